#!/usr/bin/env python3
"""Inter-kernel gaps of the graph-replayed steps in a rocprofv3 kernel trace.

  python3 tools/gap_analysis.py <*_kernel_trace.csv | dir with ONE run's trace>

Splits the dispatch stream into bursts at host-side pauses, takes the burst with
back-to-back launches (graph replays; an eager pass never overlaps its kernels),
and cuts them into train() steps at the step's last kernel.  Reports per STEP: wall span, sum of kernel
durations, idle time between consecutive kernels, and the idle time grouped by
the kernel that FOLLOWS the gap (its launch latency / dependency wait)."""
import collections, csv, glob, re, sys

d = sys.argv[1]
rows = []
for f in ([d] if d.endswith('.csv') else glob.glob(d + '/**/*kernel_trace.csv', recursive=True)):
  with open(f) as fh:
    for r in csv.DictReader(fh):
      rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                   r['Kernel_Name']))
rows.sort()


def short(n):
  n = re.sub(r'^void ', '', n)
  n = re.sub(r'calciumgan::|\(anonymous namespace\)::|at::native::', '', n)
  return n[:44]


# steps: maximal runs separated by a host-side pause (> 150 us without a kernel)
steps, cur = [], []
for s, e, n in rows:
  if cur and s - max(x[1] for x in cur[-4:]) > 150000:
    steps.append(cur)
    cur = []
  cur.append((s, e, n))
if cur:
  steps.append(cur)
bursts = steps


def back_to_back(b):
  """A graph replay: most launches start before their predecessor has ended
  (an eager pass with events between its kernels never does)."""
  ovl = sum(1 for i in range(1, len(b)) if b[i][0] <= b[i - 1][1])
  return len(b) >= 64 and ovl > 0.8 * len(b)


# the replayed steps: every back-to-back burst, in time order (the tracer itself
# can pause a replay for > 150 us: those pauses are reported, not counted as
# launch gaps)
replay = [b for b in bursts if back_to_back(b)]
big = [k for b in replay for k in b]
pause_ns = sum(replay[i + 1][0][0] - max(e for _, e, _ in replay[i][-4:])
               for i in range(len(replay) - 1))
names = [n for _, _, n in big]
# one step = the launches up to and including the step's LAST kernel, the one
# that gathers train()'s return values (graph branches may start in another
# order from replay to replay, so the name sequence is not strictly periodic)
marker = 'step_outputs_kernel'
if not any(marker in n for n in names):
  cnt = collections.Counter(names)
  marker = min((n for n in cnt if cnt[n] >= 8), key=lambda n: cnt[n])
sel, cur = [], []
for k in big:
  cur.append(k)
  if marker in k[2]:
    sel.append(cur)
    cur = []
sel = sel[1:]  # (the first one may hold capture / warm-up launches)
common = collections.Counter(len(x) for x in sel).most_common(1)[0][0]
print('%d bursts, %d of them back-to-back (graph replays): %d launches; cut at %s: %d steps, '
      '%d launches in most; %.3f ms of pauses > 150 us left out' %
      (len(bursts), len(replay), len(names), short(marker).split('(')[0], len(sel), common,
       pause_ns / 1e6))
tot_span = tot_busy = tot_gap = 0.0
by_next = collections.defaultdict(lambda: [0, 0.0])
hist = collections.Counter()
for st in sel:
  span = max(e for _, e, _ in st) - st[0][0]
  busy = sum(e - s for s, e, _ in st)
  tot_busy += busy
  end = st[0][1]
  for s, e, n in st[1:]:
    g = s - end
    if g > 150000:
      span -= g
    elif g > 0:
      tot_gap += g
      by_next[short(n)][0] += 1
      by_next[short(n)][1] += g
      hist[min(int(g / 1000), 20)] += 1
    else:
      hist[-1] += 1
    end = max(end, e)
  tot_span += span
n = len(sel)
print('per step: span %.3f ms, kernel time %.3f ms, idle between kernels %.3f ms'
      % (tot_span / n / 1e6, tot_busy / n / 1e6, tot_gap / n / 1e6))
print('gap histogram (us: launches per step):',
      ' '.join('%s:%.1f' % ('ovl' if k < 0 else k, v / n) for k, v in sorted(hist.items())))
print('idle time by the kernel after the gap (per step):')
for k, (c, g) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:25]:
  print('  %-46s %6.1f gaps  %7.1f us  (%.2f us each)' % (k, c / n, g / n / 1e3, g / c / 1e3))
