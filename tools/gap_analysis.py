#!/usr/bin/env python3
"""Inter-kernel gaps of the graph-replayed steps in a rocprofv3 kernel trace.

  python3 tools/gap_analysis.py <dir with *_kernel_trace.csv>

Splits the dispatch stream into bursts at host-side pauses, takes the burst with
the most launches (the timed loop: graph replays run back to back), finds the
period of its kernel-name sequence (= the launches of ONE train() step) and cuts
the burst into steps of that length.  Reports per STEP: wall span, sum of kernel
durations, idle time between consecutive kernels, and the idle time grouped by
the kernel that FOLLOWS the gap (its launch latency / dependency wait)."""
import collections, csv, glob, re, sys

d = sys.argv[1]
rows = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
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
big = max(bursts, key=len)
names = [n for _, _, n in big]


def period_of(seq):
  """Smallest p with seq[i] == seq[i + p] for (almost) every i: one step."""
  for p in range(8, len(seq) // 2 + 1):
    same = sum(1 for i in range(len(seq) - p) if seq[i] == seq[i + p])
    if same >= 0.98 * (len(seq) - p):
      return p
  return len(seq)


period = period_of(names)
# cut from the END of the burst: its head may hold the capture / warm-up launches
n_steps = len(big) // period
big = big[len(big) - n_steps * period:]
sel = [big[i * period:(i + 1) * period] for i in range(n_steps)]
print('%d bursts; the largest has %d launches = %d steps of %d launches' %
      (len(bursts), len(names), n_steps, period))
tot_span = tot_busy = tot_gap = 0.0
by_next = collections.defaultdict(lambda: [0, 0.0])
hist = collections.Counter()
for st in sel:
  span = max(e for _, e, _ in st) - st[0][0]
  busy = sum(e - s for s, e, _ in st)
  tot_span += span
  tot_busy += busy
  end = st[0][1]
  for s, e, n in st[1:]:
    g = s - end
    if g > 0:
      tot_gap += g
      by_next[short(n)][0] += 1
      by_next[short(n)][1] += g
      hist[min(int(g / 1000), 20)] += 1
    else:
      hist[-1] += 1
    end = max(end, e)
n = len(sel)
print('per step: span %.3f ms, kernel time %.3f ms, idle between kernels %.3f ms'
      % (tot_span / n / 1e6, tot_busy / n / 1e6, tot_gap / n / 1e6))
print('gap histogram (us: launches per step):',
      ' '.join('%s:%.1f' % ('ovl' if k < 0 else k, v / n) for k, v in sorted(hist.items())))
print('idle time by the kernel after the gap (per step):')
for k, (c, g) in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:25]:
  print('  %-46s %6.1f gaps  %7.1f us  (%.2f us each)' % (k, c / n, g / n / 1e3, g / c / 1e3))
