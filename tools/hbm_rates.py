"""Per-kernel HBM rates: the rocprofv3 --kernel-trace --stats summary of a bench
command (tools/rocprof_bench.sh -> <prefix>rocprof_kernel_stats.csv) joined with
the PMC traffic of the same command (tools/pmc_traffic.sh -> <prefix>pmc_traffic.json:
FETCH_SIZE x 2 + WRITE_SIZE per launch, corrected as MI355X_MICROARCH.md's HBM
section prescribes).  VERDICT r4 item 2 asks for the TB/s column; the
"everything but swconv + wgrad" total is the tail the round's target names.

  python3 tools/hbm_rates.py gpurun_out/r05_ STEPS [title]
"""
import csv
import json
import re
import sys

prefix, steps = sys.argv[1], float(sys.argv[2])
title = sys.argv[3] if len(sys.argv) > 3 else ''
rows = list(csv.DictReader(open(prefix + 'rocprof_kernel_stats.csv')))
traffic = json.load(open(prefix + 'pmc_traffic.json'))


def family(name):
  if 'swconv_kernel' in name or 'swconv_swp_kernel' in name:
    return 'swconv'
  if 'wgrad_reduce' in name or 'wgrad_flex_reduce' in name:
    return 'wgrad_reduce'
  if 'wgrad_multi' in name or 'wgrad_flex_kernel' in name:
    return 'wgrad_batched'
  if 'wgrad_kernel' in name and 'dense1' not in name and 'dense_wgrad' not in name:
    return 'wgrad_single'
  if 'at::' in name or 'rocclr' in name:
    return None
  m = re.search(r'(\w+_kernel)', name)
  return m.group(1) if m else None


agg = {}
other_ns = 0.0
total_ns = 0.0
for r in rows:
  ns, calls = float(r['TotalDurationNs']), int(r['Calls'])
  total_ns += ns
  fam = family(r['Name'])
  if fam is None:
    other_ns += ns
    continue
  a = agg.setdefault(fam, [0.0, 0])
  a[0] += ns
  a[1] += calls
mfma = ('swconv', 'wgrad_batched', 'wgrad_reduce', 'wgrad_single')
print(title or 'HBM rate per kernel (MB = 1e6 bytes; peak HBM ~8 TB/s, ~6.3 achievable)')
print('total kernel time %.2f ms/step over %g recorded steps' % (total_ns / 1e6 / steps, steps))
print('%-30s %8s %9s %10s %10s %7s %6s' % ('kernel', 'n/step', 'avg us', 'ms/step',
                                          'MB/launch', 'TB/s', 'of 8'))
tail = 0.0
for fam, (ns, calls) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
  t = traffic.get(fam)
  mb = t['hbm_bytes_per_launch'] / 1e6 if isinstance(t, dict) else None
  us = ns / calls / 1e3
  if fam not in mfma:
    tail += ns
  print('%-30s %8.1f %9.1f %10.3f %10s %7s %6s' % (
      fam[:30], calls / steps, us, ns / 1e6 / steps,
      '%.1f' % mb if mb is not None else '-',
      '%.2f' % (mb / us) if mb is not None else '-',
      '%.2f' % (mb / us / 8) if mb is not None else '-'))
tail += other_ns
print('%-30s %8s %9s %10.3f' % ('torch / runtime kernels', '', '', other_ns / 1e6 / steps))
print('tail (everything but swconv + cg_wgrad families): %.3f ms/step' % (tail / 1e6 / steps))
