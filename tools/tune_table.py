#!/usr/bin/env python
"""Per-geometry table from a CALCIUMGAN_TUNE_LOG file: best classic tile vs
best software-pipelined tile (development tool)."""
import json
import sys

seen = set()
print('%-44s | %-12s %7s | %-10s %7s | ratio' % ('geometry', 'classic', 'us', 'swp', 'us'))
for l in open(sys.argv[1]):
  r = json.loads(l)
  k = tuple(r['key'])
  if k in seen:
    continue
  seen.add(k)
  items = list(r['times_us'].items())
  old = next(((c, v) for c, v in items if int(c.split(',')[0]) < 9), None)
  new = next(((c, v) for c, v in items if int(c.split(',')[0]) >= 9), None)
  if new is None or old is None:
    continue
  print('R%d t%-2d nB%-3d Lu%-4d Cx%-3d N%-3d epi%d %s| %-12s %7.1f | %-10s %7.1f | %.2f' %
        (k[0], k[1], k[2], k[5], k[4], k[6], k[9], ' ' * 6, old[0], old[1],
         new[0], new[1], new[1] / old[1]))
