"""HBM traffic of the cg_swconv launches, bucketed by launch geometry (VERDICT r4
item 3a).  Inputs (written by tools/traffic_by_geometry.sh on the GPU box):

  gpurun_out/geo_launch_<COUNTER>.log      one line per cg_swconv launch, in launch
                                           order (CALCIUMGAN_LAUNCH_LOG)
  gpurun_out/geo_<COUNTER>/*/*counter_collection.csv   rocprofv3 --pmc <COUNTER>
                                           of the same eager run

for COUNTER in FETCH_SIZE, WRITE_SIZE (separate passes).  The i-th swconv-family
dispatch of the run is the i-th logged launch (eager launches, one stream).  Per
geometry: launches per step, algorithmic bytes (source once, packed weights once,
output once, the mask / pre-activation a fused epilogue reads or writes once --
bench.py's algorithmic_bytes_swconv convention), measured bytes (FETCH_SIZE x 2
+ WRITE_SIZE, corrected as MI355X_MICROARCH.md's HBM section prescribes for
gfx950) and the excess, with a reading of where the excess comes from computed
from the tile shape:

  w_reread   packed weights re-fetched by every row tile beyond the first that
             does not find them in its XCD's L2 (upper bound: all row tiles)
  halo       window rows two neighbouring row tiles both stage (taps/stride - 1
             rows per tile boundary inside a sample)
  pitch      bytes of padded channels (102 of 128) that are algorithmically dead
             but sit in the same 128-byte lines (already inside `algorithmic`,
             listed for scale)
"""
import collections
import csv
import glob
import json
import os
import re
import sys

OUT = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'
STEPS = float(os.environ.get('GEO_STEPS', '3'))  # warm-up + timed steps of the profiled run

TILE_ROWS = {9: 512, 10: 256, 11: 256, 12: 128, 13: 128, 14: 256, 15: 128}
TILE_COLS = {9: 64, 10: 64, 11: 128, 12: 128, 13: 64, 14: 64, 15: 128}


def parse_log(path):
  out = []
  for line in open(path):
    if not line.startswith('swconv '):
      continue
    out.append({k: int(v) for k, v in (kv.split('=') for kv in line.split()[1:])})
  return out


def algorithmic(g):
  src = g['nB'] * g['Lx'] * g['Cx'] * 2
  w = g['nphase'] * g['taps'] * g['Cx'] * g['N'] * 2
  out = g['nB'] * g['Ly'] * g['Cy'] * (4 if g['f32'] else 2)
  extra = 0
  if g['mask']:
    extra += g['nB'] * g['Ly'] * g['Cy'] * 2
  if g['ln'] == 2:  # fused LayerNorm with the backward's copies
    extra += g['nB'] * g['Ly'] * (g['Cy'] * 2 + 8)
  if g['ksplit'] > 1:  # f32 partial sums written, read by the finishing launch
    out = g['ksplit'] * g['nB'] * g['Ly'] * g['Cy'] * 4
  return src, w, out + extra


def counter_rows(counter):
  fs = sorted(glob.glob('%s/geo_%s/*/*counter_collection.csv' % (OUT, counter)),
              key=os.path.getmtime)
  rows = [r for r in csv.DictReader(open(fs[-1])) if r['Counter_Name'] == counter]
  rows.sort(key=lambda r: int(r['Dispatch_Id']))
  main = [r for r in rows
          if re.search(r'swconv_(swp_)?kernel', r['Kernel_Name'])]
  return main


def main():
  per = {}
  for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    log = parse_log('%s/geo_launch_%s.log' % (OUT, counter))
    rows = counter_rows(counter)
    if len(log) != len(rows):
      print('warning: %d logged launches vs %d swconv dispatches (%s)' % (
          len(log), len(rows), counter))
    for g, r in zip(log, rows):
      key = tuple(sorted(g.items()))
      d = per.setdefault(key, {'FETCH_SIZE': [0, 0.0], 'WRITE_SIZE': [0, 0.0],
                               'kernel': r['Kernel_Name']})
      d[counter][0] += 1
      d[counter][1] += float(r['Counter_Value'])
  table = []
  for key, d in per.items():
    g = dict(key)
    n = d['FETCH_SIZE'][0]
    if not n or not d['WRITE_SIZE'][0]:
      continue
    fetch = 2 * d['FETCH_SIZE'][1] / n * 1024
    write = d['WRITE_SIZE'][1] / d['WRITE_SIZE'][0] * 1024
    src, w, out = algorithmic(g)
    tm = TILE_ROWS.get(g['tile'], 256)
    tn = TILE_COLS.get(g['tile'], 64)
    row_tiles = g['nB'] * max(1, g['Lu'] // tm) if g['Lu'] >= tm else (g['nB'] * g['Lu'] + tm - 1) // tm
    col_tiles = (g['N'] + tn - 1) // tn
    # a row tile's window: stride * tm + taps - stride rows; neighbours overlap
    halo_rows = g['taps'] - g['stride'] if g['Lu'] > tm else 0
    halo = g['nB'] * max(0, g['Lu'] // tm - 1) * halo_rows * g['Cx'] * 2 * g['nphase']
    # each column tile re-reads the window unless it shares the XCD's L2
    win_reread = (col_tiles - 1) * src
    w_all_tiles = (row_tiles - 1) * w
    table.append(dict(
        geometry=g, kernel=re.sub(r'\(.*', '', d['kernel'])[:60],
        launches_per_step=n / STEPS, algorithmic=src + w + out,
        src=src, weights=w, out=out, fetch=fetch, write=write,
        measured=fetch + write, ratio=(fetch + write) / (src + w + out),
        excess_read=fetch - (src + w + (out if False else 0)) -
        (g['nB'] * g['Ly'] * g['Cy'] * 2 if g['mask'] else 0),
        excess_write=write - (out - (g['nB'] * g['Ly'] * g['Cy'] * 2 if g['mask'] else 0)),
        bound_halo=halo, bound_window_reread=win_reread,
        bound_weights_every_tile=w_all_tiles, row_tiles=row_tiles,
        col_tiles=col_tiles))
  table.sort(key=lambda t: -(t['measured'] - t['algorithmic']) * t['launches_per_step'])
  tot_m = sum(t['measured'] * t['launches_per_step'] for t in table)
  tot_a = sum(t['algorithmic'] * t['launches_per_step'] for t in table)
  nl = sum(t['launches_per_step'] for t in table)
  lines = []
  lines.append('cg_swconv HBM traffic by launch geometry (per train() step; MB = 1e6 bytes)')
  lines.append('total: %.0f launches/step, measured %.1f MB/launch vs algorithmic %.1f MB/launch = %.2f x; '
               'excess %.0f MB/step' % (nl, tot_m / nl / 1e6, tot_a / nl / 1e6, tot_m / tot_a,
                                         (tot_m - tot_a) / 1e6))
  lines.append('%-58s %5s %8s %8s %6s | %8s %8s | %7s %7s %7s' % (
      'geometry (stride taps nB Lx Cx->N Ly, tile, flags)', 'n/st', 'alg MB', 'meas MB', 'ratio',
      'xs read', 'xs write', 'halo', 'win x', 'w x'))
  for t in table:
    g = t['geometry']
    flags = ''.join(f for f, on in (('M', g['mask']), ('S', g['shifts']), ('O', g['oshifts']),
                                    ('L', g['ln']), ('Q', g['ssq']), ('R', g['rscale']),
                                    ('N', g['narrow']), ('F', g['f32'])) if on)
    name = 's%d t%-2d nB%-4d Lx%-5d %3d->%-3d Ly%-5d ph%d tile%-2d k%d %s' % (
        g['stride'], g['taps'], g['nB'], g['Lx'], g['Cx'], g['N'], g['Ly'], g['nphase'],
        g['tile'], g['ksplit'], flags)
    lines.append('%-58s %5.1f %8.1f %8.1f %6.2f | %8.1f %8.1f | %7.1f %7.1f %7.1f' % (
        name, t['launches_per_step'], t['algorithmic'] / 1e6, t['measured'] / 1e6, t['ratio'],
        t['excess_read'] / 1e6, t['excess_write'] / 1e6, t['bound_halo'] / 1e6,
        t['bound_window_reread'] / 1e6, t['bound_weights_every_tile'] / 1e6))
  lines.append('xs read / xs write: measured minus algorithmic reads / writes per launch; halo, win x, w x: '
               'upper bounds of window rows staged twice by neighbouring row tiles, of the source re-read by '
               'every further column tile, and of the weights re-read by every further row tile')
  text = '\n'.join(lines)
  print(text)
  open('%s/swconv_traffic_by_geometry.txt' % OUT, 'w').write(text + '\n')
  json.dump(table, open('%s/swconv_traffic_by_geometry.json' % OUT, 'w'), indent=1)


if __name__ == '__main__':
  main()
