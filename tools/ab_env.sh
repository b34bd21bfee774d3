#!/bin/bash
# A/B on one box, one tile table: bench.py under several environments in turn.
#   tools/ab_env.sh ROUNDS "label1:VAR=VAL VAR2=VAL" "label2:..." ...
# AB_ARGS: further bench.py arguments (another workload), AB_STEPS: timed steps.
# (the losses are printed beside the rates: an arm whose data went NaN runs at a
# higher clock -- round 4's first "15 % faster" epilogue was one)
# The first process tunes the tiles and saves its table; every timed run loads it
# (CALCIUMGAN_AUTOTUNE=0), so all arms launch the same tiles.  An arm may name
# another kernel library with CALCIUMGAN_HIP_LIB=<path>.
ROUNDS=$1; shift
export CALCIUMGAN_TILE_CACHE=${AB_TILES:-/tmp/cg_ab_tiles.json}
if [ ! -f "$CALCIUMGAN_TILE_CACHE" ]; then
  python3 bench.py $AB_ARGS --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > /dev/null 2>&1 || exit 1
fi
export CALCIUMGAN_AUTOTUNE=0
for r in $(seq 1 $ROUNDS); do
  for arm in "$@"; do
    label=${arm%%:*}; envs=${arm#*:}
    env $envs python3 bench.py $AB_ARGS --steps ${AB_STEPS:-40} --warmup 3 --no_cpu_baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('%-14s %8.1f samples/s %7.3f ms/step  swconv frac %.4f (%.1f us)  wgrad frac %.4f  losses %s' % ('$label', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r['wgrad_kernel']['frac'], ' '.join('%.4g' % v for v in d['final_losses'])))"
  done
done
