#!/bin/bash
# The headline, reproducible from profiles/ (VERDICT r3 item 5): bench.py at the
# driver's 20 steps and at 400 steps on one box, rocm-smi sampled beside both
# (shader clock, socket power, temperature, busy %).  The 20-step timed region
# (0.25 s) sits inside the first second of load; the 400-step one shows what the
# chip sustains.  -> gpurun_out/bench_20_vs_400.txt
out=gpurun_out/bench_20_vs_400.txt
mkdir -p gpurun_out
export CALCIUMGAN_TILE_CACHE=/tmp/cg_b20_tiles.json
python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > /dev/null 2>&1 || exit 1
export CALCIUMGAN_AUTOTUNE=0
sample() {  # $1 = pid to follow, $2 = output file
  : > $2
  while kill -0 $1 2>/dev/null; do
    echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showuse --showpower --showclocks --showtemp 2>/dev/null | grep -E 'GPU use|Power \(W\)|sclk clock level|Temperature \(Sensor junction\)' | sed 's/GPU\[0\]\s*: //' | tr -s ' \t' ' ' | tr '\n' ';')" >> $2
    sleep 0.2
  done
}
{
  echo "tools/bench_20_vs_400.sh (commit $(cat profiles/.head_commit 2>/dev/null)): bench.py --steps N --warmup 5, one box, tiles tuned once; rocm-smi every 0.2 s"
  for steps in 20 400 20 400; do
    python3 bench.py --steps $steps --warmup 5 --no_cpu_baseline ${EXTRA_BENCH_ARGS} > gpurun_out/_b.json 2> gpurun_out/_b.err &
    pid=$!
    sample $pid gpurun_out/_b.smi
    wait $pid
    python3 - <<P
import json
d = json.loads([l for l in open('gpurun_out/_b.json') if l.startswith('{')][-1])
r = d['roofline']
print('steps %4d: %8.1f samples/s  %7.3f ms/step  swconv frac %.4f  wgrad frac %.4f  losses %s' % (
    d['steps'], d['value'], d['ms_per_step'], r['frac'], r['wgrad_kernel']['frac'],
    ' '.join('%.4g' % v for v in d['final_losses'])))
P
    grep -E "GPU use \(%\): (9[0-9]|100)" gpurun_out/_b.smi | sed 's/^/    /' | cut -c1-230
  done
} > $out
cat $out | cut -c1-230
