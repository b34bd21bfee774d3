#!/bin/bash
# bench.py --steps 200 with rocm-smi sampling the device beside it (VERDICT r2
# hygiene 13: the default run's 0.26 s timed region is shorter than an outside
# sampler's period, so its "GPU busy" reads 0 %).  -> gpurun_out/bench_busy.txt
out=gpurun_out/bench_busy.txt
python bench.py --steps 200 --warmup 5 --no_cpu_baseline --no_kernel_timing > gpurun_out/bench_busy_line.json 2> gpurun_out/bench_busy.err &
pid=$!
: > $out.smi
while kill -0 $pid 2>/dev/null; do
  echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showuse --showpower --showclocks 2>/dev/null | grep -E 'GPU use|Average Graphics Package Power|sclk clock level' | tr -s ' ' | tr '\n' ';')" >> $out.smi
  sleep 0.2
done
wait $pid
{
  echo "tools/bench_busy.sh: python bench.py --steps 200 --warmup 5 --no_cpu_baseline --no_kernel_timing, rocm-smi sampled every 0.2 s beside it"
  echo "(commit $(cat profiles/.head_commit 2>/dev/null))"
  grep "^{" gpurun_out/bench_busy_line.json | cut -c1-400
  echo "samples (GPU use %, power, sclk):"
  cat $out.smi
} > $out
tail -25 $out | cut -c1-200
