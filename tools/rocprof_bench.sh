#!/bin/bash
# rocprofv3 kernel statistics of the default benchmark command, on exactly the
# tuned launches: pass 1 (no profiler) times the tile candidates and saves the
# choices, pass 2 is profiled with the choices loaded (no candidate launches).
# Run on the GPU box from the repo root; results under gpurun_out/.
export TMPDIR=/tmp
export CALCIUMGAN_TILE_CACHE=/tmp/cg_tiles.json
python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > gpurun_out/rocprof_tune.log 2>&1 || exit 1
export CALCIUMGAN_AUTOTUNE=0
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 20 --warmup 3 --no_cpu_baseline > gpurun_out/rocprof_bench.log 2>&1 || exit 1
grep "^{" gpurun_out/rocprof_bench.log > gpurun_out/rocprof_bench_line.json
# 3 warm-up + 20 timed + 20 eager instrumented steps
python3 tools/stats.py gpurun_out/prof 43 > gpurun_out/rocprof_kernel_summary.txt
cp gpurun_out/prof/*/*kernel_stats.csv gpurun_out/rocprof_kernel_stats.csv
cat gpurun_out/rocprof_kernel_summary.txt
