#!/bin/bash
# rocprofv3 kernel statistics of the default benchmark command, on exactly the
# tuned launches: pass 1 (no profiler) times the tile candidates and saves the
# choices, pass 2 is profiled with the choices loaded (no candidate launches).
# Run on the GPU box from the repo root; results under gpurun_out/.
# BENCH_ARGS: extra bench.py arguments (another workload), PREFIX: output name
# prefix under gpurun_out/, STEPS / WARMUP: of the profiled run.
export TMPDIR=/tmp
export CALCIUMGAN_TILE_CACHE=/tmp/cg_tiles${PREFIX:-}.json
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-3}; P=gpurun_out/${PREFIX:-}
python3 bench.py $BENCH_ARGS --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing > ${P}rocprof_tune.log 2>&1 || exit 1
export CALCIUMGAN_AUTOTUNE=0
rm -rf ${P}prof
rocprofv3 --kernel-trace --stats --output-format csv -d ${P}prof -- python3 bench.py $BENCH_ARGS --steps $STEPS --warmup $WARMUP --no_cpu_baseline > ${P}rocprof_bench.log 2>&1 || exit 1
grep "^{" ${P}rocprof_bench.log > ${P}rocprof_bench_line.json
# warm-up + timed + the eager instrumented steps
python3 tools/stats.py ${P}prof $((WARMUP + 2 * STEPS)) > ${P}rocprof_kernel_summary.txt
cp ${P}prof/*/*kernel_stats.csv ${P}rocprof_kernel_stats.csv
cat ${P}rocprof_kernel_summary.txt
