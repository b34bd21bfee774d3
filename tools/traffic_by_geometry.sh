#!/bin/bash
# HBM traffic of every cg_swconv launch geometry: two eager
# runs of the benchmark under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE: separate
# passes), each with the library's launch log on, joined by launch order.
# Run on the GPU box from the repo root; results under gpurun_out/.
export TMPDIR=/tmp
export CALCIUMGAN_GRAPH=0
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/geo_$c gpurun_out/geo_launch_$c.log
  CALCIUMGAN_LAUNCH_LOG=gpurun_out/geo_launch_$c.log rocprofv3 --pmc $c --kernel-trace --output-format csv \
      -d gpurun_out/geo_$c -- python3 bench.py $BENCH_ARGS --steps 2 --warmup 1 --no_cpu_baseline --no_kernel_timing \
      > gpurun_out/geo_$c.log 2>&1 || exit 1
done
GEO_STEPS=3 python3 tools/traffic_by_geometry.py gpurun_out
