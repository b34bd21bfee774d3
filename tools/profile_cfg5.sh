#!/bin/bash
# BASELINE configs[4] (L 8192, 512 neurons, batch 256, mixed_float16) profiled
# like the default workload: rocprofv3 kernel statistics + HBM traffic counters
# of the same command.  Results under gpurun_out/cfg5_*; the summaries to keep
# go to profiles/r03_cfg5_*.
export BENCH_ARGS="--mixed_precision --seq_len 8192 --neurons 512 --batch 256"
export PREFIX=cfg5_
STEPS=5 WARMUP=2 bash tools/rocprof_bench.sh || exit 1
PMC_STEPS=1 bash tools/pmc_traffic.sh
