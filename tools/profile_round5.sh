#!/bin/bash
# Round-5 evidence in one GPU call (results under gpurun_out/r05_*; the summaries
# to keep are copied to profiles/ by hand):
#   bench line, rocprofv3 kernel statistics, PMC traffic (per kernel), per-kernel
#   HBM rates, SQ counters, cg_swconv traffic by launch geometry, launch gaps,
#   same-box A/Bs of the round's switches, the 20-vs-400-step headline check.
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py > gpurun_out/r05_bench_n1.json 2> gpurun_out/r05_bench_n1.err || exit 1
cut -c1-200 gpurun_out/r05_bench_n1.json
PREFIX=r05_ bash tools/rocprof_bench.sh > gpurun_out/r05_rocprof.log 2>&1 || exit 1
PREFIX=r05_ bash tools/pmc_traffic.sh > gpurun_out/r05_traffic.log 2>&1 || exit 1
python3 tools/hbm_rates.py gpurun_out/r05_ 43 "cfg2, round-5 kernels: per-kernel time and HBM rate (rocprofv3 --kernel-trace --stats of python3 bench.py, tracer attached, joined with the FETCH_SIZE x 2 + WRITE_SIZE passes of the same command)" > gpurun_out/r05_hbm_rates.txt
cat gpurun_out/r05_hbm_rates.txt | cut -c1-110
python3 tools/gap_analysis.py gpurun_out/r05_prof > gpurun_out/r05_gap_analysis.txt 2>&1
bash tools/pmc_bench.sh > gpurun_out/r05_pmc_sq.log 2>&1; cp gpurun_out/pmc_sq.json gpurun_out/r05_pmc_sq.json
bash tools/traffic_by_geometry.sh > gpurun_out/r05_geo.log 2>&1
cp gpurun_out/swconv_traffic_by_geometry.txt gpurun_out/r05_swconv_traffic_by_geometry.txt
{
  echo "Same-box A/Bs of round 5's switches (tools/ab_env.sh: bench.py --steps 40 per arm, two rounds, static tiles; the final losses are printed"
  echo "beside every rate -- arms whose arithmetic is the same end on the same bits)"
  AB_STEPS=40 bash tools/ab_env.sh 2 "default:X=1" "wgrad_halves:CALCIUMGAN_WGRAD_FLEX=0" "pass_order_r4:CALCIUMGAN_SWP_CHUNK_INNER=0" "interp_separate:CALCIUMGAN_FUSE_INTERP=0" "l1_convolved:CALCIUMGAN_L1_LINEAR=0" "all_four_r4:CALCIUMGAN_WGRAD_FLEX=0 CALCIUMGAN_SWP_CHUNK_INNER=0 CALCIUMGAN_FUSE_INTERP=0 CALCIUMGAN_L1_LINEAR=0"
} > gpurun_out/r05_ab.txt 2>&1
cat gpurun_out/r05_ab.txt
bash tools/bench_20_vs_400.sh > gpurun_out/r05_b20.log 2>&1; cp gpurun_out/bench_20_vs_400.txt gpurun_out/r05_bench_20_vs_400.txt; tail -8 gpurun_out/r05_bench_20_vs_400.txt | cut -c1-200
