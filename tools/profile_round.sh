git_head=$(cat profiles/.head_commit)
bash tools/rocprof_bench.sh > gpurun_out/r3_rocprof.log 2>&1; tail -45 gpurun_out/rocprof_kernel_summary.txt | cut -c1-120
bash tools/pmc_bench.sh > gpurun_out/r3_pmc_sq.log 2>&1; tail -5 gpurun_out/r3_pmc_sq.log | cut -c1-300
bash tools/pmc_traffic.sh > gpurun_out/r3_pmc_traffic.log 2>&1; python - <<'P'
import json
d=json.load(open('gpurun_out/pmc_traffic.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, v['launches'], round(v['hbm_bytes_per_launch']/1e6,1),'MB')
P
