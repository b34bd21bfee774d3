set -u
mkdir -p gpurun_out/hunt3
for i in 1 2 3 4 5 6; do CALCIUMGAN_AUTOTUNE=0 timeout -k 10 120 python tools/stale_graph_hunt.py memsetprobe notune_$i > gpurun_out/hunt3/notune_$i.log 2>&1; grep -h "^MEMSETPROBE" gpurun_out/hunt3/notune_$i.log | cut -c1-200; done
python bench.py > gpurun_out/r3_bench1.log 2>&1; tail -1 gpurun_out/r3_bench1.log | cut -c1-1800
python -m pytest tests -x -q -m gpu > gpurun_out/r3_gpu_all1.log 2>&1; tail -5 gpurun_out/r3_gpu_all1.log
