#!/bin/bash
# A/B on one box: the working tree's kernel library against the same library
# with the previous commit's swconv_swp.hip (put a copy at
# calciumgan_amd/csrc/_head_swconv_swp.hip first: git show HEAD:... > ...).
set -e
mkdir -p /tmp/cg_old
for f in swconv _head_swconv_swp wgrad pointwise dense_rows; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c calciumgan_amd/csrc/$f.hip -o /tmp/cg_old/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/cg_old/libcalciumgan_hip.so /tmp/cg_old/*.o
run() {
  CALCIUMGAN_HIP_LIB=$2 python bench.py --steps 30 --warmup 3 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
}
for i in 1 2 3; do
  run tree ""
  run head /tmp/cg_old/libcalciumgan_hip.so
done
