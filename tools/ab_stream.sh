#!/bin/bash
# A/B on one box: the product library against the same kernels with the tiles of
# a workgroup as one pass stream (-DCG_SWP_STREAM) and against the
# previous commit's swconv_swp.hip (a copy at csrc/_head_swconv_swp.hip, if present).
set -e
NS=$(bash tools/build_variant.sh stream -DCG_SWP_STREAM | tail -1)
OLD=
if [ -f calciumgan_amd/csrc/_head_swconv_swp.hip ]; then   # (a copy of the previous commit's file)
  mkdir -p /tmp/cg_old
  for f in swconv _head_swconv_swp wgrad pointwise dense_rows; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c calciumgan_amd/csrc/$f.hip -o /tmp/cg_old/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/cg_old/libcalciumgan_hip.so /tmp/cg_old/*.o
  OLD=/tmp/cg_old/libcalciumgan_hip.so
fi
run() {
  CALCIUMGAN_HIP_LIB=$2 python bench.py --steps 30 --warmup 3 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
}
for i in 1 2; do
  run product ""
  run stream $NS
  [ -n "$OLD" ] && run head $OLD
done
