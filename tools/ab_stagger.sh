#!/bin/bash
# A/B on one box: start stagger of the persistent conv workgroups (product)
# against none and against staggering only the launches in which a workgroup
# runs one tile / more than one tile.  bench.py twice per variant.
set -e
run() {
  CALCIUMGAN_HIP_LIB=$2 python bench.py --steps 30 --warmup 3 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4))"
}
P4='((bx&1)+2*((bx>>8)&1))'
A=$(bash tools/build_variant.sh none -DCG_SWP_NO_STAGGER | tail -1)
B=$(bash tools/build_variant.sh single "-DCG_SWP_STAGGER=($P4*5*(pa.ntl<=(int)gridDim.x))" | tail -1)
C=$(bash tools/build_variant.sh multi "-DCG_SWP_STAGGER=($P4*5*(pa.ntl>(int)gridDim.x))" | tail -1)
for i in 1 2; do
  run product ""
  run none $A
  run single_tile_launches_only $B
  run multi_tile_launches_only $C
done
