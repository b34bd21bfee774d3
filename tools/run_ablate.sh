#!/bin/bash
# times the ablated swconv builds (tools/ablate_swconv.py) on two layers
B=tools/bench_conv.py
for v in base noepi noa nob nobar noab loop directb; do
  export CALCIUMGAN_HIP_LIB=$PWD/tools/probe/_abl/lib_$v.so
  export CALCIUMGAN_AUTOTUNE=0
  echo "== $v"
  python $B conv 2 24 384 2048 128 64 32 0 1 0 2
  python $B conv 2 24 384 2048 128 64 32 3 1 0 2
  python $B conv 1 12 384 256 192 128 32 0 0 0 4
  python $B conv 1 12 384 256 192 128 32 5 0 0 2
done
