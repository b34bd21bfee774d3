# time the ablation variants of swconv_swp.hip (tools/ablate_swp.py) on a few cfg2
# geometries (GPU box): args of bench_conv.py conv = R taps nB Lx Cx N CK tile epi f32 ksteps ssq sp
export CALCIUMGAN_AUTOTUNE=0
VARIANTS=${VARIANTS:-"base noepi noloop nodma noreads nobar mfmaonly"}
GEOS=${GEOS:-"1 12 384 512 128 64 32 10 2 0 2 0 0|1 12 384 128 256 192 32 10 2 0 2 0 0|2 24 384 2048 128 64 32 10 1 0 2 0 0|2 24 128 128 256 320 32 10 2 0 2 0 0"}
echo "$GEOS" | tr "|" "\n" | while read geo; do
  echo "== $geo"
  for v in $VARIANTS; do
    CALCIUMGAN_HIP_LIB=${ABL_OUT:-tools/probe/_abl}/lib_swp_$v.so python tools/bench_conv.py conv $geo 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /"
  done
done
