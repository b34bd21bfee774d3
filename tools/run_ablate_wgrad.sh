# time the wgrad.hip ring-loop ablation variants on critic layer geometries (GPU box)
VARIANTS=${VARIANTS:-"base noreads nomfma nodma nobar nodmabar noloop mfmaonly"}
for geo in "2 24 384 2048 128 64 0 128" "2 24 384 256 192 256 0 128"; do
  echo "== $geo"
  for v in $VARIANTS; do
    CALCIUMGAN_HIP_LIB=${ABL_OUT:-tools/probe/_abl}/lib_wg_$v.so python tools/bench_conv.py wgrad $geo 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /"
  done
done
