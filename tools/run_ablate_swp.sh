# time every ablation variant of swconv_swp.hip on a few geometries (GPU box)
export CALCIUMGAN_AUTOTUNE=0
for geo in "2 24 384 512 128 192 32 10 1 0 2 0 0" "2 24 384 512 128 192 32 11 1 0 2 0 0" "1 12 384 128 256 192 32 10 2 0 2 0 0" "2 24 384 2048 128 64 32 9 1 0 2 0 0"; do
  echo "== $geo"
  for v in base novm nobar nodma noreads nomfma noepi loop; do
    CALCIUMGAN_HIP_LIB=tools/probe/_abl/lib_swp_$v.so python tools/bench_conv.py conv $geo 2>&1 | grep -v amdgpu.ids | sed "s/^/$v: /"
  done
done
echo "== old kernel"
python tools/bench_conv.py conv 2 24 384 512 128 192 32 0 1 0 2 0 1 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py conv 1 12 384 128 256 192 32 0 2 0 2 0 0 2>&1 | grep -v amdgpu.ids
python tools/bench_conv.py conv 2 24 384 2048 128 64 32 0 1 0 2 0 1 2>&1 | grep -v amdgpu.ids
