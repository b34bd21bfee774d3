export CALCIUMGAN_AUTOTUNE=0
for epi in 0 1 2; do
  python tools/bench_conv.py conv 1 12 384 512 128 64 32 10 $epi 0 2 0 0 2>&1 | grep -v amdgpu.ids | sed "s/^/epi$epi: /"
done
for t in 0 3 5 10; do
  python tools/bench_conv.py conv 1 12 384 512 128 64 32 $t 2 0 2 0 0 2>&1 | grep -v amdgpu.ids | sed "s/^/tile$t: /"
done
