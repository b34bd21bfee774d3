export CALCIUMGAN_AUTOTUNE=0
r() { python tools/bench_conv.py conv "$@" 2>&1 | tail -1; }
echo "== per-timestep Dense 512->512, 256 samples x 8192, f32 out + sigmoid (epi 3), by tile"
for t in 0 1 2 3 4 5 6 7 8; do r 1 1 256 8192 512 512 0 $t 3 1; done
echo "== bf16 out (input gradient), by tile"
for t in 0 3 5 6 7 8; do r 1 1 256 8192 512 512 0 $t 0 0; done
echo "== 1280 samples f32 out"
for t in 5 7; do r 1 1 1280 8192 512 512 0 $t 3 1; done
