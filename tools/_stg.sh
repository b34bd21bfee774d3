export CALCIUMGAN_AUTOTUNE=0
r() { python tools/bench_conv.py conv "$@" 2>&1 | tail -1; }
for s in 0 1 0 1; do export CALCIUMGAN_STAGGER=$s; echo "== stagger $s"
r 2 24 384 1024 64 128 0 14 1
r 2 24 384 512 128 192 0 14 1
r 1 12 384 256 192 128 0 14 2
r 1 12 384 512 128 64 0 14 2
r 1 12 640 256 256 192 0 14 0
done
