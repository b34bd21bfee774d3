export CALCIUMGAN_AUTOTUNE=0
r() { python tools/bench_conv.py conv "$@" 2>&1 | tail -1; }
echo "== D L3 fwd (2.25 rounds)"; r 2 24 384 512 128 192 0 14 1; r 2 24 341 512 128 192 0 14 1; for t in 14 13 10 15 12; do r 2 24 43 512 128 192 0 $t 1; done
echo "== D L4 fwd (1.5 rounds)"; r 2 24 384 256 192 256 0 14 1; r 2 24 256 256 192 256 0 14 1; for t in 14 13 15 12; do r 2 24 128 256 192 256 0 $t 1; done
echo "== dgrad 128->256 rows N192 (2.25)"; r 1 12 384 128 256 192 0 14 2; r 1 12 341 128 256 192 0 14 2; for t in 14 13 15; do r 1 12 43 128 256 192 0 $t 2; done
echo "== dgrad Lu64 N256 (1.5)"; r 1 12 384 64 320 256 0 14 2; r 1 12 256 64 320 256 0 14 2; for t in 14 13 15; do r 1 12 128 64 320 256 0 $t 2; done
