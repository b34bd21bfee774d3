# ring-staged vs register-staged cg_wgrad on the critic's layer geometries (GPU box)
for geo in "2 24 384 2048 128 64" "2 24 384 1024 64 128" "2 24 384 512 128 192" "2 24 384 256 192 256" "2 24 384 128 256 320"; do
  for tt in 64 128; do
    for classic in 0 1; do
      python tools/bench_conv.py wgrad $geo 0 $tt 0 $classic 2>&1 | grep -v amdgpu.ids
    done
  done
done
