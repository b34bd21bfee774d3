B=tools/bench_conv.py
for ns in 0 8 16 21 24 32; do
python $B wgrad 2 24 384 512 128 192 $ns 128
done
for ns in 0 32 64; do
python $B wgrad 2 24 384 2048 104 64 $ns 128
done
for ns in 0 8 16 24; do
python $B wgrad 2 24 384 256 192 256 $ns 128
done
for ns in 0 8 32 64; do
python $B wgrad 2 24 384 1024 64 128 $ns 128
done
for ns in 0 8 16; do
python $B wgrad 2 24 384 128 256 320 $ns 64
done
