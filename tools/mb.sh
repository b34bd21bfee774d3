B=tools/bench_conv.py
for t in 64 128; do
python $B wgrad 2 24 384 512 128 192 0 $t
python $B wgrad 2 24 384 2048 104 64 0 $t
python $B wgrad 2 24 384 256 192 256 0 $t
python $B wgrad 2 24 384 1024 64 128 0 $t
done
python $B wgrad 2 24 384 128 256 320 0 64
