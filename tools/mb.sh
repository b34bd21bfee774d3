B=tools/bench_conv.py
for small in 0 2; do for ks in 0 2 4; do
python $B conv 2 24 384 512 128 192 32 $small 1 0 $ks
python $B conv 2 24 384 512 128 192 64 $small 1 0 $ks
done; done
for small in 0 2; do for ks in 2 4; do
python $B conv 1 12 384 256 192 128 64 $small 0 0 $ks
python $B conv 2 24 384 2048 104 64 104 $small 1 0 $ks
done; done
