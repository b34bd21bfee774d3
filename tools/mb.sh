B=tools/bench_conv.py
python $B conv 1 12 384 256 192 128 64 0 0 0 4
python $B conv 1 12 384 256 192 128 64 0 0 0 2
python $B conv 1 12 384 256 192 128 32 0 0 0 2
python $B conv 2 24 384 512 128 192 32 2 1 0 2
python $B conv 2 24 384 512 128 192 32 0 1 0 2
python $B conv 2 24 384 128 256 320 32 0 1 0 0
python $B conv 2 24 384 2048 104 64 104 0 1 0 0
python $B conv 1 12 128 1024 128 102 64 0 0 0 0
python $B conv 1 1 128 2048 104 102 104 1 3 1 0
