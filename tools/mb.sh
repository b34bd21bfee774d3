#!/bin/bash
B=tools/bench_conv.py
export CALCIUMGAN_AUTOTUNE=0
python $B conv 2 24 384 2048 128 64 32 0 1 0 2
python $B conv 2 24 384 2048 128 64 32 3 1 0 2
python $B conv 1 12 384 256 192 128 32 0 0 0 4
python $B conv 1 12 384 256 192 128 32 5 0 0 2
python $B conv 2 24 384 512 128 192 32 2 1 0 2
python $B conv 1 12 128 1024 128 102 32 0 0 0 2
python $B conv 1 12 128 1024 128 102 32 0 0 1 2
python $B conv 1 1 128 2048 128 102 32 5 3 1 0
