#!/bin/bash
# A/B of two builds of the library on the same box: new (in-tree) vs old
for r in 1 2; do
echo "== new"; tools/mb.sh 2>&1 | grep -v amdgpu.ids
echo "== old"; CALCIUMGAN_HIP_LIB=$PWD/tools/probe/_abl/lib_old.so tools/mb.sh 2>&1 | grep -v amdgpu.ids
done
