#!/bin/bash
# Build a variant of the kernel library beside the product one:
#   tools/build_variant.sh <name> [-DFLAG ...]   ->  /tmp/cg_<name>/libcalciumgan_hip.so
# (use with CALCIUMGAN_HIP_LIB=<that path>)
set -e
name=$1; shift
D=calciumgan_amd/csrc; O=${VARIANT_OUT:-/tmp}/cg_$name; mkdir -p $O
for f in swconv swconv_swp wgrad pointwise dense_rows; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $D/$f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libcalciumgan_hip.so $O/*.o
echo $O/libcalciumgan_hip.so
