#!/bin/bash
# Build the kernel library with the in-kernel cycle accounting of the
# software-pipelined conv tiles (-DCG_SWP_TRACE [+ extra flags, e.g.
# -DCG_SWP_STREAM]) beside the product library and run tools/swp_trace.py on the
# cfg2 geometries.  GPU box, repo root.
set -e
L=$(bash tools/build_variant.sh trace -DCG_SWP_TRACE "$@" | tail -1)
export CALCIUMGAN_HIP_LIB=$L
echo "flags: -DCG_SWP_TRACE $@"
#                       R taps nB   Lx  Cx   N tile [creal]
python3 tools/swp_trace.py 2 24 384 2048 128  64 14 102   # critic layer 1 (narrow chunk)
python3 tools/swp_trace.py 2 24 384 1024  64 128 14       # critic layer 2
python3 tools/swp_trace.py 2 24 384  512 128 192 14       # critic layer 3
python3 tools/swp_trace.py 2 24 384  256 192 256 14       # critic layer 4
python3 tools/swp_trace.py 1 12 384  512 128  64 14       # layer-2 input gradient
python3 tools/swp_trace.py 1 12 128 1024  64 102 14       # x^ layer-1 input gradient
