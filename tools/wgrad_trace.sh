#!/bin/bash
# -DCG_WGRAD_TRACE build beside the product library + tools/wgrad_trace.py.
set -e
L=$(bash tools/build_variant.sh wtrace -DCG_WGRAD_TRACE "$@" | tail -1)
CALCIUMGAN_HIP_LIB=$L python3 tools/wgrad_trace.py
