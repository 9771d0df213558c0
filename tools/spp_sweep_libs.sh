#!/bin/bash
# kernel time at a few sample counts for several library builds in one GPU call: tools/spp_sweep_libs.sh <spps> <lib.so>...
SPPS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for LIB in "$@"; do
  echo "== $LIB"
  RTX_HIP_LIB=$ROOT/raytracing_amd/$LIB SPPS=$SPPS timeout -k 10 200 python3 $ROOT/tools/spp_sweep.py 2>&1 | grep spp || exit 1
done
