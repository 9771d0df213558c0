#!/bin/bash
# The ray regrouping experiment: builds with -DRT_X_REGROUP=K (raytracing_amd/librtx_rgK.so,
# built beforehand, see tools/README.md) against the product build: first the images must be bit-identical on small
# renders of both list kinds (regrouping only changes which lane traces a path), then kernel times on the BASELINE workloads.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/regroup
for K in "$@"; do
  for SC in FinalStatic Random; do
    SCENE=$SC timeout -k 10 120 python3 tools/ab_image.py librtx_hip.so librtx_rg$K.so > gpurun_out/regroup/image_${SC}_rg$K.log 2>&1 || { echo "image run rg$K $SC failed"; tail -5 gpurun_out/regroup/image_${SC}_rg$K.log; exit 1; }
    grep -E "^depth|rays" gpurun_out/regroup/image_${SC}_rg$K.log | grep -v "pixels differing 0 " | grep -E "pixels differing" && { echo "rg$K $SC: images differ"; exit 1; }
    echo "rg$K $SC: images identical at depth 1, 2, 3, 50"
  done
done
LIBS="raytracing_amd/librtx_hip.so"
for K in "$@"; do LIBS="$LIBS raytracing_amd/librtx_rg$K.so"; done
tools/ab_bench.sh regroup $LIBS
