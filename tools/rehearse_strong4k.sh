#!/bin/bash
# REHEARSAL of bench.py's multi-rank strong-scaling mode (BASELINE configs[4]: one image, samples partitioned over the
# ranks, one sum-reduce onto rank 0) on a box with ONE GPU: four ranks over gloo that share GPU 0, reduced size.  It
# exercises the code path the driver launches on an 8-GPU node (partition, render per rank, reduce, max-over-ranks
# timing, rank 0's JSON line) -- it is NOT a measurement, and bench.py marks its line accordingly ("data": "...REHEARSAL").
# Usage: tools/rehearse_strong4k.sh [ranks] [width] [height] [spp]
RANKS=${1:-4}; W=${2:-480}; H=${3:-270}; SPP=${4:-67}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
RTX_BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $RANKS --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $RANKS --mode strong4k --width $W --height $H --spp $SPP --steps 2 --warmup 1
