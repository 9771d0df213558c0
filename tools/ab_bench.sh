#!/bin/bash
# A/B kernel times of library builds on the BASELINE workloads: tools/ab_bench.sh <tag> <lib.so> [<lib.so> ...]
# (each build: configs[1] headline + the extra configs, bench.py's own HIP-event kernel times)
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for LIB in "$@"; do
  NAME=$(basename $LIB .so)
  RTX_HIP_LIB=$ROOT/$LIB timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --allow-lib-override > $OUT/$NAME.json 2> $OUT/$NAME.err || { echo "$NAME failed"; tail -5 $OUT/$NAME.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$OUT/$NAME.json"))
print("$NAME", "configs[1] %.2f" % d["roofline"]["kernel_ms_per_launch"], " ".join("%s %.2f" % (c["scene"], c["kernel_ms"]) for c in d.get("configs",[])))
PY
done
