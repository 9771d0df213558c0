#!/bin/bash
# One PMC pass (VALU instruction counts, lane utilisation, SMEM/SALU) of bench.py's headline step or of --scene X.
# Usage: tools/pmc_valu.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--allow-lib-override --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline $@"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES -d $OUT/pmc_sq -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_sq.log 2>&1 || { echo pmc failed; tail -5 $OUT/pmc_sq.log; exit 1; }
python3 - <<PY
import sqlite3
q="select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"
for k,c,n,v in sqlite3.connect("$OUT/pmc_sq/pmc_results.db").execute(q):
    if "render" in k: print(k[:70], c, n, "%.5g" % (v / n))
PY
