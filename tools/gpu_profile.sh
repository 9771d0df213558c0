#!/bin/bash
# Round profile recipe (run on the GPU box via gpurun): bench line, rocprofv3 kernel stats, PMC passes.
# Usage: tools/gpu_profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-extra-configs $@"
timeout -k 10 300 python3 $ROOT/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline > $OUT/stats.log 2>&1 || { echo stats failed; tail -5 $OUT/stats.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || { echo pmc fetch failed; tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || { echo pmc write failed; tail -5 $OUT/pmc_write.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o pmc -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline > $OUT/pmc_sq.log 2>&1 || { echo pmc sq failed; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o pmc -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline > $OUT/pmc_sq2.log 2>&1 || { echo pmc sq2 failed; tail -5 $OUT/pmc_sq2.log; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA -d $OUT/pmc_lds -o pmc -- python3 $ROOT/bench.py $ARGS --no-cpu-baseline > $OUT/pmc_lds.log 2>&1 || { echo pmc lds failed; tail -5 $OUT/pmc_lds.log; }
ls $OUT
