#!/bin/bash
# One GPU-box call of the edit-measure loop: GPU test suite, parity fuzzers, a short bench line.  Every step under
# its own time limit; the script stops at the first failing step.  Usage: tools/gpu_round.sh <tag> [fuzz cases] [pytest -k expr]
set -o pipefail
TAG=${1:-x}; CASES=${2:-60}; KEXPR=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "$KEXPR" > $OUT/pytest.log 2>&1; RC=$?
else
  timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1; RC=$?
fi
tail -15 $OUT/pytest.log
[ $RC -eq 0 ] || { echo "pytest rc=$RC"; exit $RC; }
timeout -k 10 400 python3 -u tools/fuzz_parity.py $CASES 9000 > $OUT/fuzz.log 2>&1 || { echo fuzz failed; tail -5 $OUT/fuzz.log; exit 1; }
grep -cE "^case" $OUT/fuzz.log; grep -E "FAILED|note" $OUT/fuzz.log | cut -c1-400 | head -20
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("configs[1] kernel ms/launch", d["roofline"]["kernel_ms_per_launch"], "frac", d["roofline"]["frac"])
for c in d.get("configs",[]): print(c["config"], c["kernel_ms"], c["rays_per_sample"])
PY
