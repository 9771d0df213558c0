#!/bin/bash
# The round's committed profiles in one GPU call: for each of the three kernels (static list, moving list, noise) run
# tools/gpu_profile.sh (bench line, rocprofv3 kernel trace, PMC passes), condense with tools/summarize_prof.py into
# gpurun_out/profiles_<tag>/ (summaries + hbm_traffic.json + valu_pmc.json, merged with the committed ones) and delete the
# raw rocprofv3 databases (tens of MB each: gpurun_out/ is capped).  Afterwards: cp gpurun_out/profiles_<tag>/* profiles/
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $DST
cp $ROOT/profiles/hbm_traffic.json $ROOT/profiles/valu_pmc.json $DST/ 2>/dev/null
run() {  # name, bench args...
  local name=$1; shift
  $ROOT/tools/gpu_profile.sh ${TAG}_$name "$@" > $DST/${TAG}_$name.log 2>&1 || { echo "profile $name failed"; tail -5 $DST/${TAG}_$name.log; return 1; }
  python3 $ROOT/tools/summarize_prof.py $ROOT/gpurun_out/prof_${TAG}_$name $DST/${TAG}_$name > /dev/null || return 1
  cp $ROOT/gpurun_out/prof_${TAG}_$name/bench.json $DST/${TAG}_${name}_bench.json
  rm -rf $ROOT/gpurun_out/prof_${TAG}_$name
  grep -E "^- " $DST/${TAG}_${name}_summary.md | cut -c1-200
}
run static && run moving --scene Random --spp 500 && run noise --scene PerlinTurbulence --spp 500 && run noise2 --scene PerlinNoise2 --spp 500
ls -la $DST
