#!/bin/bash
# Throughput of the other BASELINE configs through the reference's command line (rtx_cli), on the GPU box.
# C3: Random (moving spheres) 1200x800 @ 500; C4: noise scenes 1200x800 @ 500; C0: defaults.
set -o pipefail
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/sweep; mkdir -p $OUT
for s in FinalStatic Random Balls TwoSpheres BirdsEyeView ValueNoise2 ValueTurbulence ValueMarbled PerlinNoise2 PerlinNoisePink PerlinTurbulence PerlinMarbled; do
  echo "== $s" | tee -a $OUT/sweep.log
  timeout -k 10 120 ./raytracing_amd/rtx_cli -w 1200 -h 800 -s 500 -m 100 -r 1 -o $OUT/$s.ppm $s 2>&1 | tee -a $OUT/sweep.log || exit 1
done
