#!/bin/bash
# A/B kernel builds on the GPU box: tools/ab.sh <spp> lib1.so lib2.so ...  (libs under raytracing_amd/)
SPP=$1; shift
for v in "$@"; do
  RTX_HIP_LIB=$PWD/raytracing_amd/$v timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline --no-extra-configs --allow-lib-override 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'Gsamples/s %.3f' % (d['value']/1e9), 'frac %.4f' % d['roofline']['frac'], 'kernel_ms %.2f' % d['roofline']['kernel_ms_per_launch'])" || exit 1
done
