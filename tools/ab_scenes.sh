#!/bin/bash
# A/B kernel builds over several scenes on the GPU box: tools/ab_scenes.sh "Scene:spp ..." lib1.so lib2.so ...
SCENES=$1; shift
for sc in $SCENES; do
  name=${sc%%:*}; spp=${sc##*:}
  for v in "$@"; do
    RTX_HIP_LIB=$PWD/raytracing_amd/$v timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --scene $name --spp $spp --no-cpu-baseline --no-extra-configs --allow-lib-override 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', '$v', 'kernel_ms %.2f' % d['roofline']['kernel_ms_per_launch'])" || exit 1
  done
done
