#!/bin/bash
# like ab.sh but prints rays and cycles per wave-bounce (128 ray slots)
SPP=$1; shift
for v in "$@"; do
  RTX_HIP_LIB=$PWD/raytracing_amd/$v timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline --allow-lib-override 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
rays=d['rays_per_sample']*d['config']['width']*d['config']['height']*d['config']['spp_per_gpu']
ms=r['kernel_ms_per_launch']
print('$v', 'Gsamples/s %.3f' % (d['value']/1e9), 'frac %.4f' % r['frac'], 'kernel_ms %.2f' % ms, 'rays/sample %.3f' % d['rays_per_sample'], 'SIMD-cycles per 128 rays %.0f' % (ms*1e-3*2.39e9*1024/(rays/128)))" || exit 1
done
