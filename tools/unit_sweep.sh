#!/bin/bash
# Debug: kernel time of configs[1] vs the unit-size knobs (RTX_UNIT_A / RTX_UNIT_B / RTX_TAIL_SPAN)
for cfg in "16 16 0" "24 24 0" "32 8 64" "24 8 48" "16 8 32" "48 8 128" "40 8 96" "48 8 64" "32 32 0"; do
  set -- $cfg
  RTX_UNIT_A=$1 RTX_UNIT_B=$2 RTX_TAIL_SPAN=$3 timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('A=$1 B=$2 span=$3', 'kernel_ms %.2f' % d['roofline']['kernel_ms_per_launch'], 'frac %.4f' % d['roofline']['frac'])" || exit 1
done
