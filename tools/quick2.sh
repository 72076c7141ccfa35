#!/bin/bash
# like quick.sh, plus the frame's remaining GPU time (accumulate + resolve + gaps)
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; o=r['other_kernels']
p,b,g=o['k_wave_primary_ms_per_step'],o['k_wave_bounce_ms_per_step'],o['gpu_ms_per_step']
print('value %.0f Mrays/s  ms/step %.3f  primary %.3f  bounce %.3f  rest %.3f  gpu %.3f' % (d['value'], d['ms_per_step'], p, b, g-p-b, g))
"
