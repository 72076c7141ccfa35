#!/bin/bash
# quick A/B: prints per-kernel ms of one frame for the given bench args
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; o=r['other_kernels']
print('value %.0f Mrays/s  ms/step %.3f  primary %.3f ms (%d launches)  bounce %.3f ms (%d)  gpu %.3f' % (d['value'], d['ms_per_step'], o['k_wave_primary_ms_per_step'], r['launches_per_step'], o['k_wave_bounce_ms_per_step'], o['k_wave_bounce_launches'], o['gpu_ms_per_step']))
"
