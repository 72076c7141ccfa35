#!/bin/bash
# quick A/B: prints per-kernel ms of one frame for the given bench args
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rank-share "$@" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['config'].get('kernel_ms',{})
print('value %.0f Mrays/s  ms/step %.3f [min %.3f max %.3f, %d blocks]  primary %.3f ms (%d launches)  bounce %.3f ms (%d)  gpu %.3f' % (d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'], d['timing']['blocks'], k.get('k_wave_primary',0), k.get('k_wave_primary_launches',0), k.get('k_wave_bounce',0), k.get('k_wave_bounce_launches',0), k.get('gpu',0)))
"
