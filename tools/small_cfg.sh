for c in cfg1 cfg3; do for f in 2 3 4; do echo "== $c in flight $f"; timeout -k 10 200 python3 bench.py --config $c --no-cpu-baseline --no-rank-share --in-flight $f | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('one_frame_at_a_time',{})
print('%.0f Mrays/s %.4f ms (one at a time %.0f, %.4f ms)'%(d['value'],d['ms_per_step'],o.get('value',0),o.get('ms_per_step',0)))
print(d.get('kernels_ms_per_step'))" || exit 1; done; done
