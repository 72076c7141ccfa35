#!/usr/bin/env python3
"""Static per-loop-depth instruction mix of one kernel in rrt_kernels.s (make asm):
   python tools/asm_blocks.py <mangled-name-substring>
lane = v_readlane/v_writelane (SGPR spill traffic, each a VALU slot)."""
import re, sys, collections
s = open('/root/repo/rustraytracer_amd/csrc/rrt_kernels.s').read()
pat = sys.argv[1]
m = re.search(r'^(\S*%s\S*):' % re.escape(pat), s, re.M)
i = m.start(); j = s.index('.Lfunc_end', i)
depth = 0
tot = collections.defaultdict(lambda: collections.Counter())
for l in s[i:j].split('\n'):
    t = l.strip()
    mm = re.match(r'^\.LBB\d+_\d+:(.*)', t)
    if mm:
        d = re.search(r'Depth=(\d+)', mm.group(1))
        depth = int(d.group(1)) if d else 0
        continue
    if not t or t.startswith(('.', ';')): continue
    op = t.split()[0]
    c = tot[depth]
    c['n'] += 1
    if op in ('v_readlane_b32', 'v_writelane_b32'): c['lane'] += 1
    elif op.startswith('s_nop'): c['nop'] += 1
    elif op.startswith('scratch_'): c['scratch'] += 1
    elif op.startswith('v_'): c['valu'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    else: c['mem'] += 1
print(m.group(1))
for d in sorted(tot): print('depth', d, dict(tot[d]))
