"""Static instruction mix of one kernel of rrt_kernels_a.o / _b.o (gfx950 code objects; OBJ=<file> picks one, default: the
bounce kernels' object for a name with "bounce" in it, else the other): python tools/isa_mix.py <name regex>"""
import collections
import re
import struct
import subprocess
import sys
import tempfile

import os
_default = "rustraytracer_amd/csrc/rrt_kernels_b.o" if len(sys.argv) > 1 and "bounce" in sys.argv[1] else "rustraytracer_amd/csrc/rrt_kernels_a.o"
obj = open(os.environ.get("OBJ", _default), "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
at = obj.find(magic)
n = struct.unpack_from("<Q", obj, at + len(magic))[0]
p = at + len(magic) + 8
co = None
for _ in range(n):
    off, size, tl = struct.unpack_from("<QQQ", obj, p)
    triple = obj[p + 24:p + 24 + tl].decode()
    p += 24 + tl
    if "gfx950" in triple:
        co = obj[at + off:at + off + size]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(co)
    f.flush()
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
pat = re.compile(sys.argv[1])
cur, mix = None, collections.Counter()
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m:
        cur = m.group(1) if pat.search(m.group(1)) else None
        continue
    if cur and line.startswith("\t"):
        op = line.split()[0]
        mix[op] += 1
tot = sum(mix.values())
print("total static instructions:", tot)
for op, c in mix.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 45):
    print("%6d  %s" % (c, op))
