"""VGPR / SGPR counts and spills of the trace kernels: parses the gfx950 code objects out of rrt_kernels_{a,b}.o.
   python tools/spills.py [regex]"""
import re
import struct
import subprocess
import sys
import tempfile

import os
# the library's kernels are two objects (RRT_TU 0 / 1); OBJ=/tmp/rrt_kernels_<variant>.o for a variant build
OBJS = os.environ.get("OBJ", "rustraytracer_amd/csrc/rrt_kernels_a.o,rustraytracer_amd/csrc/rrt_kernels_b.o").split(",")
if len(OBJS) > 1:
    for o in OBJS:
        subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, OBJ=o))
    sys.exit(0)
obj = open(OBJS[0], "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
at = obj.find(magic)
if at < 0:
    sys.exit("no uncompressed offload bundle in the object (compressed?)")
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
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else r"k_wave_(primary|bounce)ILi1ELb[01]ELb1ELb0|k_accumulate")
for blk in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if not pat.search(name):
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print("%-60s vgpr %3s spill %3s  sgpr %3s spill %3s  scratch %s B" % (name[:60], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size")))
