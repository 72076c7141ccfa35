#!/usr/bin/env python3
"""One-line summary of a bench.py --config line read from stdin: python bench.py --config cfg5 | python tools/cfgline.py [label]"""
import json, sys
d = json.loads(sys.stdin.read())
print(" ".join(sys.argv[1:]), d["metric"], "%.0f Mrays/s" % d["value"], "%.3f ms" % d["ms_per_step"], d["config"].get("kernel_ms"))
