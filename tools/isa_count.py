#!/usr/bin/env python3
"""Static instruction mix per kernel from `hipcc -S --cuda-device-only` output (tools/isa_count.py k.s [filter])."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur, fn = None, {}
for ln in lines:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        cur = m.group(1)
        fn[cur] = collections.Counter()
        continue
    t = ln.strip()
    if cur is None or not t or t[0] in ";." or t.endswith(":"):
        continue
    op = t.split()[0]
    fn[cur][op] += 1
    if op == "s_endpgm":
        cur = None
for name, c in fn.items():
    if flt not in name:
        continue
    grp = collections.Counter()
    for k, v in c.items():
        g = "valu" if k.startswith("v_") else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else "salu/other"
        grp[g] += v
    print(name[:70], dict(grp))
    print("   ", c.most_common(18))
