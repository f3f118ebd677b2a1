#!/usr/bin/env python3
"""Static instruction counts between the MZ_MARK points of one kernel (library compiled with -DMZ_MARKS -save-temps).
usage: tools/count_marks.py <file.s> <kernel name substring>"""
import re
import sys

path, pat = sys.argv[1:3]
inside = False
cur, counts, order = None, {}, []
for line in open(path):
    t = line.strip()
    if re.match(r"^_Z\w+:", t):
        inside = pat in t
        cur = None
        continue
    if not inside:
        continue
    m = re.match(r"; MZMARK (\w+)", t)
    if m:
        cur = m.group(1) + "#%d" % sum(1 for o in order if o.startswith(m.group(1) + "#"))
        order.append(cur)
        counts[cur] = dict(valu=0, salu=0, lds=0, vmem=0, other=0)
        continue
    if cur is None or not t or t[0] in ".;" or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "scratch_", "buffer_", "flat_")) else "other"
    counts[cur][kind] += 1
for o in order:
    print("%-18s" % o, " ".join("%s=%d" % kv for kv in counts[o].items()))
