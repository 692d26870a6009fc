#!/usr/bin/env python3
"""Static instruction mix of a kernel in an assembly listing (hipcc -S --cuda-device-only).
usage: tools/count_insts.py file.s kernel-substring [vertices-or-points-per-thread]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
per = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
for m in re.finditer(r"^(\S*%s\S*):.*?\n(.*?)\n\.Lfunc_end" % re.escape(sys.argv[2]), s, re.S | re.M):
    ins = [l.split()[0] for l in m.group(2).split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = Counter(ins)
    cls = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    trans = sum(v for k, v in c.items() if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", k))
    print(f"{m.group(1)}: {len(ins)} instructions; VALU {cls('v_')} (packed {cls('v_pk')}, transcendental {trans}) = {cls('v_') / per:.0f} per unit; "
          f"SALU {cls('s_')}, buffer/global {cls('buffer') + cls('global')}, LDS {cls('ds_')}, s_nop {c['s_nop']}, s_waitcnt {c['s_waitcnt']}")
    if "-v" in sys.argv:
        print("   ", c.most_common(60))
