#!/usr/bin/env python3
"""Condenses a tools/profile_gpu.sh output directory into a per-kernel table (mean per dispatch)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
def short(k):
    import re
    k = re.sub(r"\(ocean::GenLayout\)", "", k)       # ocean_gen_kernel<(ocean::GenLayout)0>
    k = re.sub(r"\([^()]*\)\s*$", "", k)             # the parameter list
    for a, b in (("void ocean::", ""), ("ocean::", ""), ("_kernel", "")):
        k = k.replace(a, b)
    return k[:40]
# kernel trace
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, d))
    for row in csv.DictReader(open(f)):
        if "ocean" in row.get("Name", ""):
            print(f"  {short(row['Name']):40s} calls {row['Calls']:>5s}  avg {float(row['AverageNs'])/1e3:9.2f} us  min {float(row['MinNs'])/1e3:9.2f}  max {float(row['MaxNs'])/1e3:9.2f}  {row['Percentage']}%")
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    seen = {}
    for row in csv.DictReader(open(f)):
        n = row.get("Kernel_Name", "")
        if "ocean" in n and n not in seen:
            seen[n] = row
    print("== launch shapes / resources")
    for n, row in seen.items():
        print(f"  {short(n):40s} grid {row.get('Grid_Size_X')}x{row.get('Grid_Size_Y')} wg {row.get('Workgroup_Size_X')} vgpr {row.get('VGPR_Count')} accum {row.get('Accum_VGPR_Count')} sgpr {row.get('SGPR_Count')} lds {row.get('LDS_Block_Size')} scratch {row.get('Scratch_Size')}")
# pmc
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        n = row.get("Kernel_Name", "")
        if "ocean" not in n:
            continue
        acc[short(n)][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("== PMC (mean per dispatch)")
for k, cs in acc.items():
    print(" ", k)
    for c, v in sorted(cs.items()):
        print(f"     {c:34s} {sum(v)/len(v):16.1f}   (n={len(v)})")
