#!/usr/bin/env python3
"""profiles/r02_traffic.json from a tools/profile_gpu.sh output directory: HBM-side bytes per launch of the two step
kernels from the FETCH_SIZE and WRITE_SIZE passes, stamped with the hash of the kernel sources it was measured on.
usage: python tools/make_traffic_json.py gpurun_out/<dir> "1024x1024 x 4 cascades" [out.json]"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

d, workload = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r03_traffic.json")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"ocean_(rowpass|colpass)_kernel<(\d+)", row.get("Kernel_Name", ""))
        if m and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            acc[f"ocean_{m.group(1)}_kernel<{m.group(2)}>"][row["Counter_Name"]].append(float(row["Counter_Value"]))
kernels = {}
for k, c in acc.items():
    fetch = sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"]))
    write = sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"]))
    kernels[k] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "launches": len(c["FETCH_SIZE"]),
                  "traffic_bytes": int(fetch * 1024 * 2 + write * 1024)}
json.dump({"comment": "HBM-side traffic per launch from rocprofv3 PMC passes (tools/profile_gpu.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc runs of "
                      "`python3 bench.py --cpu-seconds 0`). Counters are KiB. gfx950 correction from /opt/skills/guides/MI355X_MICROARCH.md (HBM section): "
                      "FETCH_SIZE tallies 128-B requests at 64 B, so reads = FETCH_SIZE x 2 (calibrated there for 16 B/lane streams); WRITE_SIZE is exact. "
                      "Infinity Cache hits are counted, not excluded.",
           "workload": workload, "kernel_source_sha256_16": bench.kernel_source_hash(), "kernels": kernels}, open(out, "w"), indent=1)
print(open(out).read())
