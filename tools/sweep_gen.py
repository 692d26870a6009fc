#!/usr/bin/env python3
"""Runs tools/gen_bench.py for the shipped library and every variant in datum_amd/lib/variants/ (timing only)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [("shipped", None)] + [(os.path.basename(l)[4:-3], l) for l in sorted(glob.glob(os.path.join(ROOT, "datum_amd/lib/variants/lib_*.so")))]
for name, lib in libs:
    env = dict(os.environ)
    if lib:
        env["DATUM_OCEAN_HIP_LIB"] = lib
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools/gen_bench.py")] + sys.argv[1:], env=env, capture_output=True, text=True, timeout=600)
    for l in out.stdout.splitlines():
        if l.startswith("gen "):
            print(f"{name:24s} {l}", flush=True)
    if out.returncode != 0:
        print(name, "FAILED", out.stderr[-400:])
