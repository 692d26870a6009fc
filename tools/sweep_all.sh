#!/bin/bash
# Run ON the GPU box: step timing (bench.py) and gen timing for the shipped library and every variant
# usage: tools/sweep_all.sh [bench args...]
python tools/sweep_gen.py 64 1024
DATUM_SHIPPED=1 python - "$@" <<'PY'
import glob, json, os, subprocess, sys
ROOT = os.getcwd()
libs = [("shipped", None)] + [(os.path.basename(l)[4:-3], l) for l in sorted(glob.glob("datum_amd/lib/variants/lib_*.so"))]
for name, lib in libs:
    env = dict(os.environ)
    if lib:
        env["DATUM_OCEAN_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "1000", "--warmup", "100", "--cpu-seconds", "0"] + sys.argv[1:], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "FAILED", out.stderr[-300:]); continue
    j = json.loads(line[-1]); r = j["roofline"]
    print(f"{name:24s} grids/s {j['value']:9.0f}  step {j['ms_per_step']*1e3:7.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us  col {r['colpass']['ms']*1e3:7.1f} us  step_frac {r['step_frac']:.3f}", flush=True)
PY
