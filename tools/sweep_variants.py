#!/usr/bin/env python3
"""Runs bench.py (no CPU leg) for every variant in datum_amd/lib/variants/; --parity adds a 1024^2 parity spot check,
--no-check lets timing-only ablation builds (wrong results by construction) through bench.py's sanity assert."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
parity = "--parity" in sys.argv
extra = [a for a in sys.argv[1:] if a != "--parity"]
for lib in sorted(glob.glob(os.path.join(ROOT, "datum_amd/lib/variants/lib_*.so"))):
    env = dict(os.environ, DATUM_OCEAN_HIP_LIB=lib)
    name = os.path.basename(lib)[4:-3]
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1000", "--warmup", "100", "--cpu-seconds", "0"] + extra,
                             env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-300:]); continue
        j = json.loads(line[-1]); r = j["roofline"]
        ok = ""
        if parity:
            chk = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(ROOT, "tests/test_gpu_parity.py"),
                                  "-k", "test_displace_end_to_end and 1024 or test_golden"], env=env, capture_output=True, text=True, timeout=600)
            ok = "parity: " + (chk.stdout.strip().splitlines()[-1] if chk.stdout.strip() else "?")
        print(f"{name:28s} grids/s {j['value']:9.0f}  step {j['ms_per_step']*1e3:7.1f} us  row {r['rowpass']['ms']*1e3:7.1f} us {r['rowpass']['GBps']:6.0f} GB/s  "
              f"col {r['colpass']['ms']*1e3:7.1f} us {r['colpass']['GBps']:6.0f} GB/s  step_frac {r['step_frac']:.3f}  {ok}", flush=True)
    except Exception as e:
        print(name, "ERROR", e)
