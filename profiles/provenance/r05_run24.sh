timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python - <<'PY'
import subprocess, json, os, sys
def frame(lib):
    env = dict(os.environ)
    if lib: env["DATUM_OCEAN_HIP_LIB"] = os.path.realpath(lib)
    out = subprocess.run([sys.executable, "bench.py", "--cpu-seconds", "0", "--no-regime", "--steps", "200", "--warmup", "20"], env=env, capture_output=True, text=True).stdout
    j = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    f = j["reference_frame_n64"]
    return j["value"], f["us_per_frame"], f["us_displace_only"]
for rep in range(3):
    for name, lib in (("fused step at 64^2", None), ("two launches (previous build)", "datum_amd/lib/variants/lib_prev.so")):
        v, a, b = frame(lib)
        print(f"  {name:<34s} 1024^2 x 4 {v:8.0f} grids/s   reference frame {a:6.2f} us   displace only at 64^2 {b:5.2f} us")
PY
echo "== 1024^2 x 4"; N=1024 C=4 STEPS=1000 REPS=3 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 2048^2 x 1"; N=2048 C=1 STEPS=500 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 64^2 x 1"; N=64 C=1 STEPS=5000 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
echo "== 64^2 x 4"; N=64 C=4 STEPS=5000 REPS=2 EXTRA="" tools/ab_4096.sh 2>&1 | grep -v amdgpu
