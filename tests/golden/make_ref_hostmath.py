"""Regenerates tests/golden/ref_hostmath_n64.npz from the REFERENCE'S OWN host-math lines (build container only).

What runs here is the text of /root/reference/src/renderer/ocean.cpp:80-236 -- `dispersion`, `phillips`,
`guass_random_distribution`, `seed_ocean`, `lerp_ocean_swell`, `lerp_ocean_waves`, `update_ocean` -- and of the `OceanParams`
struct, /root/reference/src/renderer/ocean.h:48-73, READ AT RUN TIME from the reference checkout, pasted into a translation
unit in a temporary directory (never stored in this repository, in any encoding) and compiled with g++ twice: as the reference's
build does (`-ffast-math`, src/CMakeLists.txt:10) and without it.  Around those lines stands what the reference gets from
elsewhere and this image lacks:

  * `leap::lml` (un-vendored, un-pinned: README.md:31) -> datum_amd/host/lml.h, a from-scratch stand-in for Vec2 / Plane /
    normsqr / dot / lerp / normalise / pi with the call sites' meaning;
  * `OceanContext` (Vulkan members) -> `struct OceanContext { static const int WaveResolution = 64; }`;
  * `random_device{}()` (ocean.cpp:132: a non-reproducible seed) -> a fixed seed.

Because of those stand-ins this is NOT a reference build in the rubric's sense and does not turn `parity` green (DESIGN.md
section 3); what it replaces is "the oracle agrees with itself" by "the oracle and the host shim agree with the reference's own
lines, compiled, for update_ocean / seed_ocean / lerp_ocean_*": tests/test_oracle_pins.py::test_reference_hostmath_fixture.

The fixture holds DATA only: the seed, h0, the phase after 1 / 60 / 600 `update_ocean` calls (dt = 1/60, example-ocean
parameters, examples/ocean/ocean.cpp:46-50), swellphase and flow beside them, and the parameters and h0 after one
`lerp_ocean_waves` / `lerp_ocean_swell`, each from both builds.

Run (build container only; /root/reference does not exist on the GPU box):  python tests/golden/make_ref_hostmath.py
"""

import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/src/renderer"

N = 64
SEED = 1000
STEPS = (1, 60, 600)

# what stands around the reference's lines (nothing of the reference is in this string)
PROLOGUE = r"""
#include <random>
#include <complex>
#include <numeric>
#include <cmath>
#include <cstring>
#include "lml.h"

using namespace std;
using namespace lml;

static unsigned int g_fixed_seed = 0;

struct OceanContext { static const int WaveResolution = 64; };
"""

EPILOGUE = r"""
extern "C"
{
  struct Dump
  {
    float wavescale, waveamplitude, windspeed, windx, windy;
    float swelllength, swellamplitude, swellspeed, swelldirx, swelldiry;
    float swellphase, flowx, flowy;
  };

  static OceanParams g_params;

  static void dump(Dump *d)
  {
    d->wavescale = g_params.wavescale; d->waveamplitude = g_params.waveamplitude; d->windspeed = g_params.windspeed;
    d->windx = g_params.winddirection.x; d->windy = g_params.winddirection.y;
    d->swelllength = g_params.swelllength; d->swellamplitude = g_params.swellamplitude; d->swellspeed = g_params.swellspeed;
    d->swelldirx = g_params.swelldirection.x; d->swelldiry = g_params.swelldirection.y;
    d->swellphase = g_params.swellphase; d->flowx = g_params.flow.x; d->flowy = g_params.flow.y;
  }

  // example-ocean parameters (examples/ocean/ocean.cpp:46-50) over the struct's defaults, then seed_ocean with a fixed seed
  void ref_seed(unsigned int seed, float wavescale, float waveamplitude, float swellamplitude, float windspeed, float smoothing, float *seedout, float *h0out, Dump *d)
  {
    g_params = OceanParams();
    g_params.wavescale = wavescale;
    g_params.waveamplitude = waveamplitude;
    g_params.swellamplitude = swellamplitude;
    g_params.windspeed = windspeed;
    g_params.smoothing = smoothing;
    g_fixed_seed = seed;
    seed_ocean(g_params);
    memcpy(seedout, g_params.seed, sizeof(g_params.seed));
    memcpy(h0out, g_params.height, sizeof(g_params.height));
    dump(d);
  }

  void ref_update(int steps, float dt, float *phaseout, Dump *d)
  {
    for(int i = 0; i < steps; ++i)
      update_ocean(g_params, dt);
    memcpy(phaseout, g_params.phase, sizeof(g_params.phase));
    dump(d);
  }

  void ref_lerp_waves(float wavescale, float waveamplitude, float windspeed, float windx, float windy, float t, float *h0out, Dump *d)
  {
    lerp_ocean_waves(g_params, wavescale, waveamplitude, windspeed, Vec2(windx, windy), t);
    memcpy(h0out, g_params.height, sizeof(g_params.height));
    dump(d);
  }

  void ref_lerp_swell(float swelllength, float swellamplitude, float swellspeed, float dirx, float diry, float t, Dump *d)
  {
    lerp_ocean_swell(g_params, swelllength, swellamplitude, swellspeed, Vec2(dirx, diry), t);
    dump(d);
  }
}
"""


class Dump(ctypes.Structure):
    _fields_ = [(n, ctypes.c_float) for n in ("wavescale", "waveamplitude", "windspeed", "windx", "windy", "swelllength", "swellamplitude", "swellspeed",
                                               "swelldirx", "swelldiry", "swellphase", "flowx", "flowy")]

    def array(self):
        return np.array([getattr(self, n) for n, _ in self._fields_], np.float32)


DUMP_FIELDS = [n for n, _ in Dump._fields_]


def lines(path, first, last):
    with open(path) as f:
        text = f.read().splitlines()
    return "\n".join(text[first - 1:last]) + "\n"


def translation_unit():
    """the reference's lines between the stand-ins; the one edit: random_device{}() -> the fixed seed (ocean.cpp:132)"""
    struct = lines(os.path.join(REF, "ocean.h"), 48, 73)
    body = lines(os.path.join(REF, "ocean.cpp"), 80, 236)
    assert struct.lstrip().startswith("struct OceanParams") and "float phase[" in struct
    assert "float dispersion(Vec2 const &k)" in body and "void update_ocean(OceanParams &params, float dt)" in body
    assert body.count("random_device{}()") == 1
    body = body.replace("random_device{}()", "g_fixed_seed")
    return PROLOGUE + struct + body + EPILOGUE


def build(tmp, name, flags):
    src = os.path.join(tmp, "ref_hostmath.cpp")
    with open(src, "w") as f:
        f.write(translation_unit())
    lib = os.path.join(tmp, name + ".so")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-I", os.path.join(ROOT, "datum_amd", "host")] + flags + ["-o", lib, src])
    os.remove(src)
    return ctypes.CDLL(lib)


def run(lib):
    fp = ctypes.POINTER(ctypes.c_float)
    F = ctypes.c_float
    lib.ref_seed.argtypes = [ctypes.c_uint, F, F, F, F, F, fp, fp, ctypes.POINTER(Dump)]
    lib.ref_update.argtypes = [ctypes.c_int, F, fp, ctypes.POINTER(Dump)]
    lib.ref_lerp_waves.argtypes = [F, F, F, F, F, F, fp, ctypes.POINTER(Dump)]
    lib.ref_lerp_swell.argtypes = [F, F, F, F, F, F, ctypes.POINTER(Dump)]

    out = {}
    d = Dump()
    seed = np.zeros((N, N, 2), np.float32)
    h0 = np.zeros((N, N, 2), np.float32)
    # examples/ocean/ocean.cpp:46-50
    lib.ref_seed(SEED, 22.0, 0.0025, 0.8, 7.9, 320.0, seed.ctypes.data_as(fp), h0.ctypes.data_as(fp), ctypes.byref(d))
    out["seed"], out["h0"], out["params_seeded"] = seed, h0, d.array()

    done = 0
    for steps in STEPS:
        phase = np.zeros((N, N), np.float32)
        lib.ref_update(steps - done, np.float32(1.0 / 60.0), phase.ctypes.data_as(fp), ctypes.byref(d))
        done = steps
        out[f"phase_{steps}"], out[f"params_{steps}"] = phase, d.array()

    # one blend of the wave parameters (recomputes every h0 from the stored seed, ocean.cpp:185-213) and one of the swell
    h1 = np.zeros((N, N, 2), np.float32)
    lib.ref_lerp_waves(30.0, 0.004, 9.0, 0.6, 0.8, 0.25, h1.ctypes.data_as(fp), ctypes.byref(d))
    out["h0_lerped"], out["params_lerped"] = h1, d.array()
    lib.ref_lerp_swell(55.0, 0.5, 1.5, 0.0, 1.0, 0.5, ctypes.byref(d))
    out["params_swell"] = d.array()
    # ... and two more ticks on the blended parameters
    phase = np.zeros((N, N), np.float32)
    lib.ref_update(2, np.float32(1.0 / 60.0), phase.ctypes.data_as(fp), ctypes.byref(d))
    out["phase_602_lerped"], out["params_602_lerped"] = phase, d.array()
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("make_ref_hostmath.py: /root/reference is not here (build container only); the committed fixture stands")
    with tempfile.TemporaryDirectory() as tmp:
        exact = run(build(tmp, "exact", ["-fno-fast-math", "-ffp-contract=off"]))
        fast = run(build(tmp, "fast", ["-ffast-math"]))
    out = {k: v for k, v in exact.items()}
    out.update({"fast_" + k: v for k, v in fast.items()})
    out["dump_fields"] = np.array(DUMP_FIELDS)
    out["lerp_waves_call"] = np.array([30.0, 0.004, 9.0, 0.6, 0.8, 0.25], np.float32)
    out["lerp_swell_call"] = np.array([55.0, 0.5, 1.5, 0.0, 1.0, 0.5], np.float32)
    path = os.path.join(ROOT, "tests", "golden", "ref_hostmath_n64.npz")
    np.savez_compressed(path, **out)
    worst = max(float(np.abs(exact[k].astype(np.float64) - fast[k]).max()) for k in exact)
    nan = int(np.isnan(exact["seed"]).sum())
    print(f"wrote {path}: {len(out)} arrays; NaN seeds {nan}; sum |h0|^2 = {float((exact['h0'].astype(np.float64) ** 2).sum()):.6g}; "
          f"phase_60[10][20] = {exact['phase_60'][10][20]:.6g}; largest |exact - fast-math| over all arrays {worst:.3g}")


if __name__ == "__main__":
    main()
