import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """The shared libraries are build products (git-ignored; they travel to the GPU box with the snapshot).
    On a fresh checkout build them once; hipcc cross-compiles gfx950 without a GPU."""
    need = [
        os.path.join(ROOT, "datum_amd", "lib", "libdatum_ocean_hip.so"),
        os.path.join(ROOT, "datum_amd", "lib", "libdatum_ocean_host.so"),
        os.path.join(ROOT, "oracle", "liboracle.so"),
        os.path.join(ROOT, "tests", "cpu", "libfft_core_emul.so"),
        os.path.join(ROOT, "tests", "gpu", "libextmem_helper.so"),
    ]
    if all(os.path.exists(p) for p in need):
        return
    env = dict(os.environ)
    env["PATH"] = "/opt/rocm/bin:" + env.get("PATH", "")
    subprocess.check_call(["make", "-C", ROOT, "all", "emul", "helpers"], env=env, stdout=subprocess.DEVNULL)


_ensure_built()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.build()
    return o


_REPORT = []


@pytest.fixture(scope="session")
def report():
    """Measured parity figures (RMSE per size, against the literal- and the reduced-table oracle ...) collected by the
    GPU tests and written to gpurun_out/parity_table.txt at the end of the session; the copy under profiles/ is what
    DESIGN.md quotes."""

    def add(line):
        _REPORT.append(line)

    return add


def pytest_sessionfinish(session, exitstatus):
    if not _REPORT:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_table.txt"), "w") as f:
            f.write("# measured by pytest tests -m gpu (tests/conftest.py: report fixture); one line per check\n")
            for line in _REPORT:
                f.write(line + "\n")
    except OSError:
        pass
