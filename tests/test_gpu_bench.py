"""bench.py's N-rank path on ONE GPU (the driver's 8-GPU run cannot be tried by the builder): every rank on device 0
(DATUM_BENCH_DEVICES=1), the torch.distributed group over gloo (--rendezvous gloo: two ranks on one device cannot share an RCCL
communicator).  Everything but the collective runs: the launcher, the rendezvous, the per-rank grids, the barriers, the reduction
of the timings, the one JSON line; and with the gather switched on the run reaches datum_ocean_farm_init on every rank, where
RCCL refuses -- cleanly."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

pytestmark = pytest.mark.gpu


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DATUM_BENCH_DEVICES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


COMMON = ["--steps", "6", "--warmup", "2", "--rendezvous", "gloo", "--cpu-seconds", "0", "--no-frame", "--no-regime", "--resolution", "512", "--cascades", "2"]


def test_two_ranks_without_the_gather():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--gather", "none"] + COMMON, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["grids_per_step"] == 4 and j["config"]["collective_world_size"] == 2
    assert j["value"] > 0 and j["scaling"] == "weak" and j["config"]["gather"] == "none" and j["cpu_baseline"] is None
    assert j["roofline"]["rowpass"]["ms"] > 0 and j["roofline"]["colpass"]["ms"] > 0


def test_under_torch_distributed_run():
    # the driver's way of starting it
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
                        BENCH, "--gpus", "2", "--gather", "none"] + COMMON, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["value"] > 0


def test_the_gather_reaches_rccl_and_fails_cleanly():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + COMMON, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]              # no JSON line from a failed run
    assert "Duplicate GPU" in r.stderr and "datum_ocean error -6" in r.stderr and "ranks failed" in r.stderr
