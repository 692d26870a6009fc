"""bench.py's own launcher (`python bench.py --gpus N` with WORLD_SIZE unset) on the CPU: N child ranks, a gloo rendezvous
on 127.0.0.1, rank 0's JSON line relayed, exit codes.  --plumbing leaves the ocean out; the environment, the spawn and the
process group are the ones an N-GPU run uses (with nccl in place of gloo)."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.parametrize("world", [2, 4, 8])      # 8 = BASELINE.json configs[3]'s node
def test_own_launcher_starts_n_ranks(world):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--plumbing"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # exactly one JSON line on stdout, whatever the ranks print
    j = json.loads(lines[0])
    assert j["plumbing"] == "ok" and j["world_size"] == world and j["launcher"] == "bench.py"
    assert j["ranks_seen"] == list(range(world)) and j["sum_of_ranks"] == world * (world - 1) / 2
    # the N > 1 line's self-diagnosis (every rank's report gathered as objects, then summarised) through the calls the timed run makes
    d = j["farm_diagnostics"]
    assert [r["rank"] for r in d["per_rank"]] == list(range(world)) and all(isinstance(r["rccl_env"], dict) for r in d["per_rank"])
    assert d["slowest_rank"] == world - 1 and d["compute_ms_spread"] == [1.0, pytest.approx(1.0 + 0.1 * (world - 1))]
    assert d["bus_GBps"] == pytest.approx((world - 1) * 50331648 / 0.9e-3 / 1e9) and d["gather_hidden_under_compute"] is True
    assert d["partition_on_every_rank"] is False and d["per_rank"][1]["farm_partition"].startswith("refused")


def test_under_an_external_launcher():
    # the driver's way: torch.distributed.run sets WORLD_SIZE etc.; bench.py must then NOT start ranks of its own
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", BENCH, "--gpus", "2", "--plumbing"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["world_size"] == 2 and j["launcher"] == "external"


def test_world_size_mismatch_is_an_error_before_any_gpu_work():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--plumbing"], env=_env(WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
    assert "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


def test_a_failing_rank_fails_the_launcher():
    # rank 1 cannot parse its arguments?  No: every rank gets the same argv.  A rank that dies is simulated by an
    # environment hook only the launcher test sets.
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing"], env=_env(DATUM_BENCH_FAIL_RANK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "ranks failed" in r.stderr


def test_farm_example_ends_when_its_ranks_fail():
    # examples/ocean_farm.cpp without a GPU: rank 0 cannot make the communicator id, the others never get one; the parent reaps its ranks in
    # the order they end, ends the rest and reports -- no hang in waitpid (datum_ocean_farm_init itself has no timeout: ADVICE r04)
    exe = os.path.join(ROOT, "examples", "ocean_farm")
    if not os.path.exists(exe):
        pytest.skip("examples/ocean_farm is not built")
    import torch

    if torch.cuda.is_available():
        pytest.skip("the GPU suite runs the example for real")
    r = subprocess.run([exe, "3", "256", "1", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "FAILED" in r.stdout
