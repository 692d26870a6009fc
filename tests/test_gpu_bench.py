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
    # a physical fraction: bytes moved (PMC or by design) over the longer kernel's duration over the peak
    rf = j["roofline"]
    assert 0 < rf["frac"] <= 1 and abs(rf["frac"] - rf["bytes_per_launch"] / (rf["ms_per_launch"] * 1e-3) / 8e12) < 1e-9
    # (rowpass.ms / colpass.ms are per step; a step launches either kernel once per cascade group)
    launches = j["config"]["launches_per_pass_and_step"]
    assert launches == 1 and j["config"]["cascades_per_launch"] == 2
    assert abs(rf["ms_per_launch"] * launches - max(rf["rowpass"]["ms"], rf["colpass"]["ms"])) < 1e-12 and rf["residency"] in ("infinity-cache", "hbm")


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


def test_one_rank_group_with_the_native_gather():
    # --force-collective: the N > 1 code path in a one-rank group on the real backends (RCCL process group, the module's own
    # communicator, pack + all-gather on the communication stream inside the timed region), then the same steps without the gather
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, BENCH, "--force-collective", "--steps", "6", "--warmup", "2", "--cpu-seconds", "0", "--no-frame", "--no-regime",
                        "--resolution", "512", "--cascades", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # stdout is the ONE JSON line and nothing else: RCCL's version banner (C stdio, flushed at exit, i.e. behind the line) goes to stderr
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1, r.stdout[-600:]
    j = json.loads(r.stdout)
    assert j["n_gpus"] == 1 and j["config"]["gathers_in_timed_region"] == 1 and j["config"]["payload"] == "xyz32"
    assert "datum_ocean_farm_" in j["config"]["collective_backend"] and j["config"]["payload_bytes_per_rank"] == 512 * 512 * 2 * 12
    assert j["gather_ms"] > 0 and j["value"] > 0
    assert j["without_gather"]["value"] > 0 and j["without_gather"]["ms_per_step"] > 0
    # the farm's two streams on disjoint compute units by default (datum_ocean_farm_partition: an eighth of the device for the collective)
    # (an eighth of the device in whole shares of 8 compute units: 32 of an MI355X's 256)
    import torch

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    part = j["config"]["cu_partition"]
    if cus >= 64:
        assert part and part["communication_stream_cus"] == (cus // 64) * 8 and part["communication_stream_cus"] + part["compute_stream_cus"] == cus
    else:
        assert part is None
    # the N > 1 line explains itself: every rank's own compute / collective time, its partition state, its RCCL
    d = j["farm_diagnostics"]
    assert len(d["per_rank"]) == 1 and d["per_rank"][0]["rank"] == 0 and d["per_rank"][0]["gather_ms"] > 0 and d["per_rank"][0]["compute_ms"] > 0
    assert d["per_rank"][0]["rccl_version"] > 20000 and isinstance(d["per_rank"][0]["rccl_env"], dict)
    assert d["per_rank"][0]["farm_partition"].startswith("applied" if cus >= 64 else "off")
    assert d["partition_on_every_rank"] == (cus >= 64) and d["bus_GBps"] is None          # (one rank receives nothing)


def test_the_drivers_command_line():
    # `python3 bench.py --gpus 1 --steps 20 --warmup 5` (BENCH_r03 ... r05's cmd): one line, the contract's keys, and the roofline's measurement
    # kept light inside the timed region -- two samples of each kernel there (a sampled step costs the loop ~8 us of its ~1000), every step of a
    # second pass behind it (roofline.after_region) to say that the two were representative
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert (j["n_gpus"], j["steps"], j["warmup"], j["unit"], j["scaling"], j["vs_baseline"], j["dtype"]) == (1, 20, 5, "grids/s", "weak", None, "f32")
    assert "BASELINE.json configs[2]" in j["config"]["workload"] and j["config"]["map_stores"].startswith("written through")
    rf = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] <= 1
    assert rf["launches_timed"] == 2 and rf["after_region"]["steps_timed"] == 20
    for k in ("rowpass", "colpass"):
        inside, after = rf[k]["ms"], rf["after_region"][k + "_ms"]
        assert 0.7 < inside / after < 1.4, (k, inside, after)
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert j["parity"]["displacement_rmse"]["fused_vs_reduced_table_oracle"] < 1e-5
    assert j["value"] * j["ms_per_step"] * 1e-3 == pytest.approx(4.0, rel=1e-6)      # grids per step over seconds per step
