#!/usr/bin/env python3
"""bench.py -- ocean grids/s (N x N displacement step) on MI355X, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N --steps K --warmup W          (no launcher: bench.py starts its own N ranks, see launch_ranks)
  python bench.py --gpus 8 --resolution 2048 --cascades 1      BASELINE.json configs[3]: 2048^2 x 8 tiles, one per GPU

A "step" = one update_ocean(dt = 1/60) + one displacement pass (phase advance, ocean.sim, row IFFT, column IFFT,
ocean.map: SURVEY.md 8d) over this rank's batch of independent cascades.  Default workload = BASELINE.json
configs[2]: 1024 x 1024 x 4 cascades per GPU (the size north_star's 70 % target is quoted on).  Cascades /
tiles are independent, so N GPUs run N batches (weak scaling, no data-path collective); north_star's
"single RCCL all-gather" that reassembles the displacement field is issued once per timed batch of K steps
and is inside the timed region (--gather none leaves it out; the JSON reports its cost separately).

Prints ONE JSON line on rank 0.  value = all grids of all ranks / max-over-ranks wall time.
Inputs: example-ocean parameters, seeds mt19937(1000 + global cascade index), synthetic by construction.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
DT = np.float32(1.0 / 60.0)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--resolution", type=int, default=1024)
    ap.add_argument("--cascades", type=int, default=4)
    ap.add_argument("--gather", choices=("pipelined", "serial", "none"), default="pipelined",
                    help="N > 1: the one all-gather per batch of --steps steps that reassembles the displacement field. pipelined: the collective "
                         "of the previous batch runs on a second stream under this batch's kernels (double-buffered payload); serial: after the "
                         "batch's last step, on the compute stream; none: left out")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="N > 1: 0 = ONE all-gather per timed batch of --steps steps (north_star's single all-gather); K >= 1: the displacement "
                         "field is packed and gathered every K steps (1 = every step), each gather overlapped with the following steps when "
                         "--gather pipelined")
    ap.add_argument("--payload", choices=("xyz32", "xyz16", "maps"), default="xyz32",
                    help="what a rank contributes to the all-gather: the displacement field as floats (12 B/pt, exact), as halves (8 B/pt), or both "
                         "map layers (32 B/pt)")
    ap.add_argument("--force-collective", action="store_true",
                    help="1 GPU only (plumbing check): take the N > 1 code path -- RCCL process group, pack, all-gather on the second stream, "
                         "all-reduce of the timings -- in a one-rank group")
    ap.add_argument("--standin-peers", type=int, default=0,
                    help="1 GPU only (overhead measurement): run the pack kernel and, on the second stream, the HBM writes of this many peers' payloads "
                         "in place of the collective")
    ap.add_argument("--standin-workgroups", type=int, default=32,
                    help="--standin-peers: workgroups of the stand-in kernel on the second stream (RCCL's channels are workgroups that copy); "
                         "0 = device-to-device copies, which occupy no compute unit")
    ap.add_argument("--comm-cus", type=int, default=-1,
                    help="compute units reserved for the communication stream (a multiple of 8: that many / 8 per XCD), the step's stream on the "
                         "others (datum_ocean_farm_partition); 0: both streams on the whole device.  Default: 32 where a collective (or its "
                         "stand-in) runs under the steps, else 0")
    ap.add_argument("--standin-gbps", type=float, default=300.0,
                    help="--standin-peers: bus bandwidth the stand-in is paced to (0 = as fast as HBM takes it)")
    ap.add_argument("--plumbing", action="store_true",
                    help="no GPU: every rank joins a gloo group on the CPU, all-reduces its rank and rank 0 prints one JSON line -- the launcher, "
                         "the environment and the rendezvous of an N-rank run without the ocean (tests/test_bench_launcher.py)")
    ap.add_argument("--map-stores", choices=("auto", "written through", "streamed"), default="auto",
                    help="datum_ocean_set_map_store_policy: how the column pass stores the maps.  auto (default): the module's rule -- written through while the "
                         "handle's working set is resident in the Infinity Cache and no multi-rank farm is initialised, streamed otherwise; with "
                         "--standin-peers (the collective's footprint without a farm) auto means streamed, as the module would choose on a real farm")
    ap.add_argument("--spectrum", choices=("fp32", "fp16", "fp16h0"), default="fp32",
                    help="storage of the work spectrum between the two passes (fp16: BASELINE.json configs[4]; arithmetic stays fp32); "
                         "fp16h0: h0 read as halves too (DATUM_OCEAN_SPECTRUM_FP16_H0: SURVEY.md 8d's own byte count for configs[4])")
    ap.add_argument("--cascade-group", type=int, default=0,
                    help="cascades per launch of the two passes (datum_ocean_set_cascade_group); 0 = the module's own choice (sized to the Infinity Cache)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg; 0 disables it")
    ap.add_argument("--no-frame", action="store_true", help="skip the 64 x 64 reference-frame timing")
    ap.add_argument("--no-check", action="store_true", help="skip the output sanity check (timing-only ablation builds)")
    ap.add_argument("--no-regime", action="store_true",
                    help="skip the beyond-Infinity-Cache leg (10 steps of 1024x1024 x 16 after the timed region: roofline.hbm_regime)")
    ap.add_argument("--python-gather", action="store_true",
                    help="N > 1: issue the all-gather from Python (datum_amd/farm.py over torch.distributed) instead of the module's own "
                         "RCCL communicator (datum_ocean_farm_*); for comparison only")
    ap.add_argument("--rendezvous", choices=("nccl", "gloo"), default="nccl",
                    help="N > 1: the backend of the torch.distributed group that carries the farm's id, the barriers and the reduction of the "
                         "timings (the all-gather itself is the module's own RCCL communicator either way).  gloo + DATUM_BENCH_DEVICES=1 rehearses "
                         "an N-rank run on ONE GPU (all ranks on device 0; --gather none, or up to RCCL's refusal of two ranks on one device)")
    ap.add_argument("--cpu-baseline-child", nargs=3, metavar=("N", "CASCADES", "SECONDS"), default=None,
                    help="internal: the cpu_baseline leg, run as a child process with pinned OpenMP threads; prints one JSON object")
    return ap.parse_args()


def _code_only(text):
    """C++ source without its comments and with every run of white space collapsed: what the compiler sees, not how it is documented"""
    import re

    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", " ", text).strip()


def kernel_source_hash():
    """sha256 over the CODE (comments and white space stripped: _code_only) of the two step kernels and their launch code (ocean.gen lives
    in its own file and is not one of the kernels the traffic file covers): a committed PMC measurement is only quoted for the build it
    was made on, and a comment edited afterwards does not orphan it."""
    import hashlib

    h = hashlib.sha256()
    for name in ("ocean_kernels.hip", "ocean_fft_core.h", "ocean_capi.hip"):
        with open(os.path.join(ROOT, "datum_amd", "csrc", name), "r") as f:
            h.update(_code_only(f.read()).encode())
    return h.hexdigest()[:16]


def measured_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command (profiles/):
    bench.py cannot run the profiler on itself, so the value is the latest committed measurement for exactly this
    kernel, workload AND kernel source (hash recorded by tools/profile_gpu.sh), or None when any of them differs."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            j = json.load(open(path))
            if j.get("workload") == workload and kernel in j["kernels"] and j.get("kernel_source_sha256_16") == kernel_source_hash():
                return j["kernels"][kernel]["traffic_bytes"], f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, KiB; same kernel sources)"
        except Exception:  # noqa: BLE001
            pass
    return None, None


def committed_parity(N):
    """Which oracle the timed path is exact against, for THIS resolution, from the newest committed table of `pytest tests -m gpu`
    (profiles/r??_parity_table.txt, written by tests/conftest.py's report fixture on the MI355X): displacement RMSE of
      * the fused kernels (what this line times) against the oracle with the reference's twiddle FORMULA at a reduced lane index
        (mathematically the reference's table, accurately evaluated): north_star's 1e-5 holds at every N;
      * the fused kernels against the oracle with the reference's LITERAL table (cos / sin at unreduced fp32 angles up to pi N,
        src/renderer/ocean.cpp:694-695): above 1e-5 from N = 512 up -- the literal table's own error;
      * the module's literal mode (datum_ocean_set_literal_transform: the reference's own radix-2 transforms and literal table on the
        GPU, a validation mode this benchmark never runs) against that literal-table oracle.
    bench.py cannot run the oracle inside the timed run (it is test infrastructure), so the figures are quoted, with their source."""
    import glob
    import re

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_table.txt")), reverse=True):
        try:
            text = open(path).read()
        except OSError:
            continue
        d = re.search(rf"^displace N=\s*{N}\s+rmse vs reduced-table oracle: disp (\S+) .*?vs literal-table oracle \(ocean\.cpp:694\): disp (\S+) .*?float64 transform: hip (\S+) literal oracle (\S+)", text, re.M)
        l = re.search(rf"^literal mode N=\s*{N}\s+rmse vs literal-table oracle: disp (\S+)", text, re.M)
        if not d:
            continue
        return {"displacement_rmse": {"fused_vs_reduced_table_oracle": float(d.group(1)), "fused_vs_literal_table_oracle": float(d.group(2)),
                                      "literal_mode_vs_literal_table_oracle": (float(l.group(1)) if l else None),
                                      "fused_dz_vs_float64_transform": float(d.group(3)), "literal_table_oracle_dz_vs_float64_transform": float(d.group(4))},
                "bar": 1e-5,
                "timed_path": "fused kernels: within 1e-5 of the reduced-table oracle at every N; against the reference's LITERAL table the distance is that table's "
                              "own error (1e-5 * N / 64 asserted); the literal mode meets 1e-5 against the literal table at every N and is never timed here",
                "source": f"profiles/{os.path.basename(path)} (pytest tests -m gpu on the MI355X)"}
    return None


def lavapipe_probe(N):
    """north_star asks for the reference's shaders on the lavapipe software-Vulkan driver as the CPU baseline.  The
    reference's own shader files do not travel (no /root/reference on the GPU box) and only support N = 64;
    tools/lavapipe/ holds this repository's N-generalised restatement of them plus a C harness.  It needs the Vulkan
    headers + loader, the lvp ICD and glslangValidator: probed here; where everything exists the harness is built and
    run (N <= 1024: one invocation per point of a line) and its line is returned, otherwise what is missing."""
    import ctypes.util
    import glob
    import shutil
    import subprocess

    loader = ctypes.util.find_library("vulkan")
    icd = sorted(glob.glob("/usr/share/vulkan/icd.d/lvp_icd*.json") + glob.glob("/etc/vulkan/icd.d/lvp_icd*.json"))
    glsl = shutil.which("glslangValidator")
    header = os.path.exists("/usr/include/vulkan/vulkan.h")
    cc = shutil.which("cc") or shutil.which("gcc")
    found = {"libvulkan": loader or None, "vulkan headers": "/usr/include/vulkan/vulkan.h" if header else None, "lvp ICD": icd[0] if icd else None,
             "glslangValidator": glsl, "cc": cc}
    missing = [n for n, v in found.items() if not v]
    if missing:
        return {"status": "unavailable: no " + ", no ".join(missing), "probe": found}
    if N > 1024:
        return {"status": "available, but the harness runs N <= 1024", "probe": found}
    d = os.path.join(ROOT, "tools", "lavapipe")
    try:
        subprocess.run(["make", "-C", d, f"N={N}"], check=True, capture_output=True, timeout=120)
        subprocess.run([sys.executable, os.path.join(d, "make_state.py"), str(N)], check=True, capture_output=True, timeout=120)
        out = subprocess.run([os.path.join(d, "harness"), str(N), "50"], cwd=d, env=dict(os.environ, VK_ICD_FILENAMES=icd[0]),
                             capture_output=True, text=True, timeout=300)
        return {"status": "this repository's N-generalised restatement of the reference shaders (tools/lavapipe): " + out.stdout.strip().replace("\n", " | "),
                "probe": found}
    except Exception as e:  # noqa: BLE001
        return {"status": f"toolchain present but the harness failed: {e}", "probe": found}


def cpu_baseline_child(N, cascades, budget):
    """Runs in a CHILD process of bench.py (never touches the GPU, imports neither torch nor datum_amd): the oracle
    (oracle/ocean_oracle.cpp, OpenMP over rows / columns) on the same workload -- seeds mt19937(1000 + cascade), wave scales
    22 / 64 / 176 / 512, example-ocean parameters -- with its threads pinned (the parent sets OMP_PROC_BIND=close,
    OMP_PLACES=cores before this process starts: an unpinned run moved between 14 and 27 grids/s from box to box).
    Three timed runs inside the budget; the best is the baseline."""
    from oracle import oracle

    scales = (22.0, 64.0, 176.0, 512.0)
    w = oracle.weights(N)
    scratch = np.empty(6 * N * N, np.float32)
    out = np.empty((2, N, N, 4), np.float32)
    states = [(oracle.seed(N, 1000 + c, wavescale=scales[c % 4])[1], scales[c % 4]) for c in range(cascades)]
    phases = [np.zeros((N, N), np.float32) for _ in range(cascades)]
    # untimed touch
    oracle.displace(states[0][0], phases[0], states[0][1], 1.35, dt=DT, w=w, mt=True, scratch=scratch, out=out)
    runs = []
    for _ in range(3):
        grids = 0
        t0 = time.perf_counter()
        while True:
            for c in range(cascades):
                oracle.displace(states[c][0], phases[c], states[c][1], 1.35, dt=DT, w=w, mt=True, scratch=scratch, out=out)
                grids += 1
            el = time.perf_counter() - t0
            if el >= budget / 3.0:
                break
        runs.append((grids / el, grids, el))
    best = max(runs)
    print(json.dumps({"value": best[0], "grids": best[1], "seconds": best[2], "runs_grids_per_s": [r[0] for r in runs],
                      "omp_max_threads": oracle.num_threads(), "cpu_count": os.cpu_count(),
                      "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                      "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "omp_places": os.environ.get("OMP_PLACES")}), flush=True)
    return 0


def cpu_baseline(N, cascades, budget):
    """The oracle timed on this host on a bounded sample of the same workload, in a child process (started with subprocess,
    no exec in this GPU-initialised process) so that its OpenMP runtime starts with pinned threads whatever this process has
    loaded.  A reported baseline, not a target.  kind = "port": the reference itself cannot be built or run here (no
    Vulkan / lavapipe / glslang / leap: DESIGN.md)."""
    import subprocess

    # the cores this process may really use: its affinity mask, capped by the cgroup's CPU quota (a GPU box shows 256 logical
    # CPUs and grants 16 of them: 256 pinned threads inside a 16-CPU quota ran at 0.66 grids/s, profiles/r04_cpu_baseline.txt)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(period)))
            cores = min(cores, quota)
    except Exception:  # noqa: BLE001
        pass
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_NUM_THREADS=str(cores))
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(N), str(cascades), str(budget)],
                             env=env, capture_output=True, text=True, timeout=max(120.0, 10.0 * budget))
        j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001
        return dict(value=None, unit="grids/s", cores=None, kind="port", lavapipe=lavapipe_probe(N), sample=f"the child process failed: {e}")
    return dict(value=j["value"], unit="grids/s", cores=j["omp_max_threads"], kind="port", lavapipe=lavapipe_probe(N),
                host={"omp_get_max_threads": j["omp_max_threads"], "os.cpu_count": j["cpu_count"], "sched_getaffinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                      "cgroup_cpu_quota": quota,
                      "OMP_PROC_BIND": j["omp_proc_bind"], "OMP_PLACES": j["omp_places"]},
                runs_grids_per_s=j["runs_grids_per_s"],
                sample=f"best of three runs: {j['grids']} grids = {j['grids'] // cascades} steps of {N}x{N} x {cascades} cascades in {j['seconds']:.1f} s, "
                       f"oracle/ocean_oracle.cpp with OpenMP over rows/columns, {j['omp_max_threads']} pinned threads (child process)")


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start N ranks of this same command as CHILD
    processes -- before torch is imported or the GPU touched in this process, and never by exec -- one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run would.  Rank 0's stdout (the
    JSON line) is relayed; everything else goes to stderr.  Exit code: 0 only if every rank exited 0."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]

    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DATUM_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    # rank 0's output is collected by a thread while every rank is watched: when one rank dies the others would sit in the
    # rendezvous or a collective for ever, so they are given a few seconds and then ended (by their PIDs)
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    failed_at = None
    while any(p.poll() is None for p in procs):
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > 10.0:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.1)

    reader.join(timeout=5.0)
    out = "".join(c for c in chunks if c)
    codes = [p.returncode for p in procs]

    # stdout carries the JSON line(s) only; what libraries print there (RCCL's and gloo's banners) goes to stderr
    for l in (out or "").splitlines():
        if l.strip():
            print(l, file=sys.stdout if l.lstrip().startswith("{") else sys.stderr, flush=True)

    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr, flush=True)
        return 1
    return 0


def rank_report(rank, device, compute_ms, gather_ms, wall_ms, without_gather_ms, partition_state, rccl_version):
    """What one rank saw (N > 1): gathered from every rank into `farm_diagnostics.per_rank` so that the first run on a real node explains itself"""
    return {"rank": rank, "device": device, "compute_ms": compute_ms, "gather_ms": gather_ms, "wall_ms": wall_ms, "without_gather_ms": without_gather_ms,
            "farm_partition": partition_state, "rccl_version": rccl_version,
            "rccl_env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_")) and k not in ("NCCL_DEBUG_FILE",)}}


def farm_diagnostics(per_rank, world, pbytes, gather_ms, compute_ms, gathering, pipelined):
    """bus_GBps: what every rank RECEIVES in one all-gather, (world - 1) payloads, over the slowest rank's collective time -- the figure DESIGN.md
    section 7's prediction assumes to be >= 270 GB/s; per_rank: each rank's own view; slowest_rank: whose compute the max-over-ranks timing is"""
    return {"per_rank": per_rank,
            "gather_ms_max": gather_ms,
            "bytes_received_per_rank": (world - 1) * pbytes if gathering else None,
            "bus_GBps": ((world - 1) * pbytes / (gather_ms * 1e-3) / 1e9 if (gathering and gather_ms > 0 and world > 1) else None),
            "gather_hidden_under_compute": (bool(gather_ms <= compute_ms) if (gathering and pipelined) else None),
            "slowest_rank": (max(per_rank, key=lambda r: r["compute_ms"])["rank"] if per_rank else None),
            "compute_ms_spread": ([min(r["compute_ms"] for r in per_rank), max(r["compute_ms"] for r in per_rank)] if per_rank else None),
            "partition_on_every_rank": (all(str(r["farm_partition"]).startswith("applied") for r in per_rank) if per_rank else None)}


def plumbing(rank, world):
    """--plumbing: the N-rank rendezvous on the CPU over gloo (no GPU, no ocean)."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t)
    g = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(g, torch.tensor([float(rank)], dtype=torch.float64))
    # the N > 1 line's per-rank reports through the same calls the timed run makes (made-up times: rank r computed 1 + r / 10 ms, gathered 0.9 ms)
    box = [None] * world
    dist.all_gather_object(box, rank_report(rank, 0, 1.0 + 0.1 * rank, 0.9, 1.2, 1.0, "applied: 32 of 256 compute units for the communication stream" if rank % 2 == 0 else "refused (32 asked): plumbing", None))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"plumbing": "ok", "backend": "gloo", "world_size": dist.get_world_size(), "sum_of_ranks": float(t.item()),
                          "ranks_seen": [int(v.item()) for v in g], "launcher": "bench.py" if os.environ.get("DATUM_BENCH_CHILD") else "external",
                          "farm_diagnostics": farm_diagnostics(box, world, 50331648, 0.9, 1.0 + 0.1 * (world - 1), True, True)}), flush=True)
    dist.destroy_process_group()
    return 0


SPECTRUM_WORDS = {"fp32": "fp32", "fp16": "fp32 arithmetic, fp16-stored spectrum", "fp16h0": "fp32 arithmetic, fp16-stored spectrum and h0"}


def baseline_config(N, C, world, spectrum):
    """Which BASELINE.json config a run is (or is the per-GPU share of)."""
    if spectrum in ("fp16", "fp16h0") and N == 4096:
        return "BASELINE.json configs[4]" + ("" if world == 1 else f", one such grid set per GPU x {world}")
    if (N, C) == (2048, 1) and world > 1:
        return f"BASELINE.json configs[3]: 2048x2048 tiles, one per GPU, {world} of its 8" if world != 8 else "BASELINE.json configs[3]: 2048x2048 x 8 tiles farmed across 8 GPUs"
    if (N, C) == (2048, 1):
        return "the per-GPU share of BASELINE.json configs[3] (one 2048x2048 tile)"
    if (N, C) == (1024, 4):
        return "BASELINE.json configs[2]" + ("" if world == 1 else f" per GPU, farmed x {world} (weak scaling of configs[2]; configs[3]'s own shape is --resolution 2048 --cascades 1)")
    if (N, C) == (512, 1):
        return "BASELINE.json configs[1]" + ("" if world == 1 else f" per GPU x {world}")
    return "no BASELINE.json config has this shape"


def main():
    args = parse()

    if args.cpu_baseline_child:
        return cpu_baseline_child(int(args.cpu_baseline_child[0]), int(args.cpu_baseline_child[1]), float(args.cpu_baseline_child[2]))

    # N ranks without a launcher: this process only starts them (nothing below runs here)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    if os.environ.get("DATUM_BENCH_DEVICES"):       # rehearsals only: N ranks over fewer devices
        local_rank %= max(1, int(os.environ["DATUM_BENCH_DEVICES"]))

    if args.gpus != world:      # before anything touches the GPU
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` (own launcher) or under "
              f"torch.distributed.run with --nproc-per-node equal to --gpus", file=sys.stderr, flush=True)
        return 2

    if os.environ.get("DATUM_BENCH_FAIL_RANK") == str(rank):      # tests/test_bench_launcher.py: a rank that dies at start
        return 3

    if args.plumbing:
        return plumbing(rank, world)

    # File descriptor 1 is kept for the ONE JSON line.  Libraries write to stdout too -- RCCL prints a version banner through C stdio when
    # a communicator is made, and in a pipe that buffer is flushed at exit, i.e. BEHIND the line -- so everything else of this process
    # goes to stderr from here on.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    multi = world > 1 or args.force_collective       # a process group exists and the collectives below are issued
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.rendezvous == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    from datum_amd import capi, farm, host_api

    N, C = args.resolution, args.cascades
    dev = torch.device("cuda", local_rank)

    # state: this rank's cascades, seeded on the host exactly as seed_ocean does (mt19937(1000 + global index))
    oc = capi.Ocean(N, C, device=local_rank)
    oc.set_spectrum_format(args.spectrum)
    oc.set_cascade_group(args.cascade_group)
    cascade_group, launches_per_pass = oc.cascade_group()
    for c, g in enumerate(farm.owned_grids(rank, world, C)):
        ws = farm.grid_wavescale(g, C)
        p = host_api.OceanParams(N, **dict(host_api.EXAMPLE_TUNABLES, wavescale=ws))
        p.seed_ocean(farm.grid_seed(g))
        oc.set_cascade(c, ws, 1.35)
        oc.upload_state(c, p.height)
        if c == 0:
            genset = p.oceanset()          # the OceanSet header of render_ocean_surface for the example camera
        del p

    # maps land in a torch tensor so that RCCL can gather them in place; kernels run on torch's stream
    maps = torch.empty(C * capi.map_block_floats(N), dtype=torch.float32, device=dev)
    oc.bind_maps(maps.data_ptr(), maps.numel() * 4)
    # a real (non-default) stream: events and RCCL below are ordered on it too (DATUM_COMPUTE_PRIORITY: tools/gather_overhead.sh, -1 = high)
    # (DATUM_COMPUTE_CUMASK: the step on a subset of the compute units, the stand-in's stream on the others: tools/gather_overhead_cumask.sh)
    if "DATUM_COMPUTE_CUMASK" in os.environ:
        from datum_amd.farm import cu_masked_stream

        stream = cu_masked_stream(dev, int(os.environ["DATUM_COMPUTE_CUMASK"], 16))
    else:
        stream = torch.cuda.Stream(dev, priority=int(os.environ["DATUM_COMPUTE_PRIORITY"])) if "DATUM_COMPUTE_PRIORITY" in os.environ else torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    oc.set_stream(stream.cuda_stream)
    # the all-gather of north_star.  N > 1 (or --force-collective): the module's own farm (datum_ocean_farm_*: RCCL communicator,
    # communication stream, double-buffered payload and event choreography inside the C ABI); torch.distributed only started the
    # ranks' rendezvous and carries the 128-byte id to them.  --standin-peers (one GPU, measurement aid) and --python-gather keep
    # the Python choreography of datum_amd/farm.py.
    gathering = (multi and args.gather != "none") or args.standin_peers > 0
    native = gathering and multi and not args.python_gather
    tg = None
    if gathering:
        code, pdtype, _ = farm.PAYLOADS[args.payload]
        pbytes = oc.payload_bytes(code)
        assert pbytes == farm.payload_bytes(N, C, args.payload)
    # the collective's workgroups and the step's on disjoint compute units (include/datum_ocean_hip.h: datum_ocean_farm_partition)
    # (default: an eighth of the device in whole shares of 8 -- 32 of an MI355X's 256)
    device_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    comm_cus = args.comm_cus if args.comm_cus >= 0 else ((device_cus // 64) * 8 if (gathering and args.gather == "pipelined" and "DATUM_COMM_CUMASK" not in os.environ and "DATUM_COMPUTE_CUMASK" not in os.environ) else 0)
    if not gathering:
        comm_cus = 0
    # (for the N > 1 diagnostics below) "applied: 32 of 256 compute units" / "refused: <the module's text>" / "off"
    partition_state = "off" if not comm_cus else None
    if native:
        box = [capi.farm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        oc.farm_init(box[0], rank, world, code, slots=2)
        farm_info = oc.farm_info()
        if comm_cus:
            # the module recreates its two streams; the steps, the events and the consumers below move to its (masked) own stream
            try:
                oc.farm_partition(comm_cus)
            except capi.OceanError as e:
                # (a runtime without CU masks: the farm works without the partition)
                print(f"bench.py: rank {rank}: datum_ocean_farm_partition({comm_cus}) refused, both streams on the whole device: {e}", file=sys.stderr)
                partition_state = f"refused ({comm_cus} asked): {e}"
                comm_cus = 0
            else:
                stream = torch.cuda.ExternalStream(oc.own_stream(), device=dev)
                torch.cuda.set_stream(stream)
                oc.set_stream(None)
    elif gathering:
        if comm_cus:
            try:
                stream = farm.cu_masked_stream(dev, ((1 << device_cus) - 1) ^ ((1 << comm_cus) - 1))
            except Exception as e:
                # (a runtime without CU masks: both streams on the whole device, as the native path falls back)
                print(f"bench.py: rank {rank}: no CU-masked stream ({e}); both streams on the whole device", file=sys.stderr)
                partition_state = f"refused ({comm_cus} asked): {e}"
                comm_cus = 0
            else:
                torch.cuda.set_stream(stream)
                oc.set_stream(stream.cuda_stream)
        tg = farm.TileGather(farm.payload_numel(N, C, args.payload), pdtype, dev, world, standin_peers=args.standin_peers,
                             force_collective=args.force_collective, standin_workgroups=args.standin_workgroups, standin_gbps=args.standin_gbps,
                             comm_cus=comm_cus)

    if partition_state is None:
        partition_state = f"applied: {comm_cus} of {device_cus} compute units for the communication stream"

    # the maps' store policy: the module's own rule covers a real farm (streamed while a communicator of several ranks exists); the one-GPU
    # stand-in has the collective's cache footprint without a communicator, so it asks for what the module would choose there
    if args.map_stores != "auto":
        oc.set_map_store_policy(args.map_stores)
    elif args.standin_peers > 0:
        oc.set_map_store_policy("streamed")
    map_policy, maps_streamed = oc.map_store_policy()
    cascade_group, launches_per_pass = oc.cascade_group()

    def step():
        oc.update(DT)
        oc.displace()

    def gather():
        """pack (compute stream, behind the last displace) + all-gather (communication stream); returns the slot, does not block"""
        if native:
            return oc.farm_gather()
        buf = tg.acquire()
        oc.pack_displacement(code, buf.data_ptr(), pbytes)
        return tg.launch()

    def await_gather(slot):
        """the compute stream waits for the slot's collective (a consumer on that stream would read the field now)"""
        if native:
            oc.farm_result(slot)
        else:
            tg.result()

    for _ in range(args.warmup):
        step()
    if gathering:
        # one untimed round of the whole choreography (RCCL sets its channels up on first use)
        await_gather(gather())

    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
        torch.cuda.synchronize(dev)

    # kernel durations for the roofline: HIP events around the two kernels of every `stride`-th step of the timed loop, on
    # the stream they run on.  A sampled step costs the loop ~6 us (the events ride on the dispatch packets, but the next
    # dispatch is not prepared under a sampled one's tail): at 20 steps every 2nd step sampled takes 6 % off the
    # throughput being measured, every 4th 3 %, every 8th 1.5 % (tools/stride_check.sh) -- about 5 samples of each kernel below
    # 64 steps (every 4th step at the driver's 20), every 8th step from 64 steps up (DATUM_BENCH_STRIDE overrides)
    # -- and every 32nd step from 512 steps up (the default 2000 steps: 63 samples of each kernel, 0.4 % instead of 1.5 %).
    # Round 6, measured at the driver's 20 steps (one box, --steps 20 --warmup 5): every step sampled 68.5 k grids/s, every 4th (five samples, the
    # rule up to here) 76.3 k, every 10th (two) 78.9 k, one sample 80.0 k -- a sampled step costs the loop ~8 us, i.e. the roofline's own
    # measurement took 4.6 % off the figure it rides on.  Below 64 steps the timed region therefore carries TWO samples of each kernel (a quarter
    # and three quarters of the way through: the first step behind the barrier starts on an idle device and is not typical), and the same kernels
    # are timed on EVERY step of a second, untimed pass right behind it (roofline.after_region), which says whether the two were representative.
    stride = int(os.environ.get("DATUM_BENCH_STRIDE", "0")) or (32 if args.steps >= 512 else (8 if args.steps >= 64 else max(1, args.steps // 2)))
    first_sample = args.steps // 4 if (args.steps < 64 and "DATUM_BENCH_STRIDE" not in os.environ) else 0
    nsamples = (args.steps - first_sample + stride - 1) // stride
    oc.profile_begin(nsamples, stride)             # (the events exist from here on; the sampling proper is switched on at step `first_sample`)
    oc.profile_end()
    ev0, ev1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    steps_done = [0]

    def timed_step():
        if steps_done[0] == first_sample:
            oc.profile_begin(nsamples, stride)     # a host-side switch: no GPU work, and the host runs several steps ahead of the device
        steps_done[0] += 1
        step()

    # The timed region holds exactly --steps steps, one pack and one all-gather.  pipelined: the field as it stands when the
    # batch begins (the previous batch's result) is packed and its all-gather runs on the communication stream while this
    # batch's kernels run on the compute stream.  serial: this batch's own field is packed and gathered after its last step and
    # the compute stream waits for it.
    # --gather-every K: a pack and a gather after every K steps instead (K = 1: the field is reassembled every step).
    every = args.gather_every if gathering else 0
    gathers = 0
    slot = None
    t0 = time.perf_counter()
    ev0.record(stream)
    if every > 0:
        for i in range(args.steps):
            timed_step()
            if (i + 1) % every == 0:
                slot = gather()
                gathers += 1
                if args.gather == "serial":
                    await_gather(slot)
        if slot is None:
            slot = gather()
            gathers += 1
    elif gathering and args.gather == "pipelined":
        slot = gather()
        gathers = 1
        # (tools/gather_overhead.sh, stand-in only: DATUM_STANDIN_CHUNKS slices of the transfer, one every steps / chunks steps)
        chunks = tg.standin_chunks if (tg is not None and getattr(tg, "standin_lib", None) is not None) else 1
        for i in range(args.steps):
            timed_step()
            if chunks > 1 and (i + 1) % max(1, args.steps // chunks) == 0:
                tg.launch_more()
    elif gathering:
        for _ in range(args.steps):
            timed_step()
        slot = gather()
        gathers = 1
        await_gather(slot)
    else:
        for _ in range(args.steps):
            timed_step()
    ev1.record(stream)
    torch.cuda.synchronize(dev)      # both streams
    if multi:
        dist.barrier()
        torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    row_ms, col_ms, nprof = oc.profile_end()

    # few samples inside the timed region (short runs): the same steps once more with every launch timed, outside the timed region
    after_region = None
    if nprof < 8 and not (multi and gathering):
        na = min(args.steps, 40)
        torch.cuda.synchronize(dev)
        oc.profile_begin(na, 1)
        for _ in range(na):
            step()
        torch.cuda.synchronize(dev)
        arow, acol, an = oc.profile_end()
        after_region = {"rowpass_ms": arow, "colpass_ms": acol, "steps_timed": an,
                        "what": f"the two kernels on every one of {an} further steps right behind the timed region (dispatch events on every launch; not part of `value`)"}

    # SURVEY.md 8e asks for both figures: with the gather (the timed region above, `value`) and the tiles' throughput without it.
    # The same K steps once more, outside the timed region, no pack and no collective in flight; max over ranks like `value`.
    plain_elapsed = 0.0
    if multi and gathering:
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        plain_elapsed = time.perf_counter() - t1

    # Beyond the Infinity Cache (outside the timed region): the headline's working set, 1024^2 x 4 = 252 MB, fits the 256 MiB
    # Infinity Cache and its kernels run above what HBM alone delivers; the same kernels over 16 cascades (1 GB) cannot.  Ten
    # steps after five, every step's two kernels timed with dispatch events, fractions on ALGORITHMIC bytes like `roofline`.
    regime = None
    if rank == 0 and world == 1 and not args.no_regime and (N, C, args.spectrum) == (1024, 4, "fp32"):
        RC = 16
        with capi.Ocean(N, RC, device=local_rank) as big:
            big.set_stream(stream.cuda_stream)
            for c in range(RC):
                ws = farm.CASCADE_WAVESCALES[c % 4]
                p = host_api.OceanParams(N, **dict(host_api.EXAMPLE_TUNABLES, wavescale=ws))
                p.seed_ocean(farm.grid_seed(c))
                big.set_cascade(c, ws, 1.35)
                big.upload_state(c, p.height)
                del p
            brow_b, bcol_b = big.algorithmic_bytes()

            def regime_leg(group):
                """ten steps after five with `group` cascades per launch of either pass (0: the module's choice)"""
                big.set_cascade_group(group)
                for _ in range(5):
                    big.update(DT)
                    big.displace()
                # throughput from ten steps with nothing between the launches; the kernels' durations from ten more with dispatch events
                # on every launch (a sampled launch costs the loop ~6 us and a step beyond the cache has two to eight launches: timed
                # together -- as up to this round's first runs -- the events took 10 % off grids_per_s: 65.9 k against 72-78 k)
                r0, r1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                r0.record(stream)
                for _ in range(10):
                    big.update(DT)
                    big.displace()
                r1.record(stream)
                big.profile_begin(10, 1)
                for _ in range(10):
                    big.update(DT)
                    big.displace()
                torch.cuda.synchronize(dev)
                brow_ms, bcol_ms, bn = big.profile_end()
                g, launches = big.cascade_group()
                return {"cascades_per_launch": g, "launches_per_pass_and_step": launches,
                        "step_frac": (brow_b + bcol_b) / ((brow_ms + bcol_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "col_frac": bcol_b / (bcol_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "row_frac": brow_b / (brow_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "rowpass_ms": brow_ms, "colpass_ms": bcol_ms, "steps_timed": bn,
                        "grids_per_s": 10 * RC / (r0.elapsed_time(r1) * 1e-3),
                        "frac_on_bytes_moved": ((48.0 + capi.map_layout(N)[3]) * N * N * RC) / ((brow_ms + bcol_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS}

            one_launch = regime_leg(RC)          # every cascade in one launch per pass (the form up to round 5)
            grouped = regime_leg(0)              # the module's cascade groups (what datum_ocean_displace does by default)
            big.set_stream(None)
        regime = dict(grouped, workload=f"{N}x{N} x {RC} cascades fp32 (1 GB working set: beyond the 256 MiB Infinity Cache), outside the timed region: grids_per_s over 10 steps after 5, "
                                        "the kernels' durations over the next 10 (dispatch events); rowpass_ms / colpass_ms are sums over a step's launches",
                      one_launch_per_pass=one_launch)

    # ocean.gen (reported separately, as vertices/s): the 1024 x 1024 projected-grid mesh of the example
    # (examples/ocean/ocean.cpp:59) from cascade 0's maps, outside the timed region above
    gen = None
    if rank == 0:
        sx = sy = 1024
        verts = torch.empty(sx * sy * 12, dtype=torch.float32, device=dev)
        g0, g1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
        for _ in range(3):
            oc.gen(0, genset, sx, sy, verts.data_ptr())
        g0.record(stream)
        for _ in range(20):
            oc.gen(0, genset, sx, sy, verts.data_ptr())
        g1.record(stream)
        torch.cuda.synchronize(dev)
        gms = g0.elapsed_time(g1) / 20
        gbytes = 48.0 * sx * sy + 32.0 * N * N
        gen = {"mesh": f"{sx}x{sy}", "ms": gms, "vertices_per_s": sx * sy / (gms * 1e-3), "GBps": gbytes / (gms * 1e-3) / 1e9,
               "bytes": gbytes, "frac_of_peak": gbytes / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, "finite": bool(torch.isfinite(verts).all())}

    # the reference's own frame: WaveResolution = 64, one cascade, update_ocean + the five dispatches with a
    # 1024 x 1024 mesh (examples/ocean/ocean.cpp:59,135,179) -- launch- and latency-bound, reported beside the headline
    frame = None
    if rank == 0 and not args.no_frame:
        p64 = host_api.OceanParams(64, **host_api.EXAMPLE_TUNABLES)
        p64.seed_ocean(1000)
        with capi.Ocean(64, 1, device=local_rank) as o64:
            o64.set_stream(stream.cuda_stream)
            o64.set_cascade(0, host_api.EXAMPLE_TUNABLES["wavescale"], 1.35)
            o64.upload_state(0, p64.height)
            set64 = p64.oceanset()
            f0, f1, f2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))

            def frame64():
                o64.update(DT)
                o64.displace()
                o64.gen(0, set64, sx, sy, verts.data_ptr())

            for _ in range(20):
                frame64()
            f0.record(stream)
            for _ in range(200):
                frame64()
            f1.record(stream)
            for _ in range(200):
                o64.update(DT)
                o64.displace()
            f2.record(stream)
            torch.cuda.synchronize(dev)
            frame = {"what": "update_ocean + sim/fftx/ffty/map at WaveResolution 64 + gen of a 1024x1024 mesh (the reference's shipped workload), back to back",
                     "us_per_frame": f0.elapsed_time(f1) / 200 * 1e3, "us_displace_only": f1.elapsed_time(f2) / 200 * 1e3,
                     "frames_per_s": 200 / (f0.elapsed_time(f1) * 1e-3)}
            o64.set_stream(None)
        del p64
    compute_ms = ev0.elapsed_time(ev1)     # the compute stream's share (serial: includes the gather it waits for)
    gather_ms = (oc.farm_wait(slot) if native else tg.last_collective_ms(slot)) if gathering else 0.0

    # N > 1: what every rank saw, so that the first run on a real node explains itself: each rank's compute and
    # collective time, whether its CU partition was applied or refused (and why), its RCCL and the channel settings in its environment
    per_rank = None
    if multi:
        mine = rank_report(rank, local_rank, compute_ms, gather_ms, elapsed * 1e3, plain_elapsed * 1e3, partition_state, (farm_info["rccl_version"] if native else None))
        box = [None] * world
        dist.all_gather_object(box, mine)
        per_rank = box
        t = torch.tensor([elapsed, compute_ms, gather_ms, plain_elapsed], dtype=torch.float64, device=(dev if args.rendezvous == "nccl" else "cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, compute_ms, gather_ms, plain_elapsed = (float(v) for v in t.tolist())

    if rank == 0:
        # sanity: the maps of the last step are finite and non-trivial
        chk = capi.map_layers(maps[: capi.map_block_floats(N)], N)
        if not args.no_check:
            assert bool(torch.isfinite(chk).all()) and float(chk[0, ..., 2].abs().max()) > 0

        grids = args.steps * C * world
        row_b, col_b = oc.algorithmic_bytes()
        # what the two kernels move by design: two packed fields instead of three and 24-byte texels (the
        # zero .w channels of the two map layers are not stored) -- 32 + 40 B/pt in fp32, 24 + 32 with the fp16-stored spectrum --
        # against the 40 + 56 (24 + 44) algorithmic bytes `achieved` is computed from
        pts = float(N) * N * C
        texel = capi.map_layout(N)[3]
        half = args.spectrum != "fp32"
        moved = (({"fp32": 32.0, "fp16": 24.0, "fp16h0": 20.0}[args.spectrum]) * pts, ((8.0 if half else 16.0) + texel) * pts)
        # the dominant kernel: strictly the longer of the two (round 5; round 4's tie-break towards the column pass picked the
        # kernel with the larger figure: ADVICE r04)
        dom = ("colpass", col_ms, col_b, moved[1]) if col_ms > row_ms else ("rowpass", row_ms, row_b, moved[0])
        step_ach = (row_b + col_b) / ((row_ms + col_ms) * 1e-3) / 1e9 if row_ms + col_ms > 0 else 0.0

        kernel_name = f"ocean_{dom[0]}_kernel<{N}>"
        traffic, traffic_source = measured_traffic(kernel_name, f"{N}x{N} x {C} cascades") if args.spectrum == "fp32" else (None, None)
        # roofline.frac is a PHYSICAL fraction (<= 1 by construction): the bytes that cross the L2's memory side per launch -- the PMC
        # figure for this kernel, workload and kernel source when profiles/ holds one, otherwise the bytes the kernel moves by design
        # (the two agree within 3 % wherever both exist) -- over the launch duration measured here over 8 TB/s.  The figure on SURVEY
        # 8d's ALGORITHMIC bytes (which the kernels do not all move) is kept beside it as frac_on_survey_bytes.
        phys_bytes = float(traffic) if traffic is not None else dom[3]
        ach = phys_bytes / (dom[1] * 1e-3) / 1e9 if dom[1] > 0 else 0.0
        survey_ach = dom[2] / (dom[1] * 1e-3) / 1e9 if dom[1] > 0 else 0.0
        # h0 8 (4 as halves) + phase 4 + work spectrum 16 (8 as halves) + maps: what one step touches
        working_set = (({"fp32": 28.0, "fp16": 20.0, "fp16h0": 16.0}[args.spectrum]) + texel) * pts
        residency = "infinity-cache" if working_set < 256 * 2**20 else "hbm"

        line = {
            "metric": "ocean grids/sec (N x N displacement step)",
            "value": grids / elapsed,
            "unit": "grids/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{N}x{N} x {C} cascades per GPU, {SPECTRUM_WORDS[args.spectrum]}"
                            f", phase advance + sim + row IFFT + column IFFT + map: {baseline_config(N, C, world, args.spectrum)}",
                "baseline_config": baseline_config(N, C, world, args.spectrum),
                "resolution": N,
                "cascades_per_gpu": C,
                "cascades_per_launch": cascade_group,
                "map_stores": ("streamed (nt)" if maps_streamed else "written through") + f" [policy: {map_policy}]",
                "launches_per_pass_and_step": launches_per_pass,
                "grids_per_step": C * world,
                "gather": ((f"{args.gather}: one all-gather of the {args.payload} payload every {every} step(s), {gathers} inside the timed region"
                            + (", each on a second stream under the following steps' kernels" if args.gather == "pipelined" else "")) if every > 0 else
                           (f"{args.gather}: one all-gather of the {args.payload} payload per {args.steps} steps, inside the timed region"
                            + (" (the previous batch's, on a second stream under this batch's kernels)" if args.gather == "pipelined" else "")))
                          if (multi and args.gather != "none") else ("none" if multi else
                          (f"n/a (1 GPU; stand-in for {args.standin_peers} peers' payloads on a second stream: "
                           + (f"{args.standin_workgroups} workgroups paced to {args.standin_gbps:g} GB/s" if args.standin_workgroups else "device-to-device copies")
                           + (f", every {every} step(s)" if every > 0 else f", once per {args.steps} steps") + ")" if args.standin_peers else "n/a (1 GPU)")),
                "gathers_in_timed_region": gathers,
                "collective_world_size": (dist.get_world_size() if multi else None),
                "rendezvous_backend": (args.rendezvous if multi else None),
                "collective_backend": (None if not (multi and gathering) else (f"RCCL {farm_info['rccl_version']} through the module's C ABI (datum_ocean_farm_*), {farm_info['slots']} slots" if native
                                                                              else "RCCL through torch.distributed (datum_amd/farm.py)")),
                "measured_on_hardware": ("this line" if world > 1 else "1 GPU"),
                "cu_partition": ({"communication_stream_cus": comm_cus, "compute_stream_cus": device_cus - comm_cus} if comm_cus else None),
                "payload": args.payload if gathering else None,
                "payload_bytes_per_rank": pbytes if gathering else None,
                "parallelism": f"tile-farm x{world}",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": ach,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "frac_basis": ("rocprofv3 PMC bytes per launch (traffic)" if traffic is not None else "bytes moved by design (no committed PMC pass for these kernel sources)"),
                "residency": residency,
                "working_set_bytes": working_set,
                "achieved_on_survey_bytes": survey_ach,
                "frac_on_survey_bytes": survey_ach / HBM_PEAK_GBS,
                "frac_hbm_regime_on_bytes_moved": (regime["frac_on_bytes_moved"] if regime else None),
                "note": ("frac = the dominant (longer) kernel's bytes across the L2's memory side per launch (PMC FETCH_SIZE x 2 + WRITE_SIZE where "
                         "profiles/ holds a pass of these kernel sources, else the bytes moved by design) / ms_per_launch / 8 TB/s. "
                         "frac_on_survey_bytes is the same duration on SURVEY.md 8d's ALGORITHMIC bytes (96 B/pt fp32, 68 fp16-stored; this "
                         "kernel's share in bytes_per_launch); it can pass 1.0 because the kernels move fewer bytes than the algorithm as the "
                         "reference states it -- two packed transforms instead of three (16 instead of 24 B/pt between the passes) and 24-byte "
                         "texels (the constant-zero .w channels are not stored). residency says where the working set lives: at 1024^2 x 4 "
                         "(218 MB) inside the 256 MiB Infinity Cache, so frac is a fabric rate there; frac_hbm_regime_on_bytes_moved is the same "
                         "two kernels over 16 cascades (1 GB), out of HBM."),
                "hbm_bytes_by_design": {"rowpass": moved[0], "colpass": moved[1]},
                "frac_of_peak_on_bytes_moved": {"rowpass": moved[0] / (row_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if row_ms > 0 else None,
                                                "colpass": moved[1] / (col_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if col_ms > 0 else None,
                                                "step": (moved[0] + moved[1]) / ((row_ms + col_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS if row_ms + col_ms > 0 else None},
                # (a step launches either kernel once per cascade group: config.launches_per_pass_and_step; rowpass.ms / colpass.ms below are per STEP)
                "bytes_per_launch": phys_bytes / launches_per_pass,
                "survey_bytes_per_launch": dom[2] / launches_per_pass,
                "ms_per_launch": dom[1] / launches_per_pass,
                "rowpass": {"ms": row_ms, "bytes": row_b, "GBps": row_b / (row_ms * 1e-3) / 1e9 if row_ms > 0 else 0.0},
                "colpass": {"ms": col_ms, "bytes": col_b, "GBps": col_b / (col_ms * 1e-3) / 1e9 if col_ms > 0 else 0.0},
                "step_GBps": step_ach,
                "step_frac": step_ach / HBM_PEAK_GBS,
                "launches_timed": nprof,
                "after_region": after_region,
                "hbm_regime": regime,
            },
            "parity": committed_parity(N),
            "compute_ms": compute_ms,
            "gather_ms": gather_ms,
            # N > 1 only.  bus_GBps: what every rank RECEIVES in one all-gather, (world - 1) payloads, over the slowest rank's collective
            # time -- the figure DESIGN.md section 7's prediction assumes to be >= 270 GB/s; per_rank: each rank's own view
            "farm_diagnostics": (farm_diagnostics(per_rank, world, pbytes if gathering else 0, gather_ms, compute_ms, gathering, args.gather == "pipelined") if multi else None),
            "gen": gen,
            "reference_frame_n64": frame,
            "value_compute_only": grids / (compute_ms * 1e-3) if (compute_ms > 0 and args.gather != "serial") else None,
            "without_gather": ({"value": grids / plain_elapsed, "unit": "grids/s", "ms_per_step": plain_elapsed * 1e3 / args.steps,
                                "what": f"the same {args.steps} steps once more after the timed region with no pack and no collective in flight, timed the same way "
                                        "(barriers, max over ranks): the tiles' own throughput (SURVEY.md 8e (i)); `value` is the figure WITH the gather (8e (ii))"
                                        + (f"; still on the {device_cus - comm_cus} compute units the partition leaves the step (config.cu_partition)" if comm_cus else "")}
                               if plain_elapsed > 0 else None),
        }

        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(N, C, args.cpu_seconds)
        else:
            line["cpu_baseline"] = None

        os.write(line_fd, (json.dumps(line) + "\n").encode())

    if tg is not None:
        tg.close()
    oc.bind_maps(0, 0)
    oc.set_stream(None)
    # (with the CU partition torch's current stream is the module's own stream, which dies with the handle)
    torch.cuda.synchronize(dev)
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
    farm.release_cu_masked_stream(stream)        # (a no-op for any other stream)
    oc.close()

    if multi:
        dist.destroy_process_group()

    return 0


if __name__ == "__main__":
    sys.exit(main())
