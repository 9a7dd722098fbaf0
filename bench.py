#!/usr/bin/env python3
"""bench.py -- MLUPS of the fused D2Q9-BGK timestep on MI355X, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid NXxNY] [--math exact|fast]

One "step" is one lattice timestep (accelerate_flow + propagate + rebound + collision +
av_velocity, /root/reference/SerialCode/d2q9-bgk.c:166-170) over the whole grid.  The default
workload is BASELINE.json's HBM-roofline configuration: the synthetic 8192x8192 grid whose
obstacle map is the reference's 1024x1024 map tiled 8x8, uniform-equilibrium start (no RNG).
For N > 1 (launched by torch.distributed.run, one rank per GPU) the SAME grid is row-partitioned
over the ranks (strong scaling), halo rows travelling by RCCL send/recv inside the engine.

Timing: W untimed warm-up steps, then an untimed pre-warm until the device has been busy for
LBM_BENCH_PREWARM_S (0.3 s: a cold GPU clocks up over the first ~0.1 s, which made a 20-step run
15 % slower than a 200-step one), then the K-step timed region -- barrier + device sync on both
sides, max over ranks -- LBM_BENCH_REPEATS (5) times back to back; `value`, `ms_per_step` and the
roofline come from the MEDIAN repeat, every repeat is listed in `repeats_ms_per_step`.

Launching: `python bench.py --gpus N` started WITHOUT a torch.distributed environment (no WORLD_SIZE) launches
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`
itself, as a CHILD process, before torch or HIP is touched in this process (never exec), relays the child's
single JSON line and exit status, and kills the child's process group if it outlives LBM_BENCH_LAUNCH_TIMEOUT
(default 1500 s) -- then it exits non-zero.  Started by torch.distributed.run (WORLD_SIZE set) it is a rank.
LBM_BENCH_SELF_LAUNCH=1 takes the child path for --gpus 1 too (rehearsal of the multi-GPU launch on one GPU).
LBM_BENCH_REHEARSAL=hosted runs N > 1 ranks on however few GPUs there are: the ranks share devices, their halo rows
travel through the host over gloo (lbm_create_rank_hosted_tiled) instead of RCCL -- every line of the multi-rank path of
this file (launch, agreement on step counts, max-over-ranks timing, the bitwise self-check, exit statuses) except the
RCCL transport itself; its numbers mean nothing and the line says so.

Rank 0 prints ONE JSON line; see the task contract for its keys.  `rccl` in it says which RCCL the engine bound
(library path, version) and how many ranks its communicator counts -- "did RCCL see N ranks, and which RCCL".  Roofline block: `achieved` is the
COMPULSORY traffic of one launch of the dominant kernel -- the slab's lattice read once and written
once, 72 B per cell, however many timesteps the launch advances -- divided by the launch's device
time (HIP events on the engine's compute stream), so `frac` = achieved / 8 TB/s is a true fraction
of the HBM bound (<= 1).  The BASELINE metric's bandwidth (72 B per lattice UPDATE) is reported
beside it as `algorithmic_GBps`: with k timesteps per pass over memory it is k x `achieved` and may
exceed the peak.  `limiter` names what the PMC profile of the same kernel geometry says binds it.
"""
import argparse
import importlib.util
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
BYTES_PER_UPDATE = 72.0
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def load_package():
    spec = importlib.util.spec_from_file_location(
        "lbm_asynchronous_amd", os.path.join(ROOT, "lbm-asynchronous_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["lbm_asynchronous_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def synthetic_case(lbm, nx, ny, steps):
    """params `nx ny iters 10 0.1 0.01 1.85`, obstacles = the reference's
    1024x1024 map tiled; for the reference's own four grids the matching input files are used."""
    inputs = os.path.join(ROOT, "tests", "golden", "inputs")
    own = os.path.join(inputs, f"input_{nx}x{ny}.params")
    if os.path.exists(own):
        p = lbm.read_params(own)
        ob = lbm.read_obstacles(os.path.join(inputs, f"obstacles_{nx}x{ny}.dat"), nx, ny)
        p.max_iters = steps
        return p, ob, f"reference data set {nx}x{ny}", False
    # the engine expands the tile on the device (lbm_create_tiled): no nx*ny int map on the host
    tile = lbm.read_obstacles(os.path.join(inputs, "obstacles_1024x1024.dat"), 1024, 1024)
    p = lbm.Params(nx, ny, steps, 10, 0.1, 0.01, 1.85)
    return p, tile, f"synthetic {nx}x{ny}: 1024x1024 obstacle map tiled {nx // 1024}x{ny // 1024}", True


def cpu_baseline(nx, ny, budget_s=20.0):
    """Time the CPU oracle (kind 'port': oracle/lbm_oracle_cli, single thread, four-sweep AoS form
    of SerialCode) on a bounded sample of the same workload.  Test infrastructure used as the
    reported baseline only -- never as the thing measured above."""
    cli = os.path.join(ROOT, "oracle", "lbm_oracle_cli")
    if not os.path.exists(cli):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "lbm_oracle_cli"],
                       capture_output=True)
    if not os.path.exists(cli):
        return None
    # ~30 MLUPS per core for the serial form: pick a step count that fits the budget
    steps = int(max(2, min(200, budget_s * 30e6 / (nx * ny))))
    inputs = os.path.join(ROOT, "tests", "golden", "inputs")
    env = dict(os.environ, LBM_OUTPUT="none", OMP_NUM_THREADS="1")
    with tempfile.TemporaryDirectory() as tmp:
        pf = os.path.join(tmp, "in.params")
        with open(pf, "w") as fh:
            fh.write(f"{nx}\n{ny}\n{steps}\n10\n0.1\n0.01\n1.85\n")
        if os.path.exists(os.path.join(inputs, f"obstacles_{nx}x{ny}.dat")):
            obf = os.path.join(inputs, f"obstacles_{nx}x{ny}.dat")
        else:
            obf = os.path.join(inputs, "obstacles_1024x1024.dat")
            env["LBM_TILE"] = "1024x1024"
        out = subprocess.run([cli, pf, obf], cwd=tmp, env=env, capture_output=True, text=True)
    if out.returncode != 0:
        return None
    secs = None
    for line in out.stdout.splitlines():
        if line.startswith("Elapsed Compute time"):
            secs = float(line.split()[-2])
    if not secs:
        return None
    return {"value": nx * ny * steps / secs / 1e6, "unit": "MLUPS", "cores": 1, "kind": "port",
            "sample": f"{steps} steps of the same {nx}x{ny} grid, oracle/lbm_oracle_cli "
                      f"(four-sweep AoS restatement of SerialCode, gcc -O3, 1 thread), "
                      f"{secs:.2f} s compute"}


def cpu_reference(budget_steps=300):
    """The reference program itself (oracle/_ref/d2q9-bgk-serial-portable: SerialCode/d2q9-bgk.c
    compiled unmodified by oracle/Makefile in the builder container) timed on this host on the
    reference's own 1024x1024 data set with a shortened iteration count.  Extra information beside
    cpu_baseline: the reference binary always writes its 91 MB final_state.dat, so it cannot be run
    on the 8192x8192 workload (5.7 GB of text)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "d2q9-bgk-serial-portable")
    if not os.path.exists(exe):
        return None
    inputs = os.path.join(ROOT, "tests", "golden", "inputs")
    with tempfile.TemporaryDirectory() as tmp:
        pf = os.path.join(tmp, "in.params")
        with open(pf, "w") as fh:
            fh.write(f"1024\n1024\n{budget_steps}\n10\n0.1\n0.01\n1.85\n")
        try:
            out = subprocess.run([exe, pf, os.path.join(inputs, "obstacles_1024x1024.dat")], cwd=tmp,
                                 capture_output=True, text=True, timeout=120,
                                 env=dict(os.environ, OMP_NUM_THREADS="1"))
        except Exception:
            return None
    if out.returncode != 0:
        return None
    for line in out.stdout.splitlines():
        if line.startswith("Elapsed Compute time"):
            secs = float(line.split()[-2])
            return {"value": 1024 * 1024 * budget_steps / secs / 1e6, "unit": "MLUPS", "cores": 1,
                    "kind": "reference",
                    "sample": f"{budget_steps} steps of the reference's 1024x1024 data set, "
                              f"oracle/_ref/d2q9-bgk-serial-portable, {secs:.2f} s compute"}
    return None


def cpu_baseline_multicore(nx, ny, threads=16, budget_s=6.0):
    """The reference's own fastest shared-memory formulation (OpenMP/d2q9-bgk.c:334: fused two-lattice SoA
    sweep, `omp parallel for` + reduction) as restated in the oracle, on `threads` cores with the reference's
    binding (OpenMP/env.sh:2-4: OMP_PROC_BIND=true OMP_PLACES=cores).  16 = the host-CPU share of a 1-GPU job
    on this pool.  Reported beside the 1-core figure; test infrastructure, never the thing measured above."""
    cli = os.path.join(ROOT, "oracle", "lbm_oracle_cli")
    if not os.path.exists(cli):
        return None
    threads = max(1, min(threads, os.cpu_count() or 1))
    steps = int(max(4, min(400, budget_s * 450e6 / (nx * ny))))
    inputs = os.path.join(ROOT, "tests", "golden", "inputs")
    env = dict(os.environ, LBM_OUTPUT="none", LBM_ORACLE_FORM="fused", OMP_NUM_THREADS=str(threads),
               OMP_PROC_BIND="true", OMP_PLACES="cores")
    with tempfile.TemporaryDirectory() as tmp:
        pf = os.path.join(tmp, "in.params")
        with open(pf, "w") as fh:
            fh.write(f"{nx}\n{ny}\n{steps}\n10\n0.1\n0.01\n1.85\n")
        if os.path.exists(os.path.join(inputs, f"obstacles_{nx}x{ny}.dat")):
            obf = os.path.join(inputs, f"obstacles_{nx}x{ny}.dat")
        else:
            obf = os.path.join(inputs, "obstacles_1024x1024.dat")
            env["LBM_TILE"] = "1024x1024"
        try:
            out = subprocess.run([cli, pf, obf], cwd=tmp, env=env, capture_output=True, text=True, timeout=180)
        except Exception:
            return None
    if out.returncode != 0:
        return None
    for line in out.stdout.splitlines():
        if line.startswith("Elapsed Compute time"):
            secs = float(line.split()[-2])
            if secs > 0:
                return {"value": nx * ny * steps / secs / 1e6, "unit": "MLUPS", "cores": threads, "kind": "port",
                        "sample": f"{steps} steps of the same {nx}x{ny} grid, oracle/lbm_oracle_cli LBM_ORACLE_FORM=fused "
                                  f"(two-lattice SoA pull sweep, the formulation of OpenMP/d2q9-bgk.c:334), "
                                  f"OMP_NUM_THREADS={threads} OMP_PROC_BIND=true OMP_PLACES=cores, {secs:.2f} s compute"}
    return None


def runs_resident(info, steps):
    """True when an lbm_run call of `steps` timesteps on this engine is served by the resident kernel."""
    return info.get("resident_steps", 0) > 0 and steps >= info.get("resident_min_steps", 1)


def pmc_record(nx, ny, math, info, steps=0):
    """Committed rocprofv3 --pmc figures for EXACTLY this kernel geometry (profiles/pmc_traffic.json, collected as
    MI355X_MICROARCH.md prescribes: separate passes, FETCH_SIZE x2 on gfx950), or None.  Not a measurement of this
    run: the record names the commit and the run it came from, and is dropped when grid, math, timesteps per
    launch, band height or cells per lane differ from what is running."""
    if runs_resident(info, steps):
        key = f"{nx}x{ny}/resident"         # lbm::resident_band: one geometry per grid, the same arithmetic in both math modes
    else:
        key = (f"{nx}x{ny}/math={math}/steps_per_launch={info['steps_per_launch']}"
               f"/band={info['band_rows']}/lane_cells={info['lane_cells']}")
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            rec = json.load(fh).get(key)
    except Exception:
        return None
    if rec:
        rec = dict(rec, key=key)
    return rec


def self_launch(args):
    """`bench.py --gpus N` without a torch.distributed environment: run the N ranks under torch.distributed.run as a
    child process of this one -- nothing in THIS process has imported torch or touched HIP -- relay the one JSON line
    its rank 0 prints and its exit status.  A watchdog kills the child's process group (its own session) when it
    outlives the limit; this process then exits non-zero.  Never exec: the GPU boxes forbid replacing a process
    image, and a parent that stays around is what makes the watchdog possible."""
    import signal
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, LBM_BENCH_CHILD="1")
    if args.gpus == 1:
        env["LBM_BENCH_RANK_API"] = "1"      # a world of one through the rank code path: RCCL communicator of one rank
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes needs dmabuf IPC on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    limit = float(os.environ.get("LBM_BENCH_LAUNCH_TIMEOUT", "1500"))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
    timed_out = []

    def kill_child():
        timed_out.append(True)
        try:
            os.killpg(child.pid, signal.SIGKILL)     # the exact process group this function started
        except ProcessLookupError:
            pass

    timer = threading.Timer(limit, kill_child)
    timer.daemon = True
    timer.start()
    out, _ = child.communicate()
    timer.cancel()
    try:
        os.killpg(child.pid, signal.SIGKILL)         # ranks that outlived the launcher (a hung collective)
    except (ProcessLookupError, PermissionError):
        pass
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    rc = child.returncode
    if timed_out:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank child did not finish within {limit:.0f} s and was killed "
                         f"(LBM_BENCH_LAUNCH_TIMEOUT)\n")
        rc = rc or 124
    if rc == 0 and not lines:
        sys.stderr.write("bench.py: the child printed no JSON line\n")
        rc = 6
    if rc != 0:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank run failed (exit status {rc}).  If the failure is in the "
                         f"engine's own multi-GPU issue path rather than in the launch: the one-process-per-GPU form used "
                         f"here issues from one host thread per rank; the one-process form (host program, LBM_GPUS=n) "
                         f"uses one issuing thread per slab by default on distinct devices -- LBM_THREADS=0 forces its "
                         f"single-thread issue path; LBM_RCCL_LIB=<path> selects another librccl.\n")
    sys.exit(rc if rc >= 0 else 128 - rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", default="8192x8192")
    ap.add_argument("--math", default="exact", choices=["exact", "fast"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    nx, ny = (int(v) for v in args.grid.lower().split("x"))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("LBM_BENCH_SELF_LAUNCH") == "1"):
        self_launch(args)                    # does not return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` "
                         f"(it launches its ranks itself) or under torch.distributed.run with --nproc-per-node {args.gpus}")
    repeats = max(1, int(os.environ.get("LBM_BENCH_REPEATS", "5")))
    prewarm_s = float(os.environ.get("LBM_BENCH_PREWARM_S", "0.3"))

    # stdout must carry exactly ONE JSON line, but RCCL prints a five-line version banner on the C-level stdout
    # at communicator creation (whatever NCCL_DEBUG says): keep the real stdout aside for the line and point
    # file descriptor 1 at stderr for everything else
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    lbm = load_package()
    if not os.path.exists(lbm.LIB_PATH):
        raise SystemExit("liblbm_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    # device count first (counting does not initialise the GPU): every rank of an over-subscribed launch leaves at
    # once, with a message, instead of one rank dying inside a collective the others then wait in
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    rehearsal = os.environ.get("LBM_BENCH_REHEARSAL") == "hosted" and world > 1
    if rehearsal:
        local_rank = local_rank % n_dev          # ranks share devices; halos through the host
    if (local_world > n_dev or local_rank >= n_dev) and not rehearsal:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} needs {local_world} devices on this node, {n_dev} visible "
                         f"(rank {rank})\n")
        sys.exit(3)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU path")
    torch.cuda.set_device(local_rank)

    # LBM_BENCH_RANK_API=1 takes the one-process-per-GPU code path even for a world of one
    # (rehearsal of the torchrun path on a 1-GPU box)
    use_rank_api = world > 1 or os.environ.get("LBM_BENCH_RANK_API") == "1"
    if use_rank_api and "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
    if use_rank_api and rehearsal:
        dist.init_process_group(backend="gloo")
    elif use_rank_api:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if rehearsal else "cuda"     # where the harness's own reductions live

    def host_exchange(plan, bufs):               # the four messages of a pass over gloo (rehearsal only)
        ops = [dist.P2POp(dist.isend if op["is_send"] else dist.irecv, torch.from_numpy(buf), op["peer"])
               for op, buf in zip(plan, bufs)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()

    def host_allreduce(values):
        dist.all_reduce(torch.from_numpy(values), op=dist.ReduceOp.SUM)

    device_slabs = [1]      # slabs the single-process engine cuts the grid into (all on device 0): 1 except in one side line

    def make_engine(p, ob, tiled):
        if not use_rank_api:
            return lbm.Engine(p, ob, None, n_gpus=device_slabs[0], math=args.math, tiled=tiled)
        if rehearsal:
            return lbm.Engine(p, ob, None, math=args.math, rank=rank, world_size=world, device=local_rank,
                              host_comm=(host_exchange, host_allreduce), tiled=tiled)
        uid = [lbm.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        return lbm.Engine(p, ob, None, math=args.math, rank=rank, world_size=world,
                          unique_id=uid[0], device=local_rank, tiled=tiled)

    check = {}
    also = {}
    main_result = {}
    printed = threading.Event()

    def compose_line(extras_note=None):
        r = main_result
        elapsed, kernel_ms, info = r["elapsed"], r["kernel_ms"], r["info"]
        spl = info["steps_per_launch"]
        cells = float(nx) * float(ny)
        mlups = cells * args.steps / elapsed / 1e6
        # one launch of the dominant kernel sweeps this rank's slab once (max over ranks = ceil) and advances it
        # `spl` timesteps: it must read the lattice once and write it once -- 72 B per cell whatever spl is.
        rows_per_rank = -(-ny // world)
        compulsory_bytes = BYTES_PER_UPDATE * nx * rows_per_rank
        pmc = pmc_record(nx, ny, args.math, info, args.steps) if world == 1 else None
        if runs_resident(info, args.steps):
            kernel_name = "lbm::resident_band"
            spl = min(args.steps, info["resident_steps"])   # one launch advances the whole timed region: the lattice is
            # read and written once per LAUNCH, so the HBM fraction below is tiny by construction -- the kernel's bound
            # is instruction issue (roofline.limiter), not memory
        elif info["lane_cells"] > 0:      # a stream kernel: several timesteps per pass
            packed = os.environ.get("LBM_PACKED", "1") != "0"     # the packed kernels serve both math modes
            kernel_name = ("lbm::stepk_pk" if packed else
                           ("lbm::stepk_stream" if info["lane_cells"] == 4 else "lbm::step2_stream"))
        else:
            kernel_name = "lbm::step_vec4" if spl == 1 else "lbm::step_tile"
        launch_ms = kernel_ms * spl
        achieved = compulsory_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc.get("traffic_bytes_per_launch") if pmc else None,
                "kernel": kernel_name, "steps_per_launch": spl, "launch_ms": launch_ms,
                "kernel_ms_per_step": kernel_ms,
                "compulsory_bytes_per_launch": compulsory_bytes,
                "algorithmic_bytes_per_launch": compulsory_bytes * spl,
                "algorithmic_GBps": achieved * spl,
                "geometry": {"band_rows": info["band_rows"], "lane_cells": info["lane_cells"],
                             "nontemporal_stores": info["nontemporal"]},
                "note": ("achieved = compulsory bytes of one launch (the slab read once + written once, 72 B per cell) / "
                         "launch time, so frac <= 1 is a fraction of the HBM bound; algorithmic_GBps = 72 B per lattice "
                         f"UPDATE x updates / time = {spl} x achieved (the BASELINE metric; temporal blocking x{spl}: each "
                         "byte moved serves that many updates, so it may exceed the peak)")}
        if pmc:
            t = pmc.get("traffic_bytes_per_launch")
            roof["traffic_source"] = (f"committed rocprofv3 --pmc profile {pmc.get('source')} at commit {pmc.get('commit')}, "
                                      f"same grid / math / kernel geometry ({pmc['key']}); NOT measured in this run")
            if t and launch_ms > 0:
                roof["traffic_over_compulsory"] = t / compulsory_bytes
                roof["traffic_GBps_at_this_runs_launch_time"] = t / (launch_ms * 1e-3) / 1e9
            if pmc.get("valu_busy") is not None:
                issue = pmc.get("issue_busy")
                roof["limiter"] = {"bound": "instruction issue" if (issue or 0) > 0.75 or pmc["valu_busy"] > 0.75 else "hbm",
                                   "frac": issue if issue is not None else pmc["valu_busy"],
                                   "valu_busy": pmc["valu_busy"], "issue_busy": issue,
                                   "lane_instructions_per_update": pmc.get("lane_instructions_per_update"),
                                   "what": pmc.get("limiter_note"), "source": "same committed profile"}
        line = {
            "metric": "MLUPS", "value": mlups, "unit": "MLUPS (million lattice updates/s)",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"D2Q9-BGK timestep loop, {r['workload']}, uniform-equilibrium start",
                       "grid": f"{nx}x{ny}", "math": args.math,
                       "timesteps_per_memory_pass": spl,
                       "decomposition": (f"REHEARSAL: {world} ranks sharing {n_dev} device(s), halo rows through the host over gloo "
                                         f"(lbm_create_rank_hosted_tiled) -- exercises this file's multi-rank path, measures nothing"
                                         if rehearsal else
                                         f"{world} row slab(s), one process per GPU, RCCL halo send/recv" if world > 1
                                         else ("one rank through the rank pipeline: halo rows by RCCL self-exchange, "
                                               "interior / boundary split" if use_rank_api and os.environ.get("LBM_FORCE_HALO") == "1"
                                               else ("one rank through the rank API (communicator of one), periodic in-kernel"
                                                     if use_rank_api else "single slab, periodic in-kernel")))},
            "timing": {"repeats": len(r["repeats"]), "statistic": "median",
                       "repeats_ms_per_step": [e / args.steps * 1e3 for e in r["repeats"]],
                       "prewarm_steps_untimed": r["prewarm_steps"], "prewarm_target_s": prewarm_s},
            "roofline": roof,
            "results_finite": r["finite"],
            "rccl": dict(r.get("rccl") or {}, torch_backend=(("gloo (rehearsal: no RCCL anywhere in this run)" if rehearsal else
                                                              "nccl (torch.distributed barrier / reductions of the bench harness)")
                                                             if use_rank_api else None),
                         note=("REHEARSAL: halo rows through the host over gloo, the engine holds no communicator" if rehearsal else "the engine's halo exchange: RCCL bound at first use -- LBM_RCCL_LIB, else the librccl "
                               "already in the process (torch's here), else /opt/rocm/lib; nranks = ncclCommCount of "
                               "the engine's communicator" if (r.get("rccl") or {}).get("loaded") else
                               "single slab: no communicator, RCCL not loaded by the engine")),
        }
        if check:
            line["multi_gpu_check"] = dict(check)
        if also:
            line["also"] = dict(also)
        if extras_note:
            line["extras_note"] = extras_note
        return line

    def print_once(line):
        if not printed.is_set():
            printed.set()
            os.write(real_stdout, (json.dumps(line) + "\n").encode())

    verify_wanted = [False]

    def extras_watchdog():
        """The headline measurement is done; the verification and the extra configurations that follow must not
        be able to lose it -- nor may a hang there pass for success.  If they have not finished in time, rank 0
        prints the line with what it has, marked as failed where a verification was pending, and EVERY rank exits
        non-zero (a blocked HIP / RCCL call cannot be interrupted from Python)."""
        if rank == 0:
            note = (f"verification / extra configurations did not finish within {extras_timeout:.0f} s and were abandoned")
            if verify_wanted[0] and "fields_bitwise_equal_to_single_gpu_run" not in check:
                check.update({"error": "timeout", "fields_bitwise_equal_to_single_gpu_run": False})
                main_result["finite"] = False
            line = compose_line(note)
            line["extras_failed"] = True
            print_once(line)
        os._exit(4)

    extras_timeout = float(os.environ.get("LBM_BENCH_EXTRA_TIMEOUT", "300"))

    def verify_against_single_gpu(p, ob, tiled, eng, av, total):
        """After the timed region of a multi-rank run: every rank re-runs the SAME workload as one
        periodic slab on its own GPU and compares its rows of the final u_x, u_y, |u| and pressure
        fields bit for bit, and the all-reduced av_vels (which differ by summation order only)."""
        info = eng.info()
        mine = eng.final_state()
        forced = os.environ.pop("LBM_FORCE_HALO", None)     # the reference run is a plain periodic slab
        try:
            with (lbm.Engine(p, ob, None, n_gpus=1, math=args.math, tiled=tiled) if rehearsal else
                  lbm.Engine(p, ob, None, math=args.math, rank=0, world_size=1,
                             unique_id=lbm.rccl_unique_id(), device=local_rank, tiled=tiled)) as ref:
                ref.run(total)
                ref_av = ref.av_vels(total)
                whole = ref.final_state()
        finally:
            if forced is not None:
                os.environ["LBM_FORCE_HALO"] = forced
        rows = slice(info["row_first"], info["row_first"] + info["row_count"])
        same = all(np.array_equal(mine[k].view(np.uint32), whole[k][rows].view(np.uint32)) for k in mine)
        rel = float(np.max(np.abs(av.astype(np.float64) - ref_av) / np.abs(ref_av)))
        t = torch.tensor([0.0 if same else 1.0, rel], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seen = main_result.get("rccl") or eng.rccl_info()
        check.update({"fields_bitwise_equal_to_single_gpu_run": bool(t[0] == 0.0),
                      "av_vels_max_rel_diff": float(t[1]), "ranks_checked": world, "steps": total,
                      "rccl": {k: seen.get(k) for k in ("version_string", "nranks", "nranks_min_over_ranks",
                                                         "nranks_max_over_ranks", "library")}})

    watchdog = threading.Timer(extras_timeout, extras_watchdog)
    watchdog.daemon = True

    def measure(gx, gy, steps, warmup, verify=False, headline=False, n_repeats=1, prewarm=0.0):
        """`warmup` untimed timesteps of a gx x gy grid, an untimed pre-warm of about `prewarm` seconds, then
        `n_repeats` timed regions of `steps` timesteps each: barrier + device sync on both sides, max over
        ranks.  Returns the median region: (seconds, kernel ms per step, engine info, av_vels finite, workload)."""
        prewarm_cap = 40000
        p, ob, workload, tiled = synthetic_case(lbm, gx, gy, warmup + prewarm_cap + n_repeats * steps)
        eng = make_engine(p, ob, tiled)

        def fence():
            eng.sync()
            torch.cuda.synchronize()
            if use_rank_api:
                dist.barrier()
                torch.cuda.synchronize()

        def agree(v):   # every rank must run the same number of steps: take rank 0's view
            if not use_rank_api:
                return v
            t = torch.tensor([float(v)], dtype=torch.float64, device=red_dev)
            dist.broadcast(t, src=0)
            return float(t[0])

        done = 0
        fence()
        t0 = time.perf_counter()
        if warmup > 0:
            eng.run(warmup)
            done += warmup
        fence()
        spent = time.perf_counter() - t0
        prewarm_steps = 0
        if prewarm > 0:
            # chunks of `steps` until the device has been busy for `prewarm` seconds (clock ramp-up)
            per_step = agree(spent / max(warmup, 1)) if warmup > 0 else 0.0
            while agree(spent) < prewarm and prewarm_steps + steps <= prewarm_cap:
                chunk = steps
                if per_step > 0:
                    chunk = int(min(max(steps, (prewarm - spent) / per_step), prewarm_cap - prewarm_steps))
                    chunk = int(agree(chunk))
                t1 = time.perf_counter()
                eng.run(chunk)
                fence()
                dt = time.perf_counter() - t1
                spent += dt
                per_step = agree(dt / chunk)
                prewarm_steps += chunk
            done += prewarm_steps
        runs = []
        for _ in range(n_repeats):
            fence()
            t0 = time.perf_counter()
            kernel_ms = eng.run_timed(steps)
            fence()
            elapsed = time.perf_counter() - t0
            if use_rank_api:
                t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed, kernel_ms = float(t[0]), float(t[1])
            runs.append((elapsed, kernel_ms))
            done += steps
        elapsed, kernel_ms = sorted(runs)[len(runs) // 2]
        av = eng.av_vels(done)          # forces the cross-rank reduce too
        finite = bool(np.isfinite(av).all())
        info = eng.info()
        if headline:
            rc_info = eng.rccl_info()
            if use_rank_api:
                # every rank's view of the ring: the record must show that RCCL saw `world` ranks on all of them
                t = torch.tensor([float(rc_info["nranks"]), -float(rc_info["nranks"]), float(rc_info["version"])],
                                 dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                rc_info["nranks_min_over_ranks"] = int(-t[1])
                rc_info["nranks_max_over_ranks"] = int(t[0])
                rc_info["version_max_over_ranks"] = int(t[2])
            main_result.update(elapsed=elapsed, kernel_ms=kernel_ms, info=info, finite=finite, workload=workload,
                               repeats=[e for e, _ in runs], prewarm_steps=prewarm_steps, rccl=rc_info)
            verify_wanted[0] = verify
            watchdog.start()
        if verify:
            try:
                verify_against_single_gpu(p, ob, tiled, eng, av, done)
            except Exception as exc:
                check.update({"error": str(exc), "fields_bitwise_equal_to_single_gpu_run": False})
        eng.close()
        return elapsed, kernel_ms, info, finite, workload

    # (outside the timed region) a multi-rank result is checked against a single-GPU run of the same workload
    verify = use_rank_api and (world > 1 or os.environ.get("LBM_FORCE_HALO") == "1") and \
        os.environ.get("LBM_BENCH_VERIFY", "1") != "0"
    measure(nx, ny, args.steps, args.warmup, verify, headline=True, n_repeats=repeats, prewarm=prewarm_s)

    # BASELINE.json's other named configurations, measured the same way (every rank takes part):
    # the reference's own 1024x1024 data set (20 000 steps in the reference; Infinity-Cache resident,
    # so MLUPS only, no HBM figure; strong-scaling it over several GPUs is exchange-latency bound and
    # reported as measured) and the 16384x16384 synthetic grid of the scaling configuration
    if os.environ.get("LBM_BENCH_ALSO", "1") != "0":
        extra = [(1024, 1024, 2000, 200, "reference data set 1024x1024, cache resident"),
                 (16384, 16384, 100, 10, "synthetic 16384x16384 (BASELINE.json configs[4])")]
        if world > 1:
            # the same per-GPU work as the 1-GPU line (weak scaling): 8192 x 8192 cells per rank
            extra.append((8192, 8192 * world, 100, 10, f"weak scaling: 8192x8192 cells per GPU, {world} GPUs"))
        for (gx, gy, st, wu, note) in extra:
            if (gx, gy) == (nx, ny):
                continue
            try:
                dt, k_ms, inf, fin, _ = measure(gx, gy, st, wu, n_repeats=3)
                also[f"{gx}x{gy}"] = {"value": gx * gy * st / dt / 1e6, "unit": "MLUPS", "n_gpus": args.gpus,
                                      "ms_per_step": dt / st * 1e3, "kernel_ms_per_step": k_ms, "steps": st,
                                      "warmup": wu, "repeats": 3, "steps_per_launch": inf["steps_per_launch"],
                                      "band_rows": inf["band_rows"], "lane_cells": inf["lane_cells"],
                                      "results_finite": fin, "note": note}
                if runs_resident(inf, st):
                    also[f"{gx}x{gy}"].update(kernel="lbm::resident_band", steps_per_launch=min(st, inf["resident_steps"]),
                                              kernel_note="one launch per lbm_run call: the lattice stays in registers, 4-row bands "
                                                          "per workgroup, seam rows through tagged L2 granules")
                rec = pmc_record(gx, gy, args.math, inf, st) if world == 1 else None
                if rec:     # committed PMC profile of exactly this kernel geometry (not measured in this run)
                    also[f"{gx}x{gy}"]["limiter"] = {"valu_busy": rec.get("valu_busy"), "issue_busy": rec.get("issue_busy"),
                                                     "lane_instructions_per_update": rec.get("lane_instructions_per_update"),
                                                     "wave_cycles_share": rec.get("wave_cycles_share"),
                                                     "what": rec.get("limiter_note"),
                                                     "source": f"committed profile {rec.get('source')} at commit {rec.get('commit')}, "
                                                               f"key {rec['key']}; NOT measured in this run"}
            except Exception as exc:            # never lose the main line over an extra one
                also[f"{gx}x{gy}"] = {"error": str(exc)}
                if use_rank_api:
                    break                       # ranks may have diverged: stop issuing collectives

    # the one-timestep-per-pass kernel on the headline grid, measured in the same run: the kernel that IS bound by HBM
    # bandwidth (72 B per update actually move), beside the temporally blocked one whose limit is instruction issue
    if world == 1 and not use_rank_api and os.environ.get("LBM_BENCH_ALSO", "1") != "0" and "LBM_FUSE2" not in os.environ:
        os.environ["LBM_FUSE2"] = "0"
        try:
            dt, k_ms, inf, fin, _ = measure(nx, ny, max(args.steps, 20), args.warmup, n_repeats=3)
            if inf["steps_per_launch"] == 1 and k_ms > 0:
                gbps = BYTES_PER_UPDATE * nx * ny / (k_ms * 1e-3) / 1e9
                also["one_step_kernel"] = {"kernel": "lbm::step_vec4", "grid": f"{nx}x{ny}", "ms_per_step": dt / max(args.steps, 20) * 1e3,
                                           "kernel_ms_per_step": k_ms, "value": nx * ny / (k_ms * 1e-3) / 1e6, "unit": "MLUPS",
                                           "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS, "bound": "hbm",
                                           "results_finite": fin,
                                           "note": "LBM_FUSE2=0: one timestep per pass, every update moves its 72 B"}
        except Exception as exc:
            also["one_step_kernel"] = {"error": str(exc)}
        finally:
            os.environ.pop("LBM_FUSE2", None)

    # the headline grid as TWO slabs on the one device (interior / boundary launches of both slabs in flight together, halo
    # rows by device copies): the waves of a single launch all start together and the strips that hold wall columns run
    # ~6 % longer than the others, so every launch ends with idle SIMDs; two half-size launches fill each other's tails.
    # A side line only: the headline keeps one launch per pass, whose roofline arithmetic is unambiguous.
    if world == 1 and not use_rank_api and os.environ.get("LBM_BENCH_ALSO", "1") != "0" and main_result and \
            main_result["info"]["band_rows"] > 0 and not any(k in os.environ for k in ("LBM_HALO", "LBM_BAND_ROWS", "LBM_FUSE2")):
        os.environ.update(LBM_HALO="memcpy", LBM_BAND_ROWS=str(main_result["info"]["band_rows"]))
        device_slabs[0] = 2
        try:
            st = max(args.steps, 20)
            dt, k_ms, inf, fin, _ = measure(nx, ny, st, args.warmup, n_repeats=3)
            also["two_slabs_on_the_one_device"] = {
                "grid": f"{nx}x{ny}", "value": nx * ny * st / dt / 1e6, "unit": "MLUPS", "ms_per_step": dt / st * 1e3,
                "steps_per_launch": inf["steps_per_launch"], "band_rows": inf["band_rows"], "results_finite": fin,
                "note": "same grid, same kernels, cut into two row slabs that share the device (halo rows by device copies; "
                        "lattice bit-identical to the single slab: tests/test_gpu_parity.py); two half-size launches per pass "
                        "run concurrently and fill each other's tails (DESIGN.md section 4)"}
        except Exception as exc:
            also["two_slabs_on_the_one_device"] = {"error": str(exc)}
        finally:
            device_slabs[0] = 1
            os.environ.pop("LBM_HALO", None)
            os.environ.pop("LBM_BAND_ROWS", None)

    watchdog.cancel()
    failed = bool(check) and not check.get("fields_bitwise_equal_to_single_gpu_run", False)
    if rank == 0:
        line = compose_line()
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(nx, ny)
            if base:
                line["cpu_baseline"] = base
            multi = cpu_baseline_multicore(nx, ny)
            if multi:
                line["cpu_baseline_multicore"] = multi
            ref = cpu_reference()
            if ref:
                line["cpu_reference_1024x1024"] = ref
        print_once(line)

    if use_rank_api:
        dist.destroy_process_group()
    if failed:
        sys.exit(5)     # the multi-rank result differs from the single-GPU run: not a successful bench


if __name__ == "__main__":
    main()
