"""First contact with a multi-GPU node, rehearsed on the one-GPU box: bench.py launches its own
ranks as a child process, the record says which RCCL served the run and how many ranks it saw, an over-subscribed
launch fails fast, and the RCCL self-exchange is bit-exact under both libraries a process can end up with."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(args, env_extra, timeout):
    env = dict(os.environ, LBM_BENCH_ALSO="0", LBM_BENCH_REPEATS="1", LBM_BENCH_PREWARM_S="0", **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--no-cpu-baseline"], env=env,
                         capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    return out, time.time() - t0


def test_bench_launches_its_own_ranks_and_reports_rccl():
    """`python bench.py --gpus 1` through the child-launch path (torch.distributed.run started by bench.py itself):
    exactly one JSON line, a communicator of one rank, library and version named."""
    out, _ = run_bench(["--gpus", "1", "--steps", "8", "--warmup", "4", "--grid", "2048x2048"],
                       {"LBM_BENCH_SELF_LAUNCH": "1"}, 400)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["results_finite"]
    rc = line["rccl"]
    assert rc["loaded"] and rc["nranks"] == 1 and rc["rank"] == 0 and rc["n_comms"] == 1
    assert rc["nranks_min_over_ranks"] == rc["nranks_max_over_ranks"] == 1
    assert rc["version"] >= 22000 and "librccl" in rc["library"]
    assert "rank API" in line["config"]["decomposition"]


def test_bench_oversubscribed_launch_fails_fast(lbm):
    """`python bench.py --gpus 2` on a one-GPU box: non-zero exit, no JSON line, no hang."""
    if lbm.device_count() >= 2:
        pytest.skip("needs a box with a single GPU")
    out, secs = run_bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--grid", "1024x1024"], {}, 300)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "needs 2 devices" in out.stderr
    assert "LBM_THREADS=0" in out.stderr          # the failure message names the single-thread override
    assert secs < 180


@pytest.mark.parametrize("ranks", [2, 3])
def test_bench_multi_rank_path_rehearsed_on_one_gpu(ranks):
    """`python bench.py --gpus N` with N ranks sharing the one GPU and their halo rows travelling through the host
    (LBM_BENCH_REHEARSAL=hosted): the launch by bench.py itself, the agreement of the ranks on step counts, the
    max-over-ranks timing, the bitwise self-check against a single-GPU run and the exit status -- every line of the
    multi-rank path of bench.py except the RCCL transport -- run before the first real multi-GPU node sees them."""
    out, _ = run_bench(["--gpus", str(ranks), "--steps", "8", "--warmup", "4", "--grid", "2048x1536"],
                       {"LBM_BENCH_REHEARSAL": "hosted"}, 600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["results_finite"]
    assert "REHEARSAL" in line["config"]["decomposition"]
    chk = line["multi_gpu_check"]
    assert chk["fields_bitwise_equal_to_single_gpu_run"] is True and chk["ranks_checked"] == ranks, chk
    assert chk["av_vels_max_rel_diff"] < 1e-4


@pytest.mark.parametrize("torch_first,forced", [(False, None), (True, None), (True, "/opt/rocm/lib/librccl.so.1")])
def test_rccl_self_exchange_under_both_libraries(torch_first, forced):
    """The engine binds the librccl already in the process (torch's bundled one when torch was imported first -- the
    situation of bench.py and of this test suite) or ROCm's own (the C host program), or the file LBM_RCCL_LIB names
    (here: ROCm's next to torch's HIP runtime): the ring of one must be bit-identical to the plain periodic run under
    each, and the record must name the library."""
    env = dict(os.environ)
    env.pop("LBM_RCCL_LIB", None)
    if forced:
        if not os.path.exists(forced):
            pytest.skip(f"{forced} not present")
        env["LBM_RCCL_LIB"] = forced
    cmd = [sys.executable, os.path.join(ROOT, "tools", "rccl_self.py")] + (["--torch-first"] if torch_first else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["fields_equal"], rec
    assert not rec["single_slab_loaded_rccl"]                     # a single slab binds no RCCL
    rc = rec["rccl"]
    assert rc["loaded"] and rc["nranks"] == 1 and rc["n_comms"] == 1
    if torch_first and not forced:
        assert "/torch/lib/" in rc["library"], rc
    else:
        assert rc["library"].startswith("/opt/rocm"), rc
    # one RCCL per process unless a second one was asked for by name
    assert len([m for m in rec["mapped"] if "librccl" in m]) == (2 if forced else 1), rec["mapped"]
