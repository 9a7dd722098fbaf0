"""CPU checks of the C-ABI boundary: the shared library loads, exports exactly what
include/lbm_hip.h declares, and its host-only entry points behave (no compute without a GPU)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "lbm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbm_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(lbm):
    lib = ctypes.CDLL(lbm.LIB_PATH)
    declared = header_symbols()
    assert declared, "no prototypes found in include/lbm_hip.h"
    for name in declared:
        assert hasattr(lib, name), f"liblbm_hip.so does not export {name}"
    assert sorted(lbm.ABI_SYMBOLS) == declared, "binding symbol list out of date with the header"


def test_exported_symbols_are_plain_c(lbm):
    """extern "C": the dynamic symbol table holds the unmangled names."""
    out = subprocess.run(["nm", "-D", "--defined-only", lbm.LIB_PATH], capture_output=True, text=True)
    names = {line.split()[-1] for line in out.stdout.splitlines() if line.strip()}
    for name in header_symbols():
        assert name in names


def test_version_and_device_count(lbm):
    lib = lbm.load_library()
    assert lib.lbm_version().decode().startswith("lbm_hip")
    assert lbm.device_count() >= 0


@pytest.mark.parametrize("ny,parts", [(128, 1), (128, 8), (8192, 8), (23, 4), (16384, 7), (16, 8)])
def test_partition_rows_covers_grid(lbm, ny, parts):
    seen = []
    for i in range(parts):
        first, count = lbm.partition_rows(ny, parts, i)
        assert count >= (2 if parts > 1 else 1)
        seen.append((first, count))
    assert seen[0][0] == 0
    for (f0, c0), (f1, _) in zip(seen, seen[1:]):
        assert f0 + c0 == f1
    assert seen[-1][0] + seen[-1][1] == ny
    counts = [c for _, c in seen]
    assert max(counts) - min(counts) <= 1


def test_partition_rows_rejects_thin_slabs(lbm):
    with pytest.raises(lbm.LbmError):
        lbm.partition_rows(9, 8, 0)
    with pytest.raises(lbm.LbmError):
        lbm.partition_rows(128, 0, 0)


def test_create_fails_loudly_without_device_or_with_bad_args(lbm, datasets):
    """No CPU fallback: on a box without a GPU lbm_create must fail, not compute."""
    p, ob = datasets("128x128")
    if lbm.device_count() == 0:
        with pytest.raises(lbm.LbmError, match="no HIP device"):
            lbm.Engine(p, ob)
    bad = lbm.Params(0, 128, 10, 10, 0.1, 0.005, 1.85)
    with pytest.raises(lbm.LbmError):
        lbm.Engine(bad, ob)
    with pytest.raises(lbm.LbmError):
        lbm.Engine(p, ob, n_gpus=99)
    # more cells than an int can count (the reference's tot_cells is an int): rejected by the C side
    import ctypes
    huge = lbm.Params(65536, 65536, 1, 10, 0.1, 0.005, 1.85)
    lib = lbm.load_library()
    dummy = np.zeros(4, dtype=np.int32)
    assert not lib.lbm_create(ctypes.byref(huge._c()), dummy.ctypes.data, None, 1, 0)
    assert b"invalid parameters" in lib.lbm_last_error()


def test_die_mode_matches_reference_message_shape(lbm):
    """Default error mode = the reference's die(): 'Error at line N of file F:' + message, exit 1
    (SerialCode/d2q9-bgk.c:745-751).  Checked in a child process."""
    code = (
        "import ctypes,sys\n"
        f"lib=ctypes.CDLL({lbm.LIB_PATH!r})\n"
        "lib.lbm_partition_rows.argtypes=[ctypes.c_int]*3+[ctypes.c_void_p]*2\n"
        "lib.lbm_partition_rows(9,8,0,None,None)\n"
        "print('survived')\n")
    out = subprocess.run(["python3", "-c", code], capture_output=True, text=True)
    assert out.returncode == 1
    assert "survived" not in out.stdout
    assert re.search(r"^Error at line \d+ of file .*:\n.*fewer than 2 rows", out.stderr, flags=re.M)


def test_halo_plan_pairs_up_around_the_ring(lbm):
    """lbm_halo_plan (host-only): every send has exactly one matching receive on the peer, in the same position of
    the peer's posting order when both neighbours are the same peer (two parts) -- the property RCCL / MPI rely on."""
    for parts in (1, 2, 3, 8):
        rows = [lbm.partition_rows(64, parts, i)[1] for i in range(parts)]
        for depth in (1, 2, 3):
            plans = [lbm.halo_plan(rows[i], parts, i, depth) for i in range(parts)]
            for i, plan in enumerate(plans):
                assert [op["is_send"] for op in plan] == [True, True, False, False]
                n, s_ = (i + 1) % parts, (i - 1) % parts
                assert [op["peer"] for op in plan] == [n, s_, s_, n]
                assert plan[0]["row_first"] == rows[i] - depth and plan[1]["row_first"] == 0
                assert plan[2]["row_first"] == -depth and plan[3]["row_first"] == rows[i]
                assert all(op["row_count"] == depth for op in plan)
                # k-th send to peer q pairs with the k-th receive q posts from me
                for q in set(op["peer"] for op in plan):
                    sends = [op for op in plan if op["is_send"] and op["peer"] == q]
                    recvs = [op for op in plans[q] if not op["is_send"] and op["peer"] == i]
                    assert len(sends) == len(recvs)
                    for a, b in zip(sends, recvs):
                        # my top rows land in the peer's south halo, my bottom rows in its north halo
                        assert (a["row_first"] == 0) == (b["row_first"] == rows[q])
    with pytest.raises(lbm.LbmError):
        lbm.halo_plan(4, 2, 0, 5)          # deeper than the slab


def test_plan_halo_depth_follows_the_kernel_policy(lbm, monkeypatch):
    for k in ("LBM_FUSE2", "LBM_LANE_CELLS", "LBM_PASS_STEPS", "LBM_VEC4", "LBM_PACKED"):
        monkeypatch.delenv(k, raising=False)
    small = lbm.Params(128, 256, 10, 10, 0.1, 0.005, 1.85)
    big = lbm.Params(8192, 8192, 10, 10, 0.1, 0.01, 1.85)
    thin = lbm.Params(128, 40, 10, 10, 0.1, 0.005, 1.85)
    ragged = lbm.Params(130, 64, 10, 10, 0.1, 0.005, 1.85)
    assert lbm.plan_halo_depth(small, 2) == 3          # three-step packed kernel, 2 cells per lane
    assert lbm.plan_halo_depth(small, 2, "fast") == 3  # the packed stream kernels serve both math modes
    assert lbm.plan_halo_depth(big, 8) == 4            # 8192x1024 per rank: four-step packed kernel
    assert lbm.plan_halo_depth(big, 2) == 4 and lbm.plan_halo_depth(big, 2, "fast") == 4
    assert lbm.plan_halo_depth(ragged, 2) == 1         # nx % 4 != 0: one step per pass
    assert lbm.plan_halo_depth(thin, 8) == 2           # 5-row slabs: too thin for three-row halos
    monkeypatch.setenv("LBM_PACKED", "0")               # the scalar kernels: fast math runs K = 2 / 3
    assert lbm.plan_halo_depth(small, 2, "fast") == 2 and lbm.plan_halo_depth(big, 2, "fast") == 3
    monkeypatch.delenv("LBM_PACKED")
    monkeypatch.setenv("LBM_PASS_STEPS", "2")
    assert lbm.plan_halo_depth(big, 8) == 2


def test_rccl_is_bound_at_first_use_and_named(lbm):
    """RCCL is not a link-time dependency of the engine: liblbm_hip.so carries no DT_NEEDED entry for it, and
    lbm_rccl_info(NULL) binds it by the stated rule (here, a process without torch: ROCm's own) and says which file and
    version it was -- no device needed."""
    import subprocess
    needed = subprocess.run(["readelf", "-d", lbm.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" not in needed
    info = lbm.rccl_info()
    assert info["loaded"] and info["version"] >= 22000 and info["n_comms"] == 0 and info["nranks"] == 0
    assert os.path.exists(info["library"]) and "librccl" in info["library"]
