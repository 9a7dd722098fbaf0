"""End-to-end runs on the GPU at the reference's full sizes, through the C host program.

* the four reference data sets at their full iteration counts with the command line
  `d2q9-bgk <paramfile> <obstaclefile>`: `final_state.dat` must be BYTE-IDENTICAL to the file the
  reference's SerialCode binary wrote (md5 recorded in tests/golden/serialcode_*.npz), and both
  output files must pass the check.py rule (<= 1 %) against the reference's double-precision
  goldens where those exist (check/*.dat; tests/golden/check_goldens.npz);
* BASELINE.json's 8192x8192 synthetic grid: a short run compared bit for bit with the CPU oracle's
  multi-threaded fused form, plus size-independent properties (mass conservation, zero velocity
  in blocked cells, run-splitting invariance).
"""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


@pytest.mark.parametrize("name", ["128x128", "128x256", "256x256", "1024x1024"])
def test_cli_full_run_matches_serialcode_bytes_and_goldens(lbm, tmp_path, name):
    if not os.path.exists(lbm.CLI_PATH):
        lbm.build()
    pf = os.path.join(GOLDEN, "inputs", f"input_{name}.params")
    of = os.path.join(GOLDEN, "inputs", f"obstacles_{name}.dat")
    out = subprocess.run([lbm.CLI_PATH, pf, of], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[0] == "==done==" and lines[1].startswith("Reynolds number:\t\t")

    ref = np.load(os.path.join(GOLDEN, f"serialcode_{name}.npz"))
    gold = np.load(os.path.join(GOLDEN, "check_goldens.npz"))

    # final_state.dat: the very bytes SerialCode writes
    assert md5(tmp_path / "final_state.dat") == str(ref["md5_final_state"])

    # av_vels.dat: same format, values within summation-order noise of SerialCode's
    av = np.loadtxt(tmp_path / "av_vels.dat", usecols=[1])
    assert av.size == ref["av_vels"].size
    np.testing.assert_allclose(av, ref["av_vels"].astype(np.float64), rtol=5e-4)
    # Reynolds number printed = calc_reynolds on the final lattice (:196, :637-642)
    re_printed = float(lines[1].split()[-1])
    assert re_printed == pytest.approx(float(ref["reynolds"]), rel=5e-4)

    # the reference's acceptance gate against its own goldens (check/check.py, tolerance 1 %)
    assert lbm.check_passes(gold[f"av_vels_{name}"], av)
    if f"pressure_{name}" in gold:
        pr = np.loadtxt(tmp_path / "final_state.dat", usecols=[5])
        assert lbm.check_passes(gold[f"pressure_{name}"], pr)


@pytest.mark.parametrize("name,env", [
    ("1024x1024", {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "4"}),                            # four steps on pairs, 2 LDS windows
    ("256x256", {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "4"}),
    ("128x256", {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "4", "LBM_PASS_STEPS": "3"}),       # three steps on pairs
    ("1024x1024", {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "4", "LBM_PACKED": "0", "LBM_PASS_STEPS": "3"}),   # scalar stream kernel
    ("256x256", {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "2"}),                              # two steps, one pair per lane
    ("128x128", {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "4", "LBM_GPUS": "4", "LBM_HALO": "memcpy"}),   # 4-row halos, 32-row slabs
])
def test_cli_large_grid_kernels_full_run_match_serialcode_bytes(lbm, tmp_path, name, env):
    """The kernels that large grids get by default (and their A/B alternatives), forced onto the reference's data sets:
    at the reference's full iteration counts (20 000 - 80 000 dependent steps) `final_state.dat` must still be the
    SerialCode binary's, byte for byte -- the strongest parity statement the reference's own outputs allow for the
    headline kernel."""
    pf = os.path.join(GOLDEN, "inputs", f"input_{name}.params")
    of = os.path.join(GOLDEN, "inputs", f"obstacles_{name}.dat")
    out = subprocess.run([lbm.CLI_PATH, pf, of], cwd=tmp_path, capture_output=True, text=True, env=dict(os.environ, **env))
    assert out.returncode == 0, out.stderr
    ref = np.load(os.path.join(GOLDEN, f"serialcode_{name}.npz"))
    assert md5(tmp_path / "final_state.dat") == str(ref["md5_final_state"])
    av = np.loadtxt(tmp_path / "av_vels.dat", usecols=[1])
    np.testing.assert_allclose(av, ref["av_vels"].astype(np.float64), rtol=5e-4)
    gold = np.load(os.path.join(GOLDEN, "check_goldens.npz"))
    assert lbm.check_passes(gold[f"av_vels_{name}"], av)


@pytest.mark.parametrize("name,gpus,fuse", [("128x256", "4", "0"), ("256x256", "3", "1"), ("1024x1024", "8", "1")])
def test_cli_multi_slab_full_run_matches_serialcode_bytes(lbm, tmp_path, name, gpus, fuse):
    """The command line with LBM_GPUS row slabs (here sharing the one device, halos by device copies):
    final_state.dat is still the reference program's, byte for byte, at full iteration counts."""
    pf = os.path.join(GOLDEN, "inputs", f"input_{name}.params")
    of = os.path.join(GOLDEN, "inputs", f"obstacles_{name}.dat")
    env = dict(os.environ, LBM_GPUS=gpus, LBM_HALO="memcpy", LBM_FUSE2=fuse)
    out = subprocess.run([lbm.CLI_PATH, pf, of], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    ref = np.load(os.path.join(GOLDEN, f"serialcode_{name}.npz"))
    assert md5(tmp_path / "final_state.dat") == str(ref["md5_final_state"])
    av = np.loadtxt(tmp_path / "av_vels.dat", usecols=[1])
    np.testing.assert_allclose(av, ref["av_vels"].astype(np.float64), rtol=5e-4)


@pytest.mark.parametrize("mode", ["stale", "freshest"])
@pytest.mark.parametrize("name,gpus", [("128x256", "2"), ("256x256", "4")])
def test_cli_stale_halo_mode_deviation_is_bounded(lbm, golden, tmp_path, name, gpus, mode):
    """LBM_HALO_MODE=stale / freshest through the command line (the reference ships this as a separate program,
    MPI_Testall_OptimizedVersion/d2q9-bgk.c).  The engine's stale mode is the WORST case of that
    program -- every halo exactly one step late, every step -- so it does not always meet check.py's 1 %
    rule on av_vels, which takes the maximum over all steps including the first few, where only a handful
    of rows move at all: measured 4.7 % at step 2 on 128x256 (2 slabs: the seam lies two rows from the
    accelerated row) and 1.2 % at step 9145 on 256x256 (4 slabs).  Asserted here: av_vels within 5 %
    at every step and within 1 % at the end, the pressure field within the 1 % rule.  The freshest-available mode
    lies between that and the synchronous run (on this box, where the slabs share a device, its looks nearly always
    find the fresh rows)."""
    pf = os.path.join(GOLDEN, "inputs", f"input_{name}.params")
    of = os.path.join(GOLDEN, "inputs", f"obstacles_{name}.dat")
    env = dict(os.environ, LBM_GPUS=gpus, LBM_HALO="memcpy", LBM_HALO_MODE=mode)
    out = subprocess.run([lbm.CLI_PATH, pf, of], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr
    gold = np.load(os.path.join(golden, "check_goldens.npz"))
    ref = np.load(os.path.join(GOLDEN, f"serialcode_{name}.npz"))
    assert "EXPERIMENTAL" in out.stderr
    if mode == "stale":
        assert md5(tmp_path / "final_state.dat") != str(ref["md5_final_state"])  # the mode really was on
    av = np.loadtxt(tmp_path / "av_vels.dat", usecols=[1])
    want = gold[f"av_vels_{name}"]
    assert lbm.check_passes(want, av, 5.0), lbm.check_rule(want, av)
    assert lbm.check_passes(want[-100:], av[-100:], 1.0), lbm.check_rule(want[-100:], av[-100:])
    pressure = np.loadtxt(tmp_path / "final_state.dat", usecols=[5])
    assert lbm.check_passes(ref["pressure"].ravel(), pressure, 1.0), lbm.check_rule(ref["pressure"].ravel(), pressure)


@pytest.mark.parametrize("name", ["128x128", "128x256", "256x256", "1024x1024"])
@pytest.mark.parametrize("fuse", ["0", "1", "1-scalar", "default"])
def test_fast_mode_full_run_passes_check_rule(lbm, datasets, golden, monkeypatch, name, fuse):
    """LBM_MATH_FAST over the reference's full iteration counts -- with the one-step kernel (reciprocal + FMA), the
    stream kernels it is served by (packed exact collision), the scalar fast stream kernels (LBM_PACKED=0) and the
    default of these grids (the resident kernel): passes check.py's 1 % gate against the double-precision goldens
    and against SerialCode's fp32 output on all four reference grids."""
    if fuse != "default":
        monkeypatch.setenv("LBM_FUSE2", fuse[0])
    if fuse == "1-scalar":
        monkeypatch.setenv("LBM_PACKED", "0")
    p, ob = datasets(name)
    gold = np.load(os.path.join(golden, "check_goldens.npz"))
    ref = np.load(os.path.join(golden, f"serialcode_{name}.npz"))
    with lbm.Engine(p, ob, None, math="fast") as eng:
        eng.run(p.max_iters)
        av = eng.av_vels()
        pr = eng.final_state()["pressure"]
    assert lbm.check_passes(gold[f"av_vels_{name}"], av)
    if f"pressure_{name}" in gold:
        assert lbm.check_passes(gold[f"pressure_{name}"], pr)
    assert lbm.check_passes(ref["av_vels"], av) and lbm.check_passes(ref["pressure"], pr)
    d = lbm.check_rule(ref["av_vels"], av)
    assert abs(d["max_diff_pcnt"]) < 0.5       # measured: a few 1e-2 %


@pytest.fixture(scope="module")
def big_case(lbm):
    tile = lbm.read_obstacles(os.path.join(GOLDEN, "inputs", "obstacles_1024x1024.dat"), 1024, 1024)
    ob = lbm.tile_obstacles(tile, 8192, 8192)
    p = lbm.Params(8192, 8192, 64, 10, 0.1, 0.01, 1.85)
    return p, ob


@pytest.fixture(scope="module")
def big_oracle_50(oracle, big_case):
    """50 steps of the 8192x8192 case on the CPU oracle (multi-threaded fused form): planes, av_vels."""
    p, ob = big_case
    steps = 50
    src = np.empty((9, p.ny, p.nx), dtype=np.float32)
    w = np.float32(p.density)
    src[0] = w * np.float32(4.0) / np.float32(9.0)
    src[1:5] = w / np.float32(9.0)
    src[5:9] = w / np.float32(36.0)
    dst = np.empty_like(src)
    fluid = np.float32((ob == 0).sum())
    ref_av = []
    for _ in range(steps):
        ref_av.append(np.float32(oracle.fused_step_periodic(p, src, dst, ob)) / fluid)
        src, dst = dst, src
    del dst
    return steps, src, np.array(ref_av, dtype=np.float32)


@pytest.mark.parametrize("slabs", [1, 8])
def test_8192_short_run_bitwise_vs_oracle(lbm, big_case, big_oracle_50, monkeypatch, slabs):
    """BASELINE.json config 3 (synthetic 8192x8192, 1024x1024 obstacles tiled 8x8): 50 steps (the parity run
    SURVEY.md section 8d asks for) against the oracle's multi-threaded fused form; lattice and pressure
    bit-identical, av_vels by the check.py rule (big-grid parity through binary fields, not text files).
    slabs = 8: the same workload cut into the 8 row slabs config 4 gives its ranks (1024 rows each; here sharing
    the device, halos by device copies -- RCCL cannot open 8 communicators on one device): four-step passes,
    4-row halos, interior / boundary pipeline; 50 = 12 four-step passes + one two-step pass."""
    p, ob = big_case
    steps, src, ref_av = big_oracle_50
    monkeypatch.setenv("LBM_HALO", "memcpy")
    with lbm.Engine(p, ob, None, n_gpus=slabs) as eng:
        info = eng.info()
        assert info["n_slabs"] == slabs and info["steps_per_launch"] == 4
        eng.run(steps)
        got_av = eng.av_vels(steps)
        got = eng.cells()                      # (ny, nx, 9)
        fields = eng.final_state()
    for k in range(9):
        assert np.array_equal(got[:, :, k].view(np.uint32), src[k].view(np.uint32)), f"speed {k}"
    assert lbm.check_passes(ref_av, got_av)
    np.testing.assert_allclose(got_av, ref_av, rtol=2e-3)
    rho = src[0].copy()
    for k in range(1, 9):
        rho += src[k]                          # same left-to-right fp32 order as the reference
    want_p = np.where(ob == 1, np.float32(p.density) * np.float32(1.0 / 3.0), rho * np.float32(1.0 / 3.0))
    assert np.array_equal(fields["pressure"].view(np.uint32), want_p.astype(np.float32).view(np.uint32))
    assert not fields["u"][ob == 1].any()


def test_bench_rank_pipeline_checks_itself(lbm):
    """bench.py through the one-process-per-GPU code path on the 1-GPU box (world of one: lbm_create_rank_tiled,
    RCCL self-exchange, interior / boundary split): exactly one JSON line on stdout, whose multi_gpu_check says the
    rank's fields are bit-identical to a plain single-GPU run of the same workload."""
    import json
    import sys
    env = dict(os.environ, LBM_BENCH_RANK_API="1", LBM_FORCE_HALO="1", LBM_BENCH_ALSO="0", LBM_BENCH_REPEATS="2",
               LBM_BENCH_PREWARM_S="0.05")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "4096x2048", "--steps", "31",
                          "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 31 and rec["results_finite"] is True
    assert rec["multi_gpu_check"]["fields_bitwise_equal_to_single_gpu_run"] is True
    assert rec["multi_gpu_check"]["av_vels_max_rel_diff"] < 5e-5
    assert 0.0 < rec["roofline"]["frac"] <= 1.0 and rec["roofline"]["steps_per_launch"] == 4
    assert len(rec["timing"]["repeats_ms_per_step"]) == 2


def test_cli_8192_tiled_run_against_oracle_cli(lbm, oracle, tmp_path):
    """SURVEY.md section 8d parity run at BASELINE's full size, through both command lines: the
    C host program on the GPU vs the CPU oracle program on the same files (1024x1024 obstacle file
    tiled 8x8 with LBM_TILE), 24 steps; compared on the binary fp32 pressure dump (bit for bit) and
    on av_vels.dat (check.py rule) -- the 5.7 GB text file is not written."""
    import oracle_binding
    pf = tmp_path / "big.params"
    pf.write_text("8192\n8192\n24\n10\n0.1\n0.01\n1.85\n")
    of = os.path.join(GOLDEN, "inputs", "obstacles_1024x1024.dat")
    env = dict(os.environ, LBM_TILE="1024x1024", LBM_OUTPUT="none")
    (tmp_path / "gpu").mkdir(); (tmp_path / "cpu").mkdir()
    g = subprocess.run([lbm.CLI_PATH, str(pf), of], cwd=tmp_path / "gpu", capture_output=True, text=True,
                       env=dict(env, LBM_PRESSURE_BIN="pressure.bin"))
    assert g.returncode == 0, g.stderr
    c = subprocess.run([oracle_binding.CLI, str(pf), of], cwd=tmp_path / "cpu", capture_output=True, text=True,
                       env=dict(env, LBM_PRESSURE_BIN="pressure.bin", LBM_ORACLE_FORM="fused"))
    assert c.returncode == 0, c.stderr
    pg = np.fromfile(tmp_path / "gpu" / "pressure.bin", dtype=np.uint32)
    pc = np.fromfile(tmp_path / "cpu" / "pressure.bin", dtype=np.uint32)
    assert pg.size == 8192 * 8192 and np.array_equal(pg, pc)
    av_g = np.loadtxt(tmp_path / "gpu" / "av_vels.dat", usecols=[1])
    av_c = np.loadtxt(tmp_path / "cpu" / "av_vels.dat", usecols=[1])
    assert lbm.check_passes(av_c, av_g)
    np.testing.assert_allclose(av_g, av_c, rtol=2e-3)
    re_g = float(g.stdout.splitlines()[1].split()[-1])
    re_c = float(c.stdout.splitlines()[1].split()[-1])
    assert re_g == pytest.approx(re_c, rel=2e-3)


def test_8192_three_kernels_agree_bitwise_after_1001_steps(lbm, big_case, monkeypatch):
    """Size-independent cross-check at BASELINE's full size: the one-step kernel and the
    several-steps-per-pass stream kernels (four steps on pairs of cells = the default here, three steps, two
    steps with 4 and with 2 cells per lane) are independent implementations of the same arithmetic (different data flow,
    different neighbour exchange); after 1001 steps on 8192x8192 (1001 = 333 three-step passes + one
    two-step pass = 500 two-step passes + a one-step launch) their pressure and velocity fields must be
    bit-identical and mass conserved."""
    p0, ob = big_case
    p = lbm.Params(p0.nx, p0.ny, 1001, p0.reynolds_dim, p0.density, p0.accel, p0.omega)
    out = {}
    for tag, fuse, lane_cells, pass_steps, spl in (("one", "0", "4", "3", 1), ("three", "1", "4", "3", 3), ("four", "1", "4", "4", 4),
                                                   ("two4", "1", "4", "2", 2), ("two2", "1", "2", "2", 2)):
        monkeypatch.setenv("LBM_FUSE2", fuse)
        monkeypatch.setenv("LBM_LANE_CELLS", lane_cells)
        monkeypatch.setenv("LBM_PASS_STEPS", pass_steps)
        with lbm.Engine(p, ob, None) as eng:
            assert eng.info()["steps_per_launch"] == spl
            m0 = eng.total_density()
            eng.run(1001)
            assert eng.total_density() == pytest.approx(m0, rel=1e-6)
            f = eng.final_state()
            out[tag] = (f["pressure"].copy(), f["u"].copy(), eng.av_vels(1001))
    for other in ("four", "three", "two4", "two2"):
        assert np.array_equal(out["one"][0].view(np.uint32), out[other][0].view(np.uint32)), other
        assert np.array_equal(out["one"][1].view(np.uint32), out[other][1].view(np.uint32)), other
        # the stream kernels take |u| from the pre-collision moments (same lattice, ~1e-7 per cell): while the
        # flow is still tiny (|u| ~ 3e-7 in the first steps) the average moves by up to 1.4e-5 relative
        np.testing.assert_allclose(out["one"][2], out[other][2], rtol=5e-5)
    assert np.isfinite(out["three"][2]).all() and out["three"][2][-1] > out["three"][2][0] > 0


def test_8192_properties(lbm, big_case):
    p, ob = big_case
    with lbm.Engine(p, ob, None) as eng, lbm.Engine(p, ob, None) as split:
        m0 = eng.total_density()
        assert m0 == pytest.approx(0.1 * 8192 * 8192, rel=1e-6)
        eng.run(40)
        # mass is conserved (the reference's DEBUG invariant, SerialCode/d2q9-bgk.c:175-179)
        assert eng.total_density() == pytest.approx(m0, rel=1e-6)
        av = eng.av_vels(40)
        assert np.isfinite(av).all() and (av > 0).all() and (np.diff(av) > 0).all()
        # splitting the run changes nothing, bit for bit
        for n in (1, 7, 32):
            split.run(n)
        np.testing.assert_allclose(split.av_vels(40), av, rtol=5e-5)   # summation order, and |u| from pre- vs
        assert eng.av_velocity() == pytest.approx(float(av[-1]), rel=5e-5)   # post-collision moments by kernel
        a = eng.final_state()["pressure"]
        b = split.final_state()["pressure"]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_16384_slabs_and_kernels_agree_bitwise(lbm, monkeypatch):
    """BASELINE.json configs[4] (synthetic 16384x16384, 1024x1024 obstacles tiled 16x16), 18 GiB of
    lattices on one GPU: the grid cut into the 8 row slabs an 8-GPU run gives its ranks (here sharing the
    device; halos by device copies, interior/boundary pipeline) and the one-step kernel must both
    reproduce the single-slab two-step run bit for bit after an odd number of steps; mass conserved."""
    tile = lbm.read_obstacles(os.path.join(GOLDEN, "inputs", "obstacles_1024x1024.dat"), 1024, 1024)
    ob = lbm.tile_obstacles(tile, 16384, 16384)
    p = lbm.Params(16384, 16384, 41, 10, 0.1, 0.01, 1.85)
    monkeypatch.setenv("LBM_HALO", "memcpy")
    want = None
    for label, slabs, fuse in (("single slab, two-step", 1, "1"), ("8 slabs, two-step", 8, "1"),
                               ("single slab, one-step", 1, "0")):
        monkeypatch.setenv("LBM_FUSE2", fuse)
        with lbm.Engine(p, ob, None, n_gpus=slabs) as eng:
            m0 = eng.total_density()
            assert m0 == pytest.approx(0.1 * 16384 * 16384, rel=1e-6)
            eng.run(41)
            assert eng.total_density() == pytest.approx(m0, rel=1e-6), label
            f = eng.final_state()
            got = (f["pressure"], f["u_x"], f["u_y"], eng.av_vels(41))
        if want is None:
            want = got
            assert np.isfinite(got[3]).all() and (np.diff(got[3]) > 0).all()
            continue
        for k in range(3):
            assert np.array_equal(want[k].view(np.uint32), got[k].view(np.uint32)), (label, k)
        np.testing.assert_allclose(got[3], want[3], rtol=5e-5)
