"""The resident-lattice kernel (lbm::resident_band): one launch per lbm_run call, the lattice in registers, seam rows
between 4-row bands through tagged L2 granules.  Default for single periodic slabs of at most 1024 x 4*CUs cells and
calls long enough to pay for a launch (4 steps; 8 below 48 Ki cells) -- i.e. for the reference's own data sets.  Lattice bit-identical to the oracle."""
import numpy as np
import pytest

from test_gpu_parity import AV_RTOL, random_case

pytestmark = pytest.mark.gpu


def run_resident(lbm, oracle, p, ob, cells, calls, math="exact"):
    ref = cells.copy()
    ref_av = oracle.run(p, ref, ob, sum(calls))
    with lbm.Engine(p, ob, cells, math=math) as eng:
        info = eng.info()
        assert info["resident_steps"] > 0, info
        for n in calls:
            eng.run(n)
        got = eng.cells()
        got_av = eng.av_vels(sum(calls))
        fields = eng.final_state()
    return ref, ref_av, got, got_av, fields


@pytest.mark.parametrize("name,calls", [("128x128", [37]), ("128x128", [16, 1, 40, 3, 17]), ("128x256", [300]),
                                        ("256x256", [50]), ("1024x1024", [24])])
def test_resident_reference_datasets_bitwise(lbm, oracle, datasets, monkeypatch, name, calls):
    """The reference's data sets: calls of >= 16 steps run resident here (the default threshold is 4 steps, 8 below 48 Ki cells), shorter ones launch per pass; the lid row, the walls and the periodic seam between
    the last and the first band are all live."""
    monkeypatch.setenv("LBM_RESIDENT_MIN_STEPS", "16")
    p, ob = datasets(name)
    cells = oracle.init_cells(p)
    ref, ref_av, got, got_av, fields = run_resident(lbm, oracle, p, ob, cells, calls)
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    np.testing.assert_allclose(got_av, ref_av, rtol=AV_RTOL)
    want = oracle.final_state(p, ref, ob)
    assert np.array_equal(fields["pressure"].view(np.uint32), want["pressure"].view(np.uint32))


@pytest.mark.parametrize("nx,ny,env", [(64, 8, {}), (64, 12, {"LBM_RESIDENT_JOINT": "0"}), (128, 16, {}), (192, 40, {}),
                                       (256, 64, {"LBM_RESIDENT_JOINT": "0"}), (320, 8, {}), (320, 24, {"LBM_RESIDENT_JOINT": "1"}),
                                       (512, 64, {}), (512, 32, {"LBM_RESIDENT_JOINT": "1"}), (704, 20, {}), (1024, 64, {}),
                                       (1024, 1024, {}), (960, 36, {}), (1024, 128, {"LBM_RESIDENT_XCD": "0"}),
                                       (128, 64, {"LBM_RESIDENT_XCD": "0"}), (192, 36, {}),
                                       (128, 128, {"LBM_RESIDENT_ROWS": "4"}), (256, 256, {"LBM_RESIDENT_ROWS": "2"}),
                                       (64, 6, {}), (320, 30, {}), (512, 512, {"LBM_RESIDENT_ROWS": "2"}),
                                       (1024, 512, {"LBM_RESIDENT_ROWS": "2"}), (448, 250, {"LBM_RESIDENT_ROWS": "2", "LBM_RESIDENT_XCD": "0"}),
                                       (128, 128, {}), (128, 128, {"LBM_RESIDENT_ONE_XCD": "0"}), (128, 128, {"LBM_RESIDENT_ONE_XCD": "0", "LBM_RESIDENT_GROUP": "4"}),
                                       (64, 64, {}), (64, 256, {}), (64, 64, {"LBM_RESIDENT_GROUP": "8"}), (256, 64, {}), (256, 64, {"LBM_RESIDENT_GROUP": "2"}),
                                       (128, 64, {"LBM_RESIDENT_GROUP": "2", "LBM_RESIDENT_XCD": "0"}), (64, 12, {"LBM_RESIDENT_GROUP": "2"})])
def test_resident_random_lattices_bitwise(lbm, oracle, monkeypatch, nx, ny, env):
    """Random lattices with random obstacles (also on the seam rows and at the wave edges), both periodic wraps live,
    widths of 1 to 16 waves -- full and partly filled workgroups --, 2 to 256 bands of four or two rows, 1 to 8 bands per
    workgroup, on one XCD (the default up to 128 waves) or dealt over all eight; every call resident.  Both orders
    of a step (interior pair before the halo wait / both pairs together behind it), seams kept in an XCD's L2 where
    both bands run on it and written through everywhere (band counts that are and are not multiples of 8)."""
    monkeypatch.setenv("LBM_RESIDENT_MIN_STEPS", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    p, ob, cells = random_case(lbm, nx, ny, nx * 3 + ny, blocked_frac=0.04, walls=False)
    calls = [1, 2, 19, 5] if nx * ny <= 300000 else [1, 6]
    ref, ref_av, got, got_av, _ = run_resident(lbm, oracle, p, ob, cells, calls)
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    np.testing.assert_allclose(got_av, ref_av, rtol=AV_RTOL)


def test_resident_long_run_spans_launches(lbm, oracle):
    """More timesteps than one launch advances (4096): the epoch tags go on across launches, the partial sums of every
    step land in av_vels."""
    p, ob, cells = random_case(lbm, 64, 8, 77, blocked_frac=0.05, walls=False)
    p.max_iters = 9000
    ref, ref_av, got, got_av, _ = run_resident(lbm, oracle, p, ob, cells, [4100, 4500])
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    np.testing.assert_allclose(got_av, ref_av, rtol=AV_RTOL)


def test_resident_guard_paths_bitwise(lbm, oracle):
    """Cells outside the fast divides' range (tiny / huge densities, |u| >> 1) take the IEEE path per cell, also
    here (same lattice as test_guard_paths_stay_bit_exact)."""
    p, ob, cells = random_case(lbm, 256, 40, 4242, walls=False)
    rng = np.random.default_rng(7)
    cells[5:9, 10:40] *= np.float32(1e-22)
    cells[20:23, 100:180] *= np.float32(1e20)
    for _ in range(12):
        y, x = int(rng.integers(0, 40)), int(rng.integers(0, 256))
        cells[y, x, 1] *= np.float32(1e6)
    ref, ref_av, got, got_av, _ = run_resident(lbm, oracle, p, ob, cells, [20])
    assert np.isfinite(ref).all()
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))


def test_resident_fast_mode_uses_the_exact_kernel(lbm, oracle, datasets, monkeypatch):
    """LBM_MATH_FAST on these grids is served by the resident kernel (exact arithmetic: the faster kernel here, and
    trivially inside the fast mode's tolerance)."""
    monkeypatch.setenv("LBM_RESIDENT_MIN_STEPS", "16")
    p, ob = datasets("128x128")
    cells = oracle.init_cells(p)
    ref, ref_av, got, got_av, _ = run_resident(lbm, oracle, p, ob, cells, [64], math="fast")
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))


def test_resident_policy(lbm, monkeypatch):
    """Eligibility: one lane per cell of a row (nx a multiple of 64 up to 1024), 4-row bands, at most one band per CU;
    asking for another kernel switches it off."""
    def resident(nx, ny, **env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        try:
            with lbm.Engine(lbm.Params(nx, ny, 4, 10, 0.1, 0.005, 1.85), np.zeros((ny, nx), dtype=np.int32), None) as eng:
                return eng.info()["resident_steps"] > 0
        finally:
            for k in env:
                monkeypatch.delenv(k)
    assert resident(128, 128) and resident(1024, 1024) and resident(64, 8) and resident(512, 512)
    assert resident(128, 130) and resident(128, 4)     # bands of two rows where the chip has CUs to spare
    assert not resident(100, 128) and not resident(2048, 64) and not resident(128, 131) and not resident(128, 2)
    assert not resident(1024, 1022)                    # 1024 wide: four-row bands only
    assert not resident(1024, 2048)                    # 512 bands: more than the device has CUs
    assert not resident(128, 128, LBM_FUSE2="1") and not resident(128, 128, LBM_TILE_STEPS="4")

    def placement(nx, ny):
        with lbm.Engine(lbm.Params(nx, ny, 4, 10, 0.1, 0.005, 1.85), np.zeros((ny, nx), dtype=np.int32), None) as eng:
            i = eng.info()
            return i["resident_rows"], i["resident_group"], i["resident_one_xcd"]
    # up to 128 waves: one XCD, workgroups of four waves, at most 32 of them
    assert placement(128, 128) == (2, 2, 1) and placement(64, 128) == (2, 2, 1) and placement(64, 256) == (2, 4, 1)
    assert placement(256, 64) == (2, 1, 1) and placement(128, 64) == (2, 1, 1)
    assert placement(128, 256) == (2, 1, 0) and placement(256, 256) == (2, 1, 0) and placement(1024, 1024) == (4, 1, 0)
    assert resident(128, 128, LBM_FUSE2="1", LBM_RESIDENT="1") and not resident(128, 128, LBM_RESIDENT="0")


def test_resident_gives_up_instead_of_hanging(lbm, datasets, monkeypatch):
    """A band whose workgroup never runs (simulated) starves its neighbours: they give up after the bound, the status
    reaches the host as an error from the next call that synchronises -- an error, never a hang."""
    import time
    monkeypatch.setenv("LBM_RESIDENT_MIN_STEPS", "16")
    monkeypatch.setenv("LBM_RESIDENT_ABSENT_BAND", "5")
    monkeypatch.setenv("LBM_RESIDENT_TIMEOUT_MS", "200")
    p, ob = datasets("128x128")
    with lbm.Engine(p, ob, None) as eng:
        t0 = time.time()
        eng.run(100)
        with pytest.raises(lbm.LbmError, match="resident kernel gave up"):
            eng.sync()
        assert time.time() - t0 < 20
