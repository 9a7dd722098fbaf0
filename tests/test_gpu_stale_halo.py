"""GPU tests of the stale-halo ("asynchronous") mode -- lbm_set_halo_mode(LBM_HALO_STALE).

The reference's research variant (/root/reference/MPI_Testall_OptimizedVersion/d2q9-bgk.c:256-301)
polls MPI_Testall and relaxes its boundary rows with whatever halo rows are there; its results vary
from run to run.  The engine pins the staleness to exactly one pass, which makes it reproducible and
gives it an exact oracle: tests/slab_model.py (lag = 1) replays the same protocol on the CPU oracle.

  * bit-exact against that model on random lattices (where a one-pass-old halo changes every
    boundary cell), memcpy and RCCL transports, several lbm_run calls, with the synchronous pipeline
    configured for the one-step or the two-step kernel (stale passes always advance one timestep);
  * the accuracy claim the reference makes for stale halos -- results stay within check.py's 1 %
    rule of the synchronous run -- on a full reference data set;
  * mass: a population crossing a slab boundary is delayed, never lost or duplicated beyond what is
    in flight, so the total density stays within a boundary-row's worth of the synchronous run's.
"""
import numpy as np
import pytest

import slab_model
from test_gpu_parity import AV_RTOL, CHECK_TOL_PCT, random_case

pytestmark = pytest.mark.gpu


def engine_stale(lbm, p, ob, cells, calls, n_gpus):
    with lbm.Engine(p, ob, cells, n_gpus=n_gpus) as eng:
        eng.set_halo_mode("stale")
        assert eng.info()["halo_mode"] == lbm.HALO_STALE
        for n in calls:
            eng.run(n)
        return eng.cells(), eng.av_vels(sum(calls))


@pytest.mark.parametrize("fuse,band,lane_cells", [(0, 6, 4), (1, 3, 4), (1, 5, 2), (1, 64, 4)])
@pytest.mark.parametrize("slabs,halo,calls", [(2, "memcpy", [9]), (3, "memcpy", [4, 7]), (4, "memcpy", [12]),
                                              (1, "rccl", [7, 2])])
def test_stale_bitwise_against_slab_model(lbm, oracle, monkeypatch, fuse, band, lane_cells, slabs, halo, calls):
    monkeypatch.setenv("LBM_FUSE2", str(fuse))
    monkeypatch.setenv("LBM_BAND_ROWS", str(band))
    monkeypatch.setenv("LBM_LANE_CELLS", str(lane_cells))
    monkeypatch.setenv("LBM_HALO", halo)
    monkeypatch.setenv("LBM_VEC4", "1")
    if slabs == 1:
        monkeypatch.setenv("LBM_FORCE_HALO", "1")
    p, ob, cells = random_case(lbm, 256, 37, 5 + slabs, walls=False)
    want, want_tot = slab_model.run_slabs(oracle, lbm, p, ob, calls, slabs, lag=1, two_step=bool(fuse), cells=cells)
    got, got_av = engine_stale(lbm, p, ob, cells, calls, slabs)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    want_av = (want_tot / np.float64((ob == 0).sum())).astype(np.float32)
    np.testing.assert_allclose(got_av, want_av, rtol=AV_RTOL)
    # the lag is real on this input: the synchronous protocol gives a different lattice
    sync, _ = slab_model.run_slabs(oracle, lbm, p, ob, calls, slabs, lag=0, two_step=bool(fuse), cells=cells)
    if sum(calls) > 2 * len(calls):
        assert not np.array_equal(sync.view(np.uint32), want.view(np.uint32))


def test_stale_mode_scalar_kernel_ragged_width(lbm, oracle, monkeypatch):
    monkeypatch.setenv("LBM_HALO", "memcpy")
    p, ob, cells = random_case(lbm, 101, 19, 3, walls=False)     # nx % 4 != 0: one cell per lane
    want, _ = slab_model.run_slabs(oracle, lbm, p, ob, [11], 3, lag=1, two_step=False, cells=cells)
    got, _ = engine_stale(lbm, p, ob, cells, [11], 3)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_stale_mode_rank_api(lbm, oracle, monkeypatch):
    """lbm_create_rank + stale mode: the RCCL send/recv pair that targets the other lattice buffer."""
    monkeypatch.setenv("LBM_FORCE_HALO", "1")
    monkeypatch.setenv("LBM_FUSE2", "1")
    p, ob, cells = random_case(lbm, 128, 40, 17, walls=False)
    want, _ = slab_model.run_slabs(oracle, lbm, p, ob, [13], 1, lag=1, two_step=True, cells=cells)
    with lbm.Engine(p, ob, cells, rank=0, world_size=1, unique_id=lbm.rccl_unique_id(), device=0) as eng:
        eng.set_halo_mode("stale")
        eng.run(13)
        assert np.array_equal(eng.cells().view(np.uint32), want.view(np.uint32))


def test_switching_modes_between_runs(lbm, oracle, monkeypatch):
    """sync -> stale -> sync on one context: each call starts from fresh halos, so the model is the
    concatenation of the three protocols."""
    monkeypatch.setenv("LBM_HALO", "memcpy")
    monkeypatch.setenv("LBM_FUSE2", "1")
    p, ob, cells = random_case(lbm, 128, 30, 23, walls=False)
    a, _ = slab_model.run_slabs(oracle, lbm, p, ob, [6], 3, lag=0, cells=cells)
    b, _ = slab_model.run_slabs(oracle, lbm, p, ob, [7], 3, lag=1, cells=a)
    c, _ = slab_model.run_slabs(oracle, lbm, p, ob, [5], 3, lag=0, cells=b)
    with lbm.Engine(p, ob, cells, n_gpus=3) as eng:
        eng.run(6)
        eng.set_halo_mode("stale")
        eng.run(7)
        eng.set_halo_mode("sync")
        eng.run(5)
        assert np.array_equal(eng.cells().view(np.uint32), c.view(np.uint32))


@pytest.mark.parametrize("slabs,tol_pct", [
    (2, CHECK_TOL_PCT),
    pytest.param(8, CHECK_TOL_PCT, marks=pytest.mark.xfail(strict=True, reason=(
        "EXPERIMENTAL mode, parity unpinned: with 8 slabs of 16 rows the one-step halo lag moves av_vels by up to "
        "4.0 % mid-transient (0.5 % at the end) -- outside check.py's 1 % rule, which takes the maximum over steps"))),
    (8, 5.0)])
def test_stale_halo_full_run_against_synchronous(lbm, datasets, monkeypatch, slabs, tol_pct):
    """The reference's accuracy claim for stale halos, on its 128x128 data set at full length: av_vels
    and pressure against the synchronous run by the check.py rule (check/check.py:83-99,136-148).
    A one-step delay per slab boundary leaves steady states alone but stretches the transient: with
    2 slabs the run stays inside the 1 % rule; with 8 slabs of 16 rows (one boundary every 16 rows,
    far denser than any sensible decomposition) av_vels deviate by up to 4 % mid-transient and 0.5 %
    at the end: the 1 % rule is kept as an expected failure there, and the measured deviation is
    bounded at 5 %.  Mass is conserved up to what is in flight."""
    monkeypatch.setenv("LBM_HALO", "memcpy")
    p, ob = datasets("128x128")
    out = {}
    for mode in ("sync", "stale"):
        with lbm.Engine(p, ob, None, n_gpus=slabs) as eng:
            eng.set_halo_mode(mode)
            eng.run(p.max_iters)
            out[mode] = (eng.av_vels(p.max_iters), eng.final_state()["pressure"], eng.total_density())
    assert lbm.check_passes(out["sync"][0], out["stale"][0], tol_pct), lbm.check_rule(out["sync"][0], out["stale"][0])
    assert lbm.check_passes(out["sync"][1], out["stale"][1], tol_pct), lbm.check_rule(out["sync"][1], out["stale"][1])
    assert out["stale"][2] == pytest.approx(out["sync"][2], rel=1e-4)
    print("stale vs sync", slabs, "slabs:", lbm.check_rule(out["sync"][0], out["stale"][0])["max_diff_pcnt"],
          lbm.check_rule(out["sync"][1], out["stale"][1])["max_diff_pcnt"], out["stale"][2] / out["sync"][2] - 1)


def test_set_halo_mode_rejects_unknown(lbm, datasets):
    p, ob = datasets("128x128")
    with lbm.Engine(p, ob, None) as eng:
        with pytest.raises(lbm.LbmError):
            lbm._check(eng.lib, eng.lib.lbm_set_halo_mode(eng.handle, 7))
        eng.set_halo_mode("stale")          # accepted (no effect) on a single periodic slab
        eng.run(3)
