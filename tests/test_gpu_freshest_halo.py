"""GPU tests of the freshest-available halo mode -- lbm_set_halo_mode(LBM_HALO_FRESHEST): the reference's own rule
(/root/reference/MPI_Testall_OptimizedVersion/d2q9-bgk.c:262-290: post the exchange, relax the interior rows, look
once, relax the boundary rows with whatever is there), made reproducible after the fact: the engine logs what every
look found (lbm_read_halo_log), and tests/slab_model.py replays the run from that log on the CPU oracle.

  * forced looks: LBM_FRESH_FORCE=wait (every look finds its rows) is the single-domain oracle bit for bit,
    LBM_FRESH_FORCE=never (no rows are sent ahead) is the stale mode bit for bit -- the selection logic, pinned;
  * free-running: whatever the looks found, the lattice equals the model's under the logged schedule, bit for bit;
  * the result lies within the stale mode's distance of the synchronous run (check.py's rule, 2 slabs).
Parity unpinned against the reference (its variant is not reproducible: no fixture can exist)."""
import numpy as np
import pytest

import slab_model
from test_gpu_parity import AV_RTOL, CHECK_TOL_PCT, random_case

pytestmark = pytest.mark.gpu


def engine_freshest(lbm, p, ob, cells, calls, n_gpus):
    with lbm.Engine(p, ob, cells, n_gpus=n_gpus) as eng:
        eng.set_halo_mode("freshest")
        assert eng.info()["halo_mode"] == lbm.HALO_FRESHEST
        for n in calls:
            eng.run(n)
        return eng.cells(), eng.av_vels(sum(calls)), eng.halo_log(sum(calls))


def setup(monkeypatch, slabs, halo):
    monkeypatch.setenv("LBM_HALO", halo)
    monkeypatch.setenv("LBM_VEC4", "1")
    if slabs == 1:
        monkeypatch.setenv("LBM_FORCE_HALO", "1")


CASES = [(2, "memcpy", [9]), (3, "memcpy", [4, 7]), (8, "memcpy", [12]), (1, "rccl", [7, 2])]


@pytest.mark.parametrize("slabs,halo,calls", CASES)
def test_every_look_fresh_is_the_single_domain_run(lbm, oracle, monkeypatch, slabs, halo, calls):
    setup(monkeypatch, slabs, halo)
    monkeypatch.setenv("LBM_FRESH_FORCE", "wait")
    p, ob, cells = random_case(lbm, 256, 37, 15 + slabs, walls=False)
    ref = cells.copy()
    ref_av = oracle.run(p, ref, ob, sum(calls))
    got, got_av, log = engine_freshest(lbm, p, ob, cells, calls, slabs)
    assert (log == 3).all()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    np.testing.assert_allclose(got_av, ref_av, rtol=AV_RTOL)


@pytest.mark.parametrize("slabs,halo,calls", CASES)
def test_no_look_fresh_is_the_stale_mode(lbm, oracle, monkeypatch, slabs, halo, calls):
    setup(monkeypatch, slabs, halo)
    monkeypatch.setenv("LBM_FRESH_FORCE", "never")
    p, ob, cells = random_case(lbm, 256, 37, 25 + slabs, walls=False)
    want, want_tot = slab_model.run_slabs(oracle, lbm, p, ob, calls, slabs, lag=1, two_step=False, cells=cells)
    got, got_av, log = engine_freshest(lbm, p, ob, cells, calls, slabs)
    first = np.cumsum([0] + calls[:-1])
    assert (log[first] == 3).all() and (np.delete(log, first, axis=0) == 0).all()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    np.testing.assert_allclose(got_av, (want_tot / np.float64((ob == 0).sum())).astype(np.float32), rtol=AV_RTOL)


@pytest.mark.parametrize("nx,ny,slabs,halo,calls", [(256, 37, 2, "memcpy", [30]), (256, 37, 3, "memcpy", [11, 14]),
                                                    (128, 96, 8, "memcpy", [25]), (256, 37, 1, "rccl", [9, 6]),
                                                    (1024, 600, 2, "memcpy", [12]), (2048, 1024, 4, "memcpy", [8])])
def test_free_running_equals_the_model_under_the_logged_schedule(lbm, oracle, monkeypatch, nx, ny, slabs, halo, calls):
    """Nothing forced: each look finds what it finds.  Small slabs (an interior pass of a few microseconds) mostly see
    the rows of the step before, tall ones mostly this step's; whatever the mix, the log explains the lattice."""
    setup(monkeypatch, slabs, halo)
    p, ob, cells = random_case(lbm, nx, ny, 35 + slabs, walls=False)
    got, got_av, log = engine_freshest(lbm, p, ob, cells, calls, slabs)
    assert log.shape == (sum(calls), slabs) and (log <= 3).all()
    want, want_tot, _, _ = slab_model.run_slabs_freshest(oracle, lbm, p, ob, calls, slabs, schedule=log, cells=cells)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    np.testing.assert_allclose(got_av, (want_tot / np.float64((ob == 0).sum())).astype(np.float32), rtol=AV_RTOL)
    print(f"fresh share of the looks: {np.mean([(log[1:] & 1).mean(), ((log[1:] >> 1) & 1).mean()]):.2f}")


@pytest.mark.parametrize("slabs,halo,calls", [(2, "memcpy", [24]), (3, "memcpy", [10, 9]), (8, "memcpy", [16]), (1, "rccl", [12, 5])])
def test_mixed_schedules_equal_the_model(lbm, oracle, monkeypatch, slabs, halo, calls):
    """On this box the rows always beat the look (slabs share one device and the host is the slow part).  With about
    half of the exchanges held back on their comm streams (LBM_FRESH_TEST_DELAY_US) the looks really find both -- per
    side, per slab, per step -- and the lattice is still the model's under the logged schedule."""
    setup(monkeypatch, slabs, halo)
    monkeypatch.setenv("LBM_FRESH_TEST_DELAY_US", "3000")
    p, ob, cells = random_case(lbm, 256, 40, 45 + slabs, walls=False)
    got, got_av, log = engine_freshest(lbm, p, ob, cells, calls, slabs)
    inner = np.delete(log, np.cumsum([0] + calls[:-1]), axis=0)
    # (a ring of one over RCCL never catches up once it is behind: its send/recv pair takes longer than the interior
    # rows of this small slab, and after the first delay the host is far ahead of the device)
    assert (inner != 3).any() and ((inner != 0).any() or halo == "rccl"), log.T
    want, want_tot, _, _ = slab_model.run_slabs_freshest(oracle, lbm, p, ob, calls, slabs, schedule=log, cells=cells)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    np.testing.assert_allclose(got_av, (want_tot / np.float64((ob == 0).sum())).astype(np.float32), rtol=AV_RTOL)
    stale, _ = slab_model.run_slabs(oracle, lbm, p, ob, calls, slabs, lag=1, two_step=False, cells=cells)
    assert np.array_equal(got.view(np.uint32), stale.view(np.uint32)) == (not (inner != 0).any())
    print(f"schedule (rows: slabs, columns: steps):\n{log.T}")


def test_full_run_within_the_stale_modes_distance_of_the_synchronous_run(lbm, datasets, monkeypatch):
    """128x128 over 2 slabs, 2000 steps: av_vels inside check.py's 1 % of the synchronous run (the stale mode, the
    worst case of this one, measures 0.58 % here) and total density equal to 1e-5."""
    monkeypatch.setenv("LBM_HALO", "memcpy")
    p, ob = datasets("128x128")
    steps = 2000
    with lbm.Engine(p, ob, None, n_gpus=2) as sync:
        sync.run(steps)
        ref_av, ref_mass = sync.av_vels(steps), sync.total_density()
    with lbm.Engine(p, ob, None, n_gpus=2) as eng:
        eng.set_halo_mode("freshest")
        eng.run(steps)
        av, mass = eng.av_vels(steps), eng.total_density()
        log = eng.halo_log(steps)
    assert np.isfinite(av).all()
    assert np.max(np.abs(av - ref_av) / np.abs(ref_av)) * 100 < CHECK_TOL_PCT
    assert abs(mass - ref_mass) / ref_mass < 1e-5
    assert (log[0] == 3).all()


def test_mode_switches(lbm, oracle, monkeypatch):
    """sync -> freshest -> stale -> sync in one context: every switch starts from freshly exchanged halos."""
    setup(monkeypatch, 2, "memcpy")
    monkeypatch.setenv("LBM_FRESH_FORCE", "wait")
    p, ob, cells = random_case(lbm, 128, 24, 91, walls=False)
    ref = cells.copy()
    oracle.run(p, ref, ob, 4 + 6 + 5)
    with lbm.Engine(p, ob, cells, n_gpus=2) as eng:
        eng.run(4)
        eng.set_halo_mode("freshest")
        eng.run(6)
        eng.set_halo_mode("sync")
        eng.run(5)
        assert np.array_equal(eng.cells().view(np.uint32), ref.view(np.uint32))
        with pytest.raises(lbm.LbmError, match="unknown mode"):
            lbm._check(eng.lib, eng.lib.lbm_set_halo_mode(eng.handle, 7))


def test_hosted_exchange_refuses_the_mode(lbm, monkeypatch):
    """The MPI-callback transport blocks the host in every exchange: there is nothing to poll.  Asking for the mode there
    is an error at once, not a silent synchronous run."""
    monkeypatch.setenv("LBM_FORCE_HALO", "1")
    p, ob, cells = random_case(lbm, 128, 24, 7, walls=False)

    def exchange(plan, bufs):       # a ring of one: what goes north comes back as the south halo and vice versa
        bufs[2][:] = bufs[0]
        bufs[3][:] = bufs[1]

    def allreduce(values):
        pass
    with lbm.Engine(p, ob, cells, rank=0, world_size=1, device=0, host_comm=(exchange, allreduce)) as eng:
        with pytest.raises(lbm.LbmError, match="hosted exchange"):
            eng.set_halo_mode("freshest")
        eng.run(3)                  # the context is still good, in the synchronous mode
        assert eng.info()["halo_mode"] == lbm.HALO_SYNC
