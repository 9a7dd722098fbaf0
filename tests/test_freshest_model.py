"""CPU: the model of the freshest-available halo mode (tests/slab_model.py: run_slabs_freshest) against its two
limits -- every look finds its rows = the synchronous protocol = the single-domain oracle, bit for bit; no look does =
the stale protocol (lag = 1), bit for bit -- and the two properties the engine's header claims for everything in
between: mass is conserved as well as in the limits (a halo row that is one step old delays the populations crossing
the seam, it does not lose or duplicate them), and the deviation from the synchronous run shrinks with the share of
fresh rows.  The engine itself is checked against this model, schedule by schedule, in tests/test_gpu_freshest_halo.py."""
import numpy as np
import pytest

import slab_model


def test_limits_are_the_synchronous_and_the_stale_protocol(lbm, oracle, datasets):
    p, ob = datasets("128x128")
    calls = [7, 12]
    ref = oracle.init_cells(p)
    oracle.run(p, ref, ob, sum(calls))
    for slabs in (2, 3):
        fresh, _, _, sched = slab_model.run_slabs_freshest(oracle, lbm, p, ob, calls, slabs, p_fresh=1.0)
        assert (sched == 3).all()
        assert np.array_equal(fresh.view(np.uint32), ref.view(np.uint32))
        late, late_u, _, _ = slab_model.run_slabs_freshest(oracle, lbm, p, ob, calls, slabs, p_fresh=0.0)
        want, want_u = slab_model.run_slabs(oracle, lbm, p, ob, calls, slabs, lag=1, two_step=False)
        assert np.array_equal(late.view(np.uint32), want.view(np.uint32))
        np.testing.assert_array_equal(late_u, want_u)


def test_mass_and_deviation_between_the_limits(lbm, oracle, datasets):
    p, ob = datasets("128x128")
    steps, slabs = 400, 8
    ref = oracle.init_cells(p)
    ref_av = oracle.run(p, ref, ob, steps)
    n_fluid = np.float64((ob == 0).sum())
    dev, drift = {}, {}
    for pf in (1.0, 0.9, 0.5, 0.0):
        _, tot_u, mass, _ = slab_model.run_slabs_freshest(oracle, lbm, p, ob, [steps], slabs, p_fresh=pf, seed=3)
        dev[pf] = float(np.max(np.abs(tot_u / n_fluid - ref_av) / np.abs(ref_av)))
        drift[pf] = (mass[-1] - mass[0]) / mass[0]
    # accelerate_flow adds mass-neutral momentum only; what drifts is fp32 rounding, the same in every variant
    for pf in (0.9, 0.5, 0.0):
        assert abs(drift[pf] - drift[1.0]) < 2e-6, drift
    assert dev[1.0] < 1e-4      # the same lattices; the sums of |u| are taken in another order
    assert dev[1.0] < dev[0.9] < dev[0.5] < dev[0.0], dev
