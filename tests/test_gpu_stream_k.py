"""GPU parity of round 2's additions: the K-timesteps-per-pass stream kernel (stepk_stream: K = 2, 3; next-row
prefetch; XCD-chunked workgroup order), and the device-side mask construction (global map, per-rank rows,
periodic tile) with its device-side fluid-cell count.  Everything through the C ABI against the CPU oracle;
the lattice must stay bit-identical (see test_gpu_parity.py for the av_vels tolerance)."""
import numpy as np
import pytest

from test_gpu_parity import AV_RTOL, random_case, run_both

pytestmark = pytest.mark.gpu


def force_stream(monkeypatch, k, band, prefetch=0, chunk=0, stepk=1, packed=None):
    if packed is not None:
        # packed: 0 = scalar stream kernel; 1 = collision on pairs; 11 / 12 = pairs, 1 / 2 sliding windows in LDS
        monkeypatch.setenv("LBM_PACKED", "1" if packed else "0")
        monkeypatch.setenv("LBM_LDS_WINDOWS", str(packed % 10 if packed > 1 else 0))
    monkeypatch.setenv("LBM_FUSE2", "1")
    monkeypatch.setenv("LBM_LANE_CELLS", "4")
    monkeypatch.setenv("LBM_PASS_STEPS", str(k))
    monkeypatch.setenv("LBM_BAND_ROWS", str(band))
    monkeypatch.setenv("LBM_PREFETCH", str(prefetch))
    monkeypatch.setenv("LBM_XCD_CHUNK", str(chunk))
    monkeypatch.setenv("LBM_STEPK", str(stepk))


# (timesteps per pass, arithmetic: 0 scalar / 1 pairs / 11, 12 pairs with 1, 2 windows in LDS, band height, prefetch, XCD chunk)
STREAM_VARIANTS = [(2, 0, 2, 0, 0), (2, 0, 7, 1, 3), (3, 0, 5, 0, 3), (3, 0, 64, 1, 1), (4, 0, 3, 0, 2), (4, 0, 7, 1, 0),
                   (2, 1, 3, 1, 2), (2, 11, 5, 0, 0), (3, 1, 2, 0, 0), (3, 1, 7, 1, 0), (3, 12, 5, 0, 3), (3, 12, 64, 1, 1),
                   (4, 1, 2, 0, 0), (4, 1, 5, 0, 3), (4, 11, 7, 1, 0), (4, 11, 3, 0, 2), (4, 12, 2, 1, 0), (4, 12, 7, 1, 0),
                   (4, 12, 64, 1, 1), (4, 12, 5, 1, 3)]


@pytest.mark.parametrize("k,packed,band,prefetch,chunk", STREAM_VARIANTS)
@pytest.mark.parametrize("slabs,halo", [(1, None), (1, "rccl"), (2, "memcpy"), (3, "memcpy"), (8, "memcpy")])
def test_k_steps_per_pass_bitwise(lbm, oracle, datasets, monkeypatch, k, packed, band, prefetch, chunk, slabs, halo):
    """Reference data set 128x256 (periodic wrap in y live, wall row in the middle); 76 = 25 three-step passes
    + 1 single step, or 38 two-step passes; 77 ends a three-step run with a two-step pass."""
    force_stream(monkeypatch, k, band, prefetch, chunk, packed=packed)
    if halo:
        monkeypatch.setenv("LBM_HALO", halo)
        if slabs == 1:
            monkeypatch.setenv("LBM_FORCE_HALO", "1")
    p, ob = datasets("128x256")
    cells = oracle.init_cells(p)
    for steps in (76, 77, 79):
        ref_cells, ref_av, got_cells, got_av, fields = run_both(lbm, oracle, p, ob, cells, steps, n_gpus=slabs)
        assert np.array_equal(ref_cells.view(np.uint32), got_cells.view(np.uint32)), steps
        np.testing.assert_allclose(got_av, ref_av, rtol=AV_RTOL)
        ref_f = oracle.final_state(p, ref_cells, ob)
        assert np.array_equal(ref_f["pressure"].view(np.uint32), fields["pressure"].view(np.uint32))


@pytest.mark.parametrize("k,band,prefetch,lds", [(2, 2, 0, 0), (2, 5, 0, 1), (2, 8, 1, 1), (3, 3, 0, 0), (3, 5, 1, 0), (3, 64, 0, 1),
                                                 (3, 2, 1, 1), (4, 4, 0, 0), (4, 7, 1, 1)])
@pytest.mark.parametrize("slabs,halo", [(1, None), (1, "rccl"), (3, "memcpy"), (8, "memcpy")])
def test_two_cell_packed_kernel_bitwise(lbm, oracle, datasets, monkeypatch, k, band, prefetch, lds, slabs, halo):
    """stepk_pk with ONE pair per lane (two cells; two halo lanes per side beyond two steps per pass): the kernel of
    mid-size grids (three steps per pass by default), also at two and four steps."""
    force_stream(monkeypatch, k, band, prefetch, 0, packed=1)
    monkeypatch.setenv("LBM_LANE_CELLS", "2")
    monkeypatch.setenv("LBM_LDS_WINDOWS", str(lds))
    if halo:
        monkeypatch.setenv("LBM_HALO", halo)
        if slabs == 1:
            monkeypatch.setenv("LBM_FORCE_HALO", "1")
    p, ob = datasets("128x256")
    cells = oracle.init_cells(p)
    for steps in (76, 77, 79):
        ref_cells, ref_av, got_cells, got_av, _ = run_both(lbm, oracle, p, ob, cells, steps, n_gpus=slabs)
        assert np.array_equal(ref_cells.view(np.uint32), got_cells.view(np.uint32)), steps
        np.testing.assert_allclose(got_av, ref_av, rtol=AV_RTOL)
    p, ob, cells = random_case(lbm, 250 * 2, 67, 77, walls=False)       # ragged strips, both wraps live
    ref_cells, ref_av, got_cells, got_av, _ = run_both(lbm, oracle, p, ob, cells, 21, n_gpus=slabs)
    assert np.array_equal(ref_cells.view(np.uint32), got_cells.view(np.uint32))


def test_three_step_info_and_pieces(lbm, oracle, datasets, monkeypatch):
    """lbm_run in pieces of every residue mod 3 with three slabs: each piece starts with its own accelerate
    pass and exchange and may end with a two-step or a one-step pass."""
    force_stream(monkeypatch, 3, 4)
    monkeypatch.setenv("LBM_HALO", "memcpy")
    p, ob = datasets("128x256")
    cells = oracle.init_cells(p)
    ref = cells.copy()
    ref_av = oracle.run(p, ref, ob, 70)
    with lbm.Engine(p, ob, cells, n_gpus=3) as eng:
        info = eng.info()
        assert info["steps_per_launch"] == 3 and info["band_rows"] == 4 and info["lane_cells"] == 4
        for n in (1, 2, 3, 4, 5, 7, 48):
            eng.run(n)
        assert eng.info()["steps_done"] == 70
        assert np.array_equal(eng.cells().view(np.uint32), ref.view(np.uint32))
        np.testing.assert_allclose(eng.av_vels(70), ref_av, rtol=AV_RTOL)


def test_three_step_random_lattices(lbm, oracle, monkeypatch):
    """Random populations / obstacles, both periodic wraps live, several strips per row, uneven slabs, lid row
    next to a slab edge, slabs just tall enough for a three-step pass (6 rows)."""
    monkeypatch.setenv("LBM_HALO", "memcpy")
    for nx, ny, slabs, band, pf, chunk, steps in ((256, 24, 1, 3, 0, 0, 20), (512, 25, 4, 3, 1, 2, 19), (64, 40, 5, 2, 0, 1, 21),
                                                  (1024, 13, 2, 5, 1, 0, 9), (4096, 67, 3, 8, 1, 4, 10), (2048, 9, 1, 4, 0, 3, 8),
                                                  (8, 3, 1, 2, 0, 0, 7), (12, 6, 1, 7, 1, 0, 12)):
        force_stream(monkeypatch, 3, band, pf, chunk)
        p, ob, cells = random_case(lbm, nx, ny, 300 + ny, walls=False)
        ref_cells, ref_av, got_cells, got_av, _ = run_both(lbm, oracle, p, ob, cells, steps, n_gpus=slabs)
        assert np.array_equal(ref_cells.view(np.uint32), got_cells.view(np.uint32)), (nx, ny, slabs)
        np.testing.assert_allclose(got_av, ref_av, rtol=AV_RTOL)


def test_three_step_rank_api_rccl_self(lbm, oracle, datasets, monkeypatch):
    """The one-process-per-GPU pipeline with three-step passes: 3-row halos through RCCL send/recv to itself."""
    force_stream(monkeypatch, 3, 5, 1, 0)
    monkeypatch.setenv("LBM_FORCE_HALO", "1")
    p, ob = datasets("128x256")
    cells = oracle.init_cells(p)
    ref = cells.copy()
    ref_av = oracle.run(p, ref, ob, 47)
    with lbm.Engine(p, ob, cells, rank=0, world_size=1, unique_id=lbm.rccl_unique_id(), device=0) as eng:
        eng.run(47)
        assert np.array_equal(eng.cells().view(np.uint32), ref.view(np.uint32))
        np.testing.assert_allclose(eng.av_vels(47), ref_av, rtol=AV_RTOL)


# ------------------------------------------------------------------------------------------------
# mask construction on the device
# ------------------------------------------------------------------------------------------------
def test_tiled_obstacles_equal_global_map(lbm, oracle, datasets, monkeypatch):
    """lbm_create_tiled: the mask expanded on the device from a tile equals the host-tiled global map --
    same lattice, same fluid-cell count, also across slabs (mask halo rows wrap periodically)."""
    monkeypatch.setenv("LBM_HALO", "memcpy")
    _, tile = datasets("128x128")
    for nx, ny, slabs in ((512, 384, 1), (384, 256, 3), (200, 150, 2)):          # the last one cuts tiles
        p = lbm.Params(nx, ny, 30, 10, 0.1, 0.005, 1.85)
        ob = lbm.tile_obstacles(tile, nx, ny)
        with lbm.Engine(p, ob, None, n_gpus=slabs) as a, lbm.Engine(p, tile, None, n_gpus=slabs, tiled=True) as b:
            assert a.info()["fluid_cells"] == b.info()["fluid_cells"] == int((ob == 0).sum())
            a.run(30)
            b.run(30)
            assert np.array_equal(a.cells().view(np.uint32), b.cells().view(np.uint32))
            assert np.array_equal(a.av_vels(30).view(np.uint32), b.av_vels(30).view(np.uint32))
        ref = oracle.init_cells(p)
        oracle.run(p, ref, ob, 30)
        with lbm.Engine(p, tile, None, n_gpus=slabs, tiled=True) as b:
            b.run(30)
            assert np.array_equal(b.cells().view(np.uint32), ref.view(np.uint32))


def test_rank_rows_form_equals_global_form(lbm, oracle, monkeypatch):
    """lbm_create_rank_rows (world of one, RCCL self-exchange): only the rank's rows + periodic neighbour rows
    are handed over, cells likewise."""
    monkeypatch.setenv("LBM_FORCE_HALO", "1")
    p, ob, cells = random_case(lbm, 96, 31, 5, walls=False)
    h = lbm.MASK_HALO_ROWS
    rows = np.concatenate([ob[-h:], ob, ob[:h]])            # periodic neighbour rows below and above
    ref = cells.copy()
    ref_av = oracle.run(p, ref, ob, 21)
    with lbm.Engine(p, rows, cells, rank=0, world_size=1, unique_id=lbm.rccl_unique_id(), device=0,
                    local_rows=True) as eng:
        assert eng.info()["fluid_cells"] == int((ob == 0).sum())
        eng.run(21)
        assert np.array_equal(eng.cells().view(np.uint32), ref.view(np.uint32))
        np.testing.assert_allclose(eng.av_vels(21), ref_av, rtol=AV_RTOL)
    tile_p = lbm.Params(96, 62, 10, 10, 0.1, 0.005, 1.85)
    with lbm.Engine(tile_p, ob, None, rank=0, world_size=1, unique_id=lbm.rccl_unique_id(), device=0,
                    tiled=True) as eng:
        assert eng.info()["fluid_cells"] == 2 * int((ob == 0).sum())


@pytest.mark.parametrize("kernel", ["tile", "one-step", "two-cell packed", "four-step packed", "three-step scalar"])
def test_guard_paths_stay_bit_exact(lbm, oracle, monkeypatch, kernel):
    """The fast constant divides and the shared reciprocal are guarded (density within [2^-60, 2^60), |u|^2 < 5e28);
    cells outside take the IEEE divides -- in the packed kernels per lane, after the packed computation.  A lattice
    with patches of populations scaled by 1e-22 and 1e+20 (densities 1e-23 and 1e+19) and a few cells with momenta
    far above their density drives every kernel family through those paths; results must still equal the oracle's
    bits (all values stay finite)."""
    env = {"tile": {}, "one-step": {"LBM_FUSE2": "0"},
           "two-cell packed": {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "2"},
           "four-step packed": {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "4"},
           "three-step scalar": {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "4", "LBM_PACKED": "0", "LBM_PASS_STEPS": "3"}}[kernel]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    p, ob, cells = random_case(lbm, 256, 40, 4242, walls=False)
    rng = np.random.default_rng(7)
    cells[5:9, 10:40] *= np.float32(1e-22)
    cells[20:23, 100:180] *= np.float32(1e20)
    for _ in range(12):                                   # isolated cells with |u| >> 1
        y, x = int(rng.integers(0, 40)), int(rng.integers(0, 256))
        cells[y, x, 1] *= np.float32(1e6)
    steps = 9
    ref_cells, ref_av, got_cells, got_av, _ = run_both(lbm, oracle, p, ob, cells, steps)
    assert np.isfinite(ref_cells).all()
    assert np.array_equal(ref_cells.view(np.uint32), got_cells.view(np.uint32))
    assert np.isfinite(got_av).all()


def test_fast_math_three_step_kernel_within_check_tolerance(lbm, oracle, datasets, monkeypatch):
    """The scalar stream kernel with FAST arithmetic and three steps per pass (LBM_PACKED=0; by default both math
    modes run the packed exact stream kernels, which are faster); forced here at test size (4 cells per lane) and
    held to the check.py rule against the oracle, like the other fast-mode kernels."""
    monkeypatch.setenv("LBM_FUSE2", "1")
    monkeypatch.setenv("LBM_LANE_CELLS", "4")
    monkeypatch.setenv("LBM_PACKED", "0")
    p, ob = datasets("128x128")
    cells = oracle.init_cells(p)
    steps = 2000
    ref_cells, ref_av, got_cells, got_av, fields = run_both(lbm, oracle, p, ob, cells, steps, math="fast")
    with lbm.Engine(p, ob, None, math="fast") as eng:
        assert eng.info()["steps_per_launch"] == 3
    ref_f = oracle.final_state(p, ref_cells, ob)
    assert lbm.check_passes(ref_av, got_av, 1.0)
    assert lbm.check_passes(ref_f["pressure"], fields["pressure"], 1.0)
    np.testing.assert_allclose(got_av, ref_av, rtol=2e-4)


@pytest.mark.parametrize("nx,ny,slabs", [(4100, 400, 1), (8192, 200, 1), (16384, 100, 2), (2052, 1000, 3), (6148, 300, 1),
                                         (1600, 1000, 1), (8192, 8, 1), (2048, 2048, 4)])
def test_default_policy_on_odd_shapes(lbm, oracle, monkeypatch, nx, ny, slabs):
    """Nothing forced: whatever kernel and geometry the policy picks for wide-and-short, ragged-pitch (nx % 64 != 0)
    and multi-slab shapes of >= 1.5 Mi cells must reproduce the oracle (random lattice, both wraps live)."""
    monkeypatch.setenv("LBM_HALO", "memcpy")
    p, ob, cells = random_case(lbm, nx, ny, nx + ny, blocked_frac=0.02, walls=False)
    steps = 11
    ref_cells, ref_av, got_cells, got_av, _ = run_both(lbm, oracle, p, ob, cells, steps, n_gpus=slabs)
    assert np.array_equal(ref_cells.view(np.uint32), got_cells.view(np.uint32))
    np.testing.assert_allclose(got_av, ref_av, rtol=AV_RTOL)


@pytest.mark.parametrize("slabs,env", [(2, {}), (3, {}), (8, {}), (3, {"LBM_LANE_CELLS": "4", "LBM_FUSE2": "1"}),
                                       (1, {"LBM_FORCE_HALO": "1"})])
def test_graph_chunks_of_the_device_copy_pipeline(lbm, oracle, monkeypatch, slabs, env):
    """LBM_GRAPH=1 with halos (opt-in): chunks of the I / X / B pipeline of every slab -- two streams per slab, the
    device copies of the halo exchange inside -- built node by node (GraphBuilder) and replayed by hipGraphLaunch,
    against the oracle.  3 and 8 slabs are the cases whose multi-stream CAPTURE overflowed the stack inside
    hipStreamEndCapture (profiles/r03_graph_capture_defect.md); a crash here fails the run, nothing is excused."""
    monkeypatch.setenv("LBM_HALO", "memcpy")
    monkeypatch.setenv("LBM_GRAPH", "1")
    monkeypatch.setenv("LBM_GRAPH_PASSES", "6")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    p, ob, cells = random_case(lbm, 256, 96, 31 + slabs, blocked_frac=0.03, walls=False)
    calls = [90, 1, 47]                      # several chunks, a tail of single passes, a one-step call, again
    ref = cells.copy()
    ref_av = oracle.run(p, ref, ob, sum(calls))
    with lbm.Engine(p, ob, cells, n_gpus=slabs) as eng:
        per_chunk = eng.info()["graph_steps"]
        assert per_chunk >= 2 * eng.info()["steps_per_launch"]
        for n in calls:
            eng.run(n)
        assert eng.info()["graph_steps"] == per_chunk       # the graph path was not abandoned on the way
        got = eng.cells()
        got_av = eng.av_vels(sum(calls))
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
    np.testing.assert_allclose(got_av, ref_av, rtol=AV_RTOL)


@pytest.mark.parametrize("torch_first", [False, True])
def test_graph_chunks_of_the_rank_pipeline_rccl(lbm, torch_first):
    """The same for one rank's pipeline with RCCL send/recv (a ring of one): the group is captured on its single comm
    stream and enters the chunk as a child graph.  Under ROCm's RCCL (a process without torch) and under torch's
    bundled RCCL + HIP runtime (the situation of bench.py; round 2's multi-stream capture segfaulted there).  Runs in
    a fresh process because the library choice is made at load time; a crash or a hang FAILS."""
    import re
    import subprocess
    import sys
    from conftest import ROOT
    import os
    cmd = [sys.executable, os.path.join(ROOT, "tools", "graph_rank.py"), "1024x512", "300"] + (["--torch-first"] if torch_first else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert out.returncode == 0, (out.returncode, out.stderr[-1500:])
    replayed = [int(m) for m in re.findall(r"'graph_steps': (\d+)", out.stdout)]
    assert replayed and replayed[0] == 0 and replayed[-1] >= 40, replayed      # launches first, then chunks of 20-32 passes
    assert ("/torch/lib/" in out.stdout) == torch_first, out.stdout[-800:]
    assert "fields equal: True" in out.stdout, out.stdout[-800:]
