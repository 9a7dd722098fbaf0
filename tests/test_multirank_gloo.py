"""N > 1 path on CPU: world_size 2 and 3 over torch.distributed/gloo.

The GPU engine shards the grid by rows, one slab per rank, keeps halo rows around every slab
and, once per pass, ships WHOLE boundary rows to its ring neighbours (lbm_hip.hip: exchange_halos --
the GPU analogue of /root/reference/MPI_Waitall/d2q9-bgk.c:225-253); a pass advances K timesteps
(K = 2, 3 or 4: the stream kernels, halo depth K), else one.  RCCL needs one GPU per rank, which
the builder container and the 1-GPU box lack, so this test replays the SAME protocol on CPU ranks,
with the bookkeeping taken from the product so the replay cannot drift from the engine:

  * rows from the product's own lbm_partition_rows, K from lbm_plan_halo_depth, and the four messages
    of an exchange (peer, rows, posting order) from lbm_halo_plan -- the C-ABI, host-only functions
    exchange_halos itself is built on; the order must pair correctly when rank+1 == rank-1 (world_size 2);
  * rows that touch no halo first, the rest after the halos arrived (the Waitall pattern);
  * K-step pass: step t+j-1 is relaxed on rows -(K-j)..rows-1+(K-j) (the halo-adjacent rows redundantly,
    exactly as the neighbour relaxes them), the last step on the owned rows; the lid row (global ny-2)
    is accelerated wherever a rank holds a copy of it, owned or halo;
  * a step count that is not a multiple of K ends with a two-step and / or a one-step pass;
  * per-step partial sums of |u| over OWNED rows all-reduced at the end
    (MPI/d2q9-bgk.c:298-309) and divided by the global fluid-cell count; rows gathered on rank 0
    (MPI/d2q9-bgk.c:265-295).

The slab arithmetic is the CPU oracle's fused row-range form (test infrastructure); the result
must equal the single-domain oracle bit for bit.  The same decomposition is checked on real
hardware by tests/test_gpu_parity.py (several slabs on one device + RCCL self-exchange).
"""
import os
import socket
import sys
import threading
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def random_cells(p, seed):
    rng = np.random.default_rng(seed)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float32) * np.float32(p.density)
    return (w * (1.0 + 0.05 * rng.standard_normal((p.ny, p.nx, 9)))).astype(np.float32)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_main(rank, world, port, name, steps, out_dir, lag=0, seed=None, env=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.update(env or {})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import conftest
        import oracle_binding
        lbm = conftest.load_package()
        oracle = oracle_binding.load()
        p, ob = conftest.dataset(name)
        # the decomposition, the halo depth (= timesteps per pass) and the four messages of an exchange all come
        # from the product's host-only C-ABI queries -- the very functions its own exchange_halos is built on
        first, rows = lbm.partition_rows(p.ny, world, rank)
        K = lbm.plan_halo_depth(p, world)
        H = K + 1                                              # halo rows kept around the slab
        M = max(K - 1, 1)                                      # halo rows a K-step pass may relax redundantly
        # local indices of the copies of the lid row this rank holds (owned or among the relaxed halo rows)
        lid_rows = [p.ny - 2 + shift - first for shift in (-p.ny, 0, p.ny)
                    if -M <= p.ny - 2 + shift - first < rows + M]

        start = oracle.init_cells(p) if seed is None else random_cells(p, seed)
        full = np.ascontiguousarray(start.transpose(2, 0, 1))                     # (9, ny, nx)
        bufs = [np.zeros((9, rows + 2 * H, p.nx), dtype=np.float32) for _ in range(K + 1)]   # array row = local row + H
        bufs[0][:, H:H + rows] = full[:, first:first + rows]
        # mask of local rows -(H-1) .. rows+H-2 (periodic neighbours), as the engine builds it
        mask = np.ascontiguousarray(ob[[(first + r) % p.ny for r in range(-(H - 1), rows + H - 1)]])
        tot_u = np.zeros(steps, dtype=np.float64)
        ps = (rows + 2 * H) * p.nx

        def accelerate(arr, local_row):
            oracle.lib.lbm_oracle_accelerate_row_soa(p.nx, p.density, p.accel, arr.ctypes.data, ps,
                                                     mask[local_row + H - 1].ctypes.data, local_row + H)

        def relax(src, dst, lo, hi):
            """advance local rows lo..hi (inclusive) of src into dst; returns sum |u| of those rows"""
            if hi < lo:
                return 0.0
            return oracle.fused_rows(p.nx, rows + 2 * (H - 1), p.density, p.accel, p.omega, src, dst, mask, 0,
                                     lo + H, hi + H)

        def exchange(arr, depth):
            """post the engine's four messages in the engine's order; returns the requests and where receives land"""
            ops, landing = [], []
            for op in lbm.halo_plan(rows, world, rank, depth):
                lo = op["row_first"] + H
                if op["is_send"]:
                    t = torch.from_numpy(np.ascontiguousarray(arr[:, lo:lo + op["row_count"]]))
                    ops.append(dist.P2POp(dist.isend, t, op["peer"]))
                else:
                    t = torch.empty((9, op["row_count"], p.nx), dtype=torch.float32)
                    ops.append(dist.P2POp(dist.irecv, t, op["peer"]))
                    landing.append((lo, op["row_count"], t))
            return dist.batch_isend_irecv(ops), landing

        def land(arr, landing):
            for lo, n, t in landing:
                arr[:, lo:lo + n] = t.numpy()

        t, m, stale = 0, 0, None
        looks = np.full(steps, 3, dtype=np.uint8)   # lag == 2: what each look found (bit 0 south, bit 1 north halo row)
        S = bufs[0]
        while t < steps:
            left = steps - t
            # the engine's pass: K timesteps while that many remain, then two, then one; stale passes advance one
            k = 1 if lag else (K if left >= K else (2 if left >= 2 else 1))
            depth = 1 if lag else K          # the engine ships K halo rows every pass once the stream kernel is on
            # accelerate_flow of step t on the owned copy; the halo copies arrive already accelerated
            for r in lid_rows:
                if 0 <= r < rows:
                    accelerate(S, r)
            if lag == 2 and ((m * 2654435761 >> 7) ^ rank) & 1:
                time.sleep(0.004)            # this rank's rows leave late in about half of the passes: looks will miss them
            reqs, landing = exchange(S, depth)
            if lag and m > 0:
                # stale-halo mode (run_steps_stale): this pass reads what the neighbours sent one pass
                # ago; the exchange posted above only has to land before the NEXT pass
                land(S, stale)
            others = [b for b in bufs if b is not S]
            T = others[0]
            # step t on the rows whose inputs are all owned (overlaps the exchange) ...
            s_in = relax(S, T, k, rows - 1 - k)
            if lag == 2:
                # freshest-available mode: MPI_Testall's question, asked ONCE per receive, never waited for
                # (/root/reference/MPI_Testall_OptimizedVersion/d2q9-bgk.c:279-290); a row that has arrived replaces
                # the row of the pass before, whole.  (gloo's Work.is_completed() stays False until somebody waits, so
                # a thread per receive waits and raises a flag -- the look reads the flags.)
                arrived = [False, False]

                def wait_recv(i, req):
                    req.wait()
                    arrived[i] = True
                waiters = [threading.Thread(target=wait_recv, args=(i, req)) for i, req in enumerate(reqs[2:])]
                for w in waiters:
                    w.start()
                if m > 0:
                    looks[t] = 0
                    for side, (lo, n, tt) in enumerate(landing):
                        if arrived[side]:
                            S[:, lo:lo + n] = tt.numpy()
                            looks[t] |= 1 << side
                else:
                    for w in waiters:
                        w.join()
                    land(S, landing)                            # every run starts from fresh halos
                s_b = relax(S, T, 0, 0) + relax(S, T, rows - 1, rows - 1)
                for w in waiters:
                    w.join()                                    # ... before the NEXT pass, as its guaranteed rows
                for r in reqs[:2]:
                    r.wait()
                stale = [(lo, n, tt.clone()) for lo, n, tt in landing]
                tot_u[t] = float(s_in) + float(s_b)
                S = T
                t += 1
                m += 1
                continue
            for r in reqs:
                r.wait()
            if lag:
                if m == 0:
                    land(S, landing)                            # every run starts from fresh halos
                stale = [(lo, n, tt.clone()) for lo, n, tt in landing]
            else:
                land(S, landing)
            # ... then on the halo-dependent rows: owned rows [0, k) and [rows-k, rows), and (k > 1) the k-1 halo
            # rows on each side, relaxed redundantly exactly as their owners relax them
            s_b = relax(S, T, 0, k - 1) + relax(S, T, rows - k, rows - 1)
            relax(S, T, -(k - 1), -1)
            relax(S, T, rows, rows + k - 2)
            tot_u[t] = float(s_in) + float(s_b)
            cur = T
            for j in range(2, k + 1):
                # accelerate_flow of step t+j-1 on every copy of the lid row this rank holds
                for r in lid_rows:
                    if -(k - j + 1) <= r <= rows - 1 + (k - j + 1):
                        accelerate(cur, r)
                nxt = others[j - 1]
                relax(cur, nxt, -(k - j), -1)
                tot_u[t + j - 1] = float(relax(cur, nxt, 0, rows - 1))
                relax(cur, nxt, rows, rows + (k - j) - 1)
                cur = nxt
            S = cur
            t += k
            m += 1

        if lag == 2:
            mine_looks = torch.from_numpy(looks)
            gathered = [torch.empty_like(mine_looks) for _ in range(world)]
            dist.all_gather(gathered, mine_looks)
            if rank == 0:
                np.save(os.path.join(out_dir, "looks.npy"), np.stack([g.numpy() for g in gathered], axis=1))
        tot = torch.from_numpy(tot_u)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        av = (tot.numpy().astype(np.float32) / np.float32((ob == 0).sum())).astype(np.float32)
        mine = torch.from_numpy(np.ascontiguousarray(S[:, H:H + rows]))
        if rank == 0:
            result = np.empty((9, p.ny, p.nx), dtype=np.float32)
            result[:, first:first + rows] = mine.numpy()
            for r in range(1, world):
                f, n = lbm.partition_rows(p.ny, world, r)
                buf = torch.empty((9, n, p.nx), dtype=torch.float32)
                dist.recv(buf, src=r)
                result[:, f:f + n] = buf.numpy()
            np.save(os.path.join(out_dir, "lattice.npy"), result)
            np.save(os.path.join(out_dir, "av.npy"), av)
            np.save(os.path.join(out_dir, "depth.npy"), np.array([K]))
        else:
            dist.send(mine, dst=0)
    finally:
        dist.destroy_process_group()


FOUR_STEP = {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "4"}                           # what slabs of >= 3 Mi cells get
THREE_STEP = dict(FOUR_STEP, LBM_PASS_STEPS="3")                                # ... and in the fast-math mode


@pytest.mark.parametrize("world,name,steps,env,depth", [(2, "128x128", 61, None, 3), (3, "128x256", 40, None, 3),
                                                        (2, "128x128", 61, {"LBM_PASS_STEPS": "2"}, 2),
                                                        (2, "128x128", 62, THREE_STEP, 3), (3, "128x256", 43, THREE_STEP, 3),
                                                        (4, "128x256", 31, THREE_STEP, 3), (2, "128x128", 63, FOUR_STEP, 4),
                                                        (3, "128x256", 45, FOUR_STEP, 4)])
def test_row_sharded_ring_equals_single_domain(tmp_path, oracle, datasets, lbm, world, name, steps, env, depth):
    torch.set_num_threads(1)
    mp.spawn(rank_main, args=(world, free_port(), name, steps, str(tmp_path), 0, None, env), nprocs=world, join=True)
    assert int(np.load(tmp_path / "depth.npy")[0]) == depth
    p, ob = datasets(name)
    ref = oracle.init_cells(p)
    ref_av = oracle.run(p, ref, ob, steps)
    got = np.load(tmp_path / "lattice.npy").transpose(1, 2, 0)
    assert np.array_equal(np.ascontiguousarray(got).view(np.uint32), ref.view(np.uint32))
    np.testing.assert_allclose(np.load(tmp_path / "av.npy"), ref_av, rtol=2e-4)


@pytest.mark.parametrize("world,name,steps", [(2, "128x128", 23), (3, "128x256", 12)])
def test_stale_halo_ring_equals_slab_model(tmp_path, oracle, datasets, lbm, world, name, steps):
    """Stale-halo mode (lbm_set_halo_mode(LBM_HALO_STALE), the reproducible analogue of
    /root/reference/MPI_Testall_OptimizedVersion/d2q9-bgk.c:256-301) over real message passing:
    ranks that consume each exchange one pass late must reproduce tests/slab_model.py (lag = 1),
    the model the GPU engine is held to bit for bit in tests/test_gpu_stale_halo.py.  A random start
    makes every boundary cell sensitive to the lag."""
    import slab_model
    torch.set_num_threads(1)
    mp.spawn(rank_main, args=(world, free_port(), name, steps, str(tmp_path), 1, 99), nprocs=world, join=True)
    p, ob = datasets(name)
    cells = random_cells(p, 99)
    want, want_tot = slab_model.run_slabs(oracle, lbm, p, ob, [steps], world, lag=1, cells=cells)
    got = np.ascontiguousarray(np.load(tmp_path / "lattice.npy").transpose(1, 2, 0))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    want_av = (want_tot / np.float64((ob == 0).sum())).astype(np.float32)
    np.testing.assert_allclose(np.load(tmp_path / "av.npy"), want_av, rtol=2e-4)
    # and the lag matters on this input: the synchronous protocol ends elsewhere
    sync, _ = slab_model.run_slabs(oracle, lbm, p, ob, [steps], world, lag=0, cells=cells)
    assert not np.array_equal(sync.view(np.uint32), want.view(np.uint32))
    ref = cells.copy()
    oracle.run(p, ref, ob, steps)
    assert np.array_equal(sync.view(np.uint32), ref.view(np.uint32))      # the model itself is pinned


@pytest.mark.parametrize("world,name,steps", [(2, "128x128", 40), (3, "128x256", 30)])
def test_freshest_halo_ring_equals_slab_model_under_its_own_schedule(tmp_path, oracle, datasets, lbm, world, name, steps):
    """Freshest-available mode (lbm_set_halo_mode(LBM_HALO_FRESHEST)) over real, really asynchronous messages: every
    rank posts its exchange, relaxes its interior rows, asks each receive ONCE whether it has completed
    (Work.is_completed(): MPI_Testall's question) and relaxes its boundary rows either way -- with this step's row
    where it has, with the row of the step before where it has not.  Ranks send late in about half of the passes, so
    both happen.  Whatever the looks found, the run must be the model's (tests/slab_model.py: run_slabs_freshest)
    under the schedule the ranks logged, bit for bit -- the same statement tests/test_gpu_freshest_halo.py makes of
    the engine and its lbm_read_halo_log."""
    import slab_model
    torch.set_num_threads(1)
    mp.spawn(rank_main, args=(world, free_port(), name, steps, str(tmp_path), 2, 77), nprocs=world, join=True)
    p, ob = datasets(name)
    cells = random_cells(p, 77)
    looks = np.load(tmp_path / "looks.npy")
    assert looks.shape == (steps, world) and (looks[0] == 3).all()
    assert (looks[1:] != 3).any(), "no look missed its rows: the delays did not bite"
    want, want_tot, _, _ = slab_model.run_slabs_freshest(oracle, lbm, p, ob, [steps], world, schedule=looks, cells=cells)
    got = np.ascontiguousarray(np.load(tmp_path / "lattice.npy").transpose(1, 2, 0))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    want_av = (want_tot / np.float64((ob == 0).sum())).astype(np.float32)
    np.testing.assert_allclose(np.load(tmp_path / "av.npy"), want_av, rtol=2e-4)
    print(f"looks that found this step's row: {np.mean([(looks[1:] & 1).mean(), ((looks[1:] >> 1) & 1).mean()]):.2f}")
