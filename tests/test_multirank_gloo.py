"""N > 1 path on CPU: world_size 2 and 3 over torch.distributed/gloo.

The GPU engine shards the grid by rows, one slab per rank, keeps two halo rows around every slab
and, once per pass, ships WHOLE boundary rows to its ring neighbours (lbm_hip.hip: exchange_halos --
the GPU analogue of /root/reference/MPI_Waitall/d2q9-bgk.c:225-253); a pass advances TWO timesteps
when the two-step kernel is active (halo depth 2), else one.  RCCL needs one GPU per rank, which
the builder container and the 1-GPU box lack, so this test replays the SAME protocol on CPU ranks:

  * rows from the product's own lbm_partition_rows (C ABI, host-only call);
  * per pass: my top `depth` rows -> rank+1's south halo, my bottom `depth` rows -> rank-1's north
    halo, posted in the engine's order (send N, send S, recv S, recv N), which must pair correctly
    when rank+1 == rank-1 (world_size 2);
  * rows that touch no halo first, the rest after the halos arrived (the Waitall pattern);
  * two-step pass: step t is relaxed on rows -1..rows (the two halo-adjacent rows redundantly,
    exactly as the neighbour relaxes them), step t+1 on the owned rows; the lid row (global ny-2)
    is accelerated wherever a rank holds a copy of it, owned or halo;
  * an odd step count ends with a one-step pass (which reads only the inner halo row);
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


def rank_main(rank, world, port, name, steps, out_dir, lag=0, seed=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import conftest
        import oracle_binding
        lbm = conftest.load_package()
        oracle = oracle_binding.load()
        p, ob = conftest.dataset(name)
        first, rows = lbm.partition_rows(p.ny, world, rank)
        north, south = (rank + 1) % world, (rank - 1 + world) % world
        H = 2                                                  # halo rows kept around the slab
        # local index (-1 .. rows) of the lid row if this rank holds a copy of it, else None
        lid_local = None
        for shift in (-p.ny, 0, p.ny):
            loc = p.ny - 2 + shift - first
            if -1 <= loc <= rows:
                lid_local = loc

        start = oracle.init_cells(p) if seed is None else random_cells(p, seed)
        full = np.ascontiguousarray(start.transpose(2, 0, 1))                     # (9, ny, nx)
        S = np.zeros((9, rows + 2 * H, p.nx), dtype=np.float32)                   # array row = local row + H
        S[:, H:H + rows] = full[:, first:first + rows]
        T = np.zeros_like(S)
        U = np.zeros_like(S)
        # mask of local rows -1 .. rows (periodic neighbours), as the engine uploads it
        mask = np.ascontiguousarray(ob[[(first + r) % p.ny for r in range(-1, rows + 1)]])
        tot_u = np.zeros(steps, dtype=np.float64)
        ps = (rows + 2 * H) * p.nx

        def accelerate(arr, local_row):
            oracle.lib.lbm_oracle_accelerate_row_soa(p.nx, p.density, p.accel, arr.ctypes.data, ps,
                                                     mask[local_row + 1].ctypes.data, local_row + H)

        def relax(src, dst, lo, hi):
            """advance local rows lo..hi (inclusive) of src into dst; returns sum |u| of those rows"""
            if hi < lo:
                return 0.0
            return oracle.fused_rows(p.nx, rows + 2, p.density, p.accel, p.omega, src, dst, mask, 0,
                                     lo + H, hi + H)

        def exchange(arr, depth):
            send_n = torch.from_numpy(np.ascontiguousarray(arr[:, H + rows - depth:H + rows]))
            send_s = torch.from_numpy(np.ascontiguousarray(arr[:, H:H + depth]))
            recv_s, recv_n = torch.empty_like(send_n), torch.empty_like(send_s)
            # the engine's posting order: send N, send S, recv S, recv N
            ops = [dist.P2POp(dist.isend, send_n, north), dist.P2POp(dist.isend, send_s, south),
                   dist.P2POp(dist.irecv, recv_s, south), dist.P2POp(dist.irecv, recv_n, north)]
            return dist.batch_isend_irecv(ops), recv_s, recv_n

        def land(arr, depth, recv_s, recv_n):
            arr[:, H - depth:H] = recv_s.numpy()
            arr[:, H + rows:H + rows + depth] = recv_n.numpy()

        t, m, stale = 0, 0, None
        while t < steps:
            two = t + 1 < steps and not lag      # stale passes always advance one timestep
            depth = 2          # the engine ships both halo rows every pass once the two-step kernel is on
            # accelerate_flow of step t on the owned copy; the halo copies arrive already accelerated
            if lid_local is not None and 0 <= lid_local < rows:
                accelerate(S, lid_local)
            reqs, recv_s, recv_n = exchange(S, depth)
            if lag and m > 0:
                # stale-halo mode (run_steps_stale): this pass reads what the neighbours sent one pass
                # ago; the exchange posted above only has to land before the NEXT pass
                land(S, depth, *stale)
            # step t on the rows whose inputs are all owned (overlaps the exchange) ...
            s_in = relax(S, T, 1, rows - 2)
            for r in reqs:
                r.wait()
            if lag:
                if m == 0:
                    land(S, depth, recv_s, recv_n)              # every run starts from fresh halos
                stale = (recv_s.clone(), recv_n.clone())
            else:
                land(S, depth, recv_s, recv_n)
            if two:
                # ... then on rows -1, 0 and rows-1, rows (halo-dependent; -1 and rows redundantly)
                relax(S, T, -1, -1)
                s_b = relax(S, T, 0, 0) + relax(S, T, rows - 1, rows - 1)
                relax(S, T, rows, rows)
                tot_u[t] = float(s_in) + float(s_b)
                # accelerate_flow of step t+1 on every copy of the lid row this rank holds
                if lid_local is not None:
                    accelerate(T, lid_local)
                tot_u[t + 1] = float(relax(T, U, 0, rows - 1))
                S, U = U, S
                t += 2
            else:
                tot_u[t] = float(s_in) + float(relax(S, T, 0, 0)) + float(relax(S, T, rows - 1, rows - 1))
                S, T = T, S
                t += 1
            m += 1

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
        else:
            dist.send(mine, dst=0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name,steps", [(2, "128x128", 61), (3, "128x256", 40)])
def test_row_sharded_ring_equals_single_domain(tmp_path, oracle, datasets, lbm, world, name, steps):
    torch.set_num_threads(1)
    mp.spawn(rank_main, args=(world, free_port(), name, steps, str(tmp_path)), nprocs=world, join=True)
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
