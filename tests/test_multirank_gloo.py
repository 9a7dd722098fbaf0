"""N > 1 path on CPU: world_size 2 and 3 over torch.distributed/gloo.

The GPU engine shards the grid by rows, one slab per rank, and exchanges packed halo rows with
its ring neighbours every step (lbm_hip.hip: exchange_halos -- the GPU analogue of
/root/reference/MPI_Waitall/d2q9-bgk.c:225-253).  RCCL needs one GPU per rank, which the builder
container and the 1-GPU box lack, so this test replays the SAME protocol on CPU ranks:

  * rows from the product's own lbm_partition_rows (C ABI, host-only call);
  * per step: send_north = planes 2,5,6 of the top owned row to rank+1, send_south = planes
    4,7,8 of the bottom owned row to rank-1, recv_south from rank-1, recv_north from rank+1, in
    the engine's posting order (which must pair correctly when rank+1 == rank-1, world_size 2);
  * interior rows first, boundary rows after the halos arrived (the Waitall pattern);
  * the lid row (global ny-2) accelerated by whichever rank owns it;
  * per-step partial sums of |u| all-reduced at the end (MPI/d2q9-bgk.c:298-309) and divided by
    the global fluid-cell count; rows gathered on rank 0 (MPI/d2q9-bgk.c:265-295).

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


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_main(rank, world, port, name, steps, out_dir):
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
        lid = p.ny - 2
        accel_row = lid - first + 1 if first <= lid < first + rows else 0   # 1-based slab row

        full = np.ascontiguousarray(oracle.init_cells(p).transpose(2, 0, 1))   # (9, ny, nx)
        src = np.zeros((9, rows + 2, p.nx), dtype=np.float32)
        src[:, 1:rows + 1] = full[:, first:first + rows]
        dst = np.zeros_like(src)
        mask = np.ascontiguousarray(ob[first:first + rows])
        tot_u = np.zeros(steps, dtype=np.float64)

        for t in range(steps):
            # accelerate first: the halo rows that leave this rank must already carry it
            # (the engine fuses it into the previous step's kernel; same values)
            if accel_row:
                oracle.lib.lbm_oracle_accelerate_row_soa(
                    p.nx, p.density, p.accel, src.ctypes.data, (rows + 2) * p.nx,
                    mask[accel_row - 1].ctypes.data, accel_row)
            send_north = torch.from_numpy(np.ascontiguousarray(src[[2, 5, 6], rows]))
            send_south = torch.from_numpy(np.ascontiguousarray(src[[4, 7, 8], 1]))
            recv_south = torch.empty_like(send_north)
            recv_north = torch.empty_like(send_south)
            # the engine's posting order: send N, send S, recv S, recv N
            ops = [dist.P2POp(dist.isend, send_north, north), dist.P2POp(dist.isend, send_south, south),
                   dist.P2POp(dist.irecv, recv_south, south), dist.P2POp(dist.irecv, recv_north, north)]
            reqs = dist.batch_isend_irecv(ops)
            # interior rows overlap with the exchange
            s_in = oracle.fused_rows(p.nx, rows, p.density, p.accel, p.omega, src, dst, mask, 0, 2, rows - 1)
            for r in reqs:
                r.wait()
            src[[2, 5, 6], 0] = recv_south.numpy()
            src[[4, 7, 8], rows + 1] = recv_north.numpy()
            s_b0 = oracle.fused_rows(p.nx, rows, p.density, p.accel, p.omega, src, dst, mask, 0, 1, 1)
            s_b1 = oracle.fused_rows(p.nx, rows, p.density, p.accel, p.omega, src, dst, mask, 0, rows, rows)
            tot_u[t] = float(s_in) + float(s_b0) + float(s_b1)
            src, dst = dst, src

        tot = torch.from_numpy(tot_u)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        av = (tot.numpy().astype(np.float32) / np.float32((ob == 0).sum())).astype(np.float32)
        mine = torch.from_numpy(np.ascontiguousarray(src[:, 1:rows + 1]))
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


@pytest.mark.parametrize("world,name,steps", [(2, "128x128", 60), (3, "128x256", 40)])
def test_row_sharded_ring_equals_single_domain(tmp_path, oracle, datasets, lbm, world, name, steps):
    torch.set_num_threads(1)
    mp.spawn(rank_main, args=(world, free_port(), name, steps, str(tmp_path)), nprocs=world, join=True)
    p, ob = datasets(name)
    ref = oracle.init_cells(p)
    ref_av = oracle.run(p, ref, ob, steps)
    got = np.load(tmp_path / "lattice.npy").transpose(1, 2, 0)
    assert np.array_equal(np.ascontiguousarray(got).view(np.uint32), ref.view(np.uint32))
    np.testing.assert_allclose(np.load(tmp_path / "av.npy"), ref_av, rtol=2e-4)
