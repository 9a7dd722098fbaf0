"""Several RANKS of the real engine on the one GPU of the box: the one-process-per-GPU decomposition (row partition,
K-row halos, redundantly relaxed halo rows, lid row held as a halo copy, interior / boundary split, per-rank fluid count
and av_vels sums all-reduced) with worlds of 2 and 3 -- everything but the RCCL transport, which needs one device
per rank.  The ranks exchange their boundary rows through the host (lbm_create_rank_hosted: the reference's own
arrangement, MPI_Isend / MPI_Irecv / MPI_Waitall on host buffers, MPI_Waitall/d2q9-bgk.c:225-243), here over
torch.distributed / gloo.  The gathered lattice must equal the single-domain oracle bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_main(rank, world, port, case, steps_list, out_dir, env):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.update(env or {})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import conftest
        import oracle_binding
        from test_gpu_parity import random_case
        lbm = conftest.load_package()
        oracle = oracle_binding.load()
        if isinstance(case, str):
            p, ob = conftest.dataset(case)
            cells = oracle.init_cells(p)
        else:
            p, ob, cells = random_case(lbm, *case, walls=False)

        def exchange(plan, bufs):
            ops = []
            for op, buf in zip(plan, bufs):
                t = torch.from_numpy(buf)                      # shares the pinned staging buffer
                ops.append(dist.P2POp(dist.isend if op["is_send"] else dist.irecv, t, op["peer"]))
            for req in dist.batch_isend_irecv(ops):
                req.wait()

        def allreduce(values):
            dist.all_reduce(torch.from_numpy(values), op=dist.ReduceOp.SUM)

        total = sum(steps_list)
        with lbm.Engine(p, ob, cells, rank=rank, world_size=world, device=0, host_comm=(exchange, allreduce)) as eng:
            info = eng.info()
            assert (info["world_rank"], info["world_size"]) == (rank, world)
            assert info["fluid_cells"] == int((ob == 0).sum())          # all-reduced device-side count
            for n in steps_list:
                eng.run(n)
            mine = torch.from_numpy(np.ascontiguousarray(eng.cells()))
            av = eng.av_vels(total)
            depth = info["steps_per_launch"]
        if rank == 0:
            result = np.empty((p.ny, p.nx, 9), dtype=np.float32)
            result[info["row_first"]:info["row_first"] + info["row_count"]] = mine.numpy()
            for r in range(1, world):
                f, n = lbm.partition_rows(p.ny, world, r)
                buf = torch.empty((n, p.nx, 9), dtype=torch.float32)
                dist.recv(buf, src=r)
                result[f:f + n] = buf.numpy()
            np.save(os.path.join(out_dir, "lattice.npy"), result)
            np.save(os.path.join(out_dir, "av.npy"), av)
            np.save(os.path.join(out_dir, "depth.npy"), np.array([depth]))
        else:
            dist.send(mine, dst=0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


FOUR_STEP = {"LBM_FUSE2": "1", "LBM_LANE_CELLS": "4"}


@pytest.mark.parametrize("world,case,steps_list,env,depth", [
    (2, "128x256", [76], None, 3),                     # default at this size: three-step passes on one pair per lane
    (3, "128x256", [40, 1, 22], None, 3),              # run in pieces
    (2, "128x256", [75], {"LBM_PASS_STEPS": "2"}, 2),  # two-step passes, odd tail
    (2, "128x256", [77], FOUR_STEP, 4),                # four-step packed passes + a one-step tail; 2 ranks: north == south
    (3, "128x256", [31, 2, 44], FOUR_STEP, 4),
    (3, (512, 50, 5), [26], FOUR_STEP, 4),             # random lattice, uneven slabs (17, 17, 16 rows), both wraps live
    (2, (260, 33, 6), [9], None, 3),                   # 4 | nx but nx % 64 != 0
])
def test_hosted_ranks_equal_single_domain(tmp_path, oracle, lbm, world, case, steps_list, env, depth):
    torch.set_num_threads(1)
    mp.spawn(rank_main, args=(world, free_port(), case, steps_list, str(tmp_path), env), nprocs=world, join=True)
    import conftest
    from test_gpu_parity import random_case
    if isinstance(case, str):
        p, ob = conftest.dataset(case)
        ref = oracle.init_cells(p)
    else:
        p, ob, ref = random_case(lbm, *case, walls=False)
    ref_av = oracle.run(p, ref, ob, sum(steps_list))
    assert int(np.load(tmp_path / "depth.npy")[0]) == depth
    got = np.load(tmp_path / "lattice.npy")
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    np.testing.assert_allclose(np.load(tmp_path / "av.npy"), ref_av, rtol=2e-4)
