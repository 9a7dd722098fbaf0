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


def make_case(lbm, oracle, case, form):
    """(params, global obstacle map, global initial lattice) of a test case; the tile form starts from the uniform
    equilibrium on a grid twice the data set's height (the tile repeated)."""
    import conftest
    from test_gpu_parity import random_case
    if isinstance(case, str):
        p, ob = conftest.dataset(case)
        if form == "tile":
            p.ny *= 2
            ob = np.concatenate([ob, ob])
        return p, ob, oracle.init_cells(p)
    return random_case(lbm, *case, walls=False)


def rank_main(rank, world, port, case, steps_list, out_dir, env, form="global"):
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
        p, ob, cells = make_case(lbm, oracle, case, form)

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
        first, count = lbm.partition_rows(p.ny, world, rank)
        if form == "rows":
            # the reference's scatter (MPI_Waitall/d2q9-bgk.c:816-842): this rank's rows only, plus the periodic
            # neighbour rows a multi-step pass relaxes redundantly
            h = lbm.MASK_HALO_ROWS
            idx = np.arange(first - h, first + count + h) % p.ny
            create = dict(obstacles=ob[idx], cells=cells[first:first + count], local_rows=True)
        elif form == "tile":
            create = dict(obstacles=ob[:p.ny // 2], cells=None, tiled=True)
        else:
            create = dict(obstacles=ob, cells=cells)
        with lbm.Engine(p, create["obstacles"], create["cells"], rank=rank, world_size=world, device=0,
                        host_comm=(exchange, allreduce), tiled=create.get("tiled", False),
                        local_rows=create.get("local_rows", False)) as eng:
            info = eng.info()
            assert (info["world_rank"], info["world_size"]) == (rank, world)
            assert (info["row_first"], info["row_count"]) == (first, count)
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


@pytest.mark.parametrize("world,case,steps_list,env,depth,form", [
    (2, "128x256", [76], None, 3, "global"),           # default at this size: three-step passes on one pair per lane
    (3, "128x256", [40, 1, 22], None, 3, "global"),    # run in pieces
    (2, "128x256", [75], {"LBM_PASS_STEPS": "2"}, 2, "global"),  # two-step passes, odd tail
    (2, "128x256", [77], FOUR_STEP, 4, "global"),      # four-step packed passes + a one-step tail; 2 ranks: north == south
    (3, "128x256", [31, 2, 44], FOUR_STEP, 4, "global"),
    (3, (512, 50, 5), [26], FOUR_STEP, 4, "global"),   # random lattice, uneven slabs (17, 17, 16 rows), both wraps live
    (2, (260, 33, 6), [9], None, 3, "global"),         # 4 | nx but nx % 64 != 0
    # lbm_create_rank_hosted_rows: every rank hands over its own rows only (+ the periodic neighbour rows of the mask)
    (3, (512, 50, 5), [26], FOUR_STEP, 4, "rows"),
    (2, (260, 33, 6), [9], None, 3, "rows"),
    (3, "128x256", [40, 1, 22], None, 3, "rows"),
    # lbm_create_rank_hosted_tiled: the 128x128 map repeated over a 128x256 grid, expanded on each rank's device
    (3, "128x128", [37], None, 3, "tile"),
])
def test_hosted_ranks_equal_single_domain(tmp_path, oracle, lbm, world, case, steps_list, env, depth, form):
    torch.set_num_threads(1)
    mp.spawn(rank_main, args=(world, free_port(), case, steps_list, str(tmp_path), env, form), nprocs=world, join=True)
    p, ob, ref = make_case(lbm, oracle, case, form)
    ref_av = oracle.run(p, ref, ob, sum(steps_list))
    assert int(np.load(tmp_path / "depth.npy")[0]) == depth
    got = np.load(tmp_path / "lattice.npy")
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    np.testing.assert_allclose(np.load(tmp_path / "av.npy"), ref_av, rtol=2e-4)
