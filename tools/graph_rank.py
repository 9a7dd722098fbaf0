"""The rank pipeline (RCCL self-exchange) with and without hipGraph replay (LBM_GRAPH): same fields, time per step.
python tools/graph_rank.py NXxNY steps [--torch-first]   (without the flag: ROCm's RCCL, no torch in the process)"""
import os, sys, time
if "--torch-first" in sys.argv:
    import torch
sys.path.insert(0, "tests")
import numpy as np
import conftest
lbm = conftest.load_package()
nx, ny = (int(v) for v in sys.argv[1].split("x"))
steps = int(sys.argv[2])
os.environ["LBM_FORCE_HALO"] = "1"
tile = lbm.read_obstacles(os.path.join(conftest.GOLDEN, "inputs", "obstacles_1024x1024.dat"), 1024, 1024)
p = lbm.Params(nx, ny, 2 * steps, 10, 0.1, 0.01, 1.85)
out = {}
for graph in ("0", "1"):
    os.environ["LBM_GRAPH"] = graph
    with lbm.Engine(p, tile, None, rank=0, world_size=1, unique_id=lbm.rccl_unique_id(), device=0, tiled=True) as eng:
        print("graph", graph, eng.info(), eng.rccl_info()["library"], flush=True)
        eng.run(steps); eng.sync()
        t0 = time.perf_counter(); eng.run(steps); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
        out[graph] = eng.final_state()["pressure"].copy()
        print("graph %s: host issue %.2f us/step, wall %.2f us/step" % (graph, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6), flush=True)
print("fields equal:", np.array_equal(out["0"].view(np.uint32), out["1"].view(np.uint32)))
