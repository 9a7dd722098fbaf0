"""The multi-slab pipeline on ONE device (device-copy or RCCL halos) with and without hipGraph replay: same fields,
host issue and wall time per step.  python tools/graph_slabs.py NXxNY slabs steps [memcpy|rccl]
LBM_GRAPH_DUMP=<file.dot> LBM_GRAPH_DUMP_ONLY=1 only inspect the captured graph (no instantiate)."""
import os, sys, time
sys.path.insert(0, "tests")
import numpy as np
import conftest
lbm = conftest.load_package()
nx, ny = (int(v) for v in sys.argv[1].split("x"))
slabs, steps = int(sys.argv[2]), int(sys.argv[3])
os.environ["LBM_HALO"] = sys.argv[4] if len(sys.argv) > 4 else "memcpy"
if slabs == 1:
    os.environ["LBM_FORCE_HALO"] = "1"
tile = lbm.read_obstacles(os.path.join(conftest.GOLDEN, "inputs", "obstacles_128x128.dat"), 128, 128)
p = lbm.Params(nx, ny, 2 * steps, 10, 0.1, 0.01, 1.85)
out = {}
for graph in ("0", "1"):
    os.environ["LBM_GRAPH"] = graph
    with lbm.Engine(p, tile, None, n_gpus=slabs, tiled=True) as eng:
        print("graph", graph, eng.info(), flush=True)
        eng.run(steps); eng.sync()
        t0 = time.perf_counter(); eng.run(steps); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
        out[graph] = eng.final_state()["pressure"].copy()
        print("graph %s after: %s" % (graph, eng.info()), flush=True)
        print("graph %s: host issue %.2f us/step, wall %.2f us/step" % (graph, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6), flush=True)
print("fields equal:", np.array_equal(out["0"].view(np.uint32), out["1"].view(np.uint32)))
