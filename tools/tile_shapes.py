"""us per timestep of the LDS-tile kernel by steps per launch (LBM_TILE_STEPS; 0 = the one-step kernels)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, numpy as np
lbm = conftest.load_package()
for name, steps in (("128x128", 20000), ("128x256", 20000), ("256x256", 20000), ("1024x1024", 4000)):
    p, ob = conftest.dataset(name); p.max_iters = 2 * steps
    ref = None
    for ts in ("0", "2", "4", "8"):
        os.environ["LBM_TILE_STEPS"] = ts
        os.environ["LBM_GRAPH"] = "1"
        with lbm.Engine(p, ob, None) as eng:
            eng.run(steps // 10); eng.sync()
            t0 = time.perf_counter(); eng.run(steps); eng.sync(); t2 = time.perf_counter()
            pr = eng.final_state()["pressure"]
        if ref is None: ref = pr
        print(f"{name} LBM_TILE_STEPS={ts}: {1e6*(t2-t0)/steps:6.2f} us/step  same={np.array_equal(pr.view(np.uint32), ref.view(np.uint32))}", flush=True)
