"""us per timestep with and without hipGraph replay (LBM_GRAPH) on single-slab grids."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
lbm = conftest.load_package()
for name, steps in (("128x128", 40000), ("128x256", 40000), ("256x256", 40000), ("1024x1024", 8000)):
    p, ob = conftest.dataset(name)
    p.max_iters = 2 * steps
    for g, ts in (("0", "0"), ("1", "0"), ("0", "4"), ("1", "4")):
        os.environ["LBM_GRAPH"] = g
        os.environ["LBM_TILE_STEPS"] = ts
        with lbm.Engine(p, ob, None) as eng:
            eng.run(steps // 10); eng.sync()
            t0 = time.perf_counter(); eng.run(steps); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
        print(f"{name} LBM_GRAPH={g} LBM_TILE_STEPS={ts}: host issue {1e6*(t1-t0)/steps:6.2f} us/step, wall {1e6*(t2-t0)/steps:6.2f} us/step", flush=True)
