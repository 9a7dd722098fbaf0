"""Wall time of ONE lbm_run(n) call (sync included) with the resident kernel forced on / off, by n: where does the
resident launch's fixed cost (load the lattice into registers, store it back, reduce) pay?  python tools/resident_crossover.py"""
import os, sys, time
sys.path.insert(0, "tests")
import numpy as np
import conftest
lbm = conftest.load_package()
for name in ("128x128", "256x256", "1024x1024"):
    p, ob = conftest.dataset(name)
    p.max_iters = 200000
    for n in (4, 8, 16, 32, 64, 128, 256):
        row = []
        for resident in ("0", "1"):
            os.environ["LBM_RESIDENT"] = resident
            os.environ["LBM_RESIDENT_MIN_STEPS"] = "1"
            with lbm.Engine(p, ob, None) as eng:
                for _ in range(3):
                    eng.run(n)
                eng.sync()
                t0 = time.perf_counter()
                reps = 20
                for _ in range(reps):
                    eng.run(n); eng.sync()
                row.append((time.perf_counter() - t0) / reps * 1e6)
        print(f"{name:10s} run({n:3d}) + sync: per-pass kernels {row[0]:8.1f} us   resident {row[1]:8.1f} us", flush=True)
