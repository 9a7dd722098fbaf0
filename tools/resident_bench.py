"""us per timestep of the resident kernel against the launch-per-pass kernels on the cache-resident sizes.
python tools/resident_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, "tests")
import numpy as np
import conftest
lbm = conftest.load_package()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
for (nx, ny) in ((128, 128), (128, 256), (256, 256), (512, 512), (768, 768), (1024, 1024), (1024, 512), (512, 1024)):
    row = []
    for resident in ("0", "1"):
        os.environ["LBM_RESIDENT"] = resident
        p = lbm.Params(nx, ny, 3 * steps, 10, 0.1, 0.01, 1.85)
        name = f"{nx}x{ny}"
        try:
            _, ob = conftest.dataset(name)
        except Exception:
            ob = np.zeros((ny, nx), dtype=np.int32); ob[0, :] = 1; ob[-1, :] = 1; ob[:, 0] = 1; ob[:, -1] = 1
        with lbm.Engine(p, ob, None) as eng:
            eng.run(steps); eng.sync()
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter(); ms = eng.run_timed(steps); eng.sync(); dt = time.perf_counter() - t0
                best = min(best, dt / steps * 1e6)
            row.append((best, ms * 1e3, eng.info()["resident_steps"] > 0))
    print(f"{nx:5d} x {ny:5d}: per-pass kernels {row[0][0]:7.2f} us/step (device {row[0][1]:6.2f})   resident {row[1][0]:7.2f} us/step (device {row[1][1]:6.2f}) {'' if row[1][2] else '(not eligible)'}   x{row[0][0] / row[1][0]:.2f}", flush=True)
