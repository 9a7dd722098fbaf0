"""us per timestep of the resident kernel under sets of LBM_RESIDENT_* knobs, each checked bit for bit against the
per-pass kernels after the run.
python tools/resident_variants.py 128x128,256x256 [steps] -- "" "LBM_RESIDENT_ROWS=4" "LBM_RESIDENT_ROWS=4,LBM_RESIDENT_JOINT=0" """
import os, sys, time
sys.path.insert(0, "tests")
import numpy as np
import conftest
lbm = conftest.load_package()
cut = sys.argv.index("--")
grids = [tuple(int(v) for v in g.split("x")) for g in sys.argv[1].split(",")]
steps = int(sys.argv[2]) if cut > 2 else 4000
variants = sys.argv[cut + 1:] or [""]


def obstacles(nx, ny):
    try:
        return conftest.dataset(f"{nx}x{ny}")[1]
    except Exception:
        ob = np.zeros((ny, nx), dtype=np.int32); ob[0, :] = 1; ob[-1, :] = 1; ob[:, 0] = 1; ob[:, -1] = 1
        return ob


for nx, ny in grids:
    ob = obstacles(nx, ny)
    p = lbm.Params(nx, ny, 3 * steps, 10, 0.1, 0.01, 1.85)
    os.environ["LBM_RESIDENT"] = "0"
    with lbm.Engine(p, ob, None) as eng:
        eng.run(3 * steps); want = eng.cells().view(np.uint32)
    del os.environ["LBM_RESIDENT"]
    for v in variants:
        env = dict(kv.split("=") for kv in v.split(",") if kv)
        os.environ.update(env)
        try:
            with lbm.Engine(p, ob, None) as eng:
                assert eng.info()["resident_steps"] > 0
                eng.run(steps); eng.sync()
                best = 1e9
                for _ in range(2):
                    t0 = time.perf_counter(); eng.run(steps); eng.sync(); best = min(best, (time.perf_counter() - t0) / steps * 1e6)
                same = np.array_equal(eng.cells().view(np.uint32), want)
            print(f"{nx:5d} x {ny:5d}  {v or '(default)':60s} {best:6.3f} us/step  bit-identical {same}", flush=True)
        except Exception as e:
            print(f"{nx:5d} x {ny:5d}  {v:60s} failed: {e}", flush=True)
        for k in env:
            del os.environ[k]
