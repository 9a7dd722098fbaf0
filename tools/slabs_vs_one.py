"""8192^2 on ONE device as 1, 2 or 4 slabs (device-copy halos) and under band heights: does the multi-slab pipeline really
beat the single launch, and if so by what?   python tools/slabs_vs_one.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
import numpy as np
lbm = conftest.load_package()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nx = ny = 8192
tile = lbm.read_obstacles(os.path.join(conftest.GOLDEN, "inputs", "obstacles_1024x1024.dat"), 1024, 1024)
ob = lbm.tile_obstacles(tile, nx, ny)
p = lbm.Params(nx, ny, 8 * steps, 10, 0.1, 0.01, 1.85)
os.environ["LBM_HALO"] = "memcpy"
cases = [(1, {}), (1, {"LBM_FORCE_HALO": "1"}), (2, {"LBM_BAND_ROWS": "137"}), (1, {}), (1, {"LBM_FORCE_HALO": "1"}), (2, {"LBM_BAND_ROWS": "137"}),
         (1, {"LBM_FORCE_HALO": "1", "LBM_BAND_ROWS": "136"}), (2, {}), (4, {}), (4, {"LBM_BAND_ROWS": "137"}), (4, {"LBM_BAND_ROWS": "69"}), (1, {})]
for slabs, env in cases:
    os.environ.update(env)
    try:
        with lbm.Engine(p, ob, None, n_gpus=slabs) as eng:
            eng.run(steps); eng.sync()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); eng.run(steps); eng.sync(); best = min(best, (time.perf_counter() - t0) / steps * 1e6)
            i = eng.info()
            print(f"{slabs} slab(s) {env}: band {i['band_rows']} steps/launch {i['steps_per_launch']}: {best:8.2f} us/step {nx*ny/best:9.0f} MLUPS", flush=True)
    finally:
        for k in env:
            del os.environ[k]
