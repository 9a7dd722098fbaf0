"""What the per-cell exception path (blocked cells) of the packed kernels costs: the same grid with the reference's
obstacle tile and with no obstacle at all.  python tools/no_obstacles.py NXxNY steps"""
import os, sys, time
sys.path.insert(0, "tests")
import numpy as np
import conftest
lbm = conftest.load_package()
nx, ny = (int(v) for v in sys.argv[1].split("x"))
steps = int(sys.argv[2])
tile = lbm.read_obstacles(os.path.join(conftest.GOLDEN, "inputs", "obstacles_1024x1024.dat"), 1024, 1024)
for label, t in (("reference tile", tile), ("no obstacles", np.zeros_like(tile))):
    p = lbm.Params(nx, ny, 4 * steps, 10, 0.1, 0.01, 1.85)
    with lbm.Engine(p, t, None, tiled=True) as eng:
        eng.run(steps); eng.sync()
        best = 1e9
        for _ in range(3):
            ms = eng.run_timed(steps); eng.sync()
            best = min(best, ms)
        print(f"{nx}x{ny} {label:15s}: {best:.4f} ms per step (device), info {eng.info()['band_rows']} rows", flush=True)
