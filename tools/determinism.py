import sys, os
sys.path.insert(0, "tests")
import numpy as np, conftest
lbm = conftest.load_package()
tile = lbm.read_obstacles(os.path.join(conftest.GOLDEN, "inputs", "obstacles_1024x1024.dat"), 1024, 1024)
for nx, ny, steps in ((8192, 8192, 401), (16384, 16384, 101), (1024, 1024, 2001), (4096, 4096, 801)):
    outs = []
    for _ in range(3):
        p = lbm.Params(nx, ny, steps, 10, 0.1, 0.01, 1.85)
        with lbm.Engine(p, tile, None, tiled=True) as e:
            e.run(steps)
            f = e.final_state()
            outs.append((f["pressure"].copy(), f["u"].copy(), e.av_vels(steps).copy()))
    same = all(np.array_equal(outs[0][k].view(np.uint32), o[k].view(np.uint32)) for o in outs[1:] for k in range(3))
    print(nx, ny, steps, "three runs bit-identical:", same, "finite:", bool(np.isfinite(outs[0][2]).all()), flush=True)
