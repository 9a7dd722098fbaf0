"""us per timestep of the LDS-tile kernel by shape and steps per launch (LBM_TILE_SHAPE, LBM_TILE_STEPS;
steps 0 = the one-step / two-step kernels).  python tools/tile_shapes.py [N ...] (square grids)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, numpy as np
lbm = conftest.load_package()
SHAPES = [("-", "0"), ("0", "4"), ("2", "2"), ("3", "3"), ("4", "4"), ("5", "2"), ("6", "2")]
if os.environ.get("TS_SHAPES"):      # e.g. TS_SHAPES="-:0,0:4,3:3"
    SHAPES = [tuple(x.split(":")) for x in os.environ["TS_SHAPES"].split(",")]
sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 768, 1024, 1536]
p0, ob0 = conftest.dataset("256x256")
for n in sizes:
    if n == 1024:
        p1, ob = conftest.dataset("1024x1024")
    else:
        ob = lbm.tile_obstacles(ob0, n, n)
    steps = max(600, int(6e9 / (n * n)) // 12 * 12)
    p = lbm.Params(n, n, 2 * steps, 10, 0.1, 0.005, 1.85)
    ref = None
    for shape, ts in SHAPES:
        os.environ["LBM_TILE_STEPS"] = ts
        os.environ["LBM_TILE_SHAPE"] = "0" if shape == "-" else shape
        with lbm.Engine(p, ob, None) as eng:
            eng.run(steps // 10); eng.sync()
            t0 = time.perf_counter(); eng.run(steps); eng.sync(); t2 = time.perf_counter()
            pr = eng.final_state()["pressure"]
        if ref is None: ref = pr
        print(f"{n}x{n} shape {shape} steps/launch {ts}: {1e6*(t2-t0)/steps:7.2f} us/step  same={np.array_equal(pr.view(np.uint32), ref.view(np.uint32))}", flush=True)
