"""Wall time per timestep of the synchronous, the stale-halo and the freshest-available pipeline for several slabs on ONE device
(the only multi-slab configuration a 1-GPU box offers): shows the launch/exchange overhead each
pipeline adds to the single-slab time.    python tools/stale_vs_sync.py NXxNY steps [transport]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
import numpy as np
lbm = conftest.load_package()
nx, ny = (int(v) for v in sys.argv[1].split("x"))
steps = int(sys.argv[2])
transport = sys.argv[3] if len(sys.argv) > 3 else "memcpy"
os.environ["LBM_HALO"] = transport
tile = lbm.read_obstacles(os.path.join(conftest.GOLDEN, "inputs", "obstacles_1024x1024.dat"), 1024, 1024)
ob = lbm.tile_obstacles(tile, nx, ny) if (nx, ny) != (1024, 1024) else tile
p = lbm.Params(nx, ny, 3 * steps, 10, 0.1, 0.01, 1.85)
for slabs in (1, 2, 4, 8):
    for mode in ("sync", "stale", "freshest"):
        if slabs == 1:
            if mode != "sync":
                continue
            os.environ.pop("LBM_FORCE_HALO", None)
        kw = dict(n_gpus=slabs)
        if transport == "rank":
            if slabs > 1:
                continue
        with lbm.Engine(p, ob, None, **kw) as eng:
            eng.set_halo_mode(mode)
            eng.run(steps // 4); eng.sync()
            t0 = time.perf_counter(); eng.run(steps); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
            note = ""
            if mode == "freshest":
                log = eng.halo_log(eng.info()["steps_done"])[-steps:]
                note = f", looks that found this step's row: {np.mean([(log & 1).mean(), ((log >> 1) & 1).mean()]):.2f}"
            print(f"{nx}x{ny} {slabs} slab(s) {transport:6s} {mode:8s}: host issue {1e6*(t1-t0)/steps:8.2f} us/step, "
                  f"wall {1e6*(t2-t0)/steps:8.2f} us/step, {nx*ny*steps/(t2-t0)/1e6:9.0f} MLUPS{note}", flush=True)
