"""Sustained run: python tools/soak.py NXxNY steps -- one lbm_run call per 10 % of the steps; prints MLUPS
per chunk, the av_vels range and the total-density drift."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, numpy as np
lbm = conftest.load_package()
nx, ny = (int(v) for v in sys.argv[1].split("x")); steps = int(sys.argv[2])
tile = lbm.read_obstacles(os.path.join(conftest.GOLDEN, "inputs", "obstacles_1024x1024.dat"), 1024, 1024)
ob = lbm.tile_obstacles(tile, nx, ny)
p = lbm.Params(nx, ny, steps, 10, 0.1, 0.01, 1.85)
with lbm.Engine(p, ob, None) as eng:
    m0 = eng.total_density()
    t_all = time.perf_counter()
    for k in range(10):
        t0 = time.perf_counter(); eng.run(steps // 10); eng.sync(); dt = time.perf_counter() - t0
        print(f"chunk {k}: {nx*ny*(steps//10)/dt/1e6:9.0f} MLUPS", flush=True)
    wall = time.perf_counter() - t_all
    av = eng.av_vels(steps // 10 * 10)
    m1 = eng.total_density()
print(f"{nx}x{ny} x {steps//10*10} steps: {wall:.2f} s = {nx*ny*(steps//10*10)/wall/1e6:.0f} MLUPS sustained; av_vels finite={bool(np.isfinite(av).all())} "
      f"({av[0]:.3e} -> {av[-1]:.3e}, monotone={bool((np.diff(av) > 0).all())}); density drift {m1/m0-1:+.2e}")
