"""Where a timestep of the resident kernel goes: phase sums from the instrumented build
(hipcc ... -DLBM_RESIDENT_PROFILE -o tools/liblbm_hip_prof.so; LBM_LIB points the package at it).
LBM_LIB=tools/liblbm_hip_prof.so python tools/resident_profile.py 128x128,1024x1024 [steps]"""
import os, sys, time
sys.path.insert(0, "tests")
import numpy as np
import conftest
lbm = conftest.load_package()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
for g in sys.argv[1].split(","):
    nx, ny = (int(v) for v in g.split("x"))
    try:
        ob = conftest.dataset(g)[1]
    except Exception:
        ob = np.zeros((ny, nx), dtype=np.int32); ob[0, :] = 1; ob[-1, :] = 1; ob[:, 0] = 1; ob[:, -1] = 1
    with lbm.Engine(lbm.Params(nx, ny, 2 * steps, 10, 0.1, 0.01, 1.85), ob, None) as eng:
        eng.run(steps); eng.sync()
        t0 = time.perf_counter(); eng.run(steps); eng.sync()
        print(f"{g}: {(time.perf_counter() - t0) / steps * 1e6:.3f} us/step including the profile's own sync and copy", flush=True)
