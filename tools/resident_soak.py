"""Long runs of the resident kernel against the launch-per-pass kernels on the same data set: the lattices must be
bit-identical (both are bit-identical to the oracle step by step) -- 1e10-1e11 seam-granule reads per run, any torn or
stale one would show.  python tools/resident_soak.py NAME steps [NAME steps ...]"""
import os, sys, time
sys.path.insert(0, "tests")
import numpy as np
import conftest
lbm = conftest.load_package()
args = sys.argv[1:]
for name, steps in zip(args[0::2], args[1::2]):
    steps = int(steps)
    p, ob = conftest.dataset(name)
    p.max_iters = steps
    out = {}
    for resident in ("0", "1"):
        os.environ["LBM_RESIDENT"] = resident
        with lbm.Engine(p, ob, None) as eng:
            t0 = time.perf_counter()
            for _ in range(4):                      # in pieces: launches of different lengths
                eng.run(steps // 4)
            eng.sync()
            dt = time.perf_counter() - t0
            out[resident] = (eng.cells(), eng.av_vels(steps // 4 * 4), dt, eng.info()["resident_steps"])
    same = np.array_equal(out["0"][0].view(np.uint32), out["1"][0].view(np.uint32))
    rel = float(np.max(np.abs(out["0"][1].astype(np.float64) - out["1"][1]) / np.abs(out["0"][1])))
    print(f"{name}: {steps} steps, per-pass {out['0'][2]:.2f} s, resident {out['1'][2]:.2f} s (resident_steps {out['1'][3]}): "
          f"lattices bit-identical: {same}; av_vels max rel diff {rel:.2e}; finite {bool(np.isfinite(out['1'][0]).all())}", flush=True)
    assert same
