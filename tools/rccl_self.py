"""The rank pipeline with RCCL self-exchange (a ring of one) under a chosen librccl: which library the engine bound,
what its communicator reports, and whether the fields equal a plain periodic single-slab run bit for bit.
python tools/rccl_self.py [--torch-first]      (LBM_RCCL_LIB=<path> forces a library)
Prints one JSON line."""
import json, os, sys
if "--torch-first" in sys.argv:
    import torch                      # maps torch's bundled libamdhip64 / librccl before the engine loads
sys.path.insert(0, "tests")
import numpy as np
import conftest
lbm = conftest.load_package()
p, ob = conftest.dataset("128x256")
steps = 61
with lbm.Engine(p, ob, None) as plain:
    plain.run(steps)
    want = plain.final_state()
    before = plain.rccl_info()
os.environ["LBM_FORCE_HALO"] = "1"
with lbm.Engine(p, ob, None, rank=0, world_size=1, unique_id=lbm.rccl_unique_id(), device=0) as eng:
    info = eng.rccl_info()
    eng.run(steps)
    got = eng.final_state()
    depth = eng.info()["steps_per_launch"]
maps = [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln or "libamdhip64" in ln]
print(json.dumps({"torch_first": "--torch-first" in sys.argv, "forced": os.environ.get("LBM_RCCL_LIB"),
                  "single_slab_loaded_rccl": before["loaded"], "rccl": info, "steps_per_pass": depth,
                  "fields_equal": all(np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)) for k in got),
                  "mapped": sorted(set(maps))}))
