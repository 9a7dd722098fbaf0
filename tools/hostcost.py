"""Host issue cost vs wall time per timestep of the halo pipelines on a tiny grid (256x256), where the
device work is negligible: python tools/hostcost.py self|memcpy1|rccl1|memcpy2|memcpy4|rank1
(set LBM_FORCE_HALO=1 for the single-slab halo variants)."""
import sys,time,os; sys.path.insert(0,"tests"); import conftest; lbm=conftest.load_package()
p,ob=conftest.dataset("256x256")
p.max_iters=30000
def run(n_gpus, label, **kw):
    eng=lbm.Engine(p,ob,None,n_gpus=n_gpus,**kw); eng.run(500); eng.sync(); t=time.perf_counter(); eng.run(5000); t1=time.perf_counter(); eng.sync(); t2=time.perf_counter()
    print("%-28s host-issue us/step %.2f   wall us/step %.2f"%(label,(t1-t)/5000*1e6,(t2-t)/5000*1e6)); eng.close()
mode=sys.argv[1]
if mode=="self": run(1,"1 slab, no halo")
if mode=="memcpy1": run(1,"1 slab, memcpy self-halo")
if mode=="rccl1": run(1,"1 slab, rccl self-halo")
if mode=="memcpy2": run(2,"2 slabs, memcpy")
if mode=="memcpy4": run(4,"4 slabs, memcpy")
if mode=="rank1": run(1,"rank api, rccl self-halo",rank=0,world_size=1,unique_id=lbm.rccl_unique_id(),device=0)
