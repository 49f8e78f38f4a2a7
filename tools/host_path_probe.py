import numpy as np, time, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import manipulapy_amd as mp
from manipulapy_amd import _hip
ctx = _hip.HipContext(0)
sm, dyn, lim = mp.load_robot("ur5")
model = _hip.HipModel(dyn.S_list, dyn.Mlist_per_link, dyn.Glist, sm.M_list, lim)
ctx.specialize(model)
B,N,n=4096,1000,6
rng=np.random.default_rng(0)
st, en = (rng.uniform(-1, 1, (B, n)).astype(np.float32) for _ in range(2))
def t(fn, reps=5):
    fn(); t0=time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter()-t0)/reps*1e3
def fresh_touch():
    a=np.empty((B,N,n),np.float32); a.reshape(-1)[::1024]=0; return a
print("np.empty+touch 98MB: %.2f ms"%t(fresh_touch))
print("np.empty only: %.3f ms"%t(lambda: np.empty((B,N,n),np.float32)))
keep=np.empty((B,N,n),np.float32)
print("fused fresh out: %.2f ms"%t(lambda: ctx.traj_id_fused_host(model, st, en, 2.0, N, 5)))
print("fused reused out: %.2f ms"%t(lambda: ctx.traj_id_fused_host(model, st, en, 2.0, N, 5, out=keep)))
d=ctx.alloc(B*N*n*4)
print("download() fresh: %.2f ms"%t(lambda: d.download((B,N,n),np.float32)))
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
