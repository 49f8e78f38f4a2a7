import sys, os, time
sys.path.insert(0, os.getcwd())
from manipulapy_amd import _hip
ctx = _hip.HipContext(0)
nb = (22544384000 // 4) & ~15
for reps in (20, 20, 300, 1500, 3000, 20):
    t0 = time.perf_counter()
    nt = ctx.stream_bandwidth_mix(nb, 1, 3, reps, nontemporal=True)
    pl = ctx.stream_bandwidth_mix(nb, 1, 3, reps, nontemporal=False)
    print(f"reps {reps:5d}: non-temporal {nt:7.1f} GB/s  plain {pl:7.1f} GB/s   ({time.perf_counter() - t0:.1f} s)")
ctx.destroy()
