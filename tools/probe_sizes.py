"""Device-copy probe (mp_stream_bandwidth) at several array sizes: what the GPU streams at config c2 size and beyond."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from manipulapy_amd import _hip
ctx = _hip.HipContext(0)
for mb in (98.304, 196.6, 393.2, 786.4):
    nb = int(mb * 1e6) & ~15
    for _ in range(2):
        c = ctx.stream_bandwidth(nb, reads=1, reps=50); m = ctx.stream_bandwidth(nb, reads=3, reps=50)
    print(f"arrays of {mb:7.1f} MB: copy {c:7.0f} GB/s   3 reads + 1 write {m:7.0f} GB/s   ({4*nb/ (m*1e9) * 1e3:.4f} ms per mix launch)")
