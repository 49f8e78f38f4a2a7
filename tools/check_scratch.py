#!/usr/bin/env python3
"""Where the scratch accesses of mp_spec_id_co sit (round 5: the kernel is held to five waves per SIMD and the float64 path its first
workgroups carry spills): every scratch access should lie in the carried path, so that a float32 wave never touches scratch (0 on the robots of
BASELINE's inverse-dynamics configurations - UR5, Panda, panda7 - and on iiwa14; xarm6's max-ILP program spills ONE dword, its lane id, at
the top of the kernel and reloads it twice: 3 accesses per float32 wave).

    python tools/check_scratch.py [robot ...]        exit code 1 if a float32 wave could meet a scratch access"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manipulapy_amd import _hip, robots  # noqa: E402

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-fno-slp-vectorize", "-fno-signed-zeros",
         "-ffinite-math-only", "-mllvm", "-disable-machine-licm", "-w", "-DMP_SPECIALISED=1"]


def hot_path_scratch(robot, kernel="mp_spec_id_co_f0"):
    """(scratch accesses in the kernel, how many of them lie inside the float32 rows' code).  Compiled as the specialiser compiles the
    SECOND program (csrc/mp_jit.cpp part 1: -amdgpu-sched-strategy=max-ilp), where mp_spec_id_co lives."""
    t = robots.robot_tables(robot)
    m = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
    src = m.specialize_source(part=1)
    hip, asm = f"/tmp/chk_{robot}.hip", f"/tmp/chk_{robot}.s"
    open(hip, "w").write("#include <hip/hip_runtime.h>\n" + src)
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-I", os.path.join(ROOT, "manipulapy_amd", "csrc"), "--offload-device-only", "-S", "-o", asm, hip],
                   check=True)
    s = open(asm).read()
    i = s.index(f"\n{kernel}:")
    f = s[i:s.index(".Lfunc_end", i)].splitlines()
    scratch = [k for k, l in enumerate(f) if re.search(r"\bscratch_(load|store)", l)]
    # the kernel opens with `blockIdx.x < L.blocks ?`: the branch's target label is where the carried float64 path begins; everything
    # in front of that label is what a float32 wave can execute (its rows, the tail, the in-place fallback blocks)
    m = next(re.search(r"s_cbranch_scc[01]\s+(\.LBB\d+_\d+)", l) for l in f if "s_cbranch_scc" in l)
    lead_at = next(k for k, l in enumerate(f) if l.startswith(m.group(1) + ":"))
    inside = [k for k in scratch if k < lead_at]
    return len(scratch), len(inside)


if __name__ == "__main__":
    bad = 0
    for robot in sys.argv[1:] or ["ur5", "panda", "panda7", "xarm6", "iiwa14"]:
        total, inside = hot_path_scratch(robot)
        print(f"{robot}: {total} scratch accesses in mp_spec_id_co_f0, {inside} of them inside the float32 rows' code")
        bad += inside
    sys.exit(1 if bad else 0)
