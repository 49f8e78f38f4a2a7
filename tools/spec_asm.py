#!/usr/bin/env python3
"""Dev helper: the ISA of a robot's specialised kernels without a GPU.

    python tools/spec_asm.py xarm6 mp_spec_fd_traj_tm_f1 [more kernels] [-DNAME ...]

Takes the very translation unit the run-time specialiser hands to hiprtc (mp_model_specialize_source), compiles it with
hipcc for gfx950 with the specialiser's flags (device only, -S) into /tmp/spec_<robot>_<part>.s and prints per kernel: VGPRs,
scratch, LDS, the instruction histogram and the s_waitcnt vmcnt values in program order."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from manipulapy_amd import _hip, robots  # noqa: E402

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-fno-slp-vectorize", "-fno-signed-zeros",
         "-ffinite-math-only", "-mllvm", "-disable-machine-licm", "-w", "-DMP_SPECIALISED=1"]


def main():
    robot = sys.argv[1]
    kernels = [a for a in sys.argv[2:] if not a.startswith("-")]
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    t = robots.robot_tables(robot)
    m = _hip.HipModel(t["S_list"], t["Mlist_per_link"], t["Glist"], t["M_ee"], t["joint_limits"])
    # two translation units per robot (csrc/mp_jit.cpp): the float32 one-row inverse dynamics (mp_spec_id_s / _co) is the second,
    # compiled with the max-ILP scheduling strategy; every other kernel the first
    s = ""
    for part, more in ((0, []), (1, ["-mllvm", "-amdgpu-sched-strategy=max-ilp"])):
        src = m.specialize_source(part=part)
        hip = f"/tmp/spec_{robot}_{part}.hip"
        asm = f"/tmp/spec_{robot}_{part}.s"
        open(hip, "w").write("#include <hip/hip_runtime.h>\n" + src)
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + more + extra + ["-I", os.path.join(ROOT, "manipulapy_amd", "csrc"), "--offload-device-only", "-S",
                                                                          "-o", asm, hip], check=True)
        s += open(asm).read()
    for key in kernels:
        m_ = re.search(r"^(" + re.escape(key) + r"):", s, re.M)
        if not m_:
            print(key, "not found")
            continue
        i = m_.end()
        j = s.index(".Lfunc_end", i)
        f = s[i:j]
        ops = collections.Counter(mm.group(1) for mm in re.finditer(r"^\s+([a-z_0-9]+)\s", f, re.M))
        valu = sum(v for k, v in ops.items() if k.startswith("v_"))

        def meta(name):
            mm = re.search(r"\.amdhsa_kernel " + re.escape(key) + r"\b.*?" + re.escape(name) + r" (\d+)", s, re.S)
            return mm.group(1) if mm else "?"

        print(f"{key}: VALU {valu} SALU {sum(v for k, v in ops.items() if k.startswith('s_'))} "
              f"vgpr {meta('.amdhsa_next_free_vgpr')} accum_offset {meta('.amdhsa_accum_offset')} "
              f"scratch {meta('.amdhsa_private_segment_fixed_size')} lds {meta('.amdhsa_group_segment_fixed_size')} "
              f"scratch-ops {sum(v for k, v in ops.items() if 'scratch' in k)}")
        print("   ", ops.most_common(24))
        waits = re.findall(r"s_waitcnt ([^\n]*)", f)
        print("    waitcnt:", " | ".join(w.strip() for w in waits[:60]))


if __name__ == "__main__":
    main()
