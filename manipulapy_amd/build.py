"""Builds libmanipula_hip.so (gfx950) in-tree with hipcc.  `python -m manipulapy_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmanipula_hip.so")
ARCH = "gfx950"
SOURCES = ["mp_kernels.hip", "mp_capi.cpp", "mp_comm.cpp", "mp_model_compile.cpp"]
HEADERS = ["mp_core.h", "mp_model.h", "mp_kernels.h", "mp_model_compile.h", os.path.join("..", "..", "include", "manipula_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True, variant: str = "", defines: tuple = ()) -> str:
    """`variant` / `defines` build an experimental copy (libmanipula_hip_<variant>.so with -D flags) for
    A/B measurements: select it at run time with MANIPULAPY_HIP_LIB=<path>."""
    hipcc = _hipcc()
    objdir = os.path.join(PKG, "build" + (f"_{variant}" if variant else ""))
    lib = LIB if not variant else os.path.join(PKG, f"libmanipula_hip_{variant}.so")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    common = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
              "-ffp-contract=fast", "-fno-slp-vectorize"] + [f"-D{d}" for d in defines]
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + hdrs):
            cmd = [hipcc] + common + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    if force or _stale(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs + ["-ldl"]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    variant = ""
    if "--variant" in args:
        variant = args[args.index("--variant") + 1]
    defs = tuple(a[2:] for a in args if a.startswith("-D"))
    print(build(force="--force" in sys.argv, variant=variant, defines=defs))
