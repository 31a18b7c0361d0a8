"""Builds vers_amd/lib/libvers_hip.so (HIP, gfx950 only) in-tree with hipcc.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
OBJDIR = os.path.join(_HERE, "build")
LIB = os.path.join(LIBDIR, "libvers_hip.so")

ARCH = "gfx950"
# -ffp-contract=off: the reference rounds every product and sum separately (no FMA); see csrc/scan.hip.h
CXXFLAGS = os.environ.get("VERS_EXTRA_CXXFLAGS", "").split() + ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", f"--offload-arch={ARCH}",
            "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libvers_hip.so cannot be built")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


RCCL_SRC = os.path.join(CSRC, "rccl", "vers_comm_rccl.hip")
RCCL_LIB = os.path.join(LIBDIR, "libvers_rccl.so")


def build_rccl(force: bool = False, verbose: bool = False) -> str:
    """libvers_rccl.so: the optional RCCL adapter (include/vers_comm_rccl.h).  Links librccl; libvers_hip.so does not."""
    os.makedirs(LIBDIR, exist_ok=True)
    deps = [RCCL_SRC, os.path.join(os.path.dirname(_HERE), "include", "vers_comm_rccl.h"),
            os.path.join(os.path.dirname(_HERE), "include", "vers_hip.h")]
    if force or _stale(RCCL_LIB, deps):
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(_hipcc())), "lib")
        cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-Wall", RCCL_SRC, "-o", RCCL_LIB,
               f"-L{rocm_lib}", "-lrccl", f"-Wl,-rpath,{rocm_lib}"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return RCCL_LIB


TEST_SRC_DIR = os.path.join(CSRC, "testhooks")
TEST_LIB = os.path.join(LIBDIR, "libvers_hip_test.so")


def build_test_hooks(force: bool = False, verbose: bool = False) -> str:
    """libvers_hip_test.so (include/vers_hip_test.h): the TEST / emulation hooks, a second library linked AGAINST libvers_hip.so --
    the product library exports nothing named *test* (tests/test_abi.py)."""
    hipcc = _hipcc()
    srcs = sorted(os.path.join(TEST_SRC_DIR, f) for f in os.listdir(TEST_SRC_DIR) if f.endswith(".hip"))
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".hip.h", ".h"))]
    deps += [os.path.join(os.path.dirname(_HERE), "include", h) for h in ("vers_hip.h", "vers_hip_test.h")] + [LIB]
    if force or _stale(TEST_LIB, deps):
        cmd = [hipcc] + CXXFLAGS + ["-shared"] + srcs + ["-o", TEST_LIB, f"-L{LIBDIR}", "-lvers_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return TEST_LIB


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".hip.h", ".h"))]
    headers.append(os.path.join(os.path.dirname(_HERE), "include", "vers_hip.h"))
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(OBJDIR, src[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            cmd = [hipcc] + CXXFLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_test_hooks(force, verbose)
    try:  # the adapter is OPTIONAL (libvers_hip.so does not link RCCL): a box without librccl under the hipcc prefix still builds the library
        build_rccl(force, verbose)
    except (subprocess.CalledProcessError, OSError) as e:
        import sys
        print(f"[vers build] libvers_rccl.so NOT built ({e}): the multi-GPU exchanges then need torch.distributed (vers_amd.dist)", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
