"""In-tree native build: `python -m alphapig_amd.build [host|hip|oracle|all]`.

  libalphapig_host.so  g++  (board + PUCT tree pool; -ffp-contract=off is load-bearing)
  libalphapig_hip.so   hipcc --offload-arch=gfx950 (kernels + evaluator C ABI)
  oracle/_build/...    gcc  (C restatement of the net: test infrastructure / CPU baseline)

hipcc cross-compiles for gfx950 without a GPU; the built .so files are git-ignored but
travel to the GPU box with the source snapshot.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(REPO, "include")
HOST_LIB = os.path.join(PKG, "libalphapig_host.so")
HIP_LIB = os.path.join(PKG, "libalphapig_hip.so")
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "_build", "libnet_ref.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build libalphapig_hip.so)")


def build_host(force=False):
    srcs = [os.path.join(CSRC, "host_tree.cpp"), os.path.join(INC, "alphapig_host.h")]
    if not force and _newer(HOST_LIB, srcs):
        return HOST_LIB
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
          "-Wall", "-Wextra", "-I" + INC, srcs[0], "-o", HOST_LIB])
    return HOST_LIB


def hip_sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build_hip(force=False):
    srcs = hip_sources()
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(INC, "alphapig_hip.h")]
    if not force and _newer(HIP_LIB, deps):
        return HIP_LIB
    extra = os.environ.get("APZ_HIPCC_EXTRA", "").split()     # experiment switches (-DAPZ_...), never needed for a normal build
    _run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
          "-ffp-contract=off", "-Wall", "-Wno-unused-result", "-I" + INC, "-I" + CSRC] + extra + srcs + ["-o", HIP_LIB])
    return HIP_LIB


def build_oracle(force=False):
    src = os.path.join(ORACLE_DIR, "net_ref.c")
    if not os.path.exists(src):
        return None
    os.makedirs(os.path.dirname(ORACLE_LIB), exist_ok=True)
    if not force and _newer(ORACLE_LIB, [src]):
        return ORACLE_LIB
    # -ffp-contract=fast: the vectorised forward (target("avx512f") / target("avx2,fma") functions) wants FMAs; the
    # plain restatement is compiled for the base ISA (-mavx2, no FMA), where the flag changes nothing
    _run(["gcc", "-O3", "-mavx2", "-fPIC", "-shared", "-ffp-contract=fast", "-Wall", src, "-lm", "-o", ORACLE_LIB])
    return ORACLE_LIB


def build_all(force=False):
    return {"host": build_host(force), "hip": build_hip(force), "oracle": build_oracle(force)}


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    force = "--force" in sys.argv
    out = {"host": build_host, "hip": build_hip, "oracle": build_oracle, "all": build_all}[what](force)
    print(out)
