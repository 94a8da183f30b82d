"""Builds librsgpu.so (hand-written HIP for gfx950 + the C-ABI) in-tree with hipcc.

    python -m build  (from this directory)  or  build.build()

The shared object is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librsgpu.so")
OUT_STAMPS = os.path.join(HERE, "librsgpu_stamps.so")       # RS_STAMPS=1 build (in-kernel timers): never the product library
ARCH = "gfx950"

# (source, extra flags).  The f32-gate kernels are compiled without FMA
# contraction so that they execute the oracle's IEEE operations one for one.
SOURCES = [
    ("context.hip", []),
    ("hamming.hip", []),
    ("triangulate.hip", ["-ffp-contract=off"]),
    ("reproj_match.hip", ["-ffp-contract=off"]),
    ("tracks.hip", ["-ffp-contract=off"]),
    ("ba.hip", ["-munsafe-fp-atomics"]),
    ("ba_solve.hip", ["-munsafe-fp-atomics"]),
    ("ba_solve_big.hip", ["-munsafe-fp-atomics"]),
    ("ba_imu.hip", ["-munsafe-fp-atomics"]),
    ("ba_schur.hip", ["-munsafe-fp-atomics"]),
    ("ba_update.hip", ["-munsafe-fp-atomics"]),
    ("ba_round.hip", ["-munsafe-fp-atomics"]),
    ("map.hip", []),
    ("host.cpp", ["-ffp-contract=off"]),
    ("pose_graph.cpp", ["-ffp-contract=off"]),
]
STAMPS = ["-DRS_STAMPS=1"] if os.environ.get("RS_STAMPS") else []
# A/B builds: RS_VARIANT=<name> RS_DEFS="-DK7_V2=0 ..." -> librsgpu_<name>.so from objects of its own (tools/ab_time.py
# loads it through RS_LIB); never the product library
VARIANT = os.environ.get("RS_VARIANT", "")
STAMPS = STAMPS + os.environ.get("RS_DEFS", "").split() if VARIANT else STAMPS
COMMON = STAMPS + ["-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-pass-failed", f"--offload-arch={ARCH}"]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: librsgpu cannot be built (there is no CPU fallback)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    cc = hipcc()
    objdir = os.path.join(HERE, "build_" + VARIANT if VARIANT else ("build_stamps" if STAMPS else "build"))     # (instrumented objects never mix with the product's)
    os.makedirs(objdir, exist_ok=True)
    # every header of csrc/ is a dependency of every object: a stale .so must never ship
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    headers += [os.path.join(HERE, "..", "include", "rsgpu.h"), __file__]
    objs = []
    rebuilt = False
    for src, extra in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + headers):
            lang = ["-x", "hip"] if src.endswith(".hip") else []
            cmd = [cc] + COMMON + extra + lang + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            rebuilt = True
    out = os.path.join(HERE, f"librsgpu_{VARIANT}.so") if VARIANT else (OUT_STAMPS if STAMPS else OUT)
    if rebuilt or not os.path.exists(out):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
