"""Build libposegan_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so lives next to the
sources (pose-transfer_amd/lib/) so that it travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
# PG_TIMING_EXPERIMENTS=1: a SEPARATE library (libposegan_hip_timing.so, objects in obj_timing/) built with
# -DPG_TIMING_EXPERIMENTS — the per-workgroup phase stamps and the "wrong results, right times" K-loop experiments of
# csrc/igemm_bf16.hip exist only there (tools/conv_timeline.py); the production library carries none of them.
TIMING = os.environ.get("PG_TIMING_EXPERIMENTS") == "1"
OBJDIR = os.path.join(LIBDIR, "obj_timing" if TIMING else "obj")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
LIB = os.environ.get("PG_LIB") or os.path.join(LIBDIR, "libposegan_hip_timing.so" if TIMING else "libposegan_hip.so")
SOURCES = ["api.hip", "comm.hip", "optim.hip", "norm.hip", "losses.hip", "warp.hip", "pose_geom.hip", "edge.hip", "small_cin_wgrad.hip", "out_conv_dgrad.hip", "out_conv_fwd.hip", "igemm_conv.hip", "igemm_bf16.hip", "igemm_bf16_pair.hip", "igemm_bf16_quad.hip", "wgrad_bf16.hip", "stem_bf16.hip", "igemm_wgrad.hip"]
# -pragma-unroll-threshold: the epilogue loops over a wave's MFMA tiles MUST be fully unrolled (a rolled loop indexes the
# accumulator array at run time and the compiler moves it to scratch memory); the 4x2-tile bf16 kernel exceeds the default
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-pragma-unroll-threshold=1000000",
         "-I" + INCLUDE, "-I" + CSRC] + (["-DPG_TIMING_EXPERIMENTS"] if TIMING else [])


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src,) + tuple(extra))


def build_lib(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = (os.path.join(CSRC, "common.h"), os.path.join(INCLUDE, "posegan_hip.h"))
    conv_hdrs = hdrs + (os.path.join(CSRC, "igemm_common.h"), os.path.join(CSRC, "igemm_bf16_epi.h"))
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        if force or _newer(src, obj, conv_hdrs if s in ("igemm_conv.hip", "igemm_bf16.hip", "igemm_bf16_pair.hip", "igemm_bf16_quad.hip", "wgrad_bf16.hip", "stem_bf16.hip") else hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    if jobs:
        if verbose:
            print("[build] hipcc gfx950: %s" % ", ".join(os.path.basename(j[0]) for j in jobs), file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJDIR, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    if not TIMING:
        build_example()
    return LIB


EXAMPLE = os.path.join(LIBDIR, "cabi_smoke")


def build_example():
    """examples/cabi_smoke.cpp: a plain C++ consumer of the C ABI (no Python, no torch), next to the library."""
    src = os.path.join(os.path.dirname(PKG), "examples", "cabi_smoke.cpp")
    if not os.path.exists(src) or not _newer(src, EXAMPLE, (LIB, os.path.join(INCLUDE, "posegan_hip.h"))):
        return EXAMPLE
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + INCLUDE, src, "-L" + LIBDIR, "-lposegan_hip",
           "-Wl,-rpath,$ORIGIN", "-o", EXAMPLE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        # an example must not break the library build: warn (tests/test_gpu_kernels.py::test_c_abi_from_plain_cpp fails
        # loudly if the binary is missing)
        print("[build] WARNING: examples/cabi_smoke.cpp did not build:\n%s" % r.stderr[-2000:], file=sys.stderr)
    return EXAMPLE


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
