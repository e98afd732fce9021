"""Build libgsr.so (hand-written HIP for gfx950) in-tree with hipcc.  No torch headers are involved: the library
is a plain C-ABI shared object (include/gsr.h) that the Python side loads with ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgsr.so")
ARCH = "gfx950"

# (source, extra flags).  Integer / bit-exact stages are built without FMA contraction so they match the oracle
# bit for bit; the blend and backward kernels keep the default contraction (their contract is 1e-4).
SOURCES = [
    ("geometry.hip", ["-ffp-contract=off"]),
    ("radix_sort.hip", ["-ffp-contract=off"]),
    ("binning_bucket.hip", ["-ffp-contract=off"]),
    ("knn.hip", ["-ffp-contract=off"]),
    # the blend kernels are bound by VALU issue: packed-f32 formation by the SLP vectoriser costs register-packing moves and
    # buys nothing (v_pk_fma_f32 issues at half the rate of v_fma_f32 on gfx950, profiles/r2_ubench_valu_rate.txt)
    ("blend_fwd.hip", ["-fno-slp-vectorize"]),
    ("blend_bwd.hip", ["-fno-slp-vectorize"]),
    ("preprocess_bwd.hip", []),
    ("lbs.hip", []),
    ("attributes.hip", []),
    ("activations.hip", []),
    ("pose.hip", []),
    ("sh_exchange.hip", []),
    ("rows.hip", []),
    ("ssim.hip", []),
    ("gemv.hip", []),
    ("loss.hip", []),
    ("probe.hip", []),
    ("mlp.hip", []),
    ("gsr_api.hip", []),
]
COMMON = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(force=False, verbose=False, save_temps=False, experiments=None):
    """experiments (default: the environment's GSR_BUILD_EXPERIMENTS == "1"): also compile the kernels that were built, measured and
    not adopted (gsr_has_experiments() reports it; their knob values are refused otherwise)."""
    if experiments is None:
        experiments = os.environ.get("GSR_BUILD_EXPERIMENTS") == "1"
    objdir = os.path.join(HERE, "build_experiments" if experiments else "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [os.path.join(HERE, "..", "include", "gsr.h"),
                                                                                       os.path.abspath(__file__)]
    hdr_m = max(os.path.getmtime(h) for h in hdrs)
    objs, rebuilt = [], False
    procs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_m):
            cmd = [_hipcc()] + COMMON + (["-DGSR_BUILD_EXPERIMENTS"] if experiments else []) + extra + ["-c", s, "-o", o]
            if save_temps:
                cmd += ["-save-temps=obj"]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, cwd=objdir)))
            rebuilt = True
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    stamp = os.path.join(objdir, ".linked")
    if rebuilt or not os.path.exists(LIB) or not os.path.exists(stamp):  # (the stamp: the other flavour may have linked LIB last)
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        for d in ("build", "build_experiments"):
            other = os.path.join(HERE, d, ".linked")
            if os.path.exists(other):
                os.remove(other)
        open(stamp, "w").close()
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, save_temps="--save-temps" in sys.argv,
                experiments=True if "--experiments" in sys.argv else None))
