"""Builds deepi2p_amd/lib/libdeepi2p_hip.so (gfx950) with hipcc.  In-tree so the .so travels with
the repo snapshot to the GPU box; hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdeepi2p_hip.so")
SOURCES = ["common.cpp", "index_max.hip", "ball_query.hip", "point_ops.hip", "gemm.hip", "conv.hip", "solver.hip", "prep.hip", "pnp.hip", "rng.hip", "loss.hip", "train.hip", "winograd.hip", "stem.hip", "conv_x3.hip", "head_x3.hip", "stem_x3.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"] + os.environ.get("DI2P_EXTRA_HIPCC_FLAGS", "").split()
# hipcc's SLP vectoriser turns adjacent scalar fp32 adds / multiplies (epilogues, the Winograd transforms, the solver's fp32
# pre-filter) into packed v_pk_*_f32 instructions, which are an anti-lever on gfx950 next to MFMAs (MI355X_MICROARCH.md: +22..26
# cycles per packed op beside a matrix instruction; they also run while OTHER waves of the SIMD multiply).  Measured on the
# headline: 3106-3157 -> 3183-3200 frames/s with the vectoriser off.  Files listed here keep it (measured per file).
# (pnp.hip / prep.hip are not hot, and their fp32 label / inlier arithmetic is compared with numpy value for value: without the
# vectoriser hipcc contracts other multiply-add pairs into FMAs there and two parity tests see different roundings at cell borders.)
SLP_ON = set(os.environ.get("DI2P_SLP_ON", "pnp.hip prep.hip").split())
# Per-file flags.  solver.hip: hipcc's machine-level LICM hoists the fp64 constants of the LM update (tolerances, series coefficients) out
# of the sweep loop into VGPRs and the register allocator then SPILLS them -- every use became a scratch reload with a full wait, on
# the one lane whose latency the whole workgroup waits for.  Without that pass the kernel has no scratch at all (tools/kernel_resources.py)
# and measures 7.8 vs 8.2 ms per launch; `-mllvm -sink-insts-to-avoid-spills` (the targeted fix) measures 7.95.  IR-level LICM is unaffected.
PER_FILE_FLAGS = {"solver.hip": os.environ.get("DI2P_SOLVER_FLAGS", "-mllvm -disable-machine-licm").split()}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, variant=None, extra_flags=(), csrc=None):
    """variant: build into lib/variants/<name>/ with extra hipcc flags (A/B experiments, loaded through DI2P_LIB); csrc: other source tree"""
    if variant:
        return _build_variant(variant, list(extra_flags), csrc)
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "deepi2p_hip.h"))
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            slp = [] if (src in SLP_ON or not src.endswith(".hip")) else ["-fno-slp-vectorize"]
            cmd = [HIPCC] + FLAGS + slp + PER_FILE_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


def _build_variant(name, extra_flags, csrc):
    vdir = os.path.join(HERE, "lib", "variants", name)
    os.makedirs(os.path.join(vdir, "obj"), exist_ok=True)
    src_dir = csrc or CSRC
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(src_dir, src)
        obj = os.path.join(vdir, "obj", src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        slp = [] if (src in SLP_ON or not src.endswith(".hip")) else ["-fno-slp-vectorize"]
        cmd = [HIPCC] + FLAGS + extra_flags + slp + PER_FILE_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", obj]
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    lib = os.path.join(vdir, "libdeepi2p_hip.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
