#!/usr/bin/env python3
"""A/B variants of ONE source file: compiles csrc/<file> with extra -D flags and links it with the main build's other objects into
deepi2p_amd/lib/variants/<name>/libdeepi2p_hip.so (loaded through DI2P_LIB; travels to the GPU box with the snapshot).
    python tools/build_variant.py <file.hip> <name> [-DDI2P_...=...] ..."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepi2p_amd import build as B  # noqa: E402


def main():
    base, name, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build(verbose=False)
    vdir = os.path.join(B.LIBDIR, "variants", name)
    os.makedirs(vdir, exist_ok=True)
    obj = os.path.join(vdir, base.rsplit(".", 1)[0] + ".o")
    slp = [] if base in B.SLP_ON else ["-fno-slp-vectorize"]
    cmd = [B.HIPCC] + B.FLAGS + flags + slp + B.PER_FILE_FLAGS.get(base, []) + ["-x", "hip", "-c", os.path.join(B.CSRC, base), "-o", obj]
    subprocess.check_call(cmd)
    objs = [os.path.join(B.LIBDIR, "obj", s.rsplit(".", 1)[0] + ".o") for s in B.SOURCES if s != base] + [obj]
    lib = os.path.join(vdir, "libdeepi2p_hip.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


main()
