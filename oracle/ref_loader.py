"""Loads oracle/_ref/index_max (the reference's index_max.cpp built by oracle/Makefile).

TEST INFRASTRUCTURE ONLY.  Present only where /root/reference was available at build
time; callers must handle ``None``.
"""
import glob
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def load_ref_index_max():
    cands = glob.glob(os.path.join(_HERE, "_ref", "index_max*.so"))
    if not cands:
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)
    old = sys.getdlopenflags()
    try:
        # the two CUDA entry points are undefined by construction: bind lazily, never call them
        sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
        spec = importlib.util.spec_from_file_location("index_max", cands[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    return mod
