#!/bin/bash
# round 6, call 8: the cleaned solver source (decided variants folded in, the others removed) against the build before the clean-up
set -u
ROOT=$(pwd)
python tools/dump_solve.py /tmp/new.npz | tail -1
DI2P_LIB=$ROOT/deepi2p_amd/lib/variants/prev/libdeepi2p_hip.so python tools/dump_solve.py /tmp/prev.npz | tail -1
python -c "
import numpy as np
a,b=np.load('/tmp/new.npz'),np.load('/tmp/prev.npz')
print('cleaned source bit-identical to the build before the clean-up:', all(a[k].tobytes()==b[k].tobytes() for k in a.files))"
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
