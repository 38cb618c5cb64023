#!/usr/bin/env python3
"""Register / spill / scratch / LDS use of every kernel in a HIP source, from hipcc's own remarks (no GPU needed):
    python tools/kernel_resources.py deepi2p_amd/csrc/solver.hip [name-filter]
"""
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from deepi2p_amd import build as B  # noqa: E402


def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    base = src.rsplit("/", 1)[-1]
    slp = [] if base in B.SLP_ON else ["-fno-slp-vectorize"]
    cmd = [B.HIPCC] + B.FLAGS + slp + B.PER_FILE_FLAGS.get(base, []) + ["-x", "hip", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
    cur = None
    rows = []
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+\[?[A-Za-z/ ]*\]?):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    print("%-90s %5s %5s %5s %6s %6s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "vspill", "sspill", "scratch", "occ", "LDS"))
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r["name"])
        n = re.sub(r"\(.*", "", n)
        if filt and filt not in n:
            continue
        print("%-90s %5s %5s %5s %6s %6s %7s %4s %7s" % (n[:90], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill", r.get("VGPR Spill")),
                                                      r.get("SGPRs Spill", r.get("SGPR Spill")), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))


main()
