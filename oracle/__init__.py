"""CPU oracle for the DeepI2P registration hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The product
path (``deepi2p_amd``) never imports this package and fails loudly when its HIP
library is missing.

Every function cites the reference file:line (relative to /root/reference) whose
behaviour it restates.  Pin status per piece is stated in each module header and
in DESIGN.md.
"""
