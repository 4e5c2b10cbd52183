"""CPU oracle for the xumx-sliCQ-V2 demix hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline.  The product path (``xumx_slicq_amd``) never imports this package
and fails loudly when its HIP library is missing.

The reference (sevagh/xumx-sliCQ, branch v2) is pure Python/PyTorch, so the
restatement is numpy / torch-CPU fp32 (no C).  Every function cites the
reference file:line it restates.  Parity is PINNED: ``oracle/make_golden.py``
imports the reference itself in the development container and writes the
fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every
oracle stage against them (the reference's own tests hold no numeric golden
values for this path).
"""
