"""CPU ORACLE for the DeMF fusion hot path - TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference's algorithm for the hot path, used solely as
the checker in ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``demf_amd/`` imports it.

Parity status (details in DESIGN.md "Oracle"):
  * in-tree reference code (heads/transformer/coder glue): PINNED - golden
    fixtures in tests/golden/ are produced by importing the real reference files
    (oracle/pin_reference.py) and the restatements here are checked against them;
  * third-party operators (mmdet3d.ops / mmcv.ops, absent from the reference
    tree and from this environment): "parity unpinned" against upstream binaries;
    cross-checked against independent implementations available in-container.
"""
