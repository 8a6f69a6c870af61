"""TEST INFRASTRUCTURE ONLY.  CPU oracle for the overlap hot path.

Nothing under ``phasm_amd/`` imports this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.
"""
