"""CPU oracle for the TSM hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  The product package
(``background-debiased-video-cil_amd`` / alias ``bdvcil_amd``) never imports it.
"""
