"""CPU oracles for the TSM hot path (``tsm_oracle``), the representation path (``repr_oracle``) and the augmentation /
crop stages of the data path (``augment_oracle``) -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package, and only as the checker.  The product package
(``background-debiased-video-cil_amd`` / alias ``bdvcil_amd``) never imports it.
"""
