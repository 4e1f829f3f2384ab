"""CPU oracle for the NeRF volume-rendering hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``torch-nerf_amd/`` may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.
"""
