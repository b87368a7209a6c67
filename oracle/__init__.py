"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; nothing under ``event_representation_study_amd/`` does.
"""
from .oracle import *  # noqa: F401,F403
