"""CPU oracle for the MultiAgentTracking step path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``mate_amd``) never does.
"""
from oracle.oracle import *  # noqa: F401,F403
