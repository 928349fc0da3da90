"""Closed-loop Greedy-vs-Greedy census at batch scale (not collected by pytest; run by hand on an MI355X): the body of
tests/test_gpu_policies.py::test_greedy_policies_batch_vs_oracle at hundreds of environments and hundreds of steps."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import oracle as O
import test_gpu_policies as T
for config, n, steps in (('MATE-8v8-9.yaml', 384, 250), ('MATE-4v8-9.yaml', 512, 250), ('MATE-Navigation.yaml', 256, 150)):
    t0 = time.time()
    T.test_greedy_policies_batch_vs_oracle.__wrapped__(config, n, steps, O) if hasattr(T.test_greedy_policies_batch_vs_oracle, '__wrapped__') else T.test_greedy_policies_batch_vs_oracle(config, n, steps, O)
    print(f'{config}: {n} envs x {steps} closed-loop Greedy-vs-Greedy steps against the oracle agents: joint actions within 1e-8, masks / goals / bounties / freights / deliveries exact ({time.time() - t0:.0f} s)')
