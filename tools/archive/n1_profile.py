#!/usr/bin/env python3
"""Where a step of the N = 1 NumPy API goes (BASELINE config 1 under mate_amd.evaluate): cProfile of one 1500-step episode."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mate_amd
from mate_amd.evaluate import evaluate, random_policy
env = mate_amd.MultiAgentTracking('MATE-4v2-9.yaml', max_episode_steps=1500)
env.seed(0)
evaluate(env, random_policy(0))
t0 = time.perf_counter(); h = []
evaluate(env, random_policy(1), history=h)
dt = time.perf_counter() - t0
print('%d steps in %.3f s: %.0f steps/s, %.1f us per step' % (len(h), dt, len(h) / dt, dt / len(h) * 1e6))
pr = cProfile.Profile(); pr.enable()
evaluate(env, random_policy(2))
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
