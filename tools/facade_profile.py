#!/usr/bin/env python3
"""Diagnostic: cProfile of BASELINE configs[0] through the drop-in shims (random.seed(s); GameRunner(); reset(); get_valid_moves ->
RandomAgent.get_a_output -> GameRunner.step), 40 episodes, sorted by internal time -- where the host side of a facade call spends
its microseconds (LABNOTES.md 8).  Needs a GPU.   python tools/facade_profile.py"""
import cProfile, pstats, sys, random, io
sys.path.insert(0, ".")
sys.path.insert(0, "integration")
import torch
from azulnet.game_runner import GameRunner, RandomAgent
agent = RandomAgent()
def episode(seed):
    random.seed(seed)
    r = GameRunner(); r.reset()
    done = False
    while not done:
        mask = r.get_valid_moves()
        a = agent.get_a_output(None, torch.from_numpy(mask[None, :]))
        _, done = r.step(a)
episode(999)
pr = cProfile.Profile(); pr.enable()
for s in range(40): episode(s)
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(32); print(st.getvalue()[:7000])
