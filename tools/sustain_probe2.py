#!/usr/bin/env python3
"""Development helper (GPU box): is the launch-time step of the headline kernel tied to the GAMES' position (moves played since seeding) or to
the time under load?  Env B is prepared first; env A (other seeds) then loads the GPU for `pre` launches and B's 1000 launches follow at once."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedAzul  # noqa: E402


def make(seed, G=4096, T=512):
    env = BatchedAzul(G)
    env.seed(seed)
    env.runner_init()
    env.runner_init()
    b = env.alloc_trajectory(T, packed_mask=True, mask_pitch=192, mask_bits=False)
    run = lambda: env.selfplay(T, b["mask"], b["action"], b["reward"], b["done"], packed=b["packed"])
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    return env, run


def blocks(env, run, launches, NB=10):
    env.timing_begin()
    for _ in range(launches):
        run()
    env.timing_end()
    s = env.timing_launch_ms()
    nb = len(s) // NB
    return ["%.4f" % (sum(s[i * nb:(i + 1) * nb]) / nb) for i in range(NB)]


for pre in (0, 300, 900):
    a, run_a = make(10 ** 6)
    b, run_b = make(0)
    time.sleep(1.0)
    for _ in range(pre):
        run_a()
    print("pre-load of %4d launches on other games, then 1000 launches of games seeded 0: block means (ms)" % pre, blocks(b, run_b, 1000))
    torch.cuda.synchronize()
    del a, b
# ... and games seeded elsewhere / a smaller batch
for seed, G in ((123456789, 4096), (0, 2048), (0, 8192)):
    e, run = make(seed, G)
    time.sleep(1.0)
    print("seed base %d, %d games: 1000 launches: block means (ms)" % (seed, G), blocks(e, run, 1000))
    del e
