#!/usr/bin/env python3
"""Development helper (GPU box): how fast do never-ending games (hazard H9) accumulate in a batch that just keeps playing?
tools/h9_census.py [launches] [every] [move_limit]: the bench's batch (4096 games seeded 0, 512 moves per launch, all outputs) for `launches`
launches; every `every` launches the mean launch time of the last 100 and the games whose current episode has lasted more than 100 rounds."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedAzul  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
EVERY = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
LIMIT = int(sys.argv[3]) if len(sys.argv) > 3 else 0
G, T = 4096, 512
env = BatchedAzul(G)
env.seed(0)
if LIMIT:
    env.set_move_limit(LIMIT)
env.runner_init()
env.runner_init()
b = env.alloc_trajectory(T, packed_mask=True, mask_pitch=192, mask_bits=False)
run = lambda: env.selfplay(T, b["mask"], b["action"], b["reward"], b["done"], packed=b["packed"])
print("4096 games seeded 0, 512 moves per launch, move limit %d" % LIMIT, flush=True)
t0 = time.perf_counter()
done = 0
while done < N:
    for _ in range(EVERY - 100):
        run()
    env.timing_begin()
    for _ in range(100):
        run()
    _, _, kms, kn = env.timing_end()
    done += EVERY
    rec = env.get_records()
    odd = np.flatnonzero(rec["turn_counter"] > 100)
    ep = int(np.array(env.counters()["episodes"]).sum())
    print("launch %6d (%.2e game-moves, %5.1f s): launch %.4f ms, games in an episode of more than 100 rounds: %d %s; episodes finished %d" % (
        done, float(done) * G * T, time.perf_counter() - t0, kms / kn, len(odd), odd[:12].tolist(), ep), flush=True)
