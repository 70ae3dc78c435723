#!/usr/bin/env python3
"""Development helper (GPU box): what do the per-launch HIP event pairs of a timed region cost the whole-job rate?  Wall clock of 200 back-to-back
launches of the headline kernel (4096 games, 512 moves, all outputs) with no event, with one torch event after each launch, and inside
azul_timing_begin / _end (an event pair around every launch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedAzul  # noqa: E402

G, T, N = 4096, 512, 200
env = BatchedAzul(G)
env.seed(0)
env.runner_init()
env.runner_init()
b = env.alloc_trajectory(T, packed_mask=True, mask_pitch=192, mask_bits=False)
run = lambda: env.selfplay(T, b["mask"], b["action"], b["reward"], b["done"], packed=b["packed"])
for _ in range(50):
    run()
torch.cuda.synchronize()


def wall(mode):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)] if mode == "one" else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "pairs":
        env.timing_begin()
    if evs:
        evs[0].record()
    for i in range(N):
        run()
        if evs:
            evs[i + 1].record()
    k = None
    if mode == "pairs":
        _, _, kms, kn = env.timing_end()
        k = kms / kn
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if evs:
        ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
        k = sum(ts) / N
    return dt / N * 1e3, k


for rep in range(3):
    for mode in ("none", "one", "pairs"):
        w, k = wall(mode)
        print("%-5s wall %.4f ms per launch%s" % (mode, w, "" if k is None else "; event-timed %.4f ms" % k), flush=True)
