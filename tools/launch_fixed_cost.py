#!/usr/bin/env python3
"""Development helper (GPU box): the fixed cost of a launch of the headline kernel -- mean interval between per-launch events for T = 1 .. 512 moves
per launch at 4096 games (all outputs), 200 launches each; the intercept of time(T) is prologue (state, MT19937 staging + tempering, table) +
epilogue (write-back) + launch gap."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedAzul  # noqa: E402

G = 4096
env = BatchedAzul(G)
env.seed(0)
env.runner_init()
env.runner_init()
b = env.alloc_trajectory(512, packed_mask=True, mask_pitch=192, mask_bits=False)
for T in (512, 1, 2, 4, 16, 64, 128, 256, 512):
    v = {k: t[:T] for k, t in b.items()}
    run = lambda: env.selfplay(T, v["mask"], v["action"], v["reward"], v["done"], packed=v["packed"])
    for _ in range(20):
        run()
    N = 200
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(N):
        run()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(N))
    print("T = %3d: median %.4f ms  min %.4f ms  -> %.3f us per move" % (T, ts[N // 2], ts[0], ts[N // 2] * 1e3 / T), flush=True)
