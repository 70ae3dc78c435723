#!/usr/bin/env python3
"""Development helper: window time of the network-opponent rollout kernel (product build), event-timed per launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout  # noqa: E402

torch.manual_seed(0)
for opp, trace in (("random", 0), ("net", 0), ("net", 8)):
    o = BatchedActorCritic(136, 180, 180) if opp == "net" else opp
    ro = PolicyRollout(BatchedActorCritic(136, 180, 180), n_games=4096, parts=1, window=32, opponent=o, persistent=True, opponent_trace=trace)
    for _ in range(3):
        ro.run_window()
    ro.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    s = ro.streams[0]
    ev[0].record(s)
    for i in range(20):
        ro.run_window()
        ev[i + 1].record(s)
    ro.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(20)]
    print(opp, trace, "window ms: min %.3f median %.3f max %.3f" % (min(ts), sorted(ts)[10], max(ts)))
    del ro

# what do bench.py's per-window reductions cost on the rollout's stream?
o = BatchedActorCritic(136, 180, 180)
ro = PolicyRollout(BatchedActorCritic(136, 180, 180), n_games=4096, parts=1, window=32, opponent=o, persistent=True)
for _ in range(3):
    ro.run_window()
ro.synchronize()
s = ro.streams[0]
acc = torch.zeros(2, dtype=torch.int64, device=ro.device)
for variant in ("none", "sum_u8", "full"):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for i in range(20):
        tr = ro.run_window()
        with torch.cuda.stream(s):
            if variant == "sum_u8":
                acc[0] += tr[0]["opp_replies"].sum(dtype=torch.int64)
            elif variant == "full":
                rep = tr[0]["opp_replies"]
                acc[0] += rep.sum(dtype=torch.int64)
                acc[1] += rep.view(32, -1, 16).amax(dim=2).sum(dtype=torch.int64) * 16
    e1.record(s)
    ro.synchronize()
    print(variant, "ms per window %.3f" % (e0.elapsed_time(e1) / 20))
