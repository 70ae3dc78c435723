#!/usr/bin/env python3
"""Development helper (GPU box, under rocprofv3 --kernel-trace --stats): the timed loop of bench.py --gather-c1 on one GPU -- a window of the
policy rollout kernel, then the window's C1 records packed for the trajectory all-gather -- to show which kernels produce the bytes that
would be shipped: azul_policy_rollout2_kernel and azul_pack_c1_kernel, no at::native kernel.  (The all-gather itself needs N > 1.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout  # noqa: E402
from azul_deep_reinforcement_learning_amd.parallel import C1_BYTES, pack_c1  # noqa: E402

games, window, windows = 4096, 32, 40
torch.manual_seed(0)
ro = PolicyRollout(BatchedActorCritic(136, 180, 180), n_games=games, parts=1, window=window, persistent=True)
bufs = [torch.empty(window, games, C1_BYTES, dtype=torch.uint8, device="cuda") for _ in range(2)]
for _ in range(3):
    ro.run_window()
ro.synchronize()
s = ro.streams[0]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for i in range(windows):
    tr = ro.run_window()
    with torch.cuda.stream(s):
        pack_c1(tr[0], window, out=bufs[i & 1])
e1.record(s)
ro.synchronize()
print("%d windows of %d agent steps x %d games + pack_c1: %.3f ms per window" % (windows, window, games, e0.elapsed_time(e1) / windows))
