#!/usr/bin/env python3
"""Development helper (GPU box): which game makes the launches slower after ~310 k moves (seed base 0)?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedAzul  # noqa: E402

G, T = 4096, 512
env = BatchedAzul(G)
env.seed(0)
env.runner_init()
env.runner_init()
b = env.alloc_trajectory(T, packed_mask=True, mask_pitch=192, mask_bits=False)
run = lambda: env.selfplay(T, b["mask"], b["action"], b["reward"], b["done"], packed=b["packed"])
for i in range(700):
    run()
torch.cuda.synchronize()
c0 = {k: np.array(v).copy() for k, v in env.counters().items()}
run()
torch.cuda.synchronize()
c1 = env.counters()
de = (np.array(c1["episodes"]).astype(np.int64) - c0["episodes"].astype(np.int64))
ds = (np.array(c1["stuck"]).astype(np.int64) - c0["stuck"].astype(np.int64))
print("episodes per launch: min %d median %d max %d; stuck per launch: max %d (games %s)" % (de.min(), np.median(de), de.max(), ds.max(), np.flatnonzero(ds > 0)[:10]))
odd = np.flatnonzero((de > np.median(de) * 2) | (de == 0) | (ds > 0))
print("odd games:", odd[:20], de[odd[:20]], ds[odd[:20]])
act = b["action"].cpu().numpy()
done = b["done"].cpu().numpy()
msk = b["mask"].cpu().numpy()[:, :, :180]
legal = msk.sum(axis=2)
print("legal actions per decision: mean %.1f; per-game mean min %.1f (game %d) max %.1f (game %d)" % (legal.mean(), legal.mean(0).min(), legal.mean(0).argmin(), legal.mean(0).max(), legal.mean(0).argmax()))
for g in list(odd[:4]):
    print("game", g, "actions", act[:24, g], "done", done[:24, g], "legal", legal[:24, g])
    rec = env.get_records()[g]
    print({k: rec[k].tolist() for k in rec.dtype.names})
# time the batch with the odd games' records replaced by a neighbour's
from azul_deep_reinforcement_learning_amd import _lib as L
def launch_ms(n=20):
    env.timing_begin()
    for _ in range(n):
        run()
    _, _, kms, kn = env.timing_end()
    return kms / kn
print("launch ms now: %.4f" % launch_ms())
if len(odd):
    view = env.records_dev()      # on the device: set_records would switch the batch to the self-play instantiation that marks rule-error-stopped games
    for g in odd:
        view[int(g)] = view[int((g + 1) % G if (g + 1) % G not in odd else (g + 7) % G)]
    torch.cuda.synchronize()
    print("launch ms with the odd games replaced: %.4f" % launch_ms())
