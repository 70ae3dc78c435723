#!/usr/bin/env python3
"""One-off soak (not a test): the one-launch-per-window rollout kernel against the two-launches-per-move path over many windows
(dozens of MT19937 regenerations and hundreds of episodes per game): final records, RNG positions, counters and the last window's
trajectory must be identical.   python tools/soak_rollout.py [windows] [games]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout  # noqa: E402

import hashlib  # noqa: E402
import os  # noqa: E402

from azul_deep_reinforcement_learning_amd import _lib as L  # noqa: E402

windows = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
# provenance of a raw log: which library, which harness, which network (a claim of bit-identity is only as good as the log it cites)
from provenance import csrc_hash  # noqa: E402  (tools/provenance.py: sha256 over csrc/ + include/azul_hip.h)
print("csrc sha256 %s | libazulhip.so sha256 %s | %s | soak_rollout.py sha256 %s" % (
    csrc_hash(), hashlib.sha256(open(L.LIB_PATH, "rb").read()).hexdigest()[:16], L.lib.azul_version().decode(),
    hashlib.sha256(open(__file__, "rb").read()).hexdigest()[:16]))
modes = ("net",) if (len(sys.argv) > 3 and sys.argv[3] == "net") else (None, "random")      # "net": GameRunner(opponent=Agent) -- a second network
for opponent in modes:
    runs = []
    for persistent in (False, True):
        torch.manual_seed(11)               # INSIDE the arm loop: both arms must start from the same network (see LABNOTES.md 6)
        net = BatchedActorCritic(136, 180, 180)
        opp = BatchedActorCritic(136, 180, 180) if opponent == "net" else opponent
        print("  arm persistent=%s: net checksum %.9f" % (persistent, float(sum(p.double().sum() for p in net.parameters()))))
        ro = PolicyRollout(net, n_games=n, parts=1, seed_base=90210, window=32, use_graph=False,
                           opponent=opp, persistent=persistent)
        for _ in range(windows):
            tr = ro.run_window()
        ro.synchronize()
        last = {k: v.clone() for k, v in tr[0].items()}
        mt, pos = ro.envs[0].get_rng_range()
        runs.append((last, ro.envs[0].get_records(), mt, pos, ro.counters()))
    (la, ra, ma, pa, ca), (lb, rb, mb, pb, cb) = runs
    bad = [k for k in la if k not in ("opp_action", "opp_logp") and not torch.equal(la[k], lb[k])]
    ok = not bad and ra.tobytes() == rb.tobytes() and np.array_equal(ma, mb) and np.array_equal(pa, pb) and ca == cb
    print("opponent=%s: %d games x %d windows x 32 steps, episodes per-move arm %d / one-launch arm %d, %d stuck: %s %s" % (
        opponent, n, windows, ca["episodes"], cb["episodes"], ca["stuck"], "IDENTICAL" if ok else "MISMATCH", bad), flush=True)
    assert ok
