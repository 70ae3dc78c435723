"""End-to-end sanity of training against a frozen past self (not a test): BatchedTrainer(opponent="self", opponent_refresh=50) -- the reference's
GameRunner(opponent=Agent(...)) with the opponent's weights replaced by the policy's every 50 updates -- and, every 500 updates, the policy's
strength measured OUTSIDE training against the RandomAgent (2 windows of 4096 games).  Usage: tools/learn_check_self.py [updates]"""
import sys
import time

sys.path.insert(0, '.')
import torch
from azul_deep_reinforcement_learning_amd import BatchedActorCritic, BatchedTrainer, PolicyRollout

torch.manual_seed(0)
net = BatchedActorCritic(136, 180, 180)
tr = BatchedTrainer(net, n_games=4096, window=32, results_dir="gpurun_out/learn_self", opponent="self", opponent_refresh=50, move_limit=400)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000


def versus_random():
    ro = PolicyRollout(tr.rollout.policy, n_games=4096, window=32, persistent=True, opponent="random", seed_base=10 ** 6)
    for _ in range(6):
        ro.run_window()
    ro.synchronize()
    c = [e.counters() for e in ro.envs]
    ep = sum(int(x["episodes"].sum()) for x in c)
    s = sum(x["stat_sums"].sum(axis=0) for x in c)
    return s[9] / ep, s[0] / ep, s[1] / ep


t0 = time.time()
for i in range(N):
    row = tr.run_batch(collect_stats=(i % 500 == 0 or i == N - 1))
    if row is not None:
        w, p, o = versus_random()
        print("update %5d  vs its past self: win %.3f  score %.1f : %.1f  |  vs RandomAgent: win %.3f  score %.1f : %.1f  (%.1f s)" % (
            row["batch"], row["win_percent"], row["player_score"], row["opponent_score"], w, p, o, time.time() - t0), flush=True)
