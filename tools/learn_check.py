"""End-to-end sanity of rows N1 + N2 (not a test): 3000 batched A2C updates against the RandomAgent opponent on one MI355X.
Round 1: win rate 6 % -> 99 %, mean score 0.8 -> 43.7 (opponent 10.8 -> 6.1) in 5.4 s (profiles/round1_learning_curve.txt)."""
import sys, time
sys.path.insert(0, '.')
import torch
from azul_deep_reinforcement_learning_amd import BatchedActorCritic, BatchedTrainer
torch.manual_seed(0)
tr = BatchedTrainer(BatchedActorCritic(136, 180, 180), n_games=4096, window=32, results_dir="gpurun_out/learn")
t0 = time.time()
N_UPDATES = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for i in range(N_UPDATES):
    row = tr.run_batch()
    if i % max(250, N_UPDATES // 12) == 0 or i == N_UPDATES - 1:
        print("batch %5d  win %.3f  player %.1f  opponent %.1f  rounds %.2f  reward %.2f  critic_loss %.1f  entropy %.3f  (%.1f s)" % (
            row["batch"], row["win_percent"], row["player_score"], row["opponent_score"], row["rounds"], row["reward"], row["critic_loss"], row["entropy_loss"], time.time() - t0), flush=True)
