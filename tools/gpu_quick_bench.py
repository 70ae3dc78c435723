"""Scratch timing used while developing (not a test): selfplay kernel at N=4096."""
import sys, time
import torch
sys.path.insert(0, ".")
from azul_deep_reinforcement_learning_amd import BatchedAzul

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
env = BatchedAzul(n)
env.seed(0); env.runner_init(); env.runner_init()
t = env.alloc_trajectory(T, packed_mask=True)
for _ in range(3):
    env.selfplay(T, t["mask"], t["action"], t["reward"], t["done"], maskbits=t["maskbits"], packed=t["packed"])
torch.cuda.synchronize()
for outputs in (True, False):
    env.timing_begin()
    t0 = time.time()
    reps = 20
    for _ in range(reps):
        if outputs:
            env.selfplay(T, t["mask"], t["action"], t["reward"], t["done"], maskbits=t["maskbits"], packed=t["packed"])
        else:
            env.selfplay(T)
    ms, launches, kms, kn = env.timing_end()
    ms, launches = (kms, kn) if kn else (ms, launches)     # per-launch event pairs: the kernel's own duration
    dt = time.time() - t0
    print("N=%d T=%d outputs=%s: %.3f ms/launch, %.1f us/step, %.1f M env-steps/s (wall %.1f M/s)" % (
        n, T, outputs, ms / launches, ms / launches / T * 1e3, n * T * launches / ms / 1e3, n * T * reps / dt / 1e6))
