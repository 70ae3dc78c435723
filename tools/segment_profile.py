#!/usr/bin/env python3
"""Diagnostic: where a self-play move spends its cycles (in-kernel s_memtime stamps).
Builds a SEPARATE library with -DAZ_PROFILE_SEGMENTS, loads it in place of the shipped one for this process only,
runs the bench workload and prints the per-segment shares.  Never quote this build's run time."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

lib = os.path.join(ROOT, "gpurun_out", "libazulhip_prof.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + ge.HIPCC_FLAGS + ["-DAZ_PROFILE_SEGMENTS", "-I", os.path.join(ROOT, "include"),
                      "-o", lib, os.path.join(ge.CSRC, "azul_kernels.hip")], cwd=ge.CSRC)
import azul_deep_reinforcement_learning_amd._lib as L  # noqa: E402
L.LIB_PATH = lib
L.lib = L._load()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedAzul  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ext = int(sys.argv[2]) if len(sys.argv) > 2 else 0
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if P == 2 and ext == 0:
    env = BatchedAzul(4096)
    env.seed(0); env.runner_init(); env.runner_init()
    t = env.alloc_trajectory(256, packed_mask=True)
    run = lambda: env.selfplay(256, t["mask"], t["action"], t["reward"], t["done"], maskbits=t["maskbits"], packed=t["packed"])
else:
    env = BatchedAzul(4096, players=P, ext_rules=ext)
    env.seed(0); env.init(); env.new_round()
    t = env.alloc_trajectory(256, packed_mask=True, mask_pitch={5: 192, 7: 256, 9: 320}[env.displays], mask_bits=False)
    run = lambda: env.selfplay(256, t["mask"], t["action"], t["reward"], t["done"], packed=t["packed"])
cyc = np.zeros(9, dtype=np.uint64)
for _ in range(warm):                                   # games start in lockstep: let their rounds drift apart before measuring
    run()
torch.cuda.synchronize()
L.check(L.lib.azul_batch_segment_profile(env._h, cyc.ctypes.data_as(C.c_void_p), 9, 1))
for _ in range(8):
    run()
torch.cuda.synchronize()
L.check(L.lib.azul_batch_segment_profile(env._h, cyc.ctypes.data_as(C.c_void_p), 9, 1))
names = ["mask+mask out", "sample (RandomAgent)", "do_move", "after move (what-if / next player)", "tail (reward, outputs)",
         "new_round", "count_score", "reset (ctor)", "loop overhead"]
tot = float(cyc.sum())
moves = 4096 * 256 * 8
print("players %d, rule flags %d, %d warm-up launches" % (P, ext, warm))
for n, c in zip(names, cyc):
    print("%-38s %6.2f %%   %8.1f cycles/move" % (n, 100.0 * float(c) / tot, float(c) / moves))
print("total %.0f cycles/move (diagnostic build)" % (tot / moves))
