#!/usr/bin/env python3
"""Diagnostic: per-move phase times inside azul_policy_rollout_kernel (s_memtime stamps of wave 5 of every workgroup,
-DAZ_PROFILE_SEGMENTS build loaded for this process only).  Never quote this build's run time."""
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
from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout  # noqa: E402

opp = "random" if "--opponent" in sys.argv else None
torch.manual_seed(0)
ro = PolicyRollout(BatchedActorCritic(136, 180, 180), n_games=4096, parts=1, window=32, opponent=opp, persistent=True)
for _ in range(3):
    ro.run_window()
ro.synchronize()
cyc = np.zeros(9, dtype=np.uint64)
L.check(L.lib.azul_batch_segment_profile(ro.envs[0]._h, cyc.ctypes.data_as(C.c_void_p), 9, 1))
windows = 10
for _ in range(windows):
    ro.run_window()
ro.synchronize()
L.check(L.lib.azul_batch_segment_profile(ro.envs[0]._h, cyc.ctypes.data_as(C.c_void_p), 9, 1))
names = ["env step + publish (own game)", "wait for the slowest env wave", "layer 1 (+ barrier)", "layer 2 / critic (+ barrier)", "head (+ barrier)"]
moves = 256 * 32 * windows
tot = 0.0
for nm, cv in zip(names, cyc):
    print("%-34s %8.0f cycles per move" % (nm, float(cv) / moves))
    tot += float(cv) / moves
print("sum %.0f cycles = %.2f us per move at 2.35 GHz (wave 5 of each workgroup, opponent=%s)" % (tot, tot / 2350.0, opp))
