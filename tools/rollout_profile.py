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
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + ge.HIPCC_FLAGS + ["-DAZ_PROFILE_SEGMENTS"] + [a for a in sys.argv[1:] if a.startswith("-D")] + ["-I", os.path.join(ROOT, "include"),
                      "-o", lib, os.path.join(ge.CSRC, "azul_kernels.hip")], cwd=ge.CSRC)
import azul_deep_reinforcement_learning_amd._lib as L  # noqa: E402
L.LIB_PATH = lib
L.lib = L._load()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout  # noqa: E402

opp = "random" if "--opponent" in sys.argv else None
if "--net" in sys.argv:                                   # GameRunner(opponent=Agent(...)): the reply rounds' phases are added to the agent pass's
    opp = BatchedActorCritic(136, 180, 180)
WIN = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else 32
torch.manual_seed(0)
ro = PolicyRollout(BatchedActorCritic(136, 180, 180), n_games=4096, parts=1, window=WIN, opponent=opp, persistent=True)
for _ in range(3):
    ro.run_window()
ro.synchronize()
cyc = np.zeros(32, dtype=np.uint64)
L.check(L.lib.azul_batch_segment_profile(ro.envs[0]._h, cyc.ctypes.data_as(C.c_void_p), 32, 1))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(ro.streams[0]):
    e0.record()
ro.run_window()                                            # one launch on its own: first workgroup start .. last workgroup end
with torch.cuda.stream(ro.streams[0]):
    e1.record()
ro.synchronize()
print("event bracket around that launch (+ the returns scan): %.1f us" % (e0.elapsed_time(e1) * 1e3))
L.check(L.lib.azul_batch_segment_profile(ro.envs[0]._h, cyc.ctypes.data_as(C.c_void_p), 32, 1))
span = (float(cyc[8]) - ((1 << 62) - float(cyc[5]))) / 100.0
print("one launch: loops of the 256 workgroups span %.1f us (earliest start .. latest end); mean loop %.1f us" % (span, float(cyc[7]) / 256 / 100.0))
windows = 10
for _ in range(windows):
    ro.run_window()
ro.synchronize()
L.check(L.lib.azul_batch_segment_profile(ro.envs[0]._h, cyc.ctypes.data_as(C.c_void_p), 32, 1))
names = ["env step + publish (own game)", "wait for the slowest env wave", "layer 1 (+ barrier)", "layer 2 / critic (+ barrier)", "head (+ barrier)"]
moves = 256 * WIN * windows
tot = 0.0
for nm, cv in zip(names, cyc):
    print("%-34s %8.0f cycles per move" % (nm, float(cv) / moves))
    tot += float(cv) / moves
print("wave 5's loop per launch: %.0f shader cycles, %.1f us of s_memrealtime (100 MHz)" % (float(cyc[6]) / (256 * windows), float(cyc[7]) / (256 * windows) / 100.0))
ghz = float(cyc[6]) / max(float(cyc[7]), 1.0) * 0.1
print("sum %.0f cycles per move (wave 5 of each workgroup, opponent=%s); in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz = %.2f GHz"
      " -> %.2f us per move" % (tot, opp, ghz, tot / ghz / 1e3))

# matrix sub-phases of the two waves that share SIMD 1 (agent pass only): where the layer phases' cycles go
tiles = {1: (4, 2), 5: (2, 1)}
for wv, base in ((1, 16), (5, 24)):
    v = [float(x) / moves for x in cyc[base:base + 8]]
    print("wave %d  layer 1: prologue %5.0f | MFMA loop issue span %5.0f (own MFMAs: %d tiles x 34 k-steps x 32 = %d) | epilogue %5.0f | barrier wait %5.0f   = %5.0f"
          % (wv, v[0], v[1], tiles[wv][0], tiles[wv][0] * 34 * 32, v[2], v[3], sum(v[:4])))
    print("        layer 2: prologue %5.0f | MFMA loop issue span %5.0f (own MFMAs: %d tiles x 45 k-steps x 32 = %d) | epilogue %5.0f | barrier wait %5.0f   = %5.0f"
          % (v[4], v[5], tiles[wv][1], tiles[wv][1] * 45 * 32, v[6], v[7], sum(v[4:])))
print("matrix pipe of a SIMD per move: layer 1 = 6 tiles x 34 x 32 = 6528 cycles, layer 2 = 3 x 45 x 32 = 4320 cycles")
