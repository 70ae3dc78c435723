#!/usr/bin/env python3
"""Scratch A/B: event-bracketed time of one rollout launch for the shipped library and for builds with extra -D flags.
   python tools/rollout_ab.py [--opponent] [--window W] -- <flagset1> <flagset2> ...   (flagset: comma-separated -D names, or 'ship')"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import azul_deep_reinforcement_learning_amd._lib as L  # noqa: E402
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout  # noqa: E402

opp = "random" if "--opponent" in sys.argv else None
WIN = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else 32
sets = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else ["ship"]
for fs in sets:
    if fs.endswith(".so"):                                # a prebuilt library (e.g. the previous commit's)
        L.LIB_PATH = os.path.abspath(fs)
        L.lib = L._load()
    elif fs != "ship":
        lib = os.path.join(ROOT, "gpurun_out", "libazulhip_ab_%s.so" % fs.replace(",", "_"))
        if not os.path.exists(lib):
            os.makedirs(os.path.dirname(lib), exist_ok=True)
            subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + ge.HIPCC_FLAGS + ["-D" + f for f in fs.split(",")] +
                                  ["-I", os.path.join(ROOT, "include"), "-o", lib, os.path.join(ge.CSRC, "azul_kernels.hip")], cwd=ge.CSRC)
        L.LIB_PATH = lib
        L.lib = L._load()
    torch.manual_seed(0)
    ro = PolicyRollout(BatchedActorCritic(136, 180, 180), n_games=4096, parts=1, window=WIN, opponent=opp, persistent=True)
    for _ in range(3):
        ro.run_window()
    ro.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    with torch.cuda.stream(ro.streams[0]):
        e0.record()
    for _ in range(reps):
        ro.run_window()
    with torch.cuda.stream(ro.streams[0]):
        e1.record()
    ro.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print("%-40s window %d: %.1f us per launch (+ returns scan) = %.2f us per move" % (fs, WIN, us, us / WIN))
    del ro
