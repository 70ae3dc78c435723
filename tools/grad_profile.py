#!/usr/bin/env python3
"""Diagnostic: where azul_a2c_grad_kernel spends a sample tile (s_memtime ticks of wave 0 per phase, -DAZ_LG_PROFILE build
loaded for this process only) + the shipped kernel's launch time on the same samples.  Never quote the diagnostic build's
run time.   python tools/grad_profile.py [n_samples]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import azul_deep_reinforcement_learning_amd._lib as L  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedActorCritic  # noqa: E402
from azul_deep_reinforcement_learning_amd.learner import A2CLearner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 131072
if "--lib" in sys.argv:                                   # A/B runs of experimental builds: timing only
    L.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
    L.lib = L._load()
torch.manual_seed(0)
rs = np.random.RandomState(0)
obs = torch.from_numpy(rs.randint(0, 6, size=(n, 136)).astype(np.float32)).cuda()
m = rs.rand(n, 180) < 0.2
act_np = rs.randint(0, 180, n)
m[np.arange(n), act_np] = True
mask = torch.from_numpy(m.astype(np.uint8)).cuda()
act = torch.from_numpy(act_np.astype(np.int32)).cuda()
q = torch.from_numpy(rs.randn(n).astype(np.float32) * 5).cuda()


def run(reps):
    torch.manual_seed(0)
    net = BatchedActorCritic(136, 180, 180).cuda()
    lr = A2CLearner(net, distributed=False, fused=True)
    lr._fused_gradients(obs, mask, act, q, n_total=float(n))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lr._fused_gradients(obs, mask, act, q, n_total=float(n))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, lr._ws["grad"].clone()


ms, g_ship = run(20)
flop = 391e3 * n
print("shipped kernel: %.1f us per launch of %d samples (gradients + reduction) = %.1f TFLOP/s = %.1f %% of 157.3" % (
    ms * 1e3, n, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100))

if "--lib" in sys.argv:
    sys.exit(0)
lib = os.path.join(ROOT, "gpurun_out", "libazulhip_lgprof.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + ge.HIPCC_FLAGS + ["-DAZ_LG_PROFILE", "-I", os.path.join(ROOT, "include"),
                      "-o", lib, os.path.join(ge.CSRC, "azul_kernels.hip")], cwd=ge.CSRC)
L.LIB_PATH = lib
L.lib = L._load()
cyc = np.zeros(8, dtype=np.uint64)
run(2)
L.lib.azul_debug_lg_profile.restype = C.c_int
L.lib.azul_debug_lg_profile(cyc.ctypes.data_as(C.c_void_p), 8, 1)
reps = 5
ms_d, g_diag = run(reps)
L.lib.azul_debug_lg_profile(cyc.ctypes.data_as(C.c_void_p), 8, 1)
assert torch.equal(g_ship, g_diag), "diagnostic build computes something else"
tiles = (n + 31) // 32 * (reps + 1)
names = ["P0 observations -> LDS", "P1 layer 1", "P2 layer 2 + critic", "P3 softmax / loss / dlogits", "P4ab dW2 (+ bias sums)", "P4c dz", "P5 dW1"]
tot = 0.0
for nm, cv in zip(names, cyc):
    tot += float(cv) / tiles
for nm, cv in zip(names, cyc):
    print("%-30s %8.0f ticks per tile  %5.1f %%" % (nm, float(cv) / tiles, 100.0 * float(cv) / tiles / tot))
print("sum %.0f s_memtime ticks per 32-sample pass; diagnostic build %.1f us per launch" % (tot, ms_d * 1e3))
