#!/usr/bin/env python3
"""Development helper (GPU box): A/B libazulhip.so builds on the headline kernel THROUGH THE RAW C ABI (so that a build of an older round,
which lacks newer entry points, can be compared): tools/ab_raw.py a.so b.so ...   -> ms per 512-move launch at 4096 games, event-timed."""
import ctypes as C
import sys

import torch

G, T, REPS = 4096, 512, 3
LATE = "--late" in sys.argv        # also: the launch time after 700 launches (seed base 0: game 801 can no longer end -- hazard H9 -- and every one of its moves is a floor move)


def run(path):
    L = C.CDLL(path)
    vp, i, u64 = C.c_void_p, C.c_int, C.c_uint64
    L.azul_batch_create.argtypes = [C.POINTER(vp), i, i, i]
    L.azul_batch_seed.argtypes = [vp, u64, vp, vp]
    L.azul_batch_runner_init.argtypes = [vp, vp, vp, vp]
    L.azul_batch_selfplay_strided.argtypes = [vp, i, vp, i, vp, vp, vp, vp, vp, vp, vp]
    L.azul_batch_destroy.argtypes = [vp]
    h = vp()
    assert L.azul_batch_create(C.byref(h), G, 0, 1) == 0
    L.azul_batch_seed(h, 0, None, None)
    L.azul_batch_runner_init(h, None, None, None)
    L.azul_batch_runner_init(h, None, None, None)
    mask = torch.zeros(T, G, 192, dtype=torch.uint8, device="cuda")
    act = torch.zeros(T, G, dtype=torch.int32, device="cuda")
    rew = torch.zeros_like(act)
    pk = torch.zeros_like(act)
    dn = torch.zeros(T, G, dtype=torch.uint8, device="cuda")
    p = lambda t: vp(t.data_ptr())
    st = vp(torch.cuda.current_stream().cuda_stream)
    launch = lambda: L.azul_batch_selfplay_strided(h, T, p(mask), 192, None, p(act), p(rew), p(dn), p(pk), None, st)
    out = []
    for _ in range(REPS):
        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        ev[0].record()
        for k in range(20):
            launch()
            ev[k + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(20))
        out.append("median %.4f min %.4f" % (ts[10], ts[0]))
    if LATE:
        done = 25 * REPS
        for mark in (600, 700, 800, 1000, 1500):
            for _ in range(mark - done):
                launch()
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            ev[0].record()
            for k in range(20):
                launch()
                ev[k + 1].record()
            torch.cuda.synchronize()
            done = mark + 20
            ts = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(20))
            out.append("after %d launches: median %.4f min %.4f max %.4f" % (mark, ts[10], ts[0], ts[-1]))
    L.azul_batch_destroy(h)
    return out


for rep in range(2):
    for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
        print(path, run(path), flush=True)
