#!/usr/bin/env python3
"""Development helper (GPU box): what makes the headline kernel's launch time step up ~0.45 s into sustained load?
Runs the sustained phase in variants -- all outputs / no outputs, 512 / 256 moves per launch -- and prints block means of the launch time,
the on-device clock probe and the driver's clocks (GFX / MEM / DF / SOC via amdsmi)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from azul_deep_reinforcement_learning_amd import BatchedAzul  # noqa: E402

for pth in ("/opt/rocm/share/amd_smi",):
    sys.path.append(pth)
import amdsmi  # noqa: E402

amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]
CLK = {k: getattr(amdsmi.AmdSmiClkType, k) for k in ("GFX", "MEM", "DF", "SOC") if hasattr(amdsmi.AmdSmiClkType, k)}


def clocks():
    out = {}
    for k, t in CLK.items():
        try:
            out[k] = amdsmi.amdsmi_get_clock_info(h, t).get("clk")
        except Exception as e:
            out[k] = "err"
    try:
        out["W"] = amdsmi.amdsmi_get_power_info(h).get("current_socket_power")
    except Exception:
        pass
    return out


def phase(name, G, T, launches, outputs, idle_before=0.0):
    env = BatchedAzul(G)
    env.seed(0)
    env.runner_init()
    env.runner_init()
    b = env.alloc_trajectory(T, packed_mask=True, mask_pitch=192, mask_bits=False) if outputs else None
    run = (lambda: env.selfplay(T, b["mask"], b["action"], b["reward"], b["done"], packed=b["packed"])) if outputs else (lambda: env.selfplay(T))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    time.sleep(idle_before)
    NB = 10
    probes = torch.zeros(NB + 1, 3, dtype=torch.int64, device="cuda")
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((time.perf_counter(), clocks()))
            stop.wait(0.03)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    env.timing_begin()
    env.clock_probe(probes[0])
    marks = [time.perf_counter()]
    for i in range(NB):
        for _ in range(launches // NB):
            run()
        env.clock_probe(probes[i + 1])
        torch.cuda.synchronize()
        marks.append(time.perf_counter())
    env.timing_end()
    stop.set()
    th.join()
    series = env.timing_launch_ms()
    nb = len(series) // NB
    pr = probes.cpu().tolist()
    print("== %s: %d games, %d moves per launch, %d launches, outputs=%s" % (name, G, T, launches, outputs))
    for i in range(NB):
        blk = series[i * nb:(i + 1) * nb]
        cl = [c for t, c in samples if marks[i] <= t <= marks[i + 1]]
        print("  block %d: t=%.2fs launch %.4f ms  probe %.0f MHz  %s" % (i, marks[i + 1] - marks[0], sum(blk) / max(len(blk), 1), 100.0 * pr[i + 1][0] / max(pr[i + 1][1], 1),
                                                                          cl[len(cl) // 2] if cl else None))
    del env, b
    torch.cuda.empty_cache()


phase("all outputs, T=512", 4096, 512, 1000, True)
time.sleep(2.0)
phase("no outputs, T=512", 4096, 512, 1000, False)
time.sleep(2.0)
phase("all outputs, T=256 (twice the launches)", 4096, 256, 2000, True)
time.sleep(2.0)
phase("all outputs, T=512, second time right away", 4096, 512, 1000, True)
phase("all outputs, T=512, third time with no pause", 4096, 512, 1000, True)
