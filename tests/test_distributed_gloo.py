"""world_size-2 gloo test of the N>1 path: global-id sharding + the trajectory all-gather (CPU tensors).
The per-rank trajectories come from the oracle here (no GPU); on the GPU box the same TrajectoryGather
object ships the kernel's buffers over RCCL (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

G, T = 8, 40       # games per rank, moves per chunk


def _flat_from_oracle(rank, chunk):
    from azul_deep_reinforcement_learning_amd.parallel import shard_seed_base
    from oracle import oracle as oz
    base = shard_seed_base(100, G, rank)
    sizes = [T * G * 24, T * G * 4, T * G * 4, T * G]
    flat = torch.zeros(sum(sizes), dtype=torch.uint8)
    o = np.cumsum([0] + sizes)
    maskbits = flat[o[0]:o[1]].view(torch.int64).view(T, G, 3)
    action = flat[o[1]:o[2]].view(torch.int32).view(T, G)
    reward = flat[o[2]:o[3]].view(torch.int32).view(T, G)
    done = flat[o[3]:o[4]].view(T, G)
    for g in range(G):
        s = oz.Stream(base + g)
        s.advance(chunk * T, want_records=False)
        out = s.advance(T, want_records=False)
        bits = np.packbits(np.pad(out["mask"], ((0, 0), (0, 12))), axis=1, bitorder="little").view(np.int64)
        maskbits[:, g, :] = torch.from_numpy(bits.copy())
        action[:, g] = torch.from_numpy(out["action"])
        reward[:, g] = torch.from_numpy(out["reward"])
        done[:, g] = torch.from_numpy(out["done"])
    return {"flat": flat, "maskbits": maskbits, "action": action, "reward": reward, "done": done}


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from azul_deep_reinforcement_learning_amd.parallel import TrajectoryGather
    tg = TrajectoryGather(world, torch.device("cpu"))
    bufs = []
    for chunk in range(3):                       # double-buffered like bench.py
        slot = chunk & 1
        tg.wait_buffer_free(slot)
        b = _flat_from_oracle(rank, chunk)
        bufs.append(b)
        tg.launch(slot, b, T)
    tg.finish()
    got = tg.gathered(0, T, G)                   # slot 0 holds chunk 2
    q.put((rank, {k: v.numpy() for k, v in got.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_trajectory_allgather_two_ranks():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank sees the same gathered data, and slice r equals what a single process computes for rank r
    for key in ("maskbits", "action", "reward", "done"):
        assert np.array_equal(results[0][key], results[1][key])
    for r in range(world):
        exp = _flat_from_oracle(r, 2)
        for key in ("maskbits", "action", "reward", "done"):
            assert np.array_equal(results[0][key][r], exp[key].numpy()), (r, key)
    # invariance to the GPU count: global game 8+3 (rank 1, local 3) is the same stream as a 1-rank run of 16 games
    from oracle import oracle as oz
    s1 = oz.Stream(100 + 8 + 3)
    s1.advance(2 * T, want_records=False)
    assert np.array_equal(s1.advance(T, want_records=False)["action"], results[0]["action"][1][:, 3])
