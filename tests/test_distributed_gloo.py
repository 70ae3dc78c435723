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


def _bufs_from_oracle(rank, chunk):
    from azul_deep_reinforcement_learning_amd.parallel import shard_seed_base
    from oracle import oracle as oz
    base = shard_seed_base(100, G, rank)
    maskbits = torch.zeros(T, G, 3, dtype=torch.int64)
    packed = torch.zeros(T, G, dtype=torch.int32)
    for g in range(G):
        s = oz.Stream(base + g)
        s.advance(chunk * T, want_records=False)
        out = s.advance(T, want_records=False)
        bits = np.packbits(np.pad(out["mask"], ((0, 0), (0, 12))), axis=1, bitorder="little").view(np.int64)
        maskbits[:, g, :] = torch.from_numpy(bits.copy())
        a = np.where(out["action"] < 0, 0xFF, out["action"]).astype(np.uint32) & 0xFF
        p = a | (out["done"].astype(np.uint32) << 8) | ((out["reward"].astype(np.int64) & 0xFFFF).astype(np.uint32) << 16)
        packed[:, g] = torch.from_numpy(p.astype(np.uint32).view(np.int32))
    return {"maskbits": maskbits, "packed": packed,
            "action": torch.zeros(0), "reward": torch.zeros(0), "done": torch.zeros(0)}, out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from azul_deep_reinforcement_learning_amd.parallel import TrajectoryGather
    tg = TrajectoryGather(world, torch.device("cpu"), with_masks=True)
    for chunk in range(3):                       # double-buffered like bench.py
        slot = chunk & 1
        tg.wait_buffer_free(slot)
        b, _ = _bufs_from_oracle(rank, chunk)
        tg.launch(slot, b, T)
    tg.finish()
    got = tg.gathered(0, T, G)                   # slot 0 holds chunk 2
    q.put((rank, {k: v.numpy().copy() for k, v in got.items()}))
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
    from azul_deep_reinforcement_learning_amd.parallel import unpack_moves
    for key in ("maskbits", "packed"):
        assert np.array_equal(results[0][key], results[1][key])
    for r in range(world):
        exp, _ = _bufs_from_oracle(r, 2)
        for key in ("maskbits", "packed"):
            assert np.array_equal(results[0][key][r], exp[key].numpy()), (r, key)
    # the compact record decodes back to (action, done, reward); invariance to the GPU count: global game 8+3
    # (rank 1, local 3) is the same stream as game 11 of a 1-rank run
    from oracle import oracle as oz
    s1 = oz.Stream(100 + 8 + 3)
    s1.advance(2 * T, want_records=False)
    ref = s1.advance(T, want_records=False)
    action, done, reward = unpack_moves(torch.from_numpy(results[0]["packed"][1][:, 3].copy()))
    assert np.array_equal(action.numpy(), ref["action"]) and np.array_equal(done.numpy(), ref["done"])
    assert np.array_equal(reward.numpy(), ref["reward"])


# ---- the opt-in all-gather of full C1 records (bench.py --gather-c1; BASELINE configs[4] names it; nn_runner.py:59-78) -------------------
CT, CG = 6, 5      # agent steps per window, games per rank


def _c1_window(rank, window):
    """A deterministic stand-in for one rank's PolicyRollout window (CPU tensors): env side from the oracle's streams of that rank's
    GLOBAL game ids, network side (value / log-prob / entropy / returns) from a generator keyed by (rank, window)."""
    from azul_deep_reinforcement_learning_amd.parallel import shard_seed_base
    from oracle import oracle as oz
    base = shard_seed_base(500, CG, rank)
    g = torch.Generator().manual_seed(1000 * rank + window)
    tr = {"obs": torch.zeros(CT + 1, CG, 136), "mask": torch.zeros(CT + 1, CG, 180, dtype=torch.uint8), "player": torch.zeros(CT + 1, CG, dtype=torch.uint8),
          "action": torch.zeros(CT, CG, dtype=torch.int32), "reward": torch.zeros(CT, CG, dtype=torch.int32), "done": torch.zeros(CT, CG, dtype=torch.uint8),
          "value": torch.randn(CT, CG, 1, generator=g), "log_prob": torch.randn(CT, CG, generator=g), "entropy": torch.randn(CT, CG, generator=g),
          "returns": torch.randn(CT, CG, generator=g)}
    for j in range(CG):
        s = oz.Stream(base + j)
        s.advance(window * CT, want_records=False)
        out = s.advance(CT, want_records=True)
        tr["mask"][:CT, j] = torch.from_numpy(out["mask"])
        tr["action"][:, j] = torch.from_numpy(out["action"])
        tr["reward"][:, j] = torch.from_numpy(out["reward"])
        tr["done"][:, j] = torch.from_numpy(out["done"])
        rec = np.ascontiguousarray(out["rec_after"]).view(np.uint8).reshape(CT, 128)
        tr["obs"][:CT, j, :128] = torch.from_numpy(rec.astype(np.float32))      # any integers in 0..255 do for the wire format
        tr["player"][:CT, j] = torch.from_numpy((rec[:, 31] & 7).astype(np.uint8))
    return tr


def _c1_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from azul_deep_reinforcement_learning_amd.parallel import C1_BYTES, TrajectoryGather, pack_c1
    tg = TrajectoryGather(world, torch.device("cpu"))
    bufs = [torch.empty(CT, CG, C1_BYTES, dtype=torch.uint8) for _ in range(2)]
    for window in range(3):                      # double-buffered like bench.py's extras
        slot = window & 1
        tg.wait_buffer_free(slot)
        tg.launch_c1(slot, pack_c1(_c1_window(rank, window), CT, out=bufs[slot]))
    tg.finish()
    q.put((rank, tg.gathered_c1(0, CT, CG).numpy().copy(), tg.gathered_bytes))      # slot 0 holds window 2
    dist.barrier()
    dist.destroy_process_group()


def test_c1_record_allgather_two_ranks():
    from azul_deep_reinforcement_learning_amd.parallel import C1_BYTES, pack_c1, unpack_c1
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_c1_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (a, b) for r, a, b in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][0], res[1][0])                        # every rank holds the same gathered records
    assert res[0][1] == 3 * world * CT * CG * C1_BYTES                  # bytes delivered into a rank: three windows of `world` slices
    for r in range(world):
        tr = _c1_window(r, 2)
        assert np.array_equal(res[0][0][r], pack_c1(tr, CT).numpy()), r  # slice r == what a single process records for rank r's games
        u = unpack_c1(torch.from_numpy(res[0][0][r].copy()))
        for k in ("action", "reward", "done", "log_prob", "entropy", "returns"):
            assert torch.equal(u[k], tr[k]), k
        assert torch.equal(u["value"], tr["value"].reshape(CT, CG))
        assert torch.equal(u["obs"], tr["obs"][:CT]) and torch.equal(u["mask"], tr["mask"][:CT]) and torch.equal(u["player"], tr["player"][:CT])
