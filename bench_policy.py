#!/usr/bin/env python3
"""bench_policy.py -- BASELINE configs[2]: 4096 concurrent games, ActorCritic(136,180,hidden 180) policy on
PyTorch-ROCm interleaved with the env step on separate HIP streams, 1 MI355X.  Not the driver's headline bench
(that is bench.py / configs[1]); prints one JSON line with env steps/s for the policy-driven rollout.

    python bench_policy.py [--games 4096] [--parts 1] [--window 32] [--windows 20] [--no-graph]
    python bench_policy.py --train          # the reference's training setup (NNRunner.train): policy vs RandomAgent opponent,
                                            # one A2C update (learner.py) per window; N>1 under torch.distributed.run
"""
import argparse
import json
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 1)[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--parts", type=int, default=1)
    ap.add_argument("--window", type=int, default=32)
    ap.add_argument("--windows", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--torch-head", action="store_true", help="sample with torch.multinomial instead of the fused head kernel")
    ap.add_argument("--torch-mlp", action="store_true", help="run the network as PyTorch GEMMs instead of azul_policy_forward")
    ap.add_argument("--per-move", action="store_true", help="two launches per move (forward + env step) instead of one launch per window")
    ap.add_argument("--no-compare", action="store_true", help="skip the PyTorch-GEMM / two-stream comparison run")
    ap.add_argument("--train", action="store_true", help="policy vs RandomAgent opponent + one A2C update per window")
    a = ap.parse_args()
    import torch
    from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout
    if a.train:
        return train(a)
    import os
    import torch.distributed as dist
    # N > 1 (python -m torch.distributed.run --nproc-per-node N bench_policy.py): games shard by global id, every rank keeps its own
    # trajectories (they feed that rank's share of the data-parallel update); the value is the aggregate over the ranks
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("AZUL_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
        a.no_compare = True

    def measure(parts, fused_mlp, persistent=False):
        torch.manual_seed(0)
        net = BatchedActorCritic(136, 180, 180)
        ro = PolicyRollout(net, n_games=a.games, parts=parts, window=a.window, use_graph=not a.no_graph, seed_base=rank * a.games,
                           sample_seed=0x5EED, fused_head=not a.torch_head, fused_mlp=fused_mlp, persistent=persistent)
        for _ in range(3):
            ro.run_window()
        ro.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.windows):
            ro.run_window()
        ro.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return ro, dt

    moves = a.games * world * a.window * a.windows
    extra = {}
    if not a.no_compare and not a.torch_mlp:
        # comparison runs go FIRST: measured after the one-launch-per-window kernel has run in the same process, the PyTorch-GEMM
        # configuration came out ~40 % slower than on its own (69 M -> 40 M; cause not established), the others did not.
        # 1) the configuration BASELINE.json words literally: model.py's network as PyTorch-ROCm GEMMs, interleaved with the
        #    env step on two HIP streams (same trajectories recorded, same head kernel)
        ro2, dt2 = measure(2, False)
        extra["pytorch_rocm_policy_two_streams"] = {"value": moves / dt2, "unit": "env steps/s", "hip_graph": ro2.use_graph}
        del ro2
        if not a.per_move:
            ro1, dt1 = measure(a.parts, True, persistent=False)
            extra["fused_forward_two_launches_per_move"] = {"value": moves / dt1, "unit": "env steps/s", "hip_graph": ro1.use_graph}
            del ro1
    ro, dt = measure(a.parts, not a.torch_mlp, persistent=not a.per_move)
    c = ro.counters()
    out = {"metric": "Azul env steps/sec (ActorCritic policy self-play, trajectories recorded)", "value": moves / dt,
           "unit": "env steps/s", "n_gpus": world, "config": {"workload": "BASELINE configs[2]", "games_per_gpu": a.games,
           "stream_parts": a.parts, "moves_per_graph": a.window, "hip_graph": ro.use_graph, "fused_head": ro.fused_head, "fused_mlp": ro.fused_mlp,
           "one_launch_per_window": ro.persistent, "graph_error": ro.graph_error}, "ms_per_step": dt / (a.window * a.windows) * 1e3,
           "episodes_finished": c["episodes"], "stuck": c["stuck"], "dtype": "fp32 policy / u8 env", "data": "synthetic"}
    out.update(extra)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def train(a):
    """Row N2: batched NNRunner.train -- rollout windows against the RandomAgent opponent, A2C update per window,
    gradients all-reduced over the ranks (one process per GPU)."""
    import os
    import torch
    import torch.distributed as dist
    from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout
    from azul_deep_reinforcement_learning_amd.learner import A2CLearner
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("AZUL_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    torch.manual_seed(0)                                    # same initial weights on every rank
    net = BatchedActorCritic(136, 180, 180).cuda()
    learner = A2CLearner(net)
    ro = PolicyRollout(net, n_games=a.games, parts=a.parts, window=a.window, use_graph=not a.no_graph, fused_head=not a.torch_head,
                       fused_mlp=not a.torch_mlp, persistent=not a.per_move, opponent="random", seed_base=rank * a.games,
                       sample_seed=0x5EED, kweights=learner.kweights(), ring=3 if (a.parts == 1 and not a.per_move) else 1)

    def one_window():
        tr = ro.run_window()
        ro.join()                                           # device-side dependency between the streams; no host sync per window
        out = learner.update_from_rollout(ro)
        ro.refresh_weights()
        return out

    for _ in range(3):
        one_window()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    e0 = ro.counters()["episodes"]
    t0 = time.perf_counter()
    for _ in range(a.windows):
        out = one_window()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    c = ro.counters()
    stats = {k: list(v) for k, v in learner.statistics.items()}
    if rank == 0:
        print(json.dumps({"metric": "A2C training throughput (policy vs RandomAgent opponent, one update per window)",
                          "value": a.games * world * a.window * a.windows / dt, "unit": "agent steps/s", "n_gpus": world,
                          "updates_per_s": a.windows / dt, "samples_per_update": float(out["samples"]),
                          "episodes_per_s_rank0": (c["episodes"] - e0) / dt,
                          "config": {"workload": "NNRunner.train batched: %d games/GPU, window %d agent steps" % (a.games, a.window),
                                     "hip_graph": ro.use_graph, "graph_error": ro.graph_error},
                          "first_losses": {k: float(stats[k][0]) for k in ("actor_loss", "critic_loss", "entropy_loss")},
                          "last_losses": {k: float(stats[k][-1]) for k in ("actor_loss", "critic_loss", "entropy_loss")},
                          "dtype": "fp32 policy / u8 env", "data": "synthetic"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
