#!/usr/bin/env python3
"""bench.py -- Azul env steps/sec, random-agent self-play (BASELINE.json metric), on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--games G] [--chunk T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole batch: every one of the G games per GPU plays one env
move (legal mask -> RandomAgent -> Azul.step -> reward -> done, auto-reset at game end).  Steps are issued
as ceil(K / T) launches of the persistent self-play kernel (T env moves per game per launch, state resident
in registers in between); inputs (game records, MT19937 streams) are resident in HBM before the clock starts.

Workload: BASELINE.json configs[1] -- 4096 concurrent 2-player games per GPU, rules Lid + random first player,
game g of rank r seeded random.seed(base + 4096 r + g).  With N > 1 the games shard by global id (no data-path
collective); the one exchange step of the path -- the all-gather of the trajectory buffers -- runs over RCCL on a
side stream, overlapped with the next launch, and is inside the timed region.

Rank 0 prints ONE JSON line (contract in the task description) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 445          # state R 128 + W 128 + action 4 + mask 180 + reward 4 + done 1
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s HBM3E


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--games", type=int, default=4096, help="games per GPU (BASELINE configs[1]: 4096)")
    ap.add_argument("--chunk", type=int, default=512, help="env moves per game per kernel launch")
    ap.add_argument("--seed-base", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL trajectory all-gather")
    ap.add_argument("--gather-masks", action="store_true", help="N>1: also ship the bit-packed legal masks (24 B/move)")
    return ap.parse_args()


def cpu_baseline(games, seed_base):
    """The oracle (a plain-C port of the reference path) timed on this box's host cores; bounded sample."""
    from oracle import oracle as oz
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = int(os.environ.get("AZUL_CPU_THREADS", min(cores, 16)))     # a 1-GPU box shares 16 host cores
    streams, steps = min(games, 1024), 16000            # ~16 M moves: 15-25 s of CPU work, ~1.5 s wall on 16 threads
    oz.bench_selfplay(seed_base, min(streams, 64), 200, cores)           # warm the pages
    t0 = time.perf_counter()
    moves, _ = oz.bench_selfplay(seed_base, streams, steps, cores)
    dt = time.perf_counter() - t0
    return {"value": moves / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "first %d of the %d games (same seeds, same rules), %d env moves each, oracle/azul_oracle.c on %d threads"
                      % (streams, games, steps, cores),
            "reference_python_steps_per_s_per_core": 2690.0,
            "reference_python_note": "azulnet GameRunner measured in the build container (BASELINE.md), cannot travel to the GPU box"}


def parity_gate(env, games, seed_base, steps_done):
    """Bit-exactness gate: the first games' records after `steps_done` moves must equal the oracle's."""
    from oracle import oracle as oz
    k = min(games, 16)
    recs = env.get_records(0, k)
    for g in range(k):
        s = oz.Stream(seed_base + g)
        s.advance(steps_done, want_records=False)
        if s.record().tobytes() != recs[g].tobytes():
            return "MISMATCH in game %d after %d moves" % (g, steps_done)
    return "ok (%d games x %d moves bit-exact vs oracle)" % (k, steps_done)


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # AZUL_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks (ranks share devices, the
    # all-gather goes through gloo); the driver's multi-GPU runs use the default: nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("AZUL_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd.parallel import TrajectoryGather

    G, T, K, W = args.games, args.chunk, args.steps, args.warmup
    base = args.seed_base + rank * G                     # seeds follow the GLOBAL game id
    env = BatchedAzul(G, device=dev)
    env.seed(base)
    env.runner_init()                                    # GameRunner()
    env.runner_init()                                    # reset()   (DESIGN.md "stream semantics")
    bufs = [env.alloc_trajectory(T, packed_mask=True) for _ in range(2)]
    gather = TrajectoryGather(world, dev, with_masks=args.gather_masks) if (world > 1 and not args.no_gather) else None

    def run(n_steps, timed):
        done_steps, i = 0, 0
        while done_steps < n_steps:
            t = min(T, n_steps - done_steps)
            b = bufs[i & 1]
            if gather is not None:
                gather.wait_buffer_free(i & 1)           # the all-gather that last read this buffer has finished
            env.selfplay(t, b["mask"], b["action"], b["reward"], b["done"], maskbits=b["maskbits"], packed=b["packed"])
            if gather is not None:
                gather.launch(i & 1, b, t)               # side stream, overlaps the next launch
            done_steps += t
            i += 1
        if gather is not None:
            gather.finish()

    run(W, False)
    torch.cuda.synchronize()
    gate = parity_gate(env, G, base, W) if rank == 0 else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    stuck0 = int(env.counters()["stuck"].sum())
    env.timing_begin()
    t0 = time.perf_counter()
    run(K, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = env.timing_end()
    cnt = env.counters()
    stuck = int(cnt["stuck"].sum()) - stuck0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    moves = torch.tensor([float(G * K - stuck)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(moves, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_moves = float(moves.item())

    if rank == 0:
        value = total_moves / elapsed
        avg_launch_s = kern_ms / 1e3 / max(launches, 1)
        steps_per_launch = K / max(launches, 1)
        achieved = ALGO_BYTES_PER_STEP * G * steps_per_launch / avg_launch_s / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))     # PMC-measured HBM bytes per env move (profiles/), scaled to this run's launch size
                traffic = tj["bytes_per_move"] * G * steps_per_launch
            except Exception:
                traffic = None
        out = {
            "metric": "Azul env steps/sec (random-agent self-play), bit-exact vs CPU",
            "value": value, "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/u32 (+f64 sampling)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d concurrent 2-player games per GPU, RandomAgent vs RandomAgent, "
                                   "rules Lid + random first player, seeds base+global_id, auto-reset" % G,
                       "games_per_gpu": G, "global_games": G * world, "moves_per_launch": T,
                       "parallelism": "games sharded by global id; %s" %
                                      (("%s all-gather of the compact trajectory records%s, issued async behind each launch" %
                                        ("RCCL" if backend == "nccl" else backend, " + mask bits" if args.gather_masks else "")) if gather else "no collective")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "azul_selfplay_kernel", "avg_launch_ms": avg_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_STEP * G * steps_per_launch,
                         "note": "working set is cache resident; the path is issue/latency bound, see DESIGN.md"},
            "parity_gate": gate,
            "episodes_finished": int(cnt["episodes"].sum()), "stuck_resets": stuck,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(G, args.seed_base)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
