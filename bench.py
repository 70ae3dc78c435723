#!/usr/bin/env python3
"""bench.py -- Azul env steps/sec, random-agent self-play (BASELINE.json metric), on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--games G] [--chunk T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input = ONE LAUNCH of the persistent self-play
kernel: every one of the G games per GPU plays T = 512 env moves (per move: legal mask -> RandomAgent -> Azul.step
-> reward -> done, auto-reset at game end; the state stays in registers for the whole launch) and the launch writes its
[T][G] trajectory batch.  K steps = K launches = K*T*G env moves per GPU; `value` stays in env steps/s, and
`ms_per_step` x K is the timed region.  Inputs (game records, MT19937 streams) are resident in HBM before the clock
starts.  (The driver's `--steps 20 --warmup 5` therefore times 20 launches, ~10 k moves per game, ~180 episodes each.)

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
    ap.add_argument("--steps", type=int, default=8, help="timed launches (one launch = --chunk env moves per game)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed launches before the clock starts")
    ap.add_argument("--games", type=int, default=4096, help="games per GPU (BASELINE configs[1]: 4096)")
    ap.add_argument("--chunk", type=int, default=512, help="env moves per game per kernel launch")
    ap.add_argument("--seed-base", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL trajectory all-gather")
    ap.add_argument("--gather-masks", action="store_true", help="N>1: also ship the bit-packed legal masks (24 B/move)")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[2] / training-loop lines under `extra`")
    ap.add_argument("--mask-pitch", type=int, default=192, help="byte pitch of a game's legal-mask row (180 = dense, 192 = 64-byte aligned rows)")
    return ap.parse_args()


def _host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a 1-GPU box exposes every core of
    the host in the mask but grants a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(-(-float(quota) // period))))
            break
        except Exception:
            continue
    return n


def cpu_baseline(games, seed_base):
    """The oracle (a plain-C port of the reference path) timed on this box's host cores; bounded sample (~20 s of CPU work):
    (i) all host cores this process may use, one game stream per thread at a time, (ii) one core."""
    from oracle import oracle as oz
    cores = int(os.environ.get("AZUL_CPU_THREADS", _host_cores()))
    streams = games                                      # every game of the workload
    steps = max(1000, min(16000, 250 * cores))           # ~1 M moves per core, at least ~4 M in total
    oz.bench_selfplay(seed_base, min(streams, 64), 200, cores)           # warm the pages
    t0 = time.perf_counter()
    moves, _ = oz.bench_selfplay(seed_base, streams, steps, cores)
    dt = time.perf_counter() - t0
    s1, n1 = min(games, 64), 16000                      # ~1 M moves on one core
    t0 = time.perf_counter()
    moves1, _ = oz.bench_selfplay(seed_base, s1, n1, 1)
    dt1 = time.perf_counter() - t0
    return {"value": moves / dt, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "first %d of the %d games (same seeds, same rules), %d env moves each, oracle/azul_oracle.c on %d threads"
                      % (streams, games, steps, cores),
            "one_core": {"value": moves1 / dt1, "unit": "env steps/s", "cores": 1,
                         "sample": "first %d games, %d env moves each, one thread" % (s1, n1)},
            "reference_python_steps_per_s_per_core": 2690.0,
            "reference_python_note": "azulnet GameRunner measured in the build container (BASELINE.md), cannot travel to the GPU box"}


def parity_gate(env, games, seed_base, moves_done, budget=6000000):
    """Bit-exactness gate: the first games' records AND MT19937 positions after `moves_done` env moves each must equal the
    oracle's (the oracle replays them from the seed; at most `budget` oracle moves in total)."""
    from oracle import oracle as oz
    k = max(1, min(games, 16, budget // max(moves_done, 1)))
    recs = env.get_records(0, k)
    for g in range(k):
        s = oz.Stream(seed_base + g)
        s.advance(moves_done, want_records=False)
        if s.record().tobytes() != recs[g].tobytes():
            return "MISMATCH in game %d after %d moves" % (g, moves_done)
        if s.rng_state()[1] != env.get_rng(g)[1]:
            return "RNG position MISMATCH in game %d after %d moves" % (g, moves_done)
    return "ok (%d games x %d moves bit-exact vs oracle, %d episodes each)" % (k, moves_done, int(s.episodes.value))


F32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: dense f32 matrix peak
FWD_FLOP_PER_GAME = 2 * (136 * 360 + 180 * 180 + 180)            # ActorCritic(136, 180, 180) forward: 163,080 FLOP
GRAD_FLOP_PER_SAMPLE = 391000                                     # forward + backward of the A2C loss


def extras(games):
    """Driver-observed secondary lines (N = 1 only, after the headline measurement): BASELINE configs[2] -- the policy in
    the loop, one launch per 32-move window -- and the training loop (NNRunner.train batched), each with its own roofline
    against the f32 matrix peak.  Kernel time = torch events on the stream the kernels run on."""
    import torch
    from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout
    from azul_deep_reinforcement_learning_amd.learner import A2CLearner
    res = {}
    window, windows = 32, 40
    torch.manual_seed(0)
    net = BatchedActorCritic(136, 180, 180)
    ro = PolicyRollout(net, n_games=games, parts=1, window=window, persistent=True)
    for _ in range(3):
        ro.run_window()
    ro.synchronize()
    torch.cuda.synchronize()
    s = ro.streams[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(s)
    for _ in range(windows):
        ro.run_window()
    e1.record(s)
    ro.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = e0.elapsed_time(e1)
    moves = games * window * windows
    tf = FWD_FLOP_PER_GAME * moves / (kms / 1e3) / 1e12
    res["policy_config"] = {
        "metric": "Azul env steps/sec (ActorCritic policy self-play, full C1 trajectory recorded)", "value": moves / dt, "unit": "env steps/s",
        "config": {"workload": "BASELINE configs[2]: %d games, ActorCritic(136,180,180) f32 inside azul_batch_policy_rollout, "
                               "one launch per %d-move window" % (games, window), "windows_timed": windows},
        "roofline": {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TFLOPS,
                     "traffic": None, "kernel": "azul_policy_rollout2_kernel (+ azul_returns_kernel)", "avg_window_ms": kms / windows,
                     "flop_per_env_move": FWD_FLOP_PER_GAME, "event_bracket_ms": kms, "host_elapsed_ms": dt * 1e3}}
    del ro
    torch.cuda.empty_cache()

    torch.manual_seed(0)
    net = BatchedActorCritic(136, 180, 180).cuda()
    learner = A2CLearner(net)
    ro = PolicyRollout(net, n_games=games, parts=1, window=window, persistent=True, opponent="random", kweights=learner.kweights(), ring=3)

    def one_window():
        tr = ro.run_window()
        ro.join()
        out = learner.update_from_rollout(ro)
        ro.refresh_weights()
        return out

    for _ in range(3):
        one_window()
    torch.cuda.synchronize()
    ep0 = ro.counters()["episodes"]
    t0 = time.perf_counter()
    for _ in range(windows):
        out = one_window()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    samples = float(out["samples"])
    dropped = int(learner.dropped_steps[1]) if learner.dropped_steps is not None else None
    res["training"] = {
        "metric": "A2C training throughput (policy vs RandomAgent opponent, one update per window)", "value": games * window * windows / dt,
        "unit": "agent steps/s", "updates_per_s": windows / dt, "samples_last_update": samples, "steps_dropped_from_ring": dropped,
        "episodes_per_s": (ro.counters()["episodes"] - ep0) / dt,
        "config": {"workload": "NNRunner.train batched: %d games, window %d agent steps (ring of 3 windows: every step of every episode is "
                               "trained once), rollout + selection + gradients + Adam per window" % (games, window), "windows_timed": windows},
        "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": F32_MFMA_PEAK_TFLOPS,
                     "achieved": (GRAD_FLOP_PER_SAMPLE * samples + FWD_FLOP_PER_GAME * games * window) * windows / dt / 1e12,
                     "frac": (GRAD_FLOP_PER_SAMPLE * samples + FWD_FLOP_PER_GAME * games * window) * windows / dt / 1e12 / F32_MFMA_PEAK_TFLOPS,
                     "traffic": None, "kernel": "whole training step (rollout + gradients), wall clock",
                     "note": "agent-step forwards only: the opponent's env moves inside the rollout carry no network evaluation"}}
    return res


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
    # per move: the legal mask (bytes), action, reward, done and the compact record the multi-GPU gather ships; the bit-packed mask
    # is only produced when it is shipped (--gather-masks) or by the one-game-per-wave kernel, whose full variant always writes it
    want_bits = args.gather_masks or os.environ.get("AZUL_SELFPLAY_KERNEL", "2") == "1"
    bufs = [env.alloc_trajectory(T, packed_mask=True, mask_pitch=args.mask_pitch, mask_bits=want_bits) for _ in range(2)]
    gather = TrajectoryGather(world, dev, with_masks=args.gather_masks) if (world > 1 and not args.no_gather) else None

    def run(n_launches):
        for i in range(n_launches):
            b = bufs[i & 1]
            if gather is not None:
                gather.wait_buffer_free(i & 1)           # the all-gather that last read this buffer has finished
            env.selfplay(T, b["mask"], b["action"], b["reward"], b["done"], maskbits=b.get("maskbits"), packed=b["packed"])
            if gather is not None:
                gather.launch(i & 1, b, T)               # side stream, overlaps the next launch
        if gather is not None:
            gather.finish()

    run(W)
    torch.cuda.synchronize()
    # the gate replays >= 1 full episode per checked game whatever --warmup is: top the warm-up up to 128 moves if needed
    gate_moves = W * T
    if gate_moves < 128:
        env.selfplay(128 - gate_moves)
        torch.cuda.synchronize()
        gate_moves = 128
    gate = parity_gate(env, G, base, gate_moves) if rank == 0 else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    stuck0 = int(env.counters()["stuck"].sum())
    t0 = time.perf_counter()
    env.timing_begin()                                   # events are recorded INSIDE the host-timed region
    run(K)
    bracket_ms, launches, kern_ms, kern_launches = env.timing_end()     # records the closing event, waits for it
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cnt = env.counters()
    stuck = int(cnt["stuck"].sum()) - stuck0
    gate_end = parity_gate(env, G, base, gate_moves + K * T) if rank == 0 else None      # ... and the state the timed region left
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    moves = torch.tensor([float(G * K * T - stuck)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(moves, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_moves = float(moves.item())

    if rank == 0:
        value = total_moves / elapsed
        avg_launch_s = kern_ms / 1e3 / max(kern_launches, 1)          # event pair around each launch: the kernel's own duration
        achieved = ALGO_BYTES_PER_STEP * G * T / avg_launch_s / 1e9
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))     # PMC-measured HBM bytes per env move (an EARLIER rocprofv3 --pmc run, see `source`), scaled to one launch
                traffic = tj["bytes_per_move"] * G * T
                traffic_src = "not measured in this run: %s B/move from %s" % (tj["bytes_per_move"], tj.get("source", "profiles/hbm_traffic.json"))
            except Exception:
                traffic = None
        out = {
            "metric": "Azul env steps/sec (random-agent self-play), bit-exact vs CPU",
            "value": value, "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/u32 (+f64 sampling)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d concurrent 2-player games per GPU, RandomAgent vs RandomAgent, "
                                   "rules Lid + random first player, seeds base+global_id, auto-reset" % G,
                       "step_definition": "one step = one launch of the persistent self-play kernel = %d env moves for each of the %d "
                                          "games of a GPU (%d env moves per step and GPU)" % (T, G, T * G),
                       "games_per_gpu": G, "global_games": G * world, "moves_per_launch": T, "env_moves_timed": int(total_moves),
                       "mask_row_pitch_bytes": args.mask_pitch, "mask_bits_stream": bool(want_bits), "selfplay_kernel": os.environ.get("AZUL_SELFPLAY_KERNEL", "2") + " game(s) per wavefront",
                       "parallelism": "games sharded by global id; %s" %
                                      (("%s all-gather of the compact trajectory records%s, issued async behind each launch" %
                                        ("RCCL" if backend == "nccl" else backend, " + mask bits" if args.gather_masks else "")) if gather else "no collective")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "azul_selfplay2_kernel" if os.environ.get("AZUL_SELFPLAY_KERNEL", "2")[:1] != "1" else "azul_selfplay_kernel", "avg_launch_ms": avg_launch_s * 1e3, "launches_timed": kern_launches,
                         "event_bracket_ms": bracket_ms, "host_elapsed_ms": elapsed * 1e3,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_STEP * G * T,
                         "note": "working set is cache resident; the path is issue/latency bound, see DESIGN.md"},
            "parity_gate": gate, "parity_gate_after_timed_region": gate_end,
            "episodes_finished": int(cnt["episodes"].sum()), "stuck_resets": stuck,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(G, args.seed_base)
        if world == 1 and not args.no_extras:
            del bufs, env
            torch.cuda.empty_cache()
            try:
                out["extra"] = extras(G)
            except Exception as e:           # the headline line must survive a failure of the secondary measurements
                out["extra"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
