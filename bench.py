#!/usr/bin/env python3
"""bench.py -- Azul env steps/sec, random-agent self-play (BASELINE.json metric), on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--games G] [--chunk T]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Started as a plain command with --gpus N > 1 (no WORLD_SIZE in the environment) it launches the N ranks ITSELF: fresh child
processes through torch.distributed.run, started before this process has made any GPU call; rank 0's JSON line appears on this
process's stdout and the exit code is the launcher's (non-zero if any rank failed).

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
    ap.add_argument("--extras-timeout", type=int, default=240, help="seconds after which the secondary measurements are abandoned")
    ap.add_argument("--headline-timeout", type=int, default=900, help="seconds after which a headline measurement that hangs is abandoned")
    ap.add_argument("--mask-pitch", type=int, default=192, help="byte pitch of a game's legal-mask row (180 = dense, 192 = 64-byte aligned rows)")
    ap.add_argument("--sustained", type=int, default=1000, help="launches of the `sustained` object after the timed region (0 = skip)")
    ap.add_argument("--gather-c1", action="store_true", help="N>1, secondary line configs[4]: also all-gather every window's full C1 records "
                                                             "(184 B per agent step; opt-in, the collective north_star names)")
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


def _timed(world, dev, fn):
    """barrier + synchronize on both sides of fn(); the MAX over ranks of the host-clock time."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item()), res


def policy_pytorch_two_streams(games, seed_base=0, window=32, windows=20):
    """BASELINE configs[2] AS WORDED: model.py's ActorCritic evaluated by PyTorch-ROCm (rocBLAS / hipBLASLt GEMMs: addmm + relu + addmm,
    model.py:23-41), interleaved with the env step on two HIP streams -- the batch is split in two halves, each half alternates
    [network on PyTorch -> sampling head -> azul_batch_policy_step] on its own stream (captured once per window as a HIP graph), so one
    half's env step overlaps the other half's GEMMs.  Same trajectory record as the fused kernel."""
    import torch
    from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout
    torch.manual_seed(0)
    net = BatchedActorCritic(136, 180, 180)
    ro = PolicyRollout(net, n_games=games, parts=2, window=window, use_graph=True, fused_mlp=False, persistent=False, seed_base=seed_base)
    for _ in range(3):
        ro.run_window()
    ro.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(windows):
        ro.run_window()
    ro.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    moves = games * window * windows
    c = ro.counters()
    return {"metric": "Azul env steps/sec (model.py policy on PyTorch-ROCm interleaved with env.step on two HIP streams)", "value": moves / dt,
            "unit": "env steps/s", "n_gpus": 1, "ms_per_move": dt / (window * windows) * 1e3, "hip_graph": bool(ro.use_graph),
            "graph_error": ro.graph_error, "episodes_finished": c["episodes"],
            "config": {"workload": "BASELINE configs[2] as worded: %d games, ActorCritic(136,180,180) f32 as PyTorch-ROCm GEMMs, two stream parts of %d "
                                   "games, %d-move windows, %d windows timed" % (games, games // 2, window, windows),
                       "network": "torch.addmm / relu on k-major copies of the module's parameters (rollout.py: fused_mlp=False)",
                       "env": "azul_batch_policy_step (Azul.step + reward + done + auto-reset + next obs / mask), azul_policy_head (sampling)"}}


OPP_FLOP_PER_MOVE = 2 * (136 * 180 + 180 * 180)                  # forward_actor alone (agent.py:73-81): 113,760 FLOP


def policy_vs_policy(games, seed_base=0, window=32, windows=40):
    """GameRunner(opponent=Agent(...)) batched (game_runner.py:27-30; scripts/run_batch.py:6-10; BASELINE.md's "NNRunner vs Agent opponent"
    line, ~590 env steps/s in CPython): the policy is player 1, a SECOND ActorCritic plays every opponent_move() -- replies, player 1's
    forced moves, openings -- inside azul_policy_rollout2_kernel<LID, 2> (azul_batch_policy_rollout_vs), one launch per window of agent
    steps.  env steps = accepted Azul.step calls of both sides.  Roofline against the f32 matrix peak, twice: the network evaluations the
    GAMES needed (agent forward per agent step + forward_actor per opponent move), and what the kernel EXECUTED (a reply round runs the
    opponent's matrices for all 16 games of a workgroup while any of them owes a move)."""
    import torch
    from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout
    torch.manual_seed(0)
    net, opp = BatchedActorCritic(136, 180, 180), BatchedActorCritic(136, 180, 180)
    ro = PolicyRollout(net, n_games=games, parts=1, window=window, persistent=True, opponent=opp, seed_base=seed_base)
    for _ in range(3):
        ro.run_window()
    ro.synchronize()
    torch.cuda.synchronize()
    s = ro.streams[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # every window's reply counts are kept (one 128 KB device copy per window on the rollout's stream) and reduced AFTER the timed region:
    # torch's reductions would cost ~0.7 ms per window here, the kernel's window takes ~1.1 ms
    keep = torch.zeros(windows, window, games, dtype=torch.uint8, device=ro.device)
    ep0 = ro.counters()["episodes"]
    t0 = time.perf_counter()
    e0.record(s)
    for i in range(windows):
        tr = ro.run_window()
        with torch.cuda.stream(s):
            keep[i].copy_(tr[0]["opp_replies"], non_blocking=True)
    e1.record(s)
    ro.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = e0.elapsed_time(e1)
    agent_steps = games * window * windows
    opp_moves = int(keep.sum(dtype=torch.int64))
    executed = int(keep.view(windows, window, -1, 16).amax(dim=3).sum(dtype=torch.int64)) * 16 if games % 16 == 0 else 0
    c = ro.counters()
    useful = (FWD_FLOP_PER_GAME * agent_steps + OPP_FLOP_PER_MOVE * opp_moves) / (kms / 1e3) / 1e12
    run = (FWD_FLOP_PER_GAME * agent_steps + OPP_FLOP_PER_MOVE * executed) / (kms / 1e3) / 1e12
    return {"metric": "Azul env steps/sec (ActorCritic policy vs ActorCritic opponent inside GameRunner, full C1 trajectory of the agent recorded)",
            "value": (agent_steps + opp_moves) / dt, "unit": "env steps/s", "n_gpus": 1, "agent_steps_per_s": agent_steps / dt,
            "opponent_moves_per_agent_step": opp_moves / agent_steps, "reply_rounds_per_agent_step": executed / agent_steps if executed else None,
            "episodes_finished": c["episodes"] - ep0, "stuck_resets": c["stuck"],
            "reference_python_env_steps_per_s": 590.0,
            "config": {"workload": "%d games, GameRunner(opponent=Agent) semantics: two ActorCritic(136,180,180) f32 weight sets inside "
                                   "azul_batch_policy_rollout_vs, one launch per window of %d agent steps, %d windows timed" % (games, window, windows)},
            "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": F32_MFMA_PEAK_TFLOPS, "achieved": useful, "frac": useful / F32_MFMA_PEAK_TFLOPS,
                         "executed": run, "executed_frac": run / F32_MFMA_PEAK_TFLOPS, "traffic": None,
                         "kernel": "azul_policy_rollout2_kernel<LID, 2>", "avg_window_ms": kms / windows, "event_bracket_ms": kms, "host_elapsed_ms": dt * 1e3,
                         "flop_per_agent_step": FWD_FLOP_PER_GAME, "flop_per_opponent_move": OPP_FLOP_PER_MOVE,
                         "note": "achieved = the evaluations the games needed; executed = incl. the masked games of a reply round"}}


def saturated(seed_base=0, chunk=512):
    """The headline kernel with more games than BASELINE configs[1] gives a GPU: 8192 games = four waves per SIMD, 32768 games = the
    whole of configs[3] on ONE GPU (eight waves per SIMD, two rounds).  Same kernel, same outputs, kernel time from the library's event pairs.
    Not the metric's workload: it shows what the instruction-issue bound leaves on the table at 4096 games."""
    import torch
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    res = {}
    for G, launches in ((8192, 6), (32768, 3)):
        env = BatchedAzul(G)
        env.seed(seed_base)
        env.runner_init()
        env.runner_init()
        b = env.alloc_trajectory(chunk, packed_mask=True, mask_pitch=192, mask_bits=False)
        run = lambda: env.selfplay(chunk, b["mask"], b["action"], b["reward"], b["done"], packed=b["packed"])
        run()
        torch.cuda.synchronize()
        stuck0 = int(env.counters()["stuck"].sum())
        t0 = time.perf_counter()
        env.timing_begin()
        for _ in range(launches):
            run()
        _, _, kms, kn = env.timing_end()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        moves = G * chunk * launches - (int(env.counters()["stuck"].sum()) - stuck0)
        avg = kms / max(kn, 1)
        res["games_%d" % G] = {"value": moves / dt, "unit": "env steps/s", "kernel_env_steps_per_s": G * chunk / (avg / 1e3), "avg_launch_ms": avg,
                               "launches": launches, "waves_per_simd": G / 2 / 1024.0,
                               "nominal_hbm_frac": ALGO_BYTES_PER_STEP * G * chunk / (avg / 1e3) / 1e9 / HBM_PEAK_GBS}
        del env, b
        torch.cuda.empty_cache()
    res["kernel"] = "azul_selfplay2_kernel (the headline kernel; %d moves per launch)" % chunk
    return res


def extras(games, world=1, rank=0, dev=None, seed_base=0, backend="nccl", phase=None, gather_c1=False):
    """Driver-observed secondary lines (after the headline measurement; EVERY rank runs them): BASELINE configs[2] (N = 1) /
    configs[4] (N > 1) -- the policy in the loop, one launch per 32-move window, games sharded by global id -- and the training loop
    (NNRunner.train batched; N > 1: data parallel, the step is rollout -> selection -> gradients -> ALL-REDUCE of the global sample
    count and of the flat 82,085-float gradient -> Adam, timed across the ranks), each with its own roofline against the f32 matrix
    peak.  Kernel time = torch events on the stream the kernels run on (rank 0's)."""
    import torch
    import torch.distributed as dist
    from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout
    from azul_deep_reinforcement_learning_amd.learner import A2CLearner
    res = {}
    phase = phase if phase is not None else [""]
    phase[0] = "policy_config"
    window, windows = 32, 40
    base = seed_base + rank * games                     # CPython seeds and Philox keys follow the GLOBAL game id
    coll = "RCCL" if backend == "nccl" else backend
    torch.manual_seed(0)
    net = BatchedActorCritic(136, 180, 180)
    ro = PolicyRollout(net, n_games=games, parts=1, window=window, persistent=True, seed_base=base)
    for _ in range(3):
        ro.run_window()
    ro.synchronize()
    torch.cuda.synchronize()
    s = ro.streams[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c1 = None
    if gather_c1 and world > 1:
        # opt-in: the collective BASELINE configs[4] names -- every window's full C1 records (pack_c1: 184 B per agent step) all-gathered,
        # double-buffered on the rollout's stream, inside the timed region
        from azul_deep_reinforcement_learning_amd.parallel import C1_BYTES, TrajectoryGather, pack_c1
        c1 = TrajectoryGather(world, dev)
        c1_bufs = [torch.empty(window, games, C1_BYTES, dtype=torch.uint8, device=dev) for _ in range(2)]

    def policy_windows():
        e0.record(s)
        for i in range(windows):
            tr = ro.run_window()
            if c1 is not None:
                with torch.cuda.stream(s):
                    c1.wait_buffer_free(i & 1)
                    c1.launch_c1(i & 1, pack_c1(tr[0], window, out=c1_bufs[i & 1]))
        if c1 is not None:
            with torch.cuda.stream(s):
                c1.finish()
        e1.record(s)
        ro.synchronize()

    dt, _ = _timed(world, dev, policy_windows)
    kms = e0.elapsed_time(e1)
    moves = games * window * windows
    tf = FWD_FLOP_PER_GAME * moves / (kms / 1e3) / 1e12
    res["policy_config"] = {
        "metric": "Azul env steps/sec (ActorCritic policy self-play, full C1 trajectory recorded)", "value": world * moves / dt, "unit": "env steps/s",
        "n_gpus": world,
        "config": {"workload": "BASELINE configs[%d]: %d games per GPU (%d in all), ActorCritic(136,180,180) f32 inside azul_batch_policy_rollout, "
                               "one launch per %d-move window" % (2 if world == 1 else 4, games, games * world, window), "windows_timed": windows,
                   "parallelism": ("games sharded by global id; no collective: the C1 trajectory records stay in the HBM of the rank that "
                                   "produced them (DESIGN.md 7)") if c1 is None else
                                  ("games sharded by global id; --gather-c1: every window's full C1 records (%d B per agent step, %.1f MB per rank "
                                   "and window) all-gathered over %s behind the window that produced them, inside the timed region"
                                   % (C1_BYTES, window * games * C1_BYTES / 1e6, coll)),
                   "c1_gather": None if c1 is None else {"bytes_per_agent_step": C1_BYTES, "bytes_into_each_rank_per_window": window * games * C1_BYTES * world,
                                                         "gathered_bytes_timed": c1.gathered_bytes}},
        "roofline": {"bound": "mfma", "achieved": tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TFLOPS,
                     "traffic": None, "kernel": "azul_policy_rollout2_kernel (writes the window's discounted returns itself)", "avg_window_ms": kms / windows,
                     "flop_per_env_move": FWD_FLOP_PER_GAME, "event_bracket_ms": kms, "host_elapsed_ms": dt * 1e3, "scope": "rank 0's GPU"}}
    del ro
    torch.cuda.empty_cache()
    phase[0] = "training"

    torch.manual_seed(0)                                # every rank starts from the same parameters (and keeps them: same updates)
    net = BatchedActorCritic(136, 180, 180).cuda()
    learner = A2CLearner(net)                           # distributed iff a process group exists: count + flat gradient all-reduced
    ro = PolicyRollout(net, n_games=games, parts=1, window=window, persistent=True, opponent="random", kweights=learner.kweights(), ring=3,
                       seed_base=base)

    def one_window():
        ro.run_window()
        ro.join()
        out = learner.update_from_rollout(ro)
        ro.refresh_weights()
        return out

    for _ in range(3):
        one_window()
    torch.cuda.synchronize()
    ep0 = ro.counters()["episodes"]

    def train_windows():
        out = None
        for _ in range(windows):
            out = one_window()
        return out

    dt, out = _timed(world, dev, train_windows)
    samples = float(out["samples"])                     # GLOBAL sample count of the last update
    cnt = torch.tensor([float(ro.counters()["episodes"] - ep0), float(int(learner.dropped_steps[1]) if learner.dropped_steps is not None else 0)],
                       dtype=torch.float64, device=dev)
    chk = torch.stack([p.detach().double().sum() for p in net.parameters()]).sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    if world > 1:
        dist.all_reduce(cnt)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    flop = (GRAD_FLOP_PER_SAMPLE * samples + FWD_FLOP_PER_GAME * games * world * window) * windows / dt / 1e12
    res["training"] = {
        "metric": "A2C training throughput (policy vs RandomAgent opponent, one update per window)", "value": world * games * window * windows / dt,
        "unit": "agent steps/s", "n_gpus": world, "updates_per_s": windows / dt, "samples_last_update": samples,
        "steps_dropped_from_ring": int(cnt[1].item()), "episodes_per_s": float(cnt[0].item()) / dt,
        "ranks_hold_identical_parameters": bool(lo.item() == hi.item()),
        "config": {"workload": "NNRunner.train batched%s: %d games per GPU (%d in all), window %d agent steps (ring of 3 windows: every step of "
                               "every episode is trained once), rollout + selection + gradients + Adam per window"
                               % ("" if world == 1 else " (BASELINE configs[4], data parallel)", games, games * world, window),
                   "windows_timed": windows,
                   "parallelism": "single process" if world == 1 else
                                  "dp%d: games sharded by global id; per update ONE %s all-reduce of the global sample count (4 B) and ONE of the "
                                  "flat gradient + loss sums (82,085 floats, 328 KB); no trajectory leaves its rank" % (world, coll)},
        "roofline": {"bound": "mfma", "unit": "TFLOP/s", "peak": F32_MFMA_PEAK_TFLOPS * world, "achieved": flop,
                     "frac": flop / (F32_MFMA_PEAK_TFLOPS * world),
                     "traffic": None, "kernel": "whole training step (rollout + gradients%s), wall clock" % ("" if world == 1 else " + all-reduce"),
                     "note": "agent-step forwards only: the opponent's env moves inside the rollout carry no network evaluation; "
                             "peak = %d x the f32 matrix peak of one GPU" % world}}
    return res


def facade_config1(budget_s=8.0, max_games=150):
    """BASELINE configs[0] through the drop-in shims (integration/azulnet, i.e. the reference's own single-game Python API on the GPU
    backend): SURVEY 8d config 1 -- random.seed(s); GameRunner(); reset(); loop get_valid_moves -> RandomAgent.get_a_output ->
    GameRunner.step until done, s = 0, 1, ... -- in env steps/s (1 step = 1 accepted Azul.step = GameRunner.move_counter), beside the
    2,690 steps/s/core the reference's CPython measured in the build container.  Every rule evaluation is a kernel launch on a 1-game
    batch; this measures the per-call overhead of the compatibility layer, not the kernels."""
    import random
    import torch
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    from azulnet.game_runner import GameRunner, RandomAgent
    from azul_deep_reinforcement_learning_amd import facade_backend as fb
    agent = RandomAgent()

    def episode(seed):
        random.seed(seed)
        r = GameRunner()
        r.reset()
        done, calls = False, 0
        while not done:
            mask = r.get_valid_moves()
            a = agent.get_a_output(None, torch.from_numpy(mask[None, :]))
            _, done = r.step(a)
            calls += 1
        return r.move_counter, calls

    episode(10 ** 6)                                     # warm-up: lazy backends, allocator
    fb.reset_traffic()
    t0 = time.perf_counter()
    steps = agent_steps = games = 0
    while games < max_games and time.perf_counter() - t0 < budget_s:
        m, c = episode(games)
        steps += m
        agent_steps += c
        games += 1
    dt = time.perf_counter() - t0
    tr = fb.traffic()
    return {"metric": "Azul env steps/sec through the single-game drop-in API (GameRunner loop of SURVEY 8d config 1)", "value": steps / dt,
            "unit": "env steps/s", "games": games, "env_steps": steps, "agent_steps_per_s": agent_steps / dt, "games_per_s": games / dt,
            "reference_python_steps_per_s_per_core": 2690.0,
            "pcie_bytes_per_episode": {"host_to_device": tr["h2d"] / max(games, 1), "device_to_host": tr["d2h"] / max(games, 1)},
            "launches_per_episode": tr["launches"] / max(games, 1), "syncs_per_episode": tr["syncs"] / max(games, 1),
            "config": {"workload": "BASELINE configs[0]: single 2-player game at a time, RandomAgent vs RandomAgent through "
                                   "integration/azulnet (GameRunner / RandomAgent / check_all_valid on libazulhip.so), seeds 0..%d" % (games - 1)}}


def players_selfplay(games, chunk=256, launches=6):
    """Row N4: the persistent self-play kernel for 3 and 4 players (azul_x_selfplay_kernel: two games per wavefront on the 256-byte wide
    record; the loop mask -> RandomAgent -> Azul.step, fresh game at each game end), mask + action + reward + done + compact record
    written (mask rows padded to 192 / 256 / 320 bytes).  Two lines per player count: the reference's rules (five displays), and the
    extended rules -- 2P+1 displays, end-of-game bonuses, short deal: beyond the reference, parity unpinned."""
    import torch
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd import _lib as L
    res = {}
    for P in (3, 4):
        for ext, tag in ((0, "players_%d" % P), (L.RULE_DISPLAYS_2P1 | L.RULE_END_BONUS | L.RULE_SHORT_DEAL, "players_%d_displays_%d" % (P, 2 * P + 1))):
            env = BatchedAzul(games, players=P, ext_rules=ext)
            env.seed(0)
            env.init()
            env.new_round()
            bufs = env.alloc_trajectory(chunk, packed_mask=True, mask_pitch={5: 192, 7: 256, 9: 320}[env.displays], mask_bits=False)
            run = lambda: env.selfplay(chunk, bufs["mask"], bufs["action"], bufs["reward"], bufs["done"], packed=bufs["packed"])
            run()
            torch.cuda.synchronize()
            stuck0 = int(env.counters()["stuck"].sum())
            t0 = time.perf_counter()
            env.timing_begin()
            for _ in range(launches):
                run()
            _, _, kms, kn = env.timing_end()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            c = env.counters()
            moves = games * chunk * launches - (int(c["stuck"].sum()) - stuck0)
            res[tag] = {"value": moves / dt, "unit": "env steps/s", "avg_launch_ms": kms / max(kn, 1),
                        "kernel_env_steps_per_s": games * chunk / (kms / max(kn, 1) / 1e3), "episodes_finished": int(c["episodes"].sum()),
                        "num_actions": env.num_actions,
                        "workload": "%d concurrent %d-player games, %d displays%s, RandomAgent for every seat, rules Lid + random first player, "
                                    "%d moves per launch" % (games, P, env.displays, "" if not ext else
                                                             " + end-of-game bonuses + short deal (beyond the reference, parity unpinned)", chunk)}
            del env, bufs
            torch.cuda.empty_cache()
    res["kernel"] = "azul_x_selfplay_kernel (two games per wavefront)"
    return res


WATCHDOG_EXIT_CODE = 3


def rank_identity(rank, local_rank, dev_index, ndev, backend):
    """What a rank knows about where it runs: host, the devices it can see, the one it bound (index, name, UUID when the runtime gives one)."""
    import socket
    import torch
    name, uuid = "?", None
    try:
        props = torch.cuda.get_device_properties(dev_index)
        name = props.name
        uuid = str(getattr(props, "uuid", "")) or None
    except Exception:
        pass
    return {"rank": int(rank), "local_rank": int(local_rank), "host": socket.gethostname(), "device_count": int(ndev), "device": int(dev_index),
            "name": name, "uuid": uuid, "visible": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES")),
            "shares_device": bool(backend != "nccl" and ndev and local_rank >= ndev)}


class Phase(list):
    """[label of what is running]; `armed` once a watchdog watches it."""
    armed = False


class Collectives:
    """Every collective of the headline measurement goes through here: it refuses to run unless a watchdog is armed (a rank that never
    enters a collective must not leave the others waiting for ever) and it names the phase the watchdog would report."""

    def __init__(self, dist, world, phase):
        self.dist, self.world, self.phase = dist, int(world), phase

    def enter(self, label):
        if not getattr(self.phase, "armed", False):
            raise RuntimeError("collective '%s' before the watchdog was armed" % label)
        self.phase[0] = label

    def barrier(self, label):
        if self.world > 1:
            self.enter(label)
            self.dist.barrier()

    def all_reduce(self, t, op, label):
        if self.world > 1:
            self.enter(label)
            self.dist.all_reduce(t, op=op)
        return t

    def all_gather_object(self, obj, label):
        if self.world == 1:
            return [obj]
        self.enter(label)
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


def start_watchdog(timeout_s, rank, out, phase, _exit=os._exit, what="secondary measurements"):
    """The measurements run under a deadline.  When it passes, a GPU process of this job is stuck (a kernel that does not
    finish, a collective that one rank never entered): rank 0 still prints the headline line -- with the phase that was running under
    `extra.error` (no headline yet: a line holding only `error` / `hung_phase`) -- and EVERY rank leaves with a NON-ZERO exit code, so
    that torchrun / spawn_ranks / the caller see a failed run.  No retry, no re-exec."""
    import threading

    def give_up():
        err = {"error": "%s exceeded %d s in phase '%s'; exit code %d" % (what, timeout_s, phase[0], WATCHDOG_EXIT_CODE), "hung_phase": phase[0]}
        if rank == 0 and out is not None:
            out["extra"] = err
            print(json.dumps(out), flush=True)
        elif rank == 0 and what != "secondary measurements":
            print(json.dumps(err), flush=True)
        sys.stdout.flush()
        sys.stderr.write("bench.py: watchdog fired on rank %d in phase '%s'\n" % (rank, phase[0]))
        sys.stderr.flush()
        _exit(WATCHDOG_EXIT_CODE)

    wd = threading.Timer(timeout_s, give_up)
    wd.daemon = True
    wd.start()
    if isinstance(phase, Phase):
        phase.armed = True
    return wd


def spawn_ranks(args):
    """`python bench.py --gpus N` as a plain command: start the N ranks as FRESH child processes (torch.distributed.run) before this
    process has imported torch or touched a GPU, pass every argument through, relay the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))                      # nothing below has run: no torch import, no GPU call in this process
    import datetime
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # AZUL_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks (ranks share devices, the
    # all-gather goes through gloo); the driver's multi-GPU runs use the default: nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("AZUL_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    hphase = Phase(["start"])
    hwd = start_watchdog(args.headline_timeout, rank, None, hphase, what="headline measurement")      # armed before the first collective
    coll_ = Collectives(dist, world, hphase)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        coll_.enter("init_process_group (%s)" % backend)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))

    from azul_deep_reinforcement_learning_amd import BatchedAzul
    from azul_deep_reinforcement_learning_amd.parallel import TrajectoryGather

    # who is here: every rank states the devices it sees and the one it bound (stderr), and rank 0 records the list in the JSON line
    # (`config.rccl`), so that a multi-GPU record proves that N ranks ran on N different devices over the named backend
    mine = rank_identity(rank, local_rank, dev_index, ndev, backend)
    sys.stderr.write("bench.py: rank %d/%d local_rank %d: torch.cuda.device_count() = %d, bound cuda:%d (%s), backend %s\n"
                     % (rank, world, local_rank, ndev, dev_index, mine["name"], backend if world > 1 else "none"))
    sys.stderr.flush()
    ranks_seen = coll_.all_gather_object(mine, "all_gather_object of the ranks' devices")

    G, T, K, W = args.games, args.chunk, args.steps, args.warmup
    base = args.seed_base + rank * G                     # seeds follow the GLOBAL game id
    env = BatchedAzul(G, device=dev)
    env.seed(base)
    env.runner_init()                                    # GameRunner()
    env.runner_init()                                    # reset()   (DESIGN.md "stream semantics")
    # per move: the legal mask (bytes), action, reward, done and the compact record the multi-GPU gather ships; the bit-packed mask
    # is only produced when it is shipped (--gather-masks)
    want_bits = args.gather_masks
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

    if gather is not None:
        coll_.enter("warm-up launches + trajectory all-gather")
    run(W)
    torch.cuda.synchronize()
    # the gate replays >= 1 full episode per checked game whatever --warmup is: top the warm-up up to 128 moves if needed
    gate_moves = W * T
    if gate_moves < 128:
        env.selfplay(128 - gate_moves)
        torch.cuda.synchronize()
        gate_moves = 128
    gate = parity_gate(env, G, base, gate_moves) if rank == 0 else None
    coll_.barrier("barrier before the timed region")
    torch.cuda.synchronize()
    stuck0 = int(env.counters()["stuck"].sum())
    t0 = time.perf_counter()
    if gather is not None:
        coll_.enter("timed region: launches + trajectory all-gather")
    env.timing_begin()                                   # events are recorded INSIDE the host-timed region
    run(K)
    bracket_ms, launches, kern_ms, kern_launches = env.timing_end()     # records the closing event, waits for it
    torch.cuda.synchronize()
    coll_.barrier("barrier after the timed region")
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cnt = env.counters()
    stuck = int(cnt["stuck"].sum()) - stuck0
    gate_end = parity_gate(env, G, base, gate_moves + K * T) if rank == 0 else None      # ... and the state the timed region left
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    moves = torch.tensor([float(G * K * T - stuck)], dtype=torch.float64, device=dev)
    coll_.all_reduce(el, dist.ReduceOp.MAX, "all-reduce (max) of the ranks' elapsed time")
    coll_.all_reduce(moves, dist.ReduceOp.SUM, "all-reduce (sum) of the ranks' env moves")
    elapsed = float(el.item())
    total_moves = float(moves.item())
    gathered_per_launch = (gather.gathered_bytes // max(W + K, 1)) if gather is not None else 0

    # ---- sustained: the same step, the same buffers, S further launches (~0.8 s at N = 1) after the timed region; not `value`, a check on it
    sustained = None
    S = args.sustained
    if S > 0:
        coll_.barrier("barrier before the sustained launches")
        torch.cuda.synchronize()
        stuck1 = int(env.counters()["stuck"].sum())
        t0 = time.perf_counter()
        if gather is not None:
            coll_.enter("sustained launches + trajectory all-gather")
        env.timing_begin()
        run(S)
        s_bracket, _, s_kms, s_kn = env.timing_end()
        torch.cuda.synchronize()
        coll_.barrier("barrier after the sustained launches")
        torch.cuda.synchronize()
        s_el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        s_moves = torch.tensor([float(G * S * T - (int(env.counters()["stuck"].sum()) - stuck1))], dtype=torch.float64, device=dev)
        coll_.all_reduce(s_el, dist.ReduceOp.MAX, "all-reduce (max) of the sustained region's elapsed time")
        coll_.all_reduce(s_moves, dist.ReduceOp.SUM, "all-reduce (sum) of the sustained region's env moves")
        if rank == 0:
            series = env.timing_launch_ms()
            nb = max(1, len(series) // 10)
            blocks = [sum(series[i:i + nb]) / len(series[i:i + nb]) for i in range(0, len(series), nb)][:10]
            per = sorted(series)
            sustained = {"launches": S, "env_steps_per_s": float(s_moves.item()) / float(s_el.item()), "seconds": float(s_el.item()),
                         "vs_value": float(s_moves.item()) / float(s_el.item()) / (total_moves / elapsed),
                         "launch_ms": {"n": len(per), "min": per[0], "p50": per[len(per) // 2], "p99": per[min(len(per) - 1, int(len(per) * 0.99))],
                                       "max": per[-1], "mean": s_kms / max(s_kn, 1),
                                       "means_of_ten_consecutive_blocks": blocks,
                                       "last_block_kernel_env_steps_per_s": G * T / (blocks[-1] / 1e3) if blocks else None,
                                       "mean_even_odd_launches": [sum(series[0::2]) / max(len(series[0::2]), 1), sum(series[1::2]) / max(len(series[1::2]), 1)]}
                         if per else None,
                         "event_bracket_ms": s_bracket,
                         "parity_gate": parity_gate(env, G, base, gate_moves + (K + S) * T, budget=2500000),
                         "note": "after the timed region: same buffers, same launches (N > 1: same all-gather), launch durations from the "
                                 "library's event pairs (rank 0's GPU); `value` is NOT taken from here.  The block means show the clock "
                                 "coming down under sustained load (a step of ~+10 % in launch time after ~0.45 s on the boxes measured)"}

    out = None
    if rank == 0:
        value = total_moves / elapsed
        avg_launch_s = kern_ms / 1e3 / max(kern_launches, 1)          # event pair around each launch: the kernel's own duration
        achieved = ALGO_BYTES_PER_STEP * G * T / avg_launch_s / 1e9
        traffic, traffic_src, traffic_gbs = None, None, None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))     # PMC-measured HBM bytes per env move (an EARLIER rocprofv3 --pmc run, see `source`), scaled to one launch
                traffic = tj["bytes_per_move"] * G * T
                traffic_gbs = traffic / avg_launch_s / 1e9
                traffic_src = "not measured in this run: %s B/move (kernel %s) from %s" % (tj["bytes_per_move"], tj.get("kernel", "?"),
                                                                                          tj.get("source", "profiles/hbm_traffic.json"))
            except Exception:
                traffic = None
        issue = None
        isf = os.path.join(ROOT, "profiles", "issue_rate.json")
        if os.path.exists(isf):
            try:
                ij = json.load(open(isf))    # PMC passes of an EARLIER rocprofv3 run of this command (tools/summarize_profile.py; see `source`)
                hw = ij.get("hw") or {}
                issue = {"hw_frac": ij.get("hw_frac"),
                         "hw_frac_definition": hw.get("definition"),
                         "hw_frac_if_every_valu_instruction_were_full_rate": hw.get("frac_if_every_valu_instruction_were_full_rate"),
                         "mean_pipe_cycles_per_valu_instruction": hw.get("mean_pipe_cycles_per_valu_instruction"),
                         "scalar_pipe_frac": hw.get("scalar_pipe_frac"),
                         "ceiling_of_a_dependent_chain_at_two_waves_per_simd": hw.get("ceiling_of_a_dependent_chain_at_two_waves_per_simd"),
                         "ceiling_source": hw.get("ceiling_source"),
                         "valu_pipe_share_from_SQ_ACTIVE_INST_VALU": ij.get("valu_pipe_share_from_SQ_ACTIVE_INST_VALU"),
                         "occupancy_frac": ij.get("occupancy_frac"), "occupancy_source": ij.get("occupancy_source"),
                         "wave_time_shares": ij.get("wave_time_shares")}
                issue.update({k: ij.get(k) for k in ("instr_per_game_move", "valu_per_game_move", "salu_per_game_move", "cycles_per_instr_per_simd",
                                                     "waves_per_simd")})
                issue["note"] = ("not measured in this run: %s.  hw_frac prices the kernel's vector instructions (PMC class counts x pipe cycles per "
                                 "wave64 instruction on a SIMD-32) against the SIMD cycles of the launch; occupancy_frac is the kernel against itself "
                                 "with four waves per SIMD (8192 games), not a hardware bound" % ij.get("source", isf))
            except Exception:
                issue = None
        kernel_name = "azul_selfplay2_kernel"
        coll = "RCCL" if backend == "nccl" else backend
        if gather:
            par = ("games sharded by global id (rank r owns games [%d r, %d (r + 1))), no data-path collective; the one exchange step is the %s "
                   "all-gather of the compact trajectory records (4 B per move: action | done | reward%s), issued async behind each launch and "
                   "inside the timed region; the byte masks (180 B per move) are a function of seed + actions and are not shipped"
                   % (G, G, coll, "; + the bit-packed masks, 24 B per move" if args.gather_masks else ""))
        elif world > 1:
            par = "games sharded by global id; no collective (--no-gather)"
        else:
            par = "one GPU: nothing is exchanged (N > 1: games shard by global id, all-gather of the compact trajectory records over RCCL)"
        out = {
            "metric": "Azul env steps/sec (random-agent self-play), bit-exact vs CPU",
            "value": value, "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/u32 (+f64 sampling)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %d concurrent 2-player games per GPU, RandomAgent vs RandomAgent, "
                                   "rules Lid + random first player, seeds base+global_id, auto-reset" % (1 if world == 1 else 3, G),
                       "step_definition": "one step = one launch of the persistent self-play kernel = %d env moves for each of the %d "
                                          "games of a GPU (%d env moves per step and GPU)" % (T, G, T * G),
                       "games_per_gpu": G, "global_games": G * world, "moves_per_launch": T, "env_moves_timed": int(total_moves),
                       "mask_row_pitch_bytes": args.mask_pitch, "mask_bits_stream": bool(want_bits),
                       "selfplay_kernel": "two games per wavefront",
                       "parallelism": par,
                       "rccl": {"backend": (coll if world > 1 else None), "world_size": dist.get_world_size() if world > 1 else 1,
                                "ranks_seen": len(ranks_seen), "distinct_devices": len({(r["host"], r["uuid"] or r["device"]) for r in ranks_seen}),
                                "ranks": ranks_seen, "bytes_per_launch": gathered_per_launch,
                                "bytes_per_launch_note": "bytes the trajectory all-gather delivers INTO each rank per launch (0: nothing gathered)"}},
            # `achieved` / `frac` follow the contract: ALGORITHMIC bytes (445 B per env move) over the kernel's own duration.
            # What really reaches HBM is less (the state never leaves the registers): `traffic` (PMC) and `traffic_gbs` / `traffic_frac`.
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_gbs": traffic_gbs, "traffic_frac": (traffic_gbs / HBM_PEAK_GBS) if traffic_gbs else None,
                         "kernel": kernel_name, "avg_launch_ms": avg_launch_s * 1e3, "launches_timed": kern_launches,
                         "event_bracket_ms": bracket_ms, "host_elapsed_ms": elapsed * 1e3,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_STEP * G * T, "scope": "rank 0's GPU",
                         "limiter": "instruction issue / dependent-issue latency at two waves per SIMD, not HBM (DESIGN.md 3)",
                         "issue": issue,
                         "note": "working set is cache resident; the path is issue/latency bound, see DESIGN.md"},
            "parity_gate": gate, "parity_gate_after_timed_region": gate_end,
            "episodes_finished": int(cnt["episodes"].sum()), "stuck_resets": stuck,
            "sustained": sustained,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(G, args.seed_base)
    if not args.no_extras:
        # secondary lines, run by EVERY rank (N > 1: the data-parallel training step has collectives).  The headline must survive them: an
        # exception is reported under `extra`; if a rank hangs, every rank's watchdog fires, rank 0 prints the headline and all exit.
        hwd.cancel()                                     # the headline is in `out`: from here on the secondary deadline watches
        phase = Phase(["policy_config / training"])      # what the watchdog reports: the secondary measurement that was running
        wd = start_watchdog(args.extras_timeout, rank, out, phase)
        del bufs, env, gather
        torch.cuda.empty_cache()
        first = {}
        if world == 1:
            # (first: measured after the one-launch-per-window kernel has run in the same process the PyTorch-GEMM configuration came out
            # ~40 % slower than on its own -- bench_policy.py, cause not established)
            phase[0] = "policy_pytorch_two_streams"
            try:
                first["policy_pytorch_two_streams"] = policy_pytorch_two_streams(G, args.seed_base)
            except Exception as e:
                first["policy_pytorch_two_streams"] = {"error": repr(e)}
            torch.cuda.empty_cache()
        try:
            ex = extras(G, world, rank, dev, args.seed_base, backend, phase, gather_c1=args.gather_c1)
        except Exception as e:
            ex = {"error": repr(e)}
        ex.update(first)
        if world == 1:
            for name, fn in (("policy_vs_policy", lambda: policy_vs_policy(G, args.seed_base)),
                             ("saturated", lambda: saturated(args.seed_base, T)), ("facade_config1", facade_config1),
                             ("players_selfplay", lambda: players_selfplay(G))):
                phase[0] = name
                try:
                    ex[name] = fn()
                except Exception as e:
                    ex[name] = {"error": repr(e)}
        wd.cancel()
        if rank == 0:
            out["extra"] = ex
    hwd.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
