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
    ap.add_argument("--move-limit", type=int, default=0, help="NOT the metric's workload: azul_batch_set_move_limit (beyond the reference): cut an episode at the "
                    "first end of a round after this many moves, so that games that can never end do not keep their slots (0 = the reference's behaviour)")
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


def parity_gate(env, games, seed_base, moves_done, budget=6000000, move_limit=0):
    """Bit-exactness gate: the first games' records AND MT19937 positions after `moves_done` env moves each must equal the
    oracle's (the oracle replays them from the seed; at most `budget` oracle moves in total)."""
    from oracle import oracle as oz
    k = max(1, min(games, 16, budget // max(moves_done, 1)))
    recs = env.get_records(0, k)
    for g in range(k):
        s = oz.Stream(seed_base + g)
        s.advance(moves_done, want_records=False, move_limit=move_limit)
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


def longer_launches(games, seed_base=0, chunk=2048, launches=5):
    """The headline kernel with FOUR TIMES the moves per launch (same games, same outputs; not the metric's step definition): a launch lasts as long
    as its slowest wave and starts by staging every game's 2.5 KB MT19937 state, so its fixed + tail cost (~24 us of a 512-move launch, ~3 %) is
    spread over more moves.  The grid exactly fills the GPU's two resident waves per SIMD at 4096 games: nothing backfills inside a launch."""
    import torch
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    env = BatchedAzul(games)
    env.seed(seed_base)
    env.runner_init()
    env.runner_init()
    b = env.alloc_trajectory(chunk, packed_mask=True, mask_pitch=192, mask_bits=False)
    run = lambda: env.selfplay(chunk, b["mask"], b["action"], b["reward"], b["done"], packed=b["packed"])
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    stuck0 = int(env.counters()["stuck"].sum())
    dt, avg = clocked(env, run, launches)
    moves = games * chunk * launches - (int(env.counters()["stuck"].sum()) - stuck0)
    return {"value": moves / dt, "unit": "env steps/s", "moves_per_launch": chunk, "launches": launches, "avg_launch_ms": avg,
            "us_per_move_and_launch": avg * 1e3 / chunk, "kernel_env_steps_per_s": games * chunk / (avg / 1e3),
            "nominal_hbm_frac": ALGO_BYTES_PER_STEP * games * chunk / (avg / 1e3) / 1e9 / HBM_PEAK_GBS,
            "note": "same kernel and outputs as the headline, %d instead of 512 moves per launch (profiles/round6_chunk_sweep.txt: 128 .. 4096)" % chunk}


def saturated(seed_base=0, chunk=512):
    """The headline kernel on LARGER GRIDS than BASELINE configs[1] gives a GPU: 8192 games and 32768 games (the whole of configs[3] on ONE
    GPU).  Same kernel, same outputs, launch time from per-launch events (LaunchClock).  The kernel's register allocation admits TWO resident
    waves per SIMD whatever the grid (kernel_resources: the HIP runtime's occupancy calculator on the loaded code object), so these runs
    do NOT raise occupancy: they oversubscribe the grid 2x / 8x -- finished workgroups are backfilled at once and the launch's tail
    (waves of a last partial round, games that hit rare paths) weighs less.  Not the metric's workload."""
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
        dt, avg = clocked(env, run, launches)
        moves = G * chunk * launches - (int(env.counters()["stuck"].sum()) - stuck0)
        try:
            kr = env.kernel_resources(padded_rows=True, mask_bits=False)
        except Exception as e:
            kr = {"error": repr(e)}
        grid_wps = G / 2 / 1024.0
        res["games_%d" % G] = {"value": moves / dt, "unit": "env steps/s", "kernel_env_steps_per_s": G * chunk / (avg / 1e3), "avg_launch_ms": avg,
                               "launches": launches, "grid_waves_per_simd": grid_wps,
                               "resident_waves_per_simd": min(grid_wps, kr["resident_waves_per_simd"]) if "resident_waves_per_simd" in kr else None,
                               "grid_oversubscription": grid_wps / kr["resident_waves_per_simd"] if kr.get("resident_waves_per_simd") else None,
                               "nominal_hbm_frac": ALGO_BYTES_PER_STEP * G * chunk / (avg / 1e3) / 1e9 / HBM_PEAK_GBS}
        res["kernel_resources"] = kr
        del env, b
        torch.cuda.empty_cache()
    res["kernel"] = "azul_selfplay2_kernel (the headline kernel; %d moves per launch)" % chunk
    res["note"] = "oversubscribed grids at the SAME resident occupancy (two waves per SIMD: register-limited), not higher occupancy"
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
            dt, avg = clocked(env, run, launches)
            c = env.counters()
            moves = games * chunk * launches - (int(c["stuck"].sum()) - stuck0)
            res[tag] = {"value": moves / dt, "unit": "env steps/s", "avg_launch_ms": avg,
                        "kernel_env_steps_per_s": games * chunk / (avg / 1e3), "episodes_finished": int(c["episodes"].sum()),
                        "num_actions": env.num_actions,
                        "workload": "%d concurrent %d-player games, %d displays%s, RandomAgent for every seat, rules Lid + random first player, "
                                    "%d moves per launch" % (games, P, env.displays, "" if not ext else
                                                             " + end-of-game bonuses + short deal (beyond the reference, parity unpinned)", chunk)}
            del env, bufs
            torch.cuda.empty_cache()
    res["kernel"] = "azul_x_selfplay_kernel (two games per wavefront)"
    return res


class Telemetry:
    """GPU clock / socket power / hotspot temperature sampled from a SIDE THREAD of this process through the amdsmi Python bindings (reads
    of the driver's metrics table: no new process, no HIP call, nothing re-executed) while the sustained launches run -- the evidence beside
    `sustained`'s block means for WHY the launch time steps up.  Unavailable bindings / metrics are reported, never guessed."""

    def __init__(self, dev_index, period_s=0.02):
        import threading
        self.samples, self.error, self.period, self._stop = [], None, period_s, threading.Event()
        self.power_limit_w = None
        try:
            for pth in ("/opt/rocm/share/amd_smi", "/opt/rocm/libexec/amdsmi_cli"):
                if pth not in sys.path and os.path.isdir(pth):
                    sys.path.append(pth)
            import amdsmi
            self.smi = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.h = hs[dev_index if dev_index < len(hs) else 0]
            try:
                lim = amdsmi.amdsmi_get_power_info(self.h).get("power_limit")
                self.power_limit_w = float(lim) / 1e6 if isinstance(lim, (int, float)) and lim > 1e5 else (float(lim) if isinstance(lim, (int, float)) else None)
            except Exception:
                pass
        except Exception as e:
            self.smi, self.error = None, repr(e)
        self.thread = threading.Thread(target=self._run, daemon=True) if self.smi else None

    def _num(self, v):
        return float(v) if isinstance(v, (int, float)) else None

    def _run(self):
        while not self._stop.is_set():
            try:
                m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
                p = self.smi.amdsmi_get_power_info(self.h)
                self.samples.append((time.perf_counter(), self._num(m.get("current_gfxclk")), self._num(p.get("current_socket_power")),
                                     self._num(m.get("temperature_hotspot")), self._num(m.get("current_uclk"))))
            except Exception as e:
                self.error = repr(e)
                return
            self._stop.wait(self.period)

    def start(self):
        if self.thread:
            self.thread.start()

    def stop(self):
        self._stop.set()
        if self.thread:
            self.thread.join(timeout=2.0)

    def window(self, t0, t1):
        """Means of the samples taken in [t0, t1] (host clock)."""
        rows = [r for r in self.samples if t0 <= r[0] <= t1]
        out = {"samples": len(rows)}
        for i, key in ((1, "sclk_mhz"), (2, "power_w"), (3, "temp_hotspot_c"), (4, "mclk_mhz")):
            vals = [r[i] for r in rows if r[i] is not None]
            out[key] = sum(vals) / len(vals) if vals else None
        return out


class LaunchClock:
    """The duration of every launch of a region, from ONE HIP event after each launch (and one before the first) on the stream the kernels are
    launched on (batch.py: torch's current stream): a launch = the interval between consecutive events = the kernel + the gap to its
    predecessor.  (An event PAIR around each launch -- azul_timing_begin -- costs the whole-job rate 1.2-1.7 % and puts the recording of the
    opening event inside the interval it measures: tools/event_overhead.py, profiles/round6_event_overhead.txt.)"""

    def __init__(self, device):
        import torch
        self.stream = torch.cuda.current_stream(device)
        self.segments = []

    def begin(self, launches):
        import torch
        self.segments.append([torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)])
        self.k = 0
        self.segments[-1][0].record(self.stream)

    def mark(self):
        self.k += 1
        self.segments[-1][self.k].record(self.stream)

    def series(self):
        """ms per launch, all segments in order (call after a synchronisation)"""
        return [seg[i].elapsed_time(seg[i + 1]) for seg in self.segments for i in range(len(seg) - 1)]

    def brackets(self):
        return [seg[0].elapsed_time(seg[-1]) for seg in self.segments]


def clocked(env, run, launches):
    """`launches` calls of run() with one event after each (LaunchClock): (host seconds incl. the final synchronisation, mean ms per launch)"""
    import torch
    clock = LaunchClock(env.device)
    t0 = time.perf_counter()
    clock.begin(launches)
    for _ in range(launches):
        run()
        clock.mark()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ts = clock.series()
    return dt, sum(ts) / max(len(ts), 1)


def never_ending_games(env, run, games, rounds_threshold=100, launches=20):
    """Games of the batch that have been in ONE episode for more than `rounds_threshold` rounds (a game of the reference lasts ~5, at most
    ~15): under the reference's rules with random play a game can reach a state from which it never ends -- e.g. all 20 tiles of one colour
    locked in pattern lines that cannot be completed any more -- and GameRunner's `while not done` loops for ever (the reference would too).
    Such a game keeps its slot for ever.  Every one of its moves is a floor move: until the sampler decided floor-only masks in its one-compare
    path (table rows of nine pairs, azul_tables.hpp) its wave ran ~10 % slower and the launch with it; now it costs nothing.  Evidence, measured
    here: the launch time with these games, and with their records replaced by a neighbour's (AFTER every parity gate, 25 untimed launches
    before each measurement; the batch is discarded afterwards)."""
    import numpy as np
    recs = env.get_records()
    odd = np.flatnonzero(recs["turn_counter"] >= rounds_threshold)
    out = {"count": int(len(odd)), "first_ids": [int(x) for x in odd[:8]], "rounds_threshold": rounds_threshold,
           "rounds_played_in_their_current_episode": [int(recs["turn_counter"][x]) for x in odd[:8]]}
    if len(odd) == 0 or len(odd) > games // 2:
        return out

    def launch_ms():
        run(25)              # untimed: the host work in front of each measurement leaves the GPU idle (the first ~12 launches after a gap are slower)
        return clocked(env, lambda: run(1), launches)[1]

    out["launch_ms_with"] = launch_ms()
    recs = env.get_records()
    good = np.flatnonzero(recs["turn_counter"] < rounds_threshold)
    # replaced ON THE DEVICE through the zero-copy view: the neighbours' records are play's own, and a batch the host has written records
    # into (set_records) would run the self-play instantiation that also marks rule-error-stopped games -- another kernel than the one measured
    import torch
    view = env.records_dev()
    src = torch.as_tensor([int(good[(int(g) + 1 + i) % len(good)]) for i, g in enumerate(odd)], device=view.device)
    view[torch.as_tensor(odd.astype(np.int64), device=view.device)] = view[src]
    torch.cuda.synchronize()
    out["launch_ms_replaced"] = launch_ms()
    return out


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
        elif rank == 0 and what == "headline measurement":        # (a final-barrier timeout adds no second JSON line: rank 0's line is out)
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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # ranks started by someone else's launcher: dmabuf IPC for RCCL (set before HIP initialises)
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
    if args.move_limit:
        env.set_move_limit(args.move_limit)              # (beyond the reference, not the metric's workload; the gates replay with the same limit)
    # per move: the legal mask (bytes), action, reward, done and the compact record the multi-GPU gather ships; the bit-packed mask
    # is only produced when it is shipped (--gather-masks)
    want_bits = args.gather_masks
    bufs = [env.alloc_trajectory(T, packed_mask=True, mask_pitch=args.mask_pitch, mask_bits=want_bits) for _ in range(2)]
    gather = TrajectoryGather(world, dev, with_masks=args.gather_masks) if (world > 1 and not args.no_gather) else None

    def run(n_launches, clock=None):
        for i in range(n_launches):
            b = bufs[i & 1]
            if gather is not None:
                gather.wait_buffer_free(i & 1)           # the all-gather that last read this buffer has finished
            env.selfplay(T, b["mask"], b["action"], b["reward"], b["done"], maskbits=b.get("maskbits"), packed=b["packed"])
            if clock is not None:
                clock.mark()                             # one event per launch, on the launch stream
            if gather is not None:
                gather.launch(i & 1, b, T)               # side stream, overlaps the next launch
        if gather is not None:
            gather.finish()

    # The parity gate comes FIRST: its oracle replay is seconds of host work during which the GPU idles and its clock / power state falls back
    # (after an idle gap the first ~12 launches run up to 5 % slower: tools/ramp_probe.py).  The W untimed warm-up steps the contract asks for
    # follow it, so that the timed region starts on the GPU the warm-up warmed.  (The gate replays >= 1 full episode per checked game.)
    gate_moves = 128
    env.selfplay(gate_moves)
    torch.cuda.synchronize()
    gate = parity_gate(env, G, base, gate_moves, move_limit=args.move_limit) if rank == 0 else None
    if gather is not None:
        coll_.enter("warm-up launches + trajectory all-gather")
    run(W)
    torch.cuda.synchronize()
    gate_moves += W * T
    coll_.barrier("barrier before the timed region")
    torch.cuda.synchronize()
    stuck0 = int(env.counters()["stuck"].sum())
    t0 = time.perf_counter()
    if gather is not None:
        coll_.enter("timed region: launches + trajectory all-gather")
    clock = LaunchClock(dev)
    clock.begin(K)                                       # events are recorded INSIDE the host-timed region
    run(K, clock)
    torch.cuda.synchronize()
    coll_.barrier("barrier after the timed region")
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    cnt = env.counters()
    stuck = int(cnt["stuck"].sum()) - stuck0
    gate_end = parity_gate(env, G, base, gate_moves + K * T, move_limit=args.move_limit) if rank == 0 else None      # ... and the state the timed region left
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    moves = torch.tensor([float(G * K * T - stuck)], dtype=torch.float64, device=dev)
    coll_.all_reduce(el, dist.ReduceOp.MAX, "all-reduce (max) of the ranks' elapsed time")
    coll_.all_reduce(moves, dist.ReduceOp.SUM, "all-reduce (sum) of the ranks' env moves")
    elapsed = float(el.item())
    total_moves = float(moves.item())
    gathered_per_launch = (gather.gathered_bytes // max(W + K, 1)) if gather is not None else 0

    # ---- sustained: the same step, the same buffers, S further launches (~0.8 s at N = 1) after the timed region, in TEN BLOCKS with a host
    #      synchronisation between them (ten ~20 us gaps in ~0.8 s) so that each block has a wall time, a move count and the clock / power /
    #      temperature samples that fall inside it.  `value_sustained` = the LAST block's whole-job rate: the planning number.
    sustained = None
    value_sustained = None
    S = args.sustained
    if S > 0:
        NB = min(10, S)
        sizes = [S // NB + (1 if i >= NB - S % NB else 0) for i in range(NB)]      # S launches in NB blocks (a remainder goes to the last ones)
        tele = Telemetry(dev_index) if rank == 0 else None
        coll_.barrier("barrier before the sustained launches")
        torch.cuda.synchronize()
        if tele:
            tele.start()
        stuck_b = int(env.counters()["stuck"].sum())
        t0 = time.perf_counter()
        if gather is not None:
            coll_.enter("sustained launches + trajectory all-gather")
        sclock = LaunchClock(dev)
        marks, b_moves = [t0], []
        probes = torch.zeros(NB + 1, 3, dtype=torch.int64, device=dev)      # the shader clock as a wave sees it, before block 0 and after every block
        env.clock_probe(probes[0])
        for bi, nl in enumerate(sizes):
            sclock.begin(nl)
            run(nl, sclock)
            env.clock_probe(probes[bi + 1])
            torch.cuda.synchronize()
            marks.append(time.perf_counter())
            sk = int(env.counters()["stuck"].sum())
            b_moves.append(float(G * nl * T - (sk - stuck_b)))
            stuck_b = sk
        torch.cuda.synchronize()
        coll_.barrier("barrier after the sustained launches")
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        if tele:
            tele.stop()
        s_el = torch.tensor([t_end - t0], dtype=torch.float64, device=dev)
        s_moves = torch.tensor([sum(b_moves)], dtype=torch.float64, device=dev)
        b_wall = torch.tensor([marks[i + 1] - marks[i] for i in range(NB)], dtype=torch.float64, device=dev)
        b_mv = torch.tensor(b_moves, dtype=torch.float64, device=dev)
        coll_.all_reduce(s_el, dist.ReduceOp.MAX, "all-reduce (max) of the sustained region's elapsed time")
        coll_.all_reduce(s_moves, dist.ReduceOp.SUM, "all-reduce (sum) of the sustained region's env moves")
        coll_.all_reduce(b_wall, dist.ReduceOp.MAX, "all-reduce (max) of the sustained blocks' wall times")
        coll_.all_reduce(b_mv, dist.ReduceOp.SUM, "all-reduce (sum) of the sustained blocks' env moves")
        if rank == 0:
            series = sclock.series()
            s_kms, s_kn, s_bracket = sum(series), len(series), sum(sclock.brackets())
            edges = [sum(sizes[:i]) for i in range(NB + 1)]
            blocks = [sum(series[edges[i]:edges[i + 1]]) / max(len(series[edges[i]:edges[i + 1]]), 1) for i in range(NB) if edges[i] < len(series)]
            per = sorted(series)
            b_rate = [float(b_mv[i].item()) / float(b_wall[i].item()) for i in range(NB)]
            value_sustained = b_rate[-1]
            tb = [tele.window(marks[i], marks[i + 1]) for i in range(NB)] if tele else [{} for _ in range(NB)]
            pr = probes.cpu().tolist()
            probe_mhz = [100.0 * c / r if r else None for c, r, _ in pr]
            for i in range(NB):
                tb[i]["device_clock_probe_mhz_after_block"] = probe_mhz[i + 1]
            # what this run measured, in words: the largest step between consecutive block means and what the telemetry did across it
            note = "launch time is flat over the %d sustained launches (block means within %.1f %%)" % (S, 100.0 * (max(blocks) / min(blocks) - 1.0))
            if blocks and max(blocks) / min(blocks) > 1.02:
                jumps = [blocks[i + 1] / blocks[i] for i in range(len(blocks) - 1)]
                j = max(range(len(jumps)), key=lambda i: jumps[i])
                at_s = sum(float(b_wall[i].item()) for i in range(j + 1))
                note = ("mean launch time goes from %.4f ms (block 0) to %.4f ms (block %d): %+.1f %%, the largest step (%+.1f %%) after block %d, "
                        "~%.2f s into the sustained launches" % (blocks[0], blocks[-1], len(blocks) - 1, 100.0 * (blocks[-1] / blocks[0] - 1.0),
                                                                 100.0 * (jumps[j] - 1.0), j, at_s))
                m0, m9 = probe_mhz[1], probe_mhz[-1]
                if m0 and m9:
                    note += "; shader clock MEASURED ON THE DEVICE (s_memtime / s_memrealtime probe after the block) %.0f -> %.0f MHz (%+.1f %%)" % (m0, m9, 100.0 * (m9 / m0 - 1.0))
                    note += (": the launch time follows the clock the waves really see (the kernel is issue-bound: time ~ 1 / clock)"
                             if abs((m0 / m9) / (blocks[-1] / blocks[0]) - 1.0) < 0.04 else ": the effective clock alone does not account for the step")
                c0, c9 = tb[0].get("sclk_mhz"), tb[-1].get("sclk_mhz")
                p0, p9 = tb[0].get("power_w"), tb[-1].get("power_w")
                if c0 and c9:
                    note += "; driver-REPORTED shader clock (amdsmi current_gfxclk) %.0f -> %.0f MHz (%+.1f %%)" % (c0, c9, 100.0 * (c9 / c0 - 1.0))
                    if p0 and p9:
                        note += ", socket power %.0f -> %.0f W" % (p0, p9) + (" of a %.0f W limit" % tele.power_limit_w if tele and tele.power_limit_w else "")
                elif tele is not None:
                    note += "; clock / power telemetry unavailable here (%s)" % (tele.error or "no samples")
            sustained = {"launches": S, "env_steps_per_s": float(s_moves.item()) / float(s_el.item()), "seconds": float(s_el.item()),
                         "vs_value": float(s_moves.item()) / float(s_el.item()) / (total_moves / elapsed),
                         "value_sustained": value_sustained, "value_sustained_vs_value": value_sustained / (total_moves / elapsed),
                         "blocks": [dict({"wall_s": float(b_wall[i].item()), "env_steps_per_s": b_rate[i],
                                          "mean_launch_ms": blocks[i] if i < len(blocks) else None}, **tb[i]) for i in range(NB)],
                         "device_clock_probe_mhz_before_block0": probe_mhz[0],
                         "telemetry": {"source": "amdsmi Python bindings (amdsmi_get_gpu_metrics_info: current_gfxclk, temperature_hotspot, current_uclk; "
                                                 "amdsmi_get_power_info: current_socket_power), side thread of rank 0, %d ms period" % int(1e3 * (tele.period if tele else 0)),
                                       "power_limit_w": tele.power_limit_w if tele else None, "samples": len(tele.samples) if tele else 0,
                                       "error": tele.error if tele else None},
                         "launch_ms": {"n": len(per), "min": per[0], "p50": per[len(per) // 2], "p99": per[min(len(per) - 1, int(len(per) * 0.99))],
                                       "max": per[-1], "mean": s_kms / max(s_kn, 1),
                                       "means_of_ten_consecutive_blocks": blocks,
                                       "last_block_kernel_env_steps_per_s": G * T / (blocks[-1] / 1e3) if blocks else None,
                                       "mean_even_odd_launches": [sum(series[0::2]) / max(len(series[0::2]), 1), sum(series[1::2]) / max(len(series[1::2]), 1)]}
                         if per else None,
                         "event_bracket_ms": s_bracket,
                         "parity_gate": parity_gate(env, G, base, gate_moves + (K + S) * T, budget=2500000, move_limit=args.move_limit),
                         "note": "after the timed region: same buffers, same launches (N > 1: same all-gather) in ten blocks with a host synchronisation "
                                 "between them; launch durations = intervals between the events recorded after consecutive launches (rank 0's GPU); `value` is NOT taken from here, "
                                 "`value_sustained` is the last block's whole-job rate.  This run: " + note}
            def run_local(n_launches):                       # (this rank's kernel alone: no collective outside the lock-step regions)
                for _ in range(n_launches):
                    env.selfplay(T, bufs[0]["mask"], bufs[0]["action"], bufs[0]["reward"], bufs[0]["done"], maskbits=bufs[0].get("maskbits"),
                                 packed=bufs[0]["packed"])
            sustained["never_ending_games"] = never_ending_games(env, run_local, G)
            ne = sustained["never_ending_games"]
            if ne.get("count"):
                sustained["note"] += ("; %d of the %d games (first ids %s) are NEVER-ENDING under the reference's rules -- every tile of one colour is "
                                      "locked in pattern lines that can no longer be completed, so no wall row can ever be filled (azul.py:184-191 stays "
                                      "false) and every move of the game is a floor move for ever; the sampler decides such masks in its one-compare path, so "
                                      "the game's wave costs the launch nothing any more (+10 %% in rounds 5 / 6 before): launch %.4f ms with them, %.4f ms "
                                      "with their records replaced by a neighbour's (never_ending_games)"
                                      % (ne["count"], G, ne["first_ids"], ne["launch_ms_with"], ne["launch_ms_replaced"]))

    out = None
    if rank == 0:
        value = total_moves / elapsed
        t_series = clock.series()                                     # one event per launch: kernel + gap to its predecessor
        kern_launches = len(t_series)
        bracket_ms = clock.brackets()[0]
        avg_launch_s = sum(t_series) / 1e3 / max(kern_launches, 1)
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
                                 "on an 8192-game grid -- an OVERSUBSCRIBED grid at the same two resident waves per SIMD (roofline.kernel_resources: "
                                 "the register allocation admits no more), i.e. what workgroup backfill and a shorter tail give, not higher occupancy "
                                 "and not a hardware bound" % ij.get("source", isf))
            except Exception:
                issue = None
        kernel_name = "azul_selfplay2_kernel"
        try:                                             # registers / LDS of the launched instantiation and its RESIDENT occupancy (HIP runtime, loaded code object)
            kres = env.kernel_resources(padded_rows=args.mask_pitch >= 192 and args.mask_pitch % 8 == 0, mask_bits=want_bits)
            kres["grid_waves_per_simd"] = G / 2 / 1024.0
        except Exception as e:
            kres = {"error": repr(e)}
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
            # `value` is the contract's number: K launches (~15 ms at N = 1), i.e. the boost-clock window.  `value_sustained` is the whole-job
            # rate of the LAST of ten blocks of `sustained` (~0.8 s later, clock settled): the planning number (README / NUMBERS quote both)
            "value_sustained": value_sustained,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/u32 (+f64 sampling)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %d concurrent 2-player games per GPU, RandomAgent vs RandomAgent, "
                                   "rules Lid + random first player, seeds base+global_id, auto-reset" % (1 if world == 1 else 3, G),
                       "step_definition": "one step = one launch of the persistent self-play kernel = %d env moves for each of the %d "
                                          "games of a GPU (%d env moves per step and GPU)" % (T, G, T * G),
                       "games_per_gpu": G, "global_games": G * world, "moves_per_launch": T, "env_moves_timed": int(total_moves),
                       "mask_row_pitch_bytes": args.mask_pitch, "mask_bits_stream": bool(want_bits),
                       "move_limit": args.move_limit or None,
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
                         "frac_sustained": (ALGO_BYTES_PER_STEP * G * T / (sustained["launch_ms"]["means_of_ten_consecutive_blocks"][-1] / 1e3) / 1e9 / HBM_PEAK_GBS)
                         if sustained and sustained.get("launch_ms") else None,
                         "kernel": kernel_name, "avg_launch_ms": avg_launch_s * 1e3, "launches_timed": kern_launches,
                         "avg_launch_ms_definition": "mean interval between the HIP events recorded after consecutive launches on the launch stream (one "
                                                     "event before the first): the kernel + the gap to its predecessor; an event PAIR per launch costs "
                                                     "the whole job 1.2-1.7 % (profiles/round6_event_overhead.txt)",
                         "kernel_resources": kres,
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
                             ("saturated", lambda: saturated(args.seed_base, T)), ("longer_launches", lambda: longer_launches(G, args.seed_base)),
                             ("facade_config1", facade_config1),
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
        # the last collective runs under a deadline of its own like every other one: a rank that left early must not keep the rest waiting
        # (rank 0's line is already out: the watchdog only reports and exits non-zero)
        fphase = Phase(["final barrier"])
        fwd = start_watchdog(120, rank, None, fphase, what="final barrier")
        Collectives(dist, world, fphase).barrier("final barrier")
        dist.destroy_process_group()
        fwd.cancel()


if __name__ == "__main__":
    main()
