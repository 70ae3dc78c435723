"""`python bench.py --gpus N` as a PLAIN command (no torchrun around it, no WORLD_SIZE in the environment) must start its N
ranks itself -- fresh child processes, launched before the parent has touched a GPU -- relay rank 0's ONE JSON line and the
launcher's exit code.  (The driver's multi-GPU runs may use either form; round 2's bench.py exited with a usage message here.)

CPU: the parent's argument path, the children's environment and the exit-code relay (the children cannot run without a GPU and
must fail loudly: the product has no CPU path).  GPU: two ranks sharing the test box's one GPU, gloo as the transport
(AZUL_BENCH_BACKEND=gloo, bench.py's rehearsal mode): one JSON line with n_gpus == 2, the data-parallel training line under
`extra` with both ranks holding identical parameters.
Reference: the path itself has no launcher (azulnet/game_runner.py is single-process); BASELINE.json configs[3], configs[4]."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_spawn_command_passes_every_argument_through(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1", "--games", "512"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 7                                     # the launcher's exit code is the command's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--games", "512"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    assert "torch" not in getattr(bench, "__dict__", {})          # the parent spawned before importing torch


@pytest.mark.skipif(torch.cuda.is_available(), reason="on a GPU box the -m gpu test below runs the real thing")
def test_plain_command_spawns_ranks_and_relays_failure():
    """No GPU here: both ranks must fail loudly (no CPU fallback) and the plain command must exit non-zero -- which also proves that
    the children were started with WORLD_SIZE=2 (a child without it would spawn again or complain about --gpus)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--games", "64", "--chunk", "16"],
                       env=_env(AZUL_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2 but WORLD_SIZE" not in r.stderr and "launch N>1 with" not in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_plain_command_two_ranks_one_json_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--games", "256", "--chunk", "128", "--sustained", "4",
                        "--gather-c1"], env=_env(AZUL_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_games"] == 512
    assert out["config"]["env_moves_timed"] == 2 * 256 * 3 * 128 - out["stuck_resets"]
    assert "all-gather" in out["config"]["parallelism"] and "gloo" in out["config"]["parallelism"]
    assert out["parity_gate"].startswith("ok") and out["parity_gate_after_timed_region"].startswith("ok")
    assert "cpu_baseline" not in out                              # rank 0 at N = 1 only
    # the record proves who took part: two ranks, the bytes the trajectory all-gather delivers into a rank per launch, and both ranks' lines
    rc = out["config"]["rccl"]
    assert rc["ranks_seen"] == 2 and rc["world_size"] == 2 and [x["rank"] for x in rc["ranks"]] == [0, 1]
    assert rc["bytes_per_launch"] == 2 * 256 * 128 * 4
    assert all(x["device_count"] >= 1 and x["name"] for x in rc["ranks"])
    assert r.stderr.count("torch.cuda.device_count() =") == 2
    su = out["sustained"]
    assert su["launches"] == 4 and su["launch_ms"]["n"] == 4 and su["parity_gate"].startswith("ok") and su["vs_value"] > 0
    ex = out["extra"]
    assert "error" not in ex, ex
    assert ex["policy_config"]["n_gpus"] == 2
    c1 = ex["policy_config"]["config"]["c1_gather"]               # --gather-c1: the full C1 records of every window, all-gathered
    assert c1["bytes_per_agent_step"] == 184 and c1["gathered_bytes_timed"] == 40 * 2 * 32 * 256 * 184
    assert "all-gathered" in ex["policy_config"]["config"]["parallelism"]
    tr = ex["training"]
    assert tr["n_gpus"] == 2 and tr["ranks_hold_identical_parameters"] is True
    assert "all-reduce" in tr["config"]["parallelism"]
    assert tr["samples_last_update"] > 256 * 32 * 0.5             # a GLOBAL count: more than one rank's window could hold on average


@pytest.mark.gpu
def test_single_gpu_line_keeps_its_shape():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--games", "256", "--chunk", "128", "--no-cpu-baseline"],
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["roofline"]["bound"] == "hbm" and out["roofline"]["frac"] > 0
    assert "nothing is exchanged" in out["config"]["parallelism"]
    assert out["extra"]["training"]["n_gpus"] == 1 and out["extra"]["policy_config"]["roofline"]["bound"] == "mfma"
    assert out["extra"]["policy_config"]["config"]["c1_gather"] is None
    assert out["sustained"]["launches"] == 1000 and out["sustained"]["vs_value"] > 0 and out["sustained"]["parity_gate"].startswith("ok")
    assert out["config"]["rccl"]["ranks_seen"] == 1 and out["config"]["rccl"]["bytes_per_launch"] == 0
    two = out["extra"]["policy_pytorch_two_streams"]
    assert two["value"] > 0 and "PyTorch-ROCm" in two["config"]["workload"]
    sat = out["extra"]["saturated"]
    # larger grids do NOT raise occupancy: the resident waves per SIMD come from the loaded code object (two: register-limited)
    assert sat["games_8192"]["grid_waves_per_simd"] == 4.0 and sat["games_8192"]["resident_waves_per_simd"] == 2.0 and sat["games_32768"]["value"] > 0
    assert sat["kernel_resources"]["vgprs"] > 128 and out["roofline"]["kernel_resources"]["resident_waves_per_simd"] == 2.0
    assert out["value_sustained"] > 0 and len(out["sustained"]["blocks"]) == 10 and "never_ending_games" in out["sustained"]
    vs = out["extra"]["policy_vs_policy"]
    assert vs["value"] > 0 and vs["roofline"]["bound"] == "mfma" and 0.5 < vs["opponent_moves_per_agent_step"] < 2.0


def test_watchdog_prints_the_headline_and_exits_non_zero(capsys):
    """A hang inside the secondary measurements (a GPU process that does not come back, a collective one rank never entered) must not
    be reported as success: the watchdog prints rank 0's headline with the phase that was running, then leaves with a NON-ZERO code."""
    sys.path.insert(0, ROOT)
    import bench
    codes = []
    out = {"metric": "m", "value": 1.0}
    phase = ["training"]
    wd = bench.start_watchdog(0.05, 0, out, phase, _exit=codes.append)
    wd.join(5)
    assert codes == [bench.WATCHDOG_EXIT_CODE] and bench.WATCHDOG_EXIT_CODE != 0
    line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(line) == 1
    got = json.loads(line[0])
    assert got["value"] == 1.0 and got["extra"]["hung_phase"] == "training" and "exceeded" in got["extra"]["error"]
    # other ranks print nothing but leave with the same code
    codes.clear()
    wd = bench.start_watchdog(0.05, 1, None, ["players_selfplay"], _exit=codes.append)
    wd.join(5)
    assert codes == [bench.WATCHDOG_EXIT_CODE]
    assert not [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]


def test_watchdog_exit_code_reaches_the_shell():
    """The real os._exit path, in a child process: exit code 3, the JSON line flushed before it."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "bench.start_watchdog(0.05, 0, {'value': 2.0}, ['facade_config1']); time.sleep(30)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert json.loads(r.stdout.strip().splitlines()[-1])["extra"]["hung_phase"] == "facade_config1"
    assert "watchdog fired" in r.stderr


def test_spawn_command_for_eight_gpus(monkeypatch):
    """The command the driver's 8-GPU run uses when it starts bench.py plainly: eight ranks on 127.0.0.1, every argument passed through."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5", "--gather-c1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 0
    cmd = seen["cmd"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[cmd.index(BENCH) + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5", "--gather-c1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


class _FakeDist:
    def __init__(self, phase):
        self.phase, self.calls = phase, []

    def _note(self, what):
        self.calls.append((what, self.phase[0], bool(self.phase.armed)))

    def barrier(self):
        self._note("barrier")

    def all_reduce(self, t, op=None):
        self._note("all_reduce")

    def all_gather_object(self, out, obj):
        self._note("all_gather_object")
        out[:] = [obj] * len(out)


def test_every_headline_collective_runs_under_an_armed_watchdog_with_its_phase_named():
    sys.path.insert(0, ROOT)
    import bench
    phase = bench.Phase(["start"])
    fake = _FakeDist(phase)
    coll = bench.Collectives(fake, 8, phase)
    with pytest.raises(RuntimeError, match="before the watchdog was armed"):
        coll.barrier("too early")                                  # no rank may wait in a collective nobody watches
    assert fake.calls == []
    wd = bench.start_watchdog(60, 3, None, phase, _exit=lambda c: None, what="headline measurement")
    try:
        assert phase.armed
        got = coll.all_gather_object({"rank": 3}, "ranks")
        coll.barrier("barrier before the timed region")
        coll.all_reduce(None, "MAX", "all-reduce (max)")
    finally:
        wd.cancel()
    assert len(got) == 8
    assert fake.calls == [("all_gather_object", "ranks", True), ("barrier", "barrier before the timed region", True),
                          ("all_reduce", "all-reduce (max)", True)]
    # one rank alone: nothing is called, nothing needs a watchdog
    solo = bench.Collectives(_FakeDist(bench.Phase([""])), 1, bench.Phase([""]))
    solo.barrier("x")
    assert solo.all_gather_object(5, "y") == [5]


def test_main_routes_its_headline_collectives_through_the_wrapper():
    """Source check: main() itself calls no torch.distributed collective directly before the headline is complete -- they all go through
    Collectives (the one exception is the closing barrier in front of destroy_process_group, after the line has been printed)."""
    import inspect
    sys.path.insert(0, ROOT)
    import bench
    src = inspect.getsource(bench.main)
    head = src[:src.index("print(json.dumps(out)")]
    for name in ("dist.barrier(", "dist.all_reduce(", "dist.all_gather", "dist.broadcast("):
        assert name not in head, name
    assert head.index("start_watchdog(args.headline_timeout") < head.index("init_process_group(")      # armed before the first rendezvous


def test_headline_watchdog_prints_an_error_line_when_there_is_no_headline_yet(capsys):
    sys.path.insert(0, ROOT)
    import bench
    codes = []
    phase = bench.Phase(["init_process_group (nccl)"])
    wd = bench.start_watchdog(0.05, 0, None, phase, _exit=codes.append, what="headline measurement")
    wd.join(5)
    assert codes == [bench.WATCHDOG_EXIT_CODE]
    line = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(line) == 1 and json.loads(line[0])["hung_phase"] == "init_process_group (nccl)"
