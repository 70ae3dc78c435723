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
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--games", "256", "--chunk", "128"],
                       env=_env(AZUL_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
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
    ex = out["extra"]
    assert "error" not in ex, ex
    assert ex["policy_config"]["n_gpus"] == 2
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
