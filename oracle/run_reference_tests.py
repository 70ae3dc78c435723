#!/usr/bin/env python3
"""CONTAINER-ONLY conformance run: the reference's OWN test files against this repo's drop-in package.

Builds a scratch tree   <tmp>/azulnet/{__init__,azul,game_runner}.py   (shims from integration/azulnet)
                        <tmp>/azulnet/{agent,model,nn_runner}.py       (symlinks to /root/reference, unchanged)
                        <tmp>/tests -> /root/reference/tests            (symlink)
and runs pytest there.  Without a GPU (this container) the facade is pointed at the TEST-ONLY 64-lane host
emulation of the device core; on a GPU box pass --hip (needs the reference tree, which does not travel).
Nothing is copied into the repository.
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    hip = "--hip" in sys.argv
    tmp = tempfile.mkdtemp(prefix="azul_conformance_")
    pkg = os.path.join(tmp, "azulnet")
    os.makedirs(pkg)
    for f in ("__init__.py", "azul.py", "game_runner.py"):
        os.symlink(os.path.join(ROOT, "integration", "azulnet", f), os.path.join(pkg, f))
    for f in ("agent.py", "model.py", "nn_runner.py"):
        os.symlink(os.path.join(REF, "azulnet", f), os.path.join(pkg, f))
    os.symlink(os.path.join(REF, "tests"), os.path.join(tmp, "reftests"))
    with open(os.path.join(tmp, "conftest.py"), "w") as fh:
        fh.write("import sys\nsys.path.insert(0, %r)\nsys.path.insert(0, %r)\n" % (ROOT, tmp))
        # pytest-benchmark is not installed (no network): the three tests that take its `benchmark` fixture (tests/test_nn_runner.py:45-51,
        # 84-90, 92-98) get a minimal stand-in -- the callable form and .pedantic(fn, args, kwargs, rounds, iterations) -- that RUNS the
        # benchmarked function and reports its wall time; it measures nothing else
        fh.write("import time, pytest\n"
                 "class _Bench:\n"
                 "    def __init__(self): self.seconds = []\n"
                 "    def __call__(self, fn, *a, **k):\n"
                 "        t0 = time.perf_counter(); out = fn(*a, **k); self.seconds.append(time.perf_counter() - t0); return out\n"
                 "    def pedantic(self, fn, args=(), kwargs=None, rounds=1, iterations=1, warmup_rounds=0, setup=None):\n"
                 "        out = None\n"
                 "        for _ in range(max(1, rounds) * max(1, iterations)):\n"
                 "            a, k = (setup() if setup else (args, kwargs or {}))\n"
                 "            out = self(fn, *a, **(k or {}))\n"
                 "        return out\n"
                 "@pytest.fixture\n"
                 "def benchmark(request):\n"
                 "    b = _Bench()\n"
                 "    yield b\n"
                 "    if b.seconds: print('\\n[benchmark stand-in] %s: %d calls, mean %.3f s' % (request.node.name, len(b.seconds), sum(b.seconds) / len(b.seconds)))\n")
        if not hip:
            fh.write("from tests.hostcheck import hostcheck as hc\n"
                     "import azul_deep_reinforcement_learning_amd.facade_backend as fb\n"
                     "fb._FACTORY = hc.call_backend_class()\n")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "-q", "--rootdir", tmp, "-c", os.devnull,
           "--confcutdir", tmp,
           os.path.join(tmp, "reftests", "test_azul.py"), os.path.join(tmp, "reftests", "test_game_runner.py"),
           os.path.join(tmp, "reftests", "test_random_agent.py"), os.path.join(tmp, "reftests", "test_model.py"),
           os.path.join(tmp, "reftests", "test_nn_runner.py")] + [a for a in sys.argv[1:] if a != "--hip"]
    print(" ".join(cmd))
    return subprocess.call(cmd, cwd=tmp, env=env)


if __name__ == "__main__":
    sys.exit(main())
