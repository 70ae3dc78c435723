"""One-off soak (not part of the test suite): 4096 games x 1,048,576 self-play moves each on the GPU (4.3 G moves, ~2.4 s),
then 25 games replayed by the oracle and compared bit for bit (final record).  Round 1 result: PASS, 76.2 M episodes, 0 stuck."""
import hashlib, sys, time
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import numpy as np, torch
from azul_deep_reinforcement_learning_amd import BatchedAzul
from oracle import oracle as oz
from azul_deep_reinforcement_learning_amd import _lib as L
from provenance import csrc_hash
print("csrc sha256 %s | libazulhip.so sha256 %s | %s" % (csrc_hash(), hashlib.sha256(open(L.LIB_PATH, "rb").read()).hexdigest()[:16],
                                                        L.lib.azul_version().decode()), flush=True)
n, T, launches = 4096, 2048, 512          # 1,048,576 moves per game, 4.3 G moves in total
env = BatchedAzul(n); env.seed(123456); env.runner_init(); env.runner_init()
t0 = time.time()
for i in range(launches):
    env.selfplay(T)
torch.cuda.synchronize()
print("gpu: %.1f s for %.2f G moves" % (time.time() - t0, n * T * launches / 1e9), flush=True)
recs = env.get_records()
cnt = env.counters()
print("episodes", int(cnt["episodes"].sum()), "stuck", int(cnt["stuck"].sum()), flush=True)
bad = 0
for g in list(range(0, 24)) + [n - 1]:
    s = oz.Stream(123456 + g)
    s.advance(T * launches, want_records=False)
    ok = s.record().tobytes() == recs[g].tobytes()
    mt, pos = env.get_rng(g)
    st = s.rng_state()
    ok2 = (pos == st[1]) if isinstance(st, tuple) else True
    bad += (not ok)
    print("game", g, "ok" if ok else "MISMATCH", flush=True)
print("SOAK", "PASS" if bad == 0 else "FAIL")
