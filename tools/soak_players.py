"""One-off soak of row N4's kernels (not part of the test suite): for three / four players on the reference's five displays and for the
extended rule sets (beyond the reference, parity unpinned: 2P+1 displays, end-of-game bonuses, short deal, finite bag), 4096 games x
262,144 self-play moves each on the GPU (azul_x_selfplay_kernel, no outputs), then sampled games replayed by the oracle and compared
bit for bit: final record, MT19937 words and index, episode / stuck counters and the GameStatistics sums.  Prints the csrc hash it
ran on."""
import hashlib
import sys
import time

sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import numpy as np
import torch
from azul_deep_reinforcement_learning_amd import BatchedAzul
from azul_deep_reinforcement_learning_amd import _lib as L
from oracle import oracle as oz
from provenance import csrc_hash

print("csrc sha256 %s | libazulhip.so sha256 %s | %s" % (csrc_hash(), hashlib.sha256(open(L.LIB_PATH, "rb").read()).hexdigest()[:16],
                                                        L.lib.azul_version().decode()), flush=True)
N, T, LAUNCHES, BASE = 4096, 2048, 128, 424242
WIDE = L.RULE_DISPLAYS_2P1 | L.RULE_END_BONUS | L.RULE_SHORT_DEAL
CASES = [(3, {"first_player": "Random", "tile_pool": "Lid"}, 0), (4, {"first_player": "Random", "tile_pool": "Lid"}, 0),
         (3, {"first_player": "Random", "tile_pool": "Lid"}, WIDE), (4, {"first_player": "Random", "tile_pool": "Lid"}, WIDE),
         (2, {"first_player": "Random", "tile_pool": "Lid"}, L.RULE_END_BONUS | L.RULE_SHORT_DEAL),
         (4, {"first_player": 3, "tile_pool": "Random"}, L.RULE_DISPLAYS_2P1 | L.RULE_FINITE_BAG | L.RULE_SHORT_DEAL),
         (3, {"first_player": 1, "tile_pool": "Random"}, 0)]
bad = 0
for (P, rules, ext) in CASES:
    env = BatchedAzul(N, rules=rules, players=P, ext_rules=ext)
    env.seed(BASE)
    env.init()
    env.new_round()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(LAUNCHES):
        env.selfplay(T)
    torch.cuda.synchronize()
    dt = time.time() - t0
    recs, cnt = env.get_records(), env.counters()
    print("players %d rules %s ext %d: %.2f s for %.2f G moves, episodes %d, stuck %d" %
          (P, rules, ext, dt, N * T * LAUNCHES / 1e9, int(cnt["episodes"].sum()), int(cnt["stuck"].sum())), flush=True)
    first = oz.FIRST_RANDOM if rules["first_player"] == "Random" else int(rules["first_player"])
    pool = oz.POOL_LID if rules["tile_pool"] == "Lid" else oz.POOL_RANDOM
    for g in list(range(0, 6)) + [N // 2 + 1, N - 1]:
        s = oz.StreamX(BASE + g, P, first_player=first, tile_pool=pool, ext=ext)
        left = T * LAUNCHES
        while left:
            k = min(left, 65536)
            s.advance(k, want_records=False)
            left -= k
        mt, pos = env.get_rng(g)
        omt, opos = s.rng_state()
        ok = (s.record().tobytes() == recs[g].tobytes() and int(pos) == opos and np.array_equal(np.asarray(mt), omt)
              and int(cnt["episodes"][g]) == int(s.episodes.value) and int(cnt["stuck"][g]) == int(s.stuck.value)
              and np.allclose(cnt["stat_sums"][g], s.stats_sum, rtol=0, atol=1e-6))
        bad += (not ok)
        print("  game %d %s (episodes %d)" % (g, "ok" if ok else "MISMATCH", int(s.episodes.value)), flush=True)
    del env
print("N4 SOAK", "PASS" if bad == 0 else "FAIL")
