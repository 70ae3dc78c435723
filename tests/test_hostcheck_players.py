"""Row N4: the 3 / 4 player rule entries (csrc/azul_rules_x.hpp: the body of azul_x_op_kernel, two games per wavefront), compiled for
the host and run under the lockstep 64-lane emulation (tests/hostcheck/simt), replay the reference's Azul(players=3|4) streams of
tests/golden/traj_players.npz bit for bit -- a logic check of the wave-level code before it reaches a GPU (the -m gpu twin is
tests/test_gpu_players.py).  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as oz
from tests.hostcheck import hostcheck as hc

OP_INIT, OP_NEW_ROUND, OP_MOVE, OP_NEXT_PLAYER, OP_COUNT_SCORE, OP_STEP, OP_NONE = "init", "new_round", "move", "next_player", "count_score", "step", "query"


def np_op(rec, players, first, pool, op, action, mt, pos, want_mask=False, want_flags=False, want_stats=False):
    o = hc.x_op(rec, players, first, pool, 0, op, action, mt, pos, want_mask=want_mask, want_flags=want_flags, want_stats=want_stats)
    return o["status"], o["mask"], o["flags"], o["stats"]


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "traj_players.npz"))


def expected_record(gold, key, t, P):
    rec = np.zeros((), dtype=oz.RECORD_NP_DTYPE)
    rec["displays"] = gold[key + "_displays"][t]
    rec["center"] = gold[key + "_center"][t]
    rec["flags"] = int(gold[key + "_cur"][t]) | (int(gold[key + "_nfp"][t]) << 3) | (int(gold[key + "_eog_flag"][t]) << 6)
    rec["pattern_lines"][:P] = gold[key + "_pattern_lines"][t]
    rec["floors"][:P] = gold[key + "_floors"][t]
    w = gold[key + "_walls"][t].reshape(P, 25).astype(np.uint32)
    rec["walls"][:P] = (w << np.arange(25, dtype=np.uint32)).sum(axis=1)
    rec["score"][:P] = gold[key + "_score"][t]
    rec["box"], rec["lid"] = gold[key + "_box"][t], gold[key + "_lid"][t]
    rec["turn_counter"] = gold[key + "_turn_counter"][t]
    rec["first_player_stats"][:P] = gold[key + "_first_player_stats"][t]
    rec["floor_penalty"][:P] = gold[key + "_floor_penalty"][t]
    rec["max_combo"][:P] = gold[key + "_max_combo"][t]
    rec["completed_lines"][:P] = gold[key + "_completed_lines"][t]
    rec["players"] = P
    return rec


@pytest.mark.parametrize("players", [3, 4])
def test_np_core_replays_the_reference_streams(gold, players):
    moves = 0
    for i, key in enumerate(gold["index_key"]):
        key, P, first, pool, seed = str(key), int(gold["index_players"][i]), int(gold["index_first"][i]), int(gold["index_pool"][i]), int(gold["index_seed"][i])
        if P != players:
            continue
        first = 1 if first < 0 else first                       # key absent -> player 1 (azul.py:42-43)
        mt = np.zeros(624, np.uint32)
        hc.lib().sh2_seed(seed, hc.ptr(mt))
        pos = np.array([624], np.uint32)
        rec = np.zeros(256, np.uint8)
        rec[204] = P
        assert np_op(rec, P, first, pool, OP_INIT, 0, mt, pos)[0] == 0
        r = oz.seeded_rng(seed)                                  # the oracle's stream, for the position of the generator
        g = oz.Game()
        L = oz.lib()
        assert L.oz_init(C.byref(g), P, first, pool, C.byref(r)) == 0
        assert rec.view(oz.RECORD_NP_DTYPE)[0].tobytes() == oz.pack_np(g).tobytes(), key
        assert np_op(rec, P, first, pool, OP_NEW_ROUND, 0, mt, pos)[0] == 0
        assert L.oz_new_round(C.byref(g), C.byref(r)) == 0
        assert rec.tobytes() == oz.pack_np(g).tobytes(), key
        acts = gold[key + "_action"]
        for t, a in enumerate(acts):
            _, mask, flags, _ = np_op(rec, P, first, pool, OP_NONE, 0, mt, pos, want_mask=True, want_flags=True)
            assert np.array_equal(np.packbits(mask.astype(bool), bitorder="little"), gold[key + "_mask"][t]), (key, t)
            assert bool(flags & 1) == bool(gold[key + "_eor_before_step"][t])
            if t % 7 == 3 and not mask.all():
                bad = int(np.flatnonzero(mask == 0)[t % int((mask == 0).sum())])
                before, p0 = rec.copy(), int(pos[0])
                assert np_op(rec, P, first, pool, OP_STEP, bad, mt, pos)[0] == 1           # ILLEGAL_MOVE, untouched
                assert np.array_equal(rec, before) and int(pos[0]) == p0
            st, _, flags, stats = np_op(rec, P, first, pool, OP_STEP, int(a), mt, pos, want_flags=True, want_stats=True)
            assert st == 0, (key, t)
            assert rec.tobytes() == expected_record(gold, key, t, P).tobytes(), (key, t)
            assert bool(flags & 2) == bool(gold[key + "_eog_walls"][t]) and bool(flags & 4) == bool(gold[key + "_eog_flag"][t])
            moves += 1
        assert np_op(rec, P, first, pool, OP_STEP, int(acts[-1]), mt, pos)[0] == 2             # GAME_ENDED
        assert np.allclose(stats, gold[key + "_stats"], rtol=0, atol=1e-12), key
        # the generator consumed exactly the reference's number of words
        words = int(gold[key + "_rng_words"][-1])
        r2 = oz.seeded_rng(seed)
        for _ in range(words):
            L.oz_rng_u32(C.byref(r2))
        assert int(pos[0]) == r2.idx and np.array_equal(mt, np.ctypeslib.as_array(r2.mt)), key
    assert moves > 1500


def test_np_core_with_two_players_equals_the_two_player_oracle():
    """The generic core instantiated for P = 2 follows the two-player oracle (whose pinning is tests/test_oracle_golden.py):
    single rule methods on a mid-game state -- move, count_score, next_player -- through the wide record."""
    L = oz.lib()
    r = oz.seeded_rng(11)
    g = oz.Game()
    assert L.oz_init(C.byref(g), 2, 0, oz.POOL_LID, C.byref(r)) == 0 and L.oz_new_round(C.byref(g), C.byref(r)) == 0
    mt = np.ctypeslib.as_array(r.mt).copy()
    pos = np.array([r.idx], np.uint32)
    rec = np.frombuffer(oz.pack_np(g).tobytes(), np.uint8).copy()
    rs = np.random.RandomState(3)
    for t in range(400):
        if g.end_of_game:
            break
        mask = oz.check_all_valid(g)
        a = int(rs.choice(np.flatnonzero(mask)))
        if t % 3 == 0:                                            # the unchecked single methods instead of step
            L.oz_move(C.byref(g), a % 6, (a // 6) % 5, a // 30)
            np_op(rec, 2, 0, oz.POOL_LID, OP_MOVE, a, mt, pos)
            assert rec.tobytes() == oz.pack_np(g).tobytes(), t
            if L.oz_is_end_of_round(C.byref(g)):
                L.oz_count_score(C.byref(g))
                np_op(rec, 2, 0, oz.POOL_LID, OP_COUNT_SCORE, 0, mt, pos)
                assert rec.tobytes() == oz.pack_np(g).tobytes(), t
                if L.oz_is_end_of_game(C.byref(g)):
                    break
                assert L.oz_new_round(C.byref(g), C.byref(r)) == 0
                assert np_op(rec, 2, 0, oz.POOL_LID, OP_NEW_ROUND, 0, mt, pos)[0] == 0
            else:
                L.oz_next_player(C.byref(g))
                np_op(rec, 2, 0, oz.POOL_LID, OP_NEXT_PLAYER, 0, mt, pos)
        else:
            assert L.oz_step(C.byref(g), a % 6, (a // 6) % 5, a // 30, C.byref(r)) == 0
            assert np_op(rec, 2, 0, oz.POOL_LID, OP_STEP, a, mt, pos)[0] == 0
        assert rec.tobytes() == oz.pack_np(g).tobytes(), t
        assert int(pos[0]) == r.idx
