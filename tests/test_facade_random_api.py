"""Differential test of the single-game facade against the oracle under RANDOM API USE: a few hundred calls per sequence, drawn from
everything a caller of the reference's `Azul` can do -- rule methods in any order, queries, host draws, re-seeding, writes INTO the
attribute arrays, rebound attributes, deep copies -- with the same calls made on the oracle's game and its CPython-exact stream.
After every call the facade's attributes (as the record they pack to), the answers, and the GLOBAL random state must equal the
oracle's.  The facade keeps several things between calls (the record the device holds, the last legal mask, the arrays it unpacked,
the answers it asked ahead for): this is the test that they never leak into a result.  Emulated device core on the CPU suite,
libazulhip.so under -m gpu."""
import copy
import ctypes as C
import random

import numpy as np
import pytest

from oracle import oracle as oz
from tests.facade_fixtures import facade  # noqa: F401

# name -> (rules dict, oracle first-player code, oracle pool, players, oracle extended-rule flags)
RULESETS = {"lid_random": ({"first_player": "Random", "tile_pool": "Lid"}, oz.FIRST_RANDOM, oz.POOL_LID, 2, 0),
            "random_first2": ({"first_player": 2, "tile_pool": "Random"}, 2, oz.POOL_RANDOM, 2, 0),
            "three_players": ({"first_player": "Random", "tile_pool": "Lid"}, oz.FIRST_RANDOM, oz.POOL_LID, 3, 0),
            # beyond the reference, parity unpinned: nine displays, end-of-game bonuses, short deal
            "four_players_extended_parity_unpinned": ({"first_player": "Random", "tile_pool": "Lid", "displays": "2P+1", "bonuses": "end",
                                                       "short_deal": True}, oz.FIRST_RANDOM, oz.POOL_LID, 4,
                                                      oz.EXT_DISPLAYS_2P1 | oz.EXT_END_BONUS | oz.EXT_SHORT_DEAL)}


def _same_stream(rng):
    st = random.getstate()
    return st[1][624] == int(rng.idx) and np.array_equal(np.array(st[1][:624], dtype=np.uint32), np.ctypeslib.as_array(rng.mt))


def _record_of(game):
    if game.players != 2 or game.ext:
        return oz.pack_np(game).tobytes()
    q = oz.Runner()
    C.memmove(C.byref(q.game), C.byref(game), C.sizeof(oz.Game))
    return oz.pack(q).tobytes()


def _check(g, og, rng, what):
    assert g._to_record().tobytes() == _record_of(og), what
    assert _same_stream(rng), what


@pytest.mark.parametrize("ruleset", sorted(RULESETS))
def test_random_call_sequences_match_the_oracle(facade, ruleset):
    rules, first, pool, P, ext = RULESETS[ruleset]
    wide = P != 2 or ext != 0

    def decode(og, a):
        d, c, p = C.c_int(0), C.c_int(0), C.c_int(0)
        oz.lib().oz_deserialize_x(C.byref(og), int(a), C.byref(d), C.byref(c), C.byref(p))
        return d.value, c.value, p.value

    lib = oz.lib()
    pick = random.Random(4711 + len(ruleset))              # the test's own generator: never the global stream
    agent = facade.RandomAgent()
    calls = 0
    for seq in range(3):
        def fresh(seed):
            random.seed(seed)
            rng = oz.seeded_rng(seed)
            g, og = facade.Azul(players=P, rules=rules), oz.Game()
            assert lib.oz_init_ext(C.byref(og), P, first, pool, ext, C.byref(rng)) == 0
            g.new_round()
            assert lib.oz_new_round(C.byref(og), C.byref(rng)) == 0
            _check(g, og, rng, "start")
            return g, og, rng

        g, og, rng = fresh(100 + seq)
        for t in range(260):
            op = pick.choice(["step", "step", "step", "step", "illegal", "mask", "legal", "flags", "draw", "agent", "edit_score",
                              "edit_lines", "rebind", "copy", "move", "count", "new_round", "reseed", "stats", "next"])
            what = (ruleset, seq, t, op)
            mask = oz.check_all_valid_x(og)
            if op == "step" and mask.any() and not og.end_of_game:
                a = int(pick.choice(np.flatnonzero(mask).tolist()))
                d, c, p = decode(og, a)
                st = lib.oz_step(C.byref(og), d, c, p, C.byref(rng))
                if st == 0:
                    g.step(d, c, p)
                else:                                      # (box and lid empty: the reference raises from random.choices)
                    with pytest.raises(ValueError):
                        g.step(d, c, p)
                    g, og, rng = fresh(1000 * seq + t)         # (where the reference raises mid-way the object is not used further)
                    continue
            elif op == "illegal":
                bad = np.flatnonzero(~mask)
                if bad.size and not og.end_of_game:
                    d, c, p = decode(og, int(pick.choice(bad.tolist())))
                    assert lib.oz_step(C.byref(og), d, c, p, C.byref(rng)) == 1
                    with pytest.raises(facade.IllegalMove):
                        g.step(d, c, p)
            elif op == "mask":
                assert np.array_equal(facade.check_all_valid(g), mask), what
            elif op == "legal":
                d, c, p = pick.randrange(og.n_displays + 1), pick.randrange(5), pick.randrange(6)
                assert bool(g.is_legal_move(d, c, p)) == bool(lib.oz_is_legal_move(C.byref(og), d, c, p)), what
            elif op == "flags":
                assert bool(g.is_end_of_round()) == bool(lib.oz_is_end_of_round(C.byref(og))), what
                assert bool(g.is_end_of_game()) == bool(lib.oz_is_end_of_game(C.byref(og))), what
            elif op == "draw":
                assert random.random() == lib.oz_rng_random(C.byref(rng)), what
            elif op == "agent" and mask.any():
                m8 = np.ascontiguousarray(mask.astype(np.uint8))
                want = lib.oz_random_agent_x(m8.ctypes.data_as(C.POINTER(C.c_uint8)), m8.size, C.byref(rng))
                assert agent.get_a_output(None, mask[None, :]) == want, what
            elif op == "edit_score":
                i, k = pick.randrange(P), pick.randrange(0, 40)
                g.score[i] = k                             # a write INTO the array the last call unpacked
                og.score[i] = k
            elif op == "edit_lines":
                pl, row = pick.randrange(P), pick.randrange(5)
                col = pick.randrange(5)
                if not og.walls[pl][row][col]:
                    n = pick.randrange(0, row + 2)
                    g.pattern_lines[pl, row, :] = 0
                    g.pattern_lines[pl, row, col] = n
                    for cc in range(5):
                        og.pattern_lines[pl][row][cc] = n if cc == col else 0
            elif op == "rebind":
                fl = [pick.randrange(0, 8) for _ in range(P)]
                g.floors = np.array(fl, dtype=float) if pick.random() < 0.5 else fl      # a float array / a plain list instead of the int array
                for i in range(P):
                    og.floors[i] = fl[i]
            elif op == "copy":
                g = copy.deepcopy(g)                       # (the reference's GameRunner deep-copies the game for its what-if score)
            elif op == "move" and mask.any() and not og.end_of_game:
                d, c, p = decode(og, int(pick.choice(np.flatnonzero(mask).tolist())))
                g.move(d, c, p)
                lib.oz_move(C.byref(og), d, c, p)
            elif op == "count":
                g.count_score()
                lib.oz_count_score(C.byref(og))
            elif op == "new_round":
                st = lib.oz_new_round(C.byref(og), C.byref(rng))
                if st == 0:
                    g.new_round()
                else:
                    with pytest.raises(ValueError):
                        g.new_round()
                    g, og, rng = fresh(1000 * seq + t)
                    continue
            elif op == "reseed":
                s2 = pick.randrange(10 ** 6)
                random.seed(s2)
                rng = oz.seeded_rng(s2)
            elif op == "stats":
                want = oz.get_statistics(og)
                got = g.get_statistics()
                assert all(got[k] == want[k] or (got[k] != got[k] and want[k] != want[k]) for k in want), what
            elif op == "next":
                g.next_player()
                lib.oz_next_player(C.byref(og))
            _check(g, og, rng, what)
            calls += 1
    assert calls > 500, calls                              # (a sequence stops early only where the reference itself would raise)
    random.seed()
