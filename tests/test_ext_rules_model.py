"""Row N4's extended rules -- BEYOND THE REFERENCE, PARITY UNPINNED (the reference implements none of them: azulnet/azul.py:19,72,86,
266-288; tests/test_azul.py:14) -- cross-checked between two independently written restatements: the C oracle (oracle/azul_oracle.c,
flags OZ_EXT_*) and the plain-Python model of tests/ext_rules_model.py, move by move on random-agent streams that share CPython's
generator.  With the flags off the oracle is pinned to the reference (tests/test_oracle_golden.py, tests/test_oracle_players.py), so
the same comparison pins the model's base rules."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as oz
from tests import ext_rules_model as M

RULESETS = [("Random", "Lid"), (1, "Random"), ("Random", "Random")]
FLAGSETS = [0, M.DISPLAYS_2P1, M.END_BONUS, M.SHORT_DEAL, M.FINITE_BAG, M.DISPLAYS_2P1 | M.END_BONUS | M.SHORT_DEAL,
            M.DISPLAYS_2P1 | M.END_BONUS | M.SHORT_DEAL | M.FINITE_BAG]


def _oracle_snapshot(g):
    P, D = g.players, g.n_displays
    disp = np.concatenate([g.arr("displays"), g.arr("xdisplays")])[:D]
    return {"displays": disp.tolist(), "center": g.arr("center").tolist(), "pattern_lines": g.arr("pattern_lines")[:P].tolist(),
            "walls": g.arr("walls")[:P].astype(int).tolist(), "floors": g.arr("floors")[:P].tolist(), "score": g.arr("score")[:P].tolist(),
            "current_player": g.current_player, "next_first_player": g.next_first_player, "end_of_game": g.end_of_game,
            "turn_counter": g.turn_counter, "box": g.arr("box").tolist(), "lid": g.arr("lid").tolist(),
            "first_player_stats": g.arr("first_player_stats")[:P].astype(int).tolist(),
            "floor_penalty": g.arr("floor_penalty")[:P].astype(int).tolist(), "max_combo": g.arr("max_combo")[:P].astype(int).tolist(),
            "completed_lines": g.arr("completed_lines")[:P].astype(int).tolist()}


def _compatible(pool, ext):
    return not (pool == "Lid" and ext & M.FINITE_BAG)


@pytest.mark.parametrize("players", [2, 3, 4])
@pytest.mark.parametrize("ext", FLAGSETS)
def test_beyond_the_reference_parity_unpinned_oracle_equals_the_python_model(players, ext):
    lib = oz.lib()
    moves_checked = games = short_deals = box_empty = 0
    long_run = players == 4 and (ext & M.DISPLAYS_2P1) and (ext & M.SHORT_DEAL)      # nine displays drain bag and lid: short deals happen
    steps = 1500 if long_run else 260
    for (first, pool) in RULESETS:
        if not _compatible(pool, ext):
            continue
        for seed in range(6):
            sd = 1000 * players + 10 * seed + ext
            ms = M.ModelStream(sd, players, first, pool, ext)
            xs = oz.StreamX(sd, players, oz.FIRST_RANDOM if first == "Random" else first, oz.POOL_LID if pool == "Lid" else oz.POOL_RANDOM, ext)
            assert xs.num_actions == ms.game.num_actions() == (ms.game.D + 1) * 30
            for t in range(steps):
                before = _oracle_snapshot(xs.g)
                assert before == ms.game.snapshot(), (players, ext, first, pool, seed, t, "state before the move")
                tiles = sum(before["box"]) + sum(before["lid"])
                if ms.game.tracked:           # tile conservation: bag + lid + displays + centre + pattern lines + walls = 100
                    on_table = sum(map(sum, before["displays"])) + sum(before["center"][:5])
                    held = sum(sum(map(sum, pl)) for pl in before["pattern_lines"]) + sum(sum(map(sum, w)) for w in before["walls"])
                    assert tiles + on_table + held == 100, (players, ext, seed, t)
                    if on_table < 4 * ms.game.D and before["turn_counter"] and not any(before["center"][:5]) and before["center"][5]:
                        short_deals += 1
                try:
                    mask, a, done, snap = ms.advance()
                except M.BoxEmpty:
                    # bag and lid ran dry at a deal without the short-deal rule (where the reference raises, azul.py:86-87): the oracle
                    # reports OZ_BOX_EMPTY at the same move; the stream ends here
                    assert not ext & M.SHORT_DEAL
                    with pytest.raises(RuntimeError, match="-> %d" % oz.BOX_EMPTY):
                        xs.advance(1)
                    box_empty += 1
                    break
                o = xs.advance(1)
                assert np.array_equal(o["mask"][0].astype(bool), np.array(mask)), (players, ext, seed, t, "mask")
                assert int(o["action"][0]) == a and int(o["done"][0]) == done, (players, ext, seed, t, a, done)
                rec = oz.unpack_np(o["rec_after"][0], ext=ext)
                got = _oracle_snapshot(rec)
                want = dict(snap)
                if not ms.game.tracked:       # the record holds no tile pools for the infinite "Random" pool
                    want["box"], want["lid"] = got["box"], got["lid"]
                assert got == want, (players, ext, first, pool, seed, t, "state after the move")
                st = ms.rng.getstate()
                assert st[1][624] == int(xs.r.idx) and np.array_equal(np.array(st[1][:624], dtype=np.uint32), np.ctypeslib.as_array(xs.r.mt))
                moves_checked += 1
            assert ms.episodes == int(xs.episodes.value) and ms.stuck == int(xs.stuck.value)
            games += ms.episodes
    assert moves_checked >= 2 * 6 * 200 and games > 0
    if ext & M.SHORT_DEAL:
        assert box_empty == 0
    if long_run:
        assert short_deals > 0


def test_beyond_the_reference_parity_unpinned_end_bonus_known_answers():
    """Hand-made walls: the final bonus is 2 per row, 7 per column, 10 per colour, added once and after the round's clamp."""
    lib = oz.lib()
    r = oz.seeded_rng(1)
    g = oz.Game()
    assert lib.oz_init_ext(C.byref(g), 2, 1, oz.POOL_RANDOM, oz.EXT_END_BONUS, C.byref(r)) == 0
    w = g.arr("walls")
    w[0, 0, :] = 1                                  # player 0: one complete row ...
    for row in range(5):
        w[0, row, (0 - row) % 5] = 1                # ... the complete board column 0 ...
        w[0, row, 2] = 1                            # ... and all five tiles of colour 2
    w[1, 3, :] = 1                                  # player 1: one row
    g.arr("score")[:2] = [5, 0]
    lib.oz_end_game_bonus(C.byref(g))
    rows0 = sum(int(w[0, r].all()) for r in range(5))
    assert g.arr("score")[:2].tolist() == [5 + 2 * rows0 + 7 + 10, 2]
    # the model agrees on the same walls
    m = M.ModelGame(2, 1, "Random", M.END_BONUS, __import__("random").Random(1))
    for p in range(2):
        for row in range(5):
            for c in range(5):
                m.board[p][row][(c + row) % 5] = bool(w[p, row, c])
    m.points = [5, 0]
    m.final_bonus()
    assert m.points == g.arr("score")[:2].tolist()


def test_beyond_the_reference_parity_unpinned_short_deal_and_box_empty():
    """Bag and lid both empty when a round has to be dealt: OZ_BOX_EMPTY without the flag (the reference raises, azul.py:86-87), a
    partial deal with it -- the displays are filled in order as far as the tiles go."""
    lib = oz.lib()
    for ext, pool in ((0, oz.POOL_LID), (oz.EXT_SHORT_DEAL, oz.POOL_LID), (oz.EXT_SHORT_DEAL | oz.EXT_FINITE_BAG, oz.POOL_RANDOM)):
        r = oz.seeded_rng(3)
        g = oz.Game()
        assert lib.oz_init_ext(C.byref(g), 2, 1, pool, ext, C.byref(r)) == 0
        g.arr("box")[:] = [1, 0, 2, 0, 0]
        g.arr("lid")[:] = [0, 3, 0, 0, 0]
        st = lib.oz_new_round(C.byref(g), C.byref(r))
        d = g.arr("displays")
        if ext & oz.EXT_SHORT_DEAL:
            assert st == oz.OK
            assert d.sum(axis=1).tolist() == [4, 2, 0, 0, 0] and d.sum(axis=0).tolist() == [1, 3, 2, 0, 0]
            assert g.arr("box").sum() == 0 and g.arr("lid").sum() == 0 and g.arr("center").tolist() == [0, 0, 0, 0, 0, 1]
        else:
            assert st == oz.BOX_EMPTY


def test_beyond_the_reference_parity_unpinned_rule_combinations():
    lib = oz.lib()
    r = oz.seeded_rng(0)
    g = oz.Game()
    assert lib.oz_init_ext(C.byref(g), 3, 1, oz.POOL_LID, oz.EXT_FINITE_BAG, C.byref(r)) == oz.ILLEGAL_RULE
    for P in (2, 3, 4):
        assert lib.oz_init_ext(C.byref(g), P, 1, oz.POOL_RANDOM, oz.EXT_DISPLAYS_2P1, C.byref(r)) == 0
        assert g.n_displays == 2 * P + 1 and lib.oz_num_actions(C.byref(g)) == (2 * P + 2) * 30
        assert lib.oz_obs_size(C.byref(g)) == 5 * (2 * P + 1) + 6 + 52 * P + 1
    assert lib.oz_init(C.byref(g), 2, 1, oz.POOL_RANDOM, C.byref(r)) == 0 and g.n_displays == 5 and g.ext == 0
    assert lib.oz_obs_size(C.byref(g)) == 136 and lib.oz_num_actions(C.byref(g)) == 180          # game_runner.py:65-72, 115
