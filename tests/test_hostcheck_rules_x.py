"""The P-player / D-display rules on CPU: csrc/azul_rules_x.hpp (two games per wavefront; the bodies of azul_x_op_kernel and
azul_x_selfplay_kernel) compiled UNMODIFIED by g++ and run under the lockstep 64-lane emulation of tests/hostcheck/simt, against the
oracle -- masks, actions, done flags, 256-byte record snapshots, final records, all 624 MT19937 words + positions, episode / stuck
counters, statistics sums.

  * flags off, five displays, P = 3, 4: the reference's own behaviour (azulnet/azul.py:18-33, 64-89, 118-313; game_runner.py:87-97),
    PINNED -- the same streams tests/golden/traj_players_selfplay.npz records from the real reference are replayed.
  * each extended rule on (2P+1 displays, end-of-game bonuses, short deal, finite bag): BEYOND THE REFERENCE, PARITY UNPINNED -- the
    oracle's OZ_EXT_* restatement of the rulebook (cross-checked by tests/ext_rules_model.py) is the comparison."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as oz

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
XOP = {"query": 0, "init": 1, "new_round": 2, "move": 3, "next_player": 4, "count_score": 5, "step": 6, "random_action": 7, "sample_mask": 8}


def load(name=None):
    name = name or os.environ.get("AZUL_SIMT_X_LIB", "libsimt_rules_x.so")        # run_sanitizers.sh: the _ubsan / _asan builds
    subprocess.check_call(["make", "-s", "-C", HERE, name], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(HERE, name))
    L.shx_selfplay.restype = C.c_longlong
    L.shx_selfplay.argtypes = ([C.c_int] * 3 + [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_ulonglong, C.c_int, C.c_int, C.c_void_p, C.c_int]
                               + [C.c_void_p] * 6)
    L.shx_op.restype = C.c_int
    L.shx_op.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_ulonglong, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 5
    return L


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def split_ext(pool, ext):
    """(device pool code, end_bonus, short_deal, displays rule) from the oracle's tile_pool + OZ_EXT_* flags."""
    xpool = 2 if ext & oz.EXT_FINITE_BAG else (1 if pool == oz.POOL_LID else 0)
    return xpool, int(bool(ext & oz.EXT_END_BONUS)), int(bool(ext & oz.EXT_SHORT_DEAL))


def run_streams(L, P, first, pool, ext, n, T, variant, seed0, margin=0, prepare=None):
    D = 2 * P + 1 if ext & oz.EXT_DISPLAYS_2P1 else 5
    NA = (D + 1) * 30
    streams = [oz.StreamX(seed0 + g, P, first_player=first, tile_pool=pool, ext=ext) for g in range(n)]
    if prepare:
        prepare(streams)
    state = np.stack([np.frombuffer(s.record().tobytes(), np.uint8) for s in streams]).copy()
    mt = np.stack([s.rng_state()[0] for s in streams]).astype(np.uint32).copy()
    pos = np.array([s.rng_state()[1] for s in streams], dtype=np.uint32)
    ep, stuck, ss = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, 10))
    pitch = {5: 192, 7: 256, 9: 320}[D] if variant in (0, 1) else NA
    NL = (NA + 63) // 64
    out = {}
    if variant != 4:
        out = {"mask": np.full((T, n, pitch), 0xEE, np.uint8), "action": np.full((T, n), -7, np.int32), "reward": np.full((T, n), -7, np.int32),
               "done": np.full((T, n), 9, np.uint8)}
        if variant == 0:
            out["maskbits"] = np.zeros((T, n, NL), np.uint64)
        if variant != 3:
            out["packed"] = np.zeros((T, n), np.uint32)
        else:
            out["rec"] = np.zeros((T, n, 256), np.uint8)
            out["maskbits"] = np.zeros((T, n, NL), np.uint64)
    xpool, eb, sd = split_ext(pool, ext)
    ops = L.shx_selfplay(n, P, D, ptr(state), ptr(mt), ptr(pos), ptr(ep), ptr(stuck), ptr(ss), first, xpool, eb, sd, margin, T, variant,
                         ptr(out.get("mask")), pitch, ptr(out.get("maskbits")), ptr(out.get("action")), ptr(out.get("reward")),
                         ptr(out.get("done")), ptr(out.get("packed")), ptr(out.get("rec")))
    assert ops > 0
    return streams, state, mt, pos, ep, stuck, ss, out, NA


def check_streams(L, P, first, pool, ext, n, T, variant, seed0, margin=0, prepare=None):
    streams, state, mt, pos, ep, stuck, ss, out, NA = run_streams(L, P, first, pool, ext, n, T, variant, seed0, margin, prepare)
    finished = 0
    for g, s in enumerate(streams):
        tag = (P, first, pool, ext, variant, g)
        try:
            o = s.advance(T)
        except RuntimeError:
            # bag and lid ran dry without the short-deal rule: the oracle stops at that move (OZ_BOX_EMPTY); the device game stays as it
            # was after its last complete move.  Compare the moves before it.
            assert not ext & oz.EXT_SHORT_DEAL, tag
            s2 = oz.StreamX(seed0 + g, P, first_player=first, tile_pool=pool, ext=ext)
            ok = 0
            while True:
                try:
                    o1 = s2.advance(1)
                except RuntimeError:
                    break
                if variant != 4:
                    assert np.array_equal(out["mask"][ok, g, :NA], o1["mask"][0]) and out["action"][ok, g] == o1["action"][0], tag + (ok,)
                ok += 1
            assert ok < T
            if variant != 4:
                # the slots after the stop: no move played -- marked like stuck slots (action -1, done 2) and counted with them
                assert (out["action"][ok + 1:, g] == -1).all() and (out["done"][ok + 1:, g] == 2).all(), tag
                assert not out["mask"][ok + 1:, g, :NA].any(), tag      # ... with an empty mask row (no stale bytes in the trajectory)
            assert int(stuck[g]) >= T - ok - 1, tag
            continue
        if variant != 4:
            assert np.array_equal(out["mask"][:, g, :NA], o["mask"]), tag
            assert np.array_equal(out["action"][:, g], o["action"]), tag
            assert np.array_equal(out["done"][:, g], o["done"]), tag
            assert not out["reward"][:, g].any(), tag                 # GameRunner's shaped reward is two-player (game_runner.py:50)
        if "maskbits" in out:
            nb = (NA + 7) // 8
            bits = out["maskbits"][:, g].view(np.uint8).reshape(T, -1)[:, :nb]
            assert np.array_equal(bits, np.packbits(o["mask"].astype(bool), axis=1, bitorder="little")), tag
        if "packed" in out:
            p = out["packed"][:, g]
            a = np.where((p & 0xFF) == 0xFF, (p >> 16).astype(np.int32), (p & 0xFF).astype(np.int32))
            a[a == 0xFFFF] = -1
            assert np.array_equal(a, o["action"]) and np.array_equal((p >> 8) & 0xFF, o["done"]), tag
        if "rec" in out:
            assert out["rec"][:, g].tobytes() == o["rec_after"].tobytes(), tag
        assert state[g].tobytes() == s.record().tobytes(), tag
        assert np.array_equal(mt[g], s.rng_state()[0]) and int(pos[g]) == s.rng_state()[1], tag
        assert int(ep[g]) == int(s.episodes.value) and int(stuck[g]) == int(s.stuck.value), tag
        assert np.allclose(ss[g], s.stats_sum, rtol=0, atol=1e-9), tag
        finished += int(ep[g])
    return finished


@pytest.mark.parametrize("players", [3, 4])
def test_flags_off_replays_the_reference_generated_streams(players):
    """PINNED: tests/golden/traj_players_selfplay.npz holds what the REAL reference plays (Azul(players=P) + its RandomAgent on the global
    stream); the emulated kernel body must produce those masks / actions / done flags from the same seeds."""
    L = load()
    gold = np.load(os.path.join(GOLDEN, "traj_players_selfplay.npz"))
    seen = 0
    for i, key in enumerate(gold["index_key"]):
        if int(gold["index_players"][i]) != players:
            continue
        key, seed, first, pool = str(key), int(gold["index_seed"][i]), int(gold["index_first"][i]), int(gold["index_pool"][i])
        if seed % 3:                                             # every third stream: the emulation runs a fiber per lane
            continue
        T = min(len(gold[key + "_action"]), 220)
        streams, state, mt, pos, ep, stuck, ss, out, NA = run_streams(L, players, first if first >= 0 else 1, pool, 0, 1, T, 3, seed)
        assert np.array_equal(np.packbits(out["mask"][:, 0, :180].astype(bool), axis=1, bitorder="little"), gold[key + "_mask"][:T]), key
        assert np.array_equal(out["action"][:, 0], gold[key + "_action"][:T]), key
        assert np.array_equal(out["done"][:, 0], gold[key + "_done"][:T]), key
        seen += 1
    assert seen >= 2


@pytest.mark.parametrize("players", [2, 3, 4])
def test_flags_off_two_games_per_wave_equals_the_oracle(players):
    L = load()
    eps = 0
    for (first, pool) in ((oz.FIRST_RANDOM, oz.POOL_LID), (1, oz.POOL_RANDOM), (2, oz.POOL_LID)):
        for variant in (0, 3, 4):
            eps += check_streams(L, players, first, pool, 0, n=3, T=150, variant=variant, seed0=40 + variant)
    assert eps > 0


EXT_CASES = [oz.EXT_END_BONUS, oz.EXT_SHORT_DEAL, oz.EXT_FINITE_BAG, oz.EXT_DISPLAYS_2P1,
             oz.EXT_DISPLAYS_2P1 | oz.EXT_END_BONUS | oz.EXT_SHORT_DEAL, oz.EXT_DISPLAYS_2P1 | oz.EXT_END_BONUS | oz.EXT_SHORT_DEAL | oz.EXT_FINITE_BAG]


@pytest.mark.parametrize("players", [2, 3, 4])
@pytest.mark.parametrize("ext", EXT_CASES)
def test_beyond_the_reference_parity_unpinned_each_flag_equals_the_oracle(players, ext):
    L = load()
    for (first, pool) in ((oz.FIRST_RANDOM, oz.POOL_LID), (1, oz.POOL_RANDOM)):
        if pool == oz.POOL_LID and ext & oz.EXT_FINITE_BAG:
            continue
        for variant in (1, 3):
            check_streams(L, players, first, pool, ext, n=3, T=140, variant=variant, seed0=900 + ext)


def dense_walls(streams):
    """Crafted states: every wall almost full (every row misses one cell), so that rows, columns and colours complete within a few
    moves and the bonus arithmetic -- per round (reference) or at the end (beyond the reference) -- is exercised hard."""
    rng = np.random.default_rng(7)
    for s in streams:
        s.advance(6)
        w = s.g.arr("walls")
        for p in range(s.g.players):
            w[p] = 1
            for r in range(5):
                w[p, r, rng.integers(5)] = 0
            if p == 0:
                w[p, :, 3] = 1                        # player 0: colour 3 complete already
        s.g.arr("score")[: s.g.players] = rng.integers(0, 40, s.g.players)


@pytest.mark.parametrize("players", [2, 3, 4])
@pytest.mark.parametrize("ext", [0, oz.EXT_END_BONUS, oz.EXT_END_BONUS | oz.EXT_DISPLAYS_2P1])
def test_line_bonuses_on_dense_walls_per_round_and_at_the_end(players, ext):
    """ext = 0: azul.py:266-295 (bonuses in the round the tile lands, pinned through the oracle); with the end-of-game switch: BEYOND
    THE REFERENCE, PARITY UNPINNED.  Also a negative control: the two rules really differ on these states."""
    L = load()
    eps = check_streams(L, players, oz.FIRST_RANDOM, oz.POOL_LID, ext, n=4, T=60, variant=3, seed0=77, prepare=dense_walls)
    assert eps >= 4
    if ext == oz.EXT_END_BONUS:
        a = [oz.StreamX(77 + g, players, ext=0) for g in range(4)]
        b = [oz.StreamX(77 + g, players, ext=ext) for g in range(4)]
        dense_walls(a)
        dense_walls(b)
        ra = np.stack([s.advance(60)["rec_after"]["score"] for s in a])
        rb = np.stack([s.advance(60)["rec_after"]["score"] for s in b])
        assert not np.array_equal(ra, rb)


def near_the_end_of_the_state(streams):
    """Stream positions spread over the last hundred words of the MT19937 state: the round that is dealt after the first ~11 moves then
    fetches its 40 / 56 / 72 words across a regeneration for several of the games."""
    for i, s in enumerate(streams):
        s.r.idx = 624 - 100 + 6 * i


@pytest.mark.parametrize("players,ext", [(3, 0), (3, oz.EXT_DISPLAYS_2P1), (4, oz.EXT_DISPLAYS_2P1 | oz.EXT_SHORT_DEAL)])
def test_factory_draw_across_a_regeneration_and_through_the_fp64_path(players, ext):
    """The parallel draw (five displays: az2::deal_tiles2; seven / nine: deal_parallel_x, 28 or 32 + 4 draws) with its words straddling an
    MT19937 regeneration, and -- with a draw margin that covers every draw -- the literal fp64 decision on the fetched words."""
    L = load()
    check_streams(L, players, oz.FIRST_RANDOM, oz.POOL_LID, ext, n=16, T=40, variant=3, seed0=300, prepare=near_the_end_of_the_state)
    check_streams(L, players, oz.FIRST_RANDOM, oz.POOL_LID, ext, n=4, T=60, variant=3, seed0=310, margin=0x7fffffff)
    check_streams(L, players, oz.FIRST_RANDOM, oz.POOL_LID, ext, n=8, T=30, variant=3, seed0=320, margin=0x7fffffff, prepare=near_the_end_of_the_state)
