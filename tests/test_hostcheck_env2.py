"""The policy rollout kernel's ENV SIDE on CPU: csrc/azul_env2.hpp (GameRunner.step / reset / get_state and the RandomAgent opponent,
two games per wavefront, as azul_policy_rollout2_kernel calls them between its matrix phases) compiled UNMODIFIED by g++ and run
under the lockstep 64-lane emulation of tests/hostcheck/simt against the oracle: every published legal mask (bytes and the packed
words the head reads), observation, player, reward and done flag, the final records, all 624 MT19937 words + positions and the
episode counters -- for policy-on-both-sides self-play and for the GameRunner opponent loop, an odd batch, three rule sets, and the
refusal paths (illegal action, out-of-range action, stuck slot).  The same runs under UBSan / ASan: tests/hostcheck/run_sanitizers.sh.
Reference: azulnet/game_runner.py:43-97, azulnet/azul.py:296-313, azulnet/nn_runner.py:17-47."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as oz

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck")
RULES = {"lid_randomfirst": (0, 1), "random_first1": (1, 0), "lid_first2": (2, 1)}      # (first_player code, tile_pool code)
ST_ILLEGAL, ST_STUCK, ST_BAD_ACTION = 1, 3, 4          # csrc/azul_common.hpp


def load(name=None):
    name = name or os.environ.get("AZUL_SIMT_ENV_LIB", "libsimt_env2.so")           # run_sanitizers.sh: the _ubsan / _asan builds
    subprocess.check_call(["make", "-s", "-C", HERE, name], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(HERE, name))
    L.sh2_rollout_env.restype = C.c_longlong
    L.sh2_rollout_env.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_ulonglong, C.c_int, C.c_int] + [C.c_void_p] * 8
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def oracle_play(rec, mt, pos, T, first, pool, opponent, pick, move_limit=0):
    """T agent moves in the oracle, the action of each chosen by `pick(mask, t)` from the legal mask of the state it is played in
    (what the head does); returns the actions and everything the kernel's env side publishes.  `move_limit` > 0: the move-limit extension
    (beyond the reference; oz_step_limited / oz_runner_step_limited): `done` then holds the codes 0 / 1 / 3."""
    L = oz.lib()
    fp = first if first else oz.FIRST_RANDOM
    q = oz.unpack(rec, pool, fp)
    r = oz.Rng()
    L.oz_rng_set(C.byref(r), mt.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos))
    out = {k: [] for k in ("action", "mask", "obs", "player", "reward", "done")}

    def publish():
        out["mask"].append(oz.check_all_valid(q.game))
        cur = q.game.current_player
        out["player"].append(cur)
        out["obs"].append(oz.get_state(q.game, 0 if opponent else cur - 1))

    publish()
    for t in range(T):
        a = pick(np.asarray(out["mask"][-1], bool), t)
        out["action"].append(a)
        if opponent:
            rew, dn = C.c_int64(0), C.c_int(0)
            if move_limit:
                assert L.oz_runner_step_limited(C.byref(q), int(a), C.byref(r), None, None, move_limit, C.byref(rew), C.byref(dn)) == 0
            else:
                assert L.oz_runner_step(C.byref(q), int(a), C.byref(r), C.byref(rew), C.byref(dn)) == 0     # game_runner.py:43-55
            out["reward"].append(rew.value)
            out["done"].append(int(dn.value) if move_limit else bool(dn.value))
            if dn.value:
                assert L.oz_runner_reset(C.byref(q), C.byref(r)) == 0                                   # :76-85
        else:
            st = L.oz_step_limited(C.byref(q.game), int(a) % 6, (int(a) // 6) % 5, int(a) // 30, C.byref(r), q.move_counter + 1, move_limit)
            assert st in (0, oz.TRUNCATED)
            q.move_counter += 1
            phi = L.oz_potential(C.byref(q.game))
            out["reward"].append(phi - q.player_score)
            q.player_score = phi
            dn = 3 if st == oz.TRUNCATED else int(bool(L.oz_is_end_of_game(C.byref(q.game))))
            out["done"].append(dn if move_limit else bool(dn))
            if dn:
                assert L.oz_runner_init(C.byref(q), fp, pool, C.byref(r)) == 0
        publish()
    mt_out = np.array(r.mt[:], dtype=np.uint32)
    return out, oz.pack(q), mt_out, r.idx


def start_batch(n, seed0, first, pool, warm=0):
    streams = [oz.Stream(seed0 + g, first_player=first if first else oz.FIRST_RANDOM, tile_pool=pool) for g in range(n)]
    for g, s in enumerate(streams):
        if warm:
            s.advance(warm + 3 * g)                          # the games of a batch are at different points of their rounds
    state = np.stack([np.frombuffer(s.record().tobytes(), np.uint8) for s in streams]).copy()
    mt = np.stack([s.rng_state()[0] for s in streams]).astype(np.uint32).copy()
    pos = np.array([s.rng_state()[1] for s in streams], dtype=np.uint32)
    return state, mt, pos


def emulate(L, state, mt, pos, first, pool, opponent, actions, margin=0):
    T, n = actions.shape
    ep, stuck, ss = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, 10))
    o = {"obs": np.full((T + 1, n, 136), -99, np.float32), "mask": np.full((T + 1, n, 180), 0xEE, np.uint8),
         "player": np.full((T + 1, n), 9, np.uint8), "maskbits": np.zeros((T + 1, n, 3), np.uint64),
         "reward": np.full((T, n), -7, np.int32), "done": np.full((T, n), 9, np.uint8), "status": np.full(n, 99, np.uint8)}
    acts = np.ascontiguousarray(actions, np.int32)
    ops = L.sh2_rollout_env(n, ptr(state), ptr(mt), ptr(pos), ptr(ep), ptr(stuck), ptr(ss), first, pool, margin, int(opponent), T, ptr(acts),
                            ptr(o["obs"]), ptr(o["mask"]), ptr(o["player"]), ptr(o["maskbits"]), ptr(o["reward"]), ptr(o["done"]),
                            ptr(o["status"]))
    assert ops > 0
    o.update(episodes=ep, stuck=stuck, stat_sum=ss)
    return o, ops


def check_rollout(L, first, pool, opponent, n, T, seed0, warm=0, margin=0, pos0=None):
    state, mt, pos = start_batch(n, seed0, first, pool, warm)
    if pos0 is not None:
        pos[:] = pos0                                        # any index 0..624 is a valid CPython state
    rng = np.random.default_rng(seed0)
    exp = []
    for g in range(n):
        exp.append(oracle_play(state[g].view(oz.RECORD_DTYPE)[0], mt[g], pos[g], T, first, pool, opponent,
                               lambda m, t: int(rng.choice(np.flatnonzero(m)))))
    actions = np.stack([np.array(e[0]["action"], np.int32) for e in exp], axis=1)
    got, ops = emulate(L, state, mt, pos, first, pool, opponent, actions, margin)
    episodes = 0
    for g, (e, rec, mt_e, idx) in enumerate(exp):
        tag = (first, pool, opponent, g)
        assert np.array_equal(got["mask"][:, g].astype(bool), np.array(e["mask"], bool)), tag
        assert set(np.unique(got["mask"][:, g])) <= {0, 1}, tag
        bits = got["maskbits"][:, g].view(np.uint8).reshape(T + 1, 24)[:, :23]
        assert np.array_equal(bits, np.packbits(np.array(e["mask"], bool), axis=1, bitorder="little")), tag
        assert np.array_equal(got["obs"][:, g].astype(np.int64), np.array(e["obs"])), tag
        assert np.array_equal(got["player"][:, g], np.array(e["player"])), tag
        assert np.array_equal(got["reward"][:, g], np.array(e["reward"])), tag
        assert np.array_equal(got["done"][:, g].astype(bool), np.array(e["done"])), tag
        assert state[g].tobytes() == rec.tobytes(), tag
        assert int(pos[g]) == idx, tag
        assert np.array_equal(mt[g], mt_e), tag
        assert int(got["episodes"][g]) == int(np.sum(e["done"])) and int(got["stuck"][g]) == 0 and int(got["status"][g]) == 0, tag
        episodes += int(np.sum(e["done"]))
    return ops, episodes


@pytest.mark.parametrize("ruleset", sorted(RULES))
def test_policy_self_play_env_side_under_emulation_equals_the_oracle(ruleset):
    """env_policy_step: the sampled action is played by whoever is to move; reward / done / auto-reset as NNRunner's loop sees them."""
    L = load()
    first, pool = RULES[ruleset]
    ops, episodes = check_rollout(L, first, pool, False, n=5, T=170, seed0=40, warm=2)
    assert episodes >= 5 and ops > 10000


@pytest.mark.parametrize("ruleset", sorted(RULES))
def test_game_runner_step_with_random_opponent_under_emulation_equals_the_oracle(ruleset):
    """env_agent_step: GameRunner.step (the opponent's RandomAgent moves until player 1 has a choice again) and GameRunner.reset
    with the opponent's opening at episode end."""
    L = load()
    first, pool = RULES[ruleset]
    ops, episodes = check_rollout(L, first, pool, True, n=5, T=90, seed0=140)
    assert episodes >= 5 and ops > 10000


@pytest.mark.parametrize("opponent", [False, True])
def test_rollout_env_across_an_mt19937_regeneration(opponent):
    """The streams start at indices 560 .. 624: the opponent's draws, a round's forty words or a reset's words straddle the
    regeneration of the 624-word state within the first moves (deal2's "read, regenerate, read" path and the window refill)."""
    L = load()
    n = 33
    check_rollout(L, 0, 1, opponent, n=n, T=45, seed0=2300, warm=1, pos0=np.array([560 + 2 * g for g in range(n)], np.uint32))


def test_factory_draw_fp64_path_in_the_rollout_env():
    L = load()
    check_rollout(L, 0, 1, True, n=2, T=60, seed0=77, margin=0x7fffffff)


def test_refused_actions_leave_the_game_untouched_and_the_sibling_plays_on():
    """Illegal and out-of-range actions (azul.py:298-302 raises before touching the game): status reported, record, MT19937 position
    and published state unchanged, no reward; the other game of the wave plays the oracle's game."""
    L = load()
    n, T = 2, 6
    state, mt, pos = start_batch(n, 900, 0, 1, warm=4)
    e1 = oracle_play(state[1].view(oz.RECORD_DTYPE)[0], mt[1], pos[1], T, 0, 1, False, lambda m, t: int(np.flatnonzero(m)[-1]))
    mask0 = np.asarray(oz.check_all_valid(oz.unpack(state[0].view(oz.RECORD_DTYPE)[0], oz.POOL_LID, oz.FIRST_RANDOM).game), bool)
    for bad, want in ((int(np.flatnonzero(~mask0)[0]), ST_ILLEGAL), (180, ST_BAD_ACTION), (-3, ST_BAD_ACTION)):
        st, m, p = state.copy(), mt.copy(), pos.copy()
        actions = np.stack([np.full(T, bad, np.int32), np.array(e1[0]["action"], np.int32)], axis=1)
        got, _ = emulate(L, st, m, p, 0, 1, False, actions)
        assert int(got["status"][0]) == want and int(got["status"][1]) == 0
        assert st[0].tobytes() == state[0].tobytes() and int(p[0]) == int(pos[0]) and np.array_equal(m[0], mt[0])
        assert not got["reward"][:, 0].any() and not got["done"][:, 0].any()
        assert (got["mask"][:, 0].astype(bool) == mask0).all() and (got["obs"][:, 0] == got["obs"][0, 0]).all()
        assert np.array_equal(got["reward"][:, 1], np.array(e1[0]["reward"])) and st[1].tobytes() == e1[1].tobytes()
        assert np.array_equal(got["obs"][:, 1].astype(np.int64), np.array(e1[0]["obs"]))


def test_stuck_slot_restarts_in_the_rollout_env():
    """Hazard H3: nothing legal (only the first-player token left) and the head hands over "no action" (-1): the slot restarts,
    done == 2, reward 0, status ST_STUCK, the stuck counter moves; the sibling half is not disturbed."""
    L = load()
    n, T = 2, 1
    state, mt, pos = start_batch(n, 950, 0, 1, warm=6)
    rec = state[1].view(oz.RECORD_DTYPE)[0].copy()
    rec["displays"][:] = 0
    rec["center"][:] = [0, 0, 0, 0, 0, 1]
    state[1] = np.frombuffer(rec.tobytes(), np.uint8)
    e0 = oracle_play(state[0].view(oz.RECORD_DTYPE)[0], mt[0], pos[0], T, 0, 1, False, lambda m, t: int(np.flatnonzero(m)[0]))
    actions = np.array([[e0[0]["action"][0], -1]], np.int32)
    got, _ = emulate(L, state, mt, pos, 0, 1, False, actions)
    assert not got["mask"][0, 1].any()
    assert int(got["done"][0, 1]) == 2 and int(got["reward"][0, 1]) == 0 and int(got["status"][1]) == ST_STUCK and int(got["stuck"][1]) == 1
    assert got["mask"][1, 1].any()                              # a fresh game was dealt
    fresh = state[1].view(oz.RECORD_DTYPE)[0]
    assert int(np.sum(fresh["displays"])) + int(np.sum(fresh["center"][:5])) == 20 and int(fresh["center"][5]) == 1
    assert state[0].tobytes() == e0[1].tobytes() and int(got["reward"][0, 0]) == e0[0]["reward"][0] and int(got["status"][0]) == 0


@pytest.mark.parametrize("opponent", [False, True])
def test_move_limit_in_the_rollout_env(opponent):
    """azul_batch_set_move_limit on the GameRunner / policy paths (apply_step2 -> ST_TRUNCATED; beyond the reference, off by default): a move of
    either side that ends a round without ending the game once the episode has played `limit` moves cuts the episode -- done = 3, the
    slot restarts (with the RandomAgent opponent: reward 0 and the opponent's opening), counted in `stuck` -- like the oracle's restatement."""
    L = load()
    L.sh2_set_move_limit.argtypes = [C.c_uint]
    n, T, limit = 4, 120, 24
    state, mt, pos = start_batch(n, 7300, 0, 1, 0)
    rng = np.random.default_rng(11)
    exp = [oracle_play(state[g].view(oz.RECORD_DTYPE)[0], mt[g], pos[g], T, 0, 1, opponent, lambda m, t: int(rng.choice(np.flatnonzero(m))), move_limit=limit)
           for g in range(n)]
    actions = np.stack([np.array(e[0]["action"], np.int32) for e in exp], axis=1)
    try:
        L.sh2_set_move_limit(limit)
        got, _ = emulate(L, state, mt, pos, 0, 1, opponent, actions)
    finally:
        L.sh2_set_move_limit(0)
    cuts = 0
    for g, (e, rec, mt_e, idx) in enumerate(exp):
        assert np.array_equal(got["done"][:, g], np.array(e["done"], np.uint8)), (g, got["done"][:, g], e["done"])
        assert np.array_equal(got["reward"][:, g], np.array(e["reward"])) and np.array_equal(got["mask"][:, g].astype(bool), np.array(e["mask"], bool)), g
        assert np.array_equal(got["obs"][:, g].astype(np.int64), np.array(e["obs"])), g
        assert state[g].tobytes() == rec.tobytes() and int(pos[g]) == idx and np.array_equal(mt[g], mt_e), g
        c = int((np.array(e["done"]) == 3).sum())
        assert int(got["stuck"][g]) == c and int(got["episodes"][g]) == int((np.array(e["done"]) == 1).sum()), g
        cuts += c
    assert cuts >= 4
