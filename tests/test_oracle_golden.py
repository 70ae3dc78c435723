"""Pins the oracle (oracle/azul_oracle.c) to vectors generated from the real reference.

tests/golden/*.npz were written by oracle/gen_golden.py, which imports /root/reference/azulnet and
records what the reference's own Azul / GameRunner / RandomAgent code did on seeded games.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as oz

RULESETS = {"lid_randomfirst": (oz.FIRST_RANDOM, oz.POOL_LID),
            "random_first1": (oz.FIRST_ABSENT, oz.POOL_RANDOM),
            "lid_first2": (2, oz.POOL_LID)}


@pytest.fixture(scope="module")
def pyrandom(golden_dir):
    return np.load(os.path.join(golden_dir, "pyrandom.npz"))


def test_mt19937_seed_state_and_outputs(pyrandom):
    L = oz.lib()
    for i, s in enumerate(pyrandom["seeds"]):
        r = oz.seeded_rng(int(s))
        st = pyrandom["state_%d" % i]
        assert np.array_equal(np.ctypeslib.as_array(r.mt), st[:624])
        assert r.idx == int(st[624])
        got = np.array([L.oz_rng_u32(C.byref(r)) for _ in range(1400)], dtype=np.uint32)
        assert np.array_equal(got, pyrandom["u32_%d" % i])
        r = oz.seeded_rng(int(s))
        got = np.array([L.oz_rng_random(C.byref(r)) for _ in range(700)])
        assert np.array_equal(got, pyrandom["random_%d" % i])          # bit-exact fp64
        r = oz.seeded_rng(int(s))
        got = np.array([L.oz_rng_randbelow(C.byref(r), 5) for _ in range(300)], dtype=np.uint8)
        assert np.array_equal(got, pyrandom["randbelow5_%d" % i])
        r = oz.seeded_rng(int(s))
        got = np.array([1 + L.oz_rng_randbelow(C.byref(r), 2) for _ in range(300)], dtype=np.uint8)
        assert np.array_equal(got, pyrandom["choice12_%d" % i])


def test_choices_tile_pool_weights(pyrandom):
    L = oz.lib()
    r = oz.seeded_rng(99)
    for box, pick in zip(pyrandom["choices_box"], pyrandom["choices_box_pick"]):
        total = int(box.sum())
        w = np.array([float(b) / float(total) for b in box], dtype=np.float64)
        got = L.oz_rng_choices(C.byref(r), w.ctypes.data_as(C.POINTER(C.c_double)), 5)
        assert got == int(pick)


def test_random_agent_choices(pyrandom):
    L = oz.lib()
    r = oz.seeded_rng(2024)
    for pm, pick in zip(pyrandom["choices_mask"], pyrandom["choices_mask_pick"]):
        mask = np.unpackbits(pm, bitorder="little")[:180].astype(np.uint8)
        got = L.oz_random_agent(mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(r))
        assert got == int(pick)
    assert r.idx == int(pyrandom["choices_mask_words_end"])


def _golden_record(t, k, i):
    """Golden env row i of stream k -> canonical record (player_score / move_counter filled by caller)."""
    rec = np.zeros((), dtype=oz.RECORD_DTYPE)
    g = lambda name: t["s%d_env_%s" % (k, name)][i]
    rec["displays"] = g("displays")
    rec["center"] = g("center")
    rec["flags"] = int(g("cur")) | (int(g("nfp")) << 3) | (int(g("eog_flag")) << 6)
    rec["pattern_lines"] = g("pattern_lines")
    rec["floors"] = g("floors")
    w = g("walls").reshape(2, 25).astype(np.uint32)
    rec["walls"] = (w << np.arange(25, dtype=np.uint32)).sum(axis=1)
    rec["score"] = g("score")
    rec["box"] = g("box")
    rec["lid"] = g("lid")
    rec["turn_counter"] = g("turn_counter")
    rec["first_player_stats"] = g("first_player_stats")
    rec["floor_penalty"] = g("floor_penalty")
    rec["max_combo"] = g("max_combo")
    rec["completed_lines"] = g("completed_lines")
    return rec


@pytest.mark.parametrize("ruleset", sorted(RULESETS))
def test_flat_stream_matches_reference(golden_dir, ruleset):
    """Every env move of every golden stream: mask, action, post-state, potential, done, RNG words."""
    t = np.load(os.path.join(golden_dir, "traj_%s.npz" % ruleset))
    fp, pool = RULESETS[ruleset]
    for k, seed in enumerate(t["seeds"]):
        n = len(t["s%d_env_action" % k])
        st = oz.Stream(int(seed), fp, pool)
        first = list(t["s%d_episode_first_env_index" % k]) + [n]
        out = st.advance(n)
        assert np.array_equal(out["action"], t["s%d_env_action" % k]), (ruleset, seed)
        gm = np.unpackbits(t["s%d_env_mask" % k], axis=1, bitorder="little")[:, :180]
        assert np.array_equal(out["mask"], gm)
        assert np.array_equal(out["done"].astype(bool), t["s%d_env_eog" % k])
        # reward = delta of the what-if potential, restarting at 0 with every episode
        phi = t["s%d_env_phi" % k].astype(np.int64)
        exp_r = np.empty(n, dtype=np.int64)
        for e in range(len(first) - 1):
            a, b = first[e], first[e + 1]
            exp_r[a:b] = np.diff(np.concatenate([[0], phi[a:b]]))
        assert np.array_equal(out["reward"], exp_r)
        # full post-state, incl. box/lid and the statistics counters
        ep = 0
        for i in range(n):
            while i >= first[ep + 1]:
                ep += 1
            exp = _golden_record(t, k, i)
            exp["player_score"] = phi[i]
            exp["move_counter"] = i - first[ep] + 1
            got = out["rec_after"][i]
            for name in oz.RECORD_DTYPE.names:
                assert np.array_equal(got[name], exp[name]), (ruleset, int(seed), i, name, got[name], exp[name])
        # the game ended exactly where the reference's episodes ended
        assert int(st.episodes.value) == 2 and int(st.stuck.value) == 0
        # RNG consumption: words consumed up to the last move (+ the auto-reset after it: 1 choice + 20 draws)
        words_last = int(t["s%d_env_rng_words" % k][-1])
        assert st.r.words >= words_last + (20 if pool == oz.POOL_RANDOM else 40)


@pytest.mark.parametrize("ruleset", ["lid_randomfirst", "random_first1"])
def test_runner_step_matches_reference(golden_dir, ruleset):
    """GameRunner.reset()/step() semantics: reward, done, obs, mask, move_counter, player_score."""
    L = oz.lib()
    t = np.load(os.path.join(golden_dir, "traj_%s.npz" % ruleset))
    fp, pool = RULESETS[ruleset]
    for k, seed in enumerate(t["seeds"]):
        r = oz.seeded_rng(int(seed))
        q = oz.Runner()
        assert L.oz_runner_init(C.byref(q), fp, pool, C.byref(r)) == 0
        acts = t["s%d_agent_action" % k]
        dones = t["s%d_agent_done" % k]
        j = 0
        for _ep in range(2):
            assert L.oz_runner_reset(C.byref(q), C.byref(r)) == 0
            done = False
            while not done:
                mask = oz.check_all_valid(q.game)
                gm = np.unpackbits(t["s%d_agent_mask_before" % k][j], bitorder="little")[:180].astype(bool)
                assert np.array_equal(mask, gm)
                assert np.array_equal(oz.get_state(q.game, 0), t["s%d_agent_obs_before" % k][j])
                m8 = mask.astype(np.uint8)
                a = L.oz_random_agent(m8.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(r))
                assert a == int(acts[j])
                rew, dn = C.c_int64(0), C.c_int(0)
                assert L.oz_runner_step(C.byref(q), a, C.byref(r), C.byref(rew), C.byref(dn)) == 0
                assert rew.value == int(t["s%d_agent_reward" % k][j])
                assert bool(dn.value) == bool(dones[j])
                assert q.move_counter == int(t["s%d_agent_move_counter" % k][j])
                assert q.player_score == int(t["s%d_agent_player_score" % k][j])
                done = bool(dn.value)
                j += 1
            stats = oz.get_statistics(q.game)
            exp = t["s%d_episode_stats" % k][_ep]
            assert np.array_equal(np.array([stats[key] for key in oz.STAT_KEYS]), exp)
        assert j == len(acts)


def test_observation_both_perspectives(golden_dir):
    t = np.load(os.path.join(golden_dir, "traj_lid_randomfirst.npz"))
    st = oz.Stream(int(t["seeds"][3]), oz.FIRST_RANDOM, oz.POOL_LID)
    n = len(t["s3_env_action"])
    out = st.advance(n)
    for i in range(n):
        q = oz.unpack(out["rec_after"][i], oz.POOL_LID, oz.FIRST_RANDOM)
        for persp in (0, 1):
            assert np.array_equal(oz.get_state(q.game, persp), t["s3_env_obs"][i][persp])


def test_board_fixtures_known_answers(golden_dir, resources_dir):
    import json
    b = np.load(os.path.join(golden_dir, "boards.npz"))
    L = oz.lib()
    for f in sorted(os.listdir(resources_dir)):
        key = f[:-5]
        data = json.load(open(os.path.join(resources_dir, f)))
        g = oz.Game()
        r = oz.seeded_rng(0)
        assert L.oz_init(C.byref(g), 2, oz.FIRST_ABSENT, oz.POOL_RANDOM, C.byref(r)) == 0
        g.arr("displays")[:] = data["game_board_displays"]
        g.arr("center")[:] = data["game_board_center"]
        g.arr("pattern_lines")[:2] = data["pattern_lines"]
        g.arr("walls")[:2] = data["walls"]
        g.arr("floors")[:2] = data["floors"]
        g.arr("score")[:2] = data["score"]
        g.current_player = data["current_player"]
        g.next_first_player = data["next_first_player"]
        g.turn_counter = data["turn_counter"]
        gm = np.unpackbits(b[key + "_mask"], bitorder="little")[:180].astype(bool)
        assert np.array_equal(oz.check_all_valid(g), gm), key
        for persp in (0, 1):
            assert np.array_equal(oz.get_state(g, persp), b[key + "_obs"][persp]), key
        assert bool(L.oz_is_end_of_round(C.byref(g))) == bool(b[key + "_eor"])
        assert bool(L.oz_is_end_of_game(C.byref(g))) == bool(b[key + "_eog"])
        L.oz_count_score(C.byref(g))
        assert np.array_equal(g.arr("score")[:2], b[key + "_scored_score"]), key
        assert np.array_equal(g.arr("walls")[:2], b[key + "_scored_walls"]), key
        assert np.array_equal(g.arr("pattern_lines")[:2], b[key + "_scored_pattern_lines"]), key
        stats = np.concatenate([g.arr("floor_penalty")[:2], g.arr("max_combo")[:2], g.arr("completed_lines")[:2].flatten()])
        assert np.array_equal(stats, b[key + "_scored_stats"]), key


def test_pack_unpack_roundtrip(golden_dir):
    st = oz.Stream(5)
    out = st.advance(80)
    for rec in out["rec_after"]:
        q = oz.unpack(rec, oz.POOL_LID, oz.FIRST_RANDOM)
        assert oz.pack(q).tobytes() == rec.tobytes()


def test_threaded_bench_is_deterministic():
    m1, c1 = oz.bench_selfplay(0, 16, 200, 1)
    m4, c4 = oz.bench_selfplay(0, 16, 200, 4)
    assert m1 == m4 == 16 * 200 and c1 == c4


@pytest.mark.parametrize("ruleset", ["lid_randomfirst", "random_first1"])
def test_game_runner_with_a_network_opponent_matches_reference(golden_dir, ruleset):
    """GameRunner(opponent=Agent(...)) of the reference (game_runner.py:27-30, 37-47, 84-85; scripts/run_batch.py:6-10), recorded by
    oracle/gen_golden.py gen_net_opponent: fed the opponent's recorded answers, the oracle must hand the opponent the recorded
    perspective-rotated observation and mask at every call -- incl. player 1's forced moves and the opening moves of reset() -- and give
    the agent the recorded observation, mask, reward, done, move_counter and player_score at every step."""
    t = np.load(os.path.join(golden_dir, "net_opponent.npz"))
    fp, pool = RULESETS[ruleset]
    for seed in t["seeds"]:
        pre = "%s_s%d_" % (ruleset, seed)
        cs, cm, ca, cp = t[pre + "call_state"], np.unpackbits(t[pre + "call_mask"], axis=1, bitorder="little")[:, :180], t[pre + "call_action"], t[pre + "call_player"]
        used = [0]

        def opponent(state, mask):
            i = used[0]
            used[0] += 1
            assert np.array_equal(state, cs[i]), (pre, i)
            assert np.array_equal(mask, cm[i].astype(bool)), (pre, i)
            assert int(run.q.game.current_player) == int(cp[i]) and int(run.q.move_counter) == int(t[pre + "call_move_counter"][i])
            return int(ca[i])

        run = oz.NetRunner(opponent, fp, pool, seed=int(seed))
        sm = np.unpackbits(t[pre + "step_mask"], axis=1, bitorder="little")[:, :180].astype(bool)
        first = list(t[pre + "step_episode_first_step"]) + [len(sm)]
        for e in range(2):
            assert run.reset() == 0
            assert used[0] == int(t[pre + "step_episode_calls_before"][e])          # (recorded after reset(): the opening moves included)
            for i in range(first[e], first[e + 1]):
                assert np.array_equal(run.get_state(), t[pre + "step_obs"][i]), (pre, i)
                assert np.array_equal(run.get_valid_moves(), sm[i]), (pre, i)
                rc, rew, dn = run.step(int(t[pre + "step_action"][i]))
                assert rc == 0 and rew == int(t[pre + "step_reward"][i]) and dn == bool(t[pre + "step_done"][i]), (pre, i)
                assert int(run.q.move_counter) == int(t[pre + "step_move_counter"][i]) and int(run.q.player_score) == int(t[pre + "step_player_score"][i])
                assert used[0] == int(t[pre + "step_calls_after"][i]), (pre, i)
            assert np.array_equal(run.get_state(), t[pre + "step_final_obs"][e])
            st = oz.get_statistics(run.q.game)
            assert np.allclose([st[k] for k in oz.STAT_KEYS], t[pre + "step_stats"][e])
        assert used[0] == len(ca)
    # the fixture exercises what makes this path different from RandomAgent: forced moves of player 1 go to the opponent, and it opens
    forced = sum(int((t["%s_s%d_call_player" % (ruleset, s)] == 1).sum()) for s in t["seeds"])
    assert forced >= 4
