"""The persistent policy rollout kernel ITSELF on CPU: azul_policy_rollout2_kernel (csrc/azul_rollout2.hpp -- env phases, matrix phases
on v_mfma_f32_16x16x4_f32 with weights streamed through buffer loads, the sampling head, trajectory slots written by the idle waves,
the fused returns scan), compiled UNMODIFIED by g++ and run as a workgroup of eight emulated wavefronts (tests/hostcheck/simt:
run_workgroup, s_barrier, MFMA and buffer-load emulation).  Checked per move and game:
  * the network: value, log-prob of the sampled action and the entropy term against a numpy fp32 forward of the same weights on the
    observation the kernel recorded (model.py:22-41, agent.py:64-72), within fp32 tolerance;
  * the sampled action is legal;
  * the env: the kernel's own actions replayed through the oracle give the recorded observations, masks, players, rewards, done flags,
    final records, MT19937 words and counters (game_runner.py:43-97, nn_runner.py:17-47);
  * the returns scan (nn_runner.py:70-76) and that no buffer load ever left its resource.
Under ASan / UBSan (tests/hostcheck/run_sanitizers.sh) every LDS and global index the kernel forms is checked as well."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as oz
from tests.test_hostcheck_env2 import RULES, oracle_play, ptr, start_batch

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck")


def load(name=None):
    name = name or os.environ.get("AZUL_SIMT_ROLLOUT_LIB", "libsimt_rollout2.so")
    subprocess.check_call(["make", "-s", "-C", HERE, name], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(HERE, name))
    L.sr2_rollout.restype = C.c_longlong
    L.sr2_rollout.argtypes = ([C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_int, C.c_uint] + [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 11
                              + [C.c_float, C.c_ulonglong, C.c_ulonglong])
    L.sr2_buffer_oob.restype = C.c_ulonglong
    return L


def weights(seed):
    rs = np.random.RandomState(seed)
    w = {"w1t": rs.randn(136, 360) * 0.08, "b1": rs.randn(360) * 0.05, "w2c": rs.randn(180) * 0.1, "b2c": rs.randn(1) * 0.1,
         "w2a_t": rs.randn(180, 180) * 0.12, "b2a": rs.randn(180) * 0.05}
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}


def forward(w, obs, mask):
    """model.py:22-41 on one observation, fp32; masked log-softmax and the entropy term of agent.py:64-72 / nn_runner.py:32-40."""
    h = np.maximum(obs.astype(np.float32) @ w["w1t"] + w["b1"], np.float32(0))
    value = np.float32(h[:180] @ w["w2c"] + w["b2c"][0])
    logits = (h[180:] @ w["w2a_t"] + w["b2a"]).astype(np.float64)
    legal = mask.astype(bool)
    z = logits[legal]
    lse = z.max() + np.log(np.exp(z - z.max()).sum())
    logp = np.full(180, -np.inf)
    logp[legal] = z - lse
    return float(value), logp, float(-logp[legal].mean())


def run(L, first, pool, opponent, n, T, seed0, warm=0, gamma=0.9, with_returns=True):
    state, mt, pos = start_batch(n, seed0, first, pool, warm)
    state0, mt0, pos0 = state.copy(), mt.copy(), pos.copy()
    ep, stuck, ss = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, 10))
    w = weights(seed0)
    o = {"obs": np.full((T + 1, n, 136), -99, np.float32), "mask": np.full((T + 1, n, 180), 0xEE, np.uint8),
         "player": np.full((T + 1, n), 9, np.uint8), "action": np.full((T, n), -7, np.int32), "reward": np.full((T, n), -7777, np.int32),
         "done": np.full((T, n), 9, np.uint8), "value": np.full((T, n), np.nan, np.float32), "logp": np.full((T, n), np.nan, np.float32),
         "entropy": np.full((T, n), np.nan, np.float32), "status": np.full(n, 99, np.uint8),
         "returns": np.full((T, n), np.nan, np.float32)}
    oob0 = L.sr2_buffer_oob()
    ops = L.sr2_rollout(n, ptr(state), ptr(mt), ptr(pos), ptr(ep), ptr(stuck), ptr(ss), first, pool, int(opponent), 1000,
                        ptr(w["w1t"]), ptr(w["b1"]), ptr(w["w2c"]), ptr(w["b2c"]), ptr(w["w2a_t"]), ptr(w["b2a"]), T,
                        ptr(o["obs"]), ptr(o["mask"]), ptr(o["player"]), ptr(o["action"]), ptr(o["reward"]), ptr(o["done"]),
                        ptr(o["value"]), ptr(o["logp"]), ptr(o["entropy"]), ptr(o["status"]), ptr(o["returns"]) if with_returns else None,
                        gamma, 4242, 17)
    assert ops > 0
    assert L.sr2_buffer_oob() == oob0                       # no weight fragment was ever requested outside its matrix
    episodes = 0
    for g in range(n):
        tag = (first, pool, opponent, g)
        # -- the network and the head, move by move, on what the kernel recorded
        for t in range(T):
            a = int(o["action"][t, g])
            mask = o["mask"][t, g]
            assert set(np.unique(mask)) <= {0, 1} and 0 <= a < 180 and mask[a] == 1, (tag, t)
            value, logp, ent = forward(w, o["obs"][t, g], mask)
            assert abs(o["value"][t, g] - value) < 2e-4 * max(1.0, abs(value)), (tag, t)
            assert abs(o["logp"][t, g] - logp[a]) < 2e-4, (tag, t)
            assert abs(o["entropy"][t, g] - ent) < 2e-4 * max(1.0, ent), (tag, t)
        # -- the env: the kernel's actions through the oracle
        acts = o["action"][:, g]
        e, rec, mt_e, idx = oracle_play(state0[g].view(oz.RECORD_DTYPE)[0], mt0[g], pos0[g], T, first, pool, opponent, lambda m, t: int(acts[t]))
        assert np.array_equal(o["mask"][:, g].astype(bool), np.array(e["mask"], bool)), tag
        assert np.array_equal(o["obs"][:, g].astype(np.int64), np.array(e["obs"])), tag
        assert np.array_equal(o["player"][:, g], np.array(e["player"])), tag
        assert np.array_equal(o["reward"][:, g], np.array(e["reward"])), tag
        assert np.array_equal(o["done"][:, g].astype(bool), np.array(e["done"])), tag
        assert state[g].tobytes() == rec.tobytes() and int(pos[g]) == idx and np.array_equal(mt[g], mt_e), tag
        assert int(ep[g]) == int(np.sum(e["done"])) and int(stuck[g]) == 0 and int(o["status"][g]) == 0, tag
        episodes += int(np.sum(e["done"]))
        if with_returns:
            q, want = np.float32(0), np.zeros(T, np.float32)
            for t in range(T - 1, -1, -1):
                q = np.float32(o["reward"][t, g]) + np.float32(gamma) * (np.float32(0) if o["done"][t, g] else q)
                want[t] = q
            assert np.array_equal(o["returns"][:, g], want), tag
    return ops, episodes


@pytest.mark.parametrize("ruleset", ["lid_randomfirst", "random_first1"])
def test_policy_on_both_sides(ruleset):
    L = load()
    first, pool = RULES[ruleset]
    ops, _ = run(L, first, pool, False, n=16, T=7, seed0=60, warm=30)
    assert ops > 5000


def test_random_agent_opponent_and_a_ragged_last_workgroup():
    L = load()
    first, pool = RULES["lid_randomfirst"]
    ops, _ = run(L, first, pool, True, n=19, T=5, seed0=80, warm=45)
    assert ops > 5000


# ---- the NETWORK opponent (azul_policy_rollout2_kernel<LID, 2>; game_runner.py:27-30 GameRunner(opponent=Agent(...))) ----------------------
def load_vs():
    L = load()
    L.sr2_rollout_vs.restype = C.c_longlong
    L.sr2_rollout_vs.argtypes = ([C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 11
                                 + [C.c_float, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong] + [C.c_void_p] * 3 + [C.c_int])
    return L


def to_agent_decision(rec, mt, pos, first, pool, rng):
    """Bring a flat self-play state to a point where GameRunner hands the turn to the agent (game_runner.py:46: player 1 to move with at
    least two legal moves, game not over) by letting a stand-in opponent play random legal moves."""
    run = oz.NetRunner(lambda s, m: int(rng.choice(np.flatnonzero(m))), first if first else oz.FIRST_RANDOM, pool, rec=rec, mt=mt, pos=pos)
    Lz = oz.lib()
    for _ in range(200):
        legal = int(run.get_valid_moves().sum())
        over = bool(Lz.oz_is_end_of_game(C.byref(run.q.game)))
        if over:
            assert run.reset() == 0
            continue
        if run.q.game.current_player == 1 and legal >= 2:
            break
        assert Lz.oz_runner_opponent_move_with(C.byref(run.q), C.byref(run.r), run._cb, None) == 0
    run.q.player_score = int(Lz.oz_potential(C.byref(run.q.game)))          # GameRunner.player_score follows the potential (game_runner.py:52)
    m, p = run.rng_state()
    return np.frombuffer(run.record().tobytes(), np.uint8).copy(), m, p


def actor_forward(w, obs, mask):
    h = np.maximum(obs.astype(np.float32) @ w["w1t"][:, 180:] + w["b1"][180:], np.float32(0))
    logits = (h @ w["w2a_t"] + w["b2a"]).astype(np.float64)
    legal = mask.astype(bool)
    z = logits[legal]
    lse = z.max() + np.log(np.exp(z - z.max()).sum())
    logp = np.full(180, -np.inf)
    logp[legal] = z - lse
    return logp


def run_vs(L, first, pool, n, T, seed0, warm, slots=8, argmax=False, move_limit=0):
    state, mt, pos = start_batch(n, seed0, first, pool, warm)
    rng = np.random.default_rng(seed0)
    for g in range(n):
        state[g], mt[g], pos[g] = to_agent_decision(state[g].view(oz.RECORD_DTYPE)[0], mt[g], pos[g], first, pool, rng)
    state0, mt0, pos0 = state.copy(), mt.copy(), pos.copy()
    ep, stuck, ss = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros((n, 10))
    wa, wo = weights(seed0), weights(seed0 + 1)
    keys = ("w1t", "b1", "w2c", "b2c", "w2a_t", "b2a")
    pa = (C.c_void_p * 6)(*[wa[k].ctypes.data for k in keys])
    po = (C.c_void_p * 6)(*[wo[k].ctypes.data for k in keys])
    o = {"obs": np.full((T + 1, n, 136), -99, np.float32), "mask": np.full((T + 1, n, 180), 0xEE, np.uint8),
         "player": np.full((T + 1, n), 9, np.uint8), "action": np.full((T, n), -7, np.int32), "reward": np.full((T, n), -7777, np.int32),
         "done": np.full((T, n), 9, np.uint8), "value": np.full((T, n), np.nan, np.float32), "logp": np.full((T, n), np.nan, np.float32),
         "entropy": np.full((T, n), np.nan, np.float32), "status": np.full(n, 99, np.uint8), "returns": np.full((T, n), np.nan, np.float32),
         "opp_action": np.full((T, slots, n), -9, np.int32), "opp_logp": np.full((T, slots, n), np.nan, np.float32),
         "opp_replies": np.full((T, n), 200, np.uint8)}
    oob0 = L.sr2_buffer_oob()
    AM = 0xFFFFFFFFFFFFFFFF
    L.sr2_set_move_limit.argtypes = [C.c_uint]
    L.sr2_set_move_limit(move_limit)
    ops = L.sr2_rollout_vs(n, ptr(state), ptr(mt), ptr(pos), ptr(ep), ptr(stuck), ptr(ss), first, pool, 1000, C.cast(pa, C.c_void_p), C.cast(po, C.c_void_p), T,
                           ptr(o["obs"]), ptr(o["mask"]), ptr(o["player"]), ptr(o["action"]), ptr(o["reward"]), ptr(o["done"]), ptr(o["value"]),
                           ptr(o["logp"]), ptr(o["entropy"]), ptr(o["status"]), ptr(o["returns"]), 0.9, 4242, AM if argmax else 777, 17,
                           ptr(o["opp_action"]), ptr(o["opp_logp"]), ptr(o["opp_replies"]), slots)
    L.sr2_set_move_limit(0)
    assert ops > 0 and L.sr2_buffer_oob() == oob0
    episodes = forced = opening = cuts = 0
    for g in range(n):
        tag = (first, pool, g)
        assert int(o["opp_replies"][:, g].max()) <= slots, "raise `slots`: a step had more replies than the trace holds"
        cur = {"t": 0, "j": 0}
        handed = []

        def opponent(s, m):
            t, j = cur["t"], cur["j"]
            a = int(o["opp_action"][t, j, g])
            assert 0 <= a < 180 and m[a], (tag, t, j, a)
            handed.append((t, j, s.copy(), m.copy(), int(run.q.game.current_player)))
            cur["j"] += 1
            return a

        run = oz.NetRunner(opponent, first if first else oz.FIRST_RANDOM, pool, rec=state0[g].view(oz.RECORD_DTYPE)[0], mt=mt0[g], pos=pos0[g])
        for t in range(T):
            cur["t"], cur["j"] = t, 0
            mask = run.get_valid_moves()
            assert np.array_equal(o["mask"][t, g].astype(bool), mask), (tag, t)
            assert np.array_equal(o["obs"][t, g].astype(np.int64), run.get_state(0)), (tag, t)
            assert int(o["player"][t, g]) == 1 == int(run.q.game.current_player) and mask.sum() >= 2, (tag, t)
            a = int(o["action"][t, g])
            assert mask[a]
            value, logp, ent = forward(wa, o["obs"][t, g], o["mask"][t, g])
            assert abs(o["value"][t, g] - value) < 2e-4 * max(1.0, abs(value)) and abs(o["logp"][t, g] - logp[a]) < 2e-4, (tag, t)
            assert abs(o["entropy"][t, g] - ent) < 2e-4 * max(1.0, ent), (tag, t)
            rc, rew, dn = run.step(a, move_limit)                           # game_runner.py:43-55 with the recorded opponent
            assert rc == 0 and rew == int(o["reward"][t, g]) and int(dn) == int(o["done"][t, g]), (tag, t, rew, dn, int(o["done"][t, g]))
            cuts += int(dn) == 3
            if dn:
                episodes += 1
                before = cur["j"]
                assert run.reset() == 0                                     # nn_runner.py:20: the next episode, the opponent opens (:84-85)
                opening += cur["j"] - before
            assert cur["j"] == int(o["opp_replies"][t, g]), (tag, t, cur["j"], int(o["opp_replies"][t, g]))
        assert np.array_equal(o["mask"][T, g].astype(bool), run.get_valid_moves()) and np.array_equal(o["obs"][T, g].astype(np.int64), run.get_state(0)), tag
        m_e, idx = run.rng_state()
        assert state[g].tobytes() == run.record().tobytes() and int(pos[g]) == idx and np.array_equal(mt[g], m_e), tag
        assert int(ep[g]) == int((o["done"][:, g] == 1).sum()) and int(stuck[g]) == int((o["done"][:, g] == 3).sum()), tag
        assert int(o["status"][g]) == (6 if int(o["done"][T - 1, g]) == 3 else 0), tag
        # the opponent saw the MOVER's perspective and sampled from ITS net: log-prob of every answer against a numpy forward_actor
        for t, j, s, m, player in handed:
            lp = actor_forward(wo, s, m)
            a = int(o["opp_action"][t, j, g])
            assert abs(o["opp_logp"][t, j, g] - lp[a]) < 2e-4, (tag, t, j)
            if argmax:
                assert a == int(np.argmax(np.where(m, lp, -np.inf))) or abs(lp[a] - lp[m].max()) < 1e-5, (tag, t, j)
            forced += player == 1
        q, want = np.float32(0), np.zeros(T, np.float32)
        for t in range(T - 1, -1, -1):
            q = np.float32(o["reward"][t, g]) + np.float32(0.9) * (np.float32(0) if o["done"][t, g] else q)
            want[t] = q
        assert np.array_equal(o["returns"][:, g], want), tag
    return (ops, episodes, forced, opening, cuts) if move_limit else (ops, episodes, forced, opening)


def test_network_opponent_with_a_move_limit():
    """The move limit (beyond the reference, off by default) on the network-opponent protocol: a move of either side that ends a round
    without ending the game once the episode has played `limit` moves cuts the episode (done = 3, reward 0, slot restarted and opened)."""
    L = load_vs()
    first, pool = RULES["lid_randomfirst"]
    ops, episodes, forced, opening, cuts = run_vs(L, first, pool, n=16, T=24, seed0=900, warm=20, move_limit=22)
    assert cuts >= 6


def test_network_opponent_in_the_rollout_kernel_replays_through_the_oracle():
    """GameRunner(opponent=Agent(...)) inside the kernel: the agent's records as with the RandomAgent opponent, every opponent_move() --
    replies, player 1's forced moves, the opening moves after an episode end -- answered by the second net on the mover's perspective."""
    L = load_vs()
    first, pool = RULES["lid_randomfirst"]
    ops, episodes, forced, opening = run_vs(L, first, pool, n=19, T=6, seed0=300, warm=40)        # (a ragged last workgroup)
    assert ops > 5000 and episodes >= 1 and forced >= 1
    ops, episodes, forced, opening = run_vs(L, first, pool, n=16, T=30, seed0=500, warm=50)
    assert episodes >= 10 and forced >= 8 and opening >= 3


def test_network_opponent_argmax_and_the_random_pool():
    L = load_vs()
    first, pool = RULES["random_first1"]
    run_vs(L, first, pool, n=16, T=5, seed0=410, warm=30, argmax=True)
