"""The MOVE LIMIT (azul_batch_set_move_limit; beyond the reference, off by default -- "parity unpinned", restated in the oracle as
oz_step_limited / oz_stream_advance_limited / oz_runner_step_limited) on the GPU.

Why it exists: under the reference's rules a game can reach a state from which it never ends (GameRunner's callers loop `while not done`,
/root/reference/azulnet/nn_runner.py:24): e.g. the game seeded 801 of the benchmark batch after ~310,000 moves -- all 20 tiles of one colour
locked in pattern lines that can no longer be completed, so no wall row can ever fill (azul.py:184-191).  It keeps its slot for ever and
slows its wavefront (bench.py: sustained.never_ending_games)."""
import numpy as np
import pytest
import torch

from oracle import oracle as oz

pytestmark = pytest.mark.gpu


def _start(env, seed_base):
    env.seed(seed_base)
    env.runner_init()
    env.runner_init()


@pytest.mark.parametrize("chunks", [(400,), (64, 200, 136)])
def test_selfplay_with_a_move_limit_equals_the_oracle(chunks):
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    n, base, limit = 128, 9100, 40
    env = BatchedAzul(n)
    _start(env, base)
    env.set_move_limit(limit)
    act, rew, dn, msk = [], [], [], []
    for T in chunks:                                     # a trajectory does not depend on how the moves are split over launches
        t = env.alloc_trajectory(T)
        env.selfplay(T, t["mask"], t["action"], t["reward"], t["done"])
        torch.cuda.synchronize()
        act.append(t["action"].cpu().numpy()); rew.append(t["reward"].cpu().numpy()); dn.append(t["done"].cpu().numpy()); msk.append(t["mask"].cpu().numpy())
    act, rew, dn, msk = (np.concatenate(x) for x in (act, rew, dn, msk))
    final, cnt = env.get_records(), env.counters()
    cuts = 0
    for g in range(n):
        s = oz.Stream(base + g)
        o = s.advance(sum(chunks), want_records=False, move_limit=limit)
        assert np.array_equal(o["action"], act[:, g]) and np.array_equal(o["reward"], rew[:, g]) and np.array_equal(o["done"], dn[:, g]), g
        assert np.array_equal(o["mask"], msk[:, g]), g
        assert s.record().tobytes() == final[g].tobytes() and env.get_rng(g)[1] == s.rng_state()[1] if g % 8 == 0 else True, g
        assert int(cnt["episodes"][g]) == int(s.episodes.value) and int(cnt["stuck"][g]) == int(s.stuck.value) == int((o["done"] == 3).sum()), g
        cuts += int((o["done"] == 3).sum())
    assert cuts > n                                      # with a limit of 40 moves most episodes are cut (a game lasts ~57)
    # off again: the reference's behaviour from here on
    env.set_move_limit(0)
    t = env.alloc_trajectory(300)
    env.selfplay(300, t["mask"], t["action"], t["reward"], t["done"])
    torch.cuda.synchronize()
    assert int((t["done"] == 3).sum()) == 0


def test_the_never_ending_game_of_the_benchmark_batch_is_cut():
    """Game 801 of the seed-0 batch stops ending after ~310 k moves (turn counter in the thousands, scores 0 : 0); with a limit of 400 moves
    per episode no game of its neighbourhood is ever more than a few rounds past the limit, and episodes keep finishing."""
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    n, base, T, launches = 64, 768, 512, 680
    out = {}
    for limit in (0, 400):
        env = BatchedAzul(n)
        _start(env, base)
        env.set_move_limit(limit)
        for _ in range(launches):
            env.selfplay(T)
        torch.cuda.synchronize()
        recs, cnt = env.get_records(), env.counters()
        out[limit] = (recs["turn_counter"].astype(int), recs["move_counter"].astype(int), cnt["episodes"].astype(np.int64), cnt["stuck"].astype(np.int64))
    turns0, moves0, ep0, _ = out[0]
    g = 801 - base
    assert turns0[g] > 1000 and turns0.argmax() == g and (np.delete(turns0, g) < 40).all()            # the reference's behaviour: it never ends
    turns1, moves1, ep1, cut1 = out[400]
    assert (turns1 < 80).all() and (moves1 < 400 + 60).all() and int(cut1[g]) >= 1                   # cut once it has played 400 moves ...
    assert int(ep1[g]) > int(ep0[g]) + 300 and int(cut1.sum()) <= 3                                   # ... then its slot finishes games again; ordinary games never reach the limit


def test_the_never_ending_game_stays_bit_exact_while_every_move_is_a_floor_move():
    """Game 801 again, without a limit: after ~310 k moves every one of its decisions is a draw among 0.01-weight floor moves only (M = 0), i.e.
    the branch of the RandomAgent sampler the one-compare path never takes (sample_slow2's exact floor guess; game_runner.py:87-97).  The oracle
    replays games 800..803 move for move for 680 launches: the last launch's masks / actions / rewards / done flags, the records and the RNG
    positions are identical, and game 801's last 512 actions are all floor moves."""
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    n, base, T, launches = 4, 800, 512, 680
    env = BatchedAzul(n)
    _start(env, base)
    for _ in range(launches - 1):
        env.selfplay(T)
    t = env.alloc_trajectory(T)
    env.selfplay(T, t["mask"], t["action"], t["reward"], t["done"])
    torch.cuda.synchronize()
    final = env.get_records()
    act = t["action"].cpu().numpy()
    assert (act[:, 1] < 30).all() and int(final["turn_counter"][1]) > 1000          # game 801: floor moves only, thousands of rounds into one episode
    for g in range(n):
        s = oz.Stream(base + g)
        s.advance((launches - 1) * T, want_records=False)
        o = s.advance(T, want_records=False)
        assert np.array_equal(o["action"], act[:, g]) and np.array_equal(o["mask"], t["mask"].cpu().numpy()[:, g]), g
        assert np.array_equal(o["reward"], t["reward"].cpu().numpy()[:, g]) and np.array_equal(o["done"], t["done"].cpu().numpy()[:, g]), g
        for p in range(2):       # the record keeps an episode's floor-penalty statistic in 16 bits: game 801's has wrapped (-38,936 points in one episode)
            fp = int(s.q.game.floor_penalty[p])
            assert (g == 1) == (fp < -32768), (g, fp)
            s.q.game.floor_penalty[p] = float((fp + 32768) % 65536 - 32768)
        assert s.record().tobytes() == final[g].tobytes() and env.get_rng(g)[1] == s.rng_state()[1], g


@pytest.mark.parametrize("opponent", [None, "random", "net"])
def test_rollout_structures_agree_under_a_move_limit(golden_dir, opponent):
    """The window kernel and the per-move / per-cut path with a limit of 36 moves: every trajectory array (done codes 0 / 1 / 3), records, RNG
    positions and counters bit for bit."""
    import os
    from azul_deep_reinforcement_learning_amd import PolicyRollout
    from azul_deep_reinforcement_learning_amd.policy import BatchedActorCritic
    c = np.load(os.path.join(golden_dir, "net_opponent.npz"))
    net = lambda who: BatchedActorCritic.from_reference({k[len(who) + 4:]: torch.from_numpy(c[k]) for k in c.files if k.startswith(who + "_sd_")}).cuda()
    runs = []
    for persistent in (False, True):
        opp = net("opp") if opponent == "net" else opponent
        ro = PolicyRollout(net("agent"), n_games=80, parts=1, seed_base=31, window=40, use_graph=False, opponent=opp, persistent=persistent)
        ro.envs[0].set_move_limit(36)
        wins = []
        for _ in range(3):
            tr = ro.run_window()
            ro.synchronize()
            wins.append({k: v.clone() for k, v in tr[0].items() if k not in ("opp_action", "opp_logp")})
        runs.append((wins, ro.envs[0].get_records(), ro.envs[0].get_rng_range()[1], ro.counters()))
    (wa, ra, pa, ca), (wb, rb, pb, cb) = runs
    for wi in range(3):
        for key in wa[wi]:
            assert torch.equal(wa[wi][key], wb[wi][key]), (wi, key)
    assert ra.tobytes() == rb.tobytes() and np.array_equal(pa, pb) and ca == cb
    done = torch.cat([w["done"] for w in wa])
    assert int((done == 3).sum()) > 20 and ca["stuck"] == int((done == 3).sum())
