"""GameRunner(opponent=Agent(...)) on the device (SURVEY.md 8b `.opponent`; /root/reference/azulnet/game_runner.py:27-30, 37-47, 84-85;
scripts/run_batch.py:6-10; tests/test_nn_runner.py:63-67, 84-90): the batched rollout with a NETWORK opponent.
  * the per-cut ABI (azul_batch_net_step_begin / _reply / _reset_begin) fed the REFERENCE's recorded agent and opponent actions gives the
    reference's recorded observations, masks, rewards, dones, move counters -- and hands the opponent the recorded perspective-rotated
    observation and mask at every call (tests/golden/net_opponent.npz, generated from the real reference);
  * both rollout structures replay through the oracle with their own recorded actions, the opponent's log-probabilities are the second
    net's (torch f32, 2e-5), and the window kernel equals the per-move path bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as oz
from tests.net_replay import replay_game

pytestmark = pytest.mark.gpu
RULES = {"lid_randomfirst": ({"first_player": "Random", "tile_pool": "Lid"}, oz.FIRST_RANDOM, oz.POOL_LID),
         "random_first1": ({"first_player": 1, "tile_pool": "Random"}, 1, oz.POOL_RANDOM)}


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "net_opponent.npz"))


def _nets(golden, device="cuda"):
    from azul_deep_reinforcement_learning_amd.policy import BatchedActorCritic
    out = []
    for who in ("agent", "opp"):
        sd = {k[len(who) + 4:]: torch.from_numpy(golden[k]) for k in golden.files if k.startswith(who + "_sd_")}
        out.append(BatchedActorCritic.from_reference(sd).to(device))
    return out


@pytest.mark.parametrize("ruleset", ["lid_randomfirst", "random_first1"])
def test_cut_protocol_reproduces_the_reference_game_runner_with_an_agent_opponent(golden, ruleset):
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    t = golden
    rules = {"lid_randomfirst": {"first_player": "Random", "tile_pool": "Lid"}, "random_first1": {}}[ruleset]
    forced = 0
    for seed in t["seeds"]:
        pre = "%s_s%d_" % (ruleset, seed)
        cs, ca = t[pre + "call_state"], t[pre + "call_action"]
        cm = np.unpackbits(t[pre + "call_mask"], axis=1, bitorder="little")[:, :180]
        sm = np.unpackbits(t[pre + "step_mask"], axis=1, bitorder="little")[:, :180]
        env = BatchedAzul(1, rules=rules, device="cuda")
        env.seed(int(seed))                                         # random.seed(seed)
        env.runner_init()                                           # GameRunner(opponent=opponent, rules=rules)
        net = env.net_state()
        reward = torch.zeros(1, dtype=torch.int32, device="cuda")
        done = torch.zeros(1, dtype=torch.uint8, device="cuda")
        status = torch.zeros(1, dtype=torch.uint8, device="cuda")
        act = torch.zeros(1, dtype=torch.int32, device="cuda")
        used = 0

        def replies(may_run_out=False):
            nonlocal used, forced
            while int(net["owing"].item()) > 0:
                if may_run_out and used == len(ca):
                    return                                          # (the reference stopped after two episodes: no third opening on record)
                assert int(net["pending"][0]) in (1, 2)
                assert np.array_equal(net["obs"][0].cpu().numpy().astype(np.int64), cs[used]), (pre, used)       # game_runner.py:38
                assert np.array_equal(net["mask"][0].cpu().numpy(), cm[used]), (pre, used)                       # :39
                forced += int(t[pre + "call_player"][used]) == 1
                act[0] = int(ca[used])
                used += 1
                env.net_step_reply(act, net, reward, done, status)
            assert int(net["pending"][0]) == 0

        env.net_reset_begin(net, status)                            # runner.reset() of the first run_episode (nn_runner.py:20)
        replies()
        assert used == int(t[pre + "step_episode_calls_before"][0])
        n_steps = len(sm)
        second = int(t[pre + "step_episode_first_step"][1])
        for i in range(n_steps):
            obs, mask, player = env.observe_all(0)
            assert np.array_equal(obs[0].cpu().numpy().astype(np.int64), t[pre + "step_obs"][i]), (pre, i)
            assert np.array_equal(mask[0].cpu().numpy(), sm[i]) and int(player[0]) == 1, (pre, i)
            act[0] = int(t[pre + "step_action"][i])
            env.net_step_begin(act, net, reward, done, status)
            replies(may_run_out=i == n_steps - 1)
            assert int(reward[0]) == int(t[pre + "step_reward"][i]) and bool(done[0]) == bool(t[pre + "step_done"][i]), (pre, i)
            assert int(status[0]) == 0, (pre, i)
            if bool(done[0]):
                # the launch that closes an episode's last step also opens the next one (nn_runner.py:20 -> game_runner.py:76-85): the
                # reference's second reset() and its opening calls have been played
                if i + 1 == second:
                    assert used == int(t[pre + "step_episode_calls_before"][1]), (pre, i)
            else:
                assert used == int(t[pre + "step_calls_after"][i]), (pre, i)
                rec = env.get_records()[0]
                assert int(rec["move_counter"]) == int(t[pre + "step_move_counter"][i]) and int(rec["player_score"]) == int(t[pre + "step_player_score"][i])
        assert used == len(ca)
    assert forced >= 4


def _windows(ro, k):
    wins = []
    for _ in range(k):
        tr = ro.run_window()
        ro.synchronize()
        wins.append({key: v.cpu().numpy().copy() for key, v in tr[0].items()})
    return wins


def _check_through_the_oracle(wins, start_recs, start_rng, finals, final_rng, first, pool, opp_net, games, T):
    """Env side of a recorded rollout vs the oracle, game by game; the opponent's log-probs vs its net in torch f32."""
    cat = lambda key, sl: np.concatenate([w[key][sl] for w in wins])
    handed_all = []
    for g in games:
        mt, pos = start_rng[g]
        obs = np.concatenate([w["obs"][:T, g] for w in wins] + [wins[-1]["obs"][T:T + 1, g]])
        mask = np.concatenate([w["mask"][:T, g] for w in wins] + [wins[-1]["mask"][T:T + 1, g]])
        player = np.concatenate([w["player"][:T, g] for w in wins] + [wins[-1]["player"][T:T + 1, g]])
        run, handed = replay_game(start_recs[g], mt, pos, first, pool, cat("action", (slice(None), g)), cat("opp_action", (slice(None), slice(None), g)),
                                  cat("opp_replies", (slice(None), g)), obs, mask, player, cat("reward", (slice(None), g)), cat("done", (slice(None), g)))
        assert run.record().tobytes() == finals[g].tobytes(), g
        m_e, idx = run.rng_state()
        assert int(final_rng[1][g]) == idx and np.array_equal(final_rng[0][g], m_e), g
        lp_all = cat("opp_logp", (slice(None), slice(None), g))
        handed_all += [(s, m, int(cat("opp_action", (slice(None), slice(None), g))[t, j]), float(lp_all[t, j]), pl) for t, j, s, m, pl in handed]
    # the opponent sampled from ITS net on the MOVER's perspective: log-prob of every answer against torch f32
    S = torch.from_numpy(np.stack([h[0] for h in handed_all]).astype(np.float32)).cuda()
    M = torch.from_numpy(np.stack([h[1] for h in handed_all])).cuda()
    with torch.no_grad():
        _, logp = opp_net.forward_actor(S, M)
    a = torch.tensor([h[2] for h in handed_all], device="cuda")
    ref = logp.gather(1, a.unsqueeze(1)).squeeze(1).cpu().numpy()
    got = np.array([h[3] for h in handed_all], np.float32)
    assert np.allclose(got, ref, atol=2e-5, rtol=1e-5), float(np.abs(got - ref).max())
    return len(handed_all), sum(h[4] == 1 for h in handed_all)


@pytest.mark.parametrize("persistent", [False, True])
@pytest.mark.parametrize("ruleset", ["lid_randomfirst", "random_first1"])
def test_rollout_against_a_network_opponent_replays_through_the_oracle(golden, persistent, ruleset):
    from azul_deep_reinforcement_learning_amd import PolicyRollout
    agent, opp = _nets(golden)
    rules, first, pool = RULES[ruleset]
    T, n = 40, 64
    ro = PolicyRollout(agent, n_games=n, parts=1, seed_base=500, window=T, use_graph=False, opponent=opp, persistent=persistent, opponent_trace=12,
                       rules=rules)
    assert ro.persistent == persistent and ro.opponent == "net"
    env = ro.envs[0]
    start = env.get_records()
    rng0 = [env.get_rng(g) for g in range(n)]
    wins = _windows(ro, 2)
    calls, forced = _check_through_the_oracle(wins, start, rng0, env.get_records(), env.get_rng_range(), first, pool, opp, range(0, n, 3), T)
    assert calls > 1500 and forced >= 10
    # the agent's own records: value / log-prob / entropy are the FIRST net's on the recorded observation
    w = wins[0]
    obs = torch.from_numpy(w["obs"][:T].reshape(-1, 136)).cuda()
    msk = torch.from_numpy(w["mask"][:T].reshape(-1, 180)).cuda().bool()
    with torch.no_grad():
        v = agent.forward_critic(obs).squeeze(1).cpu().numpy()
        _, lp = agent.forward_actor(obs, msk)
    a = torch.from_numpy(w["action"].reshape(-1).astype(np.int64)).cuda()
    assert np.allclose(w["value"].reshape(-1), v, atol=2e-5, rtol=1e-5)
    assert np.allclose(w["log_prob"].reshape(-1), lp.gather(1, a.unsqueeze(1)).squeeze(1).cpu().numpy(), atol=2e-5, rtol=1e-5)
    assert ro.counters()["episodes"] >= 40 and ro.counters()["stuck"] == 0


@pytest.mark.parametrize("n,rules,selection", [(100, None, "Distribution"), (16, None, "Max"), (37, {"first_player": 2, "tile_pool": "Random"}, "Distribution"),
                                               (2048, None, "Distribution"), (1, None, "Distribution")])
def test_window_kernel_with_a_network_opponent_is_bit_identical_to_the_cut_protocol(golden, n, rules, selection):
    """azul_batch_policy_rollout_vs (matrix phases on the second weight set inside the window kernel) against the per-cut path (azul_policy_forward
    on the opponent's weights + azul_batch_net_step_*): every trajectory array, the opponent's answers and their log-probs, the final records,
    RNG states, counters and the Philox step counter -- over two windows, with a ragged last workgroup (n = 100)."""
    from azul_deep_reinforcement_learning_amd import PolicyRollout
    T = 33
    runs = []
    for persistent in (False, True):
        agent, opp = _nets(golden)
        kw = {} if rules is None else {"rules": rules}
        ro = PolicyRollout(agent, n_games=n, parts=1, seed_base=77, window=T, use_graph=False, opponent=opp, persistent=persistent, opponent_trace=10,
                           opponent_selection=selection, **kw)
        wins = _windows(ro, 2)
        mt, pos = ro.envs[0].get_rng_range()
        runs.append((wins, ro.envs[0].get_records(), mt, pos, ro.counters(), ro.work[0]["counter"].tolist()))
    (wa, ra, ma, pa, ca, cta), (wb, rb, mb, pb, cb, ctb) = runs
    for wi in range(2):
        for key in ("obs", "mask", "player", "action", "reward", "done", "value", "log_prob", "entropy", "returns", "opp_replies"):
            assert np.array_equal(wa[wi][key], wb[wi][key]), (wi, key)
        rep = wa[wi]["opp_replies"]
        assert int(rep.max()) <= 10
        valid = np.arange(10)[None, :, None] < rep[:, None, :]              # [T][R][N]: slot j of a step holds a reply
        assert np.array_equal(wa[wi]["opp_action"][valid], wb[wi]["opp_action"][valid]), wi
        assert np.array_equal(wa[wi]["opp_logp"][valid], wb[wi]["opp_logp"][valid]), wi
        assert valid.sum() > T * n // 2 or n == 1
    assert ra.tobytes() == rb.tobytes() and np.array_equal(ma, mb) and np.array_equal(pa, pb) and ca == cb and cta == ctb
    assert ca["episodes"] > 0


def test_training_against_a_frozen_past_self(golden):
    """BatchedTrainer(opponent="self"): the rollout kernel plays the policy being trained against a frozen COPY of it (the second weight set
    of azul_batch_policy_rollout_vs), replaced by the current policy every `opponent_refresh` updates; the A2C update trains on the agent's
    C1 records exactly as with the RandomAgent opponent."""
    from azul_deep_reinforcement_learning_amd.training import BatchedTrainer
    agent, _ = _nets(golden)
    tr = BatchedTrainer(agent, n_games=256, window=16, opponent="self", opponent_refresh=3, move_limit=300, seed_base=5)
    ro = tr.rollout
    assert ro.opponent == "net" and ro.persistent
    frozen = ro.ow1t.clone()
    rows = [tr.run_batch() for _ in range(2)]
    torch.cuda.synchronize()
    assert torch.equal(ro.ow1t, frozen) and not torch.equal(ro.ow1t, ro.w1t)           # the policy moved, its past self did not
    rows.append(tr.run_batch())                                                        # third update: the opponent catches up
    torch.cuda.synchronize()
    assert torch.equal(ro.ow1t, ro.w1t) and torch.equal(ro.ow2a_t, ro.w2a_t)
    for _ in range(6):
        rows.append(tr.run_batch())
    assert all(np.isfinite(r["ac_loss"]) for r in rows) and tr.learner.updates == 9
    assert ro.counters()["episodes"] > 0


def test_network_opponent_structures_agree_from_unusual_states(golden):
    """The paths ordinary play rarely reaches, on purpose, with the NETWORK opponent: games handed in without any tile on the table (nothing
    legal: the head answers -1 -> BAD_ACTION for the agent, the slot is left as it is, game_runner.py:44), games whose record says "ended"
    (GameEnded -> done, slot restarted, the opponent opens the next episode if it starts), games whose wall already holds a complete row
    while the flag is clear (is_end_of_game() reads the walls, azul.py:184-191), and games in which it is the OPPONENT's turn when the agent is
    asked (its action is played for player 2, then the protocol carries on).  The window kernel must do what the per-cut path does: every
    trajectory array, the opponent's answers, final records, RNG positions, episode and stuck counters."""
    from azul_deep_reinforcement_learning_amd import PolicyRollout
    T, n = 12, 64
    runs = []
    for persistent in (False, True):
        agent, opp = _nets(golden)
        ro = PolicyRollout(agent, n_games=n, parts=1, seed_base=4242, window=T, use_graph=False, opponent=opp, persistent=persistent, opponent_trace=12)
        env = ro.envs[0]
        recs = env.get_records()
        raw = recs.view(np.uint8).reshape(n, -1).copy()
        for g in (3, 20, 41):
            raw[g, 0:31] = 0                                   # displays, centre and token empty
        for g in (5, 33):
            raw[g, 31] |= 0x40                                 # the record's "ended" flag
        for g in (7, 40, 63):
            raw[g, 84] |= 0x1f                                 # player 1's wall: row 0 complete
        for g in (9, 50):
            raw[g, 31] = (int(raw[g, 31]) & 0xF8) | 2            # current_player = 2 at an agent decision
        env.set_records(raw.view(recs.dtype).reshape(-1))
        t = ro.traj[0]
        with torch.cuda.stream(ro.streams[0]):
            env.observe_all(ro._persp(), t["obs"][T], t["mask"][T], t["player"][T])      # slot 0 of the first window
        ro.synchronize()
        wins = _windows(ro, 2)
        runs.append((wins, env.get_records(), env.get_rng_range()[1], ro.counters()))
    (wa, ra, pa, ca), (wb, rb, pb, cb) = runs
    for wi in range(2):
        for key in ("obs", "mask", "player", "action", "reward", "done", "value", "log_prob", "entropy", "returns", "opp_replies"):
            assert np.array_equal(wa[wi][key], wb[wi][key]), (wi, key)
        rep = wa[wi]["opp_replies"]
        valid = np.arange(12)[None, :, None] < np.minimum(rep, 12)[:, None, :]
        assert np.array_equal(wa[wi]["opp_action"][valid], wb[wi]["opp_action"][valid]) and np.array_equal(wa[wi]["opp_logp"][valid], wb[wi]["opp_logp"][valid])
    assert ra.tobytes() == rb.tobytes() and np.array_equal(pa, pb) and ca == cb
    d0 = wa[0]["done"][0]
    assert (d0[[5, 33]] == 1).all()                            # GameEnded: reported done, slot restarted
    assert (wa[0]["action"][:, [3, 20, 41]] == -1).all()       # nothing legal for the agent: the slot stays as it is


def test_training_against_a_past_self_resumes_bit_for_bit(golden, tmp_path):
    """A checkpoint of a run with a network opponent carries the opponent's frozen weights: a fresh trainer (other seeds, other weights,
    another opponent) restored from it plays the NEXT window exactly as the uninterrupted run does."""
    import os
    from azul_deep_reinforcement_learning_amd import BatchedActorCritic
    from azul_deep_reinforcement_learning_amd.training import BatchedTrainer
    agent, _ = _nets(golden)
    kw = dict(n_games=128, window=24, results_dir=str(tmp_path), opponent="self", opponent_refresh=2)
    tr = BatchedTrainer(agent, seed_base=7, **kw)
    for _ in range(3):                                   # the opponent was refreshed after update 2 and is one update behind now
        tr.run_batch(collect_stats=False)
    ck = os.path.join(str(tmp_path), "vs.pt")
    tr.save_checkpoint(ck)
    nxt = tr.rollout.run_window()
    tr.rollout.synchronize()
    want = {k: v.clone() for k, v in nxt[0].items()}
    torch.manual_seed(99)
    tr2 = BatchedTrainer(BatchedActorCritic(136, 180, 180), seed_base=9000, **kw)
    tr2.load_checkpoint(ck)
    assert torch.equal(tr2.rollout.ow1t, tr.rollout.ow1t) and not torch.equal(tr2.rollout.ow1t, tr2.rollout.w1t)
    got = tr2.rollout.run_window()
    tr2.rollout.synchronize()
    for k in want:
        assert torch.equal(want[k], got[0][k]), k
