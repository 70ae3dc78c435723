"""Row N1: policy bridge for BASELINE configs[2] -- ActorCritic contract, fused policy_step,
discounted returns, two-stream rollout.  Golden (weights, obs, mask) -> (probs, log-probs, value) triples come from
the reference's own model.py (tests/golden/policy_contract.npz, oracle/gen_golden.py)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import oracle as oz


@pytest.fixture(scope="module")
def contract(golden_dir):
    return np.load(os.path.join(golden_dir, "policy_contract.npz"))


def _net(contract, device):
    from azul_deep_reinforcement_learning_amd.policy import BatchedActorCritic
    sd = {k[3:]: torch.from_numpy(contract[k]) for k in contract.files if k.startswith("sd_")}
    return BatchedActorCritic.from_reference(sd).to(device)


def test_actor_critic_contract_cpu(contract):
    """Same parameter names / shapes / outputs as the reference module (82,081 parameters)."""
    from azul_deep_reinforcement_learning_amd.policy import BatchedActorCritic, IllegalMask
    net = _net(contract, "cpu")
    assert sum(p.numel() for p in net.parameters()) == 82081
    obs, mask = torch.from_numpy(contract["obs"]), torch.from_numpy(contract["mask"])
    with torch.no_grad():
        value = net.forward_critic(obs)
        probs, logp = net.forward_actor(obs, mask)
    assert value.shape == (64, 1) and probs.shape == (64, 180)
    assert torch.allclose(value, torch.from_numpy(contract["value"]), atol=1e-5)
    assert torch.allclose(probs, torch.from_numpy(contract["probs"]), atol=1e-6)
    finite = torch.from_numpy(contract["mask"])
    assert torch.allclose(logp[finite], torch.from_numpy(contract["logp"])[finite], atol=1e-5)
    assert torch.isinf(logp[~finite]).all() and (probs[~finite] == 0).all()
    assert torch.allclose(probs.sum(dim=1), torch.ones(64), atol=1e-5)
    one = torch.zeros(1, 180, dtype=torch.bool)
    one[0, 1] = True
    assert net.forward_actor(obs[:1], one)[0][0, 1] == 1            # reference tests/test_model.py:27-35
    with pytest.raises(IllegalMask):
        net.forward_actor(obs[:1], torch.zeros(1, 180, dtype=torch.bool), check=True)   # tests/test_model.py:36-39
    assert set(BatchedActorCritic().state_dict()) == {k[3:] for k in contract.files if k.startswith("sd_")}


@pytest.mark.gpu
def test_actor_critic_contract_gpu(contract):
    net = _net(contract, "cuda")
    obs, mask = torch.from_numpy(contract["obs"]).cuda(), torch.from_numpy(contract["mask"]).cuda()
    with torch.no_grad():
        value = net.forward_critic(obs).cpu()
        probs, logp = [x.cpu() for x in net.forward_actor(obs, mask)]
    # fp32 GEMM accumulation order differs on the GPU: tolerance, not bit-exactness
    assert torch.allclose(value, torch.from_numpy(contract["value"]), atol=1e-4, rtol=1e-4)
    assert torch.allclose(probs, torch.from_numpy(contract["probs"]), atol=1e-5, rtol=1e-4)


@pytest.mark.gpu
def test_discounted_returns_matches_reference_loop():
    from azul_deep_reinforcement_learning_amd import _lib as L
    T, n, gamma = 57, 300, 0.99
    rs = np.random.RandomState(0)
    reward = rs.randint(-30, 25, size=(T, n)).astype(np.int32)
    done = (rs.rand(T, n) < 0.03).astype(np.uint8)
    exp = np.zeros((T, n), dtype=np.float64)
    for g in range(n):
        q = 0.0
        for t in reversed(range(T)):                 # nn_runner.py:72-76, restarted at every episode end
            if done[t, g]:
                q = 0.0
            q = reward[t, g] + gamma * q
            exp[t, g] = q
    r, d = torch.from_numpy(reward).cuda(), torch.from_numpy(done).cuda()
    out = torch.zeros(T, n, device="cuda")
    L.check(L.lib.azul_discounted_returns(C.c_void_p(r.data_ptr()), C.c_void_p(d.data_ptr()), C.c_void_p(out.data_ptr()), None,
                                          C.c_float(gamma), T, n, None))
    torch.cuda.synchronize()
    assert np.allclose(out.cpu().numpy(), exp, rtol=1e-5, atol=1e-3)     # fp32 recurrence vs fp64


def _oracle_replay(rec, mt, pos, fp, pool, actions):
    """Flat policy-driven self-play in the oracle from a given record + CPython stream, with auto-reset."""
    L = oz.lib()
    q = oz.unpack(rec, pool, fp)
    r = oz.Rng()
    L.oz_rng_set(C.byref(r), mt.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos))
    out = {"mask": [], "obs": [], "player": [], "reward": [], "done": []}
    for a in actions:
        out["mask"].append(oz.check_all_valid(q.game))
        cur = q.game.current_player
        out["player"].append(cur)
        out["obs"].append(oz.get_state(q.game, cur - 1))
        d, c, p = int(a) % 6, (int(a) // 6) % 5, int(a) // 30
        assert L.oz_step(C.byref(q.game), d, c, p, C.byref(r)) == 0
        q.move_counter += 1
        phi = L.oz_potential(C.byref(q.game))
        out["reward"].append(phi - q.player_score)
        q.player_score = phi
        dn = bool(L.oz_is_end_of_game(C.byref(q.game)))
        out["done"].append(dn)
        if dn:
            assert L.oz_runner_init(C.byref(q), fp, pool, C.byref(r)) == 0
    return out, oz.pack(q)


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph,fused,mlp", [(False, True, True), (True, True, True), (True, True, False), (False, False, False)])
def test_policy_rollout_replays_exactly_through_the_oracle(contract, use_graph, fused, mlp):
    """Everything the rollout records about the ENV (mask, observation, player, reward, done, final state) must be
    what the oracle computes when it is fed the actions the policy sampled -- for both stream parts, eager launches
    and HIP-graph replays alike."""
    from azul_deep_reinforcement_learning_amd import PolicyRollout
    torch.manual_seed(1)
    net = _net(contract, "cuda")
    T, n = 40, 64
    ro = PolicyRollout(net, n_games=n, parts=2, seed_base=500, window=T, use_graph=use_graph, fused_head=fused, fused_mlp=mlp)
    assert ro.use_graph == use_graph, ro.graph_error
    assert ro.fused_mlp == mlp
    start = [(env.get_records(), [env.get_rng(g) for g in range(ro.h)]) for env in ro.envs]
    windows = []
    for _ in range(2):
        tr = ro.run_window()
        ro.synchronize()
        windows.append([{k: v.cpu().numpy().copy() for k, v in part.items()} for part in tr])
    finals = [env.get_records() for env in ro.envs]
    for p in (0, 1):
        for g in range(0, ro.h, 5):
            acts = np.concatenate([w[p]["action"][:, g] for w in windows])
            assert (acts >= 0).all()
            mt, pos = start[p][1][g]
            exp, final = _oracle_replay(start[p][0][g], mt, pos, oz.FIRST_RANDOM, oz.POOL_LID, acts)
            got_mask = np.concatenate([w[p]["mask"][:T, g] for w in windows]).astype(bool)
            assert np.array_equal(got_mask, np.array(exp["mask"]))
            assert np.array_equal(np.concatenate([w[p]["obs"][:T, g] for w in windows]).astype(np.int64), np.array(exp["obs"]))
            assert np.array_equal(np.concatenate([w[p]["player"][:T, g] for w in windows]), np.array(exp["player"]))
            assert np.array_equal(np.concatenate([w[p]["reward"][:, g] for w in windows]), np.array(exp["reward"]))
            assert np.array_equal(np.concatenate([w[p]["done"][:, g] for w in windows]).astype(bool), np.array(exp["done"]))
            assert final.tobytes() == finals[p][g].tobytes()
    # policy-side records are self-consistent: the sampled action is legal, log_prob/entropy finite, returns follow the scan
    for w in windows:
        for part in w:
            a = part["action"].astype(np.int64)
            assert np.take_along_axis(part["mask"][:T], a[..., None], axis=2).all()
            assert np.isfinite(part["log_prob"]).all() and np.isfinite(part["entropy"]).all() and (part["entropy"] >= 0).all()
            q = np.zeros(ro.h)
            for t in reversed(range(T)):
                q = np.where(part["done"][t] != 0, 0.0, q)
                q = part["reward"][t] + 0.99 * q
                assert np.allclose(part["returns"][t], q, rtol=1e-4, atol=1e-2)


@pytest.mark.gpu
def test_policy_head_matches_torch_and_samples_the_distribution(contract):
    """azul_policy_head vs PyTorch: log-prob of the chosen action and the entropy term within fp32 tolerance, sampled
    actions always legal, and the empirical distribution of many draws follows softmax(logits | mask)."""
    from azul_deep_reinforcement_learning_amd import _lib as L
    rs = np.random.RandomState(5)
    n = 256
    logits = torch.from_numpy(rs.randn(n, 180).astype(np.float32) * 2).cuda()
    mask_np = rs.rand(n, 180) < 0.2
    mask_np[np.arange(n), rs.randint(0, 180, n)] = True
    mask_np[7] = False                                     # a stuck row
    mask = torch.from_numpy(mask_np.astype(np.uint8)).cuda()
    action = torch.zeros(n, dtype=torch.int32, device="cuda")
    logp = torch.zeros(n, device="cuda")
    ent = torch.zeros(n, device="cuda")
    ref_logp = torch.log_softmax(logits.masked_fill(~mask.bool(), float("-inf")), dim=1)
    ref_ent = -(torch.where(mask.bool(), ref_logp, torch.zeros_like(ref_logp)).sum(1) / mask.sum(1).clamp(min=1))
    counts = torch.zeros(180, device="cuda")
    draws = 4000
    for c in range(draws):
        L.check(L.lib.azul_policy_head(C.c_void_p(logits.data_ptr()), C.c_void_p(mask.data_ptr()), 1234, c, None, n, 0,
                                       C.c_void_p(action.data_ptr()), C.c_void_p(logp.data_ptr()), C.c_void_p(ent.data_ptr()), None))
        if c < 3:
            a = action.long()
            assert a[7] == -1 and logp[7] == 0 and ent[7] == 0
            rows = torch.arange(n, device="cuda")
            keep = rows != 7
            assert mask.bool()[rows[keep], a[keep]].all()
            assert torch.allclose(logp[keep], ref_logp[rows[keep], a[keep]], atol=2e-5, rtol=1e-5)
            assert torch.allclose(ent[keep], ref_ent[keep], atol=2e-5, rtol=1e-5)
        counts[action[0].long()] += 1
    torch.cuda.synchronize()
    p0 = ref_logp[0].exp().cpu().numpy()
    emp = (counts / draws).cpu().numpy()
    legal = mask_np[0]
    assert emp[~legal].sum() == 0
    # chi-square over the legal actions of row 0 (expected counts >= ~5 for the bulk): loose 5-sigma style bound
    e = p0[legal] * draws
    chi2 = float((((emp[legal] * draws) - e) ** 2 / np.maximum(e, 1e-9)).sum())
    dof = int(legal.sum()) - 1
    assert chi2 < dof + 6 * np.sqrt(2 * dof) + 10, (chi2, dof)
    # a different counter or seed gives a different (but reproducible) stream
    L.check(L.lib.azul_policy_head(C.c_void_p(logits.data_ptr()), C.c_void_p(mask.data_ptr()), 1234, 17, None, n, 0,
                                   C.c_void_p(action.data_ptr()), C.c_void_p(logp.data_ptr()), C.c_void_p(ent.data_ptr()), None))
    a1 = action.clone()
    L.check(L.lib.azul_policy_head(C.c_void_p(logits.data_ptr()), C.c_void_p(mask.data_ptr()), 1234, 17, None, n, 0,
                                   C.c_void_p(action.data_ptr()), C.c_void_p(logp.data_ptr()), C.c_void_p(ent.data_ptr()), None))
    assert torch.equal(a1, action)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 16, 45, 2048])
def test_fused_forward_matches_the_reference_network(contract, n):
    """azul_policy_forward (f32 matrix cores, one launch) vs the PyTorch module carrying the reference's weights: value and
    logits within f32 round-off, and the head's outputs equal to azul_policy_head run on the kernel's own logits (same Philox
    stream => same action).  n = 1 / 45 exercise the ragged last tile; the weights/obs are asymmetric random data."""
    from azul_deep_reinforcement_learning_amd import _lib as L
    torch.manual_seed(n)
    net = _net(contract, "cuda")
    with torch.no_grad():                       # de-symmetrise: the golden net is a default init, give every bias a value
        for p_ in net.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    rs = np.random.RandomState(n)
    obs = torch.from_numpy(rs.randint(0, 6, size=(n, 136)).astype(np.float32)).cuda()
    mask_np = rs.rand(n, 180) < 0.15
    mask_np[np.arange(n), rs.randint(0, 180, n)] = True
    if n > 20:
        mask_np[7] = False                      # a stuck game: action -1
    mask = torch.from_numpy(mask_np.astype(np.uint8)).cuda()
    w1t = torch.cat([net.critic_linear1.weight, net.actor_linear1.weight], dim=0).t().contiguous()
    b1 = torch.cat([net.critic_linear1.bias, net.actor_linear1.bias]).contiguous()
    w2c = net.critic_linear2.weight.reshape(-1).contiguous()
    w2a_t = net.actor_linear2.weight.t().contiguous()
    value = torch.zeros(n, device="cuda")
    action = torch.zeros(n, dtype=torch.int32, device="cuda")
    logp = torch.zeros(n, device="cuda")
    ent = torch.zeros(n, device="cuda")
    logits = torch.zeros(n, 180, device="cuda")
    counter = torch.tensor([5, 0], dtype=torch.int64, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    for rep in range(2):                        # second launch: the device counter advanced by itself
        L.check(L.lib.azul_policy_forward(p(obs), p(mask), p(w1t), p(b1), p(w2c), p(net.critic_linear2.bias), p(w2a_t),
                                          p(net.actor_linear2.bias), 136, 180, 180, 77, 1000, p(counter), 1, n, 0, p(value), p(action),
                                          p(logp), p(ent), p(logits), None))
        torch.cuda.synchronize()
        assert counter.tolist() == [6 + rep, 0]
        with torch.no_grad():
            v_ref = net.forward_critic(obs).squeeze(1)
            lg_ref = net.actor_linear2(torch.relu(net.actor_linear1(obs)))
        assert torch.allclose(value, v_ref, rtol=1e-5, atol=2e-5)
        assert torch.allclose(logits, lg_ref, rtol=1e-5, atol=2e-5)
        a2 = torch.zeros_like(action)
        lp2 = torch.zeros_like(logp)
        e2 = torch.zeros_like(ent)
        L.check(L.lib.azul_policy_head(p(logits), p(mask), 77, 1000 + 5 + rep, None, n, 0, p(a2), p(lp2), p(e2), None))
        torch.cuda.synchronize()
        assert torch.equal(action, a2) and torch.equal(logp, lp2) and torch.equal(ent, e2)
        a = action.cpu().numpy()
        legal_rows = mask_np.any(axis=1)
        assert (a[~legal_rows] == -1).all() and mask_np[np.flatnonzero(legal_rows), a[legal_rows]].all()
    # other network shapes are refused, not silently mis-computed
    assert L.lib.azul_policy_forward(p(obs), p(mask), p(w1t), p(b1), p(w2c), p(net.critic_linear2.bias), p(w2a_t), p(net.actor_linear2.bias),
                                     136, 128, 180, 77, 0, None, 0, n, 0, p(value), p(action), p(logp), p(ent), None, None) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("opponent,n,rules", [(None, 100, None), ("random", 100, None), (None, 16, None), ("random", 2048, None), (None, 4096, None),
                                              (None, 37, {"first_player": 2, "tile_pool": "Random"}),
                                              ("random", 37, {"first_player": 1, "tile_pool": "Random"})])
def test_persistent_rollout_is_bit_identical_to_the_per_move_path(contract, opponent, n, rules):
    """azul_batch_policy_rollout (a whole window in one launch, games resident in registers / LDS) must reproduce the
    two-launches-per-move path bit for bit: observations, masks, players, actions, rewards, dones, values, log-probs,
    entropies, returns, the final game records and RNG positions -- over two windows, with a ragged last workgroup (n = 100)."""
    from azul_deep_reinforcement_learning_amd import PolicyRollout
    T = 45
    runs = []
    for persistent in (False, True):
        torch.manual_seed(3)
        net = _net(contract, "cuda")
        kw = {} if rules is None else {"rules": rules}
        ro = PolicyRollout(net, n_games=n, parts=1, seed_base=77, window=T, use_graph=False, opponent=opponent, persistent=persistent, **kw)
        assert ro.persistent == persistent
        wins = []
        for _ in range(2):
            tr = ro.run_window()
            ro.synchronize()
            wins.append({k: v.clone() for k, v in tr[0].items()})
        runs.append((wins, ro.envs[0].get_records(), ro.envs[0].get_rng_range()[1], ro.counters(), ro.work[0]["counter"].tolist()))
    (wa, ra, pa, ca, cta), (wb, rb, pb, cb, ctb) = runs
    for wi in range(2):
        for key in ("obs", "mask", "player", "action", "reward", "done", "value", "log_prob", "entropy", "returns"):
            assert torch.equal(wa[wi][key], wb[wi][key]), (wi, key)
    assert ra.tobytes() == rb.tobytes() and np.array_equal(pa, pb) and ca == cb and cta == ctb
    assert ca["episodes"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("opponent", [None, "random"])
def test_rollout_kernel_follows_the_per_move_path_from_unusual_states(contract, opponent):
    """The paths ordinary self-play rarely reaches, taken on purpose: games handed in without any tile on the table (nothing legal: the
    policy head answers -1, hazard H3 -> stuck counter, slot restarted), games whose record says "ended" (GameEnded -> done, restarted),
    and games whose wall already holds a complete row while the record's flag is still clear (is_end_of_game() reads the walls,
    azul.py:184-191; Azul.step's GameEnded reads the flag, :298-299).  The one-launch-per-window kernel must do what the per-move path
    does: every trajectory array, the final records, RNG positions, episode and stuck counters."""
    from azul_deep_reinforcement_learning_amd import PolicyRollout
    T, n = 12, 64
    runs = []
    for persistent in (False, True):
        torch.manual_seed(3)
        ro = PolicyRollout(_net(contract, "cuda"), n_games=n, parts=1, seed_base=4242, window=T, use_graph=False, opponent=opponent, persistent=persistent)
        env = ro.envs[0]
        recs = env.get_records()
        raw = recs.view(np.uint8).reshape(n, -1).copy()
        for g in (3, 20, 41):
            raw[g, 0:31] = 0                                   # displays, centre and token empty
        for g in (5, 33):
            raw[g, 31] |= 0x40                                 # the record's "ended" flag
        for g in (7, 40, 63):
            raw[g, 84] |= 0x1f                                 # player 1's wall: row 0 complete
        env.set_records(raw.view(recs.dtype).reshape(-1))
        t = ro.traj[0]
        with torch.cuda.stream(ro.streams[0]):
            env.observe_all(ro._persp(), t["obs"][T], t["mask"][T], t["player"][T])      # slot 0 of the first window
        ro.synchronize()
        wins = []
        for _ in range(2):
            tr = ro.run_window()
            ro.synchronize()
            wins.append({k: v.clone() for k, v in tr[0].items()})
        runs.append((wins, env.get_records(), env.get_rng_range()[1], ro.counters()))
    (wa, ra, pa, ca), (wb, rb, pb, cb) = runs
    for wi in range(2):
        for key in ("obs", "mask", "player", "action", "reward", "done", "value", "log_prob", "entropy", "returns"):
            assert torch.equal(wa[wi][key], wb[wi][key]), (wi, key)
    assert ra.tobytes() == rb.tobytes() and np.array_equal(pa, pb) and ca == cb
    d0 = wa[0]["done"][0].cpu().numpy()
    assert (d0[[5, 33]] == 1).all()                            # GameEnded: reported done, slot restarted
    if opponent is None:
        assert ca["stuck"] >= 3 and (d0[[3, 20, 41]] == 2).all()
    # (GameRunner.step with "no action" is a BAD_ACTION for the agent, game_runner.py:44: such a slot is left as it is)


@pytest.mark.gpu
@pytest.mark.parametrize("opponent", [None, "random"])
def test_rollout_kernel_long_run_equals_the_per_move_path(contract, opponent):
    """40 windows of 32 steps on 512 games (about a hundred episodes and several MT19937 regenerations per game): the one-launch-per-
    window kernel and the two-launches-per-move path must end with the same records, the same 624-word generator states, counters and
    last window (tools/soak_rollout.py is the 300-window, 4096-game form of this)."""
    from azul_deep_reinforcement_learning_amd import PolicyRollout
    runs = []
    for persistent in (False, True):
        torch.manual_seed(11)
        ro = PolicyRollout(_net(contract, "cuda"), n_games=512, parts=1, seed_base=90210, window=32, use_graph=False, opponent=opponent,
                           persistent=persistent)
        for _ in range(40):
            tr = ro.run_window()
        ro.synchronize()
        mt, pos = ro.envs[0].get_rng_range()
        runs.append(({k: v.clone() for k, v in tr[0].items()}, ro.envs[0].get_records(), mt, pos, ro.counters()))
    (la, ra, ma, pa, ca), (lb, rb, mb, pb, cb) = runs
    for key in la:
        assert torch.equal(la[key], lb[key]), key
    assert ra.tobytes() == rb.tobytes() and np.array_equal(ma, mb) and np.array_equal(pa, pb) and ca == cb
    assert ca["episodes"] > 5000


@pytest.mark.gpu
def test_argmax_action_selection(contract):
    """seed = AZUL_POLICY_ARGMAX: Agent.get_ac_output(action_selection="Max") = np.argmax of the masked softmax (first maximum),
    with the same log-prob / entropy outputs; and a greedy rollout is the same whichever launch structure plays it."""
    from azul_deep_reinforcement_learning_amd import PolicyRollout, _lib as L
    rs = np.random.RandomState(5)
    n = 500
    logits = torch.from_numpy(rs.randn(n, 180).astype(np.float32) * 2).cuda()
    logits[3, 40:60] = 7.5                                                     # a tie: the FIRST maximum wins
    m = rs.rand(n, 180) < 0.25
    m[np.arange(n), rs.randint(0, 180, n)] = True
    m[3, 40:60] = True
    m[9] = False
    mask = torch.from_numpy(m.astype(np.uint8)).cuda()
    action = torch.zeros(n, dtype=torch.int32, device="cuda")
    logp = torch.zeros(n, device="cuda")
    ent = torch.zeros(n, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    L.check(L.lib.azul_policy_head(p(logits), p(mask), L.POLICY_ARGMAX, 123, None, n, 0, p(action), p(logp), p(ent), None))
    torch.cuda.synchronize()
    masked = logits.masked_fill(~mask.bool(), float("-inf"))
    want = masked.argmax(dim=1)
    ok = mask.bool().any(dim=1)
    assert torch.equal(action[ok].long(), want[ok]) and int(action[9]) == -1 and int(action[3]) == 40
    ref_lp = torch.log_softmax(masked[ok], dim=1).gather(1, want[ok].unsqueeze(1)).squeeze(1)
    assert torch.allclose(logp[ok], ref_lp, rtol=1e-5, atol=2e-5)
    runs = []
    for persistent, seed in ((False, 1), (True, 999)):                        # the sampling seed must not matter
        net = _net(contract, "cuda")
        ro = PolicyRollout(net, n_games=48, parts=1, seed_base=5, window=30, use_graph=False, opponent="random", persistent=persistent,
                           action_selection="Max", sample_seed=seed)
        tr = ro.run_window()
        ro.synchronize()
        runs.append({k: v.clone() for k, v in tr[0].items()})
    for k in ("obs", "mask", "action", "reward", "done", "value", "log_prob"):
        assert torch.equal(runs[0][k], runs[1][k]), k
