"""The matrix-core kernels of rows N1 / N2 on CPU: azul_policy_forward_kernel (csrc/azul_policy.hpp: the ActorCritic forward of
model.py:22-41 + the sampling head) and azul_a2c_grad_kernel (csrc/azul_learner.hpp: forward + backward of the loss of agent.py:39-62,
weight-gradient tiles in registers), compiled UNMODIFIED by g++ and run as workgroups of emulated wavefronts (tests/hostcheck/simt:
MFMA, buffer loads, s_barrier) against PyTorch on the CPU: values / logits / log-probs / entropies of the forward, and the full
82,081-element gradient + loss sums against autograd of the reference's loss.  Under ASan / UBSan (run_sanitizers.sh) every LDS and
global index these kernels form is checked too."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostcheck")


def load(name=None):
    name = name or os.environ.get("AZUL_SIMT_LEARNER_LIB", "libsimt_learner.so")
    subprocess.check_call(["make", "-s", "-C", HERE, name], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(HERE, name))
    L.sl_gradients.restype = C.c_longlong
    L.sl_gradients.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 9
    L.sl_forward.restype = C.c_longlong
    L.sl_forward.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_ulonglong, C.c_ulonglong, C.c_uint] + [C.c_void_p] * 5
    L.sl_buffer_oob.restype = C.c_ulonglong
    return L


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def make(seed, n):
    rs = np.random.RandomState(seed)
    w = {"w1t": rs.randn(136, 360) * 0.08, "b1": rs.randn(360) * 0.05, "w2c": rs.randn(180) * 0.1, "b2c": rs.randn(1) * 0.1,
         "w2a_t": rs.randn(180, 180) * 0.12, "b2a": rs.randn(180) * 0.05}
    w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}
    w["w2a"] = np.ascontiguousarray(w["w2a_t"].T)                      # actor_linear2.weight as PyTorch stores it
    obs = rs.randint(0, 6, size=(n, 136)).astype(np.float32)
    mask = rs.rand(n, 180) < 0.2
    action = rs.randint(0, 180, n).astype(np.int32)
    mask[np.arange(n), action] = True
    q = (rs.randn(n) * 5).astype(np.float32)
    return w, obs, np.ascontiguousarray(mask.astype(np.uint8)), action, q


def torch_forward(w, obs, mask):
    t = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in w.items() if k != "w2a"}
    x = torch.tensor(obs)
    h = torch.relu(x @ t["w1t"] + t["b1"])
    value = h[:, :180] @ t["w2c"] + t["b2c"]
    logits = h[:, 180:] @ t["w2a_t"] + t["b2a"]
    legal = torch.tensor(mask.astype(bool))
    logp = torch.log_softmax(logits.masked_fill(~legal, float("-inf")), dim=1)
    ent = -(torch.where(legal, logp, torch.zeros_like(logp)).sum(1) / legal.sum(1))
    return t, value, logits, logp, ent


def test_forward_kernel_under_emulation_matches_torch():
    L = load()
    n = 37                                                              # ragged last workgroup of 16
    w, obs, mask, _, _ = make(3, n)
    mask[5] = 0                                                         # nothing legal: action -1, log-prob / entropy 0
    value, logp, ent = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    action, logits = np.zeros(n, np.int32), np.zeros((n, 180), np.float32)
    oob0 = L.sl_buffer_oob()
    ops = L.sl_forward(n, ptr(obs), ptr(mask), ptr(w["w1t"]), ptr(w["b1"]), ptr(w["w2c"]), ptr(w["b2c"]), ptr(w["w2a_t"]), ptr(w["b2a"]),
                       99, 7, 500, ptr(value), ptr(action), ptr(logp), ptr(ent), ptr(logits))
    assert ops > 1000 and L.sl_buffer_oob() == oob0
    keep = np.arange(n) != 5
    mk = mask.copy()
    mk[5, 0] = 1
    _, tv, tl, tlogp, tent = torch_forward(w, obs, mk)
    assert np.allclose(value, tv.detach().numpy(), atol=2e-4, rtol=1e-4)
    assert np.allclose(logits, tl.detach().numpy(), atol=2e-4, rtol=1e-4)
    assert action[5] == -1 and logp[5] == 0 and ent[5] == 0
    rows = np.arange(n)[keep]
    assert (mask[rows, action[rows]] == 1).all()
    assert np.allclose(logp[keep], tlogp.detach().numpy()[rows, action[rows]], atol=2e-4)
    assert np.allclose(ent[keep], tent.detach().numpy()[keep], atol=2e-4, rtol=1e-4)
    # the same call again: the same draws (Philox keyed by seed, counter and global id)
    a2 = np.zeros(n, np.int32)
    L.sl_forward(n, ptr(obs), ptr(mask), ptr(w["w1t"]), ptr(w["b1"]), ptr(w["w2c"]), ptr(w["b2c"]), ptr(w["w2a_t"]), ptr(w["b2a"]),
                 99, 7, 500, ptr(value), ptr(a2), ptr(logp), ptr(ent), ptr(logits))
    assert np.array_equal(a2, action)


def test_gradient_kernel_under_emulation_matches_autograd_of_the_references_loss():
    L = load()
    lay = (C.c_int * 8)()
    L.sl_layout(lay)
    W1, B1, W2C, B2C, W2A, B2A, LOSS, TOTAL = list(lay)
    n, parts = 75, 2                                                    # 32-sample passes, a ragged last one, two workgroups
    w, obs, mask, action, q = make(11, n)
    partial, grad = np.zeros((parts, TOTAL), np.float32), np.zeros(TOTAL, np.float32)
    oob0 = L.sl_buffer_oob()
    ops = L.sl_gradients(n, parts, ptr(obs), ptr(mask), ptr(action), ptr(q), None, 1.0 / n, ptr(w["w1t"]), ptr(w["b1"]), ptr(w["w2c"]),
                         ptr(w["b2c"]), ptr(w["w2a_t"]), ptr(w["b2a"]), ptr(w["w2a"]), ptr(partial), ptr(grad))
    assert ops > 1000 and L.sl_buffer_oob() == oob0
    # agent.py:45-57: adv = q - v (not detached), actor = -logp[a] * adv, critic = 0.5 adv^2, entropy term 0.1 * (-mean legal logp)
    t, value, _, logp, ent = torch_forward(w, obs, mask)
    adv = torch.tensor(q) - value
    lp = logp[torch.arange(n), torch.tensor(action.astype(np.int64))]
    actor, critic, entropy = (-lp * adv).sum(), (0.5 * adv * adv).sum(), ent.sum()
    loss = (actor + critic + 0.1 * entropy) / n
    loss.backward()
    ref = {k: v.grad.numpy() for k, v in t.items()}

    def close(got, want, what):
        scale = max(1e-3, float(np.abs(want).max()))
        assert np.abs(got - want).max() < 3e-4 * scale, (what, float(np.abs(got - want).max()), scale)

    close(grad[W1:W1 + 136 * 360].reshape(136, 360), ref["w1t"], "dw1t")
    close(grad[B1:B1 + 360], ref["b1"], "db1")
    close(grad[W2C:W2C + 180], ref["w2c"], "dw2c")
    close(grad[B2C:B2C + 1], ref["b2c"], "db2c")
    close(grad[W2A:W2A + 180 * 180].reshape(180, 180), ref["w2a_t"], "dw2a_t")
    close(grad[B2A:B2A + 180], ref["b2a"], "db2a")
    sums = grad[LOSS:LOSS + 4]
    assert abs(sums[3] - n) < 1e-3
    # the logged sums: actor term, critic term as the reference logs it (advantage^2: agent.py:50 takes half of it into the loss), entropy
    want = np.array([float(actor.detach()), 2.0 * float(critic.detach()), float(entropy.detach())])
    assert np.allclose(sums[:3], want, rtol=2e-4, atol=2e-3), (sums, want)
    # a device-built selection: sample s lives in row index[s] of the arrays -- the same samples in the same order, the same bits
    perm = np.random.RandomState(2).permutation(n)
    index = np.ascontiguousarray(np.argsort(perm).astype(np.int32))
    stored = [np.ascontiguousarray(x[perm]) for x in (obs, mask, action, q)]
    g2, p2 = np.zeros(TOTAL, np.float32), np.zeros((parts, TOTAL), np.float32)
    L.sl_gradients(n, parts, ptr(stored[0]), ptr(stored[1]), ptr(stored[2]), ptr(stored[3]), ptr(index), 1.0 / n, ptr(w["w1t"]), ptr(w["b1"]),
                   ptr(w["w2c"]), ptr(w["b2c"]), ptr(w["w2a_t"]), ptr(w["b2a"]), ptr(w["w2a"]), ptr(p2), ptr(g2))
    assert np.array_equal(g2, grad)


def test_ring_selection_kernels_under_emulation_hand_out_every_step_exactly_once():
    """azul_select_ring_count_kernel / _write_kernel (NNRunner.train's choice of samples when episodes straddle windows, nn_runner.py:59-76)
    on synthetic rings against a direct model: after every window each game contributes the steps from its first untrained step up to its
    last episode end inside the window -- those still intact in the ring, carrying an action -- game by game, steps ascending; what has
    fallen out of the ring is counted.  Episodes longer than the ring occur (they must be dropped, not mis-indexed)."""
    L = load()
    L.sl_select_ring.restype = C.c_longlong
    L.sl_select_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
    L.sl_returns_ring.restype = C.c_longlong
    L.sl_returns_ring.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_longlong, C.c_int, C.c_int]
    rs = np.random.RandomState(9)
    n, T, D, windows = 37, 8, 4, 40
    R = T * D
    done = np.zeros((R, n), np.uint8)
    action = np.zeros((R, n), np.int32)
    reward, returns = np.zeros((R, n), np.int32), np.full((R, n), np.nan, np.float32)
    hist_r, hist_d, gamma = [], [], np.float32(0.97)
    pend, mpend = np.zeros(n, np.int32), np.zeros(n, np.int64)
    index, count, countf = np.zeros(R * n, np.int32), np.zeros(2, np.int32), np.zeros(2, np.float32)
    scratch = np.zeros(3 * n + (n + 3) // 4, np.int32)
    seen, dropped_model, total = set(), 0, 0
    for w in range(windows):
        for t in range(T):
            s = w * T + t
            p_end = np.where(np.arange(n) % 5 == 0, 0.02, 0.14)          # every fifth game plays long episodes (longer than the ring)
            done[s % R] = rs.rand(n) < p_end
            action[s % R] = np.where(rs.rand(n) < 0.06, -1, rs.randint(0, 180, n))
            reward[s % R] = rs.randint(-9, 10, n)
            hist_r.append(reward[s % R].copy())
            hist_d.append(done[s % R].copy())
        s_end = (w + 1) * T
        # the returns of every step the ring holds, in one scan from the newest step backwards (azul_returns_ring_kernel)
        assert L.sl_returns_ring(ptr(reward), ptr(done), ptr(returns), gamma, R, s_end, min(R, s_end), n) > 0
        assert L.sl_select_ring(ptr(done), ptr(action), T, D, n, s_end, ptr(pend), ptr(index), ptr(count), ptr(countf), ptr(scratch)) > 0
        want = []
        lo = max(0, s_end - R + (1 if s_end % R else 0))
        for g in range(n):
            ends = [s for s in range(s_end - T, s_end) if done[s % R, g]]
            if not ends:
                continue
            start = max(int(mpend[g]), lo)
            dropped_model += start - int(mpend[g])
            q = np.float32(0)
            exact = {}
            for s in range(ends[-1], start - 1, -1):               # nn_runner.py:70-76 over the game's own history
                q = np.float32(hist_r[s][g]) + gamma * (np.float32(0) if hist_d[s][g] else q)
                exact[s] = q
            for s in range(start, ends[-1] + 1):
                if action[s % R, g] >= 0:
                    want.append((s % R) * n + g)
                    assert (g, s) not in seen
                    seen.add((g, s))
                    assert returns[s % R, g] == exact[s], (w, g, s)       # the same additions in the same order: the same bits
            mpend[g] = ends[-1] + 1
        cnt = int(count[0])
        assert cnt == len(want) and index[:cnt].tolist() == want, (w, cnt, len(want))
        assert countf[0] == cnt and abs(countf[1] - 1.0 / max(cnt, 1)) < 1e-7
        assert np.array_equal(pend, mpend.astype(np.int32))
        total += cnt
    assert int(count[1]) == dropped_model and dropped_model > 0 and total > n * 100


def test_adam_kernel_under_emulation_matches_torch_adam_and_keeps_both_layouts_in_step():
    """azul_a2c_step_kernel + azul_a2c_apply_kernel (agent.py:37,60: torch.optim.Adam, defaults) on the flat k-major master copy AND the
    eight nn.Linear tensors: three updates with random gradients against torch.optim.Adam on a CPU module fed the same gradients in its
    own layouts; afterwards the master copy is still the module, transposed; an update without samples moves nothing."""
    L = load()
    L.sl_adam.restype = C.c_longlong
    L.sl_adam.argtypes = [C.c_void_p] * 4 + [C.c_float] * 4 + [C.c_void_p] * 11
    lay = (C.c_int * 8)()
    L.sl_layout(lay)
    W1, B1, W2C, B2C, W2A, B2A, LOSS, TOTAL = list(lay)
    rs = np.random.RandomState(4)
    mod = {"c1w": rs.randn(180, 136) * 0.1, "c1b": rs.randn(180) * 0.1, "c2w": rs.randn(1, 180) * 0.1, "c2b": rs.randn(1) * 0.1,
           "a1w": rs.randn(180, 136) * 0.1, "a1b": rs.randn(180) * 0.1, "a2w": rs.randn(180, 180) * 0.1, "a2b": rs.randn(180) * 0.1}
    mod = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in mod.items()}

    def flat_of(d):
        f = np.zeros(TOTAL, np.float32)
        f[W1:W1 + 136 * 360] = np.concatenate([d["c1w"], d["a1w"]], axis=0).T.reshape(-1)
        f[B1:B1 + 360] = np.concatenate([d["c1b"], d["a1b"]])
        f[W2C:W2C + 180] = d["c2w"][0]
        f[B2C] = d["c2b"][0]
        f[W2A:W2A + 180 * 180] = d["a2w"].T.reshape(-1)
        f[B2A:B2A + 180] = d["a2b"]
        return f

    flat, m, v = flat_of(mod), np.zeros(TOTAL, np.float32), np.zeros(TOTAL, np.float32)
    tp = {k: torch.nn.Parameter(torch.tensor(x.copy())) for k, x in mod.items()}
    opt = torch.optim.Adam(list(tp.values()), lr=1e-3)
    step, n_total, stats = np.zeros(1, np.int32), np.array([100.0], np.float32), np.zeros(5, np.float32)
    for it in range(3):
        gd = {k: (rs.randn(*x.shape) * 0.5).astype(np.float32) for k, x in mod.items()}
        grad = flat_of(gd)
        grad[LOSS:LOSS + 4] = [3.0, 8.0, 5.0, 100.0]
        assert L.sl_adam(ptr(grad), ptr(flat), ptr(m), ptr(v), 1e-3, 0.9, 0.999, 1e-8, *[ptr(mod[k]) for k in ("c1w", "c1b", "c2w", "c2b", "a1w", "a1b", "a2w", "a2b")],
                         ptr(step), ptr(n_total), ptr(stats)) > 0
        for k in tp:
            tp[k].grad = torch.tensor(gd[k])
        opt.step()
        assert int(step[0]) == it + 1
        for k in tp:
            assert np.allclose(mod[k], tp[k].detach().numpy(), rtol=0, atol=2e-6), (it, k)
        assert np.array_equal(flat[:LOSS], flat_of(mod)[:LOSS])                  # the master copy IS the module, k-major
        assert np.allclose(stats, [0.03, 0.08, 0.05, 0.03 + 0.5 * 0.08 + 0.1 * 0.05, 100.0], atol=1e-6)
    # a window without a finished episode: no samples, no step, nothing moves
    before, n0 = {k: x.copy() for k, x in mod.items()}, np.array([0.0], np.float32)
    L.sl_adam(ptr(grad), ptr(flat), ptr(m), ptr(v), 1e-3, 0.9, 0.999, 1e-8, *[ptr(mod[k]) for k in ("c1w", "c1b", "c2w", "c2b", "a1w", "a1b", "a2w", "a2b")],
              ptr(step), ptr(n0), ptr(stats))
    assert int(step[0]) == 3 and all(np.array_equal(before[k], mod[k]) for k in mod)


def test_the_remaining_small_kernels_under_emulation():
    """azul_policy_head_kernel (the head alone: the forward kernel's draws, log-probs and entropies on its own logits, bit for bit),
    azul_select_complete_kernel (one window: the steps of episodes that end inside it, game by game), azul_returns_kernel (nn_runner.py:70-76
    with a carry between windows), azul_seed_kernel (random.seed(base + g) for every game: the oracle's seeding) and azul_a2c_reduce_kernel
    (the partials added in workgroup order)."""
    from oracle import oracle as oz
    L = load()
    for name, res in (("sl_head", None), ("sl_select_complete", None), ("sl_returns", None), ("sl_seed", None), ("sl_reduce", None)):
        getattr(L, name).restype = C.c_longlong
    L.sl_head.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sl_select_complete.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.sl_returns.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int]
    L.sl_seed.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p]
    L.sl_reduce.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    # -- the head on the forward kernel's logits
    n = 22
    w, obs, mask, _, _ = make(8, n)
    value, logp, ent = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    action, logits = np.zeros(n, np.int32), np.zeros((n, 180), np.float32)
    L.sl_forward(n, ptr(obs), ptr(mask), ptr(w["w1t"]), ptr(w["b1"]), ptr(w["w2c"]), ptr(w["b2c"]), ptr(w["w2a_t"]), ptr(w["b2a"]),
                 5, 40, 900, ptr(value), ptr(action), ptr(logp), ptr(ent), ptr(logits))
    a2, lp2, en2 = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    assert L.sl_head(n, ptr(logits), ptr(mask), 5, 40, 900, ptr(a2), ptr(lp2), ptr(en2)) > 0
    assert np.array_equal(a2, action) and np.array_equal(lp2, logp) and np.array_equal(en2, ent)
    # -- one window's complete episodes
    rs = np.random.RandomState(21)
    T, N = 70, 45                                             # more than 64 steps: both index-writing paths of the kernel
    done = (rs.rand(T, N) < 0.05).astype(np.uint8)
    act = np.where(rs.rand(T, N) < 0.1, -1, rs.randint(0, 180, (T, N))).astype(np.int32)
    done[:, 3] = 0                                            # a game without a finished episode contributes nothing
    index, count = np.zeros(T * N, np.int32), np.zeros(1, np.int32)
    assert L.sl_select_complete(ptr(done), ptr(act), T, N, ptr(index), ptr(count)) > 0
    want = []
    for g in range(N):
        ends = np.flatnonzero(done[:, g])
        if ends.size:
            want += [t * N + g for t in range(ends[-1] + 1) if act[t, g] >= 0]
    assert int(count[0]) == len(want) and index[:len(want)].tolist() == want
    # -- returns of a window, the carry flowing between windows
    reward = rs.randint(-9, 10, (T, N)).astype(np.int32)
    out, carry = np.zeros((T, N), np.float32), (rs.randn(N) * 3).astype(np.float32)
    carry0, gamma = carry.copy(), np.float32(0.95)
    assert L.sl_returns(ptr(reward), ptr(done), ptr(out), ptr(carry), gamma, T, N) > 0
    for g in range(N):
        q = carry0[g]
        for t in range(T - 1, -1, -1):
            if done[t, g]:
                q = np.float32(0)
            q = np.float32(reward[t, g]) + gamma * q
            assert out[t, g] == q
        assert carry[g] == q
    # -- seeding: CPython's random.seed(base + g) / an explicit seed per game
    n = 70
    mt, pos = np.zeros((n, 624), np.uint32), np.zeros(n, np.uint32)
    assert L.sl_seed(n, ptr(mt), ptr(pos), 1000, None) > 0
    for g in (0, 1, 63, 64, 69):
        r = oz.seeded_rng(1000 + g)
        assert np.array_equal(mt[g], np.ctypeslib.as_array(r.mt)) and int(pos[g]) == int(r.idx) == 624
    seeds = rs.randint(0, 2 ** 62, n).astype(np.uint64)
    assert L.sl_seed(n, ptr(mt), ptr(pos), 0, ptr(seeds)) > 0
    r = oz.seeded_rng(int(seeds[37]))
    assert np.array_equal(mt[37], np.ctypeslib.as_array(r.mt))
    # -- the partial sums, in workgroup order
    lay = (C.c_int * 8)()
    L.sl_layout(lay)
    TOTAL = lay[7]
    parts = 11
    partial = (rs.randn(parts, TOTAL) * 3).astype(np.float32)
    grad = np.zeros(TOTAL, np.float32)
    assert L.sl_reduce(ptr(partial), parts, ptr(grad)) > 0
    s = np.zeros(TOTAL, np.float32)
    for i in range(parts):
        s = s + partial[i]
    assert np.array_equal(grad, s)
