"""A2C learner for batched rollouts: the reference's ``Agent.update`` (azulnet/agent.py:39-62) and the batch assembly of
``NNRunner.train`` (azulnet/nn_runner.py:57-79), restated for whole trajectory windows and for data-parallel ranks.

Loss, exactly as the reference writes it (agent.py:45-58), over the n samples of a batch:
    advantage    = qvals - values
    actor_loss   = mean(-log_prob(action) * advantage)          (advantage NOT detached, like the reference)
    critic_loss  = mean(advantage^2)
    entropy_loss = mean(-mean(log p over legal actions))        (nn_runner.py:36-40; coefficient +0.1, sic)
    ac_loss      = 1 * actor_loss + 0.5 * critic_loss + 0.1 * entropy_loss
then one Adam step (lr 3e-4).  The reference evaluates the network once per agent step while it plays and keeps the autograd
graph; here the rollout runs without autograd and the learner re-evaluates the network on the recorded (observation, mask,
action) samples in one batched forward, which yields the same values, log-probabilities and gradients.

Data parallel (one process per GPU): every rank holds the samples of its own games; the loss terms are SUMS over the local
samples divided by the GLOBAL sample count, and gradients are summed over ranks (one flat all-reduce over RCCL), so the
update equals the single-process update on the union of all ranks' samples whatever the per-rank counts are.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

ACTOR_COEFF, CRITIC_COEFF, ENTROPY_COEFF = 1.0, 0.5, 0.1        # agent.py:47-49


def complete_episode_samples(done):
    """[T, N] done flags of a window -> bool [T, N]: the steps whose episode ENDS inside the window, i.e. whose discounted
    return (nn_runner.py:70-76) is exact without bootstrapping.  Steps after a game's last `done` are left for a later window."""
    d = done != 0
    return torch.flip(torch.cummax(torch.flip(d, dims=[0]).to(torch.uint8), dim=0).values, dims=[0]).bool()


N_PARAMS = 82081            # ActorCritic(136, 180, 180)


class A2CLearner:
    def __init__(self, policy, learning_rate=3e-4, gamma=0.99, process_group=None, distributed=None, fused=None):
        """fused=True: gradients from the hand-written kernel azul_a2c_gradients (forward + backward on the f32 matrix cores, one
        launch + a deterministic reduction) instead of PyTorch autograd; needs CUDA tensors and the reference's network shape.
        Default: fused whenever that is possible."""
        self.policy = policy
        self.fused = fused
        self._ws = None
        self.gamma = gamma
        self.optimizer = torch.optim.Adam(policy.parameters(), lr=learning_rate)        # agent.py:37
        self.group = process_group
        self.distributed = (dist.is_available() and dist.is_initialized()) if distributed is None else distributed
        self.statistics = {"actor_loss": [], "critic_loss": [], "entropy_loss": [], "ac_loss": [], "samples": []}

    def loss_terms(self, obs, mask, action, qvals, weight=None, n_total=None):
        """obs [n,136] f32, mask [n,180] bool/u8, action [n] int, qvals [n] -> the four loss terms of agent.py:51-57.
        `weight` [n] (0/1) drops samples without changing shapes; `n_total` is the divisor (default: the local count)."""
        pol = self.policy
        legal = mask.bool()
        if hasattr(pol, "critic_linear1") and hasattr(pol, "actor_linear1"):
            # same arithmetic as forward_critic / forward_actor (model.py:22-41) with both first layers in ONE GEMM (and one
            # weight-gradient GEMM in the backward pass): autograd splits the concatenated gradient back onto the two modules
            H = pol.critic_linear1.out_features
            w1 = torch.cat([pol.critic_linear1.weight, pol.actor_linear1.weight], dim=0)
            b1 = torch.cat([pol.critic_linear1.bias, pol.actor_linear1.bias])
            hidden = F.relu(F.linear(obs, w1, b1))
            values = pol.critic_linear2(hidden[:, :H]).squeeze(1)
            logits = pol.actor_linear2(hidden[:, H:]).masked_fill(~legal, float("-inf"))
            logp = F.log_softmax(logits, dim=1)
        else:
            values = pol.forward_critic(obs).squeeze(1)
            _, logp = pol.forward_actor(obs, legal)
        log_prob = logp.gather(1, action.long().clamp(min=0).unsqueeze(1)).squeeze(1)                 # nn_runner.py:32
        entropy = -(torch.where(legal, logp, torch.zeros_like(logp)).sum(dim=1) / legal.sum(dim=1).clamp(min=1))   # :36-40
        advantage = qvals.to(values.dtype) - values
        w = torch.ones_like(values) if weight is None else weight.to(values.dtype)
        n = w.sum() if n_total is None else n_total
        actor_loss = (-log_prob * advantage * w).sum() / n
        critic_loss = (advantage.pow(2) * w).sum() / n
        entropy_loss = (entropy * w).sum() / n
        ac_loss = ACTOR_COEFF * actor_loss + CRITIC_COEFF * critic_loss + ENTROPY_COEFF * entropy_loss
        return actor_loss, critic_loss, entropy_loss, ac_loss

    # ---- hand-written gradient path -------------------------------------------------------------------------------------
    def _can_fuse(self, obs):
        pol = self.policy
        return (obs.is_cuda and hasattr(pol, "critic_linear1") and pol.critic_linear1.in_features == 136 and
                pol.critic_linear1.out_features == 180 and pol.actor_linear2.out_features == 180 and pol.actor_linear1.out_features == 180)

    def _fused_gradients(self, obs, mask, action, qvals, n_total=None, index=None, count=None, kweights=None):
        """Fills every parameter's .grad from azul_a2c_gradients; returns (actor, critic, entropy, samples): loss sums / n_total.
        Either `n_total` (host number; all rows are samples) or `index` + `count` (device selection of rows; the global count
        is then formed on the device, no host round trip).  `kweights`: the k-major weight copies a PolicyRollout keeps."""
        import ctypes as C
        from . import _lib as L
        pol, dev = self.policy, obs.device
        if self._ws is None or self._ws["ws"].device != dev:
            self._ws = {"ws": torch.empty(256, N_PARAMS + 4, device=dev), "grad": torch.empty(N_PARAMS + 4, device=dev)}
        ws = self._ws
        with torch.no_grad():
            if kweights is None:
                w1t = torch.cat([pol.critic_linear1.weight, pol.actor_linear1.weight], dim=0).t().contiguous()
                b1 = torch.cat([pol.critic_linear1.bias, pol.actor_linear1.bias]).contiguous()
                w2c = pol.critic_linear2.weight.reshape(-1).contiguous()
                w2a_t = pol.actor_linear2.weight.t().contiguous()
            else:
                w1t, b1, w2c, w2a_t = kweights["w1t"], kweights["b1"], kweights["w2c"], kweights["w2a_t"]
            w2a = pol.actor_linear2.weight.contiguous()
            obs = obs.contiguous().float()
            mask = mask.contiguous().to(torch.uint8)
            action = action.contiguous().to(torch.int32)
            qvals = qvals.contiguous().float()
            p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            inv_dev = None
            if index is not None:
                n_dev = count.to(torch.float32)
                if self.distributed:
                    dist.all_reduce(n_dev, group=self.group)
                inv_dev = (1.0 / n_dev.clamp(min=1.0)).contiguous()
                inv_host = 1.0
            else:
                n_dev = None
                inv_host = 1.0 / float(n_total)
            L.check(L.lib.azul_a2c_gradients(p(obs), p(mask), p(action), p(qvals), int(obs.shape[0]), C.c_float(inv_host),
                                             p(w1t), p(b1), p(w2c), p(pol.critic_linear2.bias), p(w2a_t), p(pol.actor_linear2.bias), p(w2a),
                                             136, 180, 180, p(ws["ws"]), 256, p(ws["grad"]), p(index), p(count), p(inv_dev),
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            g = ws["grad"]
            if self.distributed:
                dist.all_reduce(g, group=self.group)                       # one flat bucket: 82 085 floats
            o = 0
            gw1 = g[o:o + 136 * 360].view(136, 360); o += 136 * 360
            gb1 = g[o:o + 360]; o += 360
            gw2c = g[o:o + 180]; o += 180
            gb2c = g[o:o + 1]; o += 1
            gw2a = g[o:o + 180 * 180].view(180, 180); o += 180 * 180
            gb2a = g[o:o + 180]; o += 180
            pairs = [(pol.critic_linear1.weight, gw1[:, :180].t()), (pol.actor_linear1.weight, gw1[:, 180:].t()),
                     (pol.critic_linear1.bias, gb1[:180]), (pol.actor_linear1.bias, gb1[180:]),
                     (pol.critic_linear2.weight, gw2c.view(1, 180)), (pol.critic_linear2.bias, gb2c),
                     (pol.actor_linear2.weight, gw2a.t()), (pol.actor_linear2.bias, gb2a)]
            for prm, grad in pairs:
                if prm.grad is None:
                    prm.grad = torch.empty_like(prm)
                prm.grad.copy_(grad)
            if index is not None:
                sums = g[o:o + 3] * inv_dev
                samples = n_dev.squeeze(0)
            else:
                sums = g[o:o + 3] / float(n_total)
                samples = torch.as_tensor(float(n_total), device=dev)
        return sums[0], sums[1], sums[2], samples

    def _finish_fused(self, a, c, e, samples):
        loss = ACTOR_COEFF * a + CRITIC_COEFF * c + ENTROPY_COEFF * e
        self.optimizer.step()
        out = {"actor_loss": a, "critic_loss": c, "entropy_loss": e, "ac_loss": loss, "samples": samples}
        for k2, v in out.items():
            self.statistics[k2].append(v)
        return out

    def update(self, obs, mask, action, qvals, weight=None):
        """One optimiser step on the given samples (this rank's share when distributed).  Returns the loss terms (global means).
        Rows with weight 0 are compacted away first (the network is not evaluated on them)."""
        if weight is not None:
            idx = torch.nonzero(weight > 0).squeeze(1)
            obs, mask, action, qvals = obs.index_select(0, idx), mask.index_select(0, idx), action.index_select(0, idx), qvals.index_select(0, idx)
            weight = None
        n_local = torch.full((1,), float(obs.shape[0]), dtype=torch.float32, device=obs.device)
        n_total = n_local.clone()
        if self.distributed:
            dist.all_reduce(n_total, group=self.group)
        use_fused = self._can_fuse(obs) if self.fused is None else bool(self.fused)
        if use_fused:
            return self._finish_fused(*self._fused_gradients(obs, mask, action, qvals, n_total=float(n_total)))
        # rows without a legal action (stuck games) carry no sample
        legal_any = mask.bool().any(dim=1)
        w = legal_any.to(torch.float32) if weight is None else weight.to(torch.float32) * legal_any
        a, c, e, loss = self.loss_terms(obs, mask, action, qvals, w, n_total.squeeze(0))
        self.optimizer.zero_grad(set_to_none=False)
        loss.backward()
        stats = torch.stack([a.detach(), c.detach(), e.detach(), loss.detach()])
        if self.distributed:
            params = [p for p in self.policy.parameters() if p.grad is not None]
            flat = torch.cat([p.grad.reshape(-1) for p in params] + [stats])           # one bucket: ~86k floats
            dist.all_reduce(flat, group=self.group)
            o = 0
            for p in params:
                p.grad.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
            stats = flat[o:]
        self.optimizer.step()
        out = {"actor_loss": stats[0], "critic_loss": stats[1], "entropy_loss": stats[2], "ac_loss": stats[3], "samples": n_total.squeeze(0)}
        for k2, v in out.items():
            self.statistics[k2].append(v)
        return out

    def update_from_windows(self, trajectories, complete_only=True, kweights=None):
        """`trajectories`: the per-part dicts PolicyRollout.run_window returns (opponent="random": every record is one agent
        step).  Uses the steps whose episode finished inside the window (exact Monte-Carlo returns, the reference's qvals).
        With one part on the GPU the whole update stays on the device: azul_select_complete_samples picks the steps,
        azul_a2c_gradients reads them through the index list -- no compaction copies, no host round trip."""
        tr0 = trajectories[0]
        if len(trajectories) == 1 and complete_only and (self._can_fuse(tr0["obs"]) if self.fused is None else bool(self.fused)):
            import ctypes as C
            from . import _lib as L
            T, N = tr0["action"].shape
            dev = tr0["obs"].device
            if getattr(self, "_sel", None) is None or self._sel[0].numel() != T * N or self._sel[0].device != dev:
                self._sel = (torch.empty(T * N, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
            index, count = self._sel
            L.check(L.lib.azul_select_complete_samples(C.c_void_p(tr0["done"].data_ptr()), C.c_void_p(tr0["action"].data_ptr()), T, N,
                                                       C.c_void_p(index.data_ptr()), C.c_void_p(count.data_ptr()),
                                                       C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            return self._finish_fused(*self._fused_gradients(
                tr0["obs"][:T].reshape(T * N, -1), tr0["mask"][:T].reshape(T * N, -1), tr0["action"].reshape(-1), tr0["returns"].reshape(-1),
                index=index, count=count, kweights=kweights))
        obs, mask, action, ret = [], [], [], []
        for tr in trajectories:
            T = tr["action"].shape[0]
            keep = complete_episode_samples(tr["done"]) if complete_only else torch.ones_like(tr["done"], dtype=torch.bool)
            keep = keep & (tr["action"] >= 0)
            idx = torch.nonzero(keep.reshape(-1)).squeeze(1)               # compact per part: the big buffers are read once
            obs.append(tr["obs"][:T].reshape(-1, tr["obs"].shape[-1]).index_select(0, idx))
            mask.append(tr["mask"][:T].reshape(-1, tr["mask"].shape[-1]).index_select(0, idx))
            action.append(tr["action"].reshape(-1).index_select(0, idx))
            ret.append(tr["returns"].reshape(-1).index_select(0, idx))
        one = len(trajectories) == 1
        return self.update(obs[0] if one else torch.cat(obs), mask[0] if one else torch.cat(mask),
                           action[0] if one else torch.cat(action), ret[0] if one else torch.cat(ret))
