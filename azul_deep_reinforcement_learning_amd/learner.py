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
import collections

import torch
import torch.distributed as dist
import torch.nn.functional as F

ACTOR_COEFF, CRITIC_COEFF, ENTROPY_COEFF = 1.0, 0.5, 0.1        # agent.py:47-49


def complete_episode_samples(done):
    """[T, N] done flags of a window -> bool [T, N]: the steps whose episode ENDS inside the window, i.e. whose discounted
    return (nn_runner.py:70-76) is exact without bootstrapping.  Steps after a game's last `done` belong to an episode that
    ends in a LATER window: with a single-window rollout they are never trained (the next window overwrites the buffers) --
    A2CLearner.update_from_rollout on a PolicyRollout(ring >= 2) keeps them and trains them when their episode ends."""
    d = done != 0
    return torch.flip(torch.cummax(torch.flip(d, dims=[0]).to(torch.uint8), dim=0).values, dims=[0]).bool()


N_PARAMS = 82081            # ActorCritic(136, 180, 180)


class A2CLearner:
    def __init__(self, policy, learning_rate=3e-4, gamma=0.99, process_group=None, distributed=None, fused=None, stat_history=4096):
        """fused=True: gradients from the hand-written kernel azul_a2c_gradients (forward + backward on the f32 matrix cores, one
        launch + a deterministic reduction) instead of PyTorch autograd; needs CUDA tensors and the reference's network shape.
        Default: fused whenever that is possible."""
        self.policy = policy
        self.fused = fused
        self._ws = None
        self.gamma = gamma
        self.optimizer = self._own_adam = torch.optim.Adam(policy.parameters(), lr=learning_rate)        # agent.py:37
        self.fused_apply = True           # the fused path steps with azul_a2c_apply_adam while `optimizer` is the learner's own Adam
        self.group = process_group
        self.distributed = (dist.is_available() and dist.is_initialized()) if distributed is None else distributed
        # the last updates' loss terms (device scalars; bounded: the reference keeps one float per logged batch, agent.py:19-24)
        self.statistics = {k: collections.deque(maxlen=stat_history) for k in ("actor_loss", "critic_loss", "entropy_loss", "ac_loss", "samples")}
        self.updates = 0                  # optimiser steps requested so far (statistics entries ever appended)
        self.dropped_steps = None         # device int32[2] of update_from_rollout: [samples of the last update, steps lost from the ring]

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

    # ---- hand-written gradient / optimiser path -------------------------------------------------------------------------
    # One flat k-major vector holds the master copy of the parameters (layout of azul_a2c_gradients' gradient):
    #     w1t [136][360] | b1 [360] | w2c [180] | b2c [1] | pad | w2a_t [180][180] | b2a [180]
    # the policy / rollout kernels read views of it (kweights()), Adam's two moments use the same layout, and
    # azul_a2c_apply_adam writes every step into the flat copy AND into the eight nn.Linear tensors of the module.
    _OFF = {"w1t": (0, 136 * 360), "b1": (48960, 360), "w2c": (49320, 180), "b2c": (49500, 1), "w2a_t": (49502, 180 * 180), "b2a": (81902, 180)}

    def _can_fuse(self, obs):
        pol = self.policy
        return (obs.is_cuda and hasattr(pol, "critic_linear1") and pol.critic_linear1.in_features == 136 and
                pol.critic_linear1.out_features == 180 and pol.actor_linear2.out_features == 180 and pol.actor_linear1.out_features == 180)

    def _ensure_flat(self, dev):
        if self._ws is None or self._ws["flat"].device != dev:
            from . import _lib as L
            n = L.A2C_FLAT_SIZE
            self._ws = {"ws": torch.empty(256, n + 4, device=dev), "grad": torch.empty(n + 4, device=dev), "flat": torch.zeros(n, device=dev),
                        "m": torch.zeros(n, device=dev), "v": torch.zeros(n, device=dev),
                        "step": torch.zeros(1, dtype=torch.int32, device=dev)}        # Adam's step counter lives on the device
            self.sync_from_module()
        return self._ws

    def sync_from_module(self):
        """(Re)build the flat k-major master copy from the module's parameters (after load_state_dict or any outside edit)."""
        if self._ws is None:
            return
        pol, f = self.policy, self._ws["flat"]
        with torch.no_grad():
            f[0:48960].view(136, 360).copy_(torch.cat([pol.critic_linear1.weight, pol.actor_linear1.weight], dim=0).t())
            f[48960:49320].copy_(torch.cat([pol.critic_linear1.bias, pol.actor_linear1.bias]))
            f[49320:49500].copy_(pol.critic_linear2.weight.reshape(-1))
            f[49500:49501].copy_(pol.critic_linear2.bias)
            f[49502:81902].view(180, 180).copy_(pol.actor_linear2.weight.t())
            f[81902:82082].copy_(pol.actor_linear2.bias)

    def kweights(self, device=None):
        """Views of the flat master copy in the layouts the policy kernels read (None when the fused path is unavailable)."""
        dev = device if device is not None else next(self.policy.parameters()).device
        if torch.device(dev).type != "cuda" or not self._can_fuse(next(self.policy.parameters())) or self.fused is False:
            return None
        f = self._ensure_flat(torch.device(dev))["flat"]
        return {"w1t": f[0:48960].view(136, 360), "b1": f[48960:49320], "w2c": f[49320:49500], "b2c": f[49500:49501],
                "w2a_t": f[49502:81902].view(180, 180), "b2a": f[81902:82082]}

    def optimizer_state(self):
        """What a checkpoint needs: torch's Adam state, plus the fused path's moments and step when it is in use."""
        st = {"torch": self.optimizer.state_dict()}
        if self._ws is not None:
            st["fused"] = {"m": self._ws["m"].cpu(), "v": self._ws["v"].cpu(), "step": int(self._ws["step"].item())}
        return st

    def load_optimizer_state(self, st):
        self.optimizer.load_state_dict(st["torch"])
        if "fused" in st:
            ws = self._ensure_flat(next(self.policy.parameters()).device)
            ws["m"].copy_(st["fused"]["m"])
            ws["v"].copy_(st["fused"]["v"])
            ws["step"].fill_(int(st["fused"]["step"]))
        self.sync_from_module()

    def _fused_gradients(self, obs, mask, action, qvals, n_total=None, index=None, count=None, kweights=None, countf=None):
        """The flat gradient from azul_a2c_gradients (summed over the ranks); returns what _finish_fused needs: (n_dev, n_host) -- the
        global sample count as a device float[1] or as a host number.  Either `n_total` (host number; all rows are samples) or
        `index` + `count` (device selection of rows; `countf` = [count, 1 / max(count, 1)] as floats when the selection kernel
        already wrote them: single-process runs then need no arithmetic outside the kernels, no host round trip either way)."""
        import ctypes as C
        from . import _lib as L
        pol, dev = self.policy, obs.device
        ws = self._ensure_flat(dev)
        kw = self.kweights(dev)
        with torch.no_grad():
            obs = obs.contiguous().float()
            mask = mask.contiguous().to(torch.uint8)
            action = action.contiguous().to(torch.int32)
            qvals = qvals.contiguous().float()
            p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            inv_dev = None
            if index is not None and countf is not None and not self.distributed:
                n_dev, inv_dev, inv_host = countf[0:1], countf[1:2], 1.0
            elif index is not None:
                n_dev = count.to(torch.float32)
                if self.distributed:
                    dist.all_reduce(n_dev, group=self.group)
                inv_dev = (1.0 / n_dev.clamp(min=1.0)).contiguous()
                inv_host = 1.0
            else:
                n_dev = None
                inv_host = 1.0 / float(n_total)
            L.check(L.lib.azul_a2c_gradients(p(obs), p(mask), p(action), p(qvals), int(obs.shape[0]), C.c_float(inv_host),
                                             p(kw["w1t"]), p(kw["b1"]), p(kw["w2c"]), p(kw["b2c"]), p(kw["w2a_t"]), p(kw["b2a"]),
                                             p(pol.actor_linear2.weight), 136, 180, 180, p(ws["ws"]), 256, p(ws["grad"]), p(index), p(count),
                                             p(inv_dev), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            g = ws["grad"]
            if self.distributed:
                dist.all_reduce(g, group=self.group)                       # one flat bucket
        return (n_dev, None) if index is not None else (None, float(n_total))

    def _record(self, out):
        for k2, v in out.items():
            self.statistics[k2].append(v)
        self.updates += 1
        return out

    def _stat_slot(self):
        """Row of the statistics ring that the next update's loss terms go to (written by the optimiser kernel itself)."""
        ws = self._ws
        if "stats" not in ws:
            ws["stats"] = torch.zeros(self.statistics["ac_loss"].maxlen, 5, device=ws["grad"].device)
        return ws["stats"][self.updates % ws["stats"].shape[0]]

    def _finish_fused(self, n_dev, n_host):
        """Optimiser step: azul_a2c_apply_adam on the flat copy + module (the learner's own Adam), or -- when the caller installed
        another optimiser -- the flat gradient scattered into the parameters' .grad and that optimiser's step.  The update's loss
        terms (agent.py:51-58) are written by the same kernel into a row of a device-resident ring: no small launches follow."""
        import ctypes as C
        from . import _lib as L
        pol, ws = self.policy, self._ws
        g = ws["grad"]
        dev = g.device
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        row = self._stat_slot()
        if self.optimizer is self._own_adam and self.fused_apply:
            grp = self.optimizer.param_groups[0]
            # the step counter and the "any samples at all?" test stay on the device: an update without samples (a window in which
            # no episode ended) leaves parameters, moments and step untouched
            L.check(L.lib.azul_a2c_apply_adam(p(g), p(ws["flat"]), p(ws["m"]), p(ws["v"]), C.c_float(grp["lr"]), C.c_float(grp["betas"][0]),
                                              C.c_float(grp["betas"][1]), C.c_float(grp["eps"]), 0,
                                              p(pol.critic_linear1.weight), p(pol.critic_linear1.bias), p(pol.critic_linear2.weight),
                                              p(pol.critic_linear2.bias), p(pol.actor_linear1.weight), p(pol.actor_linear1.bias),
                                              p(pol.actor_linear2.weight), p(pol.actor_linear2.bias), p(ws["step"]), p(n_dev),
                                              C.c_float(n_host or 0.0), p(row), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        else:
            with torch.no_grad():
                gw1 = g[0:48960].view(136, 360)
                gb1 = g[48960:49320]
                pairs = [(pol.critic_linear1.weight, gw1[:, :180].t()), (pol.actor_linear1.weight, gw1[:, 180:].t()),
                         (pol.critic_linear1.bias, gb1[:180]), (pol.actor_linear1.bias, gb1[180:]),
                         (pol.critic_linear2.weight, g[49320:49500].view(1, 180)), (pol.critic_linear2.bias, g[49500:49501]),
                         (pol.actor_linear2.weight, g[49502:81902].view(180, 180).t()), (pol.actor_linear2.bias, g[81902:82082])]
                for prm, grad in pairs:
                    if prm.grad is None:
                        prm.grad = torch.empty_like(prm)
                    prm.grad.copy_(grad)
                # the loss terms the optimiser kernel would have written (this branch is the uncommon one: a few small launches)
                o = L.A2C_FLAT_SIZE
                nn = n_dev.reshape(1).to(torch.float32) if n_dev is not None else torch.full((1,), float(n_host), device=dev)
                row[0:3] = g[o:o + 3] / nn.clamp(min=1.0)
                row[3] = ACTOR_COEFF * row[0] + CRITIC_COEFF * row[1] + ENTROPY_COEFF * row[2]
                row[4] = nn[0]
            self.optimizer.step()
            self.sync_from_module()
        return self._record({"actor_loss": row[0], "critic_loss": row[1], "entropy_loss": row[2], "ac_loss": row[3], "samples": row[4]})

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
        return self._record({"actor_loss": stats[0], "critic_loss": stats[1], "entropy_loss": stats[2], "ac_loss": stats[3],
                             "samples": n_total.squeeze(0)})

    def update_from_rollout(self, rollout):
        """One update from a PolicyRollout after run_window().  With a ring of >= 2 windows (one part, fused path) EVERY step of
        EVERY episode is trained exactly once, like NNRunner.train (nn_runner.py:59-76): azul_select_episode_samples picks, per game,
        the steps from the first one not trained yet up to its last episode end inside the newest window -- including the opening
        steps recorded in earlier windows, whose returns run_window has chained backwards through the ring -- and
        azul_a2c_gradients reads them through the index list straight from the ring.  Steps whose episode outlives the ring
        (longer than (ring - 1) * window + 1 agent steps) are dropped and counted in `dropped_steps[1]`.
        Otherwise (ring == 1, several parts, PyTorch path): update_from_windows on the newest window."""
        ro = rollout
        tr0 = ro.traj[0]
        if ro.ring < 2 or ro.parts != 1 or not (self._can_fuse(tr0["obs"]) if self.fused is None else bool(self.fused)):
            return self.update_from_windows(ro.traj)
        import ctypes as C
        from . import _lib as L
        rg, T, N, D = ro.rings[0], ro.T, ro.h, ro.ring
        dev = tr0["obs"].device
        R = D * T
        st = getattr(self, "_ring", None)
        if st is None or st["key"] != (id(ro), R, N):
            st = self._ring = {"key": (id(ro), R, N), "index": torch.empty(R * N, dtype=torch.int32, device=dev),
                               "count": torch.zeros(2, dtype=torch.int32, device=dev), "countf": torch.zeros(2, device=dev),
                               "pending": torch.zeros(N, dtype=torch.int32, device=dev),
                               "scratch": torch.empty(3 * N + (N + 3) // 4, dtype=torch.int32, device=dev), "offset": 0}
            # this learner's books start with the window that was just played (earlier windows -- warm-up -- are nobody's samples);
            # the clock is the rollout's own (absolute step s lives in ring slot s % R), shifted down by whole rings when it grows
            st["pending"].fill_((ro.windows_played - 1) * T)
            self.dropped_steps = st["count"]
        played = ro.windows_played * T - st["offset"]
        if played > (1 << 30):                            # keep the absolute step counters inside int32: rebase by whole rings
            shift = (played - R) // R * R
            st["pending"].sub_(shift)
            st["offset"] += shift
            played -= shift
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(L.lib.azul_select_episode_samples(p(rg["done"]), p(rg["action"]), T, D, N, int(played), p(st["pending"]), p(st["index"]),
                                                  p(st["count"]), p(st["countf"]), p(st["scratch"]),
                                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return self._finish_fused(*self._fused_gradients(
            rg["obs"][:R].reshape(R * N, -1), rg["mask"][:R].reshape(R * N, -1), rg["action"].reshape(-1), rg["returns"].reshape(-1),
            index=st["index"], count=st["count"][:1], countf=st["countf"]))

    def ring_state(self):
        """The books of update_from_rollout's ring selection (for a checkpoint): per game the first absolute step not trained yet, the
        clock offset and the dropped-step counter; None before the first ring update."""
        st = getattr(self, "_ring", None)
        if st is None:
            return None
        return {"pending": st["pending"].cpu(), "offset": int(st["offset"]), "count": st["count"].cpu(), "R": st["key"][1], "N": st["key"][2]}

    def load_ring_state(self, rollout, state):
        """Restore ring_state() for `rollout` (whose ring buffers and windows_played the caller has restored), or -- state None --
        forget the books: the next update_from_rollout starts them with the window played just before it."""
        self._ring = None
        self.dropped_steps = None
        if state is None or rollout.ring < 2 or rollout.parts != 1:
            return
        R, N = rollout.ring * rollout.T, rollout.h
        if (state["R"], state["N"]) != (R, N):
            raise ValueError("ring state was written for %d slots x %d games" % (state["R"], state["N"]))
        dev = rollout.device
        self._ring = {"key": (id(rollout), R, N), "index": torch.empty(R * N, dtype=torch.int32, device=dev),
                      "count": state["count"].to(dev).clone(), "countf": torch.zeros(2, device=dev),
                      "pending": state["pending"].to(dev).clone(),
                      "scratch": torch.empty(3 * N + (N + 3) // 4, dtype=torch.int32, device=dev), "offset": int(state["offset"])}
        self.dropped_steps = self._ring["count"]

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
