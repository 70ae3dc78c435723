"""Policy-driven self-play on one MI355X (BASELINE configs[2]): the batched counterpart of the reference's
``NNRunner.run_episode`` (azulnet/nn_runner.py:17-47) and of the action sampling in ``Agent.get_ac_output``
(azulnet/agent.py:64-81).

`persistent=True` (what the benches use): ONE launch per window and part -- azul_batch_policy_rollout keeps every game in
registers for the whole window and interleaves network, sampling and env step inside the kernel.  Otherwise, per move and
part of the batch (each part has its own HIP stream) -- two launches:
    azul_policy_forward      the whole ActorCritic forward on the f32 matrix cores (hidden = relu(obs @ [critic_linear1 |
                             actor_linear1]), value, logits) + masked softmax + categorical sample + log-prob + entropy;
                             value / action / log-prob / entropy land straight in trajectory slot t
    azul_batch_policy_step   Azul.step + reward + done + auto-reset + NEXT obs / mask / player, written into slot t+1
                             (opponent="random": azul_batch_agent_step, the reference's training setup)
Nothing is copied.  A window of `T` moves is captured once into a HIP graph per part and replayed (the Philox step counter
lives in device memory and is advanced by the forward launch itself); results are identical with `use_graph=False`.
`fused_mlp=False` runs the network as PyTorch GEMMs (rocBLAS/hipBLASLt) + azul_policy_head (any network shape);
`fused_head=False` keeps the all-PyTorch sampling path (torch.multinomial) for comparison.

Per (move t, game g) the record holds what the reference's run_episode keeps per agent step (C1 in SURVEY.md 8a):
observation, legal mask, action, reward, done, value, log-prob of the action and the entropy term
`-mean(log p over legal actions)` (nn_runner.py:36-40), plus the player who moved and the discounted returns.
"""
import ctypes as C

import torch

from . import _lib as L
from .batch import BatchedAzul


def _p(t):
    return C.c_void_p(t.data_ptr())


class PolicyRollout:
    def __init__(self, policy, n_games=4096, parts=1, rules={"first_player": "Random", "tile_pool": "Lid"}, seed_base=0,
                 device=None, window=32, use_graph=True, fused_head=True, sample_seed=0x5EED, opponent=None, fused_mlp=True, persistent=False,
                 action_selection="Distribution", kweights=None, game_id_base=None, ring=1, opponent_selection="Distribution",
                 opponent_seed=None, opponent_trace=0, move_limit=0):
        """opponent=None: the policy moves for both players (flat self-play, one record per env move).
        opponent="random": the reference's training setup -- the policy is player 1 of GameRunner, the opponent a RandomAgent
        inside the env step (game_runner.py:43-47); one record per AGENT step, observations from the agent's perspective.
        opponent=<a second BatchedActorCritic / any module with the reference's four layers>: GameRunner(opponent=Agent(...))
        (game_runner.py:27-30; scripts/run_batch.py:6-10) -- every opponent_move(), i.e. the opponent's replies, player 1's FORCED moves
        (:46) and the opening moves of reset() (:84-85), is sampled from that net's forward_actor on the observation from the mover's
        perspective (:38; agent.py:73-81), `opponent_selection` = its action_selection; records as with "random".  Its weights are copied
        at construction; set_opponent() installs new ones (e.g. a frozen past self of the policy being trained).  persistent=True plays it
        inside the window kernel (matrix phases on the second weight set while any game of a workgroup owes a reply); otherwise one launch
        per cut of the protocol and one host synchronisation per reply round (no HIP graph).  `opponent_trace` = R > 0 also records the
        opponent's answers: opp_action / opp_logp [T][R][N], opp_replies [T][N].
        `move_limit` > 0 (beyond the reference, off by default): cut an episode at the first end of a round with move_counter >= move_limit
        (done = 3) -- under the reference's rules some games never end and would keep their slot for ever (BatchedAzul.set_move_limit).
        `seed_base` / `game_id_base`: game i of this rollout is global game game_id_base + i (default: seed_base, so that a rank
        passes the id of its first game once); its CPython stream is random.seed(seed_base + i) and its sampling stream is
        Philox(sample_seed, step, global id) -- both independent of how the games are sharded over GPUs or split into parts.
        `ring` (persistent=True only): the trajectory buffers hold the last `ring` windows (a ring of ring * window time slots);
        run_window fills the next window of the ring and re-chains the discounted returns backwards through the older windows, so
        that the opening steps of an episode that ends in a LATER window get their exact Monte-Carlo return too
        (A2CLearner.update_from_rollout trains every step of every episode exactly once, like NNRunner.train)."""
        assert n_games % parts == 0
        self.opp_policy = None
        if opponent is not None and not isinstance(opponent, str):
            self.opp_policy, opponent = opponent, "net"
        assert opponent in (None, "random", "net")
        self.opponent = opponent
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.policy = policy.to(self.device).eval()
        self.n, self.parts, self.h, self.T = n_games, parts, n_games // parts, window
        self.fused_head = fused_head
        # the one-launch forward (azul_policy_forward) is compiled for the reference's ActorCritic(136, 180, hidden 180)
        self.fused_mlp = bool(fused_mlp and fused_head and policy.critic_linear1.in_features == L.OBS_SIZE and
                              policy.critic_linear1.out_features == 180 and policy.actor_linear2.out_features == L.NUM_ACTIONS)
        # persistent=True: the whole window runs in ONE launch per part (azul_batch_policy_rollout); same results
        self.persistent = bool(persistent and self.fused_mlp)
        self.ring = int(ring) if self.persistent else 1
        assert self.ring >= 1
        self.windows_played = 0
        # Agent.get_ac_output's two modes (agent.py:64-72): sample from the masked softmax, or take its first maximum
        assert action_selection in ("Distribution", "Max") and (fused_head or action_selection == "Distribution")
        self.action_selection = action_selection
        self.sample_seed = L.POLICY_ARGMAX if action_selection == "Max" else int(sample_seed)
        self.game_id_base = int(seed_base if game_id_base is None else game_id_base) & 0xFFFFFFFF
        assert opponent_selection in ("Distribution", "Max")
        self.opponent_seed = L.POLICY_ARGMAX if opponent_selection == "Max" else \
            (int(opponent_seed) if opponent_seed is not None else (int(sample_seed) ^ 0x4F50504F4E454E54)) & 0xFFFFFFFFFFFFFFFF
        self.opp_slots = int(opponent_trace) if self.opponent == "net" else 0
        self.envs, self.streams, self.work, self.traj, self.graphs = [], [], [], [], []
        # kweights: k-major weight tensors owned by someone else (A2CLearner.kweights(): views of its flat master copy, kept current
        # by the optimiser kernel) -- then nothing is copied here and refresh_weights() has nothing to do
        self._external_kweights = kweights is not None
        if kweights is not None:
            self.H = policy.critic_linear1.out_features
            self.w1t, self.b1, self.w2c, self.w2a_t = kweights["w1t"], kweights["b1"], kweights["w2c"], kweights["w2a_t"]
            self.w2c_t = self.w2c.view(-1, 1)
        self.refresh_weights()
        if self.opponent == "net":
            assert self.fused_mlp, "the network opponent runs on the library's forward (ActorCritic(136, 180, hidden 180))"
            self.opp_policy = self.opp_policy.to(self.device).eval()
            self.set_opponent(self.opp_policy)
        d, h, T = self.device, self.h, window
        for p in range(parts):
            env = BatchedAzul(h, rules=rules, device=d)
            env.seed(seed_base + p * h)                            # seeds follow the global game id
            env.set_id_base(self.game_id_base + p * h)             # ... and so does the sampler's Philox key
            if move_limit:
                env.set_move_limit(move_limit)
            env.runner_init()                                      # GameRunner()
            if opponent == "random":
                env.reset()                                        # GameRunner.reset(): the opponent opens when it starts
            elif opponent == "net":
                pass                                               # ... with the network opponent: below, once the work buffers exist
            else:
                env.runner_init()                                  # reset() without pre-moves (flat self-play)
            self.envs.append(env)
            self.streams.append(torch.cuda.Stream(device=d))
            R = self.ring * T
            rg = {"obs": torch.zeros(R + 1, h, L.OBS_SIZE, device=d), "mask": torch.zeros(R + 1, h, L.NUM_ACTIONS, dtype=torch.uint8, device=d),
                  "player": torch.zeros(R + 1, h, dtype=torch.uint8, device=d),
                  "action": torch.zeros(R, h, dtype=torch.int32, device=d), "reward": torch.zeros(R, h, dtype=torch.int32, device=d),
                  "done": torch.zeros(R, h, dtype=torch.uint8, device=d),
                  "value": torch.zeros(R, h, 1, device=d), "log_prob": torch.zeros(R, h, device=d), "entropy": torch.zeros(R, h, device=d),
                  "returns": torch.zeros(R, h, device=d), "carry": torch.zeros(h, device=d)}
            if self.opponent == "net":
                rg["opp_replies"] = torch.zeros(R, h, dtype=torch.uint8, device=d)
                if self.opp_slots:
                    rg["opp_action"] = torch.full((R, self.opp_slots, h), -1, dtype=torch.int32, device=d)
                    rg["opp_logp"] = torch.zeros(R, self.opp_slots, h, device=d)
            self.rings = getattr(self, "rings", [])
            self.rings.append(rg)
            t = self._window_views(rg, self.ring - 1)             # the "previous" window: its slot T seeds the first window
            w = {"hidden": torch.zeros(h, 2 * self.H, device=d), "logits": torch.zeros(h, L.NUM_ACTIONS, device=d),
                 "status": torch.zeros(h, dtype=torch.uint8, device=d),
                 "counter": torch.tensor([0, 0], dtype=torch.int64, device=d)}     # [0] Philox step counter, [1] launch ticket
            if self.opponent == "net":
                w["net"] = env.net_state()
                w["scratch_f"] = torch.zeros(3, h, device=d)       # the opponent forward's value / entropy (not recorded) and untraced log-prob
            self.traj.append(t)
            self.work.append(w)
            with torch.cuda.stream(self.streams[p]):
                if self.opponent == "net":
                    self._net_reset(p)                             # GameRunner.reset(): the network opponent opens when it starts
                env.observe_all(self._persp(), t["obs"][T], t["mask"][T], t["player"][T])     # becomes slot 0 of the first window
        torch.cuda.synchronize(d)
        self.use_graph = use_graph and not self.persistent and self.opponent != "net"     # one launch per window needs no graph; reply rounds are data-dependent
        self.graph_error = None
        if self.use_graph:
            try:
                self._capture()
            except Exception as e:          # capture is a launch-overhead optimisation only
                self.graph_error = repr(e)
                self.graphs = []
                self.use_graph = False
                torch.cuda.synchronize(d)

    def _window_views(self, rg, w):
        """Views of window `w` of a ring: T + 1 slots of obs / mask / player (slot T = the state after the window), T of the rest."""
        T = self.T
        lo = w * T
        out = {k: rg[k][lo:lo + T + 1] for k in ("obs", "mask", "player")}
        out.update({k: rg[k][lo:lo + T] for k in ("action", "reward", "done", "value", "log_prob", "entropy", "returns", "opp_replies", "opp_action",
                                                  "opp_logp") if k in rg})
        return out

    def _persp(self):
        return 0 if self.opponent in ("random", "net") else L.PERSP_CURRENT     # NNRunner observes with perspective 0 (game_runner.py:56)

    def set_opponent(self, policy_or_state_dict):
        """Install the network opponent's weights (k-major copies, updated IN PLACE): a module with the reference's four layers or its
        state_dict -- e.g. `ro.set_opponent(policy)` every so many updates trains against a frozen past self."""
        sd = policy_or_state_dict.state_dict() if hasattr(policy_or_state_dict, "state_dict") else policy_or_state_dict
        with torch.no_grad():
            g = lambda k: sd[k].detach().to(self.device, torch.float32)
            fresh = {"ow1t": torch.cat([g("critic_linear1.weight"), g("actor_linear1.weight")], dim=0).t(),
                     "ob1": torch.cat([g("critic_linear1.bias"), g("actor_linear1.bias")]), "ow2c": g("critic_linear2.weight").reshape(-1),
                     "ob2c": g("critic_linear2.bias").reshape(-1), "ow2a_t": g("actor_linear2.weight").t(), "ob2a": g("actor_linear2.bias")}
            for name, v in fresh.items():
                if hasattr(self, name):
                    getattr(self, name).copy_(v)
                else:
                    setattr(self, name, v.contiguous().clone())

    def _opp_forward(self, p, j, logp_out):
        """The opponent's get_a_output (agent.py:73-81) for the games of part p that owe an opponent_move(): forward_actor of its net on
        what net_step_* left in work["net"], reply j of the step sampled with Philox key opponent_seed + j at the step's counter."""
        w = self.work[p]
        net, sc = w["net"], w["scratch_f"]
        key = self.opponent_seed if self.opponent_seed == L.POLICY_ARGMAX else (self.opponent_seed + j) & 0xFFFFFFFFFFFFFFFF
        L.check(L.lib.azul_policy_forward(_p(net["obs"]), _p(net["mask"]), _p(self.ow1t), _p(self.ob1), _p(self.ow2c), _p(self.ob2c), _p(self.ow2a_t),
                                          _p(self.ob2a), L.OBS_SIZE, self.H, L.NUM_ACTIONS, key, 0xFFFFFFFFFFFFFFFF, _p(w["counter"]), 0, self.h,
                                          self.game_id_base + p * self.h, _p(sc[0]), _p(net["action"]), _p(logp_out), _p(sc[1]), None,
                                          C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _net_reset(self, p):
        """GameRunner.reset() with the network opponent for every game of part p (game_runner.py:76-85); its opening moves draw at the
        counter before the first step."""
        env, w = self.envs[p], self.work[p]
        env.net_reset_begin(w["net"], w["status"])
        j = 0
        while int(w["net"]["owing"].item()) > 0:
            self._opp_forward(p, j, w["scratch_f"][2])
            env.net_step_reply(w["net"]["action"], w["net"], None, None, w["status"])
            j += 1

    def refresh_weights(self):
        """(Re)build the fused first-layer weights from the policy's parameters -- call after every optimiser step.  The
        staging tensors are updated IN PLACE: a captured HIP graph keeps reading the same addresses."""
        pol = self.policy
        if self._external_kweights:
            return
        with torch.no_grad():
            self.H = pol.critic_linear1.out_features
            fresh = {"w1t": torch.cat([pol.critic_linear1.weight, pol.actor_linear1.weight], dim=0).t(),
                     "b1": torch.cat([pol.critic_linear1.bias, pol.actor_linear1.bias]),
                     "w2c_t": pol.critic_linear2.weight.t(), "w2a_t": pol.actor_linear2.weight.t(),
                     "w2c": pol.critic_linear2.weight.reshape(-1)}
            for name, v in fresh.items():
                if hasattr(self, name):
                    getattr(self, name).copy_(v)
                else:
                    setattr(self, name, v.contiguous().clone())

    def kweights(self):
        """The k-major weight copies (refresh_weights keeps them current): the learner's gradient kernel reads the same layouts."""
        return {"w1t": self.w1t, "b1": self.b1, "w2c": self.w2c, "w2a_t": self.w2a_t}

    # one move of one part, enqueued on the current stream
    def _move(self, p, t):
        env, tr, w = self.envs[p], self.traj[p], self.work[p]
        obs, mask, H = tr["obs"][t], tr["mask"][t], self.H
        pol = self.policy
        if self.fused_mlp:                                  # whole forward + head: one launch on the f32 matrix cores
            L.check(L.lib.azul_policy_forward(_p(obs), _p(mask), _p(self.w1t), _p(self.b1), _p(self.w2c), _p(pol.critic_linear2.bias),
                                              _p(self.w2a_t), _p(pol.actor_linear2.bias), L.OBS_SIZE, H, L.NUM_ACTIONS,
                                              self.sample_seed, 0, _p(w["counter"]), 1, self.h, self.game_id_base + p * self.h,
                                              _p(tr["value"][t]), _p(tr["action"][t]),
                                              _p(tr["log_prob"][t]), _p(tr["entropy"][t]), None,
                                              C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            self._env_step(p, t)
            return
        with torch.no_grad():
            torch.addmm(self.b1, obs, self.w1t, out=w["hidden"])
            w["hidden"].relu_()
            torch.addmm(pol.critic_linear2.bias, w["hidden"][:, :H], self.w2c_t, out=tr["value"][t])          # agent.py:66
            torch.addmm(pol.actor_linear2.bias, w["hidden"][:, H:], self.w2a_t, out=w["logits"])               # agent.py:67
            if self.fused_head:
                L.check(L.lib.azul_policy_head(_p(w["logits"]), _p(mask), self.sample_seed, 0, _p(w["counter"]), self.h, self.game_id_base + p * self.h,
                                               _p(tr["action"][t]), _p(tr["log_prob"][t]), _p(tr["entropy"][t]),
                                               C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
                w["counter"][:1].add_(1)
            else:
                legal = mask.bool()
                logits = w["logits"].masked_fill(~legal, float("-inf"))
                logp = torch.log_softmax(logits, dim=1)
                any_legal = legal.any(dim=1)
                safe = torch.where(any_legal.unsqueeze(1), logp.exp(), torch.full_like(logp, 1.0 / logp.shape[1]))
                action = torch.multinomial(safe, 1).squeeze(1)                                               # agent.py:69
                tr["log_prob"][t].copy_(logp.gather(1, action.unsqueeze(1)).squeeze(1))                       # nn_runner.py:32
                tr["entropy"][t].copy_(-(torch.where(legal, logp, torch.zeros_like(logp)).sum(dim=1) / legal.sum(dim=1).clamp(min=1)))
                tr["action"][t].copy_(torch.where(any_legal, action, torch.full_like(action, -1)).to(torch.int32))
        self._env_step(p, t)

    def _env_step(self, p, t):
        env, tr, w = self.envs[p], self.traj[p], self.work[p]
        if self.opponent == "net":
            # GameRunner.step cut at its opponent_move() calls: the agent's move, then reply rounds while any game owes one (one host
            # synchronisation per round: this is the per-move reference structure, the window kernel is the fast path)
            net = w["net"]
            env.net_step_begin(tr["action"][t], net, tr["reward"][t], tr["done"][t], w["status"])
            j = 0
            while int(net["owing"].item()) > 0:
                self._opp_forward(p, j, tr["opp_logp"][t][j] if j < self.opp_slots else w["scratch_f"][2])
                if j < self.opp_slots:
                    tr["opp_action"][t][j].copy_(net["action"])
                env.net_step_reply(net["action"], net, tr["reward"][t], tr["done"][t], w["status"])
                j += 1
            tr["opp_replies"][t].copy_(net["replies"])
            env.observe_all(0, tr["obs"][t + 1], tr["mask"][t + 1], tr["player"][t + 1])
        elif self.opponent == "random":
            env.agent_step(tr["action"][t], tr["reward"][t], tr["done"][t], w["status"], tr["obs"][t + 1], tr["mask"][t + 1], tr["player"][t + 1])
        else:
            env.policy_step(tr["action"][t], tr["reward"][t], tr["done"][t], w["status"], tr["obs"][t + 1], tr["mask"][t + 1], tr["player"][t + 1])

    def _window(self, p, gamma):
        T = self.T
        if self.persistent and self.ring > 1:
            wi = self.windows_played % self.ring                   # (run_window advances windows_played after all parts)
            self.traj[p] = self._window_views(self.rings[p], wi)
        tr = self.traj[p]
        if self.persistent:
            env, w, pol = self.envs[p], self.work[p], self.policy
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            if self.opponent == "net":
                wa = L.NetWeights(*[_p(x) for x in (self.w1t, self.b1, self.w2c, pol.critic_linear2.bias, self.w2a_t, pol.actor_linear2.bias)])
                wo = L.NetWeights(*[_p(x) for x in (self.ow1t, self.ob1, self.ow2c, self.ob2c, self.ow2a_t, self.ob2a)])
                out = L.RolloutBuffers(_p(tr["obs"]), _p(tr["mask"]), _p(tr["player"]), _p(tr["action"]), _p(tr["reward"]), _p(tr["done"]),
                                       _p(tr["value"]), _p(tr["log_prob"]), _p(tr["entropy"]), _p(w["status"]),
                                       _p(tr["returns"]) if self.ring == 1 else None, _p(tr["opp_action"]) if self.opp_slots else None,
                                       _p(tr["opp_logp"]) if self.opp_slots else None, _p(tr["opp_replies"]), self.opp_slots)
                L.check(L.lib.azul_batch_policy_rollout_vs(env._h, T, C.byref(wa), C.byref(wo), L.OBS_SIZE, self.H, L.NUM_ACTIONS, self.sample_seed,
                                                           self.opponent_seed, 0, _p(w["counter"]), C.byref(out), C.c_float(gamma), st))
            else:
              L.check(L.lib.azul_batch_policy_rollout_returns(
                env._h, T, 1 if self.opponent == "random" else 0, _p(self.w1t), _p(self.b1), _p(self.w2c), _p(pol.critic_linear2.bias),
                _p(self.w2a_t), _p(pol.actor_linear2.bias), L.OBS_SIZE, self.H, L.NUM_ACTIONS, self.sample_seed, 0, _p(w["counter"]),
                _p(tr["obs"]), _p(tr["mask"]), _p(tr["player"]), _p(tr["action"]), _p(tr["reward"]), _p(tr["done"]), _p(tr["value"]),
                _p(tr["log_prob"]), _p(tr["entropy"]), _p(w["status"]), _p(tr["returns"]) if self.ring == 1 else None, C.c_float(gamma), st))
            if self.ring == 1:
                return
            # returns of the new window and, chained backwards through the ring, of the older windows: the value flowing out of a
            # window's first step flows into the window before it (nn_runner.py:70-76 across window boundaries) -- one launch
            rg = self.rings[p]
            R = self.ring * T
            played = (self.windows_played + 1) * T
            L.check(L.lib.azul_discounted_returns_ring(_p(rg["reward"]), _p(rg["done"]), _p(rg["returns"]), C.c_float(gamma), R,
                                                       played % (1 << 40), min(R, played), self.h, st))
            return
        tr["obs"][0].copy_(tr["obs"][T])
        tr["mask"][0].copy_(tr["mask"][T])
        tr["player"][0].copy_(tr["player"][T])
        for t in range(T):
            self._move(p, t)
        L.check(L.lib.azul_discounted_returns(_p(tr["reward"]), _p(tr["done"]), _p(tr["returns"]), None, C.c_float(gamma), T, self.h,
                                              C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def _capture(self, gamma=0.99):
        self.gamma = gamma
        for p in range(self.parts):
            s = self.streams[p]
            with torch.cuda.stream(s):
                self._window(p, gamma)                  # warm-up (lazy inits, allocator) outside capture; advances the games
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                self._window(p, gamma)
            self.graphs.append(g)
        torch.cuda.synchronize(self.device)

    def run_window(self, gamma=0.99):
        """Advance every game by `window` moves; returns the per-part trajectory dicts (views into static buffers;
        `obs`, `mask`, `player` have T+1 slots: slot t is what the policy saw at move t)."""
        if self.use_graph and gamma != getattr(self, "gamma", gamma):
            raise ValueError("gamma is baked into the captured graph")
        cur = torch.cuda.current_stream(self.device)
        for p in range(self.parts):
            self.streams[p].wait_stream(cur)             # e.g. the optimiser step / refresh_weights enqueued by the caller
            with torch.cuda.stream(self.streams[p]):
                if self.use_graph:
                    self.graphs[p].replay()
                else:
                    self._window(p, gamma)
        self.windows_played += 1
        return self.traj

    def join(self):
        """Make the caller's current stream wait for the window(s) in flight -- on the device, the host does not block."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def counters(self):
        c = [e.counters() for e in self.envs]
        return {"episodes": sum(int(x["episodes"].sum()) for x in c), "stuck": sum(int(x["stuck"].sum()) for x in c)}
