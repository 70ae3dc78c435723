"""Policy-driven self-play on one MI355X (BASELINE configs[2]): the batched counterpart of the reference's
``NNRunner.run_episode`` (azulnet/nn_runner.py:17-47) and of the action sampling in ``Agent.get_ac_output``
(azulnet/agent.py:64-81).

The batch is split into independent parts, each with its own HIP stream: while the env kernel of one part runs
(`azul_batch_policy_step`: Azul.step + reward + done + auto-reset + next observation/mask in ONE launch), the
GEMMs/softmax/sampling of the other part run on theirs.  A window of `T` moves is captured once into a HIP graph
per part (no per-kernel host launch cost) and replayed; results are identical with `use_graph=False`.

Per (move t, game g) the record holds what the reference's run_episode keeps per agent step (C1 in SURVEY.md 8a):
observation, legal mask, action, reward, done, value, log-prob of the action and the entropy term
`-mean(log p over legal actions)` (nn_runner.py:36-40), plus the player who moved.
"""
import ctypes as C

import torch

from . import _lib as L
from .batch import BatchedAzul


class PolicyRollout:
    def __init__(self, policy, n_games=4096, parts=2, rules={"first_player": "Random", "tile_pool": "Lid"}, seed_base=0,
                 device=None, window=32, use_graph=True, record_obs=True):
        assert n_games % parts == 0
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.policy = policy.to(self.device).eval()
        self.n, self.parts, self.h, self.T = n_games, parts, n_games // parts, window
        self.record_obs = record_obs
        self.envs, self.streams, self.buf, self.traj, self.graphs = [], [], [], [], []
        for p in range(parts):
            env = BatchedAzul(self.h, rules=rules, device=self.device)
            env.seed(seed_base + p * self.h)                       # seeds follow the global game id
            env.runner_init()                                      # GameRunner()
            env.runner_init()                                      # reset() without pre-moves (flat self-play)
            self.envs.append(env)
            self.streams.append(torch.cuda.Stream(device=self.device))
            d, h, T = self.device, self.h, window
            b = {"obs": torch.zeros(h, L.OBS_SIZE, device=d), "mask": torch.zeros(h, L.NUM_ACTIONS, dtype=torch.uint8, device=d),
                 "player": torch.zeros(h, dtype=torch.uint8, device=d), "action": torch.zeros(h, dtype=torch.int32, device=d),
                 "status": torch.zeros(h, dtype=torch.uint8, device=d)}
            t = {"action": torch.zeros(T, h, dtype=torch.int32, device=d), "reward": torch.zeros(T, h, dtype=torch.int32, device=d),
                 "done": torch.zeros(T, h, dtype=torch.uint8, device=d), "player": torch.zeros(T, h, dtype=torch.uint8, device=d),
                 "value": torch.zeros(T, h, device=d), "log_prob": torch.zeros(T, h, device=d), "entropy": torch.zeros(T, h, device=d),
                 "mask": torch.zeros(T, h, L.NUM_ACTIONS, dtype=torch.uint8, device=d),
                 "returns": torch.zeros(T, h, device=d), "carry": torch.zeros(h, device=d)}
            if record_obs:
                t["obs"] = torch.zeros(T, h, L.OBS_SIZE, device=d)
            self.buf.append(b)
            self.traj.append(t)
            with torch.cuda.stream(self.streams[p]):
                env.observe_all(L.PERSP_CURRENT, b["obs"], b["mask"], b["player"])
        torch.cuda.synchronize(self.device)
        self.use_graph = use_graph
        self.graph_error = None
        if use_graph:
            try:
                self._capture()
            except Exception as e:          # capture is a launch-overhead optimisation only
                self.graph_error = repr(e)
                self.graphs = []
                self.use_graph = False
                torch.cuda.synchronize(self.device)

    # one move of one part, enqueued on the current stream
    def _move(self, p, t):
        env, b, tr = self.envs[p], self.buf[p], self.traj[p]
        obs, mask = b["obs"], b["mask"]
        with torch.no_grad():
            value = self.policy.forward_critic(obs)                              # agent.py:66
            probs, logp = self.policy.forward_actor(obs, mask)                   # agent.py:67
            legal = mask.bool()
            any_legal = legal.any(dim=1)
            safe = torch.where(any_legal.unsqueeze(1), probs, torch.full_like(probs, 1.0 / probs.shape[1]))
            action = torch.multinomial(safe, 1).squeeze(1)                       # agent.py:69 (np.random.choice(p=probs))
            lp = logp.gather(1, action.unsqueeze(1)).squeeze(1)                  # nn_runner.py:32
            ent = -(torch.where(legal, logp, torch.zeros_like(logp)).sum(dim=1) / legal.sum(dim=1).clamp(min=1))   # :36-40
            b["action"].copy_(torch.where(any_legal, action, torch.full_like(action, -1)).to(torch.int32))
        if self.record_obs:
            tr["obs"][t].copy_(obs)
        tr["mask"][t].copy_(mask)
        tr["player"][t].copy_(b["player"])
        tr["action"][t].copy_(b["action"])
        tr["value"][t].copy_(value.squeeze(1))
        tr["log_prob"][t].copy_(lp)
        tr["entropy"][t].copy_(ent)
        env.policy_step(b["action"], tr["reward"][t], tr["done"][t], b["status"], b["obs"], b["mask"], b["player"])

    def _window(self, p, gamma):
        for t in range(self.T):
            self._move(p, t)
        tr = self.traj[p]
        L.check(L.lib.azul_discounted_returns(C.c_void_p(tr["reward"].data_ptr()), C.c_void_p(tr["done"].data_ptr()),
                                              C.c_void_p(tr["returns"].data_ptr()), None, C.c_float(gamma), self.T, self.h,
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
        """Advance every game by `window` moves; returns the per-part trajectory dicts (views into static buffers)."""
        if self.use_graph and gamma != getattr(self, "gamma", gamma):
            raise ValueError("gamma is baked into the captured graph")
        for p in range(self.parts):
            with torch.cuda.stream(self.streams[p]):
                if self.use_graph:
                    self.graphs[p].replay()
                else:
                    self._window(p, gamma)
        return self.traj

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def counters(self):
        c = [e.counters() for e in self.envs]
        return {"episodes": sum(int(x["episodes"].sum()) for x in c), "stuck": sum(int(x["stuck"].sum()) for x in c)}
