"""Batched ``NNRunner.train`` (azulnet/nn_runner.py:49-84) with the on-disk surface either side of the path (row N3):
the CSV training log the reference intends to write, checkpoints, and resumable runs.

One "batch" of the reference is `batch_size` sequential episodes followed by one `Agent.update`; here it is one rollout
window of `window` agent steps for all `n_games` concurrent games (policy vs the RandomAgent opponent inside the env
step) followed by one `A2CLearner.update`: with the default ring of three windows every step of every episode is trained exactly
once, when its episode ends (A2CLearner.update_from_rollout; the reference trains on whole episodes, nn_runner.py:59-76).

CSV (nn_runner.py:51-54, 79-82): header ``batch`` + the five agent statistics keys (agent.py:13) + the ten game statistics
keys (game_runner.py:12); one row per logged batch with the batch's means.  The reference's own writer dereferences
``self.game_statistics``, which NNRunner does not have (it lives on the GameRunner), so that branch raises as soon as a
net name is given; the columns and their order here are the ones that code spells out.
  reward       mean over the batch of the per-episode reward sum (nn_runner.py:63) -- here: the window's total shaped
               reward / episodes finished in the window
  *_loss       the update's loss terms (agent.py:51-57)
  game keys    means over the episodes finished in the window of Azul.get_statistics (azul.py:314-315), accumulated on
               the device by the env step (win_percent etc. are per-episode 0/1 or counts, like the reference's buffers)

Checkpoints (`save_checkpoint` / `load_checkpoint`): the policy's state_dict (the reference's parameter names: loads into
`azulnet.model.ActorCritic` as is), the Adam state, every game's 128-byte record and CPython RNG state, the Philox step
counters, the batch index and (by default) the trajectory ring with the learner's selection books -- a restored run replays the
next window bit for bit and performs the same updates as the uninterrupted run.
"""
import csv
import os

import numpy as np
import torch

from .learner import A2CLearner, complete_episode_samples
from .records import STAT_KEYS
from .rollout import PolicyRollout

AGENT_STAT_KEYS = ("reward", "actor_loss", "critic_loss", "entropy_loss", "ac_loss")          # agent.py:13


class BatchedTrainer:
    def __init__(self, policy, n_games=4096, window=32, parts=1, learning_rate=3e-4, gamma=0.99, seed_base=0, sample_seed=0x5EED,
                 rules={"first_player": "Random", "tile_pool": "Lid"}, device=None, use_graph=True, persistent=True, results_dir="results",
                 ring=3, opponent="random", opponent_refresh=0, move_limit=0):
        """ring: trajectory windows kept (persistent rollout with one part): with ring >= 2 every step of every episode is trained
        exactly once (episodes straddle windows; an episode may span ring - 1 window boundaries), like NNRunner.train.
        opponent: "random" (GameRunner's default RandomAgent, the reference's scripts/training.py), a module (GameRunner(opponent=Agent(...)),
        game_runner.py:27-30: a frozen second net inside the rollout kernel), or "self": a frozen COPY of the policy that is replaced by the
        current policy every `opponent_refresh` updates (0: never) -- training against a past self.
        move_limit > 0: BatchedAzul.set_move_limit (beyond the reference: a game that would never end is cut, done = 3; its steps are trained like an
        episode that ended there, the return chain starts at the cut)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        policy = policy.to(dev)
        self.learner = A2CLearner(policy, learning_rate=learning_rate, gamma=gamma)
        # rollout kernels and learner share ONE k-major copy of the weights (the learner's flat master copy)
        self.opponent_refresh = int(opponent_refresh)
        self._self_opponent = opponent == "self"
        if self._self_opponent:
            import copy
            opponent = copy.deepcopy(policy)
        self.rollout = PolicyRollout(policy, n_games=n_games, parts=parts, rules=rules, seed_base=seed_base, device=dev, window=window,
                                     use_graph=use_graph, sample_seed=sample_seed, opponent=opponent, persistent=persistent,
                                     kweights=self.learner.kweights(dev), ring=ring if (persistent and parts == 1) else 1, move_limit=move_limit)
        self.gamma = gamma
        self.results_dir = results_dir
        self.batch = 0
        self._stat_base = self._stat_totals()
        self.history = {k: [] for k in ("batch",) + AGENT_STAT_KEYS + STAT_KEYS}

    # ---- statistics ----------------------------------------------------------------------------------
    def _stat_totals(self):
        """Finished episodes and the ten get_statistics() sums over all games: reduced ON THE DEVICE (the per-game counters stay
        there; 11 numbers cross PCIe instead of 88 bytes per game), on the stream that ran the games."""
        tot = None
        for p, env in enumerate(self.rollout.envs):
            with torch.cuda.stream(self.rollout.streams[p]):
                c = env.counters_dev()
                t = torch.cat([c["episodes"].sum().to(torch.float64).reshape(1), c["stat_sums"].sum(dim=0)])
            self.rollout.streams[p].synchronize()
            tot = t if tot is None else tot + t
        tot = tot.cpu().numpy()
        return int(tot[0]), tot[1:]

    def _game_statistics(self):
        """Means over the episodes finished since the last call (GameStatistics.get_stats, game_runner.py:17-22)."""
        ep, sums = self._stat_totals()
        ep0, sums0 = self._stat_base
        self._stat_base = (ep, sums)
        n = ep - ep0
        return n, {k: (float(sums[i] - sums0[i]) / n if n else float("nan")) for i, k in enumerate(STAT_KEYS)}

    # ---- one batch -----------------------------------------------------------------------------------
    def run_batch(self, collect_stats=True):
        """One rollout window + one update.  collect_stats=False keeps everything on the device (no host round trip): the
        episode statistics and the reward keep accumulating and are reported by the next collecting call, like the reference's
        GameStatistics buffers between two get_stats() calls (game_runner.py:17-22)."""
        tr = self.rollout.run_window(self.gamma)
        self.rollout.join()                              # device-side dependency: the host keeps enqueueing
        out = self.learner.update_from_rollout(self.rollout)
        self.rollout.refresh_weights()
        self.batch += 1
        if self._self_opponent and self.opponent_refresh and self.batch % self.opponent_refresh == 0:
            self.rollout.set_opponent(self.rollout.policy)         # the frozen past self moves up to the present (weights copied in place)
        r = sum(part["reward"].sum() for part in tr)
        self._reward_acc = r if getattr(self, "_reward_acc", None) is None else self._reward_acc + r
        if not collect_stats:
            return None
        episodes, game = self._game_statistics()
        row = {"batch": self.batch, "reward": float(self._reward_acc) / episodes if episodes else float("nan")}
        self._reward_acc = None
        # the loss terms of every update since the last collecting call, averaged (AgentStatistics.get_stats, agent.py:19-24)
        fresh = min(self.learner.updates - getattr(self, "_loss_mark", 0), len(self.learner.statistics["ac_loss"]))
        for k in AGENT_STAT_KEYS[1:]:
            vals = list(self.learner.statistics[k])[len(self.learner.statistics[k]) - fresh:]
            row[k] = float(torch.stack([torch.as_tensor(v, dtype=torch.float32, device=self.rollout.device) for v in vals]).mean())
        self._loss_mark = self.learner.updates
        row.update(game)
        for k, v in row.items():
            self.history[k].append(v)
        return row

    def train(self, net_name=None, batches=1000, log_every=None, checkpoint_every=1000, checkpoint_ring=False):
        """nn_runner.py:49-84: `batches` updates; with a `net_name` the CSV log and the checkpoints go to results_dir.
        checkpoint_ring: whether the periodic checkpoints carry the trajectory ring (~75 KB per game, save_checkpoint) -- off by
        default: a periodic file is a restart point, not a bit-exact continuation; call save_checkpoint(path) for the latter."""
        log_every = max(1, batches // 1000) if log_every is None else log_every             # nn_runner.py:79
        path = None
        if net_name is not None:
            os.makedirs(self.results_dir, exist_ok=True)
            path = os.path.join(self.results_dir, net_name + ".csv")
            if self.batch == 0 or not os.path.exists(path):
                with open(path, mode="w", newline="") as fh:                                # nn_runner.py:51-54
                    csv.writer(fh, delimiter=",", quotechar='"', quoting=csv.QUOTE_MINIMAL).writerow(
                        ["batch"] + list(AGENT_STAT_KEYS) + list(STAT_KEYS))
        last = None
        for i in range(batches):
            logging = (self.batch + 1) % log_every == 0 or i == batches - 1
            row = self.run_batch(collect_stats=logging)
            last = row if row is not None else last
            if path is not None and self.batch % log_every == 0:
                with open(path, mode="a+", newline="") as fh:                               # nn_runner.py:80-82
                    csv.writer(fh, delimiter=",", quotechar='"', quoting=csv.QUOTE_MINIMAL).writerow(
                        [last["batch"]] + [last[k] for k in AGENT_STAT_KEYS] + [last[k] for k in STAT_KEYS])
            if net_name is not None and self.batch % checkpoint_every == 0:                # nn_runner.py:83-84
                self.save_checkpoint(os.path.join(self.results_dir, net_name + ".pt"), save_ring=checkpoint_ring)
                self.export_mx(os.path.join(self.results_dir, net_name + ".mx"))            # the reference's file name and content
        return last

    # ---- the reference's own network file ---------------------------------------------------------------
    def export_mx(self, path, module_factory=None):
        """The file the reference writes and reads: NNRunner.train does torch.save(agent.ac_net, "/results/<name>.mx") (nn_runner.py:83-84)
        and Agent(base_net_file=...) torch.load()s that MODULE (agent.py:36).  `module_factory` builds the module to pickle -- pass the
        reference's azulnet.model.ActorCritic to get a file the unchanged reference loads; default: this package's BatchedActorCritic
        (same parameter names, so its state_dict loads into either class).  The weights are copied to the CPU first."""
        from .policy import BatchedActorCritic
        pol = self.rollout.policy
        self.rollout.synchronize()
        mod = (module_factory or (lambda: BatchedActorCritic(pol.critic_linear1.in_features, pol.actor_linear2.out_features,
                                                             pol.critic_linear1.out_features)))()
        mod.load_state_dict({k: v.detach().cpu() for k, v in pol.state_dict().items()})
        torch.save(mod, path)
        return path

    def import_mx(self, path):
        """Continue from a reference network file: a pickled module (anything with state_dict(): the reference's ActorCritic, loadable
        where its class is importable) or a plain state_dict, with the reference's parameter names."""
        obj = torch.load(path, map_location="cpu", weights_only=False)
        sd = obj.state_dict() if hasattr(obj, "state_dict") else obj
        self.rollout.synchronize()
        self.rollout.policy.load_state_dict({k: v.to(self.rollout.device) for k, v in sd.items()})
        self.learner.sync_from_module()                  # the flat k-major master copy the kernels read
        self.rollout.refresh_weights()

    # ---- checkpoints ---------------------------------------------------------------------------------
    def save_checkpoint(self, path, save_ring=True):
        """save_ring=True (default): the trajectory ring (the last `ring` windows: episodes in flight), the rollout's window clock and
        the learner's per-game "first step not trained yet" go into the file, so that a restored run EQUALS the uninterrupted one
        (~75 KB per game with the default ring of 3 x 32 steps).  save_ring=False: a small file; a run restored from it starts its
        books with the first window it plays -- the steps of the episodes in flight at the checkpoint that were recorded before it are
        not trained, and their tails are trained as if they were whole episodes (returns of those tails are still exact)."""
        ro = self.rollout
        ro.synchronize()
        torch.cuda.synchronize(ro.device)
        envs = []
        for p, env in enumerate(ro.envs):
            mt, pos = env.get_rng_range()
            envs.append({"records": env.get_records(), "mt": mt, "pos": pos,
                         "counter": ro.work[p]["counter"].cpu().numpy().copy(),
                         "next_obs": ro.traj[p]["obs"][ro.T].cpu(), "next_mask": ro.traj[p]["mask"][ro.T].cpu(),
                         "next_player": ro.traj[p]["player"][ro.T].cpu()})
        ck = {"policy": ro.policy.state_dict(), "optimizer": self.learner.optimizer_state(), "envs": envs,
              "batch": self.batch, "n_games": ro.n, "parts": ro.parts, "window": ro.T, "game_id_base": ro.game_id_base,
              "windows_played": ro.windows_played, "ring": ro.ring, "ring_saved": bool(save_ring and ro.ring > 1)}
        if ro.opponent == "net":                         # the network opponent's (frozen) weights: a resumed run plays against the same net
            ck["opponent"] = {k: getattr(ro, k).cpu() for k in ("ow1t", "ob1", "ow2c", "ob2c", "ow2a_t", "ob2a")}
        if save_ring and ro.ring > 1:
            ck["ring_buffers"] = [{k: v.cpu() for k, v in rg.items()} for rg in ro.rings]
            ck["learner_ring"] = self.learner.ring_state()
        torch.save(ck, path)

    def load_checkpoint(self, path):
        ro = self.rollout
        # (on the CPU first: the ring of a large batch is GBs and every tensor is copied into its resident buffer below, one at a time)
        ck = torch.load(path, map_location="cpu", weights_only=False)
        if (ck["n_games"], ck["parts"], ck["window"]) != (ro.n, ro.parts, ro.T):
            raise ValueError("checkpoint was written for n_games=%d parts=%d window=%d" % (ck["n_games"], ck["parts"], ck["window"]))
        ro.synchronize()
        ro.policy.load_state_dict(ck["policy"])
        self.learner.load_optimizer_state(ck["optimizer"])          # also rebuilds the flat master copy from the module
        ro.refresh_weights()
        new_base = int(ck.get("game_id_base", ro.game_id_base))            # the sampling streams are keyed by the global game id
        if new_base != ro.game_id_base:
            ro.game_id_base = new_base
            if ro.use_graph:                                                # the id base is a launch argument baked into the captured graphs
                ro.graphs = []
                ro._capture(getattr(ro, "gamma", self.gamma))               # (its warm-up window advances the games: they are restored below)
        for p, (env, e) in enumerate(zip(ro.envs, ck["envs"])):
            env.set_id_base(ro.game_id_base + p * ro.h)
            env.set_records(e["records"])
            env.set_rng_range(e["mt"], e["pos"])
            ro.work[p]["counter"].copy_(torch.from_numpy(e["counter"]))
            ro.traj[p]["obs"][ro.T].copy_(e["next_obs"])
            ro.traj[p]["mask"][ro.T].copy_(e["next_mask"])
            ro.traj[p]["player"][ro.T].copy_(e["next_player"])
        # the trajectory ring and the books of the ring selection: restored when the file has them (the run then equals the
        # uninterrupted one), otherwise the learner's books restart with the next window (nothing recorded before this call -- by the
        # checkpointed run or by this trainer -- is ever selected: `pending` starts at the next window, whose selection only reads
        # done flags of windows played from now on)
        if "ring_buffers" in ck and ck.get("ring") == ro.ring and ro.ring > 1:
            for rg, saved in zip(ro.rings, ck["ring_buffers"]):
                for k, v in saved.items():
                    rg[k].copy_(v)
            ro.windows_played = int(ck["windows_played"])
            for p in range(ro.parts):
                ro.traj[p] = ro._window_views(ro.rings[p], (ro.windows_played - 1) % ro.ring)
            self.learner.load_ring_state(ro, ck.get("learner_ring"))
        else:
            import warnings
            if "ring_buffers" in ck and ro.ring > 1:
                warnings.warn("checkpoint %s holds a trajectory ring of %s windows, this rollout uses %d: the ring is NOT restored -- the "
                              "resumed run restarts its books with the next window and no longer equals the uninterrupted one"
                              % (path, ck.get("ring"), ro.ring))
            elif ro.ring > 1:
                # a small file (save_ring=False: what train() writes periodically unless checkpoint_ring=True)
                warnings.warn("checkpoint %s carries no trajectory ring (written with save_ring=False): the steps of episodes in flight when it "
                              "was written are not trained, their tails count as whole episodes -- the resumed run does not equal the "
                              "uninterrupted one (save_checkpoint(save_ring=True) / train(checkpoint_ring=True) for an exact resume)" % path)
            self.learner.load_ring_state(ro, None)
        if ro.opponent == "net" and "opponent" in ck:
            with torch.no_grad():
                for k, v in ck["opponent"].items():
                    getattr(ro, k).copy_(v.to(ro.device))
        torch.cuda.synchronize(ro.device)
        self.batch = int(ck["batch"])
        self._stat_base = self._stat_totals()
        self._reward_acc = None
        self._loss_mark = self.learner.updates
