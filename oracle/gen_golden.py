#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (container-only; never runs on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Imports /root/reference/azulnet (with the two removed numpy aliases restored), plays seeded games
through the reference's own GameRunner / Azul / RandomAgent code and stores inputs + expected
outputs as arrays.  Only DATA is written: no reference source text leaves the container.
"""
import collections
import copy
import os
import random
import shutil
import sys

import numpy as np

np.int = int      # the reference uses aliases removed in numpy >= 1.24 (azul.py:19-26)
np.bool = bool
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
import azulnet  # noqa: E402
from azulnet import Azul, GameRunner, RandomAgent, check_all_valid, nn_serialize  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
RES = "/root/reference/tests/resources"

RULESETS = {
    # name: (rules dict handed to the reference, first_player code, tile_pool code)
    "lid_randomfirst": ({"first_player": "Random", "tile_pool": "Lid"}, 0, 1),   # GameRunner default
    "random_first1": ({}, -1, 0),                                               # Azul default
    "lid_first2": ({"first_player": 2, "tile_pool": "Lid"}, 2, 1),
}


def words_pos():
    return random.getstate()[1][624]


class Recorder:
    """Wraps Azul.step at class level and records every env move of the reference."""

    FIELDS = ["mask", "action", "phi", "eog", "displays", "center", "pattern_lines", "walls", "floors",
              "score", "cur", "nfp", "eog_flag", "turn_counter", "box", "lid", "first_player_stats",
              "floor_penalty", "max_combo", "completed_lines", "obs", "rng_words", "player_before"]

    def __init__(self):
        self.rows = {k: [] for k in self.FIELDS}
        self.words = 0
        self.last_pos = None

    def sync_words(self):
        pos = words_pos()
        if self.last_pos is not None:
            d = pos - self.last_pos
            if d < 0 or (d == 0 and False):
                d += 624
            self.words += d
        self.last_pos = pos

    def install(self, runner_ref):
        rec = self
        orig = Azul.step

        def step(game, display, color, pattern):
            mask = check_all_valid(game)
            player_before = game.current_player
            orig(game, display, color, pattern)
            g2 = copy.deepcopy(game)
            g2.count_score()
            rec.sync_words()
            r = rec.rows
            r["mask"].append(np.packbits(mask, bitorder="little"))
            r["action"].append(nn_serialize(display, color, pattern))
            r["phi"].append(int(g2.score[0] - g2.score[1]))
            r["eog"].append(bool(game.is_end_of_game()))
            r["displays"].append(np.array(game.game_board_displays, dtype=np.uint8))
            r["center"].append(np.array(game.game_board_center, dtype=np.uint8))
            r["pattern_lines"].append(np.array(game.pattern_lines, dtype=np.uint8))
            r["walls"].append(np.array(game.walls, dtype=np.uint8))
            r["floors"].append(np.array(game.floors, dtype=np.uint8))
            r["score"].append(np.array(game.score, dtype=np.int16))
            r["cur"].append(game.current_player)
            r["nfp"].append(game.next_first_player)
            r["eog_flag"].append(bool(game.end_of_game))
            r["turn_counter"].append(game.turn_counter)
            if game.tile_pool == "Lid":
                r["box"].append(np.array(game.box_tiles, dtype=np.uint8))
                r["lid"].append(np.array(game.lid_tiles, dtype=np.uint8))
            else:
                r["box"].append(np.zeros(5, np.uint8))
                r["lid"].append(np.zeros(5, np.uint8))
            r["first_player_stats"].append(np.array(game.first_player_stats, dtype=np.uint16))
            r["floor_penalty"].append(np.array(game.floor_penalty, dtype=np.int16))
            r["max_combo"].append(np.array(game.max_combo, dtype=np.uint8))
            r["completed_lines"].append(np.array(game.completed_lines, dtype=np.uint8))
            runner = runner_ref[0]
            r["obs"].append(np.stack([runner.get_state(0), runner.get_state(1)]).astype(np.int16))
            r["rng_words"].append(rec.words)
            r["player_before"].append(player_before)

        Azul.step = step
        return orig


def play_stream(seed, rules, episodes):
    """random.seed(seed); GameRunner(rules); `episodes` x (reset(); RandomAgent vs RandomAgent)."""
    rec = Recorder()
    runner_ref = [None]
    orig = rec.install(runner_ref)
    agent_rows = {"env_index": [], "reward": [], "done": [], "obs_before": [], "mask_before": [],
                  "action": [], "move_counter": [], "player_score": []}
    episode_rows = {"first_env_index": [], "stats": [], "words_at_start": []}
    try:
        random.seed(seed)
        rec.last_pos = words_pos()
        runner = GameRunner(rules=dict(rules))
        runner_ref[0] = runner
        agent = RandomAgent()
        for _ in range(episodes):
            runner.reset()
            rec.sync_words()
            episode_rows["first_env_index"].append(len(rec.rows["action"]) - runner.move_counter)
            episode_rows["words_at_start"].append(rec.words)
            done = False
            while not done:
                mask = runner.get_valid_moves()
                obs = runner.get_state()
                a = agent.get_a_output(None, torch.from_numpy(mask.reshape(1, 180)))
                reward, done = runner.step(a)
                agent_rows["env_index"].append(len(rec.rows["action"]))
                agent_rows["reward"].append(int(reward))
                agent_rows["done"].append(bool(done))
                agent_rows["obs_before"].append(obs.astype(np.int16))
                agent_rows["mask_before"].append(np.packbits(mask, bitorder="little"))
                agent_rows["action"].append(int(a))
                agent_rows["move_counter"].append(runner.move_counter)
                agent_rows["player_score"].append(int(runner.player_score))
            st = runner.game.get_statistics()
            episode_rows["stats"].append(np.array([float(st[k]) for k in
                                                   ["player_score", "opponent_score", "rounds", "percent_first_player",
                                                    "floor_penalty", "max_combo", "completed_rows", "completed_columns",
                                                    "completed_colors", "win_percent"]]))
    finally:
        Azul.step = orig
    out = {("env_" + k): np.array(v) for k, v in rec.rows.items()}
    out.update({("agent_" + k): np.array(v) for k, v in agent_rows.items()})
    out.update({("episode_" + k): np.array(v) for k, v in episode_rows.items()})
    return out


def gen_trajectories():
    for name, (rules, _fp, _pool) in RULESETS.items():
        seeds = list(range(32)) if name != "lid_first2" else list(range(8))
        seeds += [2 ** 32 + 7, 2 ** 63 + 11] if name == "lid_randomfirst" else []
        blob = {"seeds": np.array(seeds, dtype=np.uint64)}
        for i, s in enumerate(seeds):
            d = play_stream(s, rules, episodes=2)
            for k, v in d.items():
                blob["s%d_%s" % (i, k)] = v
        np.savez_compressed(os.path.join(OUT, "traj_%s.npz" % name), **blob)
        print("wrote traj_%s.npz (%d seeds)" % (name, len(seeds)))


def gen_rng():
    blob = {}
    seeds = [0, 1, 12345, 2 ** 32 - 1, 2 ** 32, 2 ** 32 + 7, 2 ** 63 + 11]
    blob["seeds"] = np.array(seeds, dtype=np.uint64)
    for i, s in enumerate(seeds):
        random.seed(s)
        st = random.getstate()
        blob["state_%d" % i] = np.array(st[1], dtype=np.uint64).astype(np.uint32)
        blob["u32_%d" % i] = np.array([random.getrandbits(32) for _ in range(1400)], dtype=np.uint32)
        random.seed(s)
        blob["random_%d" % i] = np.array([random.random() for _ in range(700)], dtype=np.float64)
        random.seed(s)
        blob["randbelow5_%d" % i] = np.array([random.randrange(0, 5, 1) for _ in range(300)], dtype=np.uint8)
        random.seed(s)
        blob["choice12_%d" % i] = np.array([random.choice([1, 2]) for _ in range(300)], dtype=np.uint8)
    # random.choices on tile-pool style weights and on RandomAgent style weights
    random.seed(99)
    rs = np.random.RandomState(7)
    boxes, picks = [], []
    for _ in range(400):
        box = rs.randint(0, 21, size=5)
        if box.sum() == 0:
            box[2] = 3
        total = np.sum(box)
        picks.append(random.choices([0, 1, 2, 3, 4], weights=[b / total for b in box])[0])
        boxes.append(box)
    blob["choices_box"] = np.array(boxes, dtype=np.uint8)
    blob["choices_box_pick"] = np.array(picks, dtype=np.uint8)
    random.seed(2024)
    agent = RandomAgent()
    masks, picks = [], []
    for i in range(600):
        dens = [0.02, 0.1, 0.3, 0.7, 1.0][i % 5]
        m = rs.rand(180) < dens
        if i % 7 == 0:
            m[:30] = False
        if i % 11 == 0:
            m[30:] = False
        if not m.any():
            m[rs.randint(0, 180)] = True
        picks.append(agent.get_a_output(None, torch.from_numpy(m.reshape(1, 180))))
        masks.append(np.packbits(m, bitorder="little"))
    blob["choices_mask"] = np.array(masks)
    blob["choices_mask_pick"] = np.array(picks, dtype=np.uint8)
    blob["choices_mask_words_end"] = np.array(random.getstate()[1][624])
    np.savez_compressed(os.path.join(OUT, "pyrandom.npz"), **blob)
    print("wrote pyrandom.npz")


def gen_boards():
    """Copy the reference's JSON board fixtures (data files) and record derived known answers."""
    dst = os.path.join(OUT, "resources")
    os.makedirs(dst, exist_ok=True)
    blob = {}
    names = sorted(f for f in os.listdir(RES) if f.endswith(".json"))
    for f in names:
        shutil.copyfile(os.path.join(RES, f), os.path.join(dst, f))
        g = Azul()
        g.import_JSON(os.path.join(RES, f))
        key = f[:-5]
        blob[key + "_mask"] = np.packbits(check_all_valid(g), bitorder="little")
        runner = GameRunner(rules={})
        runner.game = g
        blob[key + "_obs"] = np.stack([runner.get_state(0), runner.get_state(1)]).astype(np.int16)
        g2 = copy.deepcopy(g)
        g2.count_score()
        blob[key + "_scored_score"] = np.array(g2.score, dtype=np.int16)
        blob[key + "_scored_walls"] = np.array(g2.walls, dtype=np.uint8)
        blob[key + "_scored_pattern_lines"] = np.array(g2.pattern_lines, dtype=np.uint8)
        blob[key + "_scored_stats"] = np.concatenate([g2.floor_penalty, g2.max_combo, g2.completed_lines.flatten()])
        blob[key + "_eor"] = np.array(g.is_end_of_round())
        blob[key + "_eog"] = np.array(g.is_end_of_game())
    np.savez_compressed(os.path.join(OUT, "boards.npz"), **blob)
    print("wrote boards.npz + %d resource json files" % len(names))


def gen_policy_contract():
    """(state_dict, obs, mask) -> (probs, log_probs, value) of the reference's ActorCritic (model.py:12-41)."""
    from azulnet import ActorCritic
    torch.manual_seed(0)
    net = ActorCritic(136, 180)
    rs = np.random.RandomState(3)
    obs = rs.randint(0, 5, size=(64, 136)).astype(np.float32)
    mask = rs.rand(64, 180) < 0.15
    mask[np.arange(64), rs.randint(0, 180, 64)] = True
    with torch.no_grad():
        value = net.forward_critic(torch.from_numpy(obs)).numpy()
        probs, logp = net.forward_actor(torch.from_numpy(obs), torch.from_numpy(mask))
    blob = {"obs": obs, "mask": mask, "value": value, "probs": probs.numpy(), "logp": logp.numpy()}
    for k, v in net.state_dict().items():
        blob["sd_" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "policy_contract.npz"), **blob)
    print("wrote policy_contract.npz")


def gen_a2c_update():
    """Agent.update (agent.py:39-62) on fixed inputs: the four loss terms and the parameters after ONE Adam step."""
    from azulnet import Agent
    torch.manual_seed(0)
    agent = Agent()
    rs = np.random.RandomState(9)
    n = 48
    obs = torch.from_numpy(rs.randint(0, 5, size=(n, 136)).astype(np.float32))
    mask = rs.rand(n, 180) < 0.2
    mask[np.arange(n), rs.randint(0, 180, n)] = True
    mask_t = torch.from_numpy(mask)
    actions = np.array([rs.choice(np.flatnonzero(mask[i])) for i in range(n)])
    qvals = rs.randn(n, 1).astype(np.float64) * 5
    sd_before = {k: v.clone().numpy() for k, v in agent.ac_net.state_dict().items()}
    values, log_probs, entropy = [], [], []
    for i in range(n):                                   # the shapes NNRunner.run_episode produces (nn_runner.py:26-44)
        value = agent.ac_net.forward_critic(obs[i:i + 1])
        _, logp = agent.ac_net.forward_actor(obs[i:i + 1], mask_t[i:i + 1])
        values.append(value)
        log_probs.append(logp.squeeze(0)[actions[i]])
        entropy.append(-logp.masked_select(mask_t[i:i + 1]).mean())
    agent.update(qvals, [0.0], values, log_probs, entropy)
    st = agent.agent_statistics.statisticsBuffer
    blob = {"obs": obs.numpy(), "mask": mask, "actions": actions, "qvals": qvals,
            "actor_loss": st["actor_loss"], "critic_loss": st["critic_loss"], "entropy_loss": st["entropy_loss"], "ac_loss": st["ac_loss"]}
    for k, v in sd_before.items():
        blob["before_" + k] = v
    for k, v in agent.ac_net.state_dict().items():
        blob["after_" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "a2c_update.npz"), **blob)
    print("wrote a2c_update.npz")


PLAYER_RULESETS = {
    # name: (rules handed to the reference with `players` substituted for "P", first_player code, tile_pool code)
    "default": ({}, -1, 0),                                                     # Azul(players=P): random pool, player 1 starts
    "lid_randomfirst": ({"first_player": "Random", "tile_pool": "Lid"}, 0, 1),
    "lid_firstP": ({"first_player": "P", "tile_pool": "Lid"}, "P", 1),          # the LAST player starts
}
PLAYER_FIELDS = ["mask", "action", "displays", "center", "pattern_lines", "walls", "floors", "score", "cur", "nfp", "eog_flag",
                 "turn_counter", "box", "lid", "first_player_stats", "floor_penalty", "max_combo", "completed_lines", "rng_words",
                 "eor_before_step", "eog_walls"]


def play_players_stream(seed, players, rules, max_moves=400):
    """Azul(players=3|4) driven through Azul.step (azul.py:296-313) with uniformly random LEGAL actions; the action picker
    is a private random.Random so that the process-global stream is consumed by the game alone (azul.py:37,78,87)."""
    from azulnet.azul import GameEnded, IllegalMove
    from azulnet import nn_deserialize
    rows = {k: [] for k in PLAYER_FIELDS}
    random.seed(seed)
    pos0 = words_pos()
    words = [0, pos0]

    def sync():
        pos = words_pos()
        d = pos - words[1]
        if d < 0:
            d += 624
        words[0] += d
        words[1] = pos
        return words[0]

    picker = random.Random(seed * 7919 + players)
    g = Azul(players=players, rules=dict(rules))
    words_ctor = sync()
    init = {"nfp": g.next_first_player, "cur": g.current_player}
    g.new_round()
    words_round = sync()
    first = {"displays": np.array(g.game_board_displays, dtype=np.uint8), "center": np.array(g.game_board_center, dtype=np.uint8),
             "cur": g.current_player, "fps": np.array(g.first_player_stats, dtype=np.uint16)}
    end = "max_moves"
    illegal_checked = 0
    for t in range(max_moves):
        mask = check_all_valid(g)
        legal = np.flatnonzero(mask)
        if len(legal) == 0:
            end = "stuck"                            # hazard H3: nobody can move, the round is not over
            break
        if t % 9 == 4 and len(legal) < 180:           # an illegal move raises and leaves the game untouched (azul.py:301-302)
            bad = int(np.flatnonzero(~mask)[picker.randrange(180 - len(legal))])
            before = copy.deepcopy(g)
            try:
                g.step(*nn_deserialize(bad))
                raise AssertionError("illegal move accepted")
            except IllegalMove:
                assert g == before
                illegal_checked += 1
        a = int(legal[picker.randrange(len(legal))])
        eor_before = bool(g.is_end_of_round())
        try:
            g.step(*nn_deserialize(a))
        except ValueError:                            # "Lid" pool: box and lid both empty inside new_round (azul.py:85-87)
            end = "box_empty"
            rows["mask"].append(np.packbits(mask, bitorder="little"))
            rows["action"].append(a)
            break
        r = rows
        r["mask"].append(np.packbits(mask, bitorder="little"))
        r["action"].append(a)
        r["displays"].append(np.array(g.game_board_displays, dtype=np.uint8))
        r["center"].append(np.array(g.game_board_center, dtype=np.uint8))
        r["pattern_lines"].append(np.array(g.pattern_lines, dtype=np.uint8))
        r["walls"].append(np.array(g.walls, dtype=np.uint8))
        r["floors"].append(np.array(g.floors, dtype=np.uint8))
        r["score"].append(np.array(g.score, dtype=np.int16))
        r["cur"].append(g.current_player)
        r["nfp"].append(g.next_first_player)
        r["eog_flag"].append(bool(g.end_of_game))
        r["turn_counter"].append(g.turn_counter)
        r["box"].append(np.array(g.box_tiles, dtype=np.uint8) if g.tile_pool == "Lid" else np.zeros(5, np.uint8))
        r["lid"].append(np.array(g.lid_tiles, dtype=np.uint8) if g.tile_pool == "Lid" else np.zeros(5, np.uint8))
        r["first_player_stats"].append(np.array(g.first_player_stats, dtype=np.uint16))
        r["floor_penalty"].append(np.array(g.floor_penalty, dtype=np.int16))
        r["max_combo"].append(np.array(g.max_combo, dtype=np.uint8))
        r["completed_lines"].append(np.array(g.completed_lines, dtype=np.uint8))
        r["rng_words"].append(sync())
        r["eor_before_step"].append(eor_before)
        r["eog_walls"].append(bool(g.is_end_of_game()))
        if g.end_of_game:
            end = "game_end"
            try:
                g.step(*nn_deserialize(a))
                raise AssertionError("step on a finished game accepted")
            except GameEnded:
                pass
            break
    st = g.get_statistics()
    stats = np.array([float(st[k]) for k in ["player_score", "opponent_score", "rounds", "percent_first_player", "floor_penalty",
                                             "max_combo", "completed_rows", "completed_columns", "completed_colors", "win_percent"]])
    out = {k: np.array(v) for k, v in rows.items()}
    out.update({"end": np.array(end), "stats": stats, "words_ctor": np.array(words_ctor), "words_round": np.array(words_round),
                "init_nfp": np.array(init["nfp"]), "first_displays": first["displays"], "first_center": first["center"],
                "first_cur": np.array(first["cur"]), "first_fps": first["fps"], "illegal_checked": np.array(illegal_checked)})
    return out


def gen_players():
    """Row N4, first slice: what the reference itself does for 3 and 4 players (5 displays, azul.py:18-33,162-191,291-295)."""
    blob = {}
    index = []
    for players in (3, 4):
        for name, (rules, fp, pool) in PLAYER_RULESETS.items():
            rules = {k: (players if v == "P" else v) for k, v in rules.items()}
            fp = players if fp == "P" else fp
            for seed in range(8):
                key = "p%d_%s_s%d" % (players, name, seed)
                d = play_players_stream(seed, players, rules)
                for k, v in d.items():
                    blob[key + "_" + k] = v
                index.append((players, fp, pool, seed, key))
    blob["index_players"] = np.array([i[0] for i in index])
    blob["index_first"] = np.array([i[1] for i in index])
    blob["index_pool"] = np.array([i[2] for i in index])
    blob["index_seed"] = np.array([i[3] for i in index])
    blob["index_key"] = np.array([i[4] for i in index])
    np.savez_compressed(os.path.join(OUT, "traj_players.npz"), **blob)
    ends = collections.Counter(str(blob[i[4] + "_end"]) for i in index)
    print("wrote traj_players.npz (%d streams; endings: %s)" % (len(index), dict(ends)))


SELFPLAY_FIELDS = ["mask", "action", "done", "displays", "center", "pattern_lines", "walls", "floors", "score", "cur", "nfp", "eog_flag",
                   "turn_counter", "box", "lid", "first_player_stats", "floor_penalty", "max_combo", "completed_lines", "rng_words"]


def play_players_selfplay(seed, players, rules, n_moves):
    """The flat random-agent loop for P players, every decision by the reference's own RandomAgent on the process-global stream:
        random.seed(seed); g = Azul(players=P, rules=rules); g.new_round()
        repeat: a = RandomAgent().get_a_output(None, mask of g); g.step(*nn_deserialize(a)); when g.end_of_game: a fresh Azul + new_round()
    (game_runner.py:87-97 works on any mask; azul.py:296-313 is P-generic).  Recorded per move: mask before, action, done, the
    whole state after the move (before the restart), the words of the stream consumed so far (restart draws included)."""
    from azulnet import nn_deserialize
    rows = {k: [] for k in SELFPLAY_FIELDS}
    random.seed(seed)
    words = [0, words_pos()]

    def sync():
        pos = words_pos()
        d = pos - words[1]
        if d < 0:
            d += 624
        words[0] += d
        words[1] = pos
        return words[0]

    agent = RandomAgent()
    g = Azul(players=players, rules=dict(rules))
    g.new_round()
    episodes = stuck = 0
    stats_sum = np.zeros(10)
    for t in range(n_moves):
        mask = check_all_valid(g)
        try:
            a = int(agent.get_a_output(None, torch.from_numpy(np.asarray(mask)[None, :])))
        except ValueError:                             # hazard H3: nothing legal (raised before random() is drawn)
            a = -1
        r = rows
        r["mask"].append(np.packbits(mask, bitorder="little"))
        r["action"].append(a)
        if a >= 0:
            g.step(*nn_deserialize(a))
        done = 2 if a < 0 else int(bool(g.end_of_game))
        r["done"].append(done)
        r["displays"].append(np.array(g.game_board_displays, dtype=np.uint8))
        r["center"].append(np.array(g.game_board_center, dtype=np.uint8))
        r["pattern_lines"].append(np.array(g.pattern_lines, dtype=np.uint8))
        r["walls"].append(np.array(g.walls, dtype=np.uint8))
        r["floors"].append(np.array(g.floors, dtype=np.uint8))
        r["score"].append(np.array(g.score, dtype=np.int16))
        r["cur"].append(g.current_player)
        r["nfp"].append(g.next_first_player)
        r["eog_flag"].append(bool(g.end_of_game))
        r["turn_counter"].append(g.turn_counter)
        r["box"].append(np.array(g.box_tiles, dtype=np.uint8) if g.tile_pool == "Lid" else np.zeros(5, np.uint8))
        r["lid"].append(np.array(g.lid_tiles, dtype=np.uint8) if g.tile_pool == "Lid" else np.zeros(5, np.uint8))
        r["first_player_stats"].append(np.array(g.first_player_stats, dtype=np.uint16))
        r["floor_penalty"].append(np.array(g.floor_penalty, dtype=np.int16))
        r["max_combo"].append(np.array(g.max_combo, dtype=np.uint8))
        r["completed_lines"].append(np.array(g.completed_lines, dtype=np.uint8))
        if done:
            if done == 1:
                st = g.get_statistics()
                stats_sum += np.array([float(st[k]) for k in ["player_score", "opponent_score", "rounds", "percent_first_player", "floor_penalty",
                                                               "max_combo", "completed_rows", "completed_columns", "completed_colors", "win_percent"]])
                episodes += 1
            else:
                stuck += 1
            g = Azul(players=players, rules=dict(rules))
            g.new_round()
        r["rng_words"].append(sync())
    out = {k: np.array(v) for k, v in rows.items()}
    out.update({"episodes": np.array(episodes), "stuck": np.array(stuck), "stats_sum": stats_sum})
    return out


def gen_players_selfplay():
    """Row N4: the flat self-play loop for 3 and 4 players (what azul_batch_selfplay plays for such batches)."""
    blob, index = {}, []
    for players in (3, 4):
        for name, (rules, fp, pool) in PLAYER_RULESETS.items():
            rules = {k: (players if v == "P" else v) for k, v in rules.items()}
            fp = players if fp == "P" else fp
            for seed in range(3):
                key = "p%d_%s_s%d" % (players, name, seed)
                d = play_players_selfplay(1000 + seed, players, rules, 360)
                for k, v in d.items():
                    blob[key + "_" + k] = v
                index.append((players, fp, pool, 1000 + seed, key))
    blob["index_players"] = np.array([i[0] for i in index])
    blob["index_first"] = np.array([i[1] for i in index])
    blob["index_pool"] = np.array([i[2] for i in index])
    blob["index_seed"] = np.array([i[3] for i in index])
    blob["index_key"] = np.array([i[4] for i in index])
    np.savez_compressed(os.path.join(OUT, "traj_players_selfplay.npz"), **blob)
    print("wrote traj_players_selfplay.npz (%d streams, %d episodes, %d stuck)" % (
        len(index), sum(int(blob[i[4] + "_episodes"]) for i in index), sum(int(blob[i[4] + "_stuck"]) for i in index)))


def gen_net_opponent():
    """GameRunner(opponent=Agent(...)) (game_runner.py:27-30): every opponent_move -- and every forced move of player 1 -- goes through a
    second ActorCritic on the perspective-rotated observation (game_runner.py:37-42, :46; agent.py:73-81); reset() lets the opponent open
    (:84-85).  scripts/run_batch.py:6-10 and tests/test_nn_runner.py:63-67, 84-90 run exactly this.  Recorded per seed and rule set, two
    episodes: per AGENT step what NNRunner.run_episode sees (nn_runner.py:22-30: observation, mask, action, value, log-prob, entropy term,
    reward, done, move_counter, player_score) and per OPPONENT call what get_a_output was handed (state, mask, the player it moved for)
    and what it answered (action, its probability, the full distribution for the first calls).  The two nets are the reference's own
    default-initialised ActorCritic modules (torch.manual_seed(0)); the env draws from `random`, the agents from np.random."""
    from azulnet import Agent
    torch.manual_seed(0)
    agent, opponent = Agent(), Agent()
    blob = {}
    for k, v in agent.ac_net.state_dict().items():
        blob["agent_sd_" + k] = v.numpy().copy()
    for k, v in opponent.ac_net.state_dict().items():
        blob["opp_sd_" + k] = v.numpy().copy()
    calls = collections.defaultdict(list)
    orig_forward = opponent.ac_net.forward_actor
    seen = {}

    def forward_actor(state_tensor, mask=None):                     # records the distribution the opponent samples from
        out = orig_forward(state_tensor, mask)
        seen["dist"] = out[0].detach().numpy().squeeze(0).copy()
        return out

    opponent.ac_net.forward_actor = forward_actor
    orig_get = opponent.get_a_output
    runner_ref = [None]

    def get_a_output(state, valid_moves, action_selection="Distribution"):
        a = orig_get(state, valid_moves, action_selection)
        c = calls
        c["state"].append(np.asarray(state).astype(np.int16))
        c["mask"].append(np.packbits(valid_moves.numpy().reshape(180), bitorder="little"))
        c["action"].append(int(a))
        c["player"].append(int(runner_ref[0].game.current_player))
        c["move_counter"].append(int(runner_ref[0].move_counter))
        c["prob"].append(np.float32(seen["dist"][int(a)]))
        c["dist"].append(seen["dist"].astype(np.float32))
        return a

    opponent.get_a_output = get_a_output
    names, seeds = ["lid_randomfirst", "random_first1"], list(range(6))
    blob["rulesets"] = np.array(names)
    blob["seeds"] = np.array(seeds, dtype=np.uint64)
    for name in names:
        rules = RULESETS[name][0]
        for s in seeds:
            calls.clear()
            random.seed(s)
            np.random.seed(s)
            runner = GameRunner(opponent=opponent, rules=dict(rules))
            runner_ref[0] = runner
            rows = collections.defaultdict(list)
            for _ in range(2):
                runner.reset()                                      # nn_runner.py:20
                rows["episode_first_step"].append(len(rows["action"]))
                rows["episode_calls_before"].append(len(calls["action"]))
                done = False
                while not done:
                    mask = runner.get_valid_moves()
                    obs = runner.get_state()
                    a, dist, logd, value = agent.get_ac_output(obs, torch.from_numpy(mask.reshape(1, 180)))     # nn_runner.py:24-25
                    reward, done = runner.step(a)                                                            # :26
                    vm = torch.from_numpy(mask.reshape(1, 180))
                    rows["obs"].append(obs.astype(np.int16))
                    rows["mask"].append(np.packbits(mask, bitorder="little"))
                    rows["action"].append(int(a))
                    rows["value"].append(np.float32(value.item()))
                    rows["log_prob"].append(np.float32(logd.squeeze(0)[a].item()))                          # :29
                    rows["entropy"].append(np.float32((-logd.masked_select(vm).mean()).item()))            # :33-37
                    rows["reward"].append(int(reward))
                    rows["done"].append(bool(done))
                    rows["move_counter"].append(int(runner.move_counter))
                    rows["player_score"].append(int(runner.player_score))
                    rows["calls_after"].append(len(calls["action"]))
                rows["final_obs"].append(runner.get_state().astype(np.int16))
                st = runner.game.get_statistics()
                rows["stats"].append(np.array([float(st[k]) for k in
                                               ["player_score", "opponent_score", "rounds", "percent_first_player", "floor_penalty",
                                                "max_combo", "completed_rows", "completed_columns", "completed_colors", "win_percent"]]))
            pre = "%s_s%d_" % (name, s)
            for k, v in rows.items():
                blob[pre + "step_" + k] = np.array(v)
            for k, v in calls.items():
                blob[pre + "call_" + k] = np.array(v if k != "dist" else v[:24])
    np.savez_compressed(os.path.join(OUT, "net_opponent.npz"), **blob)
    print("wrote net_opponent.npz")


def main():
    os.makedirs(OUT, exist_ok=True)
    print("reference:", os.path.dirname(azulnet.__file__))
    if len(sys.argv) > 1 and sys.argv[1] == "policy":
        gen_policy_contract()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "a2c":
        gen_a2c_update()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "players":
        gen_players()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "net_opponent":
        gen_net_opponent()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "players_selfplay":
        gen_players_selfplay()
        return
    gen_rng()
    gen_boards()
    gen_trajectories()
    gen_policy_contract()
    gen_a2c_update()
    gen_players()
    gen_players_selfplay()
    gen_net_opponent()


if __name__ == "__main__":
    main()
