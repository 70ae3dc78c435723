"""``GameRunner`` / ``RandomAgent`` / ``check_all_valid`` -- mirror of azulnet/game_runner.py on the GPU backend.

``GameRunner.step`` with the default ``RandomAgent`` opponent is ONE launch (agent move, opponent loop,
shaped reward, done: azul_batch_runner_step).  With any other opponent object (anything exposing
``get_a_output(state, valid_moves)``, e.g. the reference's ``Agent``) the opponent loop runs on the host and
every rule evaluation inside it is a kernel launch.
"""
import copy

import numpy as np
import torch

from . import _lib as L
from .azul import Azul, GameEnded, IllegalMove
from .codec import nn_deserialize, nn_serialize
from .records import STAT_KEYS


class GameStatistics:
    """What GameRunner.game_statistics collects (game_runner.py:10-22): the ten get_statistics() values of every finished game
    are buffered per key; get_stats() folds the buffered games into ONE mean per key, appends it to `statistics` and empties
    the buffer.  (The batched trainer gets the same means from sums the kernels accumulate on the device.)"""

    def __init__(self):
        self._pending = {key: [] for key in STAT_KEYS}
        self.statistics = {key: np.empty(0) for key in STAT_KEYS}

    @property
    def statisticsBuffer(self):
        return {key: np.asarray(vals, dtype=float) for key, vals in self._pending.items()}

    def update(self, statistics):
        for key, value in statistics.items():
            self._pending[key].append(value)

    def get_stats(self):
        for key, vals in self._pending.items():
            if vals:
                self.statistics[key] = np.append(self.statistics[key], float(np.mean(vals)))
                vals.clear()
        return self.statistics


class GameRunner:
    GameStatistics = GameStatistics          # the reference nests the class (game_runner.py:10)

    def __init__(self, opponent=None, rules={"first_player": "Random", "tile_pool": "Lid"}):
        self.rules = rules
        self.game_statistics = GameStatistics()
        self.opponent = RandomAgent() if opponent is None else opponent
        self.player_score = 0
        self.move_counter = 0
        self.game = Azul(rules=rules)
        self.game.new_round()

    def _device_opponent(self):
        return type(self.opponent) is RandomAgent

    def opponent_move(self):
        """One move of `self.opponent` for the player to move (game_runner.py:37-42), policy evaluated on the host."""
        seat = self.game.current_player - 1
        legal = torch.from_numpy(self.get_valid_moves()[None, :])
        # (RandomAgent never looks at the state -- game_runner.py:93-97 -- so the observation is not computed for it)
        choice = self.opponent.get_a_output(None if self._device_opponent() else self.get_state(perspective=seat), legal)
        self.game.step(*nn_deserialize(choice))
        self.move_counter += 1

    def _raise_for(self, status):
        if status in (L.ILLEGAL_MOVE, L.BAD_ACTION):
            raise IllegalMove
        if status == L.GAME_ENDED:
            raise GameEnded
        if status == L.STUCK:
            raise ValueError("Total of weights must be greater than zero")   # RandomAgent with no legal move

    def step(self, i):
        if self._device_opponent():
            # the agent's move, the RandomAgent's replies, the shaped reward and `done` in ONE launch (game_runner.py:43-55)
            reward, done, status = self.game._run("op_runner_step", int(i), draws=True, runner=self)
            self._raise_for(status)
        else:
            self.game.step(*nn_deserialize(i))
            self.move_counter += 1
            # the opponent plays until it is player 1's turn with a real choice, or the game is over (game_runner.py:46-47)
            while not self.game.is_end_of_game() and (self.game.current_player != 1 or np.count_nonzero(self.get_valid_moves()) < 2):
                self.opponent_move()
            potential = self.game._run("op_potential", mutates=False)         # deepcopy + count_score, score[0] - score[1] (:48-50)
            reward, self.player_score = potential - self.player_score, potential
            done = self.game.is_end_of_game()
        if done:
            self.game_statistics.update(self.game.get_statistics())      # game_runner.py:53-54
        return reward, done

    def get_state(self, perspective=0):
        return self.game._run("op_observe", int(perspective), mutates=False)

    def get_valid_moves(self):
        return check_all_valid(self.game)

    def reset(self):
        """A fresh game, first round dealt, the opponent opening if it starts (game_runner.py:76-85)."""
        self.player_score = 0
        self.move_counter = 0
        self.game = Azul(rules=self.rules)
        self.game.new_round()
        while self.game.current_player != 1:
            self.opponent_move()


class RandomAgent:
    """Weighted random legal move (floor moves 0.01, others 1.0), one ``random.choices`` draw from the global
    CPython stream -- evaluated on the GPU (game_runner.py:87-97)."""

    def __init__(self):
        self.weight_table = np.ones(180)
        self.weight_table[:30] = 0.01

    def get_a_output(self, state, valid_moves):
        from . import facade_backend as fb
        mask = valid_moves.numpy() if hasattr(valid_moves, "numpy") else np.asarray(valid_moves)
        # 180 actions in the reference; 240 / 300 with the 2P+1 displays rule (beyond the reference): weight 0.01 for the floor row
        a = fb.sampling_backend(mask.size).sample(mask)
        if a < 0:
            raise ValueError("Total of weights must be greater than zero")
        return a


def check_all_valid(game):
    """bool[180]: is_legal_move for every action (game_runner.py:113-117), one launch."""
    return game.legal_mask()


__all__ = ["GameRunner", "RandomAgent", "check_all_valid", "nn_serialize", "nn_deserialize"]
