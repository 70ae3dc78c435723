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


class GameRunner:
    class GameStatistics:
        # game_runner.py:10-22
        def __init__(self):
            self.statisticsBuffer = {key: np.empty(0) for key in STAT_KEYS}
            self.statistics = {key: np.empty(0) for key in STAT_KEYS}

        def update(self, statistics):
            for stat in statistics:
                self.statisticsBuffer[stat] = np.append(self.statisticsBuffer[stat], statistics[stat])

        def get_stats(self):
            for stat in self.statistics:
                if len(self.statisticsBuffer[stat]) > 0:
                    self.statistics[stat] = np.append(self.statistics[stat], self.statisticsBuffer[stat].mean())
                    self.statisticsBuffer[stat] = np.empty(0)
            return self.statistics

    def __init__(self, opponent=None, rules={"first_player": "Random", "tile_pool": "Lid"}):
        self.game = Azul(rules=rules)
        self.rules = rules
        self.game_statistics = GameRunner.GameStatistics()
        self.opponent = opponent if opponent is not None else RandomAgent()
        self.game.new_round()
        self.player_score = 0
        self.move_counter = 0

    def _device_opponent(self):
        return type(self.opponent) is RandomAgent

    def opponent_move(self):
        # game_runner.py:37-42
        state = self.get_state(perspective=self.game.current_player - 1)
        valid_moves = torch.from_numpy(self.get_valid_moves().reshape(1, 180))
        action = self.opponent.get_a_output(state, valid_moves)
        self.game.step(*nn_deserialize(action))
        self.move_counter += 1

    def step(self, i):
        if self._device_opponent():
            reward, done, st = self.game._run("op_runner_step", int(i), draws=True, runner=self)
            if st == L.ILLEGAL_MOVE or st == L.BAD_ACTION:
                raise IllegalMove
            if st == L.GAME_ENDED:
                raise GameEnded
            if st == L.STUCK:
                raise ValueError("Total of weights must be greater than zero")   # RandomAgent with no legal move
        else:
            # game_runner.py:43-52 with the opponent policy evaluated on the host
            self.game.step(*nn_deserialize(i))
            self.move_counter += 1
            while (self.game.current_player != 1 or np.count_nonzero(self.get_valid_moves()) < 2) and not self.game.is_end_of_game():
                self.opponent_move()
            new_player_score = self.game._run("op_potential", mutates=False)
            reward = new_player_score - self.player_score
            self.player_score = new_player_score
            done = self.game.is_end_of_game()
        if done:
            self.game_statistics.update(self.game.get_statistics())      # game_runner.py:53-54
        return reward, done

    def get_state(self, perspective=0):
        return self.game._run("op_observe", int(perspective), mutates=False)

    def get_valid_moves(self):
        return check_all_valid(self.game)

    def reset(self):
        # game_runner.py:76-85
        self.game = Azul(rules=self.rules)
        self.game.new_round()
        self.player_score = 0
        self.move_counter = 0
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
        mask = np.asarray(valid_moves.numpy() if hasattr(valid_moves, "numpy") else valid_moves).reshape(-1)[:180]
        be = fb.backend(1, L.POOL_RANDOM)
        be.push_rng()
        a = be.op_sample_mask(mask.astype(np.uint8))
        if a < 0:
            raise ValueError("Total of weights must be greater than zero")
        be.pull_rng()
        return a


def check_all_valid(game):
    """bool[180]: is_legal_move for every action (game_runner.py:113-117), one launch."""
    return game.legal_mask()


__all__ = ["GameRunner", "RandomAgent", "check_all_valid", "nn_serialize", "nn_deserialize"]
