"""Shim: ``azulnet.game_runner`` served by the MI355X backend (replaces the reference's azulnet/game_runner.py).
The reference's tests do ``from azulnet.game_runner import *`` and then use np / torch, so those names are
re-exported like the original module did."""
import numpy as np  # noqa: F401
import torch  # noqa: F401

from azul_deep_reinforcement_learning_amd.azul import Azul  # noqa: F401
from azul_deep_reinforcement_learning_amd.game_runner import (  # noqa: F401
    GameRunner, RandomAgent, check_all_valid, nn_serialize, nn_deserialize)
