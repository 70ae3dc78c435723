"""Shim: ``azulnet.azul`` served by the MI355X backend (replaces the reference's azulnet/azul.py)."""
from azul_deep_reinforcement_learning_amd.azul import Azul, IllegalMove, GameEnded, IllegalRule  # noqa: F401
