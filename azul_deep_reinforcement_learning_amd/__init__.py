"""azul_deep_reinforcement_learning_amd -- MI355X-native batched Azul environment + self-play rollout.

Host-side mirror of the reference's ``azulnet`` API for the environment hot path (azul.py +
game_runner.py); every rule evaluation runs in hand-written gfx950 kernels behind the C ABI of
``libazulhip.so`` (include/azul_hip.h).  There is no CPU execution path.
"""
from .records import RECORD_DTYPE, RECORD_NP_DTYPE, STAT_KEYS  # noqa: F401
from .codec import nn_serialize, nn_deserialize  # noqa: F401
from .batch import BatchedAzul, IllegalRule, parse_rules  # noqa: F401
from .azul import Azul, IllegalMove, GameEnded  # noqa: F401
from .game_runner import GameRunner, RandomAgent, check_all_valid  # noqa: F401
from .policy import BatchedActorCritic, IllegalMask  # noqa: F401
from .rollout import PolicyRollout  # noqa: F401
from .learner import A2CLearner  # noqa: F401
from .training import BatchedTrainer  # noqa: F401
