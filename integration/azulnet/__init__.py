"""Drop-in ``azulnet`` package: the reference's learner files (agent.py, model.py, nn_runner.py) are used
UNCHANGED next to these shims; only azul.py and game_runner.py are replaced by the MI355X backend.
See INTEGRATION.md."""
from azulnet.azul import Azul  # noqa: F401
from azulnet.game_runner import GameRunner, RandomAgent, check_all_valid, nn_serialize, nn_deserialize  # noqa: F401

try:  # present once the reference's own agent.py / model.py / nn_runner.py are copied in beside the shims
    from azulnet.model import ActorCritic  # noqa: F401
    from azulnet.agent import Agent  # noqa: F401
    from azulnet.nn_runner import NNRunner  # noqa: F401
except ImportError:  # pragma: no cover
    pass
