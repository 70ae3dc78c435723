"""Policy/value network bridge for policy-driven self-play (BASELINE configs[2]).

``BatchedActorCritic`` honours the call contract of the reference's ``ActorCritic`` (azulnet/model.py:12-41) --
same constructor, same parameter names (``critic_linear1/2``, ``actor_linear1/2``: a reference state_dict or a
pickled reference module's weights load directly), same outputs: ``forward_critic(state) -> [B,1]`` and
``forward_actor(state, mask) -> (softmax, log_softmax)`` over the 180 actions with illegal logits at -inf -- but for
whole batches whose masks live on the GPU (the reference calls ``mask.numpy()`` and therefore needs CPU masks).
The dense 136x180 / 180x180 GEMMs are stock PyTorch-ROCm (rocBLAS/hipBLASLt): SURVEY.md 2 marks them out of scope
as kernels; only the contract matters here.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class IllegalMask(Exception):
    pass


class BatchedActorCritic(nn.Module):
    def __init__(self, num_inputs=136, num_actions=180, hidden_size=180):
        super().__init__()
        self.num_actions = num_actions
        self.critic_linear1 = nn.Linear(num_inputs, hidden_size)
        self.critic_linear2 = nn.Linear(hidden_size, 1)
        self.actor_linear1 = nn.Linear(num_inputs, hidden_size)
        self.actor_linear2 = nn.Linear(hidden_size, num_actions)

    def forward_critic(self, state):
        return self.critic_linear2(F.relu(self.critic_linear1(state)))

    def forward_actor(self, state, mask=None, check=False):
        """mask: bool/uint8 [B,180] on the same device as `state`.  `check=True` raises IllegalMask when the whole
        batch has no legal action, like the reference (model.py:33-34); it synchronises, so rollouts leave it off and
        treat mask-less rows as stuck games instead."""
        logits = self.actor_linear2(F.relu(self.actor_linear1(state)))
        if mask is not None:
            mask = mask.bool()
            if check and not bool(mask.any()):
                raise IllegalMask
            logits = logits.masked_fill(~mask, float("-inf"))
        return F.softmax(logits, dim=1), F.log_softmax(logits, dim=1)

    @classmethod
    def from_reference(cls, module_or_state_dict):
        sd = module_or_state_dict.state_dict() if hasattr(module_or_state_dict, "state_dict") else module_or_state_dict
        hidden, num_in = sd["actor_linear1.weight"].shape
        net = cls(num_in, sd["actor_linear2.weight"].shape[0], hidden)
        net.load_state_dict(sd)
        return net
