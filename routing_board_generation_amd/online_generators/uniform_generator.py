"""Generator ABC (the plug-in boundary of Connector) and UniformRandomGenerator.

Mirrors routing_board_generation/rl_training/online_generators/uniform_generator.py:26-109.
"""
import abc

import torch

from .. import ops
from ..types import Agent, State


def state_from_bufs(st, batched: bool) -> State:
    B, N = st["start"].shape[0], st["start"].shape[1]
    ids = torch.arange(N, dtype=torch.int32, device=st["start"].device).expand(B, N)
    state = State(grid=st["grid"], step_count=st["step_count"],
                  agents=Agent(id=ids, start=st["start"], target=st["target"],
                               position=st["position"]),
                  key=st["key"])
    return state if batched else unbatch_state(state)


def unbatch_state(state: State) -> State:
    return State(grid=state.grid[0], step_count=state.step_count[0],
                 agents=Agent(*[x[0] for x in state.agents]), key=state.key[0])


class Generator(abc.ABC):
    """Base class for generators for the connector environment."""

    def __init__(self, grid_size: int, num_agents: int) -> None:
        self._grid_size = grid_size
        self._num_agents = num_agents

    @property
    def grid_size(self) -> int:
        return self._grid_size

    @property
    def num_agents(self) -> int:
        return self._num_agents

    @abc.abstractmethod
    def __call__(self, key: torch.Tensor) -> State:
        """key (2,) -> State; key (B, 2) -> batched State (replaces jax.vmap(generator))."""

    def native_spec(self):
        """ops.GeneratorSpec of a generator that the env launches can run themselves (fused
        reset, next-episode stash, fused rollouts: include/rbg_prw.h::rbg_generator), or None: a
        user-defined generator is called from Python and `Connector` selects per env afterwards."""
        return None


class UniformRandomGenerator(Generator):
    """Start and target cells drawn uniformly without replacement (may be unsolvable)."""

    def __call__(self, key: torch.Tensor) -> State:
        batched = key.dim() == 2
        keys = key if batched else key[None]
        st = ops.uniform_generate(keys.contiguous(), self.grid_size, self.num_agents)
        return state_from_bufs(st, batched)

    def native_spec(self):
        from .. import _native as nat

        # drawing 2N cells costs less than copying a stash row: no stash (330 vs 323 M env-steps/s)
        return ops.GeneratorSpec(nat.GEN_UNIFORM, refill={"every": 0, "rollout_stash": False})
