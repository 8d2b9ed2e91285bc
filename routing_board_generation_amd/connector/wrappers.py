"""[3P] jumanji.wrappers.{VmapAutoResetWrapper, MultiToSingleWrapper} as used at
routing_board_generation/rl_training/setup_train.py:158-167.

The native env is already batched, so VmapAutoResetWrapper only switches on the fused auto-reset
of `rbg_connector_step`, and MultiToSingleWrapper the fused per-env reward sum / discount max.
"""
from typing import Tuple

import torch

from ..types import State, TimeStep


class _Wrapper:
    def __init__(self, env):
        self._env = env

    def __getattr__(self, name):
        return getattr(self._env, name)

    @property
    def unwrapped(self):
        return self._env.unwrapped


class MultiToSingleWrapper(_Wrapper):
    """reward = sum over agents, discount = max over agents ([3P] jumanji.wrappers).  The native
    Connector aggregates inside its step launch (rbg_timestep.reward_total / discount_max), so
    this wrapper only switches that on."""

    def __init__(self, env):
        super().__init__(env)
        env.unwrapped._configure(aggregate=True)

    def reset(self, key: torch.Tensor) -> Tuple[State, TimeStep]:
        return self._env.reset(key)

    def step(self, state: State, action: torch.Tensor) -> Tuple[State, TimeStep]:
        return self._env.step(state, action)


class VmapAutoResetWrapper(_Wrapper):
    """reset(keys (B, 2)) / step(batched state, action (B, N)); finished episodes restart inside
    the same kernel launch from split(state.key)[0], keeping reward/discount/step_type/extras of
    the terminal step and replacing only the observation."""

    def __init__(self, env, prefetch: bool = True, prefetch_delay=None, donate=None):
        """prefetch=True (default) generates every env's NEXT episode ahead of time
        (include/rbg_prw.h::rbg_stash): one refill launch per `prefetch_delay` steps runs the
        generator for all the envs that restarted since the last one, instead of every step launch
        waiting for the generator runs of its own finished envs; results are identical with and
        without it.  prefetch_delay=None takes the generator's measured best
        (ops.GeneratorSpec.refill: 20 steps in line for ParallelRandomWalk; refills rolling beside
        the stepping stream for SequentialRandomWalk / SeedExtension -- every 32 / 64 steps with
        8 / 10 slots per env, shortened to fit the time limit; none for Uniform and the offline
        dataset, whose "generation" is cheaper than a stash row copy).  It was 16 until round 2.
        donate=True/False overrides the env's setting (Connector(donate=...))."""
        super().__init__(env)
        env.unwrapped._configure(auto_reset=True,
                                 prefetch_delay=(0 if not prefetch else None
                                                 if prefetch_delay is None else int(prefetch_delay)))
        if donate is not None:
            env.unwrapped._configure(donate=bool(donate))

    def reset(self, key: torch.Tensor) -> Tuple[State, TimeStep]:
        if key.dim() != 2:
            raise ValueError("VmapAutoResetWrapper.reset expects keys of shape (B, 2)")
        return self._env.reset(key)

    def step(self, state: State, action: torch.Tensor) -> Tuple[State, TimeStep]:
        return self._env.step(state, action)

    def random_rollout(self, state: State, policy_keys: torch.Tensor, t0: int, n_steps: int,
                       want_timestep: bool = False, record: bool = False):
        """n_steps of auto-resetting random-policy stepping in one launch (Connector.random_rollout;
        record=True: the deprecated round-2 spelling of random_rollout_record)."""
        return self._env.unwrapped.random_rollout(state, policy_keys, t0, n_steps, want_timestep,
                                                  record)

    def random_rollout_record(self, state: State, policy_keys: torch.Tensor, t0: int,
                              n_steps: int):
        """... with every step's TimeStep and action recorded (Connector.random_rollout_record)."""
        return self._env.unwrapped.random_rollout_record(state, policy_keys, t0, n_steps)
