"""Pytrees of the Connector environment with torch device tensors as leaves.

Mirrors [3P] jumanji.environments.routing.connector.types (Agent, State, Observation) and
[3P] jumanji.types (StepType, TimeStep) of jumanji==0.2.2, the versions the reference pins
(requirements.txt:5).  Field names and order follow the State printed by the reference at
package_evaluation/profiling_generators.ipynb cell 13: State(grid, step_count, agents, key),
Agent(id, start, target, position).  A leading batch axis B is optional everywhere.
"""
from typing import Any, Dict, NamedTuple

import torch


class Agent(NamedTuple):
    id: torch.Tensor        # (..., N) int32
    start: torch.Tensor     # (..., N, 2) int32
    target: torch.Tensor    # (..., N, 2) int32
    position: torch.Tensor  # (..., N, 2) int32

    @property
    def connected(self) -> torch.Tensor:
        """True where the agent's head sits on its target (types.py::Agent.connected)."""
        return (self.position == self.target).all(dim=-1)


class State(NamedTuple):
    grid: torch.Tensor        # (..., m, m) int32
    step_count: torch.Tensor  # (...,) int32
    agents: Agent
    key: torch.Tensor         # (..., 2) uint32


class Observation(NamedTuple):
    grid: torch.Tensor         # (..., N, m, m) int32 -- per-agent view
    action_mask: torch.Tensor  # (..., N, 5) bool
    step_count: torch.Tensor   # (...,) int32


class StepType:
    FIRST = 0
    MID = 1
    LAST = 2


class TimeStep(NamedTuple):
    step_type: torch.Tensor  # (...,) int8
    reward: torch.Tensor     # (..., N) float32  (summed to (...,) by MultiToSingleWrapper)
    discount: torch.Tensor   # (..., N) float32
    observation: Observation
    extras: Dict[str, Any]

    def first(self) -> torch.Tensor:
        return self.step_type == StepType.FIRST

    def mid(self) -> torch.Tensor:
        return self.step_type == StepType.MID

    def last(self) -> torch.Tensor:
        return self.step_type == StepType.LAST
