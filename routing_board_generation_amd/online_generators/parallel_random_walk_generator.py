"""ParallelRandomWalkGenerator -- mirrors routing_board_generation/rl_training/
online_generators/parallel_random_walk_generator.py:29-77."""
import torch

from .. import ops
from ..board_generation.parallel_random_walk import ParallelRandomWalkBoard
from ..types import State
from .uniform_generator import Generator, state_from_bufs


class ParallelRandomWalkGenerator(Generator):
    """Generates solvable `Connector` boards with the parallel random walk."""

    def __init__(self, grid_size: int, num_agents: int) -> None:
        super().__init__(grid_size, num_agents)
        self.board_generator = ParallelRandomWalkBoard(self.grid_size, self.grid_size,
                                                       self.num_agents)

    def __call__(self, key: torch.Tensor) -> State:
        batched = key.dim() == 2
        keys = key if batched else key[None]
        out = ops.prw_generate(keys.contiguous(), self.grid_size, self.num_agents,
                               split_first=True, want_solved=False, want_train=True,
                               want_state=True)
        st = {"grid": out["train_grid"], "start": out["start"], "target": out["target"],
              "position": out["position"], "key": out["key_out"],
              "step_count": out["step_count"]}
        return state_from_bufs(st, batched)

    def native_spec(self):
        from .. import _native as nat

        # one refill per 20 steps (fine sweep, DESIGN.md 4.2a: 12 / 16 / 20 / 24 / 28 / 32 steps ->
        # 15.2 / 14.7 / 14.3 / 14.4 / 14.9 / 16.3 us per step at B = 4096)
        return ops.GeneratorSpec(nat.GEN_PARALLEL_RW, refill={"every": 20})


class ParallelRandomWalkPhiloxGenerator(Generator):
    """The same walk on a counter-based Philox-4x32-10 stream instead of jax's Threefry key
    splitting (north_star's "Philox kernel replacing jax.random").  OPT-IN and
    defined by this project: boards are valid, solvable and deterministic per key, but they are
    NOT the reference's boards for that key -- the default `ParallelRandomWalkGenerator` is the
    bit-exact one.  The stream is specified in include/rbg_prw.h.  A `Connector` built on it
    gets the same fused reset / next-episode stash / fused rollouts as the other generators
    (RBG_GEN_PARALLEL_RW_PHILOX)."""

    def __call__(self, key: torch.Tensor) -> State:
        batched = key.dim() == 2
        keys = key if batched else key[None]
        out = ops.prw_generate_philox(keys.contiguous(), self.grid_size, self.num_agents)
        st = {"grid": out["train_grid"], "start": out["start"], "target": out["target"],
              "position": out["position"], "key": out["key_out"],
              "step_count": out["step_count"]}
        return state_from_bufs(st, batched)

    def native_spec(self):
        from .. import _native as nat

        return ops.GeneratorSpec(nat.GEN_PARALLEL_RW_PHILOX, refill={"every": 20})
