"""SequentialRandomWalkGenerator -- mirrors routing_board_generation/rl_training/
online_generators/sequential_random_walk_generator.py:21-66."""
import torch

from .. import ops
from ..board_generation.sequential_random_walk import SequentialRandomWalkBoard
from ..types import State
from .uniform_generator import Generator, state_from_bufs


class SequentialRandomWalkGenerator(Generator):
    def __init__(self, grid_size: int, num_agents: int) -> None:
        super().__init__(grid_size, num_agents)
        self.board_generator = SequentialRandomWalkBoard(self.grid_size, self.grid_size,
                                                         self.num_agents)

    def __call__(self, key: torch.Tensor) -> State:
        batched = key.dim() == 2
        keys = (key if batched else key[None]).contiguous()
        out = ops.seq_rw_generate(keys, self.grid_size, self.num_agents, split_first=True,
                                  want_solved=False, want_train=True)
        st = {"grid": out["train_grid"], "start": out["start"], "target": out["target"],
              "position": out["position"], "key": out["key_out"],
              "step_count": out["step_count"]}
        return state_from_bufs(st, batched)

    def native_spec(self):
        from .. import _native as nat

        # a refill launch lasts as long as its slowest board (~0.4 ms at 16x16/10) however few rows
        # it generates, and leaves most of the chip idle: rolling refills beside the stepping
        # stream (ops.EpochStash(serial=False)), one generator stream, every entry requested
        # seven episodes ahead; on misses (short episodes) the stash falls back to the serial,
        # miss-adaptive form (17.9 us per step against 22.5 for one refill per
        # 48 steps in line)
        return ops.GeneratorSpec(nat.GEN_SEQUENTIAL_RW,
                                 refill={"every": 32, "depth": 8, "serial": False, "lead": 3,
                                         "streams": 1, "rollout_stash": False})
