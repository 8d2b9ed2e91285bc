"""SeedExtensionGenerator -- mirrors routing_board_generation/rl_training/online_generators/
random_seed_generator.py:16-57 (the notebook of the reference still calls it RandomSeedGenerator,
its former name; both names are exported)."""
import torch

from .. import ops
from ..board_generation.seed_extension import SeedExtensionBoard
from ..types import State
from .uniform_generator import Generator, state_from_bufs


class SeedExtensionGenerator(Generator):
    def __init__(self, grid_size: int, num_agents: int) -> None:
        super().__init__(grid_size, num_agents)
        self.board_generator = SeedExtensionBoard(self.grid_size, self.grid_size,
                                                  self.num_agents)

    def __call__(self, key: torch.Tensor) -> State:
        batched = key.dim() == 2
        keys = (key if batched else key[None]).contiguous()
        # :34-37 key, _ = split(key); generate_starts_ends(key) with its defaults
        out = ops.seed_extension_generate(keys, self.grid_size, self.num_agents, split_first=True,
                                          want_solved=False, want_train=True)
        st = {"grid": out["train_grid"], "start": out["start"], "target": out["target"],
              "position": out["position"], "key": out["key_out"],
              "step_count": out["step_count"]}
        return state_from_bufs(st, batched)

    def native_spec(self):
        from .. import _native as nat

        # random_seed_generator.py:36-37: generate_starts_ends(key) with its defaults
        # (refill: a launch of 1.4-2.2 ms whatever its row count -- rolling refills beside the
        # stepping stream, two in flight on two generator streams, every entry requested nine
        # episodes ahead: 18.2 us per step against 42-47 with one refill per 48 steps in line;
        # DESIGN.md 4.2a)
        return ops.GeneratorSpec(nat.GEN_SEED_EXTENSION, randomness=0.0, two_sided=True,
                                 extension_iterations=1, max_extension_steps=ops.INT32_MAX,
                                 refill={"every": 64, "depth": 10, "serial": False, "lead": 3,
                                         "streams": 2, "record_stash": True})


RandomSeedGenerator = SeedExtensionGenerator
