"""BoardDatasetGenerator -- mirrors BoardDatasetGeneratorJAX of the reference
(routing_board_generation/rl_training/offline_generation/dataset_generator_jax.py:20-141) for
board_name "offline_parallel_rw" (and "offline_uniform"): a dataset of `number_of_boards` boards
is generated once from split(PRNGKey(0), number_of_boards); `__call__(key)` samples one.

The reference builds the dataset with a Python loop of one jitted call per board (:76-105); here it
is one batched launch.  Sampling follows [3P] jax.random.randint; no reference golden pins it
("parity unpinned").
"""
import torch

from .. import ops
from ..online_generators.uniform_generator import Generator, state_from_bufs
from ..random import PRNGKey, split
from ..types import State


class BoardDatasetGenerator(Generator):
    def __init__(self, grid_size: int, num_agents: int, board_name: str = "offline_parallel_rw",
                 number_of_boards: int = 10000, device=None) -> None:
        super().__init__(grid_size, num_agents)
        self.board_name = board_name
        keys = split(PRNGKey(0, device=device), number_of_boards)
        if board_name == "offline_parallel_rw":
            out = ops.prw_generate(keys, grid_size, num_agents, split_first=False,
                                   want_solved=True, want_train=False)
            self.heads, self.targets = out["start"], out["target"]
            self.solved_boards = out["solved_grid"]
        elif board_name == "offline_uniform":
            st = ops.uniform_generate(keys, grid_size, num_agents)
            self.heads, self.targets, self.solved_boards = st["start"], st["target"], None
        else:
            raise NotImplementedError(f"board_name {board_name!r} is outside the hot path")

    def __len__(self) -> int:
        return self.heads.shape[0]

    def __call__(self, key: torch.Tensor) -> State:
        batched = key.dim() == 2
        keys = (key if batched else key[None]).contiguous()
        st, _ = ops.dataset_sample(keys, self.heads, self.targets, self.grid_size)
        return state_from_bufs(st, batched)
