"""ParallelRandomWalkBoard -- drop-in for the reference class of the same name
(routing_board_generation/board_generation_methods/jax_implementation/board_generation/
parallel_random_walk.py:49-90), backed by the gfx950 kernel `prw_generate_kernel`.
"""
from typing import Tuple

import torch

from .. import ops


class ParallelRandomWalkBoard:
    def __init__(self, rows: int, cols: int, num_agents: int):
        if rows != cols:
            # the reference's _adjacent_cells mixes `rows` and `cols` (parallel_random_walk.py
            # :276-285) and every caller passes a square grid
            raise ValueError("ParallelRandomWalkBoard supports square grids only")
        self.rows = rows
        self.cols = cols
        self.num_agents = num_agents

    def generate_board(self, key: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """key (2,) -> heads (2, N), targets (2, N), solved_grid (m, m);
        key (B, 2) -> the same with a leading B (what `jax.vmap(generate_board)` returns)."""
        batched = key.dim() == 2
        keys = key if batched else key[None]
        out = ops.prw_generate(keys.contiguous(), self.rows, self.num_agents, split_first=False,
                               want_solved=True, want_train=False)
        heads = out["start"].transpose(-1, -2)
        targets = out["target"].transpose(-1, -2)
        grid = out["solved_grid"]
        if batched:
            return heads, targets, grid
        return heads[0], targets[0], grid[0]

    def generate_board_with_stats(self, keys: torch.Tensor):
        """Batched variant that also returns the while_loop trip count per board."""
        out = ops.prw_generate(keys.contiguous(), self.rows, self.num_agents, split_first=False,
                               want_solved=True, want_train=False, want_iters=True)
        return (out["start"].transpose(-1, -2), out["target"].transpose(-1, -2),
                out["solved_grid"], out["iters"])
