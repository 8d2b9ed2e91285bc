"""SequentialRandomWalkBoard -- drop-in for the reference class of the same name
(routing_board_generation/board_generation_methods/jax_implementation/board_generation/
sequential_random_walk.py:23-431), backed by `seq_rw_generate_kernel`."""
from typing import Tuple

import torch

from .. import ops


class SequentialRandomWalkBoard:
    def __init__(self, rows: int, cols: int, num_agents: int = 3):
        if rows != cols:
            raise ValueError("SequentialRandomWalkBoard supports square grids only")
        self._rows = rows
        self._cols = cols
        self._num_agents = num_agents

    def _run(self, key: torch.Tensor):
        batched = key.dim() == 2
        keys = (key if batched else key[None]).contiguous()
        out = ops.seq_rw_generate(keys, self._rows, self._num_agents, split_first=False,
                                  want_solved=True, want_train=False)
        return out, batched

    def generate(self, key: torch.Tensor) -> torch.Tensor:
        """Board with all wires still present (all zero when generation failed).  The reference
        returns float32 (:351,:374); the values are the same small integers, here int32."""
        out, batched = self._run(key)
        return out["solved_grid"] if batched else out["solved_grid"][0]

    def generate_starts_ends(self, key: torch.Tensor) -> Tuple[Tuple[torch.Tensor, torch.Tensor],
                                                               Tuple[torch.Tensor, torch.Tensor]]:
        """((start rows, start cols), (end rows, end cols)), each of shape (N,) (or (B, N))."""
        out, batched = self._run(key)
        s, e = out["start"], out["target"]
        if not batched:
            s, e = s[0], e[0]
        return (s[..., 0], s[..., 1]), (e[..., 0], e[..., 1])
