"""SeedExtensionBoard -- drop-in for the reference class of the same name
(routing_board_generation/board_generation_methods/jax_implementation/board_generation/
seed_extension.py:36-304), backed by `seed_extension_kernel`.

Every method takes a key of shape (2,) or a batch (B, 2) (the latter replaces jax.vmap)."""
from typing import Tuple

import torch

from .. import ops


class SeedExtensionBoard:
    def __init__(self, rows: int, cols: int, num_agents: int = 0):
        if rows != cols:
            raise ValueError("SeedExtensionBoard supports square grids only")
        if num_agents > (rows // 2) * (cols // 2):
            # jax.random.choice(replace=False) on the (rows//2)*(cols//2) seed cells (:113-115)
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        self._rows = rows
        self._cols = cols
        self.grid_size = max(rows, cols)
        self._num_agents = num_agents
        self._wires_on_board = min(num_agents, rows * cols // 3)  # :60-63

    def _run(self, key: torch.Tensor, randomness, two_sided, extension_iterations,
             extension_steps, want_solved):
        batched = key.dim() == 2
        keys = (key if batched else key[None]).contiguous()
        out = ops.seed_extension_generate(
            keys, self._rows, self._num_agents, split_first=False, randomness=randomness,
            two_sided=two_sided, extension_iterations=extension_iterations,
            extension_steps=extension_steps, want_solved=want_solved, want_train=not want_solved)
        return out, batched

    def return_seeded_board(self, key: torch.Tensor) -> torch.Tensor:
        """:93-147 -- N adjacent (target, head) pairs, nothing extended.  `key` is the seed key
        itself (return_solved_board hands this method split(key)[1])."""
        out, batched = self._run(key, 0.0, True, -1, 0, want_solved=True)
        return out["solved_grid"] if batched else out["solved_grid"][0]

    def return_solved_board(self, key: torch.Tensor, randomness: float = 0.0,
                            two_sided: bool = True, extension_iterations: int = 1,
                            extension_steps=1e23) -> torch.Tensor:
        """:149-227 -- the board with all wires present."""
        out, batched = self._run(key, randomness, two_sided, extension_iterations,
                                 extension_steps, want_solved=True)
        return out["solved_grid"] if batched else out["solved_grid"][0]

    def return_training_board(self, key: torch.Tensor, randomness: float = 0.0,
                              two_sided: bool = True, extension_iterations: int = 1,
                              extension_steps=1e23) -> torch.Tensor:
        """:230-254 -- heads and targets only."""
        out, batched = self._run(key, randomness, two_sided, extension_iterations,
                                 extension_steps, want_solved=False)
        return out["train_grid"] if batched else out["train_grid"][0]

    def generate_starts_ends(self, key: torch.Tensor, randomness: float = 0.0,
                             two_sided: bool = True, extension_iterations: int = 1,
                             extension_steps=1e23
                             ) -> Tuple[Tuple[torch.Tensor, torch.Tensor],
                                        Tuple[torch.Tensor, torch.Tensor]]:
        """:257-304 -- ((start rows, start cols), (end rows, end cols)), each (N,) or (B, N)."""
        out, batched = self._run(key, randomness, two_sided, extension_iterations,
                                 extension_steps, want_solved=False)
        s, e = out["start"], out["target"]
        if not batched:
            s, e = s[0], e[0]
        return (s[..., 0], s[..., 1]), (e[..., 0], e[..., 1])
