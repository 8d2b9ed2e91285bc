"""BoardDatasetGeneratorJAX -- mirrors the reference class of the same name
(routing_board_generation/rl_training/offline_generation/dataset_generator_jax.py:20-141): a
dataset of `number_of_boards` boards is generated once from split(PRNGKey(0), number_of_boards)
by SeedExtensionBoard.generate_starts_ends ("offline_seed_extension", the reference's default),
ParallelRandomWalkBoard.generate_board ("offline_parallel_rw") or UniformRandomGenerator
("offline_uniform"); `__call__(key)` samples one board of it.

The reference builds the dataset with a Python loop of one jitted call per board (:76-105); here it
is one batched launch.  Constructor signature and attribute layout follow the reference
(`heads` / `targets` of shape (n_boards, 2, N), :57-63); the seed-extension keywords are accepted
and ignored for the other board names, as the reference does (setup_train.py:121-130 passes them
for "offline_parallel_rw" too).  Sampling follows [3P] jax.random.randint; no reference golden
pins it ("parity unpinned").
"""
import torch

from .. import ops
from ..online_generators.uniform_generator import Generator, state_from_bufs
from ..random import PRNGKey, split
from ..types import State


class BoardDatasetGeneratorJAX(Generator):
    def __init__(self, grid_size: int, num_agents: int, randomness: float = 1,
                 two_sided: bool = False, extension_iterations: int = 1,
                 extension_steps=1e23, board_name: str = "offline_seed_extension",
                 number_of_boards: int = 10000, generate_solved_boards: bool = False,
                 device=None) -> None:
        super().__init__(grid_size, num_agents)
        if board_name not in ("offline_seed_extension", "offline_uniform", "offline_parallel_rw"):
            # the reference's else-branch would run ParallelRandomWalk for ANY other string; a
            # typo silently changing the generator is not worth mirroring
            raise ValueError(f"unknown board_name {board_name!r}")
        self.board_name = board_name
        self.device = device
        if board_name != "offline_parallel_rw":  # (:36-52: kept as attributes for these two)
            self.randomness = randomness
            self.two_sided = two_sided
            self.extension_iterations = extension_iterations
            self.extension_steps = extension_steps
        # :57-63 -- the dataset is generate_n_boards(PRNGKey(0), number_of_boards, ...)
        self._heads, self._targets, solved = self._generate(
            split(PRNGKey(0, device=device), number_of_boards), generate_solved_boards)
        self.solved_boards = solved

    def _generate(self, keys: torch.Tensor, generate_solved_boards: bool):
        """One batched launch over `keys` [n, 2]: (heads [n, N, 2], targets [n, N, 2] agent-major,
        solved boards [n, m, m] or None)."""
        m, N = self.grid_size, self.num_agents
        if self.board_name == "offline_seed_extension":
            out = ops.seed_extension_generate(
                keys, m, N, split_first=False, randomness=self.randomness,
                two_sided=self.two_sided, extension_iterations=self.extension_iterations,
                extension_steps=self.extension_steps, want_solved=bool(generate_solved_boards),
                want_train=False)
            return out["start"], out["target"], out.get("solved_grid")
        if self.board_name == "offline_uniform":
            st = ops.uniform_generate(keys, m, N)
            return st["start"], st["target"], None
        out = ops.prw_generate(keys, m, N, split_first=False, want_solved=True, want_train=False)
        return out["start"], out["target"], out["solved_grid"]

    def generate_n_boards(self, key: torch.Tensor, n_boards: int = 10,
                          generate_solved_boards: bool = False):
        """dataset_generator_jax.py:65-110 -- (heads, targets, solved_boards) of the boards
        generated from split(key, n_boards): heads / targets of shape (n_boards, 2, N) (row and
        column arrays per board, what `jnp.array` makes of the reference's lists), solved_boards
        (n_boards, m, m) -- or [] where the reference returns an empty list (seed extension
        without generate_solved_boards; the uniform generator has no solved board).  One batched
        launch instead of the reference's loop of one jitted call per board.  Where the
        reference's seed-extension branch returns None for heads and targets when it generates
        solved boards (:84-95), this returns them as well: the same launch produces them."""
        key = torch.as_tensor(key)
        if key.shape != (2,):
            raise ValueError("generate_n_boards expects one PRNG key (shape (2,))")
        key = key.to(self._heads.device)  # (the dataset's device)
        n = int(n_boards)
        keys = split(key, n)
        heads, targets, solved = self._generate(keys, generate_solved_boards)
        return (heads.transpose(-1, -2), targets.transpose(-1, -2),
                solved if solved is not None else [])

    @property
    def heads(self) -> torch.Tensor:
        """(n_boards, 2, N): row and column arrays per board, as the reference holds them."""
        return self._heads.transpose(-1, -2)

    @property
    def targets(self) -> torch.Tensor:
        return self._targets.transpose(-1, -2)

    def __len__(self) -> int:
        return self._heads.shape[0]

    def __call__(self, key: torch.Tensor) -> State:
        batched = key.dim() == 2
        keys = (key if batched else key[None]).contiguous()
        st, _ = ops.dataset_sample(keys, self._heads, self._targets, self.grid_size)
        return state_from_bufs(st, batched)

    def native_spec(self):
        from .. import _native as nat

        # sampling a stored board is cheaper than a stash row copy: no stash
        return ops.GeneratorSpec(nat.GEN_DATASET, heads=self._heads, targets=self._targets,
                                 refill={"every": 0, "rollout_stash": False})

    def print_board(self, which_board: int) -> torch.Tensor:
        """:143-155 -- the training grid of board `which_board`."""
        grid = torch.zeros((self.grid_size, self.grid_size), dtype=torch.int32,
                           device=self._heads.device)
        ids = torch.arange(self.num_agents, dtype=torch.int32, device=grid.device)
        h, t = self._heads[which_board].long(), self._targets[which_board].long()
        grid[h[:, 0], h[:, 1]] = 2 + 3 * ids
        grid[t[:, 0], t[:, 1]] = 3 + 3 * ids
        return grid


# round-1 name of this class (its heads/targets were agent-major then)
BoardDatasetGenerator = BoardDatasetGeneratorJAX
