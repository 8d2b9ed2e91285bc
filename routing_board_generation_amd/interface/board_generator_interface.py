"""Name -> generator-class registry; mirrors routing_board_generation/interface/
board_generator_interface.py:32-71 for the boards this package implements.

`offline_parallel_rw` (the hot path) and `offline_seed_extension` (8f) map
to native classes; the other names of the reference enum -- its unseeded NumPy generators -- are
kept so that `BoardName(value)` parses every string the reference accepts, but asking for their
class raises NotImplementedError instead of importing a NumPy zoo.
"""
from enum import Enum

from ..board_generation.parallel_random_walk import ParallelRandomWalkBoard
from ..board_generation.seed_extension import SeedExtensionBoard


class BoardName(str, Enum):
    RANDOM_WALK = "random_walk"
    BFS_BASE = "bfs_base"
    BFS_MIN_BENDS = "bfs_min_bend"
    BFS_FIFO = "bfs_fifo"
    BFS_SHORTEST = "bfs_short"
    BFS_LONGEST = "bfs_long"
    LSYSTEMS = "lsystems_standard"
    WFC = "wfc"
    NUMBERLINK = "numberlink"

    JAX_PARALLEL_RW = "offline_parallel_rw"
    JAX_SEED_EXTENSION = "offline_seed_extension"


class BoardGenerator:
    """Maps BoardName to the class of the generator."""

    board_generator_dict = {
        BoardName.JAX_PARALLEL_RW: ParallelRandomWalkBoard,
        BoardName.JAX_SEED_EXTENSION: SeedExtensionBoard,  # reference :64
    }

    @classmethod
    def get_board_generator(cls, board_enum: BoardName):
        try:
            return cls.board_generator_dict[BoardName(board_enum)]
        except KeyError:
            raise NotImplementedError(
                f"{board_enum!r} is outside the MI355X hot path of this package "
                "(implemented natively: 'offline_parallel_rw', 'offline_seed_extension')") from None


def make_online_generator(generation_type: str, grid_size: int, num_agents: int, **kwargs):
    """The generator dispatch of setup_train._make_raw_env
    (routing_board_generation/rl_training/setup_train.py:112-152) for the generation types this
    package implements; kwargs (randomness, two_sided, extension_iterations, extension_steps,
    number_of_boards -- cfg.env.seed_extension.*, :121-130) go to the offline dataset generator."""
    from ..offline_generation import BoardDatasetGeneratorJAX
    from ..online_generators import (ParallelRandomWalkGenerator, SeedExtensionGenerator,
                                     SequentialRandomWalkGenerator, UniformRandomGenerator)

    if generation_type in ("offline_seed_extension", "offline_parallel_rw"):
        return BoardDatasetGeneratorJAX(grid_size=grid_size, num_agents=num_agents,
                                        board_name=generation_type, **kwargs)
    if generation_type == "online_uniform":
        return UniformRandomGenerator(grid_size=grid_size, num_agents=num_agents)
    if generation_type == "online_seq_rw":
        return SequentialRandomWalkGenerator(grid_size=grid_size, num_agents=num_agents)
    if generation_type == "online_random_seed":
        return SeedExtensionGenerator(grid_size=grid_size, num_agents=num_agents)
    if generation_type == "online_parallel_rw":
        return ParallelRandomWalkGenerator(grid_size=grid_size, num_agents=num_agents)
    if generation_type in ("offline", "online_lsystems"):
        raise NotImplementedError(f"generation_type {generation_type!r} (NumPy generators, "
                                  "unseeded) is outside the hot path")
    raise ValueError("Your connector.yml parameters do not exist")  # setup_train.py:154
