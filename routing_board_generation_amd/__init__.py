"""routing_board_generation_amd -- MI355X-native ParallelRandomWalk board generation and Connector
reset/step, behind the reference's Generator / Environment API.

Importing this package never touches the GPU and never falls back to a CPU path: the first call
that needs compute loads lib/librbg_prw.so and raises if it is missing.
"""
from . import _native, specs  # noqa: F401
from .board_generation.parallel_random_walk import ParallelRandomWalkBoard  # noqa: F401
from .connector import (Connector, DenseRewardFn, MultiToSingleWrapper,  # noqa: F401
                        VmapAutoResetWrapper)
from .interface.board_generator_interface import BoardGenerator, BoardName  # noqa: F401
from .offline_generation import BoardDatasetGenerator, BoardDatasetGeneratorJAX  # noqa: F401
from .board_generation.seed_extension import SeedExtensionBoard  # noqa: F401
from .board_generation.sequential_random_walk import SequentialRandomWalkBoard  # noqa: F401
from .online_generators import (Generator, ParallelRandomWalkGenerator,  # noqa: F401
                                ParallelRandomWalkPhiloxGenerator,
                                RandomSeedGenerator, SeedExtensionGenerator,
                                SequentialRandomWalkGenerator, UniformRandomGenerator)
from .random import PRNGKey, split  # noqa: F401
from .types import Agent, Observation, State, StepType, TimeStep  # noqa: F401

__version__ = "0.1.0"
