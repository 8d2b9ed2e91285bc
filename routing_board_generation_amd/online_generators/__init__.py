from .uniform_generator import Generator, UniformRandomGenerator  # noqa: F401
from .parallel_random_walk_generator import (ParallelRandomWalkGenerator,  # noqa: F401
                                             ParallelRandomWalkPhiloxGenerator)
from .sequential_random_walk_generator import SequentialRandomWalkGenerator  # noqa: F401
from .random_seed_generator import RandomSeedGenerator, SeedExtensionGenerator  # noqa: F401
