from .uniform_generator import Generator, UniformRandomGenerator  # noqa: F401
from .parallel_random_walk_generator import ParallelRandomWalkGenerator  # noqa: F401
