from .env import Connector, DenseRewardFn  # noqa: F401
from .wrappers import MultiToSingleWrapper, VmapAutoResetWrapper  # noqa: F401
