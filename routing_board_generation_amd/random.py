"""Key plumbing with jax.random's Threefry-2x32 semantics ([3P] jax==0.4.8, jax/_src/prng.py).

Keys are uint32 tensors of shape (2,) or (B, 2) on the HIP device.  `split` runs on the GPU
(rbg_split / rbg_split_batch); nothing here is a CPU restatement of the stream.
"""
from typing import Optional

import torch

from . import ops


def _default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device())


def PRNGKey(seed: int, device: Optional[torch.device] = None) -> torch.Tensor:
    """jax.random.PRNGKey(seed): key data (seed >> 32, seed & 0xffffffff)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    dev = device if device is not None else _default_device()
    return torch.tensor([seed >> 32, seed & 0xFFFFFFFF], dtype=torch.int64).to(torch.uint32).to(dev)


def split(key: torch.Tensor, num: int = 2) -> torch.Tensor:
    """jax.random.split(key, num) -> (num, 2); for batched keys (B, 2) -> (B, num, 2)."""
    if key.dim() == 1:
        k = key.to("cpu")  # two words: the parent key is passed by value through the C-ABI
        return ops.split(int(k[0]), int(k[1]), num, key.device)
    return ops.split_batch(key.contiguous(), num)
