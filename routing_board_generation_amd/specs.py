"""Shape / dtype / bounds descriptions of what the environment consumes and produces, after
[3P] jumanji.specs (jumanji==0.2.2): Array, BoundedArray, DiscreteArray, MultiDiscreteArray and
the structured Spec.  Network builders read them (`env.action_spec().num_values`,
`env.observation_spec().grid.shape`); nothing here touches the GPU.  Values are checked as
torch tensors or anything `torch.as_tensor` accepts.
"""
from typing import Any, Callable, Dict, Generic, Sequence, TypeVar

import torch

T = TypeVar("T")


class Array:
    """A tensor of a fixed shape and dtype."""

    def __init__(self, shape: Sequence[int], dtype: torch.dtype, name: str = "") -> None:
        self.shape = tuple(int(s) for s in shape)
        self.dtype = dtype
        self.name = name

    def __repr__(self) -> str:
        return f"{type(self).__name__}(shape={self.shape}, dtype={self.dtype}, name={self.name!r})"

    def _fail(self, msg: str) -> None:
        raise ValueError(f"{self.name or type(self).__name__}: {msg}")

    def validate(self, value: Any) -> torch.Tensor:
        """Returns `value` as a tensor if it conforms to the spec, raises ValueError otherwise.
        A leading batch axis is not part of a spec (as in jumanji): validate one element."""
        v = torch.as_tensor(value)
        if tuple(v.shape) != self.shape:
            self._fail(f"expected shape {self.shape}, got {tuple(v.shape)}")
        if v.dtype != self.dtype:
            self._fail(f"expected dtype {self.dtype}, got {v.dtype}")
        return v

    def generate_value(self) -> torch.Tensor:
        return torch.zeros(self.shape, dtype=self.dtype)

    def replace(self, **kwargs: Any) -> "Array":
        args = dict(self._args(), **kwargs)
        return type(self)(**args)

    def _args(self) -> Dict[str, Any]:
        return {"shape": self.shape, "dtype": self.dtype, "name": self.name}


class BoundedArray(Array):
    """An Array with inclusive element-wise bounds."""

    def __init__(self, shape: Sequence[int], dtype: torch.dtype, minimum, maximum,
                 name: str = "") -> None:
        super().__init__(shape, dtype, name)
        self.minimum = torch.as_tensor(minimum)
        self.maximum = torch.as_tensor(maximum)
        if bool((self.minimum.to(torch.float64) > self.maximum.to(torch.float64)).any()):
            raise ValueError(f"{name}: minimum exceeds maximum")

    def __repr__(self) -> str:
        return (f"{type(self).__name__}(shape={self.shape}, dtype={self.dtype}, "
                f"minimum={self.minimum.tolist()}, maximum={self.maximum.tolist()}, "
                f"name={self.name!r})")

    def validate(self, value: Any) -> torch.Tensor:
        v = super().validate(value)
        x = v.to(torch.float64)
        if bool((x < self.minimum.to(torch.float64).to(v.device)).any()) or bool(
                (x > self.maximum.to(torch.float64).to(v.device)).any()):
            self._fail("value out of bounds")
        return v

    def generate_value(self) -> torch.Tensor:
        return (torch.ones(self.shape, dtype=torch.float64) * self.minimum).to(self.dtype)

    def _args(self) -> Dict[str, Any]:
        return dict(super()._args(), minimum=self.minimum, maximum=self.maximum)


class DiscreteArray(BoundedArray):
    """A scalar in {0, ..., num_values - 1}."""

    def __init__(self, num_values: int, dtype: torch.dtype = torch.int32, name: str = "") -> None:
        if num_values <= 0:
            raise ValueError("num_values must be positive")
        super().__init__((), dtype, 0, num_values - 1, name)
        self.num_values = int(num_values)

    def _args(self) -> Dict[str, Any]:
        return {"num_values": self.num_values, "dtype": self.dtype, "name": self.name}


class MultiDiscreteArray(BoundedArray):
    """A tensor of independent discrete values; element i lies in {0, ..., num_values[i] - 1}."""

    def __init__(self, num_values, dtype: torch.dtype = torch.int32, name: str = "") -> None:
        nv = torch.as_tensor(num_values)
        if bool((nv <= 0).any()):
            raise ValueError("num_values must be positive")
        super().__init__(tuple(nv.shape), dtype, torch.zeros_like(nv), nv - 1, name)
        self.num_values = nv

    def _args(self) -> Dict[str, Any]:
        return {"num_values": self.num_values, "dtype": self.dtype, "name": self.name}


class Spec(Generic[T]):
    """A named structure of specs that builds a `constructor` object (e.g. Observation)."""

    def __init__(self, constructor: Callable[..., T], name: str = "", **specs: Any) -> None:
        self._constructor = constructor
        self.name = name
        self._specs = dict(specs)
        for k, v in specs.items():
            setattr(self, k, v)

    def __repr__(self) -> str:
        inner = ", ".join(f"{k}={v!r}" for k, v in self._specs.items())
        return f"{self.name or 'Spec'}({inner})"

    def validate(self, value: T) -> T:
        return self._constructor(**{k: s.validate(getattr(value, k)) for k, s in self._specs.items()})

    def generate_value(self) -> T:
        return self._constructor(**{k: s.generate_value() for k, s in self._specs.items()})

    def replace(self, **kwargs: Any) -> "Spec[T]":
        return Spec(self._constructor, self.name, **dict(self._specs, **kwargs))
