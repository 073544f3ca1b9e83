"""Operator boundary: registers the hot-path kernels as `torch.ops.teal.*` custom ops.

Mirrors the role of the reference's kernels/compile_wrapper.py:124-208 (BaseKernel: derive a
torch.library schema from `forward`'s annotations, define + abstract impl + device impl under
namespace "teal", hand back `torch.ops.teal.<name>`), re-implemented on the current
torch.library API (`register_fake`; `impl_abstract` is deprecated in torch 2.10) and without
msgspec.  The resulting schemas are the ones the reference produces:

    teal::sparse_gemv(Tensor hidden_states, Tensor weights, float threshold, int sparsity_bin) -> Tensor
    teal::sparse_qkv_gemv(Tensor x, Tensor weight, float threshold_q, float threshold_k,
                          float threshold_v, int sparsity_bin, int kv_size) -> Tensor
"""
from __future__ import annotations

import inspect
import typing
from typing import Any, Callable

import torch

__all__ = ["BaseKernel", "NAMESPACE"]

NAMESPACE = "teal"

_SCALARS = {float: "float", int: "int", bool: "bool", str: "str", torch.dtype: "ScalarType",
            torch.device: "Device"}


def _schema_type(ann: Any) -> str:
    """python annotation -> torch.library schema type token."""
    if ann is inspect.Parameter.empty:
        raise TypeError("every parameter of `forward` needs a type annotation to derive the op schema")
    origin = typing.get_origin(ann)
    if origin is typing.Annotated:  # Annotated[int, "SymInt"] style overrides
        return str(typing.get_args(ann)[1])
    if origin is typing.Union:
        args = [a for a in typing.get_args(ann) if a is not type(None)]
        if len(args) != 1 or len(typing.get_args(ann)) != 2:
            raise TypeError(f"only Optional[T] unions are supported in op schemas, got {ann}")
        return _schema_type(args[0]) + "?"
    if origin in (list, typing.List):
        return _schema_type(typing.get_args(ann)[0]) + "[]"
    if ann is torch.Tensor or (isinstance(ann, type) and issubclass(ann, torch.Tensor)):
        return "Tensor"
    if ann in _SCALARS:
        return _SCALARS[ann]
    raise TypeError(f"cannot express annotation {ann!r} in a torch.library schema")


def _schema_return(ann: Any) -> str:
    if typing.get_origin(ann) in (tuple, typing.Tuple):
        return "(" + ", ".join(_schema_type(a) for a in typing.get_args(ann)) + ")"
    return _schema_type(ann)


class BaseKernel:
    """Wraps one kernel for registration with torch.library.

    Subclasses provide `forward` (device implementation) and `meta` (fake-tensor shape function);
    `initialize(name, target)` builds an instance, `operator(compiled=True)` registers it (once
    per op name) and returns `torch.ops.teal.<name>`; `operator(False)` returns `forward` itself.
    """

    def __init__(self, name: str, target: str, schema: str):
        self.name = name      # op name inside the namespace
        self.target = target  # dispatch key / device type, e.g. "cuda" (also on ROCm)
        self.schema = schema  # "(Tensor a, float b) -> Tensor"

    @classmethod
    def initialize(cls, name: str, target: str, **_unused) -> "BaseKernel":
        return cls(name, target, cls.schematize())

    @classmethod
    def schematize(cls) -> str:
        hints = typing.get_type_hints(cls.forward, include_extras=True)
        sig = inspect.signature(cls.forward)
        args = [f"{_schema_type(hints.get(n, p.annotation))} {n}" for n, p in sig.parameters.items() if n != "self"]
        return f"({', '.join(args)}) -> {_schema_return(hints.get('return', sig.return_annotation))}"

    @property
    def qualname(self) -> str:
        return f"{NAMESPACE}::{self.name}"

    @property
    def is_registered(self) -> bool:
        return hasattr(getattr(torch.ops, NAMESPACE), self.name)

    def operator(self, compiled: bool = False) -> Callable:
        if not compiled:
            return self.forward
        self.register()
        return getattr(getattr(torch.ops, NAMESPACE), self.name)

    def meta(self, *args, **kwargs) -> Any:
        raise NotImplementedError(f"{type(self).__name__}.meta (fake-tensor impl) is required for registration")

    def forward(self, *args, **kwargs) -> Any:
        raise NotImplementedError(f"{type(self).__name__}.forward (device impl) is required for registration")

    def register(self) -> None:
        if self.is_registered:
            return
        torch.library.define(self.qualname, self.schema)
        torch.library.register_fake(self.qualname)(self.meta)
        torch.library.impl(self.qualname, self.target)(self.forward)
