"""Flat-parameter layout and the `state` contract.

Mirrors the reference's ``src/utils.py:12-17`` (``flatten_nn_params``): the flat
parameter vector theta in R^D is the ravel of ``state.params`` minus the top-level
keys ``logvar`` and ``batch_stats``.  The reference uses ``jax.flatten_util.ravel_pytree``
which walks dicts in *sorted key order* and ravels every leaf in C order; this file
restates exactly that traversal on nested dicts of torch tensors and additionally
returns the layout table (path -> offset, shape) that the HIP engine needs.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch

_EXCLUDED_TOP_LEVEL = ("logvar", "batch_stats")  # reference src/utils.py:14-16


def _walk(tree, prefix=()):
    """Yield (path, leaf) in sorted-key order (what ravel_pytree does for dicts)."""
    if isinstance(tree, dict):
        for k in sorted(tree.keys()):
            yield from _walk(tree[k], prefix + (k,))
    elif isinstance(tree, (list, tuple)):
        for i, v in enumerate(tree):
            yield from _walk(v, prefix + (i,))
    else:
        yield prefix, tree


def nn_param_tree(params: Dict[str, Any]) -> Dict[str, Any]:
    return {k: v for k, v in params.items() if k not in _EXCLUDED_TOP_LEVEL}


def param_layout(params: Dict[str, Any]) -> List[Tuple[Tuple, int, Tuple[int, ...]]]:
    """[(path, offset, shape)] of the flat vector, in flat order."""
    out, off = [], 0
    for path, leaf in _walk(nn_param_tree(params)):
        leaf = torch.as_tensor(leaf)
        out.append((path, off, tuple(leaf.shape)))
        off += leaf.numel()
    return out


def _set_path(tree, path, value):
    for k in path[:-1]:
        tree = tree.setdefault(k, {})
    tree[path[-1]] = value


def flatten_nn_params(params: Dict[str, Any]):
    """``(flat, unravel_fn)`` — reference ``src/utils.py:12-17``.

    ``unravel_fn(flat)`` rebuilds the nested dict (without ``logvar`` /
    ``batch_stats``); it is differentiable (views of ``flat``), so it can sit
    inside ``torch.func`` transforms.
    """
    layout = param_layout(params)
    leaves = [torch.as_tensor(leaf) for _, leaf in _walk(nn_param_tree(params))]
    if leaves:
        flat = torch.cat([l.reshape(-1) for l in leaves])
    else:
        flat = torch.zeros(0)

    def unravel_fn(flatp):
        tree: Dict[str, Any] = {}
        for path, off, shape in layout:
            n = 1
            for s in shape:
                n *= s
            _set_path(tree, path, flatp[off:off + n].reshape(shape))
        return tree

    return flat, unravel_fn


def count_model_params(params) -> int:
    """Reference ``src/utils.py:84``: number of scalars in a pytree."""
    return sum(torch.as_tensor(l).numel() for _, l in _walk(params))


def is_pd(M: torch.Tensor) -> bool:
    """Reference ``src/utils.py:9``."""
    return bool(torch.all(torch.linalg.eigvals(M).real >= 1e-9))


def tree_map(fn: Callable, tree):
    if isinstance(tree, dict):
        return {k: tree_map(fn, v) for k, v in tree.items()}
    return fn(tree)


@dataclasses.dataclass
class TrainState:
    """The ``state`` object every factory takes (reference ``src/scalemodels.py:161-163``,
    ``tests/fixtures.py:64-69``): ``.params`` (nested dict; top-level ``logvar`` /
    ``batch_stats`` are not part of theta), ``.apply_fn(variables, x, **kw)``,
    ``.batch_stats``, ``.replace(...)``.

    ``net`` is the layer program (:class:`netspec.NetSpec`) the HIP engine compiles;
    product paths refuse a state without it (there is no CPU fallback).
    """
    params: Dict[str, Any]
    apply_fn: Callable
    batch_stats: Dict[str, Any] = dataclasses.field(default_factory=dict)
    net: Any = None
    alpha: Optional[float] = None

    def replace(self, **kw) -> "TrainState":
        return dataclasses.replace(self, **kw)

    def to(self, device=None, dtype=None) -> "TrainState":
        f = lambda t: torch.as_tensor(t).to(device=device, dtype=dtype)
        return self.replace(params=tree_map(f, self.params),
                            batch_stats=tree_map(f, self.batch_stats))


def save_array_checkpoint(array, ckpt_dir, name, step):
    """Reference ``src/utils.py:20`` (implemented in ``checkpoint.py``)."""
    from .checkpoint import save_array_checkpoint as f
    return f(array, ckpt_dir, name, step)


def load_array_checkpoint(ckpt_dir, name, step, device=None):
    """Reference ``src/utils.py:33``."""
    from .checkpoint import load_array_checkpoint as f
    return f(ckpt_dir, name, step, device=device)


def save_checkpoint(train_state, ckpt_dir, prefix, step):
    """Reference ``src/utils.py:46``."""
    from .checkpoint import save_checkpoint as f
    return f(train_state, ckpt_dir, prefix, step)


def load_checkpoint(ckpt_dir, prefix, target=None, **kw):
    """Reference ``src/utils.py:63``."""
    from .checkpoint import load_checkpoint as f
    return f(ckpt_dir, prefix, target=target, **kw)
