"""Checkpoint / array formats of the reference ("next" row N4, ``src/utils.py:20-75``).

* inducing points: ``ckpt_dir/name_step.npy`` — ``save_array_checkpoint`` / ``load_array_checkpoint`` (``:20-43``).
* model state: the reference calls ``flax.training.checkpoints.save_checkpoint(ckpt_dir, target=train_state,
  step, prefix=prefix + "_")`` (``:46-60``), i.e. one file ``ckpt_dir/<prefix>_<step>`` holding
  ``flax.serialization.to_bytes(state)``: a msgpack map of the TrainState's state-dict
  (``{'step', 'params', 'opt_state', 'batch_stats', ...}``) in which every array is a msgpack ExtType 1 whose
  payload is ``msgpack.packb((shape, dtype.name, raw_bytes))`` (numpy scalars: ExtType 3; arrays above 2**30 bytes
  are split into ``{'__msgpack_chunked_array__': True, 'shape': ..., 'chunks': {...}}``).
  flax is not installed here and the reference's ``checkpoint/`` directory is absent, so the
  reader/writer below are restated from that documented wire format and pinned by a round-trip test only —
  "parity unpinned" against a real reference checkpoint.
The file carries data only (no code is executed on load: msgpack + numpy.frombuffer).
"""
from __future__ import annotations

import os
import re
from typing import Any, Dict, Optional

import msgpack
import numpy as np
import torch

from .utils import TrainState, tree_map

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
_MAX_CHUNK = 2 ** 30


# ------------------------------------------------------------------------------------------ arrays (.npy)
def save_array_checkpoint(array, ckpt_dir, name, step):
    """``src/utils.py:20-30``: ``ckpt_dir/name_step.npy``."""
    ckpt_dir = os.path.abspath(ckpt_dir)
    os.makedirs(ckpt_dir, exist_ok=True)
    filename = os.path.join(ckpt_dir, f"{name}_{step}.npy")
    np.save(filename, torch.as_tensor(array).detach().cpu().numpy())
    return filename


def load_array_checkpoint(ckpt_dir, name, step, device=None):
    """``src/utils.py:33-43`` (``FileNotFoundError`` when absent, ``:39-40``)."""
    filename = os.path.join(os.path.abspath(ckpt_dir), f"{name}_{step}.npy")
    if not os.path.exists(filename):
        raise FileNotFoundError(f"Checkpoint file {filename} not found")
    return torch.from_numpy(np.load(filename, allow_pickle=False)).to(device=device)


# ------------------------------------------------------------------------------------------ flax msgpack
def _ext_hook(code, data):
    if code == _EXT_NDARRAY:
        shape, dtype_name, buf = msgpack.unpackb(data, raw=True)
        if isinstance(dtype_name, bytes):
            dtype_name = dtype_name.decode()
        return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(tuple(shape)).copy()
    if code == _EXT_NPSCALAR:
        shape, dtype_name, buf = msgpack.unpackb(data, raw=True)
        if isinstance(dtype_name, bytes):
            dtype_name = dtype_name.decode()
        return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(())[()]
    if code == _EXT_COMPLEX:
        re_, im_ = msgpack.unpackb(data)
        return complex(re_, im_)
    return msgpack.ExtType(code, data)


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get("__msgpack_chunked_array__"):
            # flax writes tuples through _tuple_to_dict: {'0': d0, '1': d1, ...}; a plain list is accepted as well
            shp = tree["shape"]
            shape = tuple(int(shp[str(i)] if str(i) in shp else shp[i]) for i in range(len(shp))) if isinstance(shp, dict) \
                else tuple(int(v) for v in shp)
            chunks = tree["chunks"]
            parts = [chunks[str(i)] if str(i) in chunks else chunks[i] for i in range(len(chunks))]
            return np.concatenate([np.asarray(p).reshape(-1) for p in parts]).reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def msgpack_restore(encoded: bytes) -> Dict[str, Any]:
    """``flax.serialization.msgpack_restore``: bytes -> nested dict of numpy arrays / scalars."""
    tree = msgpack.unpackb(encoded, ext_hook=_ext_hook, raw=False, strict_map_key=False)
    return _unchunk(tree)


def _pack_ndarray(a: np.ndarray):
    a = np.ascontiguousarray(a)
    if a.nbytes > _MAX_CHUNK:
        flat = a.reshape(-1)
        per = max(1, _MAX_CHUNK // a.dtype.itemsize)
        chunks = {str(i): _pack_ndarray(flat[s:s + per]) for i, s in enumerate(range(0, flat.size, per))}
        return {"__msgpack_chunked_array__": True, "shape": {str(i): int(d) for i, d in enumerate(a.shape)}, "chunks": chunks}
    return msgpack.ExtType(_EXT_NDARRAY, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes()), use_bin_type=True))


def _to_packable(tree):
    if isinstance(tree, dict):
        return {str(k): _to_packable(v) for k, v in tree.items()}
    if torch.is_tensor(tree):
        return _pack_ndarray(tree.detach().cpu().numpy())
    if isinstance(tree, np.ndarray):
        return _pack_ndarray(tree)
    if isinstance(tree, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, msgpack.packb(([], tree.dtype.name, tree.tobytes()), use_bin_type=True))
    return tree


def msgpack_serialize(tree: Dict[str, Any]) -> bytes:
    """``flax.serialization.msgpack_serialize`` of a nested dict of arrays."""
    return msgpack.packb(_to_packable(tree), use_bin_type=True)


def _latest(ckpt_dir: str, prefix: str) -> Optional[str]:
    best, best_step = None, -1
    for f in os.listdir(ckpt_dir):
        m = re.fullmatch(re.escape(prefix) + r"(\d+)", f)
        if m and int(m.group(1)) > best_step:
            best, best_step = os.path.join(ckpt_dir, f), int(m.group(1))
    return best


def save_checkpoint(train_state: TrainState, ckpt_dir, prefix, step):
    """``src/utils.py:46-60``: ``ckpt_dir/<prefix>_<step>`` in the Flax msgpack format (state-dict keys ``step``,
    ``params``, ``batch_stats``; optimiser state is not part of this build's TrainState)."""
    ckpt_dir = os.path.abspath(ckpt_dir)
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f"{prefix}_{step}")
    with open(path, "wb") as f:
        f.write(msgpack_serialize({"step": int(step), "params": train_state.params, "batch_stats": train_state.batch_stats}))
    return path


def load_checkpoint(ckpt_dir, prefix, target: Optional[TrainState] = None, dtype=torch.float32, device=None):
    """``src/utils.py:63-75``: restore the latest ``<prefix>_<step>``.  With ``target`` (a TrainState built for the
    right NetSpec) the restored ``params`` / ``batch_stats`` replace the target's and are shape-checked against it;
    without, the raw nested dict is returned.  The reference has two conventions for ``state.params`` (SURVEY
    §4.1-3): a checkpoint whose ``params`` lacks the top-level ``'params'`` key is wrapped into it."""
    ckpt_dir = os.path.abspath(ckpt_dir)
    path = _latest(ckpt_dir, prefix + "_")
    if path is None:
        raise FileNotFoundError(f"no checkpoint with prefix {prefix}_ in {ckpt_dir}")
    with open(path, "rb") as f:
        raw = msgpack_restore(f.read())
    if target is None:
        return raw
    params = raw.get("params", {})
    if "params" not in params:
        logvar = params.pop("logvar", None) if isinstance(params, dict) else None
        params = {"params": params}
        if logvar is not None:
            params["logvar"] = logvar
    to_t = lambda a: torch.as_tensor(np.asarray(a)).to(dtype=dtype, device=device)
    params = tree_map(to_t, params)
    stats = tree_map(to_t, raw.get("batch_stats", {}) or {})
    _check_like(params, target.params, "params")
    if target.batch_stats:
        _check_like(stats, target.batch_stats, "batch_stats")
    # Dense kernels: Flax stores (in, out) like this build; conv kernels HWIO like this build -> no transposes
    new = target.replace(params=params, batch_stats=stats if stats else target.batch_stats)
    if target.net is not None:
        new = new.replace(apply_fn=target.net.make_apply_fn(getattr(target.net, "model_type", "classifier"), new.batch_stats))
    return new


def _check_like(got, want, where):
    if isinstance(want, dict):
        if not isinstance(got, dict) or set(got) != set(want):
            raise ValueError(f"checkpoint {where}: keys {sorted(got) if isinstance(got, dict) else type(got)} != {sorted(want)}")
        for k in want:
            _check_like(got[k], want[k], f"{where}/{k}")
    else:
        gs, ws = tuple(torch.as_tensor(got).shape), tuple(torch.as_tensor(want).shape)
        # exact shapes only (a kernel stored HWOI instead of HWIO has the same number of elements); the one tolerated
        # difference is a scalar written as () against (1,)
        if gs != ws and not ({gs, ws} <= {(), (1,)}):
            raise ValueError(f"checkpoint {where}: shape {gs} != {ws}")
