"""Layer program ("NetSpec") for the networks the reference differentiates.

The reference hands an opaque Flax ``apply_fn`` to ``jax.jvp`` / ``jax.vjp``
(``src/ggn.py:59,75,139,142``).  A hand-written HIP engine cannot differentiate an
opaque Python function, so the build's workload definition is an explicit layer
program: a short list of *units* over NHWC activation tensors

    conv unit :  dst = act( BN( conv(src, kernel) + bias ) + residual )
                 (a Dense layer is the 1x1 conv of a 1x1 image; BN is eval-mode)
    meanpool  :  dst = mean over pixels of src            (``jnp.mean(x,(1,2))``)

which covers ``SimpleRegressor`` / ``SimpleClassifier`` (``src/toymodels.py:4-37``),
``LargeClassifier`` and ``ResNet1M`` (``src/scalemodels.py:52-157``).  The same
NetSpec drives (a) the torch functional forward used by the CPU oracle and the CPU
baseline and (b) the op-tape compiler of the HIP engine (``engine.py``).

Parameter names follow Flax auto-naming (``Dense_0/kernel`` ...), Dense kernels are
(in, out), conv kernels HWIO, activations NHWC — so the flat-theta order equals what
``ravel_pytree`` gives for the reference's models (``src/utils.py:12-17``).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Path = Tuple[str, ...]

ACT_NONE, ACT_RELU, ACT_TANH, ACT_GELU = 0, 1, 2, 3
ACT_IDS = {"none": ACT_NONE, "relu": ACT_RELU, "tanh": ACT_TANH, "gelu": ACT_GELU}

_GELU_C = math.sqrt(2.0 / math.pi)


def act_fn(name: str, y: torch.Tensor) -> torch.Tensor:
    if name == "none":
        return y
    if name == "relu":
        return torch.relu(y)
    if name == "tanh":
        return torch.tanh(y)
    if name == "gelu":  # flax.linen.gelu default: approximate=True (tanh form)
        return 0.5 * y * (1.0 + torch.tanh(_GELU_C * (y + 0.044715 * y ** 3)))
    raise ValueError(name)


def _same_pad(n_in: int, k: int, stride: int) -> Tuple[int, int]:
    """XLA/Flax 'SAME': (n_out, pad_lo); the odd extra pad goes to the high side."""
    n_out = -(-n_in // stride)
    total = max((n_out - 1) * stride + k - n_in, 0)
    return n_out, total // 2


@dataclasses.dataclass
class Unit:
    kind: str                       # 'conv' | 'meanpool' | 'maxpool' | 'avgpool' | 'view'
    src: int
    dst: int
    kh: int = 1
    kw: int = 1
    stride: int = 1
    pad_h: int = 0                  # low-side padding
    pad_w: int = 0
    cin: int = 0
    cout: int = 0
    kernel: Optional[Path] = None   # path below params root
    bias: Optional[Path] = None
    bn_scale: Optional[Path] = None
    bn_bias: Optional[Path] = None
    bn_mean: Optional[Path] = None  # path below batch_stats root
    bn_var: Optional[Path] = None
    bn_eps: float = 1e-5
    res: Optional[int] = None       # tensor id added before the activation
    act: str = "none"
    flat_kernel: bool = False       # Dense on a flattened (H,W,C) tensor: kernel stored as (H*W*C, out)


def _get(tree, path):
    for k in path:
        tree = tree[k]
    return tree


class NetSpec:
    """A straight-line program over NHWC tensors; tensor 0 is the input."""

    def __init__(self, input_shape: Sequence[int], param_root: Path = ("params",), tile_channels: int = 1):
        ishape = tuple(int(s) for s in input_shape)
        self.input_shape_raw = ishape
        if len(ishape) == 1:
            ishape = (1, 1, ishape[0])
        elif len(ishape) == 2:
            ishape = (ishape[0], ishape[1], 1)
        # ``jnp.tile(x, (1,1,1,3))`` of a grayscale input (reference src/scalemodels.py:127-128): tensor 0 is the
        # tiled image; ``prepare_input`` does the replication (the input carries no tangent, so this is host-side)
        self.tile_channels = int(tile_channels)
        if self.tile_channels > 1:
            ishape = (ishape[0], ishape[1], ishape[2] * self.tile_channels)
        self.tensors: List[Tuple[int, int, int]] = [ishape]
        self.units: List[Unit] = []
        self.param_root = tuple(param_root)
        self.out: int = 0

    # ---------------------------------------------------------------- builders
    def _new(self, shape) -> int:
        self.tensors.append(tuple(shape))
        return len(self.tensors) - 1

    def flatten(self, src: int) -> int:
        """(H,W,C) -> (1,1,HWC).  NHWC memory order == flatten order: a free view."""
        h, w, c = self.tensors[src]
        if h == 1 and w == 1:
            return src
        t = self._new((1, 1, h * w * c))
        self.units.append(Unit("view", src, t))
        return t

    def dense(self, src: int, name: str, features: int, act: str = "none",
              scope: Path = ()) -> int:
        p = self.param_root + tuple(scope) + (name,)
        h, w, c = self.tensors[src]
        dst = self._new((1, 1, features))
        if src != 0 and h * w > 1:
            # Dense on a flattened feature map == an (h x w) VALID convolution whose HWIO kernel, raveled, is the
            # (h*w*c, features) Dense kernel (NHWC flatten order) -- no copy, no view op on the device
            self.units.append(Unit("conv", src, dst, h, w, 1, 0, 0, c, features,
                                   kernel=p + ("kernel",), bias=p + ("bias",), act=act, flat_kernel=True))
            self.out = dst
            return dst
        src = self.flatten(src)
        cin = self.tensors[src][2]
        self.units.append(Unit("conv", src, dst, 1, 1, 1, 0, 0, cin, features,
                               kernel=p + ("kernel",), bias=p + ("bias",), act=act))
        self.out = dst
        return dst

    def conv(self, src: int, name: str, features: int, k: int, stride: int = 1,
             padding: str = "SAME", bn: Optional[str] = None, res: Optional[int] = None,
             act: str = "none", use_bias: bool = False, scope: Path = ()) -> int:
        h, w, cin = self.tensors[src]
        oh, ph = self._out_and_pad(h, k, stride, padding)
        ow, pw = self._out_and_pad(w, k, stride, padding)
        dst = self._new((oh, ow, features))
        p = self.param_root + tuple(scope) + (name,)
        u = Unit("conv", src, dst, k, k, stride, ph, pw, cin, features,
                 kernel=p + ("kernel",), bias=(p + ("bias",)) if use_bias else None,
                 res=res, act=act)
        if bn is not None:
            pb = self.param_root + tuple(scope) + (bn,)
            sb = tuple(scope) + (bn,)
            u.bn_scale, u.bn_bias = pb + ("scale",), pb + ("bias",)
            u.bn_mean, u.bn_var = sb + ("mean",), sb + ("var",)
        if res is not None:
            assert self.tensors[res] == self.tensors[dst], "residual shape mismatch"
        self.units.append(u)
        self.out = dst
        return dst

    @staticmethod
    def _out_and_pad(n_in: int, k: int, stride: int, padding):
        """(n_out, pad_lo) for 'SAME' (XLA/Flax: odd extra pad on the high side), 'VALID', or an explicit
        symmetric integer padding (torch convention)."""
        if padding == "SAME":
            return _same_pad(n_in, k, stride)
        if padding == "VALID":
            return (n_in - k) // stride + 1, 0
        if isinstance(padding, int):
            return (n_in + 2 * padding - k) // stride + 1, padding
        raise ValueError(padding)

    def maxpool(self, src: int, k: int = 3, stride: int = 2, padding="SAME") -> int:
        """``nn.max_pool`` (window k x k); padded positions never win."""
        h, w, c = self.tensors[src]
        oh, ph = self._out_and_pad(h, k, stride, padding)
        ow, pw = self._out_and_pad(w, k, stride, padding)
        dst = self._new((oh, ow, c))
        self.units.append(Unit("maxpool", src, dst, k, k, stride, ph, pw, c, c))
        self.out = dst
        return dst

    def avgpool(self, src: int, k: int = 2, stride: int = 2, padding="VALID") -> int:
        """``nn.avg_pool`` (window k x k, sum / k^2 with padding counted -- Flax's default)."""
        h, w, c = self.tensors[src]
        oh, ph = self._out_and_pad(h, k, stride, padding)
        ow, pw = self._out_and_pad(w, k, stride, padding)
        dst = self._new((oh, ow, c))
        self.units.append(Unit("avgpool", src, dst, k, k, stride, ph, pw, c, c))
        self.out = dst
        return dst

    def meanpool(self, src: int) -> int:
        h, w, c = self.tensors[src]
        dst = self._new((1, 1, c))
        self.units.append(Unit("meanpool", src, dst, cin=c, cout=c))
        self.out = dst
        return dst

    def prepare_input(self, x: torch.Tensor) -> torch.Tensor:
        """(B, *input_shape) -> (B, H, W, C) of tensor 0 (reshape + channel tiling)."""
        h, w, c = self.tensors[0]
        t = self.tile_channels
        xb = x.reshape(-1, h, w, c // t)
        return xb.repeat(1, 1, 1, t) if t > 1 else xb

    # ------------------------------------------------------------- properties
    @property
    def num_outputs(self) -> int:
        h, w, c = self.tensors[self.out]
        return h * w * c

    def macs_per_example(self) -> int:
        tot = 0
        for u in self.units:
            if u.kind == "conv":
                oh, ow, _ = self.tensors[u.dst]
                tot += oh * ow * u.kh * u.kw * u.cin * u.cout
        return tot

    # ----------------------------------------------------------------- params
    def init_params(self, seed: int, dtype=torch.float32, randomize_bn: bool = True):
        """Seeded random init (the reference's MAP checkpoints are absent —
        SURVEY G3).  LeCun-normal kernels like Flax; biases get small noise and BN
        statistics are randomised so that every term of the JVP/VJP is exercised."""
        g = torch.Generator().manual_seed(int(seed) % (2 ** 63 - 1))
        params: Dict[str, Any] = {}
        stats: Dict[str, Any] = {}

        def put(tree, path, val):
            for k in path[:-1]:
                tree = tree.setdefault(k, {})
            tree[path[-1]] = val.to(dtype)

        for u in self.units:
            if u.kind != "conv":
                continue
            fan_in = u.kh * u.kw * u.cin
            shape = (u.cin, u.cout) if (u.kh == 1 and u.kw == 1 and self._is_dense(u)) \
                else (u.kh * u.kw * u.cin, u.cout) if u.flat_kernel else (u.kh, u.kw, u.cin, u.cout)
            put(params, u.kernel, torch.randn(shape, generator=g, dtype=torch.float64) / math.sqrt(fan_in))
            if u.bias is not None:
                put(params, u.bias, 0.1 * torch.randn(u.cout, generator=g, dtype=torch.float64))
            if u.bn_scale is not None:
                if randomize_bn:
                    put(params, u.bn_scale, 1.0 + 0.2 * torch.randn(u.cout, generator=g, dtype=torch.float64))
                    put(params, u.bn_bias, 0.1 * torch.randn(u.cout, generator=g, dtype=torch.float64))
                    put(stats, u.bn_mean, 0.1 * torch.randn(u.cout, generator=g, dtype=torch.float64))
                    put(stats, u.bn_var, 0.5 + torch.rand(u.cout, generator=g, dtype=torch.float64))
                else:
                    put(params, u.bn_scale, torch.ones(u.cout, dtype=torch.float64))
                    put(params, u.bn_bias, torch.zeros(u.cout, dtype=torch.float64))
                    put(stats, u.bn_mean, torch.zeros(u.cout, dtype=torch.float64))
                    put(stats, u.bn_var, torch.ones(u.cout, dtype=torch.float64))
        return params, stats

    def _is_dense(self, u: Unit) -> bool:
        return u.bias is not None and u.bn_scale is None and self.tensors[u.src][:2] == (1, 1)

    # ---------------------------------------------------------------- forward
    def forward(self, params: Dict[str, Any], batch_stats: Dict[str, Any], x: torch.Tensor,
                return_all: bool = False):
        """Torch functional forward, any dtype/device, differentiable in ``params``.

        ``x`` is one example (``input_shape``) or a batch ``(B, *input_shape)``;
        returns ``(K,)`` or ``(B, K)``.  ``params`` is the tree containing
        ``param_root`` (e.g. ``{'params': {...}}``).
        """
        raw = self.input_shape_raw
        if tuple(x.shape) == raw:
            single = True
            xb = self.prepare_input(x[None])
        else:
            single = False
            xb = self.prepare_input(x)
        vals: Dict[int, torch.Tensor] = {0: xb}
        for u in self.units:
            a = vals[u.src]
            if u.kind == "view":
                vals[u.dst] = a.reshape((a.shape[0],) + self.tensors[u.dst])
                continue
            if u.kind == "meanpool":
                vals[u.dst] = a.mean(dim=(1, 2), keepdim=True)
                continue
            if u.kind == "maxpool":
                oh, ow, _ = self.tensors[u.dst]
                h, w = a.shape[1], a.shape[2]
                hi_h = max((oh - 1) * u.stride + u.kh - h - u.pad_h, 0)
                hi_w = max((ow - 1) * u.stride + u.kw - w - u.pad_w, 0)
                an = F.pad(a.permute(0, 3, 1, 2), (u.pad_w, hi_w, u.pad_h, hi_h), value=float("-inf"))
                vals[u.dst] = F.max_pool2d(an, (u.kh, u.kw), stride=u.stride).permute(0, 2, 3, 1)
                continue
            if u.kind == "avgpool":
                oh, ow, _ = self.tensors[u.dst]
                h, w = a.shape[1], a.shape[2]
                hi_h = max((oh - 1) * u.stride + u.kh - h - u.pad_h, 0)
                hi_w = max((ow - 1) * u.stride + u.kw - w - u.pad_w, 0)
                an = F.pad(a.permute(0, 3, 1, 2), (u.pad_w, hi_w, u.pad_h, hi_h))
                vals[u.dst] = F.avg_pool2d(an, (u.kh, u.kw), stride=u.stride).permute(0, 2, 3, 1)
                continue
            W = _get(params, u.kernel)
            if u.flat_kernel:
                z = (a.reshape(a.shape[0], -1) @ W).reshape(a.shape[0], 1, 1, u.cout)
            elif u.kh == 1 and u.kw == 1 and u.stride == 1:
                z = a @ W.reshape(u.cin, u.cout)
            elif u.kh == 1 and u.kw == 1:
                z = a[:, ::u.stride, ::u.stride, :] @ W.reshape(u.cin, u.cout)
            else:
                oh, ow, _ = self.tensors[u.dst]
                h, w = a.shape[1], a.shape[2]
                pad_hi_h = max((oh - 1) * u.stride + u.kh - h - u.pad_h, 0)
                pad_hi_w = max((ow - 1) * u.stride + u.kw - w - u.pad_w, 0)
                an = F.pad(a.permute(0, 3, 1, 2), (u.pad_w, pad_hi_w, u.pad_h, pad_hi_h))
                z = F.conv2d(an, W.permute(3, 2, 0, 1), stride=u.stride).permute(0, 2, 3, 1)
            if u.bias is not None:
                z = z + _get(params, u.bias)
            if u.bn_scale is not None:
                mean, var = _get(batch_stats, u.bn_mean), _get(batch_stats, u.bn_var)
                z = (z - mean) * torch.rsqrt(var + u.bn_eps) * _get(params, u.bn_scale) + _get(params, u.bn_bias)
            if u.res is not None:
                z = z + vals[u.res]
            vals[u.dst] = act_fn(u.act, z)
        out = vals[self.out].reshape(xb.shape[0], -1)
        if return_all:
            return out, vals
        return out[0] if single else out

    # --------------------------------------------------------------- apply_fn
    def make_apply_fn(self, model_type: str, batch_stats: Optional[Dict[str, Any]] = None):
        """A Flax-shaped ``apply_fn(variables, x, **kw)``.

        regressor:  ``apply_fn(p, x, return_logvar=True)`` -> ``mu`` or ``(mu, logvar)``
                    (``src/toymodels.py:9-24``)
        classifier: ``apply_fn(variables, x, train=False, mutable=False)`` -> logits
        ``variables`` may carry ``'batch_stats'``; otherwise the stats given here.
        """
        default_stats = batch_stats if batch_stats is not None else {}

        def apply_fn(variables, x, return_logvar: bool = True, train: bool = False, mutable=False, **_):
            stats = variables.get("batch_stats", default_stats) if isinstance(variables, dict) else default_stats
            if stats is None or (isinstance(stats, dict) and not stats):
                stats = default_stats
            out = self.forward(variables, stats, x)
            if model_type == "regressor" and return_logvar:
                return out, variables["logvar"]["logvar"]
            return out

        return apply_fn
