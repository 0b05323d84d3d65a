"""Second-order pass of the inducing-point gradient on the engine's kernels: the input derivative of the
parameter-JVP pairing

    phi(Z) = sum_{i, k} < J(z_i) m_{ik} , c L(z_i) e_k >,        grad_{z_i} phi

(``jax.value_and_grad(alternative_objective_scalable)`` differentiates through it, ``src/train_inducing.py:195-232``;
the directions m_ik = (Q W)_(i,k) are frozen, see ``train_inducing.py``).  It is reverse mode over the tangent tape:

  forward   da_l = act'(y_l) dy_l,   dy_l = s (conv(da_{l-1}, W) + conv(a_{l-1}, dW_ik)) + dgamma_ik xhat_l + dbeta_ik + dres
  reverse   given the adjoints DA_l (per probe) of da_l and A_l of a_l:
            DY = act' DA ;  Y = act' A + sum_k act'' dy_k DA_k ;  residual branches take DY / Y
            Z' = Y + sum_k (dgamma_ik / gamma) DY_k                      (BN: xhat depends on z)
            DA_{l-1} += convT(s DY, W) ;  A_{l-1} += convT(s Z', W) + sum_k convT(s DY_k, dW_ik)

The direction differs per EXAMPLE, so the segments with a weight tangent run with the examples on the probe axis
(``n_img = 1``, one launch per direction k: "probe" i reads a_i and dW_ik — every operand offset is linear in i); the
shared-weight segments run over all examples at once.  Every convolution
/ transposed convolution is an ``LIP_OP_IGEMM`` launched through ``lip_engine_run_op`` — the transposed ones read
the HWIO kernel in place (``LIP_SEG_B_TRANS``: a weight tangent has no transposed copy) — the per-element glue between
them (act', act'', BN factors, sums over the K probes) is torch algebra on the arena the ops read and write.

Supported: the units of the reference's MLPs and ResNet1M (Dense / conv with bias or eval-mode BN, residual adds,
ReLU / tanh / identity anywhere, GELU on Dense layers, global mean pool, flatten views) and the window pools of
LeNet5 / the ResNet-50 stem (max pool is piecewise linear: tangents and adjoints gather / scatter at the cached argmax).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import _native as nv
from .netspec import _get


class EngineExecutor:
    """Runs host-built ops on a :class:`~lip_amd.engine.LinearizedNet` binding (float32, device)."""

    def __init__(self, eng):
        self.eng = eng
        self.cn, self.device, self.dtype = eng.cn, eng.device, torch.float32
        self.prim, self.consts, self.theta = eng.prim, eng.consts, eng.theta
        self.max_probes = eng.chunk

    def run(self, op, P, V, Y):
        self.eng.run_op(op, P, V=V, Y=Y)


def _ref(space, off=0, pstride=0):
    return nv.Ref(int(space), 0, int(off), int(pstride))


NONE = _ref(nv.SP_NONE)


def _igemm(n_img, OH, OW, N, seg, out, res=None):
    o = nv.Op()
    o.kind, o.nseg = nv.OP_IGEMM, 1
    for name in nv.REF_FIELDS:
        setattr(o, name, NONE)
    for i in range(3):
        o.seg[i].a = NONE
        o.seg[i].b = NONE
    s = o.seg[0]
    s.a, s.b = seg["a"], seg["b"]
    for k in ("IH", "IW", "C", "KH", "KW", "stride", "pad_h", "pad_w", "mode", "flags"):
        setattr(s, k, int(seg.get(k, 0)))
    o.n_img, o.OH, o.OW, o.N = int(n_img), int(OH), int(OW), int(N)
    o.out = out
    if res is not None:
        o.res = res
    return o


def _pool(kind, n_img, u, tens, a_ref, out_ref, amax):
    """window pool of unit u (``LIP_OP_MAXPOOL_FWD``: gather at the cached argmax / window average) or its transpose
    (``LIP_OP_MAXPOOL_BWD``) — both linear in the pooled tensor, so they serve tangents and adjoints alike."""
    o = nv.Op()
    o.kind, o.nseg = kind, 1
    for name in nv.REF_FIELDS:
        setattr(o, name, NONE)
    for i in range(3):
        o.seg[i].a = NONE
        o.seg[i].b = NONE
    ih, iw, cc = tens[u.src]
    oh, ow, _ = tens[u.dst]
    sg = o.seg[0]
    sg.a = a_ref
    sg.IH, sg.IW, sg.C, sg.KH, sg.KW, sg.stride, sg.pad_h, sg.pad_w = ih, iw, cc, u.kh, u.kw, u.stride, u.pad_h, u.pad_w
    o.n_img, o.OH, o.OW, o.N = int(n_img), oh, ow, cc
    o.out = out_ref
    o.aux0 = amax
    return o


def _act2(name: str, a: torch.Tensor, dphi: Optional[torch.Tensor], y: Optional[torch.Tensor]):
    """act''(y) from what the primal pass cached (a = act(y), dphi = act'(y)); None when it vanishes."""
    if name in ("none", "relu"):
        return None
    if name == "tanh":
        return -2.0 * a * dphi
    if name == "gelu":
        if y is None:
            raise NotImplementedError("second-order pass: GELU is supported on Dense layers only")
        c0 = math.sqrt(2.0 / math.pi)
        u = c0 * (y + 0.044715 * y ** 3)
        t = torch.tanh(u)
        du = c0 * (1.0 + 3 * 0.044715 * y ** 2)
        d2u = c0 * 6 * 0.044715 * y
        sech2 = 1.0 - t * t
        # phi = 0.5 y (1 + t):  phi' = 0.5 (1 + t) + 0.5 y sech2 du ;  phi'' = sech2 du + 0.5 y (sech2 d2u - 2 t sech2 du^2)
        return sech2 * du + 0.5 * y * (sech2 * d2u - 2.0 * t * sech2 * du * du)
    raise ValueError(name)


def input_grad_of_pairing(ex, Mdir: torch.Tensor, c_out: float, model_type: str, out_weights: Optional[torch.Tensor] = None,
                          shared: bool = False) -> torch.Tensor:
    """grad_Z of the parameter-JVP pairing of the binding behind ``ex``, T directions per example:

      per-example directions (default):  sum_{i,t} < J(z_i) Mdir[i, t] , c_out L(z_i) x_it >,   Mdir (n, T, D)
      ``shared=True``:                   sum_{i,t} < J(z_i) Mdir[t]    , c_out L(z_i) x_it >,   Mdir (T, D)

    with x_it = ``out_weights[i, t]`` (n, T, K); ``None`` means x_it = e_t (T = K, the exact objective's pairing).
    Shared directions are what the rank-one cotangent of the stochastic objective consists of (``stochastic_grad.py``):
    the tangent pass is then the engine's ordinary probe-batched one (one launch per segment for all examples and
    directions).  Returns (n, *input_shape_raw)."""
    cn = ex.cn
    net, n, D, K = cn.net, cn.n, cn.D, cn.K
    dev, dt = ex.device, ex.dtype
    tens = net.tensors
    meta = cn.meta
    layout = meta["layout"]
    if shared:
        if Mdir.dim() != 2 or Mdir.shape[1] != D:
            raise ValueError(f"shared directions must be (T, D) with D = {D}, got {tuple(Mdir.shape)}")
        T = int(Mdir.shape[0])
    else:
        if Mdir.dim() != 3 or Mdir.shape[0] != n or Mdir.shape[2] != D:
            raise ValueError(f"directions must be (n, T, D) = ({n}, T, {D}), got {tuple(Mdir.shape)}")
        T = int(Mdir.shape[1])
    if out_weights is None:
        if T != K:
            raise ValueError(f"without out_weights the pairing takes T = K = {K} directions (x_it = e_t), got {T}")
        Xw = torch.eye(K, device=dev, dtype=torch.float64)[None].expand(n, K, K)
    else:
        if tuple(out_weights.shape) != (n, T, K):
            raise ValueError(f"out_weights must be (n, T, K) = {(n, T, K)}, got {tuple(out_weights.shape)}")
        Xw = out_weights.to(device=dev, dtype=torch.float64)
    if T > ex.max_probes:
        raise NotImplementedError(f"second-order pass: {T} directions exceed the engine's chunk of {ex.max_probes} "
                                  "(split the directions: the pairing is a sum over them)")
    Mdir = Mdir.to(device=dev, dtype=dt).contiguous()
    classifier = model_type == "classifier"

    def size(t):
        h, w, c = tens[t]
        return h * w * c

    def poff(path):
        return layout[path][0]

    def prim(off, count):
        return ex.prim[off:off + count]

    def pdir(path):
        """(T, n, numel) view — (T, 1, numel) for shared directions — of the direction's slice for one parameter leaf."""
        off, shape = layout[path]
        cnt = math.prod(shape) if shape else 1
        if shared:
            return Mdir[:, None, off:off + cnt]
        return Mdir[:, :, off:off + cnt].permute(1, 0, 2)

    has_tan: Dict[int, bool] = {0: False}
    for u in net.units:
        has_tan[u.dst] = True if u.kind == "conv" else has_tan[u.src]

    # ---------------------------------------------------------------- arena: every buffer the ops touch
    alias: Dict[int, int] = {}
    for u in net.units:
        if u.kind == "view":
            alias[u.dst] = alias.get(u.src, u.src)
    root = lambda t: alias.get(t, t)
    slots: Dict[str, int] = {}
    top = 0

    def alloc(name, count):
        nonlocal top
        slots[name] = top
        top += (count + 3) // 4 * 4

    maxsz = max(size(t) for t in range(len(tens)))
    for t in range(len(tens)):
        if root(t) != t:
            continue
        if has_tan.get(t, False):
            alloc(f"T{t}", T * n * size(t))
            alloc(f"DA{t}", T * n * size(t))
        alloc(f"A{t}", n * size(t))
    for nm, cnt in (("ACC", T * n * maxsz), ("SDY", T * n * maxsz), ("SZB", n * maxsz), ("TMP", T * n * maxsz)):
        alloc(nm, cnt)
    arena = torch.zeros(top, device=dev, dtype=dt)

    def buf(name, *shape):
        cnt = math.prod(shape)
        return arena[slots[name]:slots[name] + cnt].view(*shape)

    Y = lambda name, off=0, ps=0: _ref(nv.SP_YOUT, slots[name] + off, ps)
    dys: Dict[int, torch.Tensor] = {}

    def geom(u):
        ih, iw, _ = tens[u.src]
        return dict(IH=ih, IW=iw, C=u.cin, KH=u.kh, KW=u.kw, stride=u.stride, pad_h=u.pad_h, pad_w=u.pad_w, mode=0, flags=0)

    def geom_t(u):
        """transposed convolution of unit u: gathers the cotangent of u.dst, produces a tensor shaped like u.src"""
        oh, ow, _ = tens[u.dst]
        return dict(IH=oh, IW=ow, C=u.cout, KH=u.kh, KW=u.kw, stride=u.stride, pad_h=u.pad_h, pad_w=u.pad_w, mode=1,
                    flags=nv.SEG_B_TRANS)

    def bn_factors(u):
        if u.bn_scale is None:
            return None
        so, ro = meta["s_off"][u.dst], meta["rstd_off"][u.dst]
        return ex.consts[so:so + u.cout], ex.consts[ro:ro + u.cout]          # s = gamma rstd, rstd

    def unit_primals(u):
        """(a, dphi or None, act'' or None) of unit u's output, each (n, size) or None"""
        sz = size(u.dst)
        a = prim(cn.a_off[u.dst], n * sz).view(n, sz)
        dphi = prim(meta["dphi_off"][u.dst], n * sz).view(n, sz) if u.dst in meta["dphi_off"] else None
        y = None
        if u.act == "gelu":
            ih, iw, _ = tens[u.src]
            if not (u.kh == ih and u.kw == iw and tens[u.dst][:2] == (1, 1) and u.bn_scale is None and u.res is None):
                raise NotImplementedError("second-order pass: GELU is supported on Dense layers only")
            a_src = prim(cn.a_off[u.src], n * size(u.src)).view(n, size(u.src))
            W = ex.theta[poff(u.kernel):poff(u.kernel) + u.kh * u.kw * u.cin * u.cout].view(-1, u.cout)
            y = a_src @ W
            if u.bias is not None:
                y = y + ex.theta[poff(u.bias):poff(u.bias) + u.cout]
        return a, dphi, _act2(u.act, a, dphi, y)

    # ---------------------------------------------------------------- forward: tangents of every tensor, kept
    for u in net.units:
        if u.kind == "view":
            continue
        src, dst = root(u.src), u.dst
        if u.kind == "meanpool":
            h, w, c = tens[u.src]
            Ts = buf(f"T{src}", T, n, h * w, c)
            buf(f"T{dst}", T, n, c).copy_(Ts.mean(2))
            continue
        if u.kind in ("maxpool", "avgpool"):
            am = meta["amax_off"][dst]
            amax = NONE if am is None else _ref(nv.SP_PRIM, am)
            ex.run(_pool(nv.OP_MAXPOOL_FWD, n, u, tens, Y(f"T{src}", 0, n * size(u.src)), Y(f"T{dst}", 0, n * size(dst)), amax),
                   T, Mdir, arena)
            continue
        if u.kind != "conv":
            raise NotImplementedError(f"second-order pass: unit kind '{u.kind}' is not supported")
        oh, ow, co = tens[dst]
        sz, ssz = size(dst), size(u.src)
        acc = buf("ACC", T, n, sz)
        has_in = has_tan[u.src]
        if has_in:   # conv(da, W) over all examples
            ex.run(_igemm(n, oh, ow, co, dict(geom(u), a=Y(f"T{src}", 0, n * ssz), b=_ref(nv.SP_THETA, poff(u.kernel))),
                          Y("ACC", 0, n * sz)), T, Mdir, arena)
        if shared:
            # + conv(a, dW_t): the direction is the same for every example — the engine's own probe-batched segment
            seg = dict(geom(u), a=_ref(nv.SP_PRIM, cn.a_off[u.src], 0), b=_ref(nv.SP_VIN, poff(u.kernel), D))
            o = Y("ACC", 0, n * sz)
            ex.run(_igemm(n, oh, ow, co, seg, o, res=o if has_in else None), T, Mdir, arena)
        else:
            # + conv(a_i, dW_it): the weight tangent is the example's own.  One launch per direction t with the EXAMPLES
            # on the probe axis (n_img = 1, "probe" i reads a_i, dW_it and writes slot (t, i)): every operand offset is
            # linear in i
            for k in range(T):
                for i0 in range(0, n, ex.max_probes):
                    seg = dict(geom(u), a=_ref(nv.SP_PRIM, cn.a_off[u.src] + i0 * ssz, ssz),
                               b=_ref(nv.SP_VIN, (i0 * T + k) * D + poff(u.kernel), T * D))
                    o = Y("ACC", (k * n + i0) * sz, sz)
                    ex.run(_igemm(1, oh, ow, co, seg, o, res=o if has_in else None), min(ex.max_probes, n - i0), Mdir, arena)
        a, dphi, _ = unit_primals(u)
        dy = acc.view(T, n, oh * ow, co)
        bn = bn_factors(u)
        if bn is not None:
            s, _ = bn
            xhat = prim(meta["xhat_off"][dst], n * sz).view(1, n, oh * ow, co)
            dy = dy * s + pdir(u.bn_scale).unsqueeze(2) * xhat + pdir(u.bn_bias).unsqueeze(2)
        elif u.bias is not None:
            dy = dy + pdir(u.bias).unsqueeze(2)
        dy = dy.reshape(T, n, sz)
        if u.res is not None and has_tan[u.res]:
            dy = dy + buf(f"T{root(u.res)}", T, n, sz)
        if u.act in ("tanh", "gelu"):
            dys[dst] = dy.clone()
        buf(f"T{dst}", T, n, sz).copy_(dy * dphi if dphi is not None else dy)

    # ---------------------------------------------------------------- head: adjoints of the logits' tangent / value
    # phi_i = sum_t < U_it , c L(p(f_i)) x_it >:  DA = d phi / d U = c L x (per direction),  A = d phi / d f through p —
    # a K-sized closed form per example, differentiated by autograd on the (n, K) logits (T n K elements)
    out_t = root(net.out)
    U = buf(f"T{out_t}", T, n, K).permute(1, 0, 2).double()           # U[i, t, a] = (J(z_i) m_it)_a
    if classifier:
        f = prim(cn.a_off[net.out], n * K).view(n, K).double().clone().requires_grad_(True)
        with torch.enable_grad():
            p = torch.softmax(f, dim=-1)
            sq = torch.sqrt(p)
            Lx = sq[:, None, :] * Xw - p[:, None, :] * (sq[:, None, :] * Xw).sum(-1, keepdim=True)   # (L x)_a = s_a x_a - p_a <s, x>
            phi = c_out * (U.detach() * Lx).sum()
            fbar, = torch.autograd.grad(phi, f)
        ubar = (c_out * Lx).detach()
    else:
        ubar = float(c_out) * Xw
        fbar = torch.zeros(n, K, device=dev, dtype=torch.float64)
    buf(f"DA{out_t}", T, n, K).copy_(ubar.permute(1, 0, 2).to(dt))
    buf(f"A{out_t}", n, K).copy_(fbar.to(dt))

    # ---------------------------------------------------------------- reverse
    for u in reversed(net.units):
        if u.kind == "view":
            continue
        src, dst = root(u.src), u.dst
        if u.kind == "meanpool":
            h, w, c = tens[u.src]
            if has_tan[u.src]:
                buf(f"DA{src}", T, n, h * w, c).add_(buf(f"DA{dst}", T, n, 1, c) / (h * w))
            buf(f"A{src}", n, h * w, c).add_(buf(f"A{dst}", n, 1, c) / (h * w))
            continue
        if u.kind in ("maxpool", "avgpool"):
            am = meta["amax_off"][dst]
            amax = NONE if am is None else _ref(nv.SP_PRIM, am)
            sz, ssz = size(dst), size(u.src)
            if has_tan[u.src]:
                ex.run(_pool(nv.OP_MAXPOOL_BWD, n, u, tens, Y(f"DA{dst}", 0, n * sz), Y("TMP", 0, n * ssz), amax), T, Mdir, arena)
                buf(f"DA{src}", T, n, ssz).add_(buf("TMP", T, n, ssz))
            ex.run(_pool(nv.OP_MAXPOOL_BWD, n, u, tens, Y(f"A{dst}", 0, 0), Y("TMP", 0, 0), amax), 1, Mdir, arena)
            buf(f"A{src}", n, ssz).add_(buf("TMP", n, ssz))
            continue
        oh, ow, co = tens[dst]
        sz, ssz = size(dst), size(u.src)
        sh, sw, sc = tens[u.src]
        a, dphi, ddphi = unit_primals(u)
        DA, AB = buf(f"DA{dst}", T, n, sz), buf(f"A{dst}", n, sz)
        DYb = DA * dphi if dphi is not None else DA
        Yb = AB * dphi if dphi is not None else AB.clone()
        if ddphi is not None:
            Yb = Yb + (ddphi * dys[dst] * DA).sum(0)
        if u.res is not None:
            r = root(u.res)
            if has_tan[u.res]:
                buf(f"DA{r}", T, n, sz).add_(DYb)
            buf(f"A{r}", n, sz).add_(Yb)
        bn = bn_factors(u)
        sdy, szb = buf("SDY", T, n, oh * ow, co), buf("SZB", n, oh * ow, co)
        DYv, Ybv = DYb.reshape(T, n, oh * ow, co), Yb.reshape(n, oh * ow, co)
        if bn is not None:
            s, rstd = bn
            # x-hat depends on z: its adjoint is dgamma_t rstd DY_t (written with rstd, not (dgamma / gamma) s, so a
            # zero-initialised BN scale gives 0, not inf * 0)
            sdy.copy_(DYv * s)
            szb.copy_(Ybv * s + (pdir(u.bn_scale).unsqueeze(2) * rstd * DYv).sum(0))
        else:
            sdy.copy_(DYv)
            szb.copy_(Ybv)
        tmp = buf("TMP", T, n, ssz)
        wref = _ref(nv.SP_THETA, poff(u.kernel))
        if has_tan[u.src]:   # DA_src += convT(s DY, W)
            ex.run(_igemm(n, sh, sw, sc, dict(geom_t(u), a=Y("SDY", 0, n * sz), b=wref), Y("TMP", 0, n * ssz)), T, Mdir, arena)
            buf(f"DA{src}", T, n, ssz).add_(tmp)
        # A_src += convT(s Z', W)   (one "probe")
        ex.run(_igemm(n, sh, sw, sc, dict(geom_t(u), a=Y("SZB", 0, 0), b=wref), Y("TMP", 0, 0)), 1, Mdir, arena)
        buf(f"A{src}", n, ssz).add_(buf("TMP", n, ssz))
        # A_src_i += sum_t convT(s DY_it, dW_it)
        if shared:
            seg = dict(geom_t(u), a=Y("SDY", 0, n * sz), b=_ref(nv.SP_VIN, poff(u.kernel), D))
            ex.run(_igemm(n, sh, sw, sc, seg, Y("TMP", 0, n * ssz)), T, Mdir, arena)
        else:
            for k in range(T):   # examples on the probe axis, as in the forward pass
                for i0 in range(0, n, ex.max_probes):
                    seg = dict(geom_t(u), a=Y("SDY", (k * n + i0) * sz, sz), b=_ref(nv.SP_VIN, (i0 * T + k) * D + poff(u.kernel), T * D))
                    ex.run(_igemm(1, sh, sw, sc, seg, Y("TMP", (k * n + i0) * ssz, ssz)), min(ex.max_probes, n - i0), Mdir, arena)
        buf(f"A{src}", n, ssz).add_(tmp.sum(0))

    h0, w0, c0 = tens[0]
    g = buf("A0", n, h0, w0, c0)
    t = net.tile_channels
    if t > 1:
        g = g.reshape(n, h0, w0, t, c0 // t).sum(3)
    return g.reshape((n,) + tuple(net.input_shape_raw)).clone()


def input_grad_of_rank_one_terms(ex, terms, c_out: float, model_type: str, max_directions: Optional[int] = None) -> torch.Tensor:
    """grad_Z of  sum over (U, X) in ``terms``, rows t:  U[t]^T W(Z) X[t]  =  sum_t sum_i < J(z_i) U[t] , c L(z_i) X[t, i] >
    — the cotangent of the factor W as ``stochastic_grad.stochastic_objective_and_cotangent`` returns it (U (T_b, D),
    X (T_b, n K)).  The pairing is a sum over the directions, so they are consumed in chunks of at most
    ``max_directions`` (default: what the engine's probe chunk and ~1/4 of the free device memory allow)."""
    cn = ex.cn
    n, K = cn.n, cn.K
    per_dir = 4 * n * (2 * sum(h * w * c for (h, w, c) in cn.net.tensors) + 3 * max(h * w * c for (h, w, c) in cn.net.tensors)) + 4 * cn.D
    if max_directions is None:
        free = torch.cuda.mem_get_info(ex.device)[0] if ex.device.type == "cuda" else (8 << 30)
        max_directions = max(1, min(ex.max_probes, int(free // 4 // per_dir)))
    max_directions = min(int(max_directions), ex.max_probes)
    g = None
    for U, X in terms:
        for t0 in range(0, U.shape[0], max_directions):
            Ub = U[t0:t0 + max_directions]
            Xb = X[t0:t0 + max_directions].reshape(Ub.shape[0], n, K).permute(1, 0, 2)
            gi = input_grad_of_pairing(ex, Ub, c_out, model_type, out_weights=Xb, shared=True)
            g = gi if g is None else g + gi
    return g
