"""NetSpec -> op tapes for the HIP engine, and the handle wrapper around the C ABI.

``compile_net`` is pure Python (no GPU): it lays out the cached primal tensors, the
per-binding constants and the probe workspace, and emits the three op tapes of
``include/lip.h``:

  primal   : per unit  IGEMM(z = conv(a, W))  ->  PRIMAL_POST(xhat, a = act(y), act'(y));  SOFTMAX
  tangent  : per unit  IGEMM(conv(da, W) + conv(a, dW_p), epilogue: BN/bias tangent, residual, act');
             HEAD
  backward : HEAD; per tensor (reverse)  IGEMM(act' * (sum_consumers convT(g, W^T s) + res),
             reductions into the bias / BN cotangents);  WGRAD(dW_p)

The linearisation point never changes across Krylov iterations, so everything primal is
cached once per binding — the reference re-evaluates the forward pass three times per example
per matvec (``src/ggn.py:139,140,142``).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from typing import Any, Dict, List, Optional, Tuple

import torch

from . import _native as nv
from .netspec import ACT_IDS, NetSpec, Unit, _get
from .utils import param_layout


def _r4(x: int) -> int:
    return (x + 3) // 4 * 4


class VT:
    """A virtual workspace tensor (per-probe size in floats), placed by the allocator."""
    __slots__ = ("size", "off", "name")

    def __init__(self, size: int, name: str = ""):
        self.size, self.off, self.name = _r4(size), None, name


NONE = (nv.SP_NONE, 0, 0)


@dataclasses.dataclass
class SymOp:
    kind: int
    f: Dict[str, Any]                 # scalar fields
    refs: Dict[str, Any]              # ref fields: (space, off, pstride) or ('work', VT, pstride)
    segs: List[Dict[str, Any]] = dataclasses.field(default_factory=list)

    def work_tensors(self):
        for r in list(self.refs.values()) + [s[k] for s in self.segs for k in ("a", "b")]:
            if r[0] == "work":
                yield r[1]


@dataclasses.dataclass
class CompiledNet:
    net: NetSpec
    n: int
    D: int
    K: int
    classifier: bool
    offsets: Dict[Tuple, Tuple[int, Tuple[int, ...]]]
    prim_floats: int
    const_floats: int
    work_pp: int
    tapes: List[List[nv.Op]]
    a_off: Dict[int, int]
    const_plan: List[Tuple]           # how to fill the constants buffer
    prob_off: int
    input_off: int
    meta: Dict[str, Any] = dataclasses.field(default_factory=dict)   # per-tensor offsets the second-order pass needs


def _seg(a, b, IH, IW, Cc, KH, KW, stride, pad_h, pad_w, mode):
    return dict(a=a, b=b, IH=IH, IW=IW, C=Cc, KH=KH, KW=KW, stride=stride, pad_h=pad_h, pad_w=pad_w, mode=mode)


def compile_net(net: NetSpec, n: int, params: Dict[str, Any]) -> CompiledNet:
    layout = {path: (off, shape) for path, off, shape in param_layout(params)}
    D = sum(int(torch.Size(s).numel()) for _, s in layout.values())
    K = net.num_outputs
    classifier = getattr(net, "model_type", "classifier") == "classifier"
    tens = net.tensors

    def poff(path) -> int:
        if path not in layout:
            raise KeyError(f"parameter {path} is not in state.params")
        return layout[path][0]

    # ------------------------------------------------------------------ primal / const layout
    prim = 0

    def palloc(sz):
        nonlocal prim
        o = prim
        prim += _r4(sz)
        return o

    a_off: Dict[int, int] = {0: palloc(n * tens[0][0] * tens[0][1] * tens[0][2])}
    dphi_off: Dict[int, int] = {}
    xhat_off: Dict[int, int] = {}
    amax_off: Dict[int, int] = {}
    producer: Dict[int, Unit] = {}
    zmax = 0
    for u in net.units:
        producer[u.dst] = u
        h, w, c = tens[u.dst]
        if u.kind == "view":
            a_off[u.dst] = a_off[u.src]
            continue
        a_off[u.dst] = palloc(n * h * w * c)
        if u.kind == "maxpool":
            amax_off[u.dst] = palloc(n * h * w * c)
        if u.kind == "avgpool":
            amax_off[u.dst] = None             # a NONE argmax ref selects the window average (include/lip.h)
        if u.kind == "conv":
            zmax = max(zmax, n * h * w * c)
            if u.act != "none":
                dphi_off[u.dst] = palloc(n * h * w * c)
            if u.bn_scale is not None:
                xhat_off[u.dst] = palloc(n * h * w * c)
    z_off = palloc(zmax)
    prob_off = palloc(n * K)
    sqrtp_off = palloc(n * K)

    cst = 0
    const_plan: List[Tuple] = []
    s_off: Dict[int, int] = {}
    mean_off: Dict[int, int] = {}
    rstd_off: Dict[int, int] = {}
    wt_off: Dict[int, int] = {}

    def calloc(sz):
        nonlocal cst
        o = cst
        cst += _r4(sz)
        return o

    for u in net.units:
        if u.kind != "conv":
            continue
        if u.bn_scale is not None:
            s_off[u.dst] = calloc(u.cout)
            mean_off[u.dst] = calloc(u.cout)
            rstd_off[u.dst] = calloc(u.cout)
            const_plan.append(("bn", u, s_off[u.dst], mean_off[u.dst], rstd_off[u.dst]))
        if u.src != 0 and producer.get(u.src, None) is not None and _has_grad(net, u.src):
            wt_off[u.dst] = calloc(u.kh * u.kw * u.cout * u.cin)
            const_plan.append(("wt", u, wt_off[u.dst], s_off.get(u.dst)))
    cst = max(cst, 4)

    P = lambda off: (nv.SP_PRIM, off, 0)
    TH = lambda off: (nv.SP_THETA, off, 0)
    CS = lambda off: (nv.SP_CONST, off, 0)
    VIN = lambda off: (nv.SP_VIN, off, D)
    YO = lambda off: (nv.SP_YOUT, off, D)

    def geom(u: Unit):
        ih, iw, _ = tens[u.src]
        return dict(IH=ih, IW=iw, Cc=u.cin, KH=u.kh, KW=u.kw, stride=u.stride, pad_h=u.pad_h, pad_w=u.pad_w)

    def AMAX(t):
        return NONE if amax_off[t] is None else P(amax_off[t])

    def pool_seg(u: Unit, a_ref):
        ih, iw, cc = tens[u.src]
        return _seg(a_ref, NONE, ih, iw, cc, u.kh, u.kw, u.stride, u.pad_h, u.pad_w, 0)

    # ------------------------------------------------------------------ primal tape
    primal: List[SymOp] = []
    for u in net.units:
        if u.kind == "view":
            continue
        oh, ow, c = tens[u.dst]
        if u.kind == "meanpool":
            ih, iw, _ = tens[u.src]
            primal.append(SymOp(nv.OP_POOL_FWD, dict(n_img=n, OH=ih, OW=iw, N=c, fscale=1.0 / (ih * iw)),
                                dict(out=P(a_off[u.dst])), [dict(a=P(a_off[u.src]), b=NONE)]))
            continue
        if u.kind in ("maxpool", "avgpool"):
            primal.append(SymOp(nv.OP_MAXPOOL_PRIMAL, dict(n_img=n, OH=oh, OW=ow, N=c),
                                dict(out=P(a_off[u.dst]), aux0=AMAX(u.dst)), [pool_seg(u, P(a_off[u.src]))]))
            continue
        g = geom(u)
        primal.append(SymOp(nv.OP_IGEMM, dict(n_img=n, OH=oh, OW=ow, N=c), dict(out=P(z_off)),
                            [_seg(P(a_off[u.src]), TH(poff(u.kernel)), g["IH"], g["IW"], g["Cc"], g["KH"], g["KW"],
                                  g["stride"], g["pad_h"], g["pad_w"], 0)]))
        refs = dict(out=P(a_off[u.dst]))
        if u.dst in dphi_off:
            refs["out2"] = P(dphi_off[u.dst])
        if u.bias is not None:
            refs["e0"] = TH(poff(u.bias))
        if u.bn_scale is not None:
            refs["out3"] = P(xhat_off[u.dst])
            refs["e1"] = TH(poff(u.bn_scale))
            refs["scale"] = TH(poff(u.bn_bias))
            refs["aux0"] = CS(mean_off[u.dst])
            refs["aux1"] = CS(rstd_off[u.dst])
        if u.res is not None:
            refs["res"] = P(a_off[u.res])
        primal.append(SymOp(nv.OP_PRIMAL_POST, dict(n_img=n, OH=oh, OW=ow, N=c, act=ACT_IDS[u.act]), refs,
                            [dict(a=P(z_off), b=NONE)]))
    if classifier:
        primal.append(SymOp(nv.OP_SOFTMAX, dict(n_img=n, OH=1, OW=1, N=K),
                            dict(out=P(prob_off), out2=P(sqrtp_off)), [dict(a=P(a_off[net.out]), b=NONE)]))

    # ------------------------------------------------------------------ tangent tape
    def tsize(t):
        h, w, c = tens[t]
        return n * h * w * c

    W = lambda vt: ("work", vt, vt.size)
    g_head = VT(n * K, "g_head")           # reserved: head output / backward input, offset 0 in both tapes
    tangent: List[SymOp] = []
    da: Dict[int, Optional[VT]] = {0: None}
    for u in net.units:
        if u.kind == "view":
            da[u.dst] = da[u.src]
            continue
        oh, ow, c = tens[u.dst]
        out = VT(tsize(u.dst), f"da{u.dst}")
        if u.kind == "meanpool":
            ih, iw, _ = tens[u.src]
            tangent.append(SymOp(nv.OP_POOL_FWD, dict(n_img=n, OH=ih, OW=iw, N=c, fscale=1.0 / (ih * iw)),
                                 dict(out=W(out)), [dict(a=W(da[u.src]), b=NONE)]))
            da[u.dst] = out
            continue
        if u.kind in ("maxpool", "avgpool"):
            tangent.append(SymOp(nv.OP_MAXPOOL_FWD, dict(n_img=n, OH=oh, OW=ow, N=c),
                                 dict(out=W(out), aux0=AMAX(u.dst)), [pool_seg(u, W(da[u.src]))]))
            da[u.dst] = out
            continue
        g = geom(u)
        segs = []
        if da[u.src] is not None:
            segs.append(_seg(W(da[u.src]), TH(poff(u.kernel)), g["IH"], g["IW"], g["Cc"], g["KH"], g["KW"],
                             g["stride"], g["pad_h"], g["pad_w"], 0))
        segs.append(_seg(P(a_off[u.src]), VIN(poff(u.kernel)), g["IH"], g["IW"], g["Cc"], g["KH"], g["KW"],
                         g["stride"], g["pad_h"], g["pad_w"], 0))
        refs = dict(out=W(out))
        if u.bias is not None:
            refs["e0"] = VIN(poff(u.bias))
        if u.bn_scale is not None:
            refs["scale"] = CS(s_off[u.dst])
            refs["e0"] = VIN(poff(u.bn_bias))
            refs["e1"] = VIN(poff(u.bn_scale))
            refs["xhat"] = P(xhat_off[u.dst])
        if u.res is not None and da[u.res] is not None:
            refs["res"] = W(da[u.res])
        if u.dst in dphi_off:
            refs["dphi"] = P(dphi_off[u.dst])
        tangent.append(SymOp(nv.OP_IGEMM, dict(n_img=n, OH=oh, OW=ow, N=c), refs, segs))
        da[u.dst] = out
    head_refs = dict(out=W(g_head), out2=(nv.SP_HEAD, 0, n * K))
    if classifier:
        head_refs["aux0"] = P(prob_off)
        head_refs["aux1"] = P(sqrtp_off)
    tangent.append(SymOp(nv.OP_HEAD, dict(n_img=n, OH=1, OW=1, N=K, classifier=int(classifier)),
                         dict(head_refs), [dict(a=W(da[net.out]), b=NONE)]))

    # ------------------------------------------------------------------ backward tape
    backward: List[SymOp] = []
    # head input lives in HEAD space (vjp calls); its seg a is unused in modes L / IN
    backward.append(SymOp(nv.OP_HEAD, dict(n_img=n, OH=1, OW=1, N=K, classifier=int(classifier)),
                          dict(head_refs), [dict(a=W(g_head), b=NONE)]))
    consumers: Dict[int, List[Tuple[Unit, str]]] = {}
    for u in net.units:
        consumers.setdefault(u.src, []).append((u, "src"))
        if u.res is not None:
            consumers.setdefault(u.res, []).append((u, "res"))
    gp: Dict[int, VT] = {net.out: g_head}

    def param_reds(u: Unit) -> Dict[str, Any]:
        """reductions into the bias / BN cotangents of the unit that produced the tensor"""
        r: Dict[str, Any] = {}
        if u.kind != "conv":
            return r
        if u.bn_scale is not None:
            r["red0"] = YO(poff(u.bn_bias))
            r["red1"] = YO(poff(u.bn_scale))
            r["xhat2"] = P(xhat_off[u.dst])
        elif u.bias is not None:
            r["red0"] = YO(poff(u.bias))
        return r

    order = [u.dst for u in net.units if u.kind != "view"]
    for t in reversed(order):
        u = producer[t]
        if not _has_grad(net, t):
            continue
        h, w, c = tens[t]
        if t != net.out:
            cons = consumers.get(t, [])
            conv_c = [cu for cu, role in cons if role == "src" and cu.kind == "conv"]
            res_c = [cu for cu, role in cons if role == "res"]
            pool_c = [cu for cu, role in cons if role == "src" and cu.kind == "meanpool"]
            mpool_c = [cu for cu, role in cons if role == "src" and cu.kind in ("maxpool", "avgpool")]
            view_c = [cu for cu, role in cons if cu.kind == "view"]
            if view_c:
                raise NotImplementedError("flatten of a non-input tensor is not supported by the HIP engine yet")
            if len(res_c) > 1 or len(conv_c) > 3 or ((pool_c or mpool_c) and (conv_c or res_c)) or len(mpool_c) > 1:
                raise NotImplementedError(f"unsupported fan-out at tensor {t}")
            reds = param_reds(u)
            if mpool_c:
                mu = mpool_c[0]
                moh, mow, _ = tens[mu.dst]
                out = VT(tsize(t), f"g{t}")
                refs = dict(out=W(out), aux0=AMAX(mu.dst), **reds)
                if t in dphi_off:
                    refs["dphi"] = P(dphi_off[t])
                backward.append(SymOp(nv.OP_MAXPOOL_BWD, dict(n_img=n, OH=moh, OW=mow, N=c), refs,
                                      [pool_seg(mu, W(gp[mu.dst]))]))
                gp[t] = out
            elif pool_c:
                pu = pool_c[0]
                out = VT(tsize(t), f"g{t}")
                refs = dict(out=W(out), **reds)
                if t in dphi_off:
                    refs["dphi"] = P(dphi_off[t])
                backward.append(SymOp(nv.OP_POOL_BWD, dict(n_img=n, OH=h, OW=w, N=c, fscale=1.0 / (h * w)), refs,
                                      [dict(a=W(gp[pu.dst]), b=NONE)]))
                gp[t] = out
            elif not conv_c:
                # only a residual consumer and no activation of its own: the cotangent is an alias
                if t in dphi_off:
                    raise NotImplementedError("activated tensor consumed only as a residual")
                gp[t] = gp[res_c[0].dst]
                if reds:
                    backward.append(SymOp(nv.OP_REDUCE, dict(n_img=n, OH=h, OW=w, N=c), dict(reds),
                                          [dict(a=W(gp[t]), b=NONE)]))
            else:
                out = VT(tsize(t), f"g{t}")
                segs = []
                for cu in conv_c:
                    coh, cow, cc = tens[cu.dst]
                    segs.append(_seg(W(gp[cu.dst]), CS(wt_off[cu.dst]), coh, cow, cu.cout, cu.kh, cu.kw, cu.stride,
                                     cu.pad_h, cu.pad_w, 1))
                refs = dict(out=W(out), **reds)
                if res_c:
                    refs["res"] = W(gp[res_c[0].dst])
                if t in dphi_off:
                    refs["dphi"] = P(dphi_off[t])
                backward.append(SymOp(nv.OP_IGEMM, dict(n_img=n, OH=h, OW=w, N=c), refs, segs))
                gp[t] = out
        else:
            reds = param_reds(u)
            if reds:
                backward.append(SymOp(nv.OP_REDUCE, dict(n_img=n, OH=h, OW=w, N=c), dict(reds),
                                      [dict(a=W(gp[t]), b=NONE)]))
        if u.kind == "conv":
            g = geom(u)
            refs = dict(out=YO(poff(u.kernel)))
            if u.bn_scale is not None:
                refs["scale"] = CS(s_off[u.dst])
            backward.append(SymOp(nv.OP_WGRAD, dict(n_img=n, OH=h, OW=w, N=c, ksplit=0, M=u.kh * u.kw * u.cin), refs,
                                  [_seg(P(a_off[u.src]), W(gp[t]), g["IH"], g["IW"], g["Cc"], g["KH"], g["KW"],
                                        g["stride"], g["pad_h"], g["pad_w"], 0)]))

    # ------------------------------------------------------------------ workspace placement
    work_pp = 0
    for tape in (tangent, backward):
        work_pp = max(work_pp, _place(tape, g_head))

    tapes = [[_lower(op) for op in tape] for tape in (primal, tangent, backward)]
    return CompiledNet(net=net, n=n, D=D, K=K, classifier=classifier, offsets=layout, prim_floats=max(prim, 4),
                       const_floats=cst, work_pp=work_pp, tapes=tapes, a_off=a_off, const_plan=const_plan,
                       prob_off=prob_off, input_off=a_off[0],
                       meta=dict(dphi_off=dphi_off, xhat_off=xhat_off, s_off=s_off, rstd_off=rstd_off, wt_off=wt_off, amax_off=amax_off,
                                 sqrtp_off=sqrtp_off, layout=layout))


def _has_grad(net: NetSpec, t: int) -> bool:
    """Does tensor t depend on theta (i.e. is it downstream of a parameterised unit)?"""
    if t == 0:
        return False
    for u in net.units:
        if u.dst == t:
            if u.kind == "conv":
                return True
            return _has_grad(net, u.src)
    return False


def _place(tape: List[SymOp], reserved: VT) -> int:
    """Liveness-based placement of the virtual workspace tensors of one tape; returns floats/probe."""
    for op in tape:
        for vt in op.work_tensors():
            if vt is not reserved:
                vt.off = None
    last: Dict[int, int] = {}
    for i, op in enumerate(tape):
        for vt in op.work_tensors():
            last[id(vt)] = i
    reserved.off = 0
    top = reserved.size
    free: Dict[int, List[int]] = {}
    for i, op in enumerate(tape):
        touched = list(op.work_tensors())
        for vt in touched:
            if vt.off is None:
                lst = free.get(vt.size)
                if lst:
                    vt.off = lst.pop()
                else:
                    vt.off = top
                    top += vt.size
        # lowering happens later; remember placement per op now (offsets may be reused afterwards)
        op._placed = {id(vt): vt.off for vt in touched}            # type: ignore[attr-defined]
        seen = set()
        for vt in touched:
            if vt is reserved or id(vt) in seen:
                continue
            seen.add(id(vt))
            if last[id(vt)] == i:
                free.setdefault(vt.size, []).append(vt.off)
    return top


def _mkref(op: SymOp, r) -> nv.Ref:
    if r[0] == "work":
        return nv.Ref(nv.SP_WORK, 0, op._placed[id(r[1])], r[2])       # type: ignore[attr-defined]
    return nv.Ref(int(r[0]), 0, int(r[1]), int(r[2]))


def _lower(op: SymOp) -> nv.Op:
    o = nv.Op()
    o.kind = op.kind
    o.nseg = len(op.segs)
    for name in nv.REF_FIELDS:
        setattr(o, name, _mkref(op, op.refs.get(name, NONE)))
    for i in range(3):
        if i < len(op.segs):
            s = op.segs[i]
            o.seg[i].a = _mkref(op, s["a"])
            o.seg[i].b = _mkref(op, s.get("b", NONE))
            for k, fld in (("IH", "IH"), ("IW", "IW"), ("C", "C"), ("KH", "KH"), ("KW", "KW"), ("stride", "stride"),
                           ("pad_h", "pad_h"), ("pad_w", "pad_w"), ("mode", "mode"), ("flags", "flags")):
                setattr(o.seg[i], fld, int(s.get(k, 0)))
        else:
            o.seg[i].a = _mkref(op, NONE)
            o.seg[i].b = _mkref(op, NONE)
    for k, v in op.f.items():
        setattr(o, k, v)
    return o


def build_consts(cn: CompiledNet, params: Dict[str, Any], batch_stats: Dict[str, Any], device, dtype=torch.float32):
    """Per-binding constants: BN factors s = gamma * rsqrt(var + eps), mean, rsqrt; and the transposed,
    BN-scaled kernels  Wt[kh][kw][co][ci] = W[kh][kw][ci][co] * s[co]  the data-gradient layers read."""
    out = torch.zeros(cn.const_floats, device=device, dtype=dtype)
    svals: Dict[int, torch.Tensor] = {}
    for item in cn.const_plan:
        if item[0] == "bn":
            _, u, so, mo, ro = item
            var = _get(batch_stats, u.bn_var).to(device=device, dtype=dtype)
            mean = _get(batch_stats, u.bn_mean).to(device=device, dtype=dtype)
            gamma = _get(params, u.bn_scale).to(device=device, dtype=dtype)
            rstd = torch.rsqrt(var + u.bn_eps)
            s = gamma * rstd
            svals[u.dst] = s
            out[so:so + u.cout] = s
            out[mo:mo + u.cout] = mean
            out[ro:ro + u.cout] = rstd
    for item in cn.const_plan:
        if item[0] == "wt":
            _, u, wo, so = item
            Wk = _get(params, u.kernel).to(device=device, dtype=dtype).reshape(u.kh, u.kw, u.cin, u.cout)
            if so is not None:
                Wk = Wk * svals[u.dst]
            out[wo:wo + Wk.numel()] = Wk.permute(0, 1, 3, 2).reshape(-1)
    return out


def set_precision(mode: str = "f32") -> None:
    """Process-wide arithmetic of the MFMA kernels: ``"f32"`` — exact f32 MFMA (default; bit-for-bit an fmaf
    chain) — or ``"bf16x3"`` — split-precision operands (x = hi + lo in bf16, three bf16 MFMAs per product, f32
    accumulation; ~1e-5 relative error per product, measured 6e-6 on a GGN-vp) at ~5x fewer matrix-pipe cycles."""
    modes = {"f32": 0, "bf16x3": 1}
    if mode not in modes:
        raise ValueError(f"precision must be one of {sorted(modes)}")
    nv.check(nv.load().lip_set_precision(modes[mode]), "lip_set_precision")


def get_precision() -> str:
    return {0: "f32", 1: "bf16x3"}[nv.load().lip_get_precision()]


def tape_flops_per_probe(cn: CompiledNet):
    """ALGORITHMIC FLOPs of one probe's tangent-forward + backward sweep, per op kind, from the tapes:
    conv segment 2 R N Ktot; transposed (data-gradient) segment 2 x the MACs of the conv it differentiates
    (the gather also visits stride-masked taps — those are not counted); WGRAD 2 R N M.
    Totals 8 MACs_fwd per example minus the input layer's two absent terms."""
    out = {}
    for which in (1, 2):
        for op in cn.tapes[which]:
            R = op.n_img * op.OH * op.OW
            if op.kind == nv.OP_IGEMM:
                fl = 0
                for i in range(op.nseg):
                    sg = op.seg[i]
                    if sg.mode == 0:
                        fl += 2 * R * op.N * sg.KH * sg.KW * sg.C
                    else:
                        fl += 2 * op.n_img * sg.IH * sg.IW * sg.C * sg.KH * sg.KW * op.N
                out[nv.OP_IGEMM] = out.get(nv.OP_IGEMM, 0) + fl
            elif op.kind == nv.OP_WGRAD:
                out[nv.OP_WGRAD] = out.get(nv.OP_WGRAD, 0) + 2 * R * op.N * op.M
    return out


def tape_executed_flops_per_probe(cn: CompiledNet):
    """Matrix-pipe FLOPs the kernels EXECUTE for one probe's sweep when the Winograd route is on: the ops that
    csrc/lip_mfma.hip sends to igemm_wino_kernel / wgrad_wino_kernel (3x3, stride 1, pad 1, channel counts multiples
    of 32, even maps) multiply 16 times per 2x2 output patch and (c, n)
    instead of 36 — 4/9 of the algorithmic count (padded tile blocks not counted); every other op executes its
    algorithmic FLOPs.  The utilisation of the f32 MFMA pipe follows from THIS count; `tape_flops_per_probe` stays the
    algorithmic figure the throughput is quoted in."""
    out = {}
    for which in (1, 2):
        for op in cn.tapes[which]:
            R = op.n_img * op.OH * op.OW
            even = op.OH % 2 == 0 and op.OW % 2 == 0
            if op.kind == nv.OP_IGEMM:
                segs = [op.seg[i] for i in range(op.nseg)]
                wino = (even and op.N % 32 == 0 and op.N >= 32 and
                        all(sg.KH == 3 and sg.KW == 3 and sg.stride == 1 and sg.pad_h == 1 and sg.pad_w == 1 and
                            sg.C % 32 == 0 and sg.IH == op.OH and sg.IW == op.OW for sg in segs))
                fl = 0
                for sg in segs:
                    if sg.mode == 0:
                        fl += 2 * R * op.N * sg.KH * sg.KW * sg.C
                    else:
                        fl += 2 * op.n_img * sg.IH * sg.IW * sg.C * sg.KH * sg.KW * op.N
                out[nv.OP_IGEMM] = out.get(nv.OP_IGEMM, 0) + (fl * 4 // 9 if wino else fl)
            elif op.kind == nv.OP_WGRAD:
                sg = op.seg[0]
                wino = (even and op.N % 32 == 0 and sg.C % 32 == 0 and sg.KH == 3 and sg.KW == 3 and
                        sg.stride == 1 and sg.pad_h == 1 and sg.pad_w == 1 and sg.IH == op.OH and sg.IW == op.OW)
                fl = 2 * R * op.N * op.M
                out[nv.OP_WGRAD] = out.get(nv.OP_WGRAD, 0) + (fl * 4 // 9 if wino else fl)
    return out


class LinearizedNet:
    """The linearised-network operator bound to (network, theta_MAP, data slice Z) on one GPU.

    Block operators on (P, D) row-major float32 device tensors:
      ``ggn_vp(V, scale, alpha)`` -> (P, D)      Y = scale * sum_i J_i^T H_i J_i V + alpha V
      ``jvp(V, mode, c)``         -> (P, n, K)   c * L^T J V   ('lt')   or   J V   ('raw')
      ``vjp(U, mode, c)``         -> (P, D)      J^T (c * L U) ('l')    or   J^T U ('raw')
    """

    def __init__(self, state, Z: torch.Tensor, model_type: Optional[str] = None, device=None,
                 workspace_bytes: int = 8 << 30, max_chunk: int = 1024, work: Optional[torch.Tensor] = None):
        net = getattr(state, "net", None)
        if net is None:
            raise TypeError("the HIP engine needs state.net (a NetSpec layer program); an opaque apply_fn cannot "
                            "be differentiated by hand-written kernels, and there is no CPU fallback")
        if Z.shape[0] == 0:
            raise ValueError("empty data block: the engine needs at least one example (an empty sum is the zero operator)")
        self.lib = nv.load()
        if not torch.cuda.is_available():
            raise nv.NativeError("no GPU: the product path has no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda")
        if model_type is not None:
            net.model_type = model_type
        self.n = int(Z.shape[0])
        self.cn = compile_net(net, self.n, state.params)
        cn = self.cn
        self.D, self.K = cn.D, cn.K
        f32 = dict(device=self.device, dtype=torch.float32)
        from .utils import flatten_nn_params
        flat, _ = flatten_nn_params(state.params)
        self.theta = flat.detach().to(**f32).contiguous()
        self.consts = build_consts(cn, state.params, state.batch_stats, self.device)
        self.prim = torch.zeros(cn.prim_floats, **f32)
        nin = self.n * net.tensors[0][0] * net.tensors[0][1] * net.tensors[0][2]
        self.prim[cn.input_off:cn.input_off + nin] = net.prepare_input(Z.detach().to(**f32)).reshape(-1)
        if work is not None:
            # a probe workspace shared with other engines on the same stream (ExampleChunkedGGN: the chunks of a large
            # data set run one after the other, so one workspace serves them all)
            if not (work.is_cuda and work.dtype == torch.float32 and work.is_contiguous()) or work.numel() < cn.work_pp:
                raise ValueError("shared workspace must be a contiguous float32 device tensor of >= work_pp floats")
            chunk = int(min(max_chunk, work.numel() // cn.work_pp))
            self.work = work
        else:
            chunk = int(max(1, min(max_chunk, workspace_bytes // (4 * cn.work_pp))))
            self.work = torch.empty(cn.work_pp * chunk, **f32)
        self.chunk = chunk
        h = C.c_void_p()
        nv.check(self.lib.lip_engine_create(C.byref(h), cn.D, self.n, cn.K), "lip_engine_create")
        self.h = h
        self._tapes = []
        for which, tape in enumerate(cn.tapes):
            arr = (nv.Op * len(tape))(*tape)
            self._tapes.append(arr)
            nv.check(self.lib.lip_engine_set_tape(h, which, arr, len(tape)), "lip_engine_set_tape")
        nv.check(self.lib.lip_engine_bind(h, nv.ptr(self.theta), nv.ptr(self.consts), nv.ptr(self.prim),
                                          nv.ptr(self.work), cn.work_pp, chunk), "lip_engine_bind")
        nv.check(self.lib.lip_engine_primal(h, nv.stream_ptr()), "lip_engine_primal")

    def __del__(self):
        h = getattr(self, "h", None)
        if h is not None and h.value:
            try:
                self.lib.lip_engine_destroy(h)
            except Exception:
                pass
            self.h = None

    # -------------------------------------------------------------------------------- helpers
    def _block(self, V: torch.Tensor, width: int) -> torch.Tensor:
        if V.dim() == 1:
            V = V[None]
        if V.shape[-1] != width:
            raise ValueError(f"expected trailing dimension {width}, got {tuple(V.shape)}")
        return V.reshape(-1, width).to(device=self.device, dtype=torch.float32).contiguous()

    def outputs(self) -> torch.Tensor:
        """primal network outputs f(z_i; theta) (n, K)"""
        o = self.cn.a_off[self.cn.net.out]
        return self.prim[o:o + self.n * self.K].reshape(self.n, self.K).clone()

    def probs(self) -> torch.Tensor:
        o = self.cn.prob_off
        return self.prim[o:o + self.n * self.K].reshape(self.n, self.K).clone()

    # ------------------------------------------------------------------------------ measurement
    def profile(self, enable: bool):
        nv.check(self.lib.lip_engine_profile(self.h, int(enable)), "lip_engine_profile")

    def profile_read(self):
        """{op kind: (ms, launches)} accumulated since the last read (HIP events on the launch stream)."""
        nk = 16
        ms = (C.c_double * nk)()
        cnt = (C.c_int64 * nk)()
        nv.check(self.lib.lip_engine_profile_read(self.h, ms, cnt, nk), "lip_engine_profile_read")
        return {k: (ms[k], cnt[k]) for k in range(nk) if cnt[k]}

    def flops_per_probe(self):
        return tape_flops_per_probe(self.cn)

    def executed_flops_per_probe(self):
        """matrix-pipe FLOPs actually issued per probe (Winograd ops at 4/9 of their algorithmic count) when the route
        is on (``lip_get_winograd() != 0``), else the algorithmic count"""
        if self.lib.lip_get_winograd() == 0 or self.lib.lip_get_precision() != 0:
            return tape_flops_per_probe(self.cn)
        return tape_executed_flops_per_probe(self.cn)

    # ------------------------------------------------------------------------------ operators
    def ggn_vp(self, V: torch.Tensor, scale: float = 1.0, alpha: float = 0.0, out: Optional[torch.Tensor] = None):
        Vb = self._block(V, self.D)
        Y = out if out is not None else torch.empty_like(Vb)
        nv.check(self.lib.lip_ggn_vp(self.h, nv.ptr(Vb), nv.ptr(Y), Vb.shape[0], float(scale), float(alpha),
                                     nv.stream_ptr()), "lip_ggn_vp")
        return Y

    def run_op(self, op: "nv.Op", P: int, V: Optional[torch.Tensor] = None, Y: Optional[torch.Tensor] = None,
               H: Optional[torch.Tensor] = None, head_mode: int = 0, head_c: float = 1.0) -> None:
        """Launch ONE host-built op against this binding (``lip_engine_run_op``): operands in the THETA / CONST / PRIM /
        WORK spaces resolve to the engine's buffers, VIN / YOUT / HEAD to the blocks passed here."""
        nv.check(self.lib.lip_engine_run_op(self.h, C.byref(op), nv.ptr(V), nv.ptr(Y), nv.ptr(H), int(P), int(head_mode),
                                            float(head_c), nv.stream_ptr()), "lip_engine_run_op")

    def jvp(self, V: torch.Tensor, mode: str = "raw", c: float = 1.0) -> torch.Tensor:
        Vb = self._block(V, self.D)
        U = torch.empty(Vb.shape[0], self.n, self.K, device=self.device, dtype=torch.float32)
        m = nv.HEAD_LT if mode == "lt" else nv.HEAD_OUT
        nv.check(self.lib.lip_jvp(self.h, nv.ptr(Vb), nv.ptr(U), Vb.shape[0], m, float(c), nv.stream_ptr()), "lip_jvp")
        return U

    def vjp(self, U: torch.Tensor, mode: str = "raw", c: float = 1.0) -> torch.Tensor:
        Ub = U.reshape(-1, self.n * self.K).to(device=self.device, dtype=torch.float32).contiguous()
        Y = torch.empty(Ub.shape[0], self.D, device=self.device, dtype=torch.float32)
        m = nv.HEAD_L if mode == "l" else nv.HEAD_IN
        nv.check(self.lib.lip_vjp(self.h, nv.ptr(Ub), nv.ptr(Y), Ub.shape[0], m, float(c), nv.stream_ptr()), "lip_vjp")
        return Y

    def vjp_rows(self, U: torch.Tensor, mode: str = "raw", c: float = 1.0) -> torch.Tensor:
        """Per-example rows of :meth:`vjp`: ``out[p, i] = J_i^T (c L_i U[p, i])`` -> (P, n, D).  Nothing is summed
        over examples, so the cost is that of ONE backward sweep per probe for all n rows."""
        Ub = U.reshape(-1, self.n * self.K).to(device=self.device, dtype=torch.float32).contiguous()
        Y = torch.empty(Ub.shape[0], self.n, self.D, device=self.device, dtype=torch.float32)
        m = nv.HEAD_L if mode == "l" else nv.HEAD_IN
        nv.check(self.lib.lip_vjp_rows(self.h, nv.ptr(Ub), nv.ptr(Y), Ub.shape[0], m, float(c), nv.stream_ptr()),
                 "lip_vjp_rows")
        return Y
