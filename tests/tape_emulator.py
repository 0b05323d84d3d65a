"""CPU (float64) emulator of the op-tape semantics of ``include/lip.h`` — TEST INFRASTRUCTURE.

It executes the *lowered* tapes (the exact ctypes structs the HIP engine receives) with plain
torch indexing, following the op definitions literally (gather formulas of ``lip_seg_t``).  Used
(a) on the CPU to validate the NetSpec -> tape compiler against the oracle and (b) on the GPU box
to localise a kernel bug op by op (``lip_debug_run_ops``).
"""
import torch

from lip_amd import _native as nv

F64 = torch.float64


class TapeMachine:
    def __init__(self, cn, theta, consts, Z, chunk):
        self.cn, self.chunk = cn, chunk
        self.theta = theta.to(F64).reshape(-1)
        self.consts = consts.to(F64).reshape(-1)
        self.prim = torch.zeros(cn.prim_floats, dtype=F64)
        zin = cn.net.prepare_input(Z.to(F64)).reshape(-1)
        self.prim[cn.input_off:cn.input_off + zin.numel()] = zin
        self.work = torch.zeros(cn.work_pp * chunk, dtype=F64)
        self.V = self.Y = self.H = None

    # ------------------------------------------------------------------ operand access
    def _buf(self, space):
        return {nv.SP_THETA: self.theta, nv.SP_CONST: self.consts, nv.SP_PRIM: self.prim, nv.SP_WORK: self.work,
                nv.SP_VIN: self.V, nv.SP_YOUT: self.Y, nv.SP_HEAD: self.H}[space]

    def view(self, ref, P, count):
        """(P, count) strided view of an operand (P = 1 rows for shared operands)."""
        if ref.space == nv.SP_NONE:
            return None
        buf = self._buf(ref.space)
        off = ref.off * (self.chunk if ref.space == nv.SP_WORK else 1)
        ps = ref.pstride
        return torch.as_strided(buf, (P, count), (ps, 1), off + buf.storage_offset())     # as_strided offsets are storage-absolute

    # ------------------------------------------------------------------ ops
    def _gather(self, seg, P, n, OH, OW):
        """im2col per lip_seg_t: returns A (Pa, n*OH*OW, KH*KW*C) with Pa in {1, P}."""
        Pa = P if seg.a.pstride != 0 else 1
        a = self.view(seg.a, Pa, n * seg.IH * seg.IW * seg.C).reshape(Pa, n, seg.IH, seg.IW, seg.C)
        cols = torch.zeros(Pa, n, OH, OW, seg.KH, seg.KW, seg.C, dtype=F64)
        for kh in range(seg.KH):
            for kw in range(seg.KW):
                for oh in range(OH):
                    if seg.mode == 0:
                        ih = oh * seg.stride + kh - seg.pad_h
                    else:
                        t = oh + seg.pad_h - kh
                        if t < 0 or t % seg.stride:
                            continue
                        ih = t // seg.stride
                    if ih < 0 or ih >= seg.IH:
                        continue
                    for ow in range(OW):
                        if seg.mode == 0:
                            iw = ow * seg.stride + kw - seg.pad_w
                        else:
                            t = ow + seg.pad_w - kw
                            if t < 0 or t % seg.stride:
                                continue
                            iw = t // seg.stride
                        if iw < 0 or iw >= seg.IW:
                            continue
                        cols[:, :, oh, ow, kh, kw, :] = a[:, :, ih, iw, :]
        return cols.reshape(Pa, n * OH * OW, seg.KH * seg.KW * seg.C)

    def igemm(self, op, P):
        n, OH, OW, N = op.n_img, op.OH, op.OW, op.N
        R = n * OH * OW
        acc = torch.zeros(P, R, N, dtype=F64)
        for s in range(op.nseg):
            seg = op.seg[s]
            A = self._gather(seg, P, n, OH, OW)
            Kt = seg.KH * seg.KW * seg.C
            Pb = P if seg.b.pstride != 0 else 1
            if seg.flags & nv.SEG_B_TRANS:          # B[(tap*C + c)][n] = b[(tap*N + n)*C + c]
                B = self.view(seg.b, Pb, Kt * N).reshape(Pb, seg.KH * seg.KW, N, seg.C).permute(0, 1, 3, 2).reshape(Pb, Kt, N)
            else:
                B = self.view(seg.b, Pb, Kt * N).reshape(Pb, Kt, N)
            acc = acc + torch.matmul(A, B)
        v = acc
        if op.scale.space != nv.SP_NONE:
            v = v * self.view(op.scale, 1, N).reshape(1, 1, N)
        if op.e0.space != nv.SP_NONE:
            Pe = P if op.e0.pstride != 0 else 1
            v = v + self.view(op.e0, Pe, N).reshape(Pe, 1, N)
        if op.e1.space != nv.SP_NONE:
            Pe = P if op.e1.pstride != 0 else 1
            v = v + self.view(op.e1, Pe, N).reshape(Pe, 1, N) * self.view(op.xhat, 1, R * N).reshape(1, R, N)
        if op.res.space != nv.SP_NONE:
            Pr = P if op.res.pstride != 0 else 1
            v = v + self.view(op.res, Pr, R * N).reshape(Pr, R, N)
        if op.dphi.space != nv.SP_NONE:
            v = v * self.view(op.dphi, 1, R * N).reshape(1, R, N)
        self.view(op.out, P, R * N).copy_(v.reshape(P, R * N))
        self._reds(op, v, P, R, N)

    def _reds(self, op, v, P, R, N):
        if op.red0.space != nv.SP_NONE:
            self.view(op.red0, P, N).add_(v.sum(1))
        if op.red1.space != nv.SP_NONE:
            xh = self.view(op.xhat2, 1, R * N).reshape(1, R, N)
            self.view(op.red1, P, N).add_((v * xh).sum(1))

    def wgrad(self, op, P):
        seg = op.seg[0]
        n, OH, OW, N = op.n_img, op.OH, op.OW, op.N
        R = n * OH * OW
        A = self._gather(seg, 1, n, OH, OW)[0]                      # (R, M)
        g = self.view(seg.b, P, R * N).reshape(P, R, N)
        dW = torch.matmul(A.T.unsqueeze(0), g)                      # (P, M, N)
        if op.scale.space != nv.SP_NONE:
            dW = dW * self.view(op.scale, 1, N).reshape(1, 1, N)
        M = seg.KH * seg.KW * seg.C
        self.view(op.out, P, M * N).add_(dW.reshape(P, M * N))

    def reduce(self, op, P):
        R, N = op.n_img * op.OH * op.OW, op.N
        g = self.view(op.seg[0].a, P, R * N).reshape(P, R, N)
        self._reds(op, g, P, R, N)

    def pool_fwd(self, op, P):
        n, HW, Cc = op.n_img, op.OH * op.OW, op.N
        Pa = P if op.seg[0].a.pstride != 0 else 1
        x = self.view(op.seg[0].a, Pa, n * HW * Cc).reshape(Pa, n, HW, Cc)
        self.view(op.out, Pa, n * Cc).copy_((x.sum(2) / HW).reshape(Pa, n * Cc))   # exact 1/HW (op.fscale is a C float)

    def pool_bwd(self, op, P):
        n, HW, Cc = op.n_img, op.OH * op.OW, op.N
        g = self.view(op.seg[0].a, P, n * Cc).reshape(P, n, 1, Cc)
        v = (g / HW).expand(P, n, HW, Cc).reshape(P, n * HW, Cc)
        if op.dphi.space != nv.SP_NONE:
            v = v * self.view(op.dphi, 1, n * HW * Cc).reshape(1, n * HW, Cc)
        self.view(op.out, P, n * HW * Cc).copy_(v.reshape(P, -1))
        self._reds(op, v, P, n * HW, Cc)

    def _maxpool_geom(self, op):
        g = op.seg[0]
        return op.n_img, g.IH, g.IW, op.OH, op.OW, op.N, g.KH, g.KW, g.stride, g.pad_h, g.pad_w

    def _avgpool(self, op, x, transpose=False):
        """window average (NONE argmax ref): x (B, n, IH, IW, C) -> (B, n, OH, OW, C), or its transpose."""
        n, IH, IW, OH, OW, Cc, KH, KW, st, ph, pw = self._maxpool_geom(op)
        B = x.shape[0]
        out = torch.zeros((B, n, IH, IW, Cc) if transpose else (B, n, OH, OW, Cc), dtype=F64)
        for kh in range(KH):
            for kw in range(KW):
                for oh in range(OH):
                    ih = oh * st + kh - ph
                    if ih < 0 or ih >= IH:
                        continue
                    for ow in range(OW):
                        iw = ow * st + kw - pw
                        if iw < 0 or iw >= IW:
                            continue
                        if transpose:
                            out[:, :, ih, iw, :] += x[:, :, oh, ow, :]
                        else:
                            out[:, :, oh, ow, :] += x[:, :, ih, iw, :]
        return out / (KH * KW)

    def maxpool_primal(self, op):
        n, IH, IW, OH, OW, Cc, KH, KW, st, ph, pw = self._maxpool_geom(op)
        x = self.view(op.seg[0].a, 1, n * IH * IW * Cc).reshape(n, IH, IW, Cc)
        if op.aux0.space == nv.SP_NONE:
            self.view(op.out, 1, n * OH * OW * Cc).copy_(self._avgpool(op, x[None]).reshape(1, -1))
            return
        out = torch.full((n, OH, OW, Cc), -3.0e38, dtype=F64)
        am = torch.full((n, OH, OW, Cc), -1.0, dtype=F64)
        for kh in range(KH):
            for kw in range(KW):
                for oh in range(OH):
                    ih = oh * st + kh - ph
                    if ih < 0 or ih >= IH:
                        continue
                    for ow in range(OW):
                        iw = ow * st + kw - pw
                        if iw < 0 or iw >= IW:
                            continue
                        v = x[:, ih, iw, :]
                        better = v > out[:, oh, ow, :]
                        out[:, oh, ow, :] = torch.where(better, v, out[:, oh, ow, :])
                        am[:, oh, ow, :] = torch.where(better, torch.full_like(v, float(ih * IW + iw)), am[:, oh, ow, :])
        self.view(op.out, 1, n * OH * OW * Cc).copy_(out.reshape(1, -1))
        self.view(op.aux0, 1, n * OH * OW * Cc).copy_(am.reshape(1, -1))

    def maxpool_fwd(self, op, P):
        n, IH, IW, OH, OW, Cc, KH, KW, st, ph, pw = self._maxpool_geom(op)
        if op.aux0.space == nv.SP_NONE:
            x = self.view(op.seg[0].a, P, n * IH * IW * Cc).reshape(P, n, IH, IW, Cc)
            self.view(op.out, P, n * OH * OW * Cc).copy_(self._avgpool(op, x).reshape(P, -1))
            return
        x = self.view(op.seg[0].a, P, n * IH * IW * Cc).reshape(P, n, IH * IW, Cc)
        am = self.view(op.aux0, 1, n * OH * OW * Cc).reshape(n, OH * OW, Cc).long()
        idx = am.clamp_min(0).unsqueeze(0).expand(P, -1, -1, -1)
        out = torch.gather(x, 2, idx) * (am >= 0).unsqueeze(0)
        self.view(op.out, P, n * OH * OW * Cc).copy_(out.reshape(P, -1))

    def maxpool_bwd(self, op, P):
        n, IH, IW, OH, OW, Cc, KH, KW, st, ph, pw = self._maxpool_geom(op)
        g = self.view(op.seg[0].a, P, n * OH * OW * Cc).reshape(P, n, OH * OW, Cc)
        if op.aux0.space == nv.SP_NONE:
            out = self._avgpool(op, g.reshape(P, n, OH, OW, Cc), transpose=True)
        else:
            am = self.view(op.aux0, 1, n * OH * OW * Cc).reshape(n, OH * OW, Cc).long()
            out = torch.zeros(P, n, IH * IW, Cc, dtype=F64)
            idx = am.clamp_min(0).unsqueeze(0).expand(P, -1, -1, -1)
            out.scatter_add_(2, idx, g * (am >= 0).unsqueeze(0))
        v = out.reshape(P, n * IH * IW, Cc)
        if op.dphi.space != nv.SP_NONE:
            v = v * self.view(op.dphi, 1, n * IH * IW * Cc).reshape(1, n * IH * IW, Cc)
        self.view(op.out, P, n * IH * IW * Cc).copy_(v.reshape(P, -1))
        self._reds(op, v, P, n * IH * IW, Cc)

    def primal_post(self, op):
        R, N = op.n_img * op.OH * op.OW, op.N
        y = self.view(op.seg[0].a, 1, R * N).reshape(R, N).clone()
        if op.e0.space != nv.SP_NONE:
            y = y + self.view(op.e0, 1, N)
        if op.e1.space != nv.SP_NONE:
            xh = (y - self.view(op.aux0, 1, N)) * self.view(op.aux1, 1, N)
            self.view(op.out3, 1, R * N).copy_(xh.reshape(1, -1))
            y = xh * self.view(op.e1, 1, N) + self.view(op.scale, 1, N)
        if op.res.space != nv.SP_NONE:
            y = y + self.view(op.res, 1, R * N).reshape(R, N)
        y = y.clone().requires_grad_(True)
        from lip_amd.netspec import act_fn
        name = {0: "none", 1: "relu", 2: "tanh", 3: "gelu"}[op.act]
        a = act_fn(name, y)
        d, = torch.autograd.grad(a.sum(), y)
        self.view(op.out, 1, R * N).copy_(a.detach().reshape(1, -1))
        if op.out2.space != nv.SP_NONE:
            self.view(op.out2, 1, R * N).copy_(d.reshape(1, -1))

    def softmax(self, op):
        n, K = op.n_img, op.N
        f = self.view(op.seg[0].a, 1, n * K).reshape(n, K)
        p = torch.softmax(f, -1)
        self.view(op.out, 1, n * K).copy_(p.reshape(1, -1))
        self.view(op.out2, 1, n * K).copy_(p.sqrt().reshape(1, -1))

    def head(self, op, P, mode, c):
        n, K = op.n_img, op.N
        if mode == nv.HEAD_GGN:
            rin, rout = op.seg[0].a, op.out
        elif mode in (nv.HEAD_LT, nv.HEAD_OUT):
            rin, rout = op.seg[0].a, op.out2
        else:
            rin, rout = op.out2, op.out
        u = self.view(rin, P, n * K).reshape(P, n, K)
        if (not op.classifier) or mode in (nv.HEAD_OUT, nv.HEAD_IN):
            v = c * u
        else:
            p = self.view(op.aux0, 1, n * K).reshape(1, n, K)
            s = self.view(op.aux1, 1, n * K).reshape(1, n, K)
            if mode == nv.HEAD_GGN:
                v = c * p * (u - (p * u).sum(-1, keepdim=True))
            elif mode == nv.HEAD_LT:
                v = c * s * (u - (p * u).sum(-1, keepdim=True))
            else:
                v = c * (s * u - (s * u).sum(-1, keepdim=True) * p)
        self.view(rout, P, n * K).copy_(v.reshape(P, n * K))

    def run_op(self, op, P, mode=0, c=1.0):
        k = op.kind
        if k == nv.OP_IGEMM:
            self.igemm(op, P)
        elif k == nv.OP_WGRAD:
            self.wgrad(op, P)
        elif k == nv.OP_REDUCE:
            self.reduce(op, P)
        elif k == nv.OP_POOL_FWD:
            self.pool_fwd(op, P)
        elif k == nv.OP_POOL_BWD:
            self.pool_bwd(op, P)
        elif k == nv.OP_PRIMAL_POST:
            self.primal_post(op)
        elif k == nv.OP_SOFTMAX:
            self.softmax(op)
        elif k == nv.OP_HEAD:
            self.head(op, P, mode, c)
        elif k == nv.OP_MAXPOOL_PRIMAL:
            self.maxpool_primal(op)
        elif k == nv.OP_MAXPOOL_FWD:
            self.maxpool_fwd(op, P)
        elif k == nv.OP_MAXPOOL_BWD:
            self.maxpool_bwd(op, P)
        else:
            raise ValueError(k)

    # ------------------------------------------------------------------ entry points (mirror lip.h)
    def primal(self):
        for op in self.cn.tapes[0]:
            self.run_op(op, 1)

    def ggn_vp(self, V, scale, alpha):
        P = V.shape[0]
        assert P <= self.chunk
        self.V = V.to(F64).contiguous().reshape(-1)
        Y = alpha * V.to(F64)
        self.Y = Y.contiguous().reshape(-1)
        for op in self.cn.tapes[1]:
            self.run_op(op, P, nv.HEAD_GGN, scale)
        for op in self.cn.tapes[2]:
            if op.kind != nv.OP_HEAD:
                self.run_op(op, P)
        return self.Y.reshape(P, -1)

    def jvp(self, V, mode, c):
        P = V.shape[0]
        self.V = V.to(F64).contiguous().reshape(-1)
        self.H = torch.zeros(P * self.cn.n * self.cn.K, dtype=F64)
        for op in self.cn.tapes[1]:
            self.run_op(op, P, mode, c)
        return self.H.reshape(P, self.cn.n, self.cn.K)

    def vjp(self, U, mode, c):
        P = U.shape[0]
        self.H = U.to(F64).contiguous().reshape(-1)
        self.Y = torch.zeros(P * self.cn.D, dtype=F64)
        for op in self.cn.tapes[2]:
            self.run_op(op, P, mode, c)
        return self.Y.reshape(P, -1)
