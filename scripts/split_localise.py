"""Localise a precision-mode discrepancy op by op at the bench geometry (ResNet1M 32x32, n = 50).

Every op of the tangent and backward tapes is launched through ``lip_debug_run_ops`` three times on the SAME input
state: twice with ``lip_set_precision(1)`` (bf16x3) and once in exact f32.  Printed per op: the relative difference of
the split-precision result to the f32 one (scale = max|f32| of the buffer) and whether the two split runs agree bit
for bit.  Usage: python scripts/split_localise.py [P] [n]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd  # noqa: E402,F401
from lip_amd import _native as nv  # noqa: E402
from lip_amd import krylov  # noqa: E402
from lip_amd.engine import LinearizedNet, set_precision  # noqa: E402
from lip_amd.scalemodels import ResNet1M  # noqa: E402
from lip_amd.toymodels import create_state  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda", 0)
    net = ResNet1M(10)
    state = create_state(net, seed=1231231234, dtype=torch.float32)
    Z = torch.rand(n, 32, 32, 3, generator=torch.Generator().manual_seed(280300))
    eng = LinearizedNet(state, Z.to(dev), "classifier", device=dev, workspace_bytes=8 << 30, max_chunk=P)
    assert eng.chunk == P
    V = krylov.fill_rademacher(P, eng.D, 1234, dev)
    scale, alpha = 49000 / n, 0.005

    # whole product first: f32 vs split (twice)
    Y32 = eng.ggn_vp(V, scale, alpha).clone()
    set_precision("bf16x3")
    Ya = eng.ggn_vp(V, scale, alpha).clone()
    Yb = eng.ggn_vp(V, scale, alpha).clone()
    set_precision("f32")
    Y32b = eng.ggn_vp(V, scale, alpha).clone()
    m = Y32.abs().max()
    print(f"whole product P={P}: split vs f32 {((Ya - Y32).abs().max() / m).item():.3e}, split run-to-run "
          f"{((Ya - Yb).abs().max() / m).item():.3e}, f32 run-to-run {((Y32 - Y32b).abs().max() / m).item():.3e}", flush=True)

    Yd = torch.zeros(P, eng.D, device=dev)
    Hd = torch.zeros(P, eng.n * eng.K, device=dev)
    eng.work.zero_()
    bufs = (("work", eng.work), ("Y", Yd), ("H", Hd))

    def run(which, i):
        nv.check(eng.lib.lip_debug_run_ops(eng.h, which, i, 1, nv.ptr(V), nv.ptr(Yd), nv.ptr(Hd), P, nv.HEAD_GGN, scale,
                                           nv.stream_ptr()), "debug_run_ops")
        torch.cuda.synchronize()

    for which in (1, 2):
        for i, op in enumerate(eng.cn.tapes[which]):
            if which == 2 and op.kind == nv.OP_HEAD:
                continue
            snap = [b.clone() for _, b in bufs]
            res = []
            for mode in ("bf16x3", "bf16x3", "f32"):
                for (_, b), s in zip(bufs, snap):
                    b.copy_(s)
                set_precision(mode)
                run(which, i)
                res.append([b.clone() for _, b in bufs])
            set_precision("f32")
            line = []
            for k, (name, _) in enumerate(bufs):
                ref = res[2][k]
                sc = ref.abs().max().item() + 1e-30
                d = ((res[0][k] - ref).abs().max() / sc).item()
                rr = ((res[0][k] - res[1][k]).abs().max() / sc).item()
                if d > 0 or rr > 0:
                    line.append(f"{name}: split-f32 {d:.2e} run-to-run {rr:.2e}")
            sg = op.seg[0]
            geo = f"N={op.N} OH={op.OH} OW={op.OW} nseg={op.nseg} C={sg.C} K={sg.KH}x{sg.KW} s={sg.stride} mode={sg.mode}"
            print(f"tape {which} op {i:3d} kind {op.kind} {geo}: " + ("; ".join(line) if line else "identical"), flush=True)
            del snap, res


if __name__ == "__main__":
    main()
