"""Detail of ONE backward-tape op under both precision modes at the bench geometry: where in Y the two modes (and two
runs of the same mode) differ.  Usage: python scripts/split_detail.py <op index> [P]"""
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
    target = int(sys.argv[1])
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda", 0)
    net = ResNet1M(10)
    state = create_state(net, seed=1231231234, dtype=torch.float32)
    Z = torch.rand(50, 32, 32, 3, generator=torch.Generator().manual_seed(280300))
    eng = LinearizedNet(state, Z.to(dev), "classifier", device=dev, workspace_bytes=8 << 30, max_chunk=P)
    V = krylov.fill_rademacher(P, eng.D, 1234, dev)
    Yd = torch.zeros(P, eng.D, device=dev)
    Hd = torch.zeros(P, eng.n * eng.K, device=dev)
    eng.work.zero_()

    def run(which, i):
        nv.check(eng.lib.lip_debug_run_ops(eng.h, which, i, 1, nv.ptr(V), nv.ptr(Yd), nv.ptr(Hd), P, nv.HEAD_GGN, 980.0,
                                           nv.stream_ptr()), "debug_run_ops")
        torch.cuda.synchronize()

    for i in range(len(eng.cn.tapes[1])):
        run(1, i)
    for i, op in enumerate(eng.cn.tapes[2]):
        if op.kind == nv.OP_HEAD:
            continue
        if i == target:
            break
        run(2, i)
    op = eng.cn.tapes[2][target]
    print("op", target, "kind", op.kind, "N", op.N, "red0", op.red0.space, op.red0.off, op.red0.pstride, "red1", op.red1.space,
          op.red1.off, op.red1.pstride)
    Yd.zero_()                      # only this op's contribution
    w0 = eng.work.clone()
    outs = {}
    for tag, mode in (("s1", "bf16x3"), ("s2", "bf16x3"), ("f1", "f32"), ("f2", "f32")):
        eng.work.copy_(w0)
        Yd.zero_()
        set_precision(mode)
        run(2, target)
        outs[tag] = Yd.clone()
    set_precision("f32")
    ref = outs["f1"]
    for a, b in (("s1", "f1"), ("s1", "s2"), ("f1", "f2")):
        d = (outs[a] - outs[b]).abs()
        nz = d.nonzero()
        print(f"{a} vs {b}: max abs diff {d.max().item():.4e} (max|ref| {ref.abs().max().item():.4e}), differing entries {nz.shape[0]}")
        if nz.shape[0]:
            flat = d.reshape(-1)
            top = torch.topk(flat, min(12, nz.shape[0]))
            for v, ix in zip(top.values.tolist(), top.indices.tolist()):
                p, j = divmod(ix, eng.D)
                print(f"   probe {p} param {j}: {a} {outs[a][p, j].item():.6e} {b} {outs[b][p, j].item():.6e} diff {v:.3e}")


if __name__ == "__main__":
    main()
