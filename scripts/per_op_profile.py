"""Per-op timing of the CIFAR ResNet1M sweep (or, with argument `resnet50`, of the full-resolution ResNet-50 slice) (each IGEMM / WGRAD op of the tangent and backward tapes timed alone
with events through lip_debug_run_ops): which layer shapes sit furthest from the MFMA roof."""
import sys, torch
sys.path.insert(0, '.')
import lip_amd
from lip_amd import _native as nv, krylov
from lip_amd.engine import LinearizedNet
from lip_amd.scalemodels import ResNet1M
from lip_amd.toymodels import create_state

if len(sys.argv) > 1 and sys.argv[1] == "resnet50":
    from lip_amd.scalemodels import ResNet50
    n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    net = ResNet50(1000); st = create_state(net, seed=1, dtype=torch.float32)
    eng = LinearizedNet(st, torch.rand(n_img, 224, 224, 3).cuda(), "classifier", workspace_bytes=64 << 30, max_chunk=P)
elif len(sys.argv) > 1 and sys.argv[1] == "mlp":          # BASELINE configs[2]: MNIST-MLP 784-1024-512-256-128-10, n = 50, P = 64
    from lip_amd.scalemodels import LargeClassifier
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    net = LargeClassifier((28, 28, 1), [1024, 512, 256, 128], 4, 10); st = create_state(net, seed=12345, dtype=torch.float32)
    eng = LinearizedNet(st, torch.rand(50, 28, 28, 1).cuda(), "classifier", workspace_bytes=2 << 30, max_chunk=P)
else:
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    net = ResNet1M(10); st = create_state(net, seed=1, dtype=torch.float32)
    eng = LinearizedNet(st, torch.rand(50, 32, 32, 3).cuda(), "classifier", workspace_bytes=24 << 30, max_chunk=P)
V = krylov.fill_rademacher(P, eng.D, 1, "cuda")
Y = torch.zeros(P, eng.D, device="cuda"); H = torch.zeros(P, eng.n * eng.K, device="cuda")
eng.ggn_vp(V, 1.0, 0.0); torch.cuda.synchronize()
rows = []
for which in (1, 2):
    for i, op in enumerate(eng.cn.tapes[which]):
        R = op.n_img * op.OH * op.OW
        if op.kind not in (nv.OP_IGEMM, nv.OP_WGRAD):
            fl = 0; name = f"kind {op.kind} R={R} N={op.N}"          # non-GEMM ops: time only
        elif op.kind == nv.OP_IGEMM:
            fl = 0; desc = []
            for q in range(op.nseg):
                sg = op.seg[q]
                fl += 2 * R * op.N * sg.KH * sg.KW * sg.C if sg.mode == 0 else 2 * op.n_img * sg.IH * sg.IW * sg.C * sg.KH * sg.KW * op.N
                desc.append(f"m{sg.mode} {sg.KH}x{sg.KW}x{sg.C} s{sg.stride}")
            name = f"igemm R={R} N={op.N} [" + "; ".join(desc) + "]"
        else:
            fl = 2 * R * op.N * op.M
            name = f"wgrad R={R} N={op.N} M={op.M}"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        for r in range(reps + 1):
            if r == 1:
                e0.record()
            nv.check(eng.lib.lip_debug_run_ops(eng.h, which, i, 1, nv.ptr(V), nv.ptr(Y), nv.ptr(H), P, nv.HEAD_GGN, 1.0,
                                               nv.stream_ptr()), "debug_run_ops")
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rows.append((which, i, name, ms, fl * P / ms / 1e9))
tot = sum(r[3] for r in rows)
for which, i, name, ms, tf in rows:
    if tf == 0 and ms < 0.05 and not (len(sys.argv) > 1 and sys.argv[1] == "mlp"):
        continue
    print(f"t{which} op{i:3d} {ms:7.3f} ms {tf:6.1f} TF  {name}")
print("total ms", tot, " non-GEMM ms", sum(r[3] for r in rows if r[4] == 0))
# whole product (fused alpha where the weight gradients allow it)
for al in (0.0, 1e-3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.ggn_vp(V, 1.0, al)
    e0.record()
    for _ in range(10):
        eng.ggn_vp(V, 1.0, al)
    e1.record(); torch.cuda.synchronize()
    print(f"ggn_vp alpha={al}: {e0.elapsed_time(e1) / 10:.3f} ms per {P}-probe block")
