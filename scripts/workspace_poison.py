"""Read-before-write check of the probe workspace: every product is run with the workspace filled with NaN, with zeros and
with large finite garbage beforehand; the results must be finite and identical (up to the float atomics' run-to-run
rounding) — a kernel that reads a workspace element the same call has not written would show up here."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd
from lip_amd import krylov
from lip_amd.ggn import get_engine, shared_workspace
from lip_amd.scalemodels import ResNet1M, LargeClassifier
from lip_amd.toymodels import create_state
dev = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "cifar"
if which == "cifar":
    st = create_state(ResNet1M(10), seed=1231231234, dtype=torch.float32).to(device=dev, dtype=torch.float32)
    Z = torch.rand(50, 32, 32, 3, generator=torch.Generator().manual_seed(280300)).to(dev)
else:
    st = create_state(LargeClassifier((28, 28, 1), [1024, 512, 256, 128], 4, 10), seed=3, dtype=torch.float32).to(device=dev, dtype=torch.float32)
    Z = torch.rand(50, 28, 28, 1, generator=torch.Generator().manual_seed(1)).to(dev)
eng = get_engine(st, Z, "classifier")
W = shared_workspace(dev)
assert eng.work.data_ptr() == W.data_ptr()
for P in (1, 8, 16, 64, 256):
    V = krylov.fill_normal(P, eng.D, 7 + P, dev)
    U = torch.randn(P, eng.n, eng.K, device=dev)
    outs = {}
    for fill in ("zero", "nan", "big"):
        res = []
        for name, fn in (("ggn_vp", lambda: eng.ggn_vp(V, 980.0, 0.005)), ("jvp", lambda: eng.jvp(V, "lt", 1.0)),
                         ("vjp", lambda: eng.vjp(U, "l", 1.0)), ("vjp_rows", lambda: eng.vjp_rows(U[:min(P, 4)].contiguous(), "l", 1.0))):
            if fill == "zero": W.zero_()
            elif fill == "nan": W.fill_(float("nan"))
            else: W.fill_(1e30)
            res.append((name, fn().double()))
        outs[fill] = res
    for i, (name, ref) in enumerate(outs["zero"]):
        msg = []
        for fill in ("nan", "big"):
            o = outs[fill][i][1]
            fin = bool(torch.isfinite(o).all())
            d = float((o - ref).abs().max() / ref.abs().max()) if fin else float("nan")
            msg.append(f"{fill}: finite {fin} rel diff {d:.2e}")
        print(f"P={P:3d} {name:9s} " + "; ".join(msg))
