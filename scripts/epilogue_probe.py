"""What does the fused epilogue of the 32-column implicit GEMM cost?  The same 3x3x32 -> 32 convolution over
R = 51200 rows x 256 probes (CIFAR stage 1) launched through lip_engine_run_op with 1 or 2 K-segments and with /
without epilogue operands."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd  # noqa
from lip_amd import _native as nv
from lip_amd.engine import LinearizedNet
from lip_amd.scalemodels import ResNet1M
from lip_amd.second_order import _igemm, _ref, NONE
from lip_amd.toymodels import create_state

P, n = 256, 50
net = ResNet1M(10)
st = create_state(net, seed=1, dtype=torch.float32)
eng = LinearizedNet(st, torch.rand(n, 32, 32, 3).cuda(), "classifier", workspace_bytes=12 << 30, max_chunk=P)
sz = 32 * 32 * 32
arena = torch.randn(3 * P * n * sz + 1024, device="cuda") * 0.1
V = torch.randn(P, eng.D, device="cuda") * 0.01
u = [x for x in net.units if x.kind == "conv"][2]            # a 3x3x32 -> 32 stride-1 unit
lay = eng.cn.meta["layout"]
geom = dict(IH=32, IW=32, C=32, KH=3, KW=3, stride=1, pad_h=1, pad_w=1, mode=0, flags=0)
Y = lambda slot, ps: _ref(nv.SP_YOUT, slot * P * n * sz, ps)
def op(nseg, res=False, dphi=False, xhat=False):
    o = _igemm(n, 32, 32, 32, dict(geom, a=Y(0, n * sz), b=_ref(nv.SP_THETA, lay[u.kernel][0])), Y(1, n * sz))
    if nseg == 2:
        o.nseg = 2
        s = o.seg[1]
        s.a, s.b = _ref(nv.SP_PRIM, eng.cn.a_off[u.src]), _ref(nv.SP_VIN, lay[u.kernel][0], eng.D)
        for k, v in geom.items():
            setattr(s, k, v)
    if res: o.res = Y(2, n * sz)
    if dphi: o.dphi = _ref(nv.SP_PRIM, eng.cn.meta["dphi_off"][u.dst])
    if xhat:
        o.xhat = _ref(nv.SP_PRIM, eng.cn.meta["xhat_off"][u.dst]); o.e1 = _ref(nv.SP_VIN, lay[u.bn_scale][0], eng.D)
        o.e0 = _ref(nv.SP_VIN, lay[u.bn_bias][0], eng.D); o.scale = _ref(nv.SP_CONST, eng.cn.meta["s_off"][u.dst])
    return o
def timeit(o, reps=5):
    eng.run_op(o, P, V=V, Y=arena); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.run_op(o, P, V=V, Y=arena)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
fl1 = 2.0 * n * 1024 * 32 * 288 * P
for name, o, fl in (("1 seg, plain store", op(1), fl1), ("1 seg + res + dphi", op(1, True, True), fl1),
                    ("2 seg, plain store", op(2), 2 * fl1), ("2 seg + res + dphi", op(2, True, True), 2 * fl1),
                    ("2 seg + res + dphi + BN tangent (the tangent layer)", op(2, True, True, True), 2 * fl1)):
    ms = timeit(o)
    print(f"{name:55s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF")

# which operand makes the per-probe-A segment slow?  one segment each:
def op1(a_ref, b_ref):
    return _igemm(n, 32, 32, 32, dict(geom, a=a_ref, b=b_ref), Y(1, n * sz))
W_shared, W_probe = _ref(nv.SP_THETA, lay[u.kernel][0]), _ref(nv.SP_VIN, lay[u.kernel][0], eng.D)
A_probe, A_same, A_prim = Y(0, n * sz), Y(0, 0), _ref(nv.SP_PRIM, eng.cn.a_off[u.src])
for name, o in (("A per probe (workspace), B shared", op1(A_probe, W_shared)),
                ("A one tensor for all probes (workspace, pstride 0), B shared", op1(A_same, W_shared)),
                ("A primal (shared), B shared", op1(A_prim, W_shared)),
                ("A primal (shared), B per probe", op1(A_prim, W_probe)),
                ("A per probe, B per probe", op1(A_probe, W_probe))):
    ms = timeit(o)
    print(f"{name:62s} {ms:7.3f} ms  {fl1 / ms / 1e9:6.1f} TF")
