"""HBM roofline of the Krylov / trace primitives at the CIFAR config (D = 1 084 586): algorithmic bytes / time."""
import sys, time, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lip_amd
from lip_amd import krylov, _native as nv

def timeit(f, n=20):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

def run(D=1084586, P=256, k=36):
    lib = nv.load(); st = nv.stream_ptr()
    X = krylov.fill_normal(P, D, 1); Y = krylov.fill_normal(P, D, 2)
    out = {}
    t = timeit(lambda: krylov.bdot(X, Y)); out["bdot"] = dict(ms=t*1e3, GBps=8*D*P/t/1e9)
    t = timeit(lambda: krylov.axpby(Y, X, None, 0.5, None, 1.0)); out["axpby"] = dict(ms=t*1e3, GBps=12*D*P/t/1e9)
    t = timeit(lambda: krylov.fill_rademacher(P, D, 3)); out["fill_rademacher"] = dict(ms=t*1e3, GBps=4*D*P/t/1e9)
    t = timeit(lambda: krylov.fill_normal(P, D, 3)); out["fill_normal"] = dict(ms=t*1e3, GBps=4*D*P/t/1e9)
    # CG step: cg_update (read x,r,p,Ap ; write x,r) + cg_direction (read p,r ; write p) + dot(p,Ap) = 11 passes = 44 D bytes
    Pc = 64
    x, r, p, Ap = (krylov.fill_normal(Pc, D, s) for s in (4, 5, 6, 7))
    rr = krylov.bdot(r, r); pAp = krylov.bdot(p, Ap).abs() + 1.0; rr_new = torch.empty_like(rr)
    act = torch.ones(Pc, dtype=torch.int32, device="cuda")
    def cg_step():
        krylov.bdot(p, Ap)
        nv.check(lib.lip_cg_update(nv.ptr(x), nv.ptr(r), nv.ptr(p), nv.ptr(Ap), nv.ptr(rr), nv.ptr(pAp), nv.ptr(act), nv.ptr(rr_new), Pc, D, st))
        nv.check(lib.lip_cg_direction(nv.ptr(p), nv.ptr(r), nv.ptr(rr_new), nv.ptr(rr), nv.ptr(act), Pc, D, st))
    t = timeit(cg_step); out["cg_step"] = dict(ms=t*1e3, GBps=44*D*Pc/t/1e9, probes=Pc)
    # Lanczos step j with full re-orthogonalisation (CGS2): 2 x (multi_dot: (j+2) D reads; multi_axpy_norm: (j+2) D reads + D write) + scale_store 2D
    Pl = 8
    ldq = (D + 3) // 4 * 4
    Q = torch.zeros(Pl, k, ldq, device="cuda"); Q[:, :, :D] = krylov.fill_normal(Pl * k, D, 8).reshape(Pl, k, D)
    w = krylov.fill_normal(Pl, D, 9); c = torch.empty(Pl, k, device="cuda"); nrm = torch.empty(Pl, device="cuda")
    res = {}
    for j in (4, k - 1):
        def lz():
            for _ in range(2):
                nv.check(lib.lip_multi_dot(nv.ptr(Q), nv.ptr(w), nv.ptr(c), Pl, j + 1, k, D, ldq, st))
                nv.check(lib.lip_multi_axpy_norm(nv.ptr(Q), nv.ptr(c), nv.ptr(w), nv.ptr(nrm), Pl, j + 1, k, D, ldq, st))
            nv.check(lib.lip_scale_store(nv.ptr(w), nv.ptr(nrm), nv.ptr(Q), min(j + 1, k - 1), Pl, k, D, ldq, st))
        t = timeit(lz, 10)
        bytes_ = 4 * D * Pl * (2 * ((j + 2) + (j + 3)) + 2)
        res[f"j={j}"] = dict(ms=t*1e3, GBps=bytes_/t/1e9, GBps_survey_model=4 * D * Pl * (7 + 2 * j) / t / 1e9)
    out["lanczos_step_cgs2"] = res
    del Q, w
    # float64-accumulated tall-skinny products and the streaming row combination (Hutch++ orthonormalisation, s = 20)
    s_ = 20
    Yb = krylov.fill_normal(s_, D, 10); Cm = torch.randn(s_, s_, dtype=torch.float64, device="cuda")
    t = timeit(lambda: krylov.dot_nt(Yb, Yb)); out["dot_nt_f64"] = dict(ms=t*1e3, GBps=4*D*s_/t/1e9, shape=f"({s_} x D)({s_} x D)^T")
    Ob = torch.empty_like(Yb)
    t = timeit(lambda: krylov.rows_combine(Cm, Yb, out=Ob)); out["rows_combine"] = dict(ms=t*1e3, GBps=8*D*s_/t/1e9, shape=f"({s_} x {s_})({s_} x D)")
    return out

if __name__ == "__main__":
    print(json.dumps(run(), indent=1))
