"""GPU parity of the Krylov / trace primitives (lip_krylov.hip) against float64 torch."""
import ctypes as C

import pytest
import torch

from lip_amd import _native as nv

pytestmark = pytest.mark.gpu


def _blk(P, N, seed):
    return torch.randn(P, N, generator=torch.Generator().manual_seed(seed)).cuda()


@pytest.mark.parametrize("P,N", [(1, 1), (3, 241), (4, 1027), (5, 100003), (2, 1084586)])
def test_bdot_axpby(P, N):
    lib = nv.load()
    X, Y = _blk(P, N, 1), _blk(P, N, 2)
    out = torch.empty(P, device="cuda")
    nv.check(lib.lip_bdot(nv.ptr(X), nv.ptr(Y), nv.ptr(out), P, N, nv.stream_ptr()), "bdot")
    ref = (X.double() * Y.double()).sum(1)
    assert torch.allclose(out.double(), ref, rtol=2e-5, atol=1e-3 * (N ** 0.5) * 1e-2)
    a = torch.rand(P, device="cuda") + 0.5
    b = torch.rand(P, device="cuda") - 0.5
    Y2 = Y.clone()
    nv.check(lib.lip_axpby(nv.ptr(Y2), nv.ptr(X), nv.ptr(a), 2.0, nv.ptr(b), -1.0, P, N, nv.stream_ptr()), "axpby")
    ref2 = 2.0 * a[:, None] * X - b[:, None] * Y
    assert torch.allclose(Y2, ref2, rtol=1e-5, atol=1e-5)
    # b_s == 0: Y may hold NaN garbage
    Y3 = torch.full_like(Y, float("nan"))
    nv.check(lib.lip_axpby(nv.ptr(Y3), nv.ptr(X), 0, 3.0, 0, 0.0, P, N, nv.stream_ptr()), "axpby copy")
    assert torch.allclose(Y3, 3.0 * X)


@pytest.mark.parametrize("P,k,kmax,N", [(2, 1, 4, 500), (3, 7, 9, 4099), (2, 33, 40, 70001)])
def test_lanczos_primitives(P, k, kmax, N):
    lib = nv.load()
    ldq = (N + 3) // 4 * 4
    Qbuf = torch.full((P, kmax, ldq), float("nan"), device="cuda")        # NaN padding must never leak
    Qbuf[:, :, :N] = _blk(P * kmax, N, 3).reshape(P, kmax, N)
    Q = Qbuf[:, :, :N]
    w = _blk(P, N, 4)
    c = torch.full((P, kmax), 7.0, device="cuda")
    nv.check(lib.lip_multi_dot(nv.ptr(Qbuf), nv.ptr(w), nv.ptr(c), P, k, kmax, N, ldq, nv.stream_ptr()), "multi_dot")
    ref = torch.einsum("pkn,pn->pk", Q[:, :k].double(), w.double())
    assert torch.allclose(c[:, :k].double(), ref, rtol=1e-4, atol=1e-2)
    assert torch.all(c[:, k:] == 7.0)
    w2 = w.clone()
    nrm = torch.empty(P, device="cuda")
    nv.check(lib.lip_multi_axpy_norm(nv.ptr(Qbuf), nv.ptr(c), nv.ptr(w2), nv.ptr(nrm), P, k, kmax, N, ldq, nv.stream_ptr()), "maxpy")
    refw = w.double() - torch.einsum("pk,pkn->pn", c[:, :k].double(), Q[:, :k].double())
    assert torch.allclose(w2.double(), refw, rtol=1e-4, atol=1e-2 * refw.abs().max().item() * 1e-2)
    assert torch.allclose(nrm.double(), (w2.double() ** 2).sum(1), rtol=1e-4)
    j = k - 1
    nv.check(lib.lip_scale_store(nv.ptr(w2), nv.ptr(nrm), nv.ptr(Qbuf), j, P, kmax, N, ldq, nv.stream_ptr()), "scale_store")
    assert torch.allclose(Q[:, j], w2 / nrm.sqrt()[:, None], rtol=1e-5, atol=1e-6)
    assert torch.all(Qbuf[:, j, N:] == 0)


@pytest.mark.parametrize("P,N", [(3, 1027), (2, 250001)])
def test_cg_primitives(P, N):
    lib = nv.load()
    x, r, p, Ap = (_blk(P, N, s) for s in (5, 6, 7, 8))
    rr_old = (r.double() ** 2).sum(1).float()
    pAp = torch.rand(P, device="cuda") + 1.0
    active = torch.tensor([1, 0, 1][:P], dtype=torch.int32, device="cuda")
    x0, r0 = x.clone(), r.clone()
    rr_new = torch.empty(P, device="cuda")
    nv.check(lib.lip_cg_update(nv.ptr(x), nv.ptr(r), nv.ptr(p), nv.ptr(Ap), nv.ptr(rr_old), nv.ptr(pAp), nv.ptr(active),
                               nv.ptr(rr_new), P, N, nv.stream_ptr()), "cg_update")
    a = (rr_old / pAp)[:, None]
    act = active.bool()[:, None]
    assert torch.allclose(x, torch.where(act, x0 + a * p, x0), rtol=1e-5, atol=1e-5)
    assert torch.allclose(r, torch.where(act, r0 - a * Ap, r0), rtol=1e-5, atol=1e-5)
    ref_rr = (r.double() ** 2).sum(1)
    assert torch.allclose(rr_new.double()[active.bool()], ref_rr[active.bool()], rtol=1e-4)
    p0 = p.clone()
    nv.check(lib.lip_cg_direction(nv.ptr(p), nv.ptr(r), nv.ptr(rr_new), nv.ptr(rr_old), nv.ptr(active), P, N,
                                  nv.stream_ptr()), "cg_direction")
    beta = (rr_new / rr_old)[:, None]
    assert torch.allclose(p, torch.where(act, r + beta * p0, p0), rtol=1e-5, atol=1e-5)


def test_fill_statistics_and_determinism():
    lib = nv.load()
    P, N = 4, 100003
    A, B = torch.empty(P, N, device="cuda"), torch.empty(P, N, device="cuda")
    nv.check(lib.lip_fill_rademacher(nv.ptr(A), P, N, 1234, nv.stream_ptr()), "rademacher")
    nv.check(lib.lip_fill_rademacher(nv.ptr(B), P, N, 1234, nv.stream_ptr()), "rademacher")
    assert torch.equal(A, B) and torch.all(A.abs() == 1.0)
    assert abs(A.mean().item()) < 0.01
    nv.check(lib.lip_fill_rademacher(nv.ptr(B), P, N, 1235, nv.stream_ptr()), "rademacher")
    assert (A != B).float().mean().item() > 0.4
    nv.check(lib.lip_fill_normal(nv.ptr(A), P, N, 99, nv.stream_ptr()), "normal")
    assert abs(A.mean().item()) < 0.01 and abs(A.std().item() - 1.0) < 0.01
    assert abs((A ** 4).mean().item() - 3.0) < 0.1


@pytest.mark.parametrize("m,n,K", [(1, 1, 1), (3, 5, 241), (8, 33, 4099), (40, 40, 100003), (70, 9, 1084586)])
def test_dot_nt_f64(m, n, K):
    """float64-accumulated A B^T of float32 rows: exact to float64 rounding — including unaligned rows (odd K as the
    row stride) and strided row views."""
    from lip_amd import krylov
    A, B = _blk(m, K, 11), _blk(n, K, 12)
    C = krylov.dot_nt(A, B)
    ref = A.double() @ B.double().T
    assert C.dtype == torch.float64 and tuple(C.shape) == (m, n)
    assert torch.allclose(C, ref, rtol=1e-12, atol=1e-12 * K ** 0.5)
    if K > 8:
        Av = _blk(m, K + 3, 13)[:, 1:K + 1]                    # rows at stride K + 3, base offset by one float
        assert torch.allclose(krylov.dot_nt(Av, B), Av.double() @ B.double().T, rtol=1e-12, atol=1e-12 * K ** 0.5)


@pytest.mark.parametrize("r,s,N", [(1, 1, 1), (3, 7, 241), (13, 20, 4099), (36, 36, 100003), (5, 300, 70001)])
def test_rows_combine(r, s, N):
    from lip_amd import krylov
    Y, Z = _blk(s, N, 21), _blk(r, N, 22)
    Cm = torch.randn(r, s, dtype=torch.float64, generator=torch.Generator().manual_seed(23)).cuda()
    out = krylov.rows_combine(Cm, Y)
    ref = Cm @ Y.double()
    tol = 2e-6 * (s ** 0.5) * ref.abs().max().item()
    assert (out.double() - ref).abs().max().item() <= tol
    out2 = krylov.rows_combine(Cm, Y, Z=Z, zscale=-0.5)
    assert (out2.double() - (ref - 0.5 * Z.double())).abs().max().item() <= tol + 1e-6


@pytest.mark.parametrize("s,N,rank", [(20, 100003, 20), (36, 1084586, 36), (12, 5000, 7)])
def test_gram_orthonormalize(s, N, rank):
    """CholeskyQR2-style orthonormalisation on the HIP kernels: orthonormal to float32 rounding, same span as the
    input rows, and a rank-deficient block yields rank rows."""
    from lip_amd import krylov
    Y = _blk(rank, N, 31)
    Y = Y * torch.logspace(0, 3, rank, device="cuda")[:, None]          # cond(Y) = 1e3
    if rank < s:
        mix = torch.randn(s, rank, generator=torch.Generator().manual_seed(32)).cuda()
        Y = (mix.double() @ Y.double()).float()
    Q = krylov.gram_orthonormalize(Y)
    assert Q.shape[0] == rank
    G = Q.double() @ Q.double().T
    assert (G - torch.eye(rank, device="cuda", dtype=torch.float64)).abs().max().item() <= 2e-6
    proj = Y.double() - (Y.double() @ Q.double().T) @ Q.double()     # rows of Y lie in span(Q)
    assert proj.norm().item() <= 1e-5 * Y.double().norm().item()


@pytest.mark.parametrize("m,n,K", [(1, 1, 1), (5, 7, 241), (130, 33, 4099), (256, 450, 100003), (8, 500, 1084586)])
def test_gemm_nt(m, n, K):
    """split-K MFMA A B^T of K-contiguous float32 rows against float64 (float32 accumulation: eps * sqrt(K) relative to
    the row norms), unaligned row strides included."""
    from lip_amd import krylov
    A, B = _blk(m, K, 41), _blk(n, K, 42)
    C = krylov.gemm_nt(A, B)
    ref = A.double() @ B.double().T
    scale = (A.double().norm(dim=1)[:, None] * B.double().norm(dim=1)[None, :])
    assert ((C.double() - ref).abs() / scale).max().item() <= 2e-6
    if K > 8:
        Av = _blk(m, K + 3, 43)[:, 1:K + 1]
        Cv = krylov.gemm_nt(Av, B)
        refv = Av.double() @ B.double().T
        assert ((Cv.double() - refv).abs() / (Av.double().norm(dim=1)[:, None] * B.double().norm(dim=1)[None, :])).max().item() <= 2e-6


def test_gemm_nn_axpy_matches_float64():
    """lip_gemm_nn_axpy: Out = T B + beta V (second pass of a block of posterior draws; factor-mode second product) for
    ragged m / k / N, with and without the addend, out of place and in place; 2e-6 * max|ref| (f32 MFMA = fmaf chain)."""
    from lip_amd import krylov
    g = torch.Generator().manual_seed(3)
    for (m, k, N) in ((256, 450, 100003), (7, 33, 1300), (130, 16, 257), (1, 500, 4099)):
        T = torch.randn(m, k, generator=g).cuda()
        B = torch.randn(k, N, generator=g).cuda()
        V = torch.randn(m, N, generator=g).cuda()
        ref = T.double() @ B.double() + 0.37 * V.double()
        out = krylov.gemm_nn_axpy(T, B, V, 0.37)
        assert (out.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item(), (m, k, N)
        out0 = krylov.gemm_nn_axpy(T, B)
        ref0 = T.double() @ B.double()
        assert (out0.double() - ref0).abs().max().item() <= 2e-6 * ref0.abs().max().item()
        Vc = V.clone()
        krylov.gemm_nn_axpy(T, B, Vc, 0.37, out=Vc)             # in place: the addend is the output
        assert torch.equal(Vc, out)
    # strided rows (a slice of a wider matrix) for B and V
    Bw = torch.randn(40, 3000, generator=g).cuda()
    Tw = torch.randn(9, 40, generator=g).cuda()
    Vw = torch.randn(9, 3000, generator=g).cuda()
    o = krylov.gemm_nn_axpy(Tw, Bw[:, 100:1100], Vw[:, 5:1005], 2.0)
    r = Tw.double() @ Bw[:, 100:1100].double() + 2.0 * Vw[:, 5:1005].double()
    assert (o.double() - r).abs().max().item() <= 2e-6 * r.abs().max().item()
