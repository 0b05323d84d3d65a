"""Restatement of the reference's ``tests/test_stochtrace.py`` (x64 on there, ``:8``).  M3 is 3000 x 3000
in the reference; the CPU oracle runs on a 600-dim instance to keep the suite in minutes, the HIP path on
the full size.  RNG-dependent tolerances hold for JAX's seeds only: the build keeps the
tolerances where they are RNG-free and states its own seeds elsewhere."""
import pytest
import torch

from fixtures import make_matrix_test_suite
from impl import cpu64, impl  # noqa: F401


@pytest.fixture(scope="module")
def suites():
    return {600: make_matrix_test_suite(n3=600), 3000: None}


def _suite(impl, suites):
    n3 = 3000 if impl.is_hip else 600
    if suites[n3] is None:
        suites[n3] = make_matrix_test_suite(n3=n3)
    return tuple(impl.tensor(M) for M in suites[n3])


def _fun(impl, M):
    """an oracle the way the reference's tests build one: ``def M1fun(v): return M1 @ v``"""
    if impl.is_hip:
        from lip_amd.ggn import BlockOperator
        return BlockOperator(lambda V: (V @ M.T).contiguous(), (M.shape[0],), (M.shape[0],), None, "Mfun")
    return lambda v: M @ v


def test_hutchinson_dense(impl, suites):
    M1, M2, M3 = _suite(impl, suites)
    st = impl.stochtrace
    # Rademacher probes are exact on a diagonal matrix (RNG-free known answer: 6)
    assert torch.isclose(cpu64(st.stochastic_trace_estimator_dense(M1, 2894598, num_samples=3)), torch.tensor(6.0, dtype=torch.float64), rtol=1e-2)
    tr3 = cpu64(st.stochastic_trace_estimator_dense(M3, 2894598, num_samples=200))
    assert torch.isclose(tr3, cpu64(torch.trace(M3)), rtol=3e-2)


def test_hutchinson_mvp(impl, suites):
    M1, M2, M3 = _suite(impl, suites)
    st = impl.stochtrace
    kw = dict(dtype=impl.dtype) if not impl.is_hip else {}
    tr1 = st.stochastic_trace_estimator_mvp(_fun(impl, M1), 3, 2894598, num_samples=3, **kw)
    assert torch.isclose(cpu64(tr1), torch.tensor(6.0, dtype=torch.float64), rtol=1e-2)
    tr3 = st.stochastic_trace_estimator_mvp(_fun(impl, M3), M3.shape[0], 2894598, num_samples=200, **kw)
    assert torch.isclose(cpu64(tr3), cpu64(torch.trace(M3)), rtol=3e-2)


def test_hutchpp_dense(impl, suites):
    M1, M2, M3 = _suite(impl, suites)
    st = impl.stochtrace
    # s = 3 probes span the whole 3-dim space: Hutch++ is exact there
    assert torch.isclose(cpu64(st.hutchpp_dense(M1, 284598, num_samples=3)), torch.tensor(6.0, dtype=torch.float64), rtol=1e-2)
    assert torch.isclose(cpu64(st.hutchpp_dense(M2, 284598, num_samples=3)), cpu64(torch.trace(M2)), rtol=impl.tol(1e-2, 1e-2))
    assert torch.isclose(cpu64(st.hutchpp_dense(M3, 284598, num_samples=100)), cpu64(torch.trace(M3)), rtol=2e-2)


def test_hutchpp_mvp(impl, suites):
    """reference :70-97: hutchpp_v2 with k >= n probes is exact — rtol 1e-8 in float64 (fp32 HIP: 2e-4)."""
    M1, M2, M3 = _suite(impl, suites)
    st = impl.stochtrace
    n = M3.shape[0]
    kw = dict(dtype=impl.dtype) if not impl.is_hip else {}
    tr = st.hutchpp_mvp(_fun(impl, M3) if impl.is_hip else (lambda V: M3 @ V), n, 284598, num_samples=40, **kw)
    assert torch.isclose(cpu64(tr), cpu64(torch.trace(M3)), rtol=5e-2)
    k = n + 200
    g = torch.Generator().manual_seed(284598)
    eps = impl.tensor((torch.randint(0, 2, (k, n), generator=g) * 2 - 1).double())
    s2 = 32
    trace_term = st.hutchpp_v2(_fun(impl, M3), lambda _: eps, s1=k - s2, s2=s2)
    assert torch.isclose(cpu64(trace_term), cpu64(torch.trace(M3)), rtol=impl.tol(1e-8, 2e-4))


def test_na_hutchpp_dense(impl, suites):
    M1, M2, M3 = _suite(impl, suites)
    st = impl.stochtrace
    assert torch.isclose(cpu64(st.na_hutchpp_dense(M3, 2894598, num_samples=100)), cpu64(torch.trace(M3)), rtol=3e-2)


def test_na_hutchpp_mvp(impl, suites):
    M1, M2, M3 = _suite(impl, suites)
    st = impl.stochtrace
    f = _fun(impl, M3) if impl.is_hip else (lambda V: M3 @ V)
    kw = dict(dtype=impl.dtype) if not impl.is_hip else {}
    tr = st.na_hutchpp_mvp(f, M3.shape[0], 2894598, num_samples=100, **kw)
    assert torch.isclose(cpu64(tr), cpu64(torch.trace(M3)), rtol=3e-2)


def test_inv_hutchpp(impl, suites):
    """reference :139-160: tr(M^-1) through CG; M1 = diag(1,2,3) -> 11/6 (RNG-free: 3 probes span R^3)."""
    M1, M2, M3 = _suite(impl, suites)
    st = impl.stochtrace
    f1 = _fun(impl, M1) if impl.is_hip else (lambda V: M1 @ V)
    tr1 = st.hutchpp_inv_mvp(f1, 3, 2894598, num_samples=3)
    assert torch.isclose(cpu64(tr1), torch.tensor(11.0 / 6.0, dtype=torch.float64), rtol=impl.tol(1e-2, 1e-2))
    # a well-conditioned SPD matrix (the reference's M3 needs ~1e4 CG iterations per probe)
    n = 300
    g = torch.Generator().manual_seed(3)
    A = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = impl.tensor(A @ A.T / n + torch.eye(n, dtype=torch.float64))
    fA = _fun(impl, A) if impl.is_hip else (lambda V: A @ V)
    tr = st.hutchpp_inv_mvp(fA, n, 2894598, num_samples=40)
    assert torch.isclose(cpu64(tr), torch.trace(torch.linalg.inv(cpu64(A))), rtol=2e-2)


def test_inv_na_hutchpp(impl, suites):
    M1, M2, M3 = _suite(impl, suites)
    st = impl.stochtrace
    n = 300
    g = torch.Generator().manual_seed(3)
    A = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = impl.tensor(A @ A.T / n + torch.eye(n, dtype=torch.float64))
    fA = _fun(impl, A) if impl.is_hip else (lambda V: A @ V)
    tr = st.na_hutchpp_inv_mvp(fA, n, 2894598, num_samples=60)
    assert torch.isclose(cpu64(tr), torch.trace(torch.linalg.inv(cpu64(A))), rtol=5e-2)


@pytest.mark.gpu
def test_quadratic_forms_of_the_ggn_operators_equal_their_products():
    """v^T (GGN + alpha I) v through the tangent sweep alone (``attach_quadratic_forms``: sum_i ||L_i^T J_i v||^2 + alpha ||v||^2)
    against <v, X v> from the block product, classifier and regressor; and the trace estimators give the same number
    with and without the shortcut (the same operator stripped of it)."""
    from lip_amd import krylov, stochtrace
    from lip_amd.ggn import BlockOperator, compute_ggn_vp
    from lip_amd.lla import compute_curvature_approx
    from lip_amd.toymodels import SimpleClassifier, SimpleRegressor, create_state
    g = torch.Generator().manual_seed(3)
    for net, Z, mt in ((SimpleClassifier(16, 2, 3), torch.randn(9, 2, generator=g), "classifier"),
                       (SimpleRegressor(12, 2), torch.randn(7, 1, generator=g), "regressor")):
        st = create_state(net, 7, dtype=torch.float32, **({"logvar": -0.7} if mt == "regressor" else {})).to(device="cuda", dtype=torch.float32)
        Zd = Z.cuda()
        for op in (compute_ggn_vp(st, Zd, mt, full_set_size=40), compute_curvature_approx(st, Zd, mt, alpha=0.37, full_set_size=40)):
            D = op.engine.D
            V = krylov.fill_normal(11, D, 5, "cuda")
            q = op.quadratic_forms(V)
            ref = (V.double() * op.rows(V).double()).sum(1)
            assert float((q - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
            plain = BlockOperator(op.rows, (D,), (D,), op.engine, "no shortcut")
            sampler = lambda *_: krylov.fill_normal(24, D, 9, "cuda")
            for est in (lambda X: stochtrace.stochastic_trace_estimator_mvp(X, D, 4, num_samples=32),
                        lambda X: stochtrace.hutchpp_mvp(X, D, 4, num_samples=6),
                        lambda X: stochtrace.hutchpp(X, sampler),
                        lambda X: stochtrace.hutchpp_v2(X, sampler, s1=8, s2=16)):
                a, b = float(est(op)), float(est(plain))
                assert abs(a - b) <= 1e-4 * abs(b), (mt, a, b)
