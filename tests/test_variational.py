"""Restatement of the reference's ``tests/test_variational.py`` on the HIP path.  The reference file is stale against
its own source (it passes ``return_Hinv=`` to ``compute_curvature_approx_dense``, SURVEY §4.1-1, and needs a MAP
checkpoint that is not shipped); kept are its two estimator structures and tolerances:
  * trace term (``:88-113``): Hutch++ (150 probes) of the composite v -> P (P_z^-1 v), P_z^-1 by conjugate gradients,
    against the dense trace, rtol 1e-2;
  * log-det term (``:126-150``): stochastic Lanczos quadrature (10 matvecs, 150 normal probes) of P_z against the
    dense slogdet, rtol 1e-1.
Dense reference values come from the float64 oracle on the same random-init classifier."""
import pytest
import torch

from fixtures import classification_2d_data, classifier_state  # noqa: F401
import oracle.lla as olla

pytestmark = pytest.mark.gpu


def _setup(classification_2d_data, classifier_state):
    X, y = classification_2d_data
    N, M = X.shape[0], 10
    Z = torch.rand(M, X.shape[1], dtype=torch.float64, generator=torch.Generator().manual_seed(178189))
    alpha = 0.5
    P, *_ = olla.compute_curvature_approx_dense(classifier_state, X, alpha=alpha, model_type="classifier", full_set_size=N)
    Pz, *_ = olla.compute_curvature_approx_dense(classifier_state, Z, alpha=alpha, model_type="classifier", full_set_size=N)
    return X, Z, N, alpha, P, Pz


def test_scalable_trace_term(classification_2d_data, classifier_state):
    from src import krylov
    from src.lla import compute_curvature_approx
    from src.stochtrace import hutchpp_mvp
    X, Z, N, alpha, P, Pz = _setup(classification_2d_data, classifier_state)
    trace_dense = torch.trace(P @ torch.linalg.inv(Pz)).item()
    st = classifier_state.to(device="cuda", dtype=torch.float32)
    S_full_vp = compute_curvature_approx(st, X.float().cuda(), alpha=alpha, model_type="classifier", full_set_size=N)
    S_induc_vp = compute_curvature_approx(st, Z.float().cuda(), alpha=alpha, model_type="classifier", full_set_size=N)
    D = P.shape[0]

    def composite_vp(V):                       # columns in, columns out like the reference's vmap(in_axes=1)
        Xs, _ = krylov.cg(S_induc_vp.rows, V.T.contiguous(), tol=1e-6, maxiter=10 * D)
        return S_full_vp.rows(Xs).T

    trace_scalable = float(hutchpp_mvp(composite_vp, D=D, seed=7, num_samples=150))
    assert abs(trace_scalable - trace_dense) <= 1e-2 * abs(trace_dense), (trace_scalable, trace_dense)


def test_scalable_logdet_term(classification_2d_data, classifier_state):
    from src import krylov
    from src.lla import compute_curvature_approx
    X, Z, N, alpha, P, Pz = _setup(classification_2d_data, classifier_state)
    logdet_dense = torch.linalg.slogdet(Pz)[1].item()
    st = classifier_state.to(device="cuda", dtype=torch.float32)
    S_induc_vp = compute_curvature_approx(st, Z.float().cuda(), alpha=alpha, model_type="classifier", full_set_size=N)
    D = P.shape[0]
    vals = []
    for key in (1, 2):                          # jax.random.split(key, 2) -> two estimator calls, averaged (:80-82)
        V0 = krylov.fill_normal(150, D, 1000 + key, "cuda")
        Q, diag, off = krylov.lanczos_tridiag(S_induc_vp.rows, V0, 10)
        T = krylov.tridiag_dense(diag, off).double()
        ev, U = torch.linalg.eigh(T)
        quad = (U[:, 0, :] ** 2 * torch.log(ev.clamp_min(1e-30))).sum(-1)          # e1^T log(T) e1
        vals.append((quad * (V0.double() ** 2).sum(-1)).mean().item())
    logdet_scalable = sum(vals) / len(vals)
    assert abs(logdet_scalable - logdet_dense) <= 1e-1 * abs(logdet_dense), (logdet_scalable, logdet_dense)
