"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A restatement, in PyTorch ``torch.func`` on the CPU (float64 by default), of the
reference's algorithm for the hot path: ``src/ggn.py``, ``src/lla.py``,
``src/sample.py``, ``src/stochtrace.py``, ``src/matfree_monkeypatch.py`` and the pieces of
the third-party libraries those files call (``matfree`` — un-pinned in
``requirements.txt:5`` —, ``jax.scipy.sparse.linalg.cg``).  Every function cites the
reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker / reported baseline.  The product path
(``laplace-inducing-points_amd/``) never imports it and has no CPU fallback.

PINNING.  The reference itself cannot be imported here (jax / flax / matfree /
tensorflow_probability are not installed; plain ``ModuleNotFoundError``) and
it stores no golden vectors.  The oracle is pinned by the RNG-free known answers the
reference's own tests hold, restated in ``tests/test_oracle_*.py``:
  * linear-model GGN == exp(-logvar) * [[14.46, 3.6], [3.6, 4]] on X = [-1, 0, 1.1, 3.5]
    (``tests/fixtures.py:24``, ``tests/test_ggn.py:21-54,87-102``);
  * ``vmap(ggn_vp)(I) == dense GGN`` and ``W(W^T(I)) == dense GGN`` at atol 1e-8
    (``tests/test_ggn.py:102,131``, ``tests/test_sample.py:42,49,83,90``);
  * Hutchinson with Rademacher probes exact on diag(1,2,3) -> 6; Hutch++ exact when
    s1 >= n (3200 probes on a 3000-dim PSD matrix, rtol 1e-8); tr(M1^-1) = 11/6
    (``tests/test_stochtrace.py:16-18,90-97,144-154``);
  * Lanczos inverse square root on diag(1..100)/100, 20 steps, rtol 1e-1
    (``tests/test_sample.py:334-355``);
  * sampler moments vs dense posterior, atol 1e-1 (``tests/test_sample.py:467-508``).
The reference's known answers are all regressor / dense-matrix cases; the softmax Hessian and its square-root
forms, eval-mode BN, strided SAME convolutions, residual adds and the flat-theta order are pinned in addition by
anchors written by hand WITHOUT this package and without ``NetSpec.forward`` (``tests/test_oracle_anchors.py``):
a linear softmax classifier's GGN against ``torch.autograd.functional.hessian`` of a hand-written cross-entropy (the
idea of the reference's ``tests/test_ggn.py:21-54``), a conv + BN + residual + stride-2 net written from the Flax
definition with per-example ``jacrev`` and an explicit diag(p) - p p^T, and L L^T = diag(p) - p p^T.
Unpinned by any reference fixture (stated here and in DESIGN.md): the eigenvalue clip of
``src/matfree_monkeypatch.py:19`` and the bidiagonalisation log-det of
``src/train_inducing.py:156-157`` — "parity unpinned" for those two.
"""
