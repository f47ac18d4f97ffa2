"""CPU oracle for the CG + kernel-matvec hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy (fp64 / fp32) restatement of the algorithm of the
reference `awav/conjugate-gradient-sparse-gp` for the rows of SURVEY.md §8(a).
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it -- and there only as the checker.  The product
package (`conjugate-gradient-sparse-gp_amd/cggp`) never imports it and raises if
the HIP library is missing.

Parity status (be precise about what is and is not pinned):

* The reference ships NO golden vectors and cannot be imported here
  (TensorFlow / GPflow / TFP are not installed and there is no network; this
  is unavailability, not a permission denial -- SURVEY.md §8c).
* What the reference's own tests DO pin (`cggp/cg_test.py:12-77`) are
  closed-form identities: CG(K+s2 I, b) == solve(K+s2 I, b); the custom CG
  gradient == autodiff through solve; eval_logdet forward == 0 and its
  backward == d logdet.  `tests/test_oracle.py` checks this restatement against
  exactly those identities (seeded, at 1e-10 instead of 1e-3), so rows
  CG1-CG5, K4, M5 are pinned by the reference's known-answer tests.
* The GPflow arithmetic (rows K1-K3, S1, Gaussian likelihood) is a
  third-party dependency that is absent from /root/reference
  (`requirements.txt:1` gpflow>=2.5.2, no lockfile).  It is restated from the
  published formulas (gpflow/kernels/stationaries.py, utilities/ops.py,
  covariances/kuus.py+kufs.py, models/sgpr.py, likelihoods Gaussian) and pinned
  by hand-computable known answers and cross-identities (CG model == Cholesky
  twin `cggp/models.py:250-276`, SGPR-CG == two-Cholesky closed form) only:
  **parity unpinned** against GPflow outputs.

* Independent third-party implementations installed here (scikit-learn's RBF / Matern kernels and
  GaussianProcessRegressor, SciPy's conjugate gradient) agree with the restated formulas
  (`tests/test_oracle_thirdparty.py`); that is a check of the restatement, not of GPflow's outputs,
  and does not change the status above.

Every function cites the reference file:line it follows.
"""

from . import kernels, cg, models, distance, cluster, selection  # noqa: F401
