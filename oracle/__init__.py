"""CPU oracle for the CCA Gram/eigen hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (NumPy / SciPy / CPU-torch) of the reference
algorithms for the hot path named in BASELINE.json.  It exists so that the HIP
path can be checked; it is never the thing that is shipped or measured as the
product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  ``cca_zoo_amd`` never imports it.

Two layers:

* :mod:`oracle.reference_form` -- the algorithms *as the reference structures
  them* (thin SVD of each n x d view, ``np.cov``, ``scipy.linalg.eigh`` with
  ``subset_by_index``, torch ``eigh`` + autograd for the loss).  This is what
  ``cpu_baseline`` times.  Every function cites the reference file:line.
* :mod:`oracle.gram_form` -- the same results restated from second moments
  only (Gram + column sums -> Cholesky whitening -> top-k eigen/SVD).  This is
  the algorithmic specification the HIP kernels follow, written with dense
  NumPy so it can be compared against ``reference_form`` and the goldens.

Parity pin: ``tests/golden/*.npz`` hold outputs of the *real* reference
(`/root/reference`, imported through the shim in ``tools/gen_golden.py``) on
stored inputs; ``tests/test_oracle_golden.py`` checks both layers against
them.  The arithmetic of the reference lives in un-vendored third-party
packages (numpy 2.2.6 LAPACK gesdd/syevd, scipy 1.15.3 syevr/sygvx,
scikit-learn 1.7.2 PCA, torch eigh) -- SURVEY.md section 8(c).
"""
