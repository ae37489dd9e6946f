"""Host-side helpers with the signatures of the reference's Cython module ``occuspytial/distributions.pyx``.

They are not on the device path (the engine draws its p x p / q x q Gaussians in registers and projects eta
inside ``k_iter``); they exist because the reference exposes them as public functions and because the host
uses the same draw for default start values.  numpy / scipy only.
"""
import numpy as np
from scipy.linalg import cholesky, solve_triangular

__all__ = ('precision_mvnorm', 'ensure_sums_to_zero')

_FAILURE_MESSAGE = 'Cholesky factorization/solver failed!'   # distributions.pyx:21


def ensure_sums_to_zero(x, z, out):
    """``out = x - (sum(x) / sum(z)) z``: a draw conditioned on ``sum(out) == 0`` (distributions.pyx:24-39)."""
    x = np.asarray(x, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64)
    a = -x.sum() / z.sum()
    out[...] = x + a * z


def precision_mvnorm(b, prec, random_state=None):
    r"""One draw from :math:`N(\Lambda^{-1} b, \Lambda^{-1})` given ``b`` and the precision ``prec``
    (distributions.pyx:42-110).

    As in the reference the standard normals come from ``numpy.random.default_rng(random_state)`` (one
    ``standard_normal(n)`` call) and ``prec`` is overwritten: its lower triangle receives the transposed upper
    Cholesky factor.  Unlike the reference -- whose check is dead code, ``dpotrs`` overwrites ``dpotrf``'s
    ``info`` -- a matrix that is not positive definite raises ``RuntimeError``.
    """
    b = np.ascontiguousarray(b, dtype=np.float64)
    if prec.shape != (b.size, b.size):
        raise ValueError('prec must be a square matrix of the size of b')
    rng = np.random.default_rng(random_state)
    eps = rng.standard_normal(b.size)
    try:
        U = cholesky(prec, lower=False, check_finite=False)        # prec = U'U
    except np.linalg.LinAlgError:
        raise RuntimeError(_FAILURE_MESSAGE) from None
    out = U.T @ eps + b                                            # dtrmv('U', 'T') + b
    out = solve_triangular(U, solve_triangular(U, out, trans='T', check_finite=False), check_finite=False)  # dpotrs
    il = np.tril_indices(b.size)
    prec[il] = U.T[il]
    return out
