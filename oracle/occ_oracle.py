"""ctypes bindings of the CPU oracle (``liboccoracle.so``, built from ``occ_oracle.c``).

TEST INFRASTRUCTURE ONLY -- imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the ``occuspytial_amd`` package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags='C_CONTIGUOUS')
_ip = np.ctypeslib.ndpointer(dtype=np.int64, flags='C_CONTIGUOUS')
_bp = np.ctypeslib.ndpointer(dtype=np.uint8, flags='C_CONTIGUOUS')
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags='C_CONTIGUOUS')


def build():
    """Compile the oracle with gcc (idempotent)."""
    subprocess.run(['make', '-s', '-C', _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboccoracle.so')
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        L.orc_philox4x32_10.restype = None
        L.orc_u01.argtypes = [C.c_uint64]
        L.orc_u01.restype = C.c_double
        for f in (L.orc_block_normal, L.orc_block_uniform):
            f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
            f.restype = C.c_double
        L.orc_pg1_array.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_long, _dp, _dp]
        L.orc_pg1_array.restype = None
        L.orc_std_gamma_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_double]
        L.orc_std_gamma_draw.restype = C.c_double
        L.orc_tau_rate.argtypes = [C.c_long, _ip, _ip, _dp, _dp, C.c_double]
        L.orc_tau_rate.restype = C.c_double
        L.orc_minres_joint.argtypes = [C.c_long, _ip, _ip, _dp, _dp, C.c_double, _dp, _dp, C.c_double,
                                       C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_int)]
        L.orc_minres_joint.restype = C.c_long
        L.orc_ensure_sums_to_zero.argtypes = [C.c_long, _dp, _dp, _dp]
        L.orc_ensure_sums_to_zero.restype = None
        L.orc_precision_mvnorm.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
        L.orc_precision_mvnorm.restype = C.c_int
        L.orc_beta_system.argtypes = [C.c_long, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_beta_system.restype = None
        L.orc_alpha_system.argtypes = [C.c_long, C.c_int, _ip, _bp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_alpha_system.restype = None
        L.orc_z_prob.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_long, _dp, _dp]
        L.orc_z_prob.restype = C.c_double
        L.orc_expit.argtypes = [C.c_double]
        L.orc_expit.restype = C.c_double
        L.orc_edge_prior_term.argtypes = [C.c_long, _ip, _ip, _dp, C.c_uint64, C.c_uint32, _dp]
        L.orc_edge_prior_term.restype = None
        L.orc_create.argtypes = [C.c_long, C.c_int, C.c_int, C.c_long, _ip, _ip, _dp, _dp, _ip, _ip, _dp,
                                 _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.c_uint64]
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_destroy.restype = None
        L.orc_set_start.argtypes = [C.c_void_p, _dp, _dp, C.c_double, _dp]
        L.orc_set_start.restype = None
        for name in ('orc_update_omega_b', 'orc_update_tau', 'orc_update_eta', 'orc_update_beta',
                     'orc_update_omega_a', 'orc_update_alpha', 'orc_update_z', 'orc_step'):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_int
        L.orc_rsr_theta.argtypes = [C.c_long, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp]
        L.orc_rsr_theta.restype = C.c_int
        L.orc_set_rsr.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
        L.orc_set_rsr.restype = C.c_int
        L.orc_set_dense_eigen.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_set_dense_eigen.restype = None
        L.orc_run.argtypes = [C.c_void_p, C.c_long, C.c_long, _dp, _dp, _dp]
        L.orc_run.restype = C.c_int
        L.orc_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_long]
        L.orc_get.restype = C.c_long
        L.orc_set.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_long]
        L.orc_set.restype = C.c_int
        _LIB = L
    return _LIB


_ERRORS = {1: 'MINRES solver did not converge!', 2: 'Cholesky factorization/solver failed!'}


def _csr(Q):
    return (np.ascontiguousarray(Q.indptr, dtype=np.int64), np.ascontiguousarray(Q.indices, dtype=np.int64),
            np.ascontiguousarray(Q.data, dtype=np.float64))


# ---- piece-wise wrappers (injected variates) ------------------------------------------------------
def philox(ctr, key):
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(np.asarray(ctr, dtype=np.uint32), np.asarray(key, dtype=np.uint32), out)
    return out


def pg1(z, key=1, it=0, stream=1):
    z = np.ascontiguousarray(z, dtype=np.float64)
    out = np.empty_like(z)
    lib().orc_pg1_array(key, it, stream, z.size, z, out)
    return out


def std_gamma(shape, key=1, it=0, stream=2):
    return lib().orc_std_gamma_draw(key, it, stream, shape)


def tau_rate(Q, eta, tau_rate_prior):
    ip, ix, d = _csr(Q)
    return lib().orc_tau_rate(Q.shape[0], ip, ix, d, np.ascontiguousarray(eta), tau_rate_prior)


def minres_joint(Q, omega, tau, rhs, x0=None, rtol=1e-5, maxiter=None):
    n = Q.shape[0]
    ip, ix, d = _csr(Q)
    xz = np.zeros(2 * n) if x0 is None else np.array(x0, dtype=np.float64)
    itn, istop = C.c_long(0), C.c_int(0)
    info = lib().orc_minres_joint(n, ip, ix, d, np.ascontiguousarray(omega), tau, np.ascontiguousarray(rhs), xz,
                                  rtol, 10 * n if maxiter is None else maxiter, C.byref(itn), C.byref(istop))
    return xz, info, itn.value, istop.value


def ensure_sums_to_zero(x, z):
    out = np.empty_like(x)
    lib().orc_ensure_sums_to_zero(x.size, np.ascontiguousarray(x), np.ascontiguousarray(z), out)
    return out


def precision_mvnorm(b, prec, eps):
    """Returns (draw, overwritten prec, status)."""
    d = b.size
    work = np.array(prec, dtype=np.float64, order='C')
    out = np.empty(d)
    st = lib().orc_precision_mvnorm(d, np.ascontiguousarray(b), work, np.ascontiguousarray(eps), out)
    return out, work, st


def beta_system(X, omega, k, spat, b_prec, b_prec_by_mu):
    n, p = X.shape
    A, r = np.empty((p, p)), np.empty(p)
    lib().orc_beta_system(n, p, np.ascontiguousarray(X), np.ascontiguousarray(omega), np.ascontiguousarray(k),
                          np.ascontiguousarray(spat), np.ascontiguousarray(b_prec),
                          np.ascontiguousarray(b_prec_by_mu), A, r)
    return A, r


def alpha_system(site_ptr, exists_site, W, yrow, omega_a, a_prec, a_prec_by_mu):
    q = W.shape[1]
    A, r = np.empty((q, q)), np.empty(q)
    lib().orc_alpha_system(len(site_ptr) - 1, q, np.ascontiguousarray(site_ptr, dtype=np.int64),
                           np.ascontiguousarray(exists_site, dtype=np.uint8), np.ascontiguousarray(W),
                           np.ascontiguousarray(yrow, dtype=np.float64), np.ascontiguousarray(omega_a),
                           np.ascontiguousarray(a_prec), np.ascontiguousarray(a_prec_by_mu), A, r)
    return A, r


def z_prob(xrow, beta, eta_i, Wrows, alpha):
    Wrows = np.ascontiguousarray(Wrows, dtype=np.float64).reshape(-1, alpha.size)
    return lib().orc_z_prob(beta.size, alpha.size, np.ascontiguousarray(xrow), np.ascontiguousarray(beta),
                            float(eta_i), Wrows.shape[0], Wrows, np.ascontiguousarray(alpha))


def rsr_theta(K, Qr, Er, b, omega, tau, eps1, eps2):
    """theta of the reduced-rank conditional from injected standard normals (orc_rsr_theta)."""
    K = np.ascontiguousarray(K, dtype=np.float64)
    n, r = K.shape
    theta = np.zeros(r)
    code = lib().orc_rsr_theta(n, r, K, np.ascontiguousarray(Qr, dtype=np.float64), np.ascontiguousarray(Er, dtype=np.float64),
                               np.ascontiguousarray(b, dtype=np.float64), np.ascontiguousarray(omega, dtype=np.float64), float(tau),
                               np.ascontiguousarray(eps1, dtype=np.float64), np.ascontiguousarray(eps2, dtype=np.float64), theta)
    return theta, code


def edge_prior_term(Q, key, it):
    ip, ix, d = _csr(Q)
    u = np.empty(Q.shape[0])
    lib().orc_edge_prior_term(Q.shape[0], ip, ix, d, key, it, u)
    return u


def dense_eigenfactor(Q):
    """``E = U[:, 1:] sqrt(s[1:])`` from the dense ``eigh`` of ``Q`` exactly as the reference's
    ``_EtaICARPosterior.__init__`` forms it (logit.py:66-67): O(n^3) time, O(n^2) memory."""
    s, u = np.linalg.eigh(Q.toarray() if hasattr(Q, 'toarray') else np.asarray(Q))
    return np.ascontiguousarray(u[:, 1:] * np.sqrt(np.clip(s[1:], 0.0, None)))


# ---- whole sampler --------------------------------------------------------------------------------
class OracleSampler:
    """One chain of the CPU restatement, driven from a ``FlatProblem`` (occuspytial_amd._problem)."""

    def __init__(self, prob, key):
        self.prob = prob
        ip, ix, d = _csr(prob.Q)
        self._h = lib().orc_create(prob.n, prob.p, prob.q, prob.S, ip, ix, d, prob.X, prob.site_id,
                                   prob.site_ptr, prob.W, prob.y, prob.a_mu, prob.a_prec, prob.b_mu,
                                   prob.b_prec, prob.tau_rate, prob.tau_shape, int(key))
        self.key = int(key)
        self.rsr = getattr(prob, 'rsr', None)
        if self.rsr is not None:   # LogitRSRGibbs: eta = K theta
            if lib().orc_set_rsr(self._h, int(self.rsr['dim']), self.rsr['K'], self.rsr['Q'], self.rsr['E']):
                raise ValueError('bad reduced-rank basis')
        elif getattr(prob, 'prior_factor', None) is not None:   # the problem asks for the reference-form prior draw
            self.set_dense_eigen(prob.prior_factor)

    def __del__(self):
        if getattr(self, '_h', None):
            lib().orc_destroy(self._h)
            self._h = None

    def set_dense_eigen(self, E):
        """Reference-faithful prior draw (logit.py:64-67, 77): ``E`` is the n x (n-1) eigenfactor of Q
        (:func:`dense_eigenfactor`), shared read-only between samplers; ``None`` returns to the edge form."""
        if E is None:
            self._E = None
            lib().orc_set_dense_eigen(self._h, None)
            return
        E = np.ascontiguousarray(E, dtype=np.float64)
        if E.shape != (self.prob.n, self.prob.n - 1):
            raise ValueError('E must be n x (n - 1)')
        self._E = E   # keep it alive: the C side borrows the pointer
        lib().orc_set_dense_eigen(self._h, E.ctypes.data_as(C.c_void_p))

    @staticmethod
    def _check(code):
        if code:
            raise RuntimeError(_ERRORS.get(code, f'oracle error {code}'))

    def set_start(self, alpha, beta, tau, eta):
        """``eta``: the n spatial effects, or -- reduced-rank model -- the coefficients theta of the basis."""
        eta = np.ascontiguousarray(eta, dtype=np.float64)
        theta = None
        if self.rsr is not None:
            theta, eta = eta, np.zeros(self.prob.n)
        lib().orc_set_start(self._h, np.ascontiguousarray(alpha, dtype=np.float64),
                            np.ascontiguousarray(beta, dtype=np.float64), float(tau), eta)
        if theta is not None:
            self.set('theta', theta)

    def update(self, name):
        self._check(getattr(lib(), 'orc_update_' + name)(self._h))

    def step(self):
        self._check(lib().orc_step(self._h))

    def run(self, n_iter, burnin=0):
        keep = n_iter - burnin
        a, b, t = np.zeros((keep, self.prob.q)), np.zeros((keep, self.prob.p)), np.zeros(keep)
        self._check(lib().orc_run(self._h, n_iter, burnin, a, b, t))
        return a, b, t

    def get(self, name):
        size = lib().orc_get(self._h, name.encode(), None, 0)
        if size < 0:
            raise KeyError(name)
        out = np.empty(size)
        lib().orc_get(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), size)
        return out[0] if name in ('tau', 'minres_itn', 'iter') else out

    def set(self, name, value):
        v = np.ascontiguousarray(np.atleast_1d(value), dtype=np.float64)
        if lib().orc_set(self._h, name.encode(), v, v.size):
            raise KeyError(name)
