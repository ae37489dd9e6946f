"""Host-side flattening of the sampler inputs into the plain arrays the C ABI takes.

The reference keeps ``W`` and ``y`` as ``Dict[int, ndarray]`` behind the Cython ``Data`` class
(``data.pyx:34-147``) and gathers/concatenates per-site blocks on every iteration
(``gibbs/logit.py:187-189, 221``).  Here the ragged per-site blocks are laid out ONCE as a flat
``(R, q)`` matrix plus a ``site_ptr`` offset array (surveyed sites in dict order, exactly the order
``Data.surveyed`` reports), and ``Q`` becomes CSR with sorted column indices.  Set-up only; numpy
and scipy are used here and nowhere on the per-iteration path.
"""
import numpy as np
from scipy import sparse
from scipy.linalg import solve_triangular

MAX_COVARIATES = 32  # limit of the engine (up to 8: register-resident accumulators; beyond: its generic kernels)
DENSE_PRIOR_MAX_SITES = 16384  # the reference-form prior draw keeps an n x (n - 1) factor: 2 GB at this size


class FlatProblem:
    """Plain-array view of ``(Q, W, X, y, hparams)``.

    Index sets follow ``gibbs/base.py:112-152``: ``obs_site[s]`` says whether surveyed site ``s``
    had any detection; sites without one are the reference's ``not_obs``; sites absent from ``W`` are
    ``not_surveyed``.
    """

    def __init__(self, Q, W, X, y, hparams=None, check_singular=True, prior_draw='auto'):
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError('X must be a 2-D array')
        self.n, self.p = X.shape
        self.X = X

        Qc = sparse.csr_matrix(Q).astype(np.float64)
        Qc.sum_duplicates()
        Qc.sort_indices()
        if Qc.shape != (self.n, self.n):
            raise ValueError('Q must be n x n with n = X.shape[0]')
        if prior_draw not in ('auto', 'edge', 'dense'):
            raise ValueError("prior_draw must be 'auto', 'edge' or 'dense'")
        # How the N(0, Q) prior term of the eta conditional is drawn.  'edge' (what 'auto' picks for an ICAR precision
        # D - W, W >= 0): u = B'eps with Q = B'B, no set-up, O(nnz) work.  'dense': the reference's own form
        # (gibbs/logit.py:64-67, 77): E = U[:, 1:] sqrt(s[1:]) from the dense eigh of Q, u = E eps -- O(n^3) set-up,
        # O(n^2) bytes per iteration, but any symmetric positive semi-definite SINGULAR Q is accepted (the reference's
        # own acceptance test, base.py:166-170, asks for nothing more); 'auto' falls back to it when Q is singular but
        # not an ICAR precision.
        self.prior_factor = None
        if prior_draw != 'dense':
            try:
                if check_singular or prior_draw == 'edge':
                    _verify_spatial_precision(Qc)
            except ValueError as exc:
                if prior_draw == 'edge' or 'must be singular' in str(exc) and not _is_singular_psd(Qc):
                    raise
                prior_draw = 'dense'
        if prior_draw == 'dense':
            self.prior_factor = dense_prior_factor(Qc)
        self.Q = Qc

        if list(W.keys()) != list(y.keys()):
            if set(W.keys()) != set(y.keys()):
                raise ValueError('W and y must describe the same surveyed sites')
        sites = list(W.keys())
        self.S = len(sites)
        self.site_id = np.asarray(sites, dtype=np.int64)
        if self.S and (self.site_id.min() < 0 or self.site_id.max() >= self.n):
            raise ValueError('site numbers must lie in [0, n)')
        visits = np.array([np.asarray(W[s]).shape[0] for s in sites], dtype=np.int64)
        self.site_ptr = np.zeros(self.S + 1, dtype=np.int64)
        np.cumsum(visits, out=self.site_ptr[1:])
        self.R = int(self.site_ptr[-1])
        first = np.asarray(W[sites[0]])
        self.q = first.shape[1] if first.ndim == 2 else 1
        self.W = np.ascontiguousarray(
            np.concatenate([np.asarray(W[s], dtype=np.float64).reshape(-1, self.q) for s in sites]))
        self.y = np.ascontiguousarray(
            np.concatenate([np.asarray(y[s], dtype=np.float64).ravel() for s in sites]))
        if self.y.size != self.R:
            raise ValueError('y and W disagree on the number of visits')
        if self.p > MAX_COVARIATES or self.q > MAX_COVARIATES:
            raise ValueError(f'at most {MAX_COVARIATES} occupancy and {MAX_COVARIATES} detection covariates are supported')

        # gibbs/base.py:113-137
        seg_any = np.add.reduceat(self.y != 0, self.site_ptr[:-1]) > 0 if self.R else np.zeros(0, bool)
        seg_any = np.where(visits > 0, seg_any, False)
        self.obs_site = seg_any.astype(np.uint8)
        self.surveyed = sites
        surveyed_mask = np.zeros(self.n, dtype=bool)
        surveyed_mask[self.site_id] = True
        self.not_surveyed = np.flatnonzero(~surveyed_mask).tolist()
        self.obs = [s for s, o in zip(sites, self.obs_site) if o]
        self.not_obs = [s for s, o in zip(sites, self.obs_site) if not o]
        self.z0 = np.ones(self.n)
        self.z0[self.site_id] = self.obs_site

        self._set_hyperparams(hparams)
        self.rsr = None   # set by enable_rsr(): the reduced-rank model of LogitRSRGibbs

    # ---- reduced-rank spatial effects (reference gibbs/logit.py:413-460) --------------------------------
    def enable_rsr(self, r=0.5, q=None, default_tau_shape=True):
        """Moran-operator basis ``K`` (n x m) of the reference's ``_configure_rsr``, the reduced precision
        ``K'QK`` and its eigenfactor ``E`` (``E E' = K'QK``, ``_EtaRSRPosterior.__init__``).  ``q`` fixes the
        number of columns, else eigenvalues ``>= r`` are kept.  Dense n x n linear algebra on the host, once."""
        X = self.X
        chol = np.linalg.cholesky(X.T @ X)
        zi = solve_triangular(chol, np.eye(self.p), lower=True)
        XTX_i = solve_triangular(chol, zi, lower=True, trans=1)
        P = -np.linalg.multi_dot([X, XTX_i, X.T])
        P[np.diag_indices_from(P)] += 1
        A = self.Q.copy()
        A.data = -A.data
        A.setdiag(0)
        omega = self.n * (P.T @ A @ P) / A.sum()
        w, v = np.linalg.eigh(omega)
        if q:
            m = int(q)
        else:
            if not 0 <= r <= 1:
                raise ValueError('Threshold value needs to be in [0, 1]')
            m = int(w[w >= r].size)
            if not m:
                raise ValueError('The Moran Operator Matrix of the data has no positive '
                                 'eigenvalues. Set threshold to a lower value')
        K = np.ascontiguousarray(v[:, -m:])
        Qr = np.ascontiguousarray(K.T @ (self.Q @ K))
        s, u = np.linalg.eigh(Qr)
        E = np.ascontiguousarray(u * np.sqrt(np.clip(s, 0.0, None)))
        self.rsr = {'K': K, 'Q': Qr, 'E': E, 'dim': m}
        if default_tau_shape:
            self.tau_shape = 0.5 + 0.5 * m          # logit.py:448-451 (only when no hyper-parameters were given)
            self.hparams['tau_shape'] = self.tau_shape
        return self.rsr

    # ---- plain-array round trip (what a multi-GPU launch broadcasts; see occuspytial_amd.distributed)
    _ARRAY_FIELDS = ('X', 'site_id', 'site_ptr', 'W', 'y', 'obs_site', 'z0', 'a_mu', 'a_prec', 'b_mu', 'b_prec')

    def to_arrays(self):
        """All fixed inputs as a flat ``name -> ndarray`` dict (no Python containers)."""
        d = {name: np.ascontiguousarray(getattr(self, name)) for name in self._ARRAY_FIELDS}
        d['Q_indptr'] = np.ascontiguousarray(self.Q.indptr, dtype=np.int64)
        d['Q_indices'] = np.ascontiguousarray(self.Q.indices, dtype=np.int64)
        d['Q_data'] = np.ascontiguousarray(self.Q.data, dtype=np.float64)
        d['scalars'] = np.array([self.tau_rate, self.tau_shape], dtype=np.float64)
        if self.rsr is not None:
            d['rsr_K'], d['rsr_Q'], d['rsr_E'] = self.rsr['K'], self.rsr['Q'], self.rsr['E']
        if self.prior_factor is not None:
            d['prior_factor'] = self.prior_factor
        return d

    @classmethod
    def from_arrays(cls, d):
        """Rebuild from :meth:`to_arrays` output without re-validating (the sender validated)."""
        self = cls.__new__(cls)
        for name in cls._ARRAY_FIELDS:
            setattr(self, name, np.ascontiguousarray(d[name]))
        self.n, self.p = self.X.shape
        self.q = self.W.shape[1]
        self.S = self.site_id.size
        self.R = int(self.site_ptr[-1])
        self.Q = sparse.csr_matrix((d['Q_data'], d['Q_indices'], d['Q_indptr']), shape=(self.n, self.n))
        self.tau_rate, self.tau_shape = (float(v) for v in d['scalars'])
        self.surveyed = self.site_id.tolist()
        flag = self.obs_site.astype(bool)
        self.obs = self.site_id[flag].tolist()
        self.not_obs = self.site_id[~flag].tolist()
        mask = np.zeros(self.n, dtype=bool)
        mask[self.site_id] = True
        self.not_surveyed = np.flatnonzero(~mask).tolist()
        self.hparams = dict(tau_rate=self.tau_rate, tau_shape=self.tau_shape, a_mu=self.a_mu,
                            a_prec=self.a_prec, b_mu=self.b_mu, b_prec=self.b_prec)
        self.prior_factor = np.ascontiguousarray(d['prior_factor']) if 'prior_factor' in d else None
        self.rsr = None
        if 'rsr_K' in d:
            K = np.ascontiguousarray(d['rsr_K'])
            self.rsr = {'K': K, 'Q': np.ascontiguousarray(d['rsr_Q']), 'E': np.ascontiguousarray(d['rsr_E']), 'dim': K.shape[1]}
        return self

    def _set_hyperparams(self, hparams):
        # defaults: gibbs/base.py:177-186
        hp = {
            'tau_rate': 0.005,
            'tau_shape': 0.5 + 0.5 * (self.n - 1),
            'a_mu': np.zeros(self.q),
            'a_prec': np.eye(self.q) / 10,
            'b_mu': np.zeros(self.p),
            'b_prec': np.eye(self.p) / 10,
        }
        if hparams:
            # the reference sets user keys verbatim and nothing else (base.py:154-157, 172-175):
            # a partial dict there fails later with AttributeError; defaults fill the gaps here.
            hp.update(hparams)
        self.hparams = hp
        self.tau_rate = float(hp['tau_rate'])
        self.tau_shape = float(hp['tau_shape'])
        self.a_mu = np.ascontiguousarray(hp['a_mu'], dtype=np.float64)
        self.a_prec = np.ascontiguousarray(hp['a_prec'], dtype=np.float64)
        self.b_mu = np.ascontiguousarray(hp['b_mu'], dtype=np.float64)
        self.b_prec = np.ascontiguousarray(hp['b_prec'], dtype=np.float64)
        if self.a_mu.shape != (self.q,) or self.a_prec.shape != (self.q, self.q):
            raise ValueError('a_mu / a_prec do not match the number of detection covariates')
        if self.b_mu.shape != (self.p,) or self.b_prec.shape != (self.p, self.p):
            raise ValueError('b_mu / b_prec do not match the number of occupancy covariates')
        if not (self.tau_shape > 0 and self.tau_rate > 0):
            raise ValueError('tau_shape and tau_rate must be positive')


def _verify_spatial_precision(Q):
    """ICAR precision check: ``Q 1 = 0``, symmetric, non-positive off-diagonals.

    Replaces the reference's shift-invert ``eigsh`` test (``gibbs/base.py:166-170``), which rejects
    valid lattices from 100 columns up (lambda_2 of a 100-wide rook lattice is 9.9e-4, below its
    shift) and costs a sparse factorisation.  For an ICAR precision ``D - W`` the zero row sums ARE
    the singularity, so the test is exact; the exception text is the reference's.
    """
    scale = abs(Q).sum(axis=1).max() if Q.nnz else 0.0
    rowsum = np.abs(np.asarray(Q.sum(axis=1)).ravel()).max() if Q.nnz else 1.0
    if not (scale > 0) or rowsum > 1e-10 * scale:
        raise ValueError('Spatial precision matrix Q must be singular.')
    asym = abs(Q - Q.T)
    if asym.nnz and asym.max() > 1e-12 * scale:
        raise ValueError('Spatial precision matrix Q must be symmetric.')
    off = Q - sparse.diags(Q.diagonal())
    if off.nnz and off.max() > 0:
        raise ValueError('Spatial precision matrix Q must have non-positive off-diagonal entries '
                         '(Q = D - W with non-negative weights W).')


def _is_singular_psd(Q):
    """Symmetric, positive semi-definite and singular (dense eigenvalues; small problems only)."""
    if Q.shape[0] > DENSE_PRIOR_MAX_SITES:
        return False
    A = Q.toarray()
    if np.abs(A - A.T).max() > 1e-12 * max(np.abs(A).max(), 1e-300):
        return False
    s = np.linalg.eigvalsh(A)
    return s[0] > -1e-8 * s[-1] and s[0] < 1e-8 * s[-1]


def dense_prior_factor(Q):
    """The reference's eigenfactor of the spatial precision (``_EtaICARPosterior.__init__``, gibbs/logit.py:64-67):
    ``E = U[:, 1:] * sqrt(s[1:])`` from the dense ``eigh`` of Q, so that ``E E' = Q`` and ``E eps ~ N(0, Q)``.
    Requires Q symmetric, positive semi-definite and singular (its smallest eigenvalue numerically zero -- the
    condition the reference checks with shift-invert ``eigsh``, base.py:166-170 -- and none negative)."""
    n = Q.shape[0]
    if n > DENSE_PRIOR_MAX_SITES:
        raise ValueError(f'the dense prior factor is n x (n - 1) doubles: at most {DENSE_PRIOR_MAX_SITES} sites '
                         f'(an ICAR precision D - W needs no factor: the edge form is used)')
    A = Q.toarray() if sparse.issparse(Q) else np.asarray(Q, dtype=np.float64)
    scale = max(np.abs(A).max(), 1e-300)
    if np.abs(A - A.T).max() > 1e-12 * scale:
        raise ValueError('Spatial precision matrix Q must be symmetric.')
    s, u = np.linalg.eigh(A)
    if s[0] >= 1e-8 * s[-1] or not s[-1] > 0:
        raise ValueError('Spatial precision matrix Q must be singular.')
    if s[0] < -1e-8 * s[-1]:
        raise ValueError('Spatial precision matrix Q must be positive semi-definite.')
    return np.ascontiguousarray(u[:, 1:] * np.sqrt(np.clip(s[1:], 0.0, None)))


def default_start(rng, prob):
    """Default starting values of one chain, drawn from ``rng`` in the reference's order
    (``gibbs/base.py:199-212``): tau ~ Gamma(0.5, scale 1/tau_rate); eta ~ N(0, I) centred; alpha and
    beta from ``multivariate_normal(mu, 100 * prec, method='cholesky')`` (``100 * prec`` is used as
    a covariance there)."""
    tau = rng.gamma(0.5, 1 / prob.tau_rate)
    eta = rng.standard_normal(prob.n)
    eta = eta - eta.mean()
    alpha = rng.multivariate_normal(prob.a_mu, 100 * prob.a_prec, method='cholesky')
    beta = rng.multivariate_normal(prob.b_mu, 100 * prob.b_prec, method='cholesky')
    return dict(alpha=alpha, beta=beta, tau=tau, eta=eta)


def chain_generators(random_state, n_chains):
    """Generators of chains 0..n_chains-1 exactly as the reference seeds them: chain 0 owns
    ``SFC64(SeedSequence(seed))``; chain k >= 1 the k-th child spawned from it
    (``gibbs/base.py:293-306`` called from ``gibbs/parallel.py:20-23``)."""
    parent = np.random.SFC64(random_state)
    children = parent.seed_seq.spawn(max(n_chains - 1, 0))
    return [np.random.default_rng(parent)] + [np.random.default_rng(np.random.SFC64(c)) for c in children]
