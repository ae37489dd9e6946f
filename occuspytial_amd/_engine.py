"""Thin object wrapper over the C ABI: one ``Engine`` = one device-resident batch of chains."""
import ctypes as C

import numpy as np

from . import _lib


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


class Engine:
    """Device-resident sampler state for ``n_chains`` chains of one :class:`FlatProblem`.

    ``keys`` are the 64-bit Philox keys of the chains.  ``device`` is the HIP device ordinal.
    """

    def __init__(self, prob, keys, device=0):
        lib = _lib.load()
        self.prob = prob
        self.n_chains = len(keys)
        Q = prob.Q
        self._keep = dict(
            indptr=np.ascontiguousarray(Q.indptr, dtype=np.int32),
            indices=np.ascontiguousarray(Q.indices, dtype=np.int32),
            data=np.ascontiguousarray(Q.data, dtype=np.float64),
            site_id=np.ascontiguousarray(prob.site_id, dtype=np.int32),
            site_ptr=np.ascontiguousarray(prob.site_ptr, dtype=np.int32),
        )
        k = self._keep
        pb = _lib.OccProblem(
            n=prob.n, n_surveyed=prob.S, n_rows=prob.R, p=prob.p, q=prob.q,
            q_indptr=_ptr(k['indptr']), q_indices=_ptr(k['indices']), q_data=_ptr(k['data']),
            X=_ptr(prob.X), site_id=_ptr(k['site_id']), site_ptr=_ptr(k['site_ptr']),
            W=_ptr(prob.W), y=_ptr(prob.y), a_mu=_ptr(prob.a_mu), a_prec=_ptr(prob.a_prec),
            b_mu=_ptr(prob.b_mu), b_prec=_ptr(prob.b_prec), tau_rate=prob.tau_rate, tau_shape=prob.tau_shape)
        self.rsr = getattr(prob, 'rsr', None)
        if self.rsr is not None:   # LogitRSRGibbs: eta = K theta
            pb.rsr_dim = int(self.rsr['dim'])
            pb.rsr_K, pb.rsr_Q, pb.rsr_E = _ptr(self.rsr['K']), _ptr(self.rsr['Q']), _ptr(self.rsr['E'])
        karr = (C.c_uint64 * self.n_chains)(*[int(v) & (2 ** 64 - 1) for v in keys])
        h = C.c_void_p()
        code = lib.occ_create(C.byref(pb), self.n_chains, karr, int(device), C.byref(h))
        _lib.raise_for(code, None)
        self._h = h
        self._lib = lib
        self.device = int(device)
        self.keys = [int(v) & (2 ** 64 - 1) for v in keys]

    def close(self):
        if getattr(self, '_h', None):
            self._lib.occ_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, code):
        _lib.raise_for(code, self._h)

    def set_keys(self, keys):
        if len(keys) != self.n_chains:
            raise ValueError('one key per chain is required')
        karr = (C.c_uint64 * self.n_chains)(*[int(v) & (2 ** 64 - 1) for v in keys])
        self._check(self._lib.occ_set_keys(self._h, karr))
        self.keys = [int(v) & (2 ** 64 - 1) for v in keys]

    # ---- checkpoint / resume (the reference has none) -------------------------------------
    CHECKPOINT_FIELDS = ('alpha', 'beta', 'tau', 'eta', 'z', 'xz')

    def checkpoint(self):
        """Everything a chain needs to continue exactly where it is: the state the next iteration reads
        (alpha, beta, tau, eta, z, the MINRES warm start xz), its iteration number and its Philox key.
        Variates are functions of (key, iteration, index), so a restored chain reproduces the
        uninterrupted one bit for bit; omega_b and the noise of the coming iteration are recomputed."""
        out = {'n_chains': np.int64(self.n_chains), 'keys': np.array(self.keys, dtype=np.uint64),
               'iter': np.array([int(self.get('iter', c)) for c in range(self.n_chains)], dtype=np.int64),
               'shape': np.array([self.prob.n, self.prob.p, self.prob.q, self.prob.R], dtype=np.int64)}
        fields = self.CHECKPOINT_FIELDS if self.rsr is None else ('alpha', 'beta', 'tau', 'theta', 'z')
        for name in fields:
            out[name] = np.stack([np.atleast_1d(self.get(name, c)) for c in range(self.n_chains)])
        return out

    def restore(self, ckpt):
        """Inverse of :meth:`checkpoint` (same problem, same number of chains)."""
        if int(ckpt['n_chains']) != self.n_chains:
            raise ValueError('checkpoint holds %d chains, this engine %d' % (int(ckpt['n_chains']), self.n_chains))
        if list(np.asarray(ckpt['shape'])) != [self.prob.n, self.prob.p, self.prob.q, self.prob.R]:
            raise ValueError('checkpoint belongs to a problem of different size')
        self.set_keys([int(k) for k in np.asarray(ckpt['keys'])])
        spatial = 'eta' if self.rsr is None else 'theta'   # reduced-rank model: the coefficients; eta = K theta follows
        for c in range(self.n_chains):
            self.set_start(c, ckpt['alpha'][c], ckpt['beta'][c], float(np.asarray(ckpt['tau'][c]).ravel()[0]), ckpt[spatial][c])
            self.set('z', ckpt['z'][c], c)
            if self.rsr is None:
                self.set('xz', ckpt['xz'][c], c)
            self.set('iter', float(ckpt['iter'][c]), c)

    def set_start(self, chain, alpha, beta, tau, eta):
        a = np.ascontiguousarray(alpha, dtype=np.float64)
        b = np.ascontiguousarray(beta, dtype=np.float64)
        e = np.ascontiguousarray(eta, dtype=np.float64)
        n_eta = self.prob.n if self.rsr is None else int(self.rsr['dim'])   # reduced-rank model: eta is theta
        if a.shape != (self.prob.q,) or b.shape != (self.prob.p,) or e.shape != (n_eta,):
            raise ValueError('start values have the wrong shape')
        self._check(self._lib.occ_set_start(self._h, chain, _ptr(a), _ptr(b), float(tau), _ptr(e)))

    def step(self):
        self._check(self._lib.occ_step(self._h))

    def run(self, n_iter, burnin=0):
        keep = n_iter - burnin
        C_ = self.n_chains
        a = np.zeros((C_, max(keep, 0), self.prob.q))
        b = np.zeros((C_, max(keep, 0), self.prob.p))
        t = np.zeros((C_, max(keep, 0)))
        self._check(self._lib.occ_run(self._h, n_iter, burnin, _ptr(a), _ptr(b), _ptr(t)))
        return a, b, t

    def get(self, name, chain=0):
        ln = C.c_int64(0)
        self._check(self._lib.occ_get_state(self._h, chain, name.encode(), None, 0, C.byref(ln)))
        out = np.empty(ln.value)
        self._check(self._lib.occ_get_state(self._h, chain, name.encode(), _ptr(out), out.size, C.byref(ln)))
        return out[0] if name in ('tau', 'minres_itn', 'iter') else out

    def set(self, name, value, chain=0):
        v = np.ascontiguousarray(np.atleast_1d(value), dtype=np.float64)
        self._check(self._lib.occ_set_state(self._h, chain, name.encode(), _ptr(v), v.size))

    # ---- per-conditional updates with injected variates (C ABI occ_cond_*; parity tests against the reference's fixtures)
    @staticmethod
    def _vec(a, size):
        v = np.ascontiguousarray(a, dtype=np.float64).ravel()
        if v.size != size:
            raise ValueError('wrong length: %d, expected %d' % (v.size, size))
        return v

    def cond_tau(self, gamma_variate, chain=0):
        out = C.c_double(0.0)
        self._check(self._lib.occ_cond_tau(self._h, chain, float(gamma_variate), C.byref(out)))
        return out.value

    def cond_eta(self, omega_b, eps_site, prior_term, chain=0):
        """-> (rhs, xz, eta, minres iterations) from the chain's beta, z, tau and warm start."""
        n = self.prob.n
        ob, e1, pt = self._vec(omega_b, n), self._vec(eps_site, n), self._vec(prior_term, n)
        rhs, xz, eta, itn = np.empty(n), np.empty(2 * n), np.empty(n), C.c_int32(0)
        self._check(self._lib.occ_cond_eta(self._h, chain, _ptr(ob), _ptr(e1), _ptr(pt), _ptr(rhs), _ptr(xz), _ptr(eta), C.byref(itn)))
        return rhs, xz, eta, itn.value

    def cond_beta(self, omega_b, eps, chain=0):
        ob, ep, out = self._vec(omega_b, self.prob.n), self._vec(eps, self.prob.p), np.empty(self.prob.p)
        self._check(self._lib.occ_cond_beta(self._h, chain, _ptr(ob), _ptr(ep), _ptr(out)))
        return out

    def cond_alpha(self, omega_a, eps, chain=0):
        oa, ep, out = self._vec(omega_a, self.prob.R), self._vec(eps, self.prob.q), np.empty(self.prob.q)
        self._check(self._lib.occ_cond_alpha(self._h, chain, _ptr(oa), _ptr(ep), _ptr(out)))
        return out

    def cond_z(self, u, chain=0):
        uu, out = self._vec(u, self.prob.n), np.empty(self.prob.n)
        self._check(self._lib.occ_cond_z(self._h, chain, _ptr(uu), _ptr(out)))
        return out

    def stats(self):
        st = _lib.OccStats()
        self._check(self._lib.occ_get_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    def profile(self, reps=200):
        """Average in-graph launch time per kernel kind (leaves the chains mid-solve: re-start them)."""
        counts = (C.c_int64 * _lib.N_KERNEL_KINDS)()
        total = (C.c_double * _lib.N_KERNEL_KINDS)()
        self._check(self._lib.occ_profile(self._h, reps, counts, total))
        return {k: {'launches': int(counts[i]), 'total_us': float(total[i]),
                    'avg_us': float(total[i]) / counts[i] if counts[i] else 0.0}
                for i, k in enumerate(_lib.KERNEL_KINDS)}


DRAW_KINDS = {'pg1': 0, 'std_gamma': 1, 'normal': 2, 'uniform': 3}


def device_draw(kind, param=None, n=None, key=1, it=0, stream=1, device=0):
    """Variates of the engine's own generators drawn on the device (``occ_draw``): element ``i`` comes from the
    sub-stream ``(key, i, it, stream)`` exactly as the kernels draw it.  ``kind``: ``'pg1'`` (``param`` = z),
    ``'std_gamma'`` (``param`` = shape), ``'normal'``, ``'uniform'`` (``n`` draws)."""
    lib = _lib.load()
    par = None
    if kind in ('pg1', 'std_gamma'):
        par = np.ascontiguousarray(param, dtype=np.float64).ravel()
        n = par.size
    out = np.empty(int(n))
    code = lib.occ_draw(int(device), DRAW_KINDS[kind], int(key) & (2 ** 64 - 1), int(it), int(stream), int(n),
                        _ptr(par) if par is not None else None, _ptr(out))
    _lib.raise_for(code, None)
    return out
