"""Thin object wrapper over the C ABI: one ``Engine`` = one device-resident batch of chains."""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import _lib


def _ptr(a):
    return C.c_void_p(a.ctypes.data)


def _keys_array(keys):
    return (C.c_uint64 * len(keys))(*[int(v) & (2 ** 64 - 1) for v in keys])


def problem_struct(prob):
    """``occ_problem`` of a :class:`FlatProblem` plus the arrays that must outlive the call."""
    Q = prob.Q
    k = dict(
        indptr=np.ascontiguousarray(Q.indptr, dtype=np.int32),
        indices=np.ascontiguousarray(Q.indices, dtype=np.int32),
        data=np.ascontiguousarray(Q.data, dtype=np.float64),
        site_id=np.ascontiguousarray(prob.site_id, dtype=np.int32),
        site_ptr=np.ascontiguousarray(prob.site_ptr, dtype=np.int32),
    )
    pb = _lib.OccProblem(
        n=prob.n, n_surveyed=prob.S, n_rows=prob.R, p=prob.p, q=prob.q,
        q_indptr=_ptr(k['indptr']), q_indices=_ptr(k['indices']), q_data=_ptr(k['data']),
        X=_ptr(prob.X), site_id=_ptr(k['site_id']), site_ptr=_ptr(k['site_ptr']),
        W=_ptr(prob.W), y=_ptr(prob.y), a_mu=_ptr(prob.a_mu), a_prec=_ptr(prob.a_prec),
        b_mu=_ptr(prob.b_mu), b_prec=_ptr(prob.b_prec), tau_rate=prob.tau_rate, tau_shape=prob.tau_shape)
    rsr = getattr(prob, 'rsr', None)
    if rsr is not None:   # LogitRSRGibbs: eta = K theta
        pb.rsr_dim = int(rsr['dim'])
        pb.rsr_K, pb.rsr_Q, pb.rsr_E = _ptr(rsr['K']), _ptr(rsr['Q']), _ptr(rsr['E'])
    elif getattr(prob, 'prior_factor', None) is not None:   # the reference's form of the prior draw: u = E eps
        k['prior_factor'] = np.ascontiguousarray(prob.prior_factor, dtype=np.float64)
        pb.prior_factor = _ptr(k['prior_factor'])
        pb.prior_factor_cols = k['prior_factor'].shape[1]
    return pb, k


class ProblemMeta:
    """Sizes and hyper-parameters of a problem -- all a rank of a distributed group needs on the HOST when the design
    arrays themselves arrive on its device by broadcast (start values, output shapes)."""

    FIELDS = ('n', 'p', 'q', 'S', 'R', 'tau_rate', 'tau_shape', 'a_mu', 'a_prec', 'b_mu', 'b_prec')
    rsr = None

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw[f])

    @classmethod
    def of(cls, prob):
        return cls(**{f: getattr(prob, f) for f in cls.FIELDS})

    def to_dict(self):
        return {f: getattr(self, f) for f in self.FIELDS}


_LIVE = weakref.WeakSet()   # engines not yet closed: closed at interpreter exit, before the library's own exit hook runs


@atexit.register
def _close_live_engines():
    """An engine some object still holds when the interpreter exits (an uncollected sampler, an exception path) would keep its
    CU-masked streams alive into the HIP runtime's -- and a profiler's -- finalisers (a rocprofv3 crash seen in round 3)."""
    for eng in list(_LIVE):
        try:
            eng.close()
        except Exception:
            pass


class Engine:
    """Device-resident sampler state for ``n_chains`` chains of one :class:`FlatProblem`.

    ``keys`` are the 64-bit Philox keys of the chains.  ``device`` is the HIP device ordinal.
    """

    def __init__(self, prob, keys, device=0):
        lib = _lib.load()
        pb, keep = problem_struct(prob)
        karr = _keys_array(keys)
        h = C.c_void_p()
        code = lib.occ_create(C.byref(pb), len(keys), karr, int(device), C.byref(h))
        _lib.raise_for(code, None)
        self._adopt(lib, h, prob, keys, device)
        del keep

    def _adopt(self, lib, handle, prob, keys, device):
        self._h = handle
        self._lib = lib
        self.prob = prob
        self.rsr = getattr(prob, 'rsr', None)
        self.n_chains = len(keys)
        self.device = int(device)
        self.keys = [int(v) & (2 ** 64 - 1) for v in keys]
        _LIVE.add(self)
        return self

    @classmethod
    def group(cls, prob, keys_per_device, devices):
        """One engine per entry of ``devices`` (``occ_create_group``): the problem is laid out once, uploaded to
        ``devices[0]`` and broadcast to the others device-to-device (RCCL).  ``keys_per_device[g]`` are the Philox keys
        of the chains that live on ``devices[g]``.  Each engine is then driven by its own host thread."""
        lib = _lib.load()
        pb, keep = problem_struct(prob)
        G = len(devices)
        if G < 1 or len(keys_per_device) != G or any(len(k) < 1 for k in keys_per_device):
            raise ValueError('every device of a group needs at least one chain')
        dev = (C.c_int32 * G)(*[int(d) for d in devices])
        cnt = (C.c_int32 * G)(*[len(k) for k in keys_per_device])
        karr = _keys_array([k for ks in keys_per_device for k in ks])
        hs = (C.c_void_p * G)()
        code = lib.occ_create_group(C.byref(pb), G, dev, cnt, karr, hs)
        _lib.raise_for(code, None)
        del keep
        return [cls.__new__(cls)._adopt(lib, C.c_void_p(hs[g]), prob, keys_per_device[g], devices[g]) for g in range(G)]

    @classmethod
    def distributed(cls, prob, comm, keys, root=0):
        """This process's engine of a one-process-per-GPU group (``occ_create_distributed``): ``prob`` is the full
        :class:`FlatProblem` on rank ``root`` and a :class:`ProblemMeta` (sizes and hyper-parameters only) elsewhere;
        the design arrays reach the other ranks' devices by RCCL broadcast, never their hosts.  ``comm``: an
        ``occuspytial_amd.distributed.RcclComm``."""
        lib = _lib.load()
        h = C.c_void_p()
        if comm.rank == root:
            pb, keep = problem_struct(prob)
            code = lib.occ_create_distributed(C.byref(pb), comm.handle, root, len(keys), _keys_array(keys), C.byref(h))
            del keep
        else:
            code = lib.occ_create_distributed(None, comm.handle, root, len(keys), _keys_array(keys), C.byref(h))
        _lib.raise_for(code, None)
        return cls.__new__(cls)._adopt(lib, h, prob, keys, comm.device)

    @property
    def transport(self):
        """How this engine's fixed arrays reached its device (``occ_group_transport``)."""
        return self._lib.occ_group_transport(self._h).decode()

    def synchronize(self):
        self._check(self._lib.occ_synchronize(self._h))

    def close(self):
        if getattr(self, '_h', None):
            self._lib.occ_destroy(self._h)
            self._h = None
            _LIVE.discard(self)

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


DRAW_KINDS = {'pg1': 0, 'std_gamma': 1, 'normal': 2, 'uniform': 3, 'wave_sum_check': 4}


def device_draw(kind, param=None, n=None, key=1, it=0, stream=1, device=0):
    """Variates of the engine's own generators drawn on the device (``occ_draw``): element ``i`` comes from the
    sub-stream ``(key, i, it, stream)`` exactly as the kernels draw it.  ``kind``: ``'pg1'`` (``param`` = z),
    ``'std_gamma'`` (``param`` = shape), ``'normal'``, ``'uniform'`` (``n`` draws); ``'wave_sum_check'``: a device self-test
    (tests/test_gpu_rng.py), NaN where the forms of the engine's wave sum disagree."""
    lib = _lib.load()
    par = None
    if kind in ('pg1', 'std_gamma', 'wave_sum_check'):
        par = np.ascontiguousarray(param, dtype=np.float64).ravel()
        n = par.size
    out = np.empty(int(n))
    code = lib.occ_draw(int(device), DRAW_KINDS[kind], int(key) & (2 ** 64 - 1), int(it), int(stream), int(n),
                        _ptr(par) if par is not None else None, _ptr(out))
    _lib.raise_for(code, None)
    return out


class EngineGroup:
    """Several :class:`Engine` s -- one per GPU of this process -- behind the interface of one: chain ``c`` lives on
    ``devices[c % G]`` (the reference's ``gibbs/parallel.py:20-41`` gives every chain a process of its
    own).  The problem is uploaded once and broadcast device to device (``Engine.group``); ``run`` drives every device
    from its own host thread (the C ABI holds no global state and ctypes releases the GIL)."""

    def __init__(self, prob, keys, devices, engine_factory=None):
        devices = [int(d) for d in devices]
        G = min(len(devices), len(keys))
        self.devices = devices[:G]
        self.prob = prob
        self.rsr = getattr(prob, 'rsr', None)
        self.n_chains = len(keys)
        self.where = [(c % G, c // G) for c in range(self.n_chains)]        # chain -> (engine, local index)
        per_dev = [[keys[c] for c in range(self.n_chains) if c % G == g] for g in range(G)]
        if engine_factory is None:
            self.engines = Engine.group(prob, per_dev, self.devices)
        else:   # tests: any object with the Engine interface
            self.engines = [engine_factory(prob, per_dev[g], self.devices[g]) for g in range(G)]
        self.keys = [int(k) & (2 ** 64 - 1) for k in keys]
        self.device = self.devices[0]

    @property
    def transport(self):
        return getattr(self.engines[0], 'transport', 'n/a')

    def _each(self, fn):
        """``fn(engine)`` on every engine, one host thread per engine; results in engine order."""
        if len(self.engines) == 1:
            return [fn(self.engines[0])]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(self.engines)) as pool:
            return list(pool.map(fn, self.engines))

    def close(self):
        for e in self.engines:
            e.close()

    def set_keys(self, keys):
        if len(keys) != self.n_chains:
            raise ValueError('one key per chain is required')
        G = len(self.engines)
        for g, e in enumerate(self.engines):
            e.set_keys([keys[c] for c in range(self.n_chains) if c % G == g])
        self.keys = [int(k) & (2 ** 64 - 1) for k in keys]

    def set_start(self, chain, alpha, beta, tau, eta):
        g, i = self.where[chain]
        self.engines[g].set_start(i, alpha, beta, tau, eta)

    def get(self, name, chain=0):
        g, i = self.where[chain]
        return self.engines[g].get(name, i)

    def set(self, name, value, chain=0):
        g, i = self.where[chain]
        self.engines[g].set(name, value, i)

    def step(self):
        self._each(lambda e: e.step())

    def synchronize(self):
        for e in self.engines:
            e.synchronize()

    def run(self, n_iter, burnin=0):
        parts = self._each(lambda e: e.run(n_iter, burnin))
        keep = max(n_iter - burnin, 0)
        a = np.zeros((self.n_chains, keep, self.prob.q))
        b = np.zeros((self.n_chains, keep, self.prob.p))
        t = np.zeros((self.n_chains, keep))
        for c, (g, i) in enumerate(self.where):
            a[c], b[c], t[c] = parts[g][0][i], parts[g][1][i], parts[g][2][i]
        return a, b, t

    def stats(self):
        per = [e.stats() for e in self.engines]
        out = dict(per[0])
        out['n_chains'] = self.n_chains
        out['devices'] = list(self.devices)
        out['per_device'] = per
        return out

    def checkpoint(self):
        parts = [e.checkpoint() for e in self.engines]
        out = {'n_chains': np.int64(self.n_chains), 'shape': parts[0]['shape']}
        for name in parts[0]:
            if name in ('n_chains', 'shape'):
                continue
            out[name] = np.stack([np.asarray(parts[g][name])[i] for g, i in self.where])
        return out

    def restore(self, ckpt):
        if int(ckpt['n_chains']) != self.n_chains:
            raise ValueError('checkpoint holds %d chains, this engine %d' % (int(ckpt['n_chains']), self.n_chains))
        G = len(self.engines)
        for g, e in enumerate(self.engines):
            idx = [c for c in range(self.n_chains) if c % G == g]
            part = {k: (np.asarray(v)[idx] if k not in ('n_chains', 'shape') else v) for k, v in ckpt.items()}
            part['n_chains'] = np.int64(len(idx))
            e.restore(part)
        self.keys = [int(k) for k in np.asarray(ckpt['keys'])]
