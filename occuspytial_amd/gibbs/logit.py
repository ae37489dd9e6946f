"""``LogitICARGibbs`` on the MI355X engine (API of reference ``occuspytial/gibbs/logit.py:102-266``)."""
import numpy as np

from .._engine import Engine, EngineGroup
from ..chain import Chain
from .base import GibbsBase


def _philox_key(rng):
    """64-bit key of a chain's device streams: the next raw word of the chain's own SFC64 stream,
    taken AFTER its start values were drawn, so chains and repeated runs never share a key."""
    return int(rng.bit_generator.random_raw())


class LogitICARGibbs(GibbsBase):
    r"""Gibbs sampler, logit link, ICAR spatial random effects -- computed on an AMD MI355X.

    Drop-in for the reference class of the same name (``logit.py:102-174``): same constructor
    ``(Q, W, X, y, hparams=None, random_state=None)``, same ``sample`` / ``step`` / ``copy``, same
    ``state`` / ``fixed`` attributes and error behaviour.  One iteration performs the reference's seven
    conditional updates in its order (``logit.py:254-266``): :math:`\omega_b`, :math:`\tau`,
    :math:`\eta`, :math:`\beta`, :math:`\omega_a`, :math:`\alpha`, :math:`z`.

    Differences that are visible to a user, all documented in DESIGN.md:

    * variates come from counter-based Philox streams keyed per chain, not from one sequential SFC64
      stream, so draws are equal to the reference's in distribution, not value for value;
    * the prior term of the :math:`\eta` conditional uses the edge factorisation
      :math:`Q = B^\top B` instead of a dense eigenfactor (``logit.py:66-67``): no O(n^2) memory, no
      O(n^3) set-up, for any ICAR precision (zero row sums, non-positive off-diagonals).  ``prior_draw='dense'``
      selects the reference's own form instead (dense ``eigh`` on the host, one dense matrix-vector product per
      iteration on the device); ``'auto'`` takes it by itself for a singular positive semi-definite ``Q`` that is
      not an ICAR precision, which the edge form cannot represent;
    * ``device`` selects the HIP device; all chains of one ``sample`` call run batched on it.  ``devices=[...]``
      instead fans the chains of ``sample(chains=N)`` out over several GPUs from this process, chain ``c`` on
      ``devices[c % len(devices)]`` -- the reference's one-process-per-chain fan-out (``gibbs/parallel.py:20-41``) with
      GPUs for processes: the problem is uploaded once and broadcast device to device over RCCL, every GPU is driven by
      its own host thread, and chain ``k`` still owns the generator the reference would give it, so the draws do not
      depend on how many devices share the work.
    """

    def __init__(self, Q, W, X, y, hparams=None, random_state=None, device=0, devices=None, prior_draw='auto'):
        super().__init__(Q, W, X, y, hparams, random_state)
        self.devices = [int(d) for d in devices] if devices is not None else None
        self.device = self.devices[0] if self.devices else device
        self._configure(Q, hparams, prior_draw=prior_draw)

    def _configure(self, Q, hparams, prior_draw='auto'):
        super()._configure(Q, hparams, prior_draw=prior_draw)

    # ------------------------------------------------------------------ engine management
    def _get_engine(self, keys):
        eng = self.__dict__.get('_engine')
        if eng is None or eng.n_chains != len(keys):
            if eng is not None:
                eng.close()
            if self.devices and len(self.devices) > 1 and len(keys) > 1:   # chains sharded over the GPUs of this process
                eng = EngineGroup(self._problem, keys, self.devices)
            else:
                eng = Engine(self._problem, keys, device=self.device)
            self.__dict__['_engine'] = eng
        else:
            eng.set_keys(keys)
        return eng

    def _push_start(self, eng, chain, state):
        eng.set_start(chain, np.atleast_1d(np.asarray(state.alpha, dtype=float)),
                      np.atleast_1d(np.asarray(state.beta, dtype=float)), float(state.tau),
                      np.asarray(state.eta, dtype=float))

    def _pull_state(self, eng, chain=0):
        """Mirror the device state of one chain into ``self.state`` (reference attribute names)."""
        st = self.state
        st.alpha = eng.get('alpha', chain)
        st.beta = eng.get('beta', chain)
        st.tau = float(eng.get('tau', chain))
        if self._problem.rsr is None:
            st.eta = eng.get('eta', chain)
            st.spatial = st.eta
        else:   # reduced-rank model: eta holds the basis coefficients, spatial = K eta (logit.py:484-485)
            st.eta = eng.get('theta', chain)
            st.spatial = eng.get('eta', chain)
        st.z = eng.get('z', chain)
        st.k = st.z - 0.5
        st.omega_b = eng.get('omega_b', chain)
        prob = self._problem
        exists_flag = eng.get('exists', chain).astype(bool)
        # reference order (logit.py:187-188): sites with a detection first, then newly occupied ones
        obs = prob.obs_site.astype(bool)
        order = np.concatenate([np.flatnonzero(obs), np.flatnonzero(exists_flag & ~obs)])
        st.exists = [prob.surveyed[i] for i in order]
        rows = [np.arange(prob.site_ptr[i], prob.site_ptr[i + 1]) for i in order]
        rows = np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)
        st.omega_a = eng.get('omega_a', chain)[rows]
        st.W = prob.W[rows]

    # ------------------------------------------------------------------ public stepping (logit.py:254-266)
    def step(self):
        """One Gibbs iteration of this sampler's own chain (its state must have start values:
        ``sample`` sets them, or call ``_initialize_posterior_state`` first, as ``_run`` does)."""
        if 'alpha' not in self.state.__dict__:
            self._initialize_posterior_state(None)
        eng = self.__dict__.get('_engine')
        if eng is None or eng.n_chains != 1 or not self.__dict__.get('_stepping'):
            eng = self._get_engine([_philox_key(self.rng)])
            self._push_start(eng, 0, self.state)
            eng.set('z', self.state.z, 0)
            self.__dict__['_stepping'] = True
        eng.step()
        self._pull_state(eng, 0)

    # ------------------------------------------------------------------ checkpoint / resume
    def checkpoint(self, path=None):
        """State of every chain of the last :meth:`sample` call, sufficient to continue each of them exactly
        (``Engine.checkpoint``).  Returns a dict of arrays; ``path`` additionally writes it as ``.npz``."""
        eng = self.__dict__.get('_engine')
        if eng is None:
            raise RuntimeError('nothing to checkpoint: call sample() first')
        ckpt = eng.checkpoint()
        if path is not None:
            np.savez(path, **ckpt)
        return ckpt

    def resume(self, checkpoint, size, progressbar=True):
        """Continue the chains of ``checkpoint`` (a dict from :meth:`checkpoint` or the path of its ``.npz``)
        for ``size`` more iterations on this sampler's problem.  Returns a ``PosteriorParameter`` of the new
        draws; every chain's ``Chain`` is the continuation (use ``Chain.expand`` / ``append`` to join them to
        earlier draws).  The result equals the tail of an uninterrupted run bit for bit."""
        from ..posterior import PosteriorParameter
        from tqdm.auto import tqdm
        if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, '__fspath__'):
            with np.load(checkpoint) as f:
                checkpoint = {k: f[k] for k in f.files}
        if size < 1:
            raise ValueError('size must be a positive integer')
        C = int(checkpoint['n_chains'])
        self.__dict__['_stepping'] = False
        eng = self._get_engine([int(k) for k in np.asarray(checkpoint['keys'])])
        eng.restore(checkpoint)
        alpha = np.zeros((C, size, self._problem.q))
        beta = np.zeros((C, size, self._problem.p))
        tau = np.zeros((C, size))
        bar = tqdm(total=size, disable=not progressbar)
        chunk = size if not progressbar else max(1, min(size, max(16, size // 25)))
        done = 0
        while done < size:
            step = min(chunk, size - done)
            alpha[:, done:done + step], beta[:, done:done + step], tau[:, done:done + step] = eng.run(step, 0)
            done += step
            bar.update(step)
        bar.close()
        chains = [Chain._from_arrays({'alpha': alpha[c], 'beta': beta[c], 'tau': tau[c]}) for c in range(C)]
        self.chain = chains[0]
        self._pull_state(eng, 0)
        return PosteriorParameter(*chains)

    # ------------------------------------------------------------------ batched chains
    def _run_chains(self, samplers, size, burnin=0, start=None, progressbar=True):
        """All chains of one ``sample`` call as one device batch.

        Mirrors ``GibbsBase._run`` (base.py:214-241) per chain: start values from the chain's own
        generator (or the ``start`` dict), then ``size`` iterations keeping those ``>= burnin``.
        """
        from tqdm.auto import tqdm

        for s in samplers:
            if s is not self:  # copies share `state` by reference in the reference too; give each its own
                s.__dict__['state'] = type(self.state)(**self.state.__dict__)
            s._initialize_posterior_state(start)
        keys = [_philox_key(s.rng) for s in samplers]
        self.__dict__['_stepping'] = False
        eng = self._get_engine(keys)
        z0 = self._problem.z0
        for c, s in enumerate(samplers):
            self._push_start(eng, c, s.state)
            eng.set('z', z0, c)

        C = len(samplers)
        keep = size - burnin
        alpha = np.zeros((C, keep, self._problem.q))
        beta = np.zeros((C, keep, self._problem.p))
        tau = np.zeros((C, keep))
        bars = [tqdm(total=size, disable=not progressbar, position=pos) for pos in range(C)]
        chunk = size if not progressbar else max(1, min(size, max(16, size // 25)))
        done = kept = 0
        while done < size:
            step = min(chunk, size - done)
            b = min(max(burnin - done, 0), step)
            if b == step:  # the whole chunk is burn-in: run it, keep only its last draw, drop it
                eng.run(step, step - 1)
            else:
                a_, b_, t_ = eng.run(step, b)
                m = step - b
                alpha[:, kept:kept + m], beta[:, kept:kept + m], tau[:, kept:kept + m] = a_, b_, t_
                kept += m
            done += step
            for bar in bars:
                bar.update(step)
        for bar in bars:
            bar.close()

        chains = []
        for c, s in enumerate(samplers):
            ch = Chain._from_arrays({'alpha': alpha[c], 'beta': beta[c], 'tau': tau[c]})
            s.chain = ch
            chains.append(ch)
        self._pull_state(eng, 0)
        return chains


class LogitRSRGibbs(LogitICARGibbs):
    r"""Gibbs sampler, logit link, reduced-rank (RSR) spatial random effects -- computed on an AMD MI355X.

    Drop-in for the reference class of the same name (``logit.py:340-485``): ``LogitRSRGibbs(Q, W, X, y,
    hparams=None, random_state=None, r=0.5, q=None)``.  The spatial effects are :math:`K\theta` with
    :math:`K` the eigenvectors of the Moran operator whose eigenvalues are at least ``r`` (or the ``q`` leading
    ones); ``state.eta`` holds :math:`\theta`, ``state.spatial`` holds :math:`K\theta`, ``fixed.Q`` is
    :math:`K^\top Q K`, ``fixed.K`` is :math:`K`, ``fixed.q`` its number of columns, and the default
    ``tau_shape`` becomes ``0.5 + 0.5 q`` -- all as in the reference.  The basis is computed on the host with dense
    n x n linear algebra, once (as the reference does); per iteration the device forms
    :math:`K^\top\Omega K + \tau K^\top QK` and solves the q x q system (``csrc/occ_rsr.hpp``: in LDS and registers up to
    128 basis columns, panel by panel in device memory up to 4096 -- the reference's default threshold keeps about 13 % of a
    lattice's sites: 1 280 columns at 100x100).  ``device`` selects the HIP device.
    """

    def __init__(self, Q, W, X, y, hparams=None, random_state=None, r=0.5, q=None, device=0, devices=None):
        super().__init__(Q, W, X, y, hparams, random_state, device=device, devices=devices, prior_draw='edge')
        self._configure_rsr(r, q, hparams)

    def _configure_rsr(self, r, q, hparams):
        rsr = self._problem.enable_rsr(r=r, q=q, default_tau_shape=not hparams)
        if rsr['dim'] > 4096:
            raise ValueError(f'{rsr["dim"]} basis columns selected; the device path supports at most 4096 '
                             '(raise the threshold `r` or pass `q`)')
        fixed = self.fixed
        fixed.q = rsr['dim']
        del fixed.Q
        fixed.Q = rsr['Q']
        fixed.K = rsr['K']
        if not hparams:
            del fixed.tau_shape
            fixed.tau_shape = self._problem.tau_shape

    def _initialize_default_start(self, state):
        state = super()._initialize_default_start(state)
        state.eta = self.rng.normal(scale=5, size=self.fixed.q)           # logit.py:453-455
        state.spatial = self.fixed.K @ state.eta
        return state

    def _initialize_posterior_state(self, start=None):
        if start is None:
            self._initialize_default_start(self.state)
        else:
            self.state.alpha = start['alpha']
            self.state.beta = start['beta']
            self.state.tau = start['tau']
            self.state.eta = start['eta']
            self.state.spatial = self.fixed.K @ np.asarray(self.state.eta, dtype=float)
