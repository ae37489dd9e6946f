"""Sampler base class: the public API of the reference's ``occuspytial/gibbs/base.py``.

``GibbsBase`` keeps the reference's constructor, attributes (``state``, ``fixed``, ``dists``, ``rng``,
``chain``), ``sample`` / ``copy`` / ``step`` contract and error messages.  What changes is where the
work happens: a subclass that provides ``_run_chains`` (as :class:`LogitICARGibbs` does) hands all
chains to the HIP engine in one batch instead of forking one process per chain
(reference ``gibbs/parallel.py:38-41``).
"""
import numpy as np
from scipy.sparse import csc_matrix, isspmatrix_csc

from .._problem import FlatProblem, default_start
from ..chain import Chain
from ..data import Data
from ..posterior import PosteriorParameter
from ..utils import get_generator
from .parallel import sample_parallel
from .state import FixedState, State


class _GibbsState(State):
    """Sampler state whose ``posteriors`` are the recorded parameters (reference base.py:15-27)."""

    _posterior_names = ('alpha', 'beta', 'tau')

    @property
    def posteriors(self):
        return {name: self.__dict__[name] for name in self._posterior_names}


class GibbsBase:
    """Base class of the Gibbs samplers for spatial occupancy models.

    Parameters follow the reference (``base.py:30-82``): ``Q`` spatial precision (sparse or dense),
    ``W`` / ``y`` dictionaries keyed by surveyed site, ``X`` the ``n x p`` occupancy design,
    ``hparams`` optional hyper-parameter dict, ``random_state`` None | int | SeedSequence.
    """

    def __init__(self, Q, W, X, y, hparams=None, random_state=None):
        self.W = Data(W)
        self.X = X
        self.y = Data(y)
        self.rng = get_generator(random_state)

    def step(self):
        raise NotImplementedError(f'{self.__class__.__name__} must implement a `step` method.')

    # ------------------------------------------------------------------ configuration (base.py:107-186)
    def _configure(self, Q, hparams, verify_precision=True, prior_draw='auto', **kwargs):
        prob = FlatProblem(Q, self.W._data, self.X, self.y._data, hparams, check_singular=verify_precision, prior_draw=prior_draw)
        self._problem = prob

        self.state = _GibbsState()
        self.state.z = prob.z0.copy()
        self.state.k = self.state.z - 0.5

        fixed = FixedState()
        fixed.Q = Q if isspmatrix_csc(Q) else csc_matrix(Q)
        fixed.n = prob.n
        fixed.ones = np.ones(prob.n)
        fixed.not_surveyed = prob.not_surveyed
        fixed.not_obs = prob.not_obs
        fixed.obs = prob.obs
        fixed.n_no = len(prob.not_obs)
        fixed.n_ns = len(prob.not_surveyed)
        fixed.W_not_obs = self.W[prob.not_obs] if prob.not_obs else np.zeros((0, prob.q))
        fixed.visits_not_obs = self.W.visits(prob.not_obs)
        sections = np.cumsum(fixed.visits_not_obs, dtype=np.int64)
        fixed.stacked_w_indices = np.concatenate([[0], sections])[:-1].astype(np.int64)
        if hparams:
            # user keys are set verbatim first, like base.py:172-175 (tests read them back unchanged)
            for key, value in hparams.items():
                setattr(fixed, key, value)
        for key in ('tau_rate', 'tau_shape', 'a_mu', 'a_prec', 'b_mu', 'b_prec'):
            if key not in fixed.__dict__:
                setattr(fixed, key, prob.hparams[key])
        fixed.a_prec_by_mu = prob.a_prec @ prob.a_mu
        fixed.b_prec_by_mu = prob.b_prec @ prob.b_mu
        self.fixed = fixed
        self.dists = FixedState()

    # ------------------------------------------------------------------ start values (base.py:188-212)
    def _initialize_posterior_state(self, start=None):
        if start is None:
            self._initialize_default_start(self.state)
        else:
            self.state.alpha = start['alpha']
            self.state.beta = start['beta']
            self.state.tau = start['tau']
            self.state.eta = start['eta']
            self.state.spatial = self.state.eta

    def _initialize_default_start(self, state):
        """Default start, drawn from ``self.rng`` in the reference's order (base.py:199-212):
        gamma for tau, n normals for eta (centred), then alpha and beta from
        ``multivariate_normal(mu, 100 * prec, method='cholesky')`` -- ``100 * prec`` used as a
        covariance, as the reference does."""
        st = default_start(self.rng, self._problem)
        state.tau, state.eta, state.spatial = st['tau'], st['eta'], st['eta']
        state.alpha, state.beta = st['alpha'], st['beta']
        return state

    # ------------------------------------------------------------------ generic single-chain loop (base.py:214-241)
    def _run(self, size, burnin=0, start=None, chains=2, progressbar=True, pos=0):
        from tqdm.auto import tqdm

        self._initialize_posterior_state(start)
        dims = {'alpha': np.size(self.state.alpha), 'beta': np.size(self.state.beta), 'tau': 1}
        self.chain = Chain(dims, size - burnin)
        for i in tqdm(range(size), total=size, disable=not progressbar, position=pos):
            self.step()
            if i >= burnin:
                self.chain.append(self.state.posteriors)
        return self.chain

    def sample(self, size, burnin=0, start=None, chains=2, progressbar=True):
        """Draw ``size`` iterations per chain and return the kept ``alpha``, ``beta``, ``tau`` draws.

        Same contract as the reference (``base.py:243-291``): ``burnin < size`` else ``ValueError``;
        ``chains >= 1`` else ``ValueError``; ``start`` may give ``alpha``, ``beta``, ``tau``, ``eta``;
        the result indexes as ``out['alpha'] -> (chains, size - burnin, q)`` etc.
        """
        if burnin >= size:
            raise ValueError('burnin value cannot be larger than sample size')
        if chains < 1:
            raise ValueError('chains must a positive integer.')
        samples = sample_parallel(self, size=size, burnin=burnin, chains=chains, start=start,
                                  progressbar=progressbar)
        return PosteriorParameter(*samples)

    def copy(self):
        """A shallow copy with its own generator spawned from this one's seed sequence
        (reference ``base.py:293-306``: child ``spawn_key=(j,)``, the counter persists across calls)."""
        out = type(self).__new__(self.__class__)
        out.__dict__.update(self.__dict__)
        seed_seq = self.rng.bit_generator.seed_seq.spawn(1)[0]
        out.__dict__['rng'] = get_generator(seed_seq)
        # per-object caches must not be shared with the copy
        out.__dict__.pop('_engine', None)
        return out
