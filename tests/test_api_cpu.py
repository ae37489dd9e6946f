"""Host-side behaviour that needs no GPU: the C ABI library loads and exports every declared symbol,
the reference's container/utility contracts, input flattening and seeding pinned to the reference
fixtures, argument errors, and the loud failure when no HIP device is usable."""
import os
import re

import numpy as np
import pytest
from scipy import sparse
from scipy.sparse import rand

from .conftest import GOLDEN_CASES, ROOT, load_golden


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from occuspytial_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'occ_gibbs.h')).read()
    declared = set(re.findall(r'\b(occ_[a-z_]+)\s*\(', header))
    assert {'occ_create', 'occ_destroy', 'occ_run', 'occ_step', 'occ_get_state', 'occ_profile'} <= declared
    lib = _lib.load()
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name)
    # header, library and binding agree on the ABI version (struct layouts); __graft_entry__.build() checks the same
    version = int(re.search(r'#define\s+OCC_ABI_VERSION\s+(\d+)', header).group(1))
    assert lib.occ_abi_version() == version == _lib.ABI_VERSION


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    from occuspytial_amd import LogitICARGibbs, _lib
    from occuspytial_amd.utils import make_lattice_problem
    if _lib.load().occ_device_count() > 0:
        pytest.skip('a HIP device is present')
    Q, W, X, y, *_ = make_lattice_problem(6, 6, visits=2)
    s = LogitICARGibbs(Q, W, X, y, random_state=0)   # set-up is host only
    with pytest.raises(_lib.EngineUnavailable):
        s.sample(5, chains=1, progressbar=False)
    with pytest.raises(_lib.EngineUnavailable):
        s.step()


# ------------------------------------------------------------------ containers (reference tests/test_chain.py, test_data.py, test_state.py)
def test_chain_contract():
    from occuspytial_amd.chain import Chain
    c = Chain({'p1': 2, 'p2': 1}, 1)
    assert c.full.shape[1] == 3 and len(c) == 0
    c.append({'p1': [1, 2], 'p2': 3})
    assert len(c) == 1
    with pytest.raises(ValueError, match='Chain is full'):
        c.append({'p1': [1, 2], 'p2': 3})
    c.expand(1)
    c.append({'p1': [1, 2], 'p2': 3})
    assert len(c) == 2 and np.all(c['p1'] == [[1, 2], [1, 2]])
    with pytest.raises(KeyError):
        c['p3']
    assert repr(c) == "Chain(params: ('p1', 'p2'), size: 2)"
    assert Chain({'p1': 2, 'p2': 1}, size=2)['p1'].shape == (0, 2)


def test_data_contract():
    from occuspytial_amd import Data
    dic = {1: np.random.rand(5), 2: np.random.rand(4), 3: np.random.rand(2)}
    d = Data(dic)
    assert d.surveyed == [1, 2, 3] and len(d) == 3
    assert d.visits([1, 3]) == (5, 2) and d.visits((1, 3)) == (5, 2) and d.visits(3) == 2
    assert d[1] is dic[1]
    assert np.allclose(d[[1, 3]], np.concatenate((dic[1], dic[3]))) and d[(1, 3)].size == 7
    d2 = Data({1: np.random.rand(5, 2), 3: np.random.rand(2, 2)})
    assert len(d2) == 2 and d2[[1, 3]].shape == (7, 2)


def test_state_contract():
    from occuspytial_amd.gibbs.state import FixedState, State
    s = State()
    s.a, s.b = 1, 2
    with pytest.raises(TypeError, match='does not support item assignment'):
        s['b'] = 2
    s.a = 0.5
    assert s.a == 0.5 and [i for i in s] == ['a', 'b'] and s['b'] == 2
    fs = FixedState()
    fs.c = 3
    with pytest.raises(KeyError, match='cannot change attributes already set'):
        fs.c = 2


def test_posterior_parameter_shapes_and_summary():
    from occuspytial_amd.chain import Chain
    from occuspytial_amd.posterior import PosteriorParameter
    rng = np.random.default_rng(0)
    chains = []
    for _ in range(3):
        c = Chain({'alpha': 2, 'beta': 3, 'tau': 1}, 200)
        for _ in range(200):
            c.append({'alpha': rng.standard_normal(2), 'beta': rng.standard_normal(3), 'tau': rng.gamma(2.0)})
        chains.append(c)
    post = PosteriorParameter(*chains)
    assert post['alpha'].shape == (3, 200, 2) and post['beta'].shape == (3, 200, 3) and post['tau'].shape == (3, 200)
    table = post.summary
    rows = set(getattr(table, 'index', table))
    assert any('tau' in str(r) for r in rows) and len(rows) == 6
    assert PosteriorParameter(chains[0])['tau'].shape == (1, 200)


def test_diagnostics_against_known_behaviour():
    from occuspytial_amd import diagnostics as dg
    rng = np.random.default_rng(1)
    iid = rng.standard_normal((4, 2000))
    assert 0.99 < dg.rhat(iid) < 1.01
    assert 6000 < dg.ess(iid) < 10500
    ar = np.zeros((4, 2000))
    for t in range(1, 2000):
        ar[:, t] = 0.9 * ar[:, t - 1] + rng.standard_normal(4)
    assert dg.ess(ar) < 1000            # rho = 0.9 -> about n (1-rho)/(1+rho) = 420
    assert dg.rhat(iid + np.arange(4)[:, None]) > 1.3
    lo, hi = dg.hdi(iid)
    assert -2.1 < lo < -1.7 and 1.7 < hi < 2.1


# ------------------------------------------------------------------ utils (reference tests/test_utils.py)
def test_get_generator_and_lattices():
    from occuspytial_amd.utils import get_generator, make_data, rand_precision_mat
    rng, rng2 = get_generator(0), get_generator(0)
    assert isinstance(rng.bit_generator, np.random.SFC64)
    assert np.all(rng.bit_generator.state['state']['state'] == rng2.bit_generator.state['state']['state'])
    assert rand_precision_mat(2, 4, max_neighbors=4).diagonal().max() == 3
    mat = rand_precision_mat(2, 4, max_neighbors=8)
    assert mat.diagonal().max() == 5 and np.linalg.matrix_rank(mat.toarray()) == 7
    with pytest.raises(ValueError, match='neighbors should be one of {4, 8}'):
        rand_precision_mat(2, 4, max_neighbors=9)
    assert np.linalg.matrix_rank(rand_precision_mat(2, 4, rho=0.5).toarray()) == 8
    Q = rand_precision_mat(10, 5)           # docstring example of the reference (utils.py:71-84)
    assert Q.nnz == 364 and Q.dtype == np.int64 and Q.toarray()[0, :2].tolist() == [3, -1]
    data = make_data(n=150, p=3, q=2, ns=65, random_state=10)
    assert data[0].shape[0] == 150 and data[4].shape[0] == 2 and data[5].shape[0] == 3
    assert len(data[1]) == 65 and data[2].shape[1] == 3
    assert len(make_data(n=150, p=3, q=2, random_state=10)[1]) == 75
    for kwargs, msg in (({'n': 149}, 'n cant be lower than'), ({'min_v': 0}, 'min_v needs to be at least'),
                        ({'max_v': 1}, 'max_v is too small'), ({'max_v': 151}, 'max_v cant be more than n'),
                        ({'ns': 0}, 'ns should be positive'), ({'ns': 151}, 'ns cant be more than n')):
        with pytest.raises(ValueError, match=msg):
            make_data(**kwargs)


# ------------------------------------------------------------------ set-up pinned to the reference fixtures
def _inputs(g):
    n = g['X'].shape[0]
    Q = sparse.csr_matrix((g['Q_data'], g['Q_indices'], g['Q_indptr']), shape=(n, n))
    W, y, cur = {}, {}, 0
    for s, v in zip(g['sites'], g['visits']):
        W[int(s)] = g['W_flat'][cur:cur + v]
        y[int(s)] = g['y_flat'][cur:cur + v]
        cur += v
    hp = {k[3:]: g[k] for k in g if k.startswith('hp_')} or None
    return Q, W, g['X'], y, hp


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_configuration_matches_reference(case):
    """Index sets, initial z, stacked W of not-observed sites and hyper-parameters equal what the
    reference's _configure produced (gibbs/base.py:107-164)."""
    from occuspytial_amd import LogitICARGibbs
    g = load_golden(case)
    Q, W, X, y, hp = _inputs(g)
    s = LogitICARGibbs(Q, W, X, y, hparams=hp, random_state=int(g['seed']))
    f = s.fixed
    assert f.not_obs == g['cfg_not_obs'].tolist() and f.obs == g['cfg_obs'].tolist()
    assert f.not_surveyed == g['cfg_not_surveyed'].tolist()
    assert np.array_equal(s.state.z, g['cfg_z0']) and np.array_equal(s.state.k, g['cfg_z0'] - 0.5)
    assert np.array_equal(np.asarray(f.W_not_obs), g['cfg_W_not_obs'])
    assert np.array_equal(f.stacked_w_indices, g['cfg_stacked_w_indices'])
    assert f.tau_rate == float(g['cfg_tau_rate']) and f['tau_shape'] == float(g['cfg_tau_shape'])
    for k in ('a_mu', 'a_prec', 'b_mu', 'b_prec'):
        assert np.array_equal(np.asarray(f[k]), g['cfg_' + k])
    assert f.n_no == len(f.not_obs) and f.n_ns == len(f.not_surveyed) and f.n == X.shape[0]


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_default_start_and_chain_seeding_match_reference(case):
    """Same seed => the reference's own start values (base.py:199-212), bit for bit; copies own the
    generators the reference gives them (base.py:293-306)."""
    from occuspytial_amd import LogitICARGibbs
    g = load_golden(case)
    Q, W, X, y, hp = _inputs(g)
    s = LogitICARGibbs(Q, W, X, y, hparams=hp, random_state=int(g['seed']))
    s._initialize_posterior_state(None)
    assert s.state.tau == float(g['start_tau'])
    assert np.array_equal(s.state.eta, g['start_eta']) and s.state.spatial is s.state.eta
    assert np.array_equal(s.state.alpha, g['start_alpha']) and np.array_equal(s.state.beta, g['start_beta'])
    s2 = LogitICARGibbs(Q, W, X, y, hparams=hp, random_state=int(g['seed']))
    c1, c2 = s2.copy(), s2.copy()
    assert isinstance(c1, LogitICARGibbs)
    raw = np.stack([c.rng.bit_generator.random_raw(4) for c in (c1, c2)])
    assert np.array_equal(raw, g['copy_raw'])
    from occuspytial_amd._problem import chain_generators
    gens = chain_generators(int(g['seed']), 3)
    assert np.array_equal(gens[0].bit_generator.random_raw(4), g['parent_raw'])
    assert np.array_equal(np.stack([x.bit_generator.random_raw(4) for x in gens[1:]]), g['copy_raw'])


def test_flat_problem_round_trip():
    from occuspytial_amd._problem import FlatProblem
    g = load_golden('ref_queen150_ragged')
    Q, W, X, y, hp = _inputs(g)
    p = FlatProblem(Q, W, X, y, hp)
    q = FlatProblem.from_arrays(p.to_arrays())
    assert (p.Q != q.Q).nnz == 0 and p.not_obs == q.not_obs and p.obs == q.obs and p.not_surveyed == q.not_surveyed
    for name in ('X', 'W', 'y', 'site_id', 'site_ptr', 'obs_site', 'z0', 'a_prec', 'b_mu'):
        assert np.array_equal(getattr(p, name), getattr(q, name))
    assert (p.tau_rate, p.tau_shape, p.n, p.p, p.q, p.S, p.R) == (q.tau_rate, q.tau_shape, q.n, q.p, q.q, q.S, q.R)


# ------------------------------------------------------------------ errors (reference gibbs/tests/test_samplers.py)
@pytest.fixture(scope='module')
def small():
    g = load_golden('ref_queen150_ragged')
    return _inputs(g)[:4]


def test_sample_argument_errors(small):
    from occuspytial_amd import LogitICARGibbs
    s = LogitICARGibbs(*small, random_state=10)
    with pytest.raises(ValueError, match='burnin value cannot be larger than'):
        s.sample(10, burnin=11)
    with pytest.raises(ValueError, match='burnin value cannot be larger than'):
        s.sample(10, burnin=10)
    with pytest.raises(ValueError, match='chains must a positive integer'):
        s.sample(10, chains=0)


def test_hyperparameter_input(small):
    from occuspytial_amd import LogitICARGibbs
    rng = np.random.default_rng(10)
    hypers = {'tau_rate': 1.0, 'tau_shape': 5.0, 'a_mu': rng.random(2), 'b_mu': rng.random(3),
              'a_prec': np.eye(2), 'b_prec': np.eye(3)}
    s1, s2 = LogitICARGibbs(*small), LogitICARGibbs(*small, hparams=hypers)
    assert s1.fixed['tau_shape'] != s2.fixed['tau_shape'] and s1.fixed.tau_rate != s2.fixed.tau_rate
    for k in ('a_mu', 'b_mu', 'a_prec', 'b_prec'):
        assert not np.allclose(s1.fixed[k], s2.fixed[k])
    assert s2.fixed.a_mu is hypers['a_mu']


def test_nonsingular_spatial_precision_matrix(small):
    from occuspytial_amd import LogitICARGibbs
    mat = rand(150, 150, density=0.9, format='csc', random_state=10)
    with pytest.raises(ValueError, match='Spatial precision matrix Q must be'):
        LogitICARGibbs(mat.T * mat, *small[1:])
    # lattices the reference wrongly rejects (lambda_2 below its eigsh shift) are accepted
    from occuspytial_amd._problem import _verify_spatial_precision
    from occuspytial_amd.utils import rand_precision_mat
    _verify_spatial_precision(rand_precision_mat(100, 100, max_neighbors=4).tocsr().astype(float))


def test_sampler_with_no_step_method(small):
    from occuspytial_amd.gibbs.base import GibbsBase

    class FakeSampler(GibbsBase):
        def __init__(self, Q, W, X, y):
            super().__init__(Q, W, X, y)
            super()._configure(Q, None)

    with pytest.raises(NotImplementedError, match='FakeSampler must implement a `step` method.'):
        FakeSampler(*small).sample(5, progressbar=False)


def test_too_many_covariates_is_an_error(small):
    from occuspytial_amd import LogitICARGibbs
    Q, W, X, y = small
    Xbig = np.hstack([X, np.ones((X.shape[0], 31))])
    with pytest.raises(ValueError, match='at most 32'):
        LogitICARGibbs(Q, W, Xbig, y)
    LogitICARGibbs(Q, W, np.hstack([X, np.ones((X.shape[0], 7))]), y)      # 9-32: the engine's generic kernels


def test_distribution_helpers_match_the_reference_functions():
    """``occuspytial_amd.distributions`` against outputs of the reference's Cython functions (tests/golden/
    native_helpers.npz, made by calling them with a cloned generator)."""
    from occuspytial_amd.distributions import ensure_sums_to_zero, precision_mvnorm
    from .conftest import load_golden
    g = load_golden('native_helpers')

    class Replay:  # hands out the variates the reference consumed
        def __init__(self, eps):
            self.eps = eps

        def standard_normal(self, n):
            assert n == self.eps.size
            return self.eps

    import numpy.random as npr
    real_default_rng = npr.default_rng
    for d in range(1, 9):
        prec = np.array(g[f'mvn{d}_prec'], dtype=float)
        npr.default_rng = lambda rs: rs          # a Generator is passed through unaltered, as numpy does
        try:
            draw = precision_mvnorm(g[f'mvn{d}_b'], prec, Replay(g[f'mvn{d}_eps']))
        finally:
            npr.default_rng = real_default_rng
        assert np.allclose(draw, g[f'mvn{d}_draw'], rtol=1e-11, atol=1e-13)
        assert np.allclose(np.tril(prec), np.tril(g[f'mvn{d}_prec_after']), rtol=1e-12, atol=1e-13)
    out = np.empty_like(g['proj_x'])
    ensure_sums_to_zero(g['proj_x'], g['proj_z'], out)
    assert np.allclose(out, g['proj_out'], rtol=0, atol=1e-14)
    with pytest.raises(RuntimeError, match='Cholesky factorization/solver failed!'):
        precision_mvnorm(np.zeros(2), np.array([[1.0, 2.0], [2.0, 1.0]]), 1)


def test_reduced_rank_sampler_configuration_needs_no_device():
    """LogitRSRGibbs's set-up (reference logit.py:413-455) is host work: basis size from the threshold or from
    ``q``, ``fixed`` entries, default ``tau_shape``, the reference's error for a threshold outside [0, 1]."""
    from occuspytial_amd import LogitRSRGibbs
    from .conftest import load_golden
    g = load_golden('ref_rsr150_r05')
    Q, W, X, y = _inputs(load_golden('ref_queen150_ragged'))[:4]
    s = LogitRSRGibbs(Q, W, X, y, random_state=10)
    assert s.fixed.q == int(g['rsr_dim']) and s.fixed.K.shape == (150, s.fixed.q)
    assert s.fixed.tau_shape == float(g['cfg_tau_shape'])
    assert np.allclose(s.fixed.K @ s.fixed.K.T, g['rsr_K'] @ g['rsr_K'].T, atol=1e-8)
    s._initialize_posterior_state(None)        # same generator calls as the reference: same start values
    assert np.allclose(s.state.tau, float(g['start_tau'])) and np.allclose(s.state.alpha, g['start_alpha'])
    assert s.state.eta.shape == (s.fixed.q,) and np.allclose(s.state.spatial, s.fixed.K @ s.state.eta)
    assert LogitRSRGibbs(Q, W, X, y, q=10).fixed.q == 10
    with pytest.raises(ValueError, match='Threshold value needs to be in'):
        LogitRSRGibbs(Q, W, X, y, r=1.1)
    hp = {'tau_rate': 1.0, 'tau_shape': 5.0}
    assert LogitRSRGibbs(Q, W, X, y, hparams=hp, q=10).fixed.tau_shape == 5.0


def test_prior_draw_selection_on_the_host():
    """How the N(0, Q) prior term is drawn (FlatProblem ``prior_draw``): the edge form for an ICAR precision; the
    reference's dense eigenfactor (gibbs/logit.py:64-67) on request, and by itself for a singular positive semi-definite
    Q with positive off-diagonals -- which the reference accepts (base.py:166-170) and the edge form cannot represent;
    a non-singular Q is refused with the reference's message either way."""
    from scipy import sparse
    from occuspytial_amd._problem import FlatProblem, dense_prior_factor
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(6, 7, visits=3, p=2, q=2, random_state=6)
    assert FlatProblem(Q, W, X, y).prior_factor is None
    E = FlatProblem(Q, W, X, y, prior_draw='dense').prior_factor
    assert E.shape == (42, 41) and np.abs(E @ E.T - Q.toarray()).max() < 1e-12
    row = np.zeros(42)
    row[[0, 20, 41]] = 1.0, 1.0, -2.0                              # sites 0 and 20 are not neighbours: +1 stays
    M = sparse.csr_matrix(row[None, :])
    Qg = (Q + M.T @ M).tocsr()                                     # still Q 1 = 0, PSD, but a positive off-diagonal
    assert (Qg - sparse.diags(Qg.diagonal())).max() > 0
    with pytest.raises(ValueError, match='non-positive off-diagonal'):
        FlatProblem(Qg, W, X, y, prior_draw='edge')
    Eg = FlatProblem(Qg, W, X, y).prior_factor
    assert Eg is not None and np.abs(Eg @ Eg.T - Qg.toarray()).max() < 1e-10
    for mode in ('auto', 'dense'):
        with pytest.raises(ValueError, match='Spatial precision matrix Q must be singular.'):
            FlatProblem(Q + 0.1 * sparse.identity(42), W, X, y, prior_draw=mode)
    with pytest.raises(ValueError, match='symmetric'):
        dense_prior_factor(sparse.csr_matrix(np.triu(Q.toarray())))
    # the plain-array round trip (what a multi-process launch broadcasts) keeps the factor
    assert np.array_equal(FlatProblem.from_arrays(FlatProblem(Qg, W, X, y).to_arrays()).prior_factor, Eg)


def test_diagnostics_against_closed_forms():
    """The numpy diagnostics that stand where the reference calls arviz (posterior.py:63-76), against what theory says
    for processes with known answers (arviz is not installed, so closed forms stand in for it): a stationary AR(1)
    with coefficient rho has ESS = N (1 - rho) / (1 + rho) and MCSE of the mean sd / sqrt(ESS); iid draws have ESS = N;
    the 94 % HDI of a standard normal is +-1.881; R-hat of well-mixed chains is 1; a shift of one sd between chains
    gives R-hat = sqrt(1 + 5/12) ~ 1.2 for the bulk statistic (variance of {0, 1, 0, 1} shifts)."""
    from occuspytial_amd import diagnostics as dg
    rng = np.random.default_rng(7)
    N, C = 40000, 4
    for rho in (0.0, 0.5, 0.8):
        x = np.zeros((C, N))
        x[:, 0] = rng.standard_normal(C) / np.sqrt(1 - rho * rho)
        e = rng.standard_normal((C, N))
        for t in range(1, N):
            x[:, t] = rho * x[:, t - 1] + e[:, t]
        want = C * N * (1 - rho) / (1 + rho)
        assert abs(dg.ess(x) / want - 1) < 0.12, (rho, dg.ess(x), want)
        sd = 1 / np.sqrt(1 - rho * rho)
        assert abs(dg.mcse_mean(x) / (sd / np.sqrt(want)) - 1) < 0.10
        assert 0.999 < dg.rhat(x) < 1.005
    iid = rng.standard_normal((C, N))
    lo, hi = dg.hdi(iid)
    assert abs(lo + 1.881) < 0.03 and abs(hi - 1.881) < 0.03
    shifted = iid + np.array([0.0, 1.0, 0.0, 1.0])[:, None]
    assert 1.10 < dg.rhat(shifted) < 1.25
