"""The host side of the product on a machine WITHOUT a GPU: the CPU restatement exports the engine's own C ABI
(``oracle/liboccoracle_abi.so``, SURVEY 8b: "the CPU restatement exports the identical ABI"), and these tests point the
product's ctypes binding at it -- explicitly, through a fixture; the product itself never looks for it -- so that the same
``Engine`` / ``EngineGroup`` wrappers and sampler classes that drive the HIP library on a GPU box run here: the
reference's sampler-level tests (``occuspytial/gibbs/tests/test_samplers.py``), the per-conditional entry points fed with
the reference's fixtures, checkpoints, groups, the reference-form prior draw, and one whole-chain comparison with the
reference's own chains through the drop-in class.  (On the GPU box ``tests/test_gpu_api.py`` / ``test_gpu_golden.py`` run
the same things on the device.)"""
import os

import numpy as np
import pytest

from .conftest import GOLDEN_CASES, ROOT, load_golden
from .test_api_cpu import _inputs

ABI_LIB = os.path.join(ROOT, 'oracle', 'liboccoracle_abi.so')


@pytest.fixture
def cpu_abi(monkeypatch, oracle):
    """Point occuspytial_amd._lib at the oracle's build of the C ABI for the duration of one test."""
    from occuspytial_amd import _lib
    if not os.path.exists(ABI_LIB) or os.path.getmtime(ABI_LIB) < os.path.getmtime(os.path.join(ROOT, 'oracle', 'occ_oracle_abi.c')):
        oracle.build()                            # make -C oracle (idempotent)
    monkeypatch.setattr(_lib, 'LIB_PATH', ABI_LIB)
    monkeypatch.setattr(_lib, '_lib', None)
    lib = _lib.load()
    assert lib.occ_device_count() == 0          # this is not the HIP library
    yield lib


@pytest.fixture
def data():
    return _inputs(load_golden('ref_queen150_ragged'))[:4]   # 150 sites, 100 surveyed, p=3, q=2


def test_the_restatement_exports_the_whole_abi(cpu_abi):
    from occuspytial_amd import _lib
    assert cpu_abi.occ_abi_version() == _lib.ABI_VERSION
    for name, _, _ in _lib.SYMBOLS:
        assert hasattr(cpu_abi, name), name


def test_product_default_is_the_hip_library_and_nothing_else():
    """No search path, no environment switch: the binding names one file, the HIP engine's."""
    from occuspytial_amd import _lib
    assert _lib.LIB_PATH == os.path.join(ROOT, 'occuspytial_amd', 'libocc_gibbs.so')
    src = open(os.path.join(ROOT, 'occuspytial_amd', '_lib.py')).read()
    assert 'oracle' not in src and 'environ' not in src


def test_sampler_shapes_reproducibility_burnin_chains(cpu_abi, data):
    """reference gibbs/tests/test_samplers.py:54-87 on the drop-in class."""
    from occuspytial_amd import LogitICARGibbs
    s = LogitICARGibbs(*data, random_state=10)
    samples = s.sample(5, chains=1, progressbar=False)
    assert samples['alpha'].shape == (1, 5, 2) and samples['beta'].shape == (1, 5, 3) and samples['tau'].shape == (1, 5)
    samples2 = LogitICARGibbs(*data, random_state=10).sample(5, chains=1, progressbar=False)
    for k in ('alpha', 'beta', 'tau'):
        assert np.array_equal(samples2[k], samples[k])
    assert isinstance(s.copy(), LogitICARGibbs)
    with pytest.raises(ValueError, match='burnin value cannot be larger than'):
        s.sample(10, burnin=11)
    samples = s.sample(10, burnin=3, chains=1, progressbar=False)
    assert samples['alpha'].shape == (1, 7, 2) and samples['tau'].shape == (1, 7)
    with pytest.raises(ValueError, match='chains must a positive integer'):
        s.sample(10, chains=0)
    samples = s.sample(5, chains=3, progressbar=False)
    assert samples['alpha'].shape == (3, 5, 2) and samples['beta'].shape == (3, 5, 3) and samples['tau'].shape == (3, 5)
    assert not np.allclose(samples['tau'][0], samples['tau'][1])
    assert len(s.chain) == 5 and s.state.eta.shape == (150,) and abs(s.state.eta.sum()) < 1e-9
    a = LogitICARGibbs(*data, random_state=3).sample(60, burnin=20, chains=2, progressbar=False)
    b = LogitICARGibbs(*data, random_state=3).sample(60, burnin=20, chains=2, progressbar=True)     # chunked runs
    for k in ('alpha', 'beta', 'tau'):
        assert np.array_equal(a[k], b[k])


def test_start_parameter_and_public_step(cpu_abi, data):
    from occuspytial_amd import LogitICARGibbs
    rng = np.random.default_rng(10)
    s = LogitICARGibbs(*data, random_state=10)
    samples = s.sample(5, progressbar=False)
    start = {'alpha': rng.random(2), 'beta': rng.random(3), 'tau': 2, 'eta': rng.random(150)}
    samples2 = s.sample(5, start=start, progressbar=False)
    for k in ('alpha', 'beta', 'tau'):
        assert not np.allclose(samples2[k][0, 0], samples[k][0, 0])
    s = LogitICARGibbs(*data, random_state=5)
    s._initialize_posterior_state(None)
    s.step()
    t = LogitICARGibbs(*data, random_state=5)
    t._initialize_posterior_state(None)
    t.step()
    for k in ('alpha', 'beta', 'tau', 'eta', 'z', 'omega_b'):
        assert np.array_equal(np.asarray(getattr(s.state, k)), np.asarray(getattr(t.state, k)))
    assert s.state.spatial is s.state.eta and np.array_equal(s.state.k, s.state.z - 0.5)
    assert s.state.W.shape[0] == s.state.omega_a.shape[0] == sum(s.W.visits(s.state.exists))
    assert s.state.exists[:len(s.fixed.obs)] == s.fixed.obs


def test_checkpoint_resume_and_devices_fan_out(cpu_abi, data, tmp_path):
    """Checkpoint / resume and the in-process fan-out (``devices=[...]`` -> ``occ_create_group``, one host thread per
    handle): same draws whatever the number of "devices"."""
    from occuspytial_amd import LogitICARGibbs
    whole = LogitICARGibbs(*data, random_state=21).sample(50, chains=3, progressbar=False)
    first = LogitICARGibbs(*data, random_state=21)
    first.sample(20, chains=3, progressbar=False)
    path = tmp_path / 'chains.npz'
    first.checkpoint(path)
    tail = LogitICARGibbs(*data, random_state=99).resume(str(path), 30, progressbar=False)
    for k in ('alpha', 'beta', 'tau'):
        assert np.array_equal(tail[k], whole[k][:, 20:])
    grouped = LogitICARGibbs(*data, random_state=21, devices=[0, 1]).sample(50, chains=3, progressbar=False)
    for k in ('alpha', 'beta', 'tau'):
        assert np.array_equal(grouped[k], whole[k])
    g2 = LogitICARGibbs(*data, random_state=21, devices=[0, 1, 2, 3])
    g2.sample(20, chains=3, progressbar=False)
    assert len(g2.__dict__['_engine'].engines) == 3
    tail2 = LogitICARGibbs(*data, random_state=1, devices=[0, 1]).resume(g2.checkpoint(), 30, progressbar=False)
    for k in ('alpha', 'beta', 'tau'):
        assert np.array_equal(tail2[k], whole[k][:, 20:])


def test_reduced_rank_sampler_api(cpu_abi, data):
    from occuspytial_amd import LogitRSRGibbs
    s = LogitRSRGibbs(*data, random_state=10, q=10)
    out = s.sample(12, burnin=2, chains=3, progressbar=False)
    assert s.fixed.q == 10 and out['alpha'].shape == (3, 10, 2) and s.fixed.tau_shape == 0.5 + 0.5 * 10
    assert s.state.eta.shape == (10,) and np.allclose(s.state.spatial, s.fixed.K @ s.state.eta, atol=1e-12)
    with pytest.raises(ValueError, match='Threshold value needs to be in'):
        LogitRSRGibbs(*data, r=1.1)
    # the reference's default threshold keeps ~13 % of a lattice's sites (logit.py:415-446): more than the 128 columns the
    # device's LDS-resident solve holds -- the host side takes them (the device's second path is tested with a GPU)
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(36, 36, visits=3, p=2, q=2, random_state=3)
    big = LogitRSRGibbs(Q, W, X, y, random_state=4)
    m = big.fixed.q
    assert 128 < m < 300 and big.fixed.K.shape == (1296, m)
    out = big.sample(4, chains=1, progressbar=False)
    assert out['tau'].shape == (1, 4) and np.all(out['tau'] > 0) and np.all(np.isfinite(out['beta']))
    assert np.allclose(big.state.spatial, big.fixed.K @ big.state.eta, atol=1e-11)


def test_errors_map_to_the_references_exceptions(cpu_abi, data):
    from scipy import sparse
    from occuspytial_amd import LogitICARGibbs
    from occuspytial_amd._engine import Engine
    Q, W, X, y = data
    with pytest.raises(ValueError, match='Spatial precision matrix Q must be singular.'):
        LogitICARGibbs(Q + 0.1 * sparse.identity(150), W, X, y)
    s = LogitICARGibbs(Q, W, X, y, random_state=1)
    eng = Engine(s._problem, [1])
    with pytest.raises(ValueError):
        eng.get('no_such_state')
    with pytest.raises(ValueError):
        eng.set('eta', np.zeros(3))
    eng.set_start(0, np.zeros(2), np.zeros(3), 1.0, np.zeros(150))
    with pytest.raises(RuntimeError, match='Cholesky factorization/solver failed!'):
        eng.cond_beta(-np.ones(150), np.zeros(3))          # a negative "omega": X' Omega X + prior is not positive definite
    eng.close()


def test_engine_through_the_abi_is_the_oracle(cpu_abi, oracle):
    """Nothing between the ABI and the restatement but plumbing: a run through Engine equals OracleSampler's, bitwise."""
    from occuspytial_amd._engine import Engine
    from .test_gpu_parity import KEY, _problem_from_golden
    prob, start = _problem_from_golden('ref_graph300_weighted')
    eng = Engine(prob, [KEY, KEY + 1])
    assert eng.transport.startswith('cpu restatement') and eng.stats()['persistent_solve'] == 0
    for c in range(2):
        eng.set_start(c, **start)
    A, B, T = eng.run(15, 3)
    for c in range(2):
        orc = oracle.OracleSampler(prob, KEY + c)
        orc.set_start(**start)
        a, b, t = orc.run(15, 3)
        assert np.array_equal(a, A[c]) and np.array_equal(b, B[c]) and np.array_equal(t, T[c])
        assert np.array_equal(orc.get('eta'), eng.get('eta', c)) and np.array_equal(orc.get('z'), eng.get('z', c))
    eng.close()


@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_per_conditional_entry_points_against_the_references_fixtures(cpu_abi, name):
    """The same tuples tests/test_gpu_golden.py sends to the device (inputs + the variates the reference consumed ->
    the reference's outputs), through the same entry points of the same binding."""
    from occuspytial_amd._engine import Engine
    from . import test_gpu_golden as G
    from .test_gpu_parity import _problem_from_golden
    g = load_golden(name)
    prob, start = _problem_from_golden(name)
    eng = Engine(prob, [G.KEY])
    eng.solve_bounds = (1e-10, 1e-9)    # the oracle's sequential sums against scipy's (tests/test_oracle_golden.py: 1e-10)
    case = (g, prob, eng, start)
    G.test_tau_conditional_on_device(case)
    G.test_eta_conditional_on_device(case)
    G.test_beta_conditional_on_device(case)
    G.test_alpha_conditional_on_device(case)
    G.test_z_conditional_on_device(case)
    G.test_conditionals_chain_like_the_references_step(case)
    eng.close()


def test_variate_draws_through_the_abi(cpu_abi, oracle):
    from occuspytial_amd._engine import device_draw
    z = np.linspace(-5, 5, 257)
    assert np.array_equal(device_draw('pg1', z, key=7, it=2, stream=1), oracle.pg1(z, key=7, it=2, stream=1))
    g = device_draw('std_gamma', np.full(8, 75.5), key=9, it=4, stream=2)
    assert g[0] == oracle.lib().orc_std_gamma_draw(9, 4, 2, 75.5) and len(set(g)) == 8
    assert device_draw('uniform', n=16, key=3, it=0, stream=8).min() > 0


def test_drop_in_class_against_the_references_chains(cpu_abi):
    """The whole drop-in path -- ``LogitICARGibbs(...).sample(6000, burnin=1000, chains=4)`` as a user calls it, start
    values and keys from the chains' own generators -- against the chains the REFERENCE produced for the same call
    (tests/golden/refchain_queen150_tauprior.npz), once with each form of the prior draw."""
    from occuspytial_amd import LogitICARGibbs
    from .test_reference_chains import compare_with_reference, problem_of
    case = 'refchain_queen150_tauprior'
    Q, W, X, y, hp, ch = problem_of(case)
    for mode in ('auto', 'dense'):
        s = LogitICARGibbs(Q, W, X, y, hparams=hp, random_state=314, prior_draw=mode)
        assert (s._problem.prior_factor is not None) == (mode == 'dense')
        post = s.sample(int(ch['size']), burnin=int(ch['burnin']), chains=4, progressbar=False)
        compare_with_reference(case, post['alpha'], post['beta'], post['tau'])
