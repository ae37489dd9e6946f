#!/usr/bin/env python
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE's own code.

Run in the build container only (needs ``/root/reference``; the GPU box never has it)::

    python tests/golden/make_golden.py                 # writes the per-conditional fixtures tests/golden/ref_*.npz
    python tests/golden/make_golden.py --chains-only   # writes the whole-chain fixtures tests/golden/refchain_*.npz
                                                       # (--with-chains: both)

What runs, and what does not:

* The reference's two Cython modules (``distributions.pyx``, ``data.pyx``) are cythonized from a
  scratch copy under ``/tmp`` (the committed ``.c`` files target numpy 1.x and do not compile here).
* ``gibbs/base.py``, ``gibbs/logit.py``, ``chain.py``, ``gibbs/state.py`` run unmodified.
* ``polyagamma`` (third-party C sampler, pinned 1.2.0) and ``arviz`` are NOT installed.  So that the
  module-level imports succeed, this script puts two import shims on ``sys.path`` in the scratch
  directory.  The ``polyagamma`` shim records the argument the reference passes in and returns a
  truncated sum-of-exponentials PG(1, z) variate (200 terms of the defining series) drawn from the
  generator it is handed.  PG values therefore are INPUTS to the fixtures of the other conditionals
  ("reference driver + stand-in PG"); no fixture here pins PG draws themselves.
* ``utils.make_data`` needs ``libpysal``; inputs come from ``occuspytial_amd.utils`` instead.

Every random variate the reference consumed is recovered by replaying a clone of the SFC64 state
taken just before the call, so each fixture is (inputs, variates, outputs) of one conditional.
"""
import os
import shutil
import subprocess
import sys
import textwrap

import numpy as np
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
SCRATCH = '/tmp/occ_golden_build'

sys.path.insert(0, REPO)


def build_reference():
    if not os.path.isdir(REF):
        raise SystemExit('reference tree not present; fixtures can only be regenerated in the build container')
    pkg = os.path.join(SCRATCH, 'occuspytial')
    if not os.path.exists(os.path.join(SCRATCH, '.built')):
        shutil.rmtree(SCRATCH, ignore_errors=True)
        os.makedirs(SCRATCH)
        shutil.copytree(os.path.join(REF, 'occuspytial'), pkg)
        for f in ('distributions.c', 'data.c'):
            os.remove(os.path.join(pkg, f))
        with open(os.path.join(SCRATCH, 'setup_ref.py'), 'w') as fh:
            fh.write(textwrap.dedent('''
                import os, numpy as np
                from setuptools import setup, Extension
                from Cython.Build import cythonize
                inc = np.get_include()
                lib = os.path.abspath(os.path.join(inc, '..', '..', 'random', 'lib'))
                exts = [Extension('occuspytial.distributions', ['occuspytial/distributions.pyx'],
                                  include_dirs=[inc], library_dirs=[lib], libraries=['npyrandom', 'm'],
                                  define_macros=[('NPY_NO_DEPRECATED_API', 0)]),
                        Extension('occuspytial.data', ['occuspytial/data.pyx'], include_dirs=[inc],
                                  define_macros=[('NPY_NO_DEPRECATED_API', 0)])]
                setup(name='ref', ext_modules=cythonize(exts, language_level=3),
                      script_args=['build_ext', '--inplace'])
            '''))
        subprocess.run([sys.executable, 'setup_ref.py'], cwd=SCRATCH, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        shim = os.path.join(SCRATCH, 'shims')
        os.makedirs(os.path.join(shim, 'polyagamma'))
        os.makedirs(os.path.join(shim, 'arviz'))
        with open(os.path.join(shim, 'polyagamma', '__init__.py'), 'w') as fh:
            fh.write(textwrap.dedent('''
                """Import shim (NOT the polyagamma package): truncated-series PG(1, z) stand-in."""
                import os
                import numpy as np
                calls = []
                def random_polyagamma(h, z, *, disable_checks=False, random_state=None):
                    z = np.asarray(z, dtype=float)
                    k = np.arange(1, 201) - 0.5
                    g = random_state.standard_exponential((z.size, 200))
                    c = z.reshape(-1, 1) / (2 * np.pi)
                    ser = (g / (k ** 2 + c ** 2)).sum(axis=1)
                    if os.environ.get('OCC_GOLDEN_PG_TAIL'):
                        # whole-chain runs: add the MEAN of the dropped terms k > 200, sum 1/((k-1/2)^2 + c^2)
                        # = atan(c/200)/c (midpoint rule, exact to 1e-9), so that E[PG] is exact; what is
                        # missing is the tail's variance, 3e-9 of the total
                        cc = np.abs(c.ravel())
                        ser = ser + np.where(cc > 1e-8, np.arctan(cc / 200.0) / np.maximum(cc, 1e-300), 1.0 / 200.0)
                    out = ser / (2 * np.pi ** 2)
                    if not os.environ.get('OCC_GOLDEN_PG_TAIL'):
                        calls.append((z.copy(), out.copy()))
                    return out
            '''))
        with open(os.path.join(shim, 'arviz', '__init__.py'), 'w') as fh:
            fh.write(textwrap.dedent('''
                """Import shim (NOT arviz): only what occuspytial/posterior.py touches at import."""
                class _Style:
                    def use(self, *a, **k):
                        pass
                style = _Style()
                class _Wrap:
                    def __init__(self, d):
                        self.posterior = d
                def convert_to_inference_data(d):
                    return _Wrap(d)
            '''))
        open(os.path.join(SCRATCH, '.built'), 'w').close()
    sys.path.insert(0, os.path.join(SCRATCH, 'shims'))
    sys.path.insert(0, SCRATCH)


def clone_rng(rng):
    """A generator positioned exactly where ``rng`` is now."""
    c = np.random.Generator(np.random.SFC64())
    c.bit_generator.state = rng.bit_generator.state
    return c


def flatten(W, y, q):
    sites = np.array(list(W.keys()), dtype=np.int64)
    visits = np.array([W[s].shape[0] for s in sites], dtype=np.int64)
    Wf = np.concatenate([W[s] for s in sites]).reshape(-1, q)
    yf = np.concatenate([np.asarray(y[s]) for s in sites]).astype(np.int64)
    return sites, visits, Wf, yf


def capture_case(name, Q, W, X, y, seed, hparams=None, iters=3):
    import polyagamma as pg_shim
    from occuspytial.gibbs.logit import LogitICARGibbs
    from scipy.sparse.linalg import minres

    out = {}
    Qc = sparse.csr_matrix(Q).astype(float)
    Qc.sort_indices()
    n, p = X.shape
    q = next(iter(W.values())).shape[1]
    sites, visits, Wf, yf = flatten(W, y, q)
    out.update(Q_indptr=Qc.indptr.astype(np.int64), Q_indices=Qc.indices.astype(np.int64), Q_data=Qc.data,
               X=X, sites=sites, visits=visits, W_flat=Wf, y_flat=yf, seed=np.int64(seed))
    if hparams:
        for k, v in hparams.items():
            out['hp_' + k] = np.asarray(v, dtype=float)

    s = LogitICARGibbs(Q, W, X, y, hparams=hparams, random_state=seed)
    f = s.fixed
    out.update(cfg_z0=s.state.z.copy(), cfg_not_obs=np.array(f.not_obs, dtype=np.int64),
               cfg_obs=np.array(f.obs, dtype=np.int64),
               cfg_not_surveyed=np.array(f.not_surveyed, dtype=np.int64),
               cfg_W_not_obs=np.asarray(f.W_not_obs), cfg_stacked_w_indices=np.asarray(f.stacked_w_indices),
               cfg_tau_rate=np.float64(f.tau_rate), cfg_tau_shape=np.float64(f.tau_shape),
               cfg_a_mu=f.a_mu, cfg_a_prec=f.a_prec, cfg_b_mu=f.b_mu, cfg_b_prec=f.b_prec)

    # default start (reference base.py:199-212) -- same seed => same start values
    s._initialize_posterior_state(None)
    st = s.state
    out.update(start_tau=np.float64(st.tau), start_eta=st.eta.copy(), start_alpha=st.alpha.copy(),
               start_beta=st.beta.copy())
    # chain seeding facts (reference base.py:293-306): first raw words of the copies' streams
    s2 = LogitICARGibbs(Q, W, X, y, hparams=hparams, random_state=seed)
    c1, c2 = s2.copy(), s2.copy()
    out['copy_raw'] = np.stack([c.rng.bit_generator.random_raw(4) for c in (c1, c2)])
    out['parent_raw'] = clone_rng(s2.rng).bit_generator.random_raw(4)

    post = s.dists.eta_posterior
    for it in range(iters):
        t = f'it{it}_'
        # 1. omega_b (stand-in PG; records the argument the reference formed)
        pg_shim.calls.clear()
        out[t + 'ob_beta'], out[t + 'ob_eta'] = st.beta.copy(), st.spatial.copy()
        s._update_omega_b()
        out[t + 'ob_arg'], out[t + 'omega_b'] = pg_shim.calls[0]
        # 2. tau
        c = clone_rng(s.rng)
        out[t + 'tau_eta'] = st.eta.copy()
        s._update_tau()
        g = c.standard_gamma(f.tau_shape)
        rate = 0.5 * (out[t + 'tau_eta'] @ f.Q @ out[t + 'tau_eta']) + f.tau_rate
        assert (1 / rate) * g == st.tau, 'gamma replay does not reproduce the reference draw'
        out[t + 'tau_g'], out[t + 'tau'] = np.float64(g), np.float64(st.tau)
        # 3. eta
        c = clone_rng(s.rng)
        x0 = None if post._guess is None else post._guess.copy()
        out[t + 'eta_k'], out[t + 'eta_beta'] = st.k.copy(), st.beta.copy()
        s._update_eta()
        eps = c.standard_normal(2 * n - 1)
        out[t + 'eta_eps'] = eps
        out[t + 'eta_b'] = out[t + 'eta_k'] - st.omega_b * (X @ st.beta)
        out[t + 'eta_rhs'] = post._rhs[:n].copy()
        out[t + 'eta_x0'] = np.zeros(2 * n) if x0 is None else x0
        out[t + 'eta_x0_none'] = np.bool_(x0 is None)
        out[t + 'eta_xz'] = post._guess.copy()
        out[t + 'eta'] = st.eta.copy()
        # iteration count of the same solve (scipy is the reference's own dependency)
        P = sparse.block_diag((f.Q, f.Q), format='csc') * st.tau
        P.setdiag(P.diagonal() + np.tile(st.omega_b, 2))
        cnt = [0]
        xz2, info = minres(P, np.concatenate([out[t + 'eta_rhs'], np.ones(n)]), x0=x0,
                           callback=lambda xk: cnt.__setitem__(0, cnt[0] + 1))
        assert info == 0 and np.array_equal(xz2, out[t + 'eta_xz'])
        out[t + 'eta_itn'] = np.int64(cnt[0])
        # covariance identity of the prior term: E E^T = Q  (checked, not stored: E is n x (n-1))
        if it == 0:
            assert np.allclose(post._eigen @ post._eigen.T, f.Q.toarray(), atol=1e-8)
        # 4. beta
        c = clone_rng(s.rng)
        out[t + 'beta_k'] = st.k.copy()
        s._update_beta()
        out[t + 'beta_eps'] = c.standard_normal(p)
        out[t + 'beta_A'] = (X.T * st.omega_b) @ X + f.b_prec
        out[t + 'beta_r'] = X.T @ (out[t + 'beta_k'] - st.omega_b * st.spatial) + f.b_prec_by_mu
        out[t + 'beta'] = st.beta.copy()
        # 5. omega_a
        pg_shim.calls.clear()
        out[t + 'oa_z'], out[t + 'oa_alpha'] = st.z.copy(), st.alpha.copy()
        s._update_omega_a()
        out[t + 'exists'] = np.array(st.exists, dtype=np.int64)
        out[t + 'oa_arg'], out[t + 'omega_a'] = pg_shim.calls[0]
        # 6. alpha
        c = clone_rng(s.rng)
        s._update_alpha()
        out[t + 'alpha_eps'] = c.standard_normal(q)
        WT = st.W.T
        out[t + 'alpha_A'] = (WT * st.omega_a) @ WT.T + f.a_prec
        out[t + 'alpha_r'] = WT @ (s.y[st.exists] - 0.5) + f.a_prec_by_mu
        out[t + 'alpha'] = st.alpha.copy()
        # 7. z
        c = clone_rng(s.rng)
        s._update_z()
        out[t + 'z_u_no'] = c.uniform(size=f.n_no)
        out[t + 'z_u_ns'] = c.uniform(size=f.n_ns) if f.n_ns else np.zeros(0)
        out[t + 'z'] = st.z.copy()
        out[t + 'k'] = st.k.copy()

    # API facts of reference gibbs/tests/test_samplers.py:54-87 (shapes; same seed => same samples)
    r = LogitICARGibbs(Q, W, X, y, hparams=hparams, random_state=seed).sample(5, chains=1, progressbar=False)
    out['api_alpha_shape'] = np.array(r['alpha'].shape)
    out['api_beta_shape'] = np.array(r['beta'].shape)
    out['api_tau_shape'] = np.array(r['tau'].shape)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'n =', n, 'minres its', [int(out[f'it{i}_eta_itn']) for i in range(iters)])


def capture_rsr_case(name, Q, W, X, y, seed, iters=3, **rsr_kw):
    """LogitRSRGibbs (reference gibbs/logit.py:269-485): the Moran basis K, the reduced precision K'QK, and per
    iteration the inputs, the standard normals and the output of the theta conditional -- the only conditional
    that differs from the ICAR sampler -- plus tau, beta, z, which read theta / spatial = K theta."""
    import polyagamma as pg_shim
    from occuspytial.gibbs.logit import LogitRSRGibbs

    out = {}
    Qc = sparse.csr_matrix(Q).astype(float)
    Qc.sort_indices()
    n, p = X.shape
    q = next(iter(W.values())).shape[1]
    sites, visits, Wf, yf = flatten(W, y, q)
    out.update(Q_indptr=Qc.indptr.astype(np.int64), Q_indices=Qc.indices.astype(np.int64), Q_data=Qc.data,
               X=X, sites=sites, visits=visits, W_flat=Wf, y_flat=yf, seed=np.int64(seed))
    s = LogitRSRGibbs(Q, W, X, y, random_state=seed, **rsr_kw)
    f = s.fixed
    post = s.dists.eta_posterior
    r = int(f.q)
    out.update(rsr_dim=np.int64(r), rsr_K=np.asarray(f.K), rsr_Q=np.asarray(f.Q), rsr_eigen=np.asarray(post._eigen),
               cfg_tau_rate=np.float64(f.tau_rate), cfg_tau_shape=np.float64(f.tau_shape),
               cfg_a_mu=f.a_mu, cfg_a_prec=f.a_prec, cfg_b_mu=f.b_mu, cfg_b_prec=f.b_prec)
    assert np.allclose(post._eigen @ post._eigen.T, f.Q, atol=1e-8)
    s._initialize_posterior_state(None)
    st = s.state
    out.update(start_tau=np.float64(st.tau), start_eta=st.eta.copy(), start_alpha=st.alpha.copy(),
               start_beta=st.beta.copy(), start_spatial=st.spatial.copy())
    for it in range(iters):
        t = f'it{it}_'
        pg_shim.calls.clear()
        out[t + 'ob_beta'], out[t + 'ob_spatial'] = st.beta.copy(), st.spatial.copy()
        s._update_omega_b()
        out[t + 'ob_arg'], out[t + 'omega_b'] = pg_shim.calls[0]
        c = clone_rng(s.rng)
        out[t + 'tau_theta'] = st.eta.copy()
        s._update_tau()
        out[t + 'tau_g'], out[t + 'tau'] = np.float64(c.standard_gamma(f.tau_shape)), np.float64(st.tau)
        c = clone_rng(s.rng)
        out[t + 'eta_k'], out[t + 'eta_beta'] = st.k.copy(), st.beta.copy()
        s._update_eta()
        out[t + 'eta_eps'] = c.standard_normal(r + n)      # [:r] -> eigenfactor of K'QK, [r:] -> K' sqrt(omega)
        out[t + 'theta'], out[t + 'spatial'] = st.eta.copy(), st.spatial.copy()
        c = clone_rng(s.rng)
        out[t + 'beta_k'] = st.k.copy()
        s._update_beta()
        out[t + 'beta_eps'] = c.standard_normal(p)
        out[t + 'beta'] = st.beta.copy()
        pg_shim.calls.clear()
        s._update_omega_a()
        out[t + 'omega_a'] = pg_shim.calls[0][1]
        s._update_alpha()
        out[t + 'alpha'] = st.alpha.copy()
        c = clone_rng(s.rng)
        s._update_z()
        out[t + 'z_u_no'] = c.uniform(size=f.n_no)
        out[t + 'z'] = st.z.copy()
    res = LogitRSRGibbs(Q, W, X, y, random_state=seed, **rsr_kw).sample(5, chains=1, progressbar=False)
    out['api_alpha_shape'] = np.array(res['alpha'].shape)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'n =', n, 'basis columns =', r)


def capture_reference_chains(name, Q, W, X, y, seed, hparams=None, chains=4, size=6000, burnin=1000):
    """Whole chains of the REFERENCE sampler: ``LogitICARGibbs(...).sample(size, burnin, chains=4)`` exactly as a user
    calls it (reference gibbs/base.py:243-291 -> gibbs/parallel.py:4-42: one joblib process per chain, chain k on the
    k-th generator; gibbs/logit.py:254-266 per iteration, dense eigenfactor prior draw logit.py:66-67,77, scipy MINRES).
    The only stand-in is PG(1, z): the truncated defining series of the import shim plus the mean of the dropped tail
    (``OCC_GOLDEN_PG_TAIL``).  Stored: the kept alpha / beta / tau draws -- the fixture of the DISTRIBUTIONAL parity
    tests (tests/test_reference_chains.py, tests/test_gpu_api.py), the only tests that can catch an error shared by the
    oracle and the device in the edge-form prior term or the Polya-Gamma sampler."""
    from occuspytial.gibbs.logit import LogitICARGibbs
    os.environ['OCC_GOLDEN_PG_TAIL'] = '1'
    try:
        s = LogitICARGibbs(Q, W, X, y, hparams=hparams, random_state=seed)
        res = s.sample(size, burnin=burnin, chains=chains, progressbar=False)
        # float32: the draws feed distributional tests only (half the fixture size)
        out = {'alpha': np.asarray(res['alpha'], dtype=np.float32), 'beta': np.asarray(res['beta'], dtype=np.float32),
               'tau': np.asarray(res['tau'], dtype=np.float32),
               'seed': np.int64(seed), 'size': np.int64(size), 'burnin': np.int64(burnin)}
        for k, v in (hparams or {}).items():
            out['hp_' + k] = np.asarray(v, dtype=float)
    finally:
        del os.environ['OCC_GOLDEN_PG_TAIL']
    assert out['alpha'].shape[:2] == (chains, size - burnin) and out['tau'].shape == (chains, size - burnin)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'reference chains', out['alpha'].shape, 'tau mean', out['tau'].mean(axis=1))


def capture_native_helpers():
    """precision_mvnorm / ensure_sums_to_zero (reference distributions.pyx:24-110) known answers."""
    from occuspytial.distributions import ensure_sums_to_zero, precision_mvnorm
    rng = np.random.default_rng(1234)
    out = {}
    for d in range(1, 9):
        M = rng.standard_normal((d + 3, d))
        prec = M.T @ M + 0.1 * np.eye(d)
        b = rng.standard_normal(d)
        g = np.random.default_rng(100 + d)
        eps = np.random.default_rng(100 + d).standard_normal(d)
        work = prec.copy()
        draw = precision_mvnorm(b, work, random_state=g)
        out[f'mvn{d}_prec'], out[f'mvn{d}_b'], out[f'mvn{d}_eps'] = prec, b, eps
        out[f'mvn{d}_draw'], out[f'mvn{d}_prec_after'] = np.asarray(draw), work
    x, z = rng.standard_normal(257), rng.uniform(0.5, 2.0, 257)
    o = np.empty(257)
    ensure_sums_to_zero(x, z, o)
    out.update(proj_x=x, proj_z=z, proj_out=o)
    # failure path: non positive-definite precision => RuntimeError text
    try:
        precision_mvnorm(np.zeros(2), np.array([[1.0, 2.0], [2.0, 1.0]]), random_state=1)
        msg = ''
    except RuntimeError as e:
        msg = str(e)
    out['mvn_fail_msg'] = np.array(msg)
    np.savez_compressed(os.path.join(HERE, 'native_helpers.npz'), **out)
    print('native_helpers', msg)


def weighted_graph_case():
    from occuspytial_amd.utils import get_generator, make_graph_problem
    Qg, Wg, Xg, yg, *_ = make_graph_problem(n=300, k=6, visits=4, p=2, q=3, random_state=5)
    A = -sparse.triu(Qg, k=1).tocoo()
    wts = get_generator(7).uniform(0.5, 2.0, A.nnz)
    Aw = sparse.coo_matrix((wts, (A.row, A.col)), shape=A.shape)
    Aw = Aw + Aw.T
    Qw = (sparse.diags(np.asarray(Aw.sum(axis=1)).ravel()) - Aw).tocsr()
    return Qw, Wg, Xg, yg


def main():
    build_reference()
    from occuspytial_amd.utils import get_generator, make_graph_problem, rand_precision_mat, _expit

    # case A: the reference test-suite's shape (test_samplers.py:14-16): 150 sites, 100 surveyed,
    # 2..10 visits, p=3, q=2 -- generated here because make_data needs libpysal.
    rng = get_generator(10)
    n, p, q = 150, 3, 2
    Q = rand_precision_mat(10, 15).astype(float)
    surveyed = rng.choice(range(n), size=100, replace=False)
    visits = rng.integers(2, 10, size=100, endpoint=True)
    alpha, beta = rng.standard_normal(q), rng.standard_normal(p)
    X = rng.uniform(-2, 2, n * p).reshape(n, -1)
    X[:, 0] = 1
    z = rng.binomial(1, _expit(X @ beta))
    W, y = {}, {}
    for i, j in zip(surveyed, visits):
        Wi = rng.uniform(-2, 2, size=j * q).reshape(j, -1)
        Wi[:, 0] = 1
        W[int(i)] = Wi
        y[int(i)] = rng.binomial(1, z[i] * _expit(Wi @ alpha))
    hyp = {'tau_rate': 1.0, 'tau_shape': 5.0, 'a_mu': np.array([0.3, -0.2]), 'b_mu': np.array([0.1, 0.2, -0.4]),
           'a_prec': np.eye(2), 'b_prec': np.array([[1.0, 0.2, 0.0], [0.2, 2.0, 0.1], [0.0, 0.1, 0.5]])}
    only_chains = '--chains-only' in sys.argv
    if only_chains or '--with-chains' in sys.argv:
        # whole reference chains (default hyper-parameters twice; informative tau prior once, where tau mixes)
        capture_reference_chains('refchain_queen150_ragged', Q, W, X, y, seed=10)
        # informative Gamma(25, 25) prior on tau: the conditional's shape is (n - 1)/2 + 1/2 + 25 (gibbs/base.py:177-186
        # puts the likelihood's (n - 1)/2 into `tau_shape` itself), so tau mixes and is part of the comparison
        hyp_tau = dict(hyp, tau_rate=25.0, tau_shape=0.5 + 0.5 * (n - 1) + 25.0)
        capture_reference_chains('refchain_queen150_tauprior', Q, W, X, y, seed=3, hparams=hyp_tau)
        from occuspytial_amd.utils import make_lattice_problem as _mlp
        Qb, Wb, Xb, yb, *_ = _mlp(20, 20, visits=3, p=2, q=2, max_neighbors=8, random_state=0)
        capture_reference_chains('refchain_queen400_v3', Qb, Wb, Xb, yb, seed=10)
        hyp400 = {'tau_rate': 25.0, 'tau_shape': 0.5 + 0.5 * 399 + 25.0, 'a_mu': np.zeros(2), 'a_prec': np.eye(2) / 10,
                  'b_mu': np.zeros(2), 'b_prec': np.eye(2) / 10}
        capture_reference_chains('refchain_queen400_tauprior', Qb, Wb, Xb, yb, seed=11, hparams=hyp400)
        # irregular graph with WEIGHTED edges (the edge form draws sqrt(w_ij) eps_ij per edge), informative tau prior
        Qw, Wg, Xg, yg = weighted_graph_case()
        hyp300 = {'tau_rate': 25.0, 'tau_shape': 0.5 + 0.5 * 299 + 25.0, 'a_mu': np.zeros(3), 'a_prec': np.eye(3) / 10,
                  'b_mu': np.zeros(2), 'b_prec': np.eye(2) / 10}
        capture_reference_chains('refchain_graph300_weighted_tauprior', Qw, Wg, Xg, yg, seed=21, hparams=hyp300)
        if only_chains:
            return
    capture_case('ref_queen150_ragged', Q, W, X, y, seed=10)
    capture_case('ref_queen150_hparams', Q, W, X, y, seed=3, hparams=hyp, iters=2)

    # case B: BASELINE config 1 -- 20x20 lattice, 3 visits, p=q=2, all surveyed (rook and queen)
    from occuspytial_amd.utils import make_lattice_problem
    for nb, tag in ((4, 'rook'), (8, 'queen')):
        Qb, Wb, Xb, yb, *_ = make_lattice_problem(20, 20, visits=3, p=2, q=2, max_neighbors=nb, random_state=0)
        capture_case(f'ref_{tag}400_v3', Qb, Wb, Xb, yb, seed=10)

    # case C: irregular adjacency, weighted edges (still a graph Laplacian), 300 units
    Qw, Wg, Xg, yg = weighted_graph_case()
    capture_case('ref_graph300_weighted', Qw, Wg, Xg, yg, seed=21)

    capture_native_helpers()

    # case D: the reduced-rank sampler (LogitRSRGibbs) on the 150-site problem, default threshold and q = 10
    capture_rsr_case('ref_rsr150_r05', Q, W, X, y, seed=10)
    capture_rsr_case('ref_rsr150_q10', Q, W, X, y, seed=10, q=10, iters=2)


if __name__ == '__main__':
    main()
