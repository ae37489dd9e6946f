"""Pin the CPU oracle to fixtures produced by the reference's own code (tests/golden/make_golden.py).

Each test feeds one conditional's recorded inputs and the variates the reference consumed into the
oracle's restatement of that conditional and compares with what the reference produced.
"""
import numpy as np
import pytest
from scipy import sparse

from .conftest import GOLDEN_CASES, load_golden


def _Q(g):
    n = g['X'].shape[0]
    return sparse.csr_matrix((g['Q_data'], g['Q_indices'], g['Q_indptr']), shape=(n, n))


def _iters(g):
    return sorted({int(k[2:k.index('_')]) for k in g if k.startswith('it')})


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_tau_conditional(oracle, case):
    g = load_golden(case)
    Q = _Q(g)
    for it in _iters(g):
        rate = oracle.tau_rate(Q, g[f'it{it}_tau_eta'], float(g['cfg_tau_rate']))
        tau = (1.0 / rate) * float(g[f'it{it}_tau_g'])  # Generator.gamma(shape, scale) = scale * std_gamma
        assert tau == pytest.approx(float(g[f'it{it}_tau']), rel=1e-12)


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_eta_conditional_minres_and_projection(oracle, case):
    g = load_golden(case)
    Q = _Q(g)
    n = Q.shape[0]
    for it in _iters(g):
        t = f'it{it}_'
        x0 = None if bool(g[t + 'eta_x0_none']) else g[t + 'eta_x0']
        xz, info, itn, istop = oracle.minres_joint(Q, g[t + 'omega_b'], float(g[t + 'tau']), g[t + 'eta_rhs'], x0)
        assert info == 0 and istop in (1, 2)
        assert itn == int(g[t + 'eta_itn'])  # same stopping iteration as scipy's minres
        scale = np.abs(g[t + 'eta_xz']).max()
        assert np.abs(xz - g[t + 'eta_xz']).max() <= 1e-10 * scale
        eta = oracle.ensure_sums_to_zero(xz[:n], xz[n:])
        assert np.abs(eta - g[t + 'eta']).max() <= 1e-10 * np.abs(g[t + 'eta']).max()
        assert abs(eta.sum()) < 1e-9


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_eta_rhs_pieces(oracle, case):
    """b = k - omega * (X beta) and the sqrt(omega) eps term (logit.py:213, 76); the prior term is
    the only part the build replaces (edge form), checked in test_edge_prior_term_law."""
    g = load_golden(case)
    n = g['X'].shape[0]
    for it in _iters(g):
        t = f'it{it}_'
        b = g[t + 'eta_k'] - g[t + 'omega_b'] * (g['X'] @ g[t + 'eta_beta'])
        assert np.allclose(b, g[t + 'eta_b'], rtol=0, atol=1e-13)
        prior = g[t + 'eta_rhs'] - b - np.sqrt(g[t + 'omega_b']) * g[t + 'eta_eps'][:n]
        # what is left is E @ (sqrt(tau) eps2): orthogonal to the null vector of Q
        assert abs(prior.sum()) < 1e-8 * np.abs(prior).sum()


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_edge_prior_term_law(oracle, case):
    """u = B' eps has covariance B'B = Q, the covariance of the reference's E eps (E E' = Q)."""
    g = load_golden(case)
    Q = _Q(g)
    n = Q.shape[0]
    coo = sparse.triu(Q, k=1).tocoo()
    B = sparse.csr_matrix((np.concatenate([np.sqrt(-coo.data), -np.sqrt(-coo.data)]),
                           (np.tile(np.arange(coo.nnz), 2), np.concatenate([coo.row, coo.col]))),
                          shape=(coo.nnz, n))
    assert abs(B.T @ B - Q).max() < 1e-12
    # the oracle's u equals B' eps for the per-edge normals it draws
    key, it = 12345, 7
    eps = np.array([oracle.lib().orc_block_normal(key, int(lo), int(hi), it, 4) for lo, hi in zip(coo.row, coo.col)])
    u = oracle.edge_prior_term(Q, key, it)
    assert np.allclose(u, B.T @ eps, rtol=0, atol=1e-12)
    assert abs(u.sum()) < 1e-10 * np.abs(u).sum()


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_beta_conditional(oracle, case):
    g = load_golden(case)
    bpm = g['cfg_b_prec'] @ g['cfg_b_mu']
    for it in _iters(g):
        t = f'it{it}_'
        A, r = oracle.beta_system(g['X'], g[t + 'omega_b'], g[t + 'beta_k'], g[t + 'eta'], g['cfg_b_prec'], bpm)
        assert np.allclose(A, g[t + 'beta_A'], rtol=1e-12, atol=1e-12)
        assert np.allclose(r, g[t + 'beta_r'], rtol=1e-11, atol=1e-11)
        draw, _, st = oracle.precision_mvnorm(r, A, g[t + 'beta_eps'])
        assert st == 0
        assert np.allclose(draw, g[t + 'beta'], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_alpha_conditional_and_exists(oracle, case):
    g = load_golden(case)
    sites, visits = g['sites'], g['visits']
    site_ptr = np.concatenate([[0], np.cumsum(visits)])
    obs = np.isin(sites, g['cfg_obs'])
    apm = g['cfg_a_prec'] @ g['cfg_a_mu']
    for it in _iters(g):
        t = f'it{it}_'
        z = g[t + 'oa_z']
        exists_site = (obs | (z[sites] != 0)).astype(np.uint8)
        # same SET of sites as the reference's list (logit.py:187-188); its order is obs first
        assert sorted(sites[exists_site.astype(bool)].tolist()) == sorted(g[t + 'exists'].tolist())
        n_obs = int(obs.sum())
        assert g[t + 'exists'][:n_obs].tolist() == g['cfg_obs'].tolist()
        # scatter the reference-ordered omega_a back to flat row order
        omega_flat = np.zeros(int(site_ptr[-1]))
        pos = {int(s): i for i, s in enumerate(sites)}
        cur = 0
        for s in g[t + 'exists']:
            i = pos[int(s)]
            v = int(visits[i])
            omega_flat[site_ptr[i]:site_ptr[i] + v] = g[t + 'omega_a'][cur:cur + v]
            cur += v
        assert cur == g[t + 'omega_a'].size
        A, r = oracle.alpha_system(site_ptr, exists_site, g['W_flat'], g['y_flat'], omega_flat, g['cfg_a_prec'], apm)
        assert np.allclose(A, g[t + 'alpha_A'], rtol=1e-12, atol=1e-12)
        assert np.allclose(r, g[t + 'alpha_r'], rtol=1e-11, atol=1e-11)
        draw, _, st = oracle.precision_mvnorm(r, A, g[t + 'alpha_eps'])
        assert st == 0
        assert np.allclose(draw, g[t + 'alpha'], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_z_conditional(oracle, case):
    g = load_golden(case)
    sites, visits = g['sites'], g['visits']
    site_ptr = np.concatenate([[0], np.cumsum(visits)])
    pos = {int(s): i for i, s in enumerate(sites)}
    for it in _iters(g):
        t = f'it{it}_'
        z = g[t + 'oa_z'].copy()  # z before the update
        beta, eta, alpha = g[t + 'beta'], g[t + 'eta'], g[t + 'alpha']
        near = 0
        for j, s in enumerate(g['cfg_not_obs']):
            i = pos[int(s)]
            pr = oracle.z_prob(g['X'][s], beta, eta[s], g['W_flat'][site_ptr[i]:site_ptr[i + 1]], alpha)
            u = g[t + 'z_u_no'][j]
            near += abs(u - pr) < 1e-12
            z[s] = float(u < pr)
        for j, s in enumerate(g['cfg_not_surveyed']):
            pr = oracle.lib().orc_expit(float(g['X'][s] @ beta + eta[s]))
            u = g[t + 'z_u_ns'][j]
            near += abs(u - pr) < 1e-12
            z[s] = float(u < pr)
        assert near == 0
        assert np.array_equal(z, g[t + 'z'])
        assert np.array_equal(z - 0.5, g[t + 'k'])


def test_native_helpers(oracle):
    g = load_golden('native_helpers')
    for d in range(1, 9):
        draw, work, st = oracle.precision_mvnorm(g[f'mvn{d}_b'], g[f'mvn{d}_prec'], g[f'mvn{d}_eps'])
        assert st == 0
        assert np.allclose(draw, g[f'mvn{d}_draw'], rtol=1e-11, atol=1e-13)
        # the reference leaves U' in the lower triangle of its (row-major) input
        assert np.allclose(np.triu(work), np.tril(g[f'mvn{d}_prec_after']).T, rtol=1e-12, atol=1e-13)
    out = oracle.ensure_sums_to_zero(g['proj_x'], g['proj_z'])
    assert np.allclose(out, g['proj_out'], rtol=0, atol=1e-14)


def test_cholesky_failure_is_reported(oracle):
    _, _, st = oracle.precision_mvnorm(np.zeros(2), np.array([[1.0, 2.0], [2.0, 1.0]]), np.zeros(2))
    assert st == 2


@pytest.mark.parametrize('case', ['ref_rsr150_r05', 'ref_rsr150_q10'])
def test_reduced_rank_theta_conditional_matches_the_reference(oracle, case):
    """LogitRSRGibbs (reference gibbs/logit.py:269-485): the oracle's theta conditional fed the standard normals
    the reference consumed reproduces its theta and spatial = K theta; tau's rate uses theta' K'QK theta; the
    host's Moran basis spans the reference's (eigenvectors are unique up to sign / rotation inside clusters)."""
    from occuspytial_amd._problem import FlatProblem
    g = load_golden(case)
    n = g['X'].shape[0]
    r = int(g['rsr_dim'])
    K, Qr, E = g['rsr_K'], g['rsr_Q'], g['rsr_eigen']
    assert np.allclose(E @ E.T, Qr, atol=1e-9)
    it = 0
    while f'it{it}_theta' in g:
        t = f'it{it}_'
        b = g[t + 'eta_k'] - g[t + 'omega_b'] * (g['X'] @ g[t + 'eta_beta'])
        eps = g[t + 'eta_eps']
        theta, code = oracle.rsr_theta(K, Qr, E, b, g[t + 'omega_b'], float(g[t + 'tau']), eps[r:], eps[:r])
        assert code == 0
        assert np.allclose(theta, g[t + 'theta'], rtol=1e-9, atol=1e-11)
        assert np.allclose(K @ theta, g[t + 'spatial'], rtol=1e-9, atol=1e-11)
        rate = 0.5 * g[t + 'tau_theta'] @ Qr @ g[t + 'tau_theta'] + float(g['cfg_tau_rate'])
        assert np.isclose((1.0 / rate) * float(g[t + 'tau_g']), float(g[t + 'tau']), rtol=1e-13)
        it += 1
    assert it >= 2
    # the host's own basis (FlatProblem.enable_rsr follows _configure_rsr) against the reference's
    Q = sparse.csr_matrix((g['Q_data'], g['Q_indices'], g['Q_indptr']), shape=(n, n))
    W, y, cur = {}, {}, 0
    for s, v in zip(g['sites'], g['visits']):
        W[int(s)], y[int(s)] = g['W_flat'][cur:cur + v], g['y_flat'][cur:cur + v]
        cur += v
    prob = FlatProblem(Q, W, g['X'], y)
    mine = prob.enable_rsr(q=r) if case.endswith('q10') else prob.enable_rsr()
    assert mine['dim'] == r and prob.tau_shape == float(g['cfg_tau_shape'])
    assert np.allclose(mine['K'] @ mine['K'].T, K @ K.T, atol=1e-8)       # same column space
    assert np.allclose(mine['E'] @ mine['E'].T, mine['Q'], atol=1e-9)
