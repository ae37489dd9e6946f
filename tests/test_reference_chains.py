"""Whole-chain DISTRIBUTIONAL parity with chains of the reference itself.

``tests/golden/refchain_*.npz`` hold alpha / beta / tau draws of the reference's own
``LogitICARGibbs(...).sample(6000, burnin=1000, chains=4)`` (``tests/golden/make_golden.py --chains-only``: reference
``gibbs/base.py:243-291``, ``gibbs/parallel.py:4-42``, ``gibbs/logit.py:254-266`` with its dense eigenfactor prior draw
``logit.py:66-67,77`` and scipy MINRES; PG(1, z) from the defining series, the only stand-in).  The build differs from
that code in exactly two substitutions -- the edge form ``Q = B'B`` of the prior term and its own Polya-Gamma sampler
-- and both are shared by the CPU oracle and the device, so oracle-vs-device tests cannot see an error in them.
These tests can: 4 chains of the build against the 4 reference chains, for EVERY recorded coordinate that mixes
(ESS > 400 on both sides): split R-hat over the 8 chains < 1.05, means within 3 standard errors, standard deviations
within 4 (standard errors: the larger of the autocorrelation-based and the between-chain estimate).  The CPU half (oracle chains) runs in the ``not gpu`` suite; the device half is
``tests/test_gpu_api.py::test_posterior_agrees_with_reference_chains``.
"""
import numpy as np
import pytest

from .conftest import load_golden

REFCHAIN_CASES = {   # chain fixture -> (problem fixture, coordinates that MUST qualify for the comparison)
    # default hyper-parameters: tau wanders over decades and does not mix within 5 000 draws -- in the reference's own
    # chains too (R-hat 1.1-1.3 among them) -- and beta follows it; the detection coefficients do mix
    'refchain_queen150_ragged': ('ref_queen150_ragged', ('alpha0', 'alpha1')),
    'refchain_queen400_v3': ('ref_queen400_v3', ('alpha0', 'alpha1')),
    # informative Gamma(25, 25) prior on tau: every recorded coordinate mixes, tau included
    'refchain_queen150_tauprior': ('ref_queen150_ragged', ('alpha0', 'alpha1', 'beta0', 'beta1', 'beta2', 'tau')),
    'refchain_queen400_tauprior': ('ref_queen400_v3', ('alpha0', 'alpha1', 'beta0', 'beta1', 'tau')),
    'refchain_graph300_weighted_tauprior': ('ref_graph300_weighted', ('alpha0', 'alpha1', 'alpha2', 'beta0', 'beta1', 'tau')),
}


def problem_of(case):
    """(Q, W, X, y, hparams) of a reference-chain fixture: the inputs live in the per-conditional fixture of the same
    problem, the hyper-parameters (if any) with the chains."""
    from scipy import sparse
    g, ch = load_golden(REFCHAIN_CASES[case][0]), load_golden(case)
    n = g['X'].shape[0]
    Q = sparse.csr_matrix((g['Q_data'], g['Q_indices'], g['Q_indptr']), shape=(n, n))
    W, y, cur = {}, {}, 0
    for s, v in zip(g['sites'], g['visits']):
        W[int(s)] = g['W_flat'][cur:cur + v]
        y[int(s)] = g['y_flat'][cur:cur + v]
        cur += v
    hp = {k[3:]: (float(ch[k]) if ch[k].ndim == 0 else ch[k]) for k in ch if k.startswith('hp_')} or None
    return Q, W, g['X'], y, hp, ch


def coordinates(alpha, beta, tau):
    out = {f'alpha{j}': alpha[:, :, j] for j in range(alpha.shape[2])}
    out.update({f'beta{j}': beta[:, :, j] for j in range(beta.shape[2])})
    out['tau'] = tau
    return out


def _between_chain_se(x):
    """Standard error of the pooled mean from the spread of the chain means (chains as batches): does not rely on
    the within-chain autocorrelation estimate, so it stays honest when a slowly mixing companion (tau) puts the chains
    in different regimes."""
    return x.mean(axis=1).std(ddof=1) / np.sqrt(x.shape[0])


def compare_with_reference(case, alpha, beta, tau):
    """Asserts the distributional criteria for every coordinate with ESS > 400 on both sides; returns
    {coordinate: (split R-hat over the 8 chains, |dmean| / se, |dsd| / se, ESS reference, ESS ours)}.
    se of a mean: the larger of the MCSE (autocorrelation-based) and the between-chain standard error, both sides
    combined in quadrature; se of a standard deviation: the larger of sd / sqrt(2 ESS) and the between-chain one."""
    from occuspytial_amd import diagnostics as dg
    ch = load_golden(case)
    ref = coordinates(ch['alpha'].astype(float), ch['beta'].astype(float), ch['tau'].astype(float))
    ours = coordinates(np.asarray(alpha), np.asarray(beta), np.asarray(tau))
    report = {}
    for name, r in ref.items():
        o = ours[name]
        ess_r, ess_o = dg.ess(r), dg.ess(o)
        if ess_r < 400 or ess_o < 400:
            assert name not in REFCHAIN_CASES[case][1], (case, name, ess_r, ess_o)
            continue
        rhat = dg.rhat(np.concatenate([r, o]))
        se = np.hypot(max(dg.mcse_mean(r), _between_chain_se(r)), max(dg.mcse_mean(o), _between_chain_se(o)))
        dmean = abs(r.mean() - o.mean()) / se
        sd_r, sd_o = r.std(ddof=1), o.std(ddof=1)
        se_sd_r = max(sd_r / np.sqrt(2 * ess_r), r.std(axis=1, ddof=1).std(ddof=1) / np.sqrt(r.shape[0]))
        se_sd_o = max(sd_o / np.sqrt(2 * ess_o), o.std(axis=1, ddof=1).std(ddof=1) / np.sqrt(o.shape[0]))
        dsd = abs(sd_r - sd_o) / np.hypot(se_sd_r, se_sd_o)
        report[name] = (float(rhat), float(dmean), float(dsd), float(ess_r), float(ess_o))
        assert rhat < 1.05, (case, name, report[name])
        assert dmean < 3.0, (case, name, report[name])
        assert dsd < 4.0, (case, name, report[name])
    assert set(REFCHAIN_CASES[case][1]) <= set(report), (case, sorted(report))
    return report


def test_fixture_facts():
    for case in REFCHAIN_CASES:
        ch = load_golden(case)
        assert ch['alpha'].shape[:2] == (4, 5000) and ch['tau'].shape == (4, 5000) and int(ch['size']) == 6000
        assert np.all(ch['tau'] > 0) and np.all(np.isfinite(ch['beta']))


@pytest.mark.parametrize('case', list(REFCHAIN_CASES))
def test_oracle_chains_agree_with_reference_chains(oracle, case):
    """The CPU restatement as a whole sampler (edge-form prior, own PG, own Philox streams) against the reference's
    chains: pins the two substitutions at the level of the posterior."""
    from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
    Q, W, X, y, hp, ch = problem_of(case)
    prob = FlatProblem(Q, W, X, y, hp)
    size, burnin = int(ch['size']), int(ch['burnin'])
    A, B, T = [], [], []
    for g in chain_generators(4242, 4):
        st = default_start(g, prob)
        orc = oracle.OracleSampler(prob, int(g.bit_generator.random_raw()))
        orc.set_start(st['alpha'], st['beta'], st['tau'], st['eta'])
        a, b, t = orc.run(size, burnin)
        A.append(a); B.append(b); T.append(t)
    rep = compare_with_reference(case, np.stack(A), np.stack(B), np.stack(T))
    print(case, {k: tuple(round(x, 2) for x in v) for k, v in rep.items()})


def test_oracle_dense_eigenfactor_mode_is_the_references_prior_draw(oracle):
    """The oracle's ``dense_eigen`` mode -- the prior term exactly as the reference forms it, ``E (sqrt(tau) eps_2)`` with
    ``E`` from the dense ``eigh`` of Q (logit.py:64-67, 77); it is what bench.py's ``cpu_baseline`` times: (1) the
    right-hand side it builds equals the numpy expression on the same normals; (2) its chains agree with the
    reference's chains like the edge-form chains do (same law, N(0, tau Q), by ``E E' = Q = B'B``)."""
    from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
    case = 'refchain_queen150_tauprior'
    Q, W, X, y, hp, ch = problem_of(case)
    prob = FlatProblem(Q, W, X, y, hp)
    E = oracle.dense_eigenfactor(prob.Q)
    assert E.shape == (prob.n, prob.n - 1) and np.abs(E @ E.T - prob.Q.toarray()).max() < 1e-10
    # (1) one eta update by hand
    g = chain_generators(7, 1)[0]
    st = default_start(g, prob)
    key = int(g.bit_generator.random_raw())
    orc = oracle.OracleSampler(prob, key)
    orc.set_dense_eigen(E)
    orc.set_start(st['alpha'], st['beta'], st['tau'], st['eta'])
    orc.update('omega_b'); orc.update('tau'); orc.update('eta')
    L = oracle.lib()
    n, it = prob.n, int(orc.get('iter'))
    eps1 = np.array([L.orc_block_normal(key, i, 0, it, 3) for i in range(n)])
    eps2 = np.array([L.orc_block_normal(key, j, 0, it, 10) for j in range(n - 1)])
    om, tau = orc.get('omega_b'), orc.get('tau')
    want = (orc.get('k') - om * (prob.X @ st['beta'])) + np.sqrt(om) * eps1 + E @ (np.sqrt(tau) * eps2)
    assert np.abs(orc.get('rhs') - want).max() <= 1e-12 * np.abs(want).max()
    assert abs(orc.get('eta').sum()) < 1e-9
    # (2) whole chains in dense mode against the reference's
    size, burnin = int(ch['size']), int(ch['burnin'])
    A, B, T = [], [], []
    for g in chain_generators(99, 4):
        st = default_start(g, prob)
        orc = oracle.OracleSampler(prob, int(g.bit_generator.random_raw()))
        orc.set_dense_eigen(E)
        orc.set_start(st['alpha'], st['beta'], st['tau'], st['eta'])
        a, b, t = orc.run(size, burnin)
        A.append(a); B.append(b); T.append(t)
    compare_with_reference(case, np.stack(A), np.stack(B), np.stack(T))
