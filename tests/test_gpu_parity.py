"""Parity of the HIP engine (through the C ABI) with the CPU oracle.  Needs an MI355X: ``-m gpu``.

Bar: every conditional of an iteration, computed from the same state and the same Philox streams,
agrees with the oracle to FP64 rounding amplified by the solver (stated per quantity below); z and
the MINRES iteration count agree exactly.
"""
import numpy as np
import pytest
from scipy import sparse

from .conftest import load_golden

pytestmark = pytest.mark.gpu

KEY = 0x9E3779B97F4A7C15

# relative tolerances (max-norm, relative to the max magnitude of the oracle's vector)
TOL = {
    'omega_b': 1e-10, 'omega_a': 1e-10,  # PG(1,z): same accept/reject path, libm-level differences
    'tau': 1e-11, 'rhs': 1e-11,
    'xz': 5e-8,                          # 2n MINRES: rounding in the reductions and in the scalar recurrence (the device
                                         # divides once by gamma / beta and multiplies, scipy and the oracle divide each
                                         # time) amplified by the recurrence; measured worst case 1.4e-8
    'eta': 1e-7,                         # eta = x - (sum x / sum z) z cancels leading digits of xz (|eta| << |x|);
                                         # both are far inside the solver's own stopping tolerance (rtol 1e-5)
    'beta': 1e-7, 'alpha': 1e-8,         # beta's right-hand side carries eta's difference (logit.py:128-136)
}


def _problem_from_golden(name):
    from occuspytial_amd._problem import FlatProblem
    g = load_golden(name)
    n = g['X'].shape[0]
    Q = sparse.csr_matrix((g['Q_data'], g['Q_indices'], g['Q_indptr']), shape=(n, n))
    W, y, cur = {}, {}, 0
    for s, v in zip(g['sites'], g['visits']):
        W[int(s)] = g['W_flat'][cur:cur + v]
        y[int(s)] = g['y_flat'][cur:cur + v]
        cur += v
    hp = {k[3:]: g[k] for k in g if k.startswith('hp_')} or None
    start = dict(alpha=g['start_alpha'], beta=g['start_beta'], tau=float(g['start_tau']), eta=g['start_eta'])
    return FlatProblem(Q, W, g['X'], y, hp), start


def _rel(a, b):
    a, b = np.atleast_1d(a), np.atleast_1d(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _existing_rows(prob, exists_flag):
    rows = [np.arange(prob.site_ptr[i], prob.site_ptr[i + 1]) for i in np.flatnonzero(exists_flag)]
    return np.concatenate(rows) if rows else np.zeros(0, dtype=int)


def _solve_sensitivity(orc, prob, x0):
    """How far EQUALLY VALID evaluations of the solve just made lie apart: (a) the oracle's own iterate against scipy's
    ``minres`` on the same system -- the same arithmetic (a division and a square root wherever ``minres.py`` has one) with
    another summation order -- and (b) the oracle's iterate when its right-hand side is perturbed by 1e-13 relative.  The
    joint system tau Q + diag(omega_b) is close to singular when tau is small (its z-half solves A z = 1 with A ~ tau Q), and
    the Lanczos recurrence amplifies rounding there: with tau = 7e-5 (fixture ref_queen150_hparams, iteration 5) scipy and
    the oracle differ by 8e-7.  Where the solve is that sensitive the device cannot be held to 5e-8; it is held to twice
    this spread instead (iteration counts stay exact)."""
    from scipy.sparse.linalg import minres
    from oracle.occ_oracle import minres_joint
    rhs, om, tau = orc.get('rhs'), orc.get('omega_b'), float(orc.get('tau'))
    n = prob.n
    noise = np.random.default_rng(12345).standard_normal(rhs.size)
    a = minres_joint(prob.Q, om, tau, rhs, x0=x0)[0]
    b = minres_joint(prob.Q, om, tau, rhs * (1.0 + 1e-13 * noise), x0=x0)[0]
    A = sparse.block_diag([tau * prob.Q + sparse.diags(om)] * 2).tocsr()
    c = minres(A, np.concatenate([rhs, np.ones(n)]), x0=np.array(x0), rtol=1e-5, maxiter=10 * n)[0]
    return max(_rel(b, a), _rel(c, a))


def _compare_iteration(eng, orc, prob, chain=0, x0=None):
    worst = {}
    sens = _solve_sensitivity(orc, prob, x0) if x0 is not None else 0.0
    for name in ('omega_b', 'tau', 'rhs', 'xz', 'eta', 'beta', 'alpha'):
        worst[name] = _rel(eng.get(name, chain), orc.get(name))
        tol = TOL[name]
        if name in ('xz', 'eta', 'beta'):      # what the solve's iterate enters (alpha does not: logit.py:219-224)
            tol = max(tol, 2.0 * sens * (TOL[name] / TOL['xz']))
        assert worst[name] < tol, (name, worst[name], tol, sens)
    assert int(eng.get('minres_itn', chain)) == int(orc.get('minres_itn'))
    # the oracle's `exists` is the set used by the omega_a update (z before its own update); the
    # engine derives `exists` from the current z, i.e. what the NEXT omega_a update will use
    ex_o = orc.get('exists')
    rows = _existing_rows(prob, ex_o.astype(bool))
    z_now = orc.get('z')
    assert np.array_equal(eng.get('exists', chain),
                          (prob.obs_site.astype(bool) | (z_now[prob.site_id] != 0)).astype(float))
    worst['omega_a'] = _rel(eng.get('omega_a', chain)[rows], orc.get('omega_a')[rows])
    assert worst['omega_a'] < TOL['omega_a']
    assert np.array_equal(eng.get('z', chain), orc.get('z'))
    assert np.array_equal(eng.get('k', chain), orc.get('k'))
    return worst


@pytest.fixture(params=['persistent', 'launch_per_step'])
def solve_mode(request, monkeypatch):
    """The eta solve has two implementations of the same arithmetic: one persistent launch (k_iter, taken
    when all workgroups of all chains fit on the device) and one launch per MINRES step (k_minres)."""
    if request.param == 'launch_per_step':
        monkeypatch.setenv('OCC_NO_PERSISTENT', '1')
    else:
        monkeypatch.delenv('OCC_NO_PERSISTENT', raising=False)
    return request.param


@pytest.mark.parametrize('case', ['ref_queen150_ragged', 'ref_queen150_hparams', 'ref_rook400_v3',
                                  'ref_queen400_v3', 'ref_graph300_weighted'])
def test_lockstep_iterations_match_oracle(oracle, case, solve_mode):
    """Six iterations; after each one every conditional's output is compared, then the engine is
    re-seated on the oracle's state so that each iteration is tested from identical inputs."""
    from occuspytial_amd._engine import Engine
    prob, start = _problem_from_golden(case)
    eng = Engine(prob, [KEY])
    if solve_mode == 'launch_per_step':
        assert not eng.stats()['persistent_solve']
    else:  # (ref_graph300_weighted has rows of more than 8 off-diagonals: the 16-wide window of k_iter)
        assert eng.stats()['persistent_solve']
    orc = oracle.OracleSampler(prob, KEY)
    eng.set_start(0, **start)
    orc.set_start(**start)
    for it in range(6):
        x0 = orc.get('xz')
        eng.step()
        orc.step()
        _compare_iteration(eng, orc, prob, x0=x0)
        for name in ('alpha', 'beta', 'tau', 'eta', 'z', 'xz'):
            eng.set(name, orc.get(name))
    eng.close()


def test_free_running_chain_tracks_oracle(oracle, solve_mode):
    """40 iterations without re-seating: recorded alpha/beta/tau stay together."""
    from occuspytial_amd._engine import Engine
    prob, start = _problem_from_golden('ref_queen400_v3')
    eng = Engine(prob, [KEY])
    orc = oracle.OracleSampler(prob, KEY)
    eng.set_start(0, **start)
    orc.set_start(**start)
    a, b, t = eng.run(40, 5)
    ao, bo, to = orc.run(40, 5)
    assert a.shape == (1, 35, prob.q)
    assert _rel(a[0], ao) < 1e-6 and _rel(b[0], bo) < 1e-6 and _rel(t[0], to) < 1e-6
    assert np.array_equal(eng.get('z'), orc.get('z'))
    eng.close()


def test_graph_replay_equals_eager_stepping():
    """occ_run (hipGraph replay on two streams) and occ_step (eager, one stream) run the same arithmetic: bitwise."""
    from occuspytial_amd._engine import Engine
    prob, start = _problem_from_golden('ref_queen150_ragged')
    e1, e2 = Engine(prob, [KEY]), Engine(prob, [KEY])
    e1.set_start(0, **start)
    e2.set_start(0, **start)
    a, b, t = e1.run(25, 0)
    for i in range(25):
        e2.step()
        assert np.array_equal(e2.get('alpha'), a[0, i]) and np.array_equal(e2.get('beta'), b[0, i])
        assert e2.get('tau') == t[0, i]
    st = e1.stats()
    assert st['graph_launches'] >= 20 and st['iterations'] == 25
    e1.close()
    e2.close()


def test_batched_chains_equal_single_chain_runs():
    """Chain c of a 3-chain batch == the same key run alone (chains share nothing but the inputs)."""
    from occuspytial_amd._engine import Engine
    prob, start = _problem_from_golden('ref_graph300_weighted')
    keys = [KEY, KEY ^ 0xABCDEF, 12345]
    rng = np.random.default_rng(3)
    starts = [dict(alpha=rng.standard_normal(prob.q), beta=rng.standard_normal(prob.p), tau=1.0 + c,
                   eta=(lambda e: e - e.mean())(rng.standard_normal(prob.n))) for c in range(3)]
    batch = Engine(prob, keys)
    for c in range(3):
        batch.set_start(c, **starts[c])
    A, B, T = batch.run(20, 4)
    for c in range(3):
        solo = Engine(prob, [keys[c]])
        solo.set_start(0, **starts[c])
        a, b, t = solo.run(20, 4)
        assert np.array_equal(a[0], A[c]) and np.array_equal(b[0], B[c]) and np.array_equal(t[0], T[c])
        assert np.array_equal(solo.get('eta'), batch.get('eta', c))
        solo.close()
    batch.close()


def test_calls_with_different_windows_steps_and_reads_between_them_equal_one_run():
    """Every occ_run opens its window of iterations (first / last iteration, burn-in, rows to keep) in the chains' scalars: on
    the device, by the kernel that snapshots the state (round 3: no copy engine between a call's entry and its first kernel),
    or by an upload when the call cannot take that way (occ_step; a call after something else touched the state).  The
    recorded rows come back through page-locked staging that grows between calls.  A run cut into calls of different
    lengths and burn-ins, with single steps, state reads, state writes and checkpoint / restore round trips in between, must
    leave the chains where ONE run leaves them and return the same rows."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(60, 60, visits=3, p=2, q=2, random_state=9)
    prob = FlatProblem(Q, W, X, y)
    keys = [KEY + 3 * c for c in range(4)]
    starts = [_random_start(prob, 21 + c) for c in range(4)]
    one = Engine(prob, keys)
    cut = Engine(prob, keys)
    for c in range(4):
        one.set_start(c, **starts[c])
        cut.set_start(c, **starts[c])
    A, B, T = one.run(61, 0)
    rows = []
    # ('set',): a state write between calls (the value it already has: the host's mirror of the chains' scalars is void after
    # it); ('ckpt',): checkpoint and restore (every chain's state leaves and re-enters the device)
    plan = [('run', 2, 0), ('set',), ('run', 7, 3), ('step',), ('ckpt',), ('run', 1, 0), ('get',), ('run', 30, 29), ('set',), ('step',),
            ('step',), ('ckpt',), ('run', 18, 5)]
    for item in plan:
        if item[0] == 'run':
            a, b, t = cut.run(item[1], item[2])
            rows.append((item[1], item[2], a, b, t))
        elif item[0] == 'step':
            cut.step()
            rows.append((1, 1, None, None, None))
        elif item[0] == 'set':
            for c in (1, 3):
                cut.set('alpha', cut.get('alpha', c), c)
                cut.set('tau', cut.get('tau', c), c)
        elif item[0] == 'ckpt':
            cut.restore(cut.checkpoint())
        else:
            assert np.all(np.isfinite(cut.get('eta', 2)))
    assert sum(r[0] for r in rows) == 61
    at = 0
    for n, burn, a, b, t in rows:
        if a is not None:
            assert a.shape[1] == n - burn
            assert np.array_equal(a, A[:, at + burn:at + n]) and np.array_equal(b, B[:, at + burn:at + n]) and np.array_equal(t, T[:, at + burn:at + n])
        at += n
    for c in range(4):
        for name in ('eta', 'xz', 'z'):
            assert np.array_equal(one.get(name, c), cut.get(name, c))
    assert one.stats()['fused_fallbacks'] == 0 and cut.stats()['fused_fallbacks'] == 0
    one.close()
    cut.close()


@pytest.mark.parametrize('lattice, chains, iters', [((20, 20), 3, 30), ((100, 100), 4, 60), ((37, 91), 6, 40), ((60, 60), 8, 30), ((30, 40), 10, 20), ((50, 50), 19, 20)])
def test_persistent_solve_is_bit_identical_to_launch_per_step(monkeypatch, lattice, chains, iters):
    """k_iter exchanges g between the workgroups of a chain -- one XCD per chain: plain stores, L1-bypassing loads
    and one arrival flag per workgroup through that XCD's L2; any placement: write-through stores, L1-bypassing
    loads and an agent-scope arrival counter -- and one stale or torn value would change the bits of eta.  Same
    scalars, same contractions, same summation order as k_minres: everything must agree exactly, including the
    number of MINRES iterations of every solve."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(*lattice, visits=3, p=2, q=2, random_state=5)
    prob = FlatProblem(Q, W, X, y)
    keys = [KEY + 7 * c for c in range(chains)]
    starts = [_random_start(prob, 11 + c) for c in range(chains)]
    out = {}
    modes = {'xcd_local': {}, 'any_placement': {'OCC_NO_XCD_LOCAL': '1'}, 'launch_per_step': {'OCC_NO_PERSISTENT': '1'}}
    for mode, env in modes.items():
        for k in ('OCC_NO_PERSISTENT', 'OCC_NO_XCD_LOCAL'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = Engine(prob, keys)
        # 2: one XCD per chain (at most 64 workgroups per chain; more than 8 chains: launches of eight one behind the other),
        # 1: any placement, 0: launch per step
        assert eng.stats()['persistent_solve'] == {'xcd_local': 2, 'any_placement': 1, 'launch_per_step': 0}[mode]
        for c in range(chains):
            eng.set_start(c, **starts[c])
        rec = eng.run(iters, 0)
        state = [(eng.get('eta', c), eng.get('xz', c), eng.get('z', c), eng.get('minres_itn', c)) for c in range(chains)]
        out[mode] = (rec, state, eng.stats()['krylov_mean'])
        eng.close()
    for mode in ('xcd_local', 'any_placement'):
        for u, v in zip(out[mode][0], out['launch_per_step'][0]):
            assert np.array_equal(u, v)
        for su, sv in zip(out[mode][1], out['launch_per_step'][1]):
            for u, v in zip(su, sv):
                assert np.array_equal(u, v)
        assert out[mode][2] == out['launch_per_step'][2]


@pytest.mark.parametrize('lattice, chains, tiles, iters', [((250, 250), 1, '1', 30), ((250, 250), 1, '2', 30), ((130, 170), 3, '2', 24),
                                                           ((61, 67), 2, '1', 24), ((180, 200), 2, '4', 20), ((90, 110), 1, '3', 20),
                                                           ((250, 250), 1, None, 20), ((500, 500), 1, None, 12)])
def test_tile_looping_persistent_solve_is_bit_identical_to_launch_per_step(monkeypatch, lattice, chains, tiles, iters):
    """k_tiles (occ_tiles.hpp, BASELINE config 4's path): tiles of 256 sites with their vectors in LDS, one or two tiles per
    workgroup, p exchanged through a canary-polled buffer (plain stores inside an XCD's band, write-through at the band
    edges), one record per workgroup and step.  A stale, torn or early-read value anywhere would change the bits of eta.
    Against the launch-per-step kernels in the same layout (256-thread blocks, sums grouped by T): every record, eta, xz,
    z and every solve's iteration count agree exactly -- forced on mid-size lattices (one to four tiles per workgroup, several
    chains, a ragged last tile), and at 250x250 and 500x500 where it is what the engine takes by itself."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(*lattice, visits=3, p=2, q=2, random_state=5)
    prob = FlatProblem(Q, W, X, y)
    keys = [KEY + 7 * c for c in range(chains)]
    starts = [_random_start(prob, 11 + c) for c in range(chains)]
    if tiles is not None:
        monkeypatch.setenv('OCC_FORCE_TILES', tiles)
    out = {}
    for mode in ('tiles', 'launch_per_step'):
        monkeypatch.delenv('OCC_NO_PERSISTENT', raising=False)
        if mode == 'launch_per_step':
            monkeypatch.setenv('OCC_NO_PERSISTENT', '1')
        eng = Engine(prob, keys)
        st = eng.stats()
        assert st['persistent_solve'] == (3 if mode == 'tiles' else 0) and st['threads_per_block'] == 256, st
        for c in range(chains):
            eng.set_start(c, **starts[c])
        rec = eng.run(iters, 0)
        eng.step()                                         # eager stepping through the same kernel
        state = [(eng.get('eta', c), eng.get('xz', c), eng.get('z', c), eng.get('minres_itn', c)) for c in range(chains)]
        out[mode] = (rec, state, eng.stats()['krylov_mean'], eng.stats()['fused_fallbacks'])
        eng.close()
    assert out['tiles'][3] == 0
    for u, v in zip(out['tiles'][0], out['launch_per_step'][0]):
        assert np.array_equal(u, v)
    for su, sv in zip(out['tiles'][1], out['launch_per_step'][1]):
        for u, v in zip(su, sv):
            assert np.array_equal(u, v)
    assert out['tiles'][2] == out['launch_per_step'][2]


@pytest.mark.parametrize('env', [{'OCC_EVENT_SYNC': '1'}, {'OCC_EVENT_SYNC': '1', 'OCC_STREAM_EVENTS': '1'}, {'OCC_CU_SPLIT': '0'},
                                 {'OCC_DEBUG_STREAMS_SERIALISED': '1'},
                                 {'OCC_NO_SIDE_STREAM': '1'}, {'OCC_EAGER_ONLY': '1'}, {'OCC_NO_XCD_LOCAL': '1'},
                                 {'OCC_NO_XCD_LOCAL': '1', 'OCC_CU_SPLIT': '0'},
                                 {'OCC_NO_PERSISTENT': '1', 'OCC_STREAM_EVENTS': '1'}, {'OCC_NO_PERSISTENT': '1', 'OCC_NO_SIDE_STREAM': '1'}])
def test_every_scheduling_mode_gives_the_same_chains(monkeypatch, env):
    """How an iteration is scheduled -- hand-overs by device counters (default), by event nodes, by stream
    events, without the CU partition, on one stream, launched eagerly, fused kernel or one launch per MINRES
    step -- never changes a bit of the result."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(30, 40, visits=3, p=2, q=2, random_state=2)
    prob = FlatProblem(Q, W, X, y)
    keys = [KEY, KEY + 11]
    starts = [_random_start(prob, 3 + c) for c in range(2)]

    def run():
        eng = Engine(prob, keys)
        for c in range(2):
            eng.set_start(c, **starts[c])
        rec = eng.run(33, 4)     # odd number of iterations: the graph holds two per replay
        rec2 = eng.run(10, 0)    # and a second call continues the same chains
        eta = [eng.get('eta', c) for c in range(2)]
        eng.close()
        return rec, rec2, eta

    for k in ('OCC_EVENT_SYNC', 'OCC_STREAM_EVENTS', 'OCC_CU_SPLIT', 'OCC_NO_SIDE_STREAM', 'OCC_EAGER_ONLY', 'OCC_NO_PERSISTENT'):
        monkeypatch.delenv(k, raising=False)
    ref = run()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    alt = run()
    for u, v in zip(ref[0] + ref[1], alt[0] + alt[1]):
        assert np.array_equal(u, v)
    for u, v in zip(ref[2], alt[2]):
        assert np.array_equal(u, v)


def test_wide_rows_fused_kernel_is_bit_identical_to_launch_per_step(monkeypatch):
    """Rows of 9-16 off-diagonals take k_iter's 16-wide neighbour window (one workgroup per CU): same bits as the
    launch-per-step path, whose k_minres handles the slots past its 8-wide prefetch window in a tail loop."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_graph_problem
    Q, W, X, y, *_ = make_graph_problem(n=1500, k=8, visits=4, p=2, q=2, random_state=7)
    assert 8 < np.diff(Q.indptr).max() - 1 <= 16
    prob = FlatProblem(Q, W, X, y)
    keys = [KEY + 3 * c for c in range(3)]
    starts = [_random_start(prob, 20 + c) for c in range(3)]
    out = {}
    for mode in ('persistent', 'launch_per_step'):
        if mode == 'launch_per_step':
            monkeypatch.setenv('OCC_NO_PERSISTENT', '1')
        else:
            monkeypatch.delenv('OCC_NO_PERSISTENT', raising=False)
        eng = Engine(prob, keys)
        assert bool(eng.stats()['persistent_solve']) == (mode == 'persistent')
        for c in range(3):
            eng.set_start(c, **starts[c])
        rec = eng.run(40, 0)
        out[mode] = (rec, [(eng.get('eta', c), eng.get('z', c), eng.get('minres_itn', c)) for c in range(3)])
        eng.close()
    for u, v in zip(out['persistent'][0], out['launch_per_step'][0]):
        assert np.array_equal(u, v)
    for su, sv in zip(out['persistent'][1], out['launch_per_step'][1]):
        for u, v in zip(su, sv):
            assert np.array_equal(u, v)


def test_krylov_cap_overflow_is_resumed_exactly(monkeypatch):
    """A captured graph with too few Krylov steps carries the unfinished solve into the next replay
    (same arithmetic, continued), so results equal an unconstrained run bit for bit."""
    from occuspytial_amd._engine import Engine
    prob, start = _problem_from_golden('ref_queen150_ragged')
    ref = Engine(prob, [KEY, KEY + 1])
    for c in range(2):
        ref.set_start(c, **start)
    A, B, T = ref.run(12, 0)
    monkeypatch.setenv('OCC_FORCE_KRYLOV_CAP', '4')
    monkeypatch.setenv('OCC_NO_PERSISTENT', '1')   # `ref` above runs the persistent solve: same bits again
    low = Engine(prob, [KEY, KEY + 1])
    assert not low.stats()['persistent_solve']
    for c in range(2):
        low.set_start(c, **start)
    a, b, t = low.run(12, 0)
    assert low.stats()['stalls'] > 0
    assert np.array_equal(a, A) and np.array_equal(b, B) and np.array_equal(t, T)
    ref.close()
    low.close()


def test_eta_solve_at_baseline_size_equals_the_reference_solver_call():
    """100x100 queen lattice (BASELINE config 2).  scipy is the reference's own dependency and is
    installed on the GPU box: the engine's joint solve must equal
    ``scipy.sparse.linalg.minres(P, [y;1], x0=previous)`` (reference logit.py:80-87) on the system the
    engine built -- same iteration count, same solution -- plus size-independent properties."""
    from scipy.sparse.linalg import minres

    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
    prob = FlatProblem(Q, W, X, y)
    rng = np.random.default_rng(1)
    eng = Engine(prob, [KEY])
    eta0 = rng.standard_normal(prob.n)
    eng.set_start(0, rng.standard_normal(2), rng.standard_normal(2), 0.7, eta0 - eta0.mean())
    n = prob.n
    x0 = None
    for _ in range(3):
        eng.step()
        eta, xz = eng.get('eta'), eng.get('xz')
        om, tau, rhs = eng.get('omega_b'), eng.get('tau'), eng.get('rhs')
        assert abs(eta.sum()) < 1e-8 * np.abs(eta).sum()          # sum-to-zero constraint
        L = tau * prob.Q + sparse.diags(om)
        P = sparse.block_diag((L, L), format='csc')
        cnt = [0]
        ref_xz, info = minres(P, np.concatenate([rhs, np.ones(n)]), x0=x0,
                              callback=lambda xk: cnt.__setitem__(0, cnt[0] + 1))
        assert info == 0 and cnt[0] == int(eng.get('minres_itn'))
        assert np.abs(xz - ref_xz).max() <= 1e-9 * np.abs(ref_xz).max()
        x0 = xz
        z = eng.get('z')
        assert set(np.unique(z)) <= {0.0, 1.0}
        assert np.all(z[prob.site_id[prob.obs_site.astype(bool)]] == 1.0)
        assert tau > 0 and np.all(om > 0)
    eng.close()


def _lockstep(oracle, prob, start, n_iter, key=KEY):
    from occuspytial_amd._engine import Engine
    eng = Engine(prob, [key])
    orc = oracle.OracleSampler(prob, key)
    eng.set_start(0, **start)
    orc.set_start(**start)
    for _ in range(n_iter):
        x0 = orc.get('xz')
        eng.step()
        orc.step()
        _compare_iteration(eng, orc, prob, x0=x0)
        for name in ('alpha', 'beta', 'tau', 'eta', 'z', 'xz'):
            eng.set(name, orc.get(name))
    eng.close()


def _lockstep_chains(oracle, prob, starts, keys, n_iter):
    """All-conditionals lock step of a BATCH of chains: chain c of the engine against its own oracle sampler."""
    from occuspytial_amd._engine import Engine
    eng = Engine(prob, keys)
    orcs = [oracle.OracleSampler(prob, k) for k in keys]
    for c, st in enumerate(starts):
        eng.set_start(c, **st)
        orcs[c].set_start(**st)
    worst = {}
    for _ in range(n_iter):
        eng.step()
        for c, orc in enumerate(orcs):
            x0 = orc.get('xz')
            orc.step()
            for k, v in _compare_iteration(eng, orc, prob, chain=c, x0=x0).items():
                worst[k] = max(worst.get(k, 0.0), v)
        for c, orc in enumerate(orcs):
            for name in ('alpha', 'beta', 'tau', 'eta', 'z', 'xz'):
                eng.set(name, orc.get(name), c)
    stats = eng.stats()
    eng.close()
    return worst, stats


def _random_start(prob, seed):
    rng = np.random.default_rng(seed)
    eta = rng.standard_normal(prob.n)
    return dict(alpha=rng.standard_normal(prob.q), beta=rng.standard_normal(prob.p), tau=0.9, eta=eta - eta.mean())


def test_irregular_adjacency_with_long_rows_and_unsurveyed_sites(oracle, solve_mode):
    """BASELINE config 5 in small: irregular areal graph (non-uniform row lengths, some rows longer than
    8 off-diagonals: the 16-wide neighbour window of k_iter, or -- launch-per-step -- the SELL slice table and the
    tail loop of k_minres), 10 visits, three covariates each, 5 % of the units never surveyed."""
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_graph_problem
    Q, W, X, y, *_ = make_graph_problem(n=700, k=8, visits=10, p=3, q=3, random_state=2)
    assert np.diff(Q.indptr).max() - 1 > 8
    for site in range(0, 700, 20):
        del W[site], y[site]
    prob = FlatProblem(Q, W, X, y)
    assert len(prob.not_surveyed) == 35
    _lockstep(oracle, prob, _random_start(prob, 4), 4)


def test_one_covariate_and_eight_covariates(oracle):
    """Edge sizes of the p x p / q x q systems: p = q = 1 and p = q = 8 (the supported maximum)."""
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    for d in (1, 8):
        Q, W, X, y, *_ = make_lattice_problem(9, 11, visits=4, p=d, q=d, random_state=d)
        prob = FlatProblem(Q, W, X, y)
        _lockstep(oracle, prob, _random_start(prob, d), 3)


def test_500x500_lattice_full_size_lockstep(oracle):
    """BASELINE config 4 at full size: 250 000 sites, 1.25 M visit rows (the reference cannot construct
    this case: dense eigenfactor).  Two lock-step iterations against the oracle, every conditional."""
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(500, 500, visits=5, p=2, q=2, random_state=0)
    prob = FlatProblem(Q, W, X, y)
    assert prob.n == 250_000 and prob.R == 1_250_000
    _lockstep(oracle, prob, _random_start(prob, 9), 2)


def test_headline_config_100x100_four_chains_lockstep(oracle):
    """BASELINE config 2 / the metric's workload at full size: 100x100 queen lattice, 5 visits, FOUR chains batched
    (the fused iteration kernel, one XCD per chain, 512-thread workgroups): three lock-step iterations, every
    conditional of every chain against its oracle; the measured worst-case deviations are asserted, not only the bounds."""
    from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
    prob = FlatProblem(Q, W, X, y)
    gens = chain_generators(10, 4)                      # bench.py's chains: same seeds, same default starts
    starts = [default_start(g, prob) for g in gens]
    keys = [int(g.bit_generator.random_raw()) for g in gens]
    worst, stats = _lockstep_chains(oracle, prob, starts, keys, 3)
    assert stats['persistent_solve'] == 2 and stats['n_chains'] == 4 and stats['fused_fallbacks'] == 0
    # the scalar recurrence divides once and multiplies: measured worst case 1.4e-8 on xz
    assert worst['xz'] < 3e-8 and worst['eta'] < 6e-8, worst
    print('100x100 x 4 chains lock step, worst relative deviations:', {k: float('%.2e' % v) for k, v in worst.items()})


def test_config5_full_size_four_chains_lockstep(oracle):
    """BASELINE config 5 at its stated size: irregular areal graph, 3 000 units, mean degree ~ 6 (rows of up to 16
    off-diagonals: k_iter's 16-wide window), 10 visits per unit, FOUR chains: three lock-step iterations."""
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_graph_problem
    Q, W, X, y, *_ = make_graph_problem(3000, 6, visits=10, p=2, q=2, random_state=0)
    deg = np.diff(Q.indptr) - 1
    assert Q.shape[0] == 3000 and 5.0 < deg.mean() < 8.0
    prob = FlatProblem(Q, W, X, y)
    assert prob.R == 30_000
    starts = [_random_start(prob, 40 + c) for c in range(4)]
    worst, stats = _lockstep_chains(oracle, prob, starts, [KEY + 13 * c for c in range(4)], 3)
    assert stats['n_chains'] == 4 and stats['persistent_solve'] >= 1


def test_eight_chains_of_config3_run_and_differ():
    """BASELINE config 3's chain count on one device: 8 chains of the 100x100 problem in one batch."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
    prob = FlatProblem(Q, W, X, y)
    gens = chain_generators(10, 8)
    starts = [default_start(g, prob) for g in gens]
    eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
    for c, st in enumerate(starts):
        eng.set_start(c, st['alpha'], st['beta'], st['tau'], st['eta'])
    a, b, t = eng.run(60, 10)
    assert a.shape == (8, 50, 2) and np.all(np.isfinite(a)) and np.all(np.isfinite(b)) and np.all(t > 0)
    assert len({round(float(v), 12) for v in t[:, -1]}) == 8
    for c in range(8):
        assert abs(eng.get('eta', c).sum()) < 1e-7 * np.abs(eng.get('eta', c)).sum()
    eng.close()


# ---- reduced-rank model (LogitRSRGibbs) ---------------------------------------------------------------------
def _rsr_problem(case):
    prob, start = _problem_from_golden(case)
    g = load_golden(case)
    K = np.ascontiguousarray(g['rsr_K'])
    prob.rsr = {'K': K, 'Q': np.ascontiguousarray(g['rsr_Q']), 'E': np.ascontiguousarray(g['rsr_eigen']), 'dim': K.shape[1]}
    prob.tau_shape = float(g['cfg_tau_shape'])
    return prob, start


@pytest.mark.parametrize('case', ['ref_rsr150_r05', 'ref_rsr150_q10'])
def test_reduced_rank_lockstep_iterations_match_oracle(oracle, case):
    """LogitRSRGibbs on the reference's own basis K (fixture): five iterations in lock step with the oracle, whose
    theta conditional is pinned to the reference (tests/test_oracle_golden.py); theta, eta = K theta, tau, beta,
    alpha, z after every iteration, then the engine is re-seated on the oracle's state."""
    from occuspytial_amd._engine import Engine
    prob, start = _rsr_problem(case)
    eng = Engine(prob, [KEY])
    orc = oracle.OracleSampler(prob, KEY)
    eng.set_start(0, **start)
    orc.set_start(**start)
    assert _rel(eng.get('eta'), orc.get('eta')) < 1e-13          # both form K theta from the start coefficients
    for it in range(5):
        eng.step()
        orc.step()
        for name, tol in (('omega_b', 1e-10), ('tau', 1e-11), ('theta', 1e-9), ('eta', 1e-9), ('beta', 1e-9), ('alpha', 1e-9)):
            assert _rel(eng.get(name), orc.get(name)) < tol, (it, name, _rel(eng.get(name), orc.get(name)))
        assert np.array_equal(eng.get('z'), orc.get('z'))
        for name in ('alpha', 'beta', 'tau', 'theta', 'z'):
            eng.set(name, orc.get(name))
    eng.close()


@pytest.mark.parametrize('q', [3, 16, 17, 33, 64, 97, 128])
def test_reduced_rank_every_block_count_matches_oracle(oracle, q):
    """The theta solve keeps the matrix in registers in 16-wide blocks (k_rsr_solve<2|4|6|8>): basis sizes at and
    around the block edges, up to the largest the kernel takes, three iterations in lock step with the oracle."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(20, 25, visits=3, p=2, q=2, random_state=5)
    prob = FlatProblem(Q, W, X, y)
    m = prob.enable_rsr(q=q)['dim']
    assert m == q
    rng = np.random.default_rng(q)
    start = dict(alpha=rng.standard_normal(2), beta=rng.standard_normal(2), tau=1.5, eta=0.3 * rng.standard_normal(m))
    eng = Engine(prob, [KEY])
    orc = oracle.OracleSampler(prob, KEY)
    eng.set_start(0, **start)
    orc.set_start(**start)
    for it in range(3):
        eng.step()
        orc.step()
        for name, tol in (('tau', 1e-11), ('theta', 1e-9), ('eta', 1e-9), ('beta', 1e-9), ('alpha', 1e-9)):
            assert _rel(eng.get(name), orc.get(name)) < tol, (it, name, _rel(eng.get(name), orc.get(name)))
        assert np.array_equal(eng.get('z'), orc.get('z'))
        for name in ('alpha', 'beta', 'tau', 'theta', 'z'):
            eng.set(name, orc.get(name))
    eng.close()


@pytest.mark.parametrize('q', [129, 160, 257, 333])
def test_reduced_rank_large_basis_matches_oracle(oracle, q):
    """Beyond 128 basis columns the m x m system lives in device memory and is factorised panel by panel (k_rsrb_*: 32-row
    panels; sizes at a panel edge, inside a panel, with a ragged last panel): three iterations in lock step with the
    oracle, then occ_run (graphs, two streams) against stepping for two chains.  The reference keeps every Moran
    eigenvector above its threshold -- about 13 % of a lattice's sites -- so its default arguments need this path from
    1 000 sites on (logit.py:415-446)."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(20, 25, visits=3, p=2, q=2, random_state=5)
    prob = FlatProblem(Q, W, X, y)
    m = prob.enable_rsr(q=q)['dim']
    assert m == q
    rng = np.random.default_rng(q)
    start = dict(alpha=rng.standard_normal(2), beta=rng.standard_normal(2), tau=1.5, eta=0.3 * rng.standard_normal(m))
    eng = Engine(prob, [KEY])
    orc = oracle.OracleSampler(prob, KEY)
    eng.set_start(0, **start)
    orc.set_start(**start)
    for it in range(3):
        eng.step()
        orc.step()
        for name, tol in (('tau', 1e-11), ('theta', 1e-8), ('eta', 1e-8), ('beta', 1e-8), ('alpha', 1e-9)):
            assert _rel(eng.get(name), orc.get(name)) < tol, (it, name, _rel(eng.get(name), orc.get(name)))
        assert np.array_equal(eng.get('z'), orc.get('z'))
        for name in ('alpha', 'beta', 'tau', 'theta', 'z'):
            eng.set(name, orc.get(name))
    eng.close()
    # three chains: a full pair of k_rsr_gram32's and a half-empty one; each of chains 1 and 2 alone gives the same bits
    keys = [KEY + c for c in range(3)]
    starts = [dict(alpha=rng.standard_normal(2), beta=rng.standard_normal(2), tau=1.0 + c, eta=0.3 * rng.standard_normal(m)) for c in range(3)]
    batch = Engine(prob, keys)
    for c in range(3):
        batch.set_start(c, **starts[c])
    A, B, T = batch.run(9, 1)
    for ch in (1, 2):
        solo = Engine(prob, [keys[ch]])
        solo.set_start(0, **starts[ch])
        for i in range(9):
            solo.step()
            if i >= 1:
                assert np.array_equal(solo.get('alpha'), A[ch, i - 1]) and np.array_equal(solo.get('beta'), B[ch, i - 1]) and solo.get('tau') == T[ch, i - 1]
        assert np.array_equal(solo.get('theta'), batch.get('theta', ch))
        solo.close()
    assert abs(batch.get('eta', 0) - prob.rsr['K'] @ batch.get('theta', 0)).max() < 1e-11
    batch.close()


def test_reduced_rank_basis_of_more_than_2048_columns(oracle):
    """The basis may have up to 4 096 columns (what the reference's own dense n x n set-up can reach: its default threshold
    keeps 13 % of the sites): 2 100 columns on a 50 x 50 lattice, two iterations in lock step with the oracle."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(50, 50, visits=2, p=2, q=2, random_state=6)
    prob = FlatProblem(Q, W, X, y)
    m = prob.enable_rsr(q=2100)['dim']
    assert m == 2100
    rng = np.random.default_rng(11)
    start = dict(alpha=rng.standard_normal(2), beta=rng.standard_normal(2), tau=1.5, eta=0.1 * rng.standard_normal(m))
    eng = Engine(prob, [KEY])
    orc = oracle.OracleSampler(prob, KEY)
    eng.set_start(0, **start)
    orc.set_start(**start)
    for it in range(2):
        eng.step()
        orc.step()
        for name, tol in (('tau', 1e-11), ('theta', 1e-8), ('eta', 1e-8), ('beta', 1e-8), ('alpha', 1e-9)):
            assert _rel(eng.get(name), orc.get(name)) < tol, (it, name, _rel(eng.get(name), orc.get(name)))
        assert np.array_equal(eng.get('z'), orc.get('z'))
        for name in ('alpha', 'beta', 'tau', 'theta', 'z'):
            eng.set(name, orc.get(name))
    eng.close()


@pytest.mark.parametrize('env', [{}, {'OCC_CU_SPLIT': '0'}, {'OCC_NO_SIDE_STREAM': '1'}, {'OCC_EVENT_SYNC': '1'}, {'OCC_DEBUG_STREAMS_SERIALISED': '1'}],
                         ids=['two-streams-flag-handovers', 'no-cu-partition', 'one-stream', 'no-flag-handovers', 'stream-probe-says-serialised'])
def test_reduced_rank_graph_replay_equals_stepping_and_batching(monkeypatch, env):
    """occ_run (graphs of two iterations; by default two streams on disjoint CUs handing over through device
    counters, k_rsr_gram opening the main sequence) == occ_step (one stream), and chain c of a batch == the same
    chain alone: bitwise, in every scheduling mode."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(24, 30, visits=3, p=2, q=2, random_state=4)
    prob = FlatProblem(Q, W, X, y)
    m = prob.enable_rsr(q=70)['dim']
    rng = np.random.default_rng(8)
    keys = [KEY + c for c in range(3)]
    starts = [dict(alpha=rng.standard_normal(2), beta=rng.standard_normal(2), tau=1.0 + c, eta=rng.standard_normal(m)) for c in range(3)]
    batch = Engine(prob, keys)
    for c in range(3):
        batch.set_start(c, **starts[c])
    A, B, T = batch.run(21, 2)
    solo = Engine(prob, [keys[1]])
    solo.set_start(0, **starts[1])
    rec = []
    for i in range(21):
        solo.step()
        rec.append((solo.get('alpha'), solo.get('beta'), solo.get('tau')))
    for i in range(2, 21):
        assert np.array_equal(rec[i][0], A[1, i - 2]) and np.array_equal(rec[i][1], B[1, i - 2]) and rec[i][2] == T[1, i - 2]
    assert np.array_equal(solo.get('theta'), batch.get('theta', 1)) and np.array_equal(solo.get('eta'), batch.get('eta', 1))
    assert abs(batch.get('eta', 0) - prob.rsr['K'] @ batch.get('theta', 0)).max() < 1e-12
    batch.close()
    solo.close()


# ---- residency guard and run-time fallback of the fused iteration kernel -------------------------------------
def _headline_run(iters=24):
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
    prob = FlatProblem(Q, W, X, y)
    keys = [KEY + 5 * c for c in range(4)]
    eng = Engine(prob, keys)
    for c in range(4):
        eng.set_start(c, **_random_start(prob, 60 + c))
    st0 = eng.stats()
    rec = eng.run(iters, 0)
    rec2 = eng.run(7, 2)                                  # the engine keeps working after a fallback
    state = [(eng.get('eta', c), eng.get('z', c), eng.get('xz', c)) for c in range(4)]
    st1 = eng.stats()
    eng.close()
    return rec + rec2, state, st0, st1


def test_undersized_cu_partition_is_refused_or_demoted_at_creation(monkeypatch):
    """OCC_CU_SPLIT that is not a whole-shader-engine partition is a ValueError; a valid but under-sized one (32 CUs
    for 80 workgroups of 512 threads) is caught by the arithmetic / the residency probe at creation: the engine
    takes the launch-per-step path on plain streams and returns the same bits as the default configuration."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    ref = _headline_run()
    assert ref[2]['persistent_solve'] == 2
    monkeypatch.setenv('OCC_CU_SPLIT', '26')
    Q, W, X, y, *_ = make_lattice_problem(12, 12, visits=3, p=2, q=2, random_state=1)
    with pytest.raises(ValueError, match='OCC_CU_SPLIT'):
        Engine(FlatProblem(Q, W, X, y), [KEY])
    monkeypatch.setenv('OCC_CU_SPLIT', '32')
    alt = _headline_run()
    assert alt[2]['persistent_solve'] == 0 and alt[2]['main_stream_cus'] == 0 and alt[3]['fused_fallbacks'] == 0
    for u, v in zip(ref[0], alt[0]):
        assert np.array_equal(u, v)
    for su, sv in zip(ref[1], alt[1]):
        for u, v in zip(su, sv):
            assert np.array_equal(u, v)


def test_barrier_timeout_falls_back_to_launch_per_step_with_the_same_bits(monkeypatch):
    """The time-out path for real: with the residency probe switched off (debug knob) the one-XCD-per-chain form is
    launched on a 32-CU partition -- 4 CUs per XCD for a chain's 20 workgroups of one per CU, so 16 of them are never
    resident with the first 4 and the first barrier gives up.  The call is re-run from its start state on the
    launch-per-step path: same records, same state, bit for bit; the engine stays usable (second run)."""
    ref = _headline_run()
    monkeypatch.setenv('OCC_CU_SPLIT', '32')
    monkeypatch.setenv('OCC_DEBUG_SKIP_RESIDENCY_PROBE', '1')
    monkeypatch.setenv('OCC_QUIET', '1')
    alt = _headline_run()
    assert alt[2]['persistent_solve'] == 2                       # what creation believed
    # the second call asks the residency probe before it comes back: the partition is as under-sized as before, so the
    # engine stays on the launch-per-step path (one fallback, no return)
    assert alt[3]['persistent_solve'] == 0 and alt[3]['fused_fallbacks'] == 1 and alt[3]['repromotions'] == 0 and alt[3]['demoted'] == 1
    for u, v in zip(ref[0], alt[0]):
        assert np.array_equal(u, v)
    for su, sv in zip(ref[1], alt[1]):
        for u, v in zip(su, sv):
            assert np.array_equal(u, v)


@pytest.mark.parametrize('env, main_cus', [({'OCC_NO_SCALAR_WAVE': '1'}, 160), ({'OCC_NO_XCD_SHARES': '1'}, 160), ({'OCC_CU_SPLIT': '192'}, 192),
                                           ({'OCC_CU_SPLIT': '0'}, 0)])
def test_scalar_wave_form_and_its_cu_partition_do_not_change_a_bit(monkeypatch, env, main_cus):
    """The headline workload takes the one-XCD form with a scalar wave beside seven site waves (448 sites per workgroup, 24
    CUs on the XCDs that host a chain and 16 on the others, tiles of the device-filling kernels in proportion).  The same
    run with eight site waves per workgroup, with even tile shares, on an even 192 + 64 partition and without any
    partition: same records, same state, bit for bit -- the summation order does not depend on the workgroup geometry,
    the slot a workgroup claims does not matter, and where a tile runs never changes what it computes."""
    ref = _headline_run()
    assert ref[2]['persistent_solve'] == 2 and ref[2]['main_stream_cus'] == 160
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    alt = _headline_run()
    assert alt[2]['persistent_solve'] == 2 and alt[2]['main_stream_cus'] == main_cus and alt[3]['fused_fallbacks'] == 0
    for u, v in zip(ref[0], alt[0]):
        assert np.array_equal(u, v)
    for su, sv in zip(ref[1], alt[1]):
        for u, v in zip(su, sv):
            assert np.array_equal(u, v)


def test_broken_stream_handover_falls_back_with_the_same_bits(monkeypatch):
    """The two streams hand over through device counters, which presumes they run beside each other.  Debug knob: the side
    stream's gate never announces its noise, as if the streams were served one after the other.  The first wait to give up
    ends every other wait of the enqueued batch (one time-out, not three per iteration), and the call is re-run from its
    start state without hand-overs: same records, same state, bit for bit -- on the fused ICAR path and on the reduced-rank
    model (whose two-stream schedule has no fused kernel to fall back from, only the hand-overs)."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    monkeypatch.setenv('OCC_QUIET', '1')
    ref = _headline_run(iters=10)

    def rsr_run():
        Q, W, X, y, *_ = make_lattice_problem(24, 30, visits=3, p=2, q=2, random_state=4)
        prob = FlatProblem(Q, W, X, y)
        m = prob.enable_rsr(q=40)['dim']
        rng = np.random.default_rng(8)
        eng = Engine(prob, [KEY, KEY + 1])
        for c in range(2):
            eng.set_start(c, alpha=rng.standard_normal(2), beta=rng.standard_normal(2), tau=1.0 + c, eta=rng.standard_normal(m))
        rec = eng.run(8, 0) + eng.run(5, 1)
        out = rec, [eng.get('theta', c) for c in range(2)], eng.stats()
        eng.close()
        return out

    rsr_ref = rsr_run()
    monkeypatch.setenv('OCC_DEBUG_BREAK_HANDOVER', '1')
    alt = _headline_run(iters=10)
    # ... and the engine does not STAY there: the next call finds the device as creation found it -- the
    # stream probe and the residency probe pass, the counters restart from zero (the knob's word with them) -- and runs
    # fused with device-side hand-overs again, still bit for bit
    assert alt[3]['fused_fallbacks'] == 1 and alt[3]['repromotions'] == 1 and alt[3]['persistent_solve'] == 2
    assert alt[3]['handover_mode'] == 2 and alt[3]['demoted'] == 0
    for u, v in zip(ref[0], alt[0]):
        assert np.array_equal(u, v)
    for su, sv in zip(ref[1], alt[1]):
        for u, v in zip(su, sv):
            assert np.array_equal(u, v)
    rsr_alt = rsr_run()
    assert rsr_alt[2]['fused_fallbacks'] == 1 and rsr_alt[2]['repromotions'] == 1 and rsr_alt[2]['handover_mode'] == 2
    for u, v in zip(rsr_ref[0], rsr_alt[0]):
        assert np.array_equal(u, v)
    for u, v in zip(rsr_ref[1], rsr_alt[1]):
        assert np.array_equal(u, v)


# ---- the reference's own form of the prior draw, and precisions the edge form cannot represent ---------------------
def _general_singular_precision(n, seed):
    """A symmetric positive semi-definite singular Q that is NOT an ICAR precision: Q = M'M with every row of M summing
    to zero (so Q 1 = 0) but rows like (1, 1, -2), which put POSITIVE entries off the diagonal."""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    r = 0
    for i in range(n - 1):            # a path keeps the graph connected: rank n - 1
        rows += [r, r]; cols += [i, i + 1]; vals += [1.0, -1.0]; r += 1
    for _ in range(n):
        i, j, k = rng.choice(n, size=3, replace=False)
        rows += [r, r, r]; cols += [i, j, k]; vals += [1.0, 1.0, -2.0]; r += 1
    M = sparse.csr_matrix((vals, (rows, cols)), shape=(r, n))
    Q = (M.T @ M).tocsr()
    Q.sum_duplicates()
    return Q


@pytest.mark.parametrize('which', ['icar_forced_dense', 'general_singular'])
def test_reference_form_prior_draw_lockstep(oracle, which, solve_mode):
    """prior_draw='dense': the N(0, Q) term of the eta conditional as the reference forms it (logit.py:64-67, 77) --
    E from the dense eigh of Q on the host, u = E eps2 by one pass over E on the device for all chains -- in lock step
    with the oracle's dense mode (same normals, Philox stream 10).  Once on an ICAR lattice (where the edge form is the
    default), once on a singular Q with positive off-diagonals, which only this form can sample ('auto' selects it)."""
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(13, 17, visits=4, p=2, q=2, random_state=6)
    if which == 'icar_forced_dense':
        prob = FlatProblem(Q, W, X, y, prior_draw='dense')
    else:
        Qg = _general_singular_precision(Q.shape[0], 3)
        assert (Qg - sparse.diags(Qg.diagonal())).max() > 0
        with pytest.raises(ValueError, match='non-positive off-diagonal'):
            FlatProblem(Qg, W, X, y, prior_draw='edge')
        prob = FlatProblem(Qg, W, X, y)            # 'auto': falls back to the reference's form
    E = prob.prior_factor
    assert E is not None and E.shape == (prob.n, prob.n - 1) and np.abs(E @ E.T - prob.Q.toarray()).max() < 1e-9
    starts = [_random_start(prob, 70 + c) for c in range(5)]     # five chains: two passes over E (four chains per pass)
    worst, stats = _lockstep_chains(oracle, prob, starts, [KEY + 17 * c for c in range(5)], 4)
    assert stats['n_chains'] == 5


def test_reference_form_prior_draw_free_running_equals_edge_form_in_law(oracle):
    """Same posterior whichever way the prior term is drawn (E E' = Q = B'B): 4 device chains with prior_draw='dense'
    against the REFERENCE's chains (which use exactly this form)."""
    from occuspytial_amd import LogitICARGibbs
    from .test_reference_chains import compare_with_reference, problem_of
    case = 'refchain_queen150_tauprior'
    Q, W, X, y, hp, ch = problem_of(case)
    s = LogitICARGibbs(Q, W, X, y, hparams=hp, random_state=77, prior_draw='dense')
    assert s._problem.prior_factor is not None
    post = s.sample(int(ch['size']), burnin=int(ch['burnin']), chains=4, progressbar=False)
    compare_with_reference(case, post['alpha'], post['beta'], post['tau'])


@pytest.mark.parametrize('p, q', [(12, 12), (20, 3), (2, 11), (32, 32)])
def test_more_than_eight_covariates_take_the_generic_kernels(oracle, p, q):
    """The reference's conditionals take any number of covariates (distributions.pyx:42-110 factors an n x n system for
    any n).  Up to 8 of each kind the engine keeps the p x p / q x q accumulators in registers; beyond that (to 32) the
    generic instantiations run -- run-time p and q, the terms of the systems reduced one at a time, the Cholesky factor
    in LDS -- on the launch-per-step path.  Three lock-step iterations against the oracle, two chains; then the
    per-conditional entry points on the same sizes."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(17, 19, visits=6, p=p, q=q, random_state=p + q)
    prob = FlatProblem(Q, W, X, y)
    rng = np.random.default_rng(p)
    starts = [dict(alpha=0.3 * rng.standard_normal(q), beta=0.3 * rng.standard_normal(p), tau=1.1,
                   eta=(lambda e: e - e.mean())(rng.standard_normal(prob.n))) for _ in range(2)]
    worst, stats = _lockstep_chains(oracle, prob, starts, [KEY + 1, KEY + 2], 3)
    assert stats['persistent_solve'] == 0
    eng = Engine(prob, [KEY + 1, KEY + 2])
    for c, st in enumerate(starts):
        eng.set_start(c, **st)
    rec = eng.run(12, 2)                                # graph replay through the generic kernels
    assert rec[0].shape == (2, 10, q) and rec[1].shape == (2, 10, p) and np.all(np.isfinite(rec[1])) and np.all(rec[2] > 0)
    # injected-variate entry points at these sizes: beta and alpha draws against numpy on the same sums
    st = starts[0]
    eng.set_start(0, **st)
    om = rng.uniform(0.05, 0.3, prob.n)
    eps = rng.standard_normal(p)
    beta = eng.cond_beta(om, eps)
    z = eng.get('z')
    A = (prob.X.T * om) @ prob.X + prob.b_prec
    r = prob.X.T @ ((z - 0.5) - om * st['eta']) + prob.b_prec @ prob.b_mu
    U = np.linalg.cholesky(A).T
    want = np.linalg.solve(A, r) + np.linalg.solve(U, eps)          # distributions.pyx:95-105
    assert np.allclose(beta, want, rtol=1e-9, atol=1e-11)
    eng.close()
