"""Streams are a process-wide resource: the condition behind round 2's hand-over stall, reproduced
deterministically, and the device's error exits.

An MI355X has 24 hardware queue slots per device (KFD topology ``num_cp_queues``); every CU-masked stream holds one for
itself.  Round 2 gave every engine its own two masked streams: a process with a dozen live samplers oversubscribed the
slots, the scheduler time-sliced the queues (quanta of ~10 ms, ``tools/queue_probe.hip``) and every device-side hand-over
of a NEW engine ran into its time-out.  Now the pairs are pooled per (process, device, CU partition) and capped; these
tests hold the condition (many live engines, then a fresh engine's run) and assert what the fix guarantees: no
fallback, device-side hand-overs, the same bits as in a clean process state."""
import threading

import numpy as np
import pytest

from .conftest import load_golden
from .test_api_cpu import _inputs
from .test_gpu_parity import KEY, _random_start

pytestmark = pytest.mark.gpu


def _small_problem(seed=1, rows=12, cols=13):
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(rows, cols, visits=3, p=2, q=2, random_state=seed)
    return FlatProblem(Q, W, X, y)


def _rsr_run():
    """A fresh reduced-rank engine (two streams, hand-overs on the device, no fused kernel to fall back from): the engine
    round 2's stall hit."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(24, 30, visits=3, p=2, q=2, random_state=4)
    prob = FlatProblem(Q, W, X, y)
    m = prob.enable_rsr(q=40)['dim']
    rng = np.random.default_rng(8)
    eng = Engine(prob, [KEY, KEY + 1, KEY + 2])
    for c in range(3):
        eng.set_start(c, alpha=rng.standard_normal(2), beta=rng.standard_normal(2), tau=1.0 + c, eta=rng.standard_normal(m))
    rec = eng.run(12, 2)
    out = rec, [eng.get('theta', c) for c in range(3)], eng.stats()
    eng.close()
    return out


def _icar_run(prob, chains=2, iters=16):
    from occuspytial_amd._engine import Engine
    eng = Engine(prob, [KEY + 7 * c for c in range(chains)])
    for c in range(chains):
        eng.set_start(c, **_random_start(prob, 40 + c))
    rec = eng.run(iters, 0)
    out = rec, [(eng.get('eta', c), eng.get('z', c)) for c in range(chains)], eng.stats()
    eng.close()
    return out


def _flat(x):
    if isinstance(x, (tuple, list)):
        return [a for y in x for a in _flat(y)]
    return [np.asarray(x)]


def _same(u, v):
    """records and end state of two runs: equal bit for bit"""
    fu, fv = _flat(u[0]) + _flat(u[1]), _flat(v[0]) + _flat(v[1])
    assert len(fu) == len(fv)
    for a, b in zip(fu, fv):
        assert np.array_equal(a, b)


def test_many_live_engines_then_a_fresh_engine_hands_over_on_the_device():
    """The condition of round 2's stall, held still: 16 engines alive in this process (round 2: 32 CU-masked streams, 32
    hardware queues wanted of the device's 24), every one of them used, then a FRESH reduced-rank engine's run -- and a fresh
    fused ICAR engine's.  With pooled pairs the 16 engines share ONE masked pair; the new engines hand over through the
    device counters, nothing falls back, and the draws equal those of a clean process state bit for bit."""
    from occuspytial_amd._engine import Engine
    prob = _small_problem()
    big = _small_problem(seed=3, rows=40, cols=45)
    ref_rsr, ref_icar = _rsr_run(), _icar_run(big)
    assert ref_rsr[2]['fused_fallbacks'] == 0 and ref_rsr[2]['handover_mode'] == 2
    assert ref_icar[2]['fused_fallbacks'] == 0 and ref_icar[2]['handover_mode'] == 2 and ref_icar[2]['persistent_solve'] >= 1
    live = []
    try:
        for i in range(16):
            e = Engine(prob, [KEY + i])
            e.set_start(0, **_random_start(prob, i))
            e.run(6, 0)
            live.append(e)
        st = live[-1].stats()
        assert st['stream_pairs_masked'] == 1 and st['stream_pairs_plain'] == 0, st   # one pooled pair for all sixteen
        assert all(e.stats()['fused_fallbacks'] == 0 and e.stats()['handover_mode'] == 2 for e in live)
        alt_rsr, alt_icar = _rsr_run(), _icar_run(big)
        for alt in (alt_rsr, alt_icar):
            assert alt[2]['fused_fallbacks'] == 0 and alt[2]['handover_mode'] == 2 and alt[2]['demoted'] == 0, alt[2]
            assert alt[2]['stream_pairs_masked'] <= 4
        _same(ref_rsr, alt_rsr)
        _same(ref_icar, alt_icar)
        for e in live:                                    # the old engines still work beside the new ones' pairs
            e.run(4, 0)
            assert e.stats()['fused_fallbacks'] == 0
    finally:
        for e in live:
            e.close()


def test_partitions_beyond_the_cap_take_the_unmasked_pair_by_counting(monkeypatch):
    """At most MAX_MASKED_PAIRS CU partitions are alive per device (each costs two hardware queues).  With the cap at one
    (test knob) a second, different partition is not created: that engine takes the unmasked pair and hands over through
    events -- decided by counting, at creation, not by a probe -- and returns the same bits as with its partition."""
    from occuspytial_amd._engine import Engine
    big = _small_problem(seed=3, rows=40, cols=45)
    ref = _icar_run(big)
    assert ref[2]['main_stream_cus'] > 0 and ref[2]['handover_mode'] == 2
    monkeypatch.setenv('OCC_MAX_MASKED_PAIRS', '1')
    from occuspytial_amd._problem import FlatProblem
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(20, 20, visits=3, p=2, q=2, random_state=2)
    rsr_prob = FlatProblem(Q, W, X, y)
    rsr_prob.enable_rsr(q=20)
    holder = Engine(rsr_prob, [KEY])                     # holds the one masked pair (the reduced-rank model's 192 + 64 partition)
    try:
        assert holder.stats()['main_stream_cus'] == 192 and holder.stats()['stream_pairs_masked'] == 1
        alt = _icar_run(big)
        assert alt[2]['main_stream_cus'] == 0 and alt[2]['handover_mode'] == 1 and alt[2]['fused_fallbacks'] == 0
        assert alt[2]['stream_pairs_masked'] == 1 and alt[2]['stream_pairs_plain'] == 1
        _same(ref, alt)
    finally:
        holder.close()


def test_two_host_threads_on_one_device_run_their_calls_one_after_the_other():
    """Engines of one device share its pooled streams, so the library runs their calls one after the other (a per-device
    lock): two fused engines driven from two host threads at once neither take each other's CUs (round 2: barrier
    time-outs and a fallback) nor each other's stream captures -- no fallback, and each returns the bits of a solo run."""
    from occuspytial_amd._engine import Engine
    big = _small_problem(seed=3, rows=40, cols=45)
    ref = _icar_run(big, iters=40)
    out = [None, None]

    def work(k):
        out[k] = _icar_run(big, iters=40)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for k in range(2):
        assert out[k][2]['fused_fallbacks'] == 0 and out[k][2]['handover_mode'] == 2
        _same(ref, out[k])


# ---- the device's error exits (mapped, never raised on the device) -------------------------------------------
def test_cholesky_failure_on_the_device_raises_the_references_error():
    """A prior precision that makes a conditional's system indefinite: the device's factorisation reports it
    (distributions.pyx:21, 107-108: RuntimeError('Cholesky factorization/solver failed!')) -- beta's system in k_z_ob and
    alpha's in k_noise, through sample() of the drop-in class; the sampler's next call fails as cleanly (no hang, no stale
    state) and another sampler on the same device works."""
    from occuspytial_amd import LogitICARGibbs
    data = _inputs(load_golden('ref_queen150_ragged'))[:4]
    good = LogitICARGibbs(*data, random_state=10).sample(8, chains=2, progressbar=False)
    # (start values given: the default start draws alpha and beta from N(mu, 100 prec) on the host, base.py:199-212, and
    # numpy's own Cholesky would refuse these matrices before the device sees them -- in the reference too)
    start = {'alpha': np.zeros(2), 'beta': np.zeros(3), 'tau': 1.0, 'eta': np.zeros(150)}
    for bad in ({'b_prec': np.diag([1.0, -1e6, 1.0])}, {'a_prec': np.diag([-1e7, 1.0])}):
        s = LogitICARGibbs(*data, hparams=bad, random_state=10)
        for _ in range(2):
            with pytest.raises(RuntimeError, match='Cholesky factorization/solver failed!'):
                s.sample(8, chains=2, start=start, progressbar=False)
        assert s._engine.stats()['fused_fallbacks'] == 0
    again = LogitICARGibbs(*data, random_state=10).sample(8, chains=2, progressbar=False)
    for k in ('alpha', 'beta', 'tau'):
        assert np.array_equal(good[k], again[k])


@pytest.mark.parametrize('path', ['fused', 'launch_per_step'])
def test_minres_iteration_limit_on_the_device_raises_the_references_error(monkeypatch, path):
    """``RuntimeError('MINRES solver did not converge!')`` (logit.py:91-92, scipy's info > 0 on the iteration limit): the
    limit is lowered to 3 on a live engine (debug state name), the solve stops with istop = 6 INSIDE the kernel -- k_iter's
    in-kernel exit on the fused path, k_minres / k_beta_partial on the launch-per-step path -- and the call raises; with
    the limit restored and the chains re-started the same engine returns the bits of an undisturbed one."""
    from occuspytial_amd._engine import Engine
    if path == 'launch_per_step':
        monkeypatch.setenv('OCC_NO_PERSISTENT', '1')
    prob = _small_problem(seed=3, rows=40, cols=45)

    def start(e):
        for c in range(2):
            e.set_start(c, **_random_start(prob, 40 + c))

    ref = Engine(prob, [KEY, KEY + 7])
    start(ref)
    want = ref.run(12, 0)
    assert int(ref.get('minres_itn')) > 3
    ref.close()
    eng = Engine(prob, [KEY, KEY + 7])
    assert (eng.stats()['persistent_solve'] > 0) == (path == 'fused')
    start(eng)
    eng.set('debug_maxiter', 3.0)
    with pytest.raises(RuntimeError, match='MINRES solver did not converge!'):
        eng.run(12, 0)
    start(eng)
    with pytest.raises(RuntimeError, match='MINRES solver did not converge!'):
        eng.step()
    eng.set('debug_maxiter', 0.0)                        # scipy's default 5 * (2n) again
    start(eng)
    got = eng.run(12, 0)
    assert eng.stats()['fused_fallbacks'] == 0
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    eng.close()


# ---- a call that cannot advance ends, and says what it saw (round 4: DESIGN 7) ------------------------------------------
@pytest.mark.parametrize('path', ['fused', 'launch_per_step'])
def test_a_closed_window_ends_the_call_with_the_state_it_saw(monkeypatch, path):
    """occ_run's host loop enqueues sequences until every chain has done its iterations; a chain whose window of iterations
    is not open on the device idles through every sequence and reports no error -- round 3 recorded a call that never
    returned.  The debug state name ``debug_close_window`` sends the NEXT call's window to the device with zero iterations
    (both forms of open_window: the edit by k_snapshot on the fused path, the upload on the launch-per-step path): the call
    must end at once with OCC_E_HIP and the chains' control words in the text, and the engine must go on as if the call had
    not been made -- the same bits as an undisturbed engine."""
    import time

    from occuspytial_amd._engine import Engine
    from occuspytial_amd._lib import EngineUnavailable
    if path == 'launch_per_step':
        monkeypatch.setenv('OCC_NO_PERSISTENT', '1')
    prob = _small_problem(seed=3, rows=40, cols=45)

    def fresh():
        e = Engine(prob, [KEY, KEY + 7])
        for c in range(2):
            e.set_start(c, **_random_start(prob, 40 + c))
        return e

    ref = fresh()
    want = ref.run(9, 0), ref.run(20, 3)
    want_eta = [ref.get('eta', c) for c in range(2)]
    ref.close()
    eng = fresh()
    assert (eng.stats()['persistent_solve'] > 0) == (path == 'fused')
    got0 = eng.run(9, 0)
    eng.set('debug_close_window', 1.0)
    t0 = time.perf_counter()
    with pytest.raises(EngineUnavailable) as ei:
        eng.run(20, 3)
    assert time.perf_counter() - t0 < 1.0
    text = str(ei.value)
    assert 'no progress' in text and 'it_base, it_stop' in text and 'parity' in text and '{9, 0 |' in text, text
    got1 = eng.run(20, 3)                                # the knob is consumed; the chains are where the first call left them
    st = eng.stats()
    assert st['fused_fallbacks'] == 0 and st['iterations'] == 29
    for a, b in zip(want[0] + want[1], got0 + got1):
        assert np.array_equal(a, b)
    for c in range(2):
        assert np.array_equal(want_eta[c], eng.get('eta', c))
    eng.close()


def test_a_fifth_cu_partition_evicts_an_idle_pair_and_nothing_changes(monkeypatch):
    """The pool keeps a pair whose last engine closed, idle, for the next taker of its CU partition, and destroys idle
    pairs only to make room under the cap of four masked pairs per device (acquire_pair's eviction branch): the one place
    where CU-masked streams are still destroyed and created back to back inside a running process -- the sequence round 3's
    stop inside occ_create followed.  Six distinct partitions (the default's and five more) opened, used and closed in turn: from the fifth on each evicts; then a
    default engine returns the bits it returned before any of it."""
    from occuspytial_amd._engine import Engine
    prob = _small_problem(seed=3, rows=40, cols=45)
    ref = _icar_run(prob)
    base = ref[2]['stream_pairs_evicted']
    seen_idle = []
    for split in ('64', '96', '128', '192', '224'):
        monkeypatch.setenv('OCC_CU_SPLIT', split)
        eng = Engine(prob, [KEY, KEY + 7])
        for c in range(2):
            eng.set_start(c, **_random_start(prob, 40 + c))
        rec = eng.run(6, 0)
        st = eng.stats()
        assert st['main_stream_cus'] == int(split) and st['handover_mode'] == 2 and st['fused_fallbacks'] == 0, st
        assert st['stream_pairs_masked'] == 1                     # the only pair an engine HOLDS
        assert st['stream_pairs_masked'] + st['stream_pairs_idle'] <= 4 + 1, st   # four masked pairs at most, and the device's idle plain pair
        seen_idle.append(st['stream_pairs_idle'])
        for a, b in zip(rec, ref[0]):                             # (the CU partition never changes a bit)
            assert np.array_equal(a[:, :6], b[:, :6])
        eng.close()
    monkeypatch.delenv('OCC_CU_SPLIT')
    alt = _icar_run(prob)
    assert alt[2]['stream_pairs_evicted'] >= base + 2, (base, alt[2], seen_idle)   # six distinct partitions + the default's: five more partitions + the default's again under a cap of four: at least two evictions
    assert alt[2]['fused_fallbacks'] == 0 and alt[2]['handover_mode'] == 2
    _same(ref, alt)


def test_a_host_wait_that_runs_out_names_its_call_site(monkeypatch):
    """No call of the library blocks without a limit on its own streams: every host wait polls the stream against a
    deadline (``OCC_HOST_WAIT_S``).  With the deadline at a millisecond a call of two thousand iterations runs into it: the
    call returns OCC_E_HIP with the waiting function, its source line and the engine's host-side state in the text; the engine
    is then closed without waiting for, freeing or handing on anything its streams may still hold; and the next engine -- on
    a fresh pair of streams -- returns the bits of a clean process state."""
    import time

    from occuspytial_amd._engine import Engine
    from occuspytial_amd._lib import EngineUnavailable
    prob = _small_problem(seed=3, rows=40, cols=45)
    ref = _icar_run(prob)
    eng = Engine(prob, [KEY, KEY + 7])
    for c in range(2):
        eng.set_start(c, **_random_start(prob, 40 + c))
    eng.run(4, 0)                                     # (graphs captured, everything warm)
    monkeypatch.setenv('OCC_HOST_WAIT_S', '0.001')
    t0 = time.perf_counter()
    with pytest.raises(EngineUnavailable) as ei:
        eng.run(4000, 3999)
    text = str(ei.value)
    assert 'host wait `' in text and 'occ_gibbs.hip:' in text and 'did not drain within' in text and 'parity' in text, text
    eng.close()                                       # returns at once: nothing is waited for or freed
    assert time.perf_counter() - t0 < 5.0
    monkeypatch.delenv('OCC_HOST_WAIT_S')
    time.sleep(1.0)                                   # (the abandoned batch drains on its own: 4 000 iterations of 40 us)
    alt = _icar_run(prob)
    assert alt[2]['fused_fallbacks'] == 0 and alt[2]['handover_mode'] == 2
    _same(ref, alt)
