"""Chains sharded over GPUs, on the one GPU a test box has: groups built on device 0 (twice), the RCCL
communicator with a single rank.  What these tests can and cannot show: the group / distributed creation paths,
the device-to-device hand-over of the laid-out arrays, the thread fan-out and the chain bookkeeping run for real;
a broadcast between DIFFERENT devices needs a multi-GPU node (the round-end scaling run)."""
import numpy as np
import pytest

from .conftest import load_golden
from .test_api_cpu import _inputs
from .test_gpu_parity import KEY, _problem_from_golden, _random_start

pytestmark = pytest.mark.gpu


def _run(eng, prob, n_chains, iters=30, burnin=4):
    for c in range(n_chains):
        eng.set_start(c, **_random_start(prob, 100 + c))
    rec = eng.run(iters, burnin)
    state = [(eng.get('eta', c), eng.get('z', c)) for c in range(n_chains)]
    return rec, state


def _same(u, v):
    for a, b in zip(u[0], v[0]):
        assert np.array_equal(a, b)
    for (e1, z1), (e2, z2) in zip(u[1], v[1]):
        assert np.array_equal(e1, e2) and np.array_equal(z1, z2)


def test_group_of_one_device_equals_plain_engine():
    from occuspytial_amd._engine import Engine, EngineGroup
    prob, _ = _problem_from_golden('ref_queen400_v3')
    keys = [KEY + c for c in range(3)]
    plain = Engine(prob, keys)
    ref = _run(plain, prob, 3)
    plain.close()
    grp = EngineGroup(prob, keys, [0])
    assert grp.transport == 'single device' and len(grp.engines) == 1
    _same(ref, _run(grp, prob, 3))
    grp.close()


def test_group_broadcast_through_rccl_with_one_device(monkeypatch):
    """OCC_GROUP_TRANSPORT=rccl: the in-process group path through librccl for real -- dlopen, ncclCommInitAll, one grouped
    ncclBroadcast per fixed array on the sampler's own (CU-masked) stream, ncclCommDestroy -- with the one device this box
    has (an in-place broadcast among one rank); the arrays must come out untouched: same chains as a plain engine."""
    from occuspytial_amd._engine import Engine, EngineGroup
    monkeypatch.setenv('OCC_GROUP_TRANSPORT', 'rccl')
    prob, _ = _problem_from_golden('ref_queen400_v3')
    keys = [KEY + c for c in range(3)]
    plain = Engine(prob, keys)
    ref = _run(plain, prob, 3)
    plain.close()
    grp = EngineGroup(prob, keys, [0])
    assert grp.transport.startswith('rccl broadcast (ncclCommInitAll), 1 devices'), grp.transport
    _same(ref, _run(grp, prob, 3))
    grp.close()


@pytest.mark.parametrize('transport', ['peer', 'default'])
def test_chains_sharded_over_two_samplers_equal_one_batch(monkeypatch, transport):
    """Two samplers of one group (both on device 0: the box has one GPU): the second receives the laid-out arrays
    device to device, chain c runs on sampler c % 2 from its own host thread -- same chains as one batch, bit for bit.
    `default` asks RCCL first (ncclCommInitAll refuses a device listed twice, so the hand-over falls to hipMemcpyPeer
    and says why); `peer` forces hipMemcpyPeer."""
    from occuspytial_amd._engine import Engine, EngineGroup
    if transport == 'peer':
        monkeypatch.setenv('OCC_GROUP_TRANSPORT', 'peer')
    prob, _ = _problem_from_golden('ref_graph300_weighted')
    keys = [KEY + 3 * c for c in range(5)]
    plain = Engine(prob, keys)
    ref = _run(plain, prob, 5)
    plain.close()
    grp = EngineGroup(prob, keys, [0, 0])
    assert len(grp.engines) == 2 and [e.n_chains for e in grp.engines] == [3, 2]
    assert grp.transport.startswith('hipMemcpyPeer') or grp.transport.startswith('rccl'), grp.transport
    _same(ref, _run(grp, prob, 5))
    ck = grp.checkpoint()
    assert ck['eta'].shape == (5, prob.n) and list(ck['iter']) == [30] * 5
    grp.close()


def test_sampler_class_fans_chains_out_over_devices(monkeypatch):
    """``LogitICARGibbs(..., devices=[0, 0]).sample(chains=3)``: the reference's call, the chains sharded over two
    samplers in this process -- same draws as ``device=0`` (chain k owns generator k either way), checkpoint / resume
    across the group included."""
    from occuspytial_amd import LogitICARGibbs
    monkeypatch.setenv('OCC_GROUP_TRANSPORT', 'peer')
    data = _inputs(load_golden('ref_queen150_ragged'))[:4]
    one = LogitICARGibbs(*data, random_state=10).sample(40, burnin=5, chains=3, progressbar=False)
    s2 = LogitICARGibbs(*data, random_state=10, devices=[0, 0])
    two = s2.sample(40, burnin=5, chains=3, progressbar=False)
    for k in ('alpha', 'beta', 'tau'):
        assert np.array_equal(one[k], two[k])
    assert len(s2.__dict__['_engine'].engines) == 2 and s2.state.eta.shape == (150,)
    whole = LogitICARGibbs(*data, random_state=21).sample(30, chains=3, progressbar=False)
    first = LogitICARGibbs(*data, random_state=21, devices=[0, 0])
    first.sample(12, chains=3, progressbar=False)
    tail = LogitICARGibbs(*data, random_state=5, devices=[0, 0]).resume(first.checkpoint(), 18, progressbar=False)
    for k in ('alpha', 'beta', 'tau'):
        assert np.array_equal(tail[k], whole[k][:, 12:])


def test_rccl_communicator_and_distributed_creation_with_one_rank(tmp_path):
    """The one-process-per-GPU path with world size 1: ncclGetUniqueId / ncclCommInitRank through the C ABI (librccl
    opened with dlopen), the host-side collectives, and ``occ_create_distributed`` -- layout header, deferred uploads,
    ncclBroadcast of every fixed array (a no-op copy with one rank) -- giving the engine ``occ_create`` gives."""
    from occuspytial_amd._engine import Engine
    from occuspytial_amd.distributed import FileComm, RcclComm, distributed_engine, run_sharded
    side = FileComm(0, 1, str(tmp_path / 'rdv'))
    comm = RcclComm(side, device=0)
    comm.barrier()
    assert comm.allreduce_max(3.5) == 3.5
    assert comm.allreduce_max(np.array([1.0, -2.0])).tolist() == [1.0, -2.0]
    obj = {'a': np.arange(5), 'b': 'text'}
    got = comm.bcast_obj(obj, 0)
    assert got['b'] == 'text' and np.array_equal(got['a'], obj['a'])
    assert comm.allgather_obj(7) == [7]
    prob, _ = _problem_from_golden('ref_queen150_ragged')
    keys = [KEY, KEY + 9]
    plain = Engine(prob, keys)
    ref = _run(plain, prob, 2)
    plain.close()
    eng, mine = distributed_engine(prob, comm, keys)
    assert mine is prob and eng.transport.startswith('rccl broadcast (ncclCommInitRank)')
    _same(ref, _run(eng, prob, 2))
    eng.close()
    A, B, T = run_sharded(prob, 2, size=8, burnin=2, random_state=3, device=0, comm=comm)
    assert A.shape == (2, 6, prob.q) and np.all(T > 0)
    comm.close()


def test_a_broadcast_that_arrives_damaged_is_caught_and_named(monkeypatch):
    """After the hand-over every sampler of a group checksums its copies of the fixed arrays ON ITS DEVICE and the sums are
    compared with the root's (the first real multi-GPU run must be able to tell a wrong broadcast from a right one).
    Debug knob: the peer's checksum of array 2 is flipped, as if those bytes had arrived damaged -- creation fails and
    names the array."""
    from occuspytial_amd._engine import EngineGroup
    from occuspytial_amd._lib import EngineUnavailable
    monkeypatch.setenv('OCC_GROUP_TRANSPORT', 'peer')
    prob, _ = _problem_from_golden('ref_queen150_ragged')
    keys = [KEY, KEY + 1, KEY + 2]
    monkeypatch.setenv('OCC_DEBUG_CORRUPT_BROADCAST', '2')
    with pytest.raises(EngineUnavailable, match='did not arrive intact: array `sell_val`'):
        EngineGroup(prob, keys, [0, 0])
    monkeypatch.delenv('OCC_DEBUG_CORRUPT_BROADCAST')
    grp = EngineGroup(prob, keys, [0, 0])               # ... and the undamaged hand-over passes the same check
    _run(grp, prob, 3)
    grp.close()


def test_without_librccl_the_group_copies_peer_to_peer_and_the_ranks_meet_through_files(tmp_path):
    """A machine without librccl (that branch crashed -- dlerror() called twice -- and no test reached it;
    OCC_RCCL_LIB=none skips the default names): in a fresh process (the library handle is looked up once per process) the
    group's hand-over ends on hipMemcpyPeer with the reason, occ_comm_unique_id fails cleanly, and init_comm ends on the
    file rendezvous."""
    import os
    import subprocess
    import sys
    from .conftest import ROOT
    code = r'''
import numpy as np
from tests.test_gpu_parity import KEY, _problem_from_golden, _random_start
from occuspytial_amd._engine import EngineGroup
from occuspytial_amd.distributed import init_comm
prob, _ = _problem_from_golden('ref_queen150_ragged')
grp = EngineGroup(prob, [KEY, KEY + 1], [0, 0])
assert grp.transport.startswith('hipMemcpyPeer (librccl could not be opened: OCC_RCCL_LIB=none'), grp.transport
for c in range(2):
    grp.set_start(c, **_random_start(prob, c))
a, b, t = grp.run(6, 0)
assert np.all(np.isfinite(a)) and np.all(t > 0)
grp.close()
comm, note = init_comm(device=0)
assert note.startswith('file rendezvous (RCCL unusable'), note
comm.close()
print('ok')
'''
    env = dict(os.environ, OCC_RCCL_LIB='none', PYTHONPATH=ROOT, RANK='0', WORLD_SIZE='1', MASTER_PORT='29999',
               TORCHELASTIC_RUN_ID='norccl_%d' % os.getpid())
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout + r.stderr


def test_rccl_broadcast_between_two_devices_equals_one_batch():
    """The first thing a multi-GPU node runs (skipped on a one-GPU box): chains sharded over devices 0 and 1, the laid-out
    problem broadcast device to device by RCCL (ncclCommInitAll + grouped ncclBroadcast over xGMI), checksummed on arrival
    -- same chains as one batch on device 0, bit for bit (reference semantics: gibbs/parallel.py:20-41, base.py:293-306:
    chain k owns generator k wherever it runs)."""
    from occuspytial_amd import _lib
    from occuspytial_amd._engine import Engine, EngineGroup
    if _lib.load().occ_device_count() < 2:
        pytest.skip('needs two GPUs')
    prob, _ = _problem_from_golden('ref_graph300_weighted')
    keys = [KEY + 3 * c for c in range(5)]
    plain = Engine(prob, keys)
    ref = _run(plain, prob, 5)
    plain.close()
    grp = EngineGroup(prob, keys, [0, 1])
    assert grp.transport.startswith('rccl broadcast (ncclCommInitAll), 2 devices'), grp.transport
    _same(ref, _run(grp, prob, 5))
    grp.close()
