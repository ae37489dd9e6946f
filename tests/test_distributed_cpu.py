"""The N > 1 paths on CPU: world_size-2 runs of the chain-sharding driver (``occuspytial_amd.distributed``: problem
broadcast, chain -> rank map, per-chain seeding, result gather) over a gloo group and over the product's own file
rendezvous, and the in-process fan-out (one host thread per device) over a stub engine factory."""
import os
import subprocess
import sys

import numpy as np
import pytest

from .conftest import ROOT


def test_shard_chains_round_robin():
    from occuspytial_amd.distributed import shard_chains
    assert shard_chains(5, 2, 0) == [0, 2, 4] and shard_chains(5, 2, 1) == [1, 3]
    assert shard_chains(4, 8, 6) == [] and shard_chains(8, 8, 3) == [3]
    assert sorted(sum((shard_chains(7, 3, r) for r in range(3)), [])) == list(range(7))


def _single_process_reference(oracle, n_chains):
    from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(8, 9, visits=3, p=2, q=2, random_state=4)
    del W[5], y[5]
    prob = FlatProblem(Q, W, X, y)
    gens = chain_generators(77, n_chains)
    out = []
    for c in range(n_chains):
        st = default_start(gens[c], prob)
        orc = oracle.OracleSampler(prob, int(gens[c].bit_generator.random_raw()))
        orc.set_start(st['alpha'], st['beta'], st['tau'], st['eta'])
        out.append(orc.run(12, 2))
    return out


def _check(out, oracle, n_chains):
    r0, r1 = np.load(out + '.rank0.npz'), np.load(out + '.rank1.npz')
    assert r0['mine'].tolist() == [0, 2] and r1['mine'].tolist() == [1]
    for k in ('A', 'B', 'T'):
        assert np.array_equal(r0[k], r1[k])          # every rank holds the gathered result
    assert r0['A'].shape == (3, 10, 2) and r0['T'].shape == (3, 10)
    # single-process reference: same seeding rule, chain by chain
    for c, (a, b, t) in enumerate(_single_process_reference(oracle, n_chains)):
        assert np.array_equal(a, r0['A'][c]) and np.array_equal(b, r0['B'][c]) and np.array_equal(t, r0['T'][c])
    assert not np.allclose(r0['T'][0], r0['T'][1])   # chains differ from one another


def test_two_rank_gloo_run_equals_single_process(tmp_path, oracle):
    out = str(tmp_path / 'dist')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', PYTHONPATH=ROOT)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29631',
           os.path.join(ROOT, 'tests', '_dist_worker.py'), out, '3', 'gloo']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    _check(out, oracle, 3)


def test_two_rank_file_rendezvous_run_equals_single_process(tmp_path, oracle):
    """The product's own communicator (no torch anywhere): two plain processes that find each other through RANK /
    WORLD_SIZE / MASTER_PORT and a rendezvous directory keyed by the launcher's PID."""
    out = str(tmp_path / 'dist')
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_PORT='29633', PYTHONPATH=ROOT)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_dist_worker.py'), out, '3', 'file'],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
    _check(out, oracle, 3)


def test_in_process_fan_out_over_devices_equals_one_batch(oracle):
    """``EngineGroup`` (what ``LogitICARGibbs(..., devices=[...])`` runs on): chain c on device c % G, one host thread
    per device, results and state addressed by global chain number -- over a stub engine factory (the CPU oracle), for
    G = 1, 2, 3 and more devices than chains: always the chains of a single batch, bit for bit."""
    from occuspytial_amd._engine import EngineGroup
    from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
    from occuspytial_amd.utils import make_lattice_problem
    from ._dist_worker import OracleEngine
    Q, W, X, y, *_ = make_lattice_problem(7, 8, visits=3, p=2, q=2, random_state=9)
    prob = FlatProblem(Q, W, X, y)
    gens = chain_generators(5, 5)
    starts = [default_start(g, prob) for g in gens]
    keys = [int(g.bit_generator.random_raw()) for g in gens]
    made = []

    def factory(prob_, keys_, device):
        made.append((device, list(keys_)))
        return OracleEngine(prob_, keys_, device)

    ref = None
    for devices in ([0], [0, 1], [3, 1, 2], list(range(8))):
        made.clear()
        grp = EngineGroup(prob, keys, devices, engine_factory=factory)
        G = min(len(devices), 5)
        assert [d for d, _ in made] == devices[:G]
        assert [k for _, k in made] == [[keys[c] for c in range(5) if c % G == g] for g in range(G)]
        for c, st in enumerate(starts):
            grp.set_start(c, st['alpha'], st['beta'], st['tau'], st['eta'])
        rec = grp.run(9, 3)
        assert rec[0].shape == (5, 6, 2) and rec[2].shape == (5, 6)
        if ref is None:
            ref = rec
        for u, v in zip(ref, rec):
            assert np.array_equal(u, v)
        grp.close()
    with pytest.raises(ValueError):
        EngineGroup(prob, keys, [0, 1], engine_factory=factory).set_keys(keys[:2])
