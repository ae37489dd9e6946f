"""world_size-2 gloo run of the chain-sharding driver on CPU (the N > 1 path of bench.py and
occuspytial_amd.distributed): problem broadcast, chain->rank map, per-chain seeding, result gather."""
import os
import subprocess
import sys

import numpy as np

from .conftest import ROOT


def test_shard_chains_round_robin():
    from occuspytial_amd.distributed import shard_chains
    assert shard_chains(5, 2, 0) == [0, 2, 4] and shard_chains(5, 2, 1) == [1, 3]
    assert shard_chains(4, 8, 6) == [] and shard_chains(8, 8, 3) == [3]
    assert sorted(sum((shard_chains(7, 3, r) for r in range(3)), [])) == list(range(7))


def test_two_rank_gloo_run_equals_single_process(tmp_path, oracle):
    n_chains = 3
    out = str(tmp_path / 'dist')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', PYTHONPATH=ROOT)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29631',
           os.path.join(ROOT, 'tests', '_dist_worker.py'), out, str(n_chains)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    r0, r1 = np.load(out + '.rank0.npz'), np.load(out + '.rank1.npz')
    assert r0['mine'].tolist() == [0, 2] and r1['mine'].tolist() == [1]
    for k in ('A', 'B', 'T'):
        assert np.array_equal(r0[k], r1[k])          # every rank holds the gathered result
    assert r0['A'].shape == (3, 10, 2) and r0['T'].shape == (3, 10)

    # single-process reference: same seeding rule, chain by chain
    from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
    from occuspytial_amd.utils import make_lattice_problem
    Q, W, X, y, *_ = make_lattice_problem(8, 9, visits=3, p=2, q=2, random_state=4)
    del W[5], y[5]
    prob = FlatProblem(Q, W, X, y)
    gens = chain_generators(77, n_chains)
    for c in range(n_chains):
        st = default_start(gens[c], prob)
        orc = oracle.OracleSampler(prob, int(gens[c].bit_generator.random_raw()))
        orc.set_start(st['alpha'], st['beta'], st['tau'], st['eta'])
        a, b, t = orc.run(12, 2)
        assert np.array_equal(a, r0['A'][c]) and np.array_equal(b, r0['B'][c]) and np.array_equal(t, r0['T'][c])
    # chains differ from one another
    assert not np.allclose(r0['T'][0], r0['T'][1])
