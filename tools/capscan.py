"""Developer script: iteration time of the 100x100 x 4-chain workload versus the captured Krylov cap."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
C = int(os.environ.get('CHAINS', '4'))
gens = chain_generators(10, C)
keys = [int(g.bit_generator.random_raw()) for g in gens]
starts = [default_start(g, prob) for g in gens]
for cap in os.environ.get('CAPS', 'auto').split(','):
    if cap == 'auto':
        os.environ.pop('OCC_FORCE_KRYLOV_CAP', None)
    else:
        os.environ['OCC_FORCE_KRYLOV_CAP'] = cap
    eng = Engine(prob, keys)
    for i, st in enumerate(starts):
        eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
    eng.run(1200, 1199)
    c0 = eng.stats()['stalls']
    t0 = time.perf_counter(); eng.run(1000, 999); dt = time.perf_counter() - t0
    st = eng.stats()
    print(f"cap {cap:>5} -> {1e6*dt/1000:7.1f} us/iteration  (device {1e3*st['last_run_ms']/1000:7.1f})  cap_used {st['krylov_cap']}  carries {st['stalls']-c0}  kmean {st['krylov_mean']:.2f}")
    eng.close()
