import sys, time
sys.path.insert(0, '.')
import numpy as np
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, 4)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
for i, g in enumerate(gens):
    st = default_start(g, prob)
    eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
eng.run(300, 299)
import os
os.environ['OCC_VERBOSE'] = '1'
for n in (2, 20):
    t0 = time.perf_counter(); eng.run(n, 0); dt = time.perf_counter() - t0
    print('run(%d) %.1f us  device %.1f us' % (n, 1e6 * dt, 1e3 * eng.stats()['last_run_ms']), flush=True)
os.environ.pop('OCC_VERBOSE')
# python-side cost: time the ctypes call alone vs the wrapper
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): eng.run(2, 0)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(8)
eng.close()
