import sys, time
sys.path.insert(0, '.')
import numpy as np
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, 4)
keys = [int(g.bit_generator.random_raw()) for g in gens]
Engine(prob, keys).close()   # context creation out of the way
for n in (2000, 20000):
    t0 = time.perf_counter()
    eng = Engine(prob, keys)
    for i, g in enumerate(gens):
        st = default_start(g, prob); eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
    t1 = time.perf_counter()
    rec = eng.run(n, 0)
    t2 = time.perf_counter()
    st = eng.stats()
    print(f'{n} iterations x 4 chains: create+upload+starts {1e3*(t1-t0):.1f} ms, run incl. download of the draws {1e3*(t2-t1):.1f} ms '
          f'(device {st["last_run_ms"]:.1f} ms) -> {4*n/(t2-t0):.0f} chain-it/s end to end, {4*n/(t2-t1):.0f} without the one-off set-up')
    eng.close()
