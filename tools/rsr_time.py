"""Developer script: LogitRSRGibbs engine timing.  python tools/rsr_time.py rows cols m chains iters"""
import sys, time
sys.path.insert(0, '.')
import os
import numpy as np
if os.environ.get('OCC_LIB_PATH'):
    import occuspytial_amd._lib as L
    L.LIB_PATH = os.environ['OCC_LIB_PATH']
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators
from occuspytial_amd.utils import make_lattice_problem
rows, cols, m, chains, n = (int(v) for v in sys.argv[1:6])
Q, W, X, y, *_ = make_lattice_problem(rows, cols, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
t0 = time.perf_counter(); prob.enable_rsr(q=m); t_basis = time.perf_counter() - t0
gens = chain_generators(10, chains)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
rng = np.random.default_rng(0)
for i in range(chains):
    eng.set_start(i, rng.standard_normal(2), rng.standard_normal(2), 1.0, rng.standard_normal(m))
eng.run(50, 49)
t0 = time.perf_counter(); eng.run(n, n - 1); dt = time.perf_counter() - t0
print(f'{rows}x{cols} sites, {m} basis columns, {chains} chains: {1e6*dt/n:.1f} us/iteration, {chains*n/dt:.0f} chain-it/s (basis set-up {t_basis:.1f} s on the host)')
