import sys; sys.path.insert(0, '.')
import numpy as np
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, 4)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
for i, g in enumerate(gens):
    st = default_start(g, prob); eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
ks = []
for it in range(400):
    eng.step()
    ks.append([int(eng.get('minres_itn', c)) for c in range(4)])
ks = np.array(ks)
print('mean', ks.mean(), 'sd', ks.std(), 'min', ks.min(), 'max', ks.max())
print('hist', np.bincount(ks.ravel()))
print('first 30 of chain 0', ks[:30, 0])
print('tau chain0', [round(float(eng.get('tau', c)), 4) for c in range(4)])
