import os, sys
sys.path.insert(0, '.')
import occuspytial_amd._lib as L
if len(sys.argv) > 1: L.LIB_PATH = os.path.abspath(sys.argv[1])
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, 4)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
for i, g in enumerate(gens):
    st = default_start(g, prob); eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
eng.run(200, 199)
p = eng.profile(200)
print({k: round(v['avg_us'], 2) for k, v in p.items()})
