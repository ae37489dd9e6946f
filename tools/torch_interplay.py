"""Developer script: does initialising torch's HIP context change the cost of graph launches?"""
import os, sys, time
sys.path.insert(0, '.')
mode = sys.argv[1]
if mode in ('torch_first', 'torch_mid'):
    import torch
if mode == 'torch_first':
    torch.cuda.synchronize()
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, 4)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
for i, g in enumerate(gens):
    st = default_start(g, prob); eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
eng.run(100, 99)
if mode == 'torch_mid':
    torch.cuda.synchronize()
t0 = time.perf_counter(); eng.run(600, 599); dt = time.perf_counter() - t0
print(mode, f'{1e6*dt/600:.1f} us/iteration', eng.stats()['krylov_cap'])
