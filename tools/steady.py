"""Developer script: per-batch timing over a long run (looks for clock / state changes over time)."""
import os, sys, time
sys.path.insert(0, '.')
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, 4)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
for i, g in enumerate(gens):
    st = default_start(g, prob); eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
t0 = time.perf_counter(); eng.run(n, n - 1); dt = time.perf_counter() - t0
st = eng.stats()
print(f'{1e6*dt/n:.1f} us/iteration over {n}; cap {st["krylov_cap"]} carries {st["stalls"]} kmean {st["krylov_mean"]:.2f}')
