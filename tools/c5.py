"""Developer script: BASELINE config 5 (irregular areal graph, ~3000 units, mean degree ~6, 10 visits, 4 chains) on one GPU."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_graph_problem
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Q, W, X, y, *_ = make_graph_problem(3000, 6, visits=10, p=2, q=2, random_state=0)
deg = np.diff(Q.indptr) - 1
print('n', Q.shape[0], 'mean degree %.2f' % deg.mean(), 'max degree', deg.max(), 'rows > 8 neighbours:', int((deg > 8).sum()))
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, chains)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
for i, g in enumerate(gens):
    st = default_start(g, prob); eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
eng.run(100, 99)
n = 1500
t0 = time.perf_counter(); eng.run(n, n - 1); dt = time.perf_counter() - t0
st = eng.stats()
print(f'{1e6*dt/n:.1f} us/iteration  {chains*n/dt:.0f} chain-it/s  fused={st["persistent_solve"]} kmean={st["krylov_mean"]:.1f} cap={st["krylov_cap"]}')
