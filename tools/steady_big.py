"""Developer script: like steady.py for an arbitrary lattice / chain count.  python tools/steady_big.py rows cols chains iters"""
import sys, time
sys.path.insert(0, '.')
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
rows, cols, chains, n = (int(v) for v in sys.argv[1:5])
Q, W, X, y, *_ = make_lattice_problem(rows, cols, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, chains)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
for i, g in enumerate(gens):
    st = default_start(g, prob); eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
eng.run(40, 39)
t0 = time.perf_counter(); eng.run(n, n - 1); dt = time.perf_counter() - t0
st = eng.stats()
print(f'{1e6*dt/n:.1f} us/iteration over {n}; cap {st["krylov_cap"]} kmean {st["krylov_mean"]:.2f} fused {st["persistent_solve"]}')
