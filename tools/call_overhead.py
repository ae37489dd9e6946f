"""Developer script: what one occ_run call costs beyond its iterations (the driver's bench runs 20 iterations per call)."""
import sys, time
sys.path.insert(0, '.')
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
for n in (2, 20, 200, 2000):
    reps = max(3, 2000 // n)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.run(n, 0)
    dt = (time.perf_counter() - t0) / reps
    print('run(%4d): %8.1f us per call, %6.1f us per iteration' % (n, 1e6 * dt, 1e6 * dt / n), flush=True)
eng.close()
