"""Developer script: the headline workload at three stages of a run (first iterations: K ~ 9; 200-2200: K ~ 13; after 20 000: K ~ 6)
for a given split of the CUs (OCC_DEBUG_COLD_CUS = main-stream CUs on the XCDs without a chain)."""
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
out = []
eng.run(5, 4)
for n in (20, 175, 2000, 20000, 4000):
    t0 = time.perf_counter(); eng.run(n, n - 1); dt = time.perf_counter() - t0
    st = eng.stats()
    out.append('%d: %.1f us (k_iter %.1f)' % (n, 1e6 * dt / n, st['iter_kernel_mean_us']))
print('main CUs %d | ' % eng.stats()['main_stream_cus'] + ' | '.join(out))
eng.close()
