"""Developer script: which form of the fused iteration kernel an engine takes (OCC_VERBOSE=1 prints the probe's verdict)."""
import sys
sys.path.insert(0, '.')
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
for (r, c, ch) in [(60, 60, 4), (100, 100, 4), (100, 100, 8), (100, 100, 1)]:
    Q, W, X, y, *_ = make_lattice_problem(r, c, visits=5, p=2, q=2, random_state=0)
    prob = FlatProblem(Q, W, X, y)
    gens = chain_generators(10, ch)
    eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
    for i, g in enumerate(gens):
        st = default_start(g, prob)
        eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
    eng.run(50, 49)
    st = eng.stats()
    print(r, c, ch, {k: st[k] for k in ('persistent_solve', 'main_stream_cus', 'iter_kernel_mean_us', 'fused_fallbacks')}, flush=True)
    eng.close()
