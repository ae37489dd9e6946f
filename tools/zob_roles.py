"""Developer script: what k_z_ob's two roles cost (OCC_DEBUG_ZOB_SKIP: 1 = no z update, 2 = no omega_b draw, 3 = beta draw only)."""
import os, sys
sys.path.insert(0, '.')
rows, cols, chains = (int(v) for v in sys.argv[1:4])
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
Q, W, X, y, *_ = make_lattice_problem(rows, cols, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
for skip in (0, 1, 2, 3):
    os.environ['OCC_DEBUG_ZOB_SKIP'] = str(skip)
    gens = chain_generators(10, chains)
    eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
    for i, g in enumerate(gens):
        st = default_start(g, prob)
        eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
    eng.run(50, 49)
    pr = eng.profile(200)
    print(f'{rows}x{cols} x {chains}: skip={skip}  z_ob {pr["z_ob"]["avg_us"]:.2f} us   omega_b alone {pr["omega_b"]["avg_us"]:.2f}  omega_a {pr["omega_a"]["avg_us"]:.2f}', flush=True)
    eng.close()
