"""Developer script: long runs of the fused iteration kernel in each of its forms; a single lost or torn hand-over would show
as a device-side time-out (fused_fallbacks > 0) or as chains that differ between two identical runs."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
import os
cases = [(100, 100, 4, 300000, {}), (100, 100, 2, 100000, {}), (100, 100, 8, 60000, {}), (60, 60, 8, 150000, {}), (20, 20, 3, 150000, {}),
         (100, 100, 4, 60000, {'OCC_NO_XCD_LOCAL': '1'}), (250, 250, 2, 30000, {}), (500, 500, 1, 8000, {})]
for rows, cols, chains, iters, env in cases:
    for k in ('OCC_NO_XCD_LOCAL',):
        os.environ.pop(k, None)
    os.environ.update(env)
    Q, W, X, y, *_ = make_lattice_problem(rows, cols, visits=5 if rows > 20 else 3, p=2, q=2, random_state=0)
    prob = FlatProblem(Q, W, X, y)
    finals = []
    for rep in range(2):
        gens = chain_generators(10, chains)
        eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
        for i, g in enumerate(gens):
            st = default_start(g, prob)
            eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
        t0 = time.perf_counter()
        n = iters if rep == 0 else iters // 10
        a, b, t = eng.run(n, n - 1)
        dt = time.perf_counter() - t0
        st = eng.stats()
        assert st['fused_fallbacks'] == 0 and np.all(np.isfinite(a)) and np.all(t > 0), st
        finals.append((n, a.copy(), t.copy()))
        print(f'{rows}x{cols} x {chains} {env}: {n} iterations in {dt:.1f} s ({chains * n / dt:.0f} chain-it/s), form {st["persistent_solve"]}, fallbacks {st["fused_fallbacks"]}, K mean {st["krylov_mean"]:.2f}', flush=True)
        eng.close()
print('soak ok')
