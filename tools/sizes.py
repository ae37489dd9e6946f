"""Developer script: iterations/s over a few problem sizes and chain counts (which path each one takes)."""
import sys, time
sys.path.insert(0, '.')
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
import os
cases = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(20, 20, 1, 2000), (100, 100, 1, 1500), (100, 100, 2, 1500), (100, 100, 4, 1500), (100, 100, 6, 1500),
                                     (100, 100, 8, 1000), (100, 100, 16, 600), (250, 250, 1, 400), (500, 500, 1, 200)]
for (rows, cols, chains, iters) in cases:
    Q, W, X, y, *_ = make_lattice_problem(rows, cols, visits=5 if rows > 20 else 3, p=2, q=2, random_state=0)
    prob = FlatProblem(Q, W, X, y)
    gens = chain_generators(10, chains)
    eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
    for i, g in enumerate(gens):
        st = default_start(g, prob)
        eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
    eng.run(100, 99)
    t0 = time.perf_counter(); eng.run(iters, iters - 1); dt = time.perf_counter() - t0
    st = eng.stats()
    print(f'{rows}x{cols} x {chains} chains: {1e6*dt/iters:8.1f} us/iteration  {chains*iters/dt:9.0f} chain-it/s  '
          f'fused={st["persistent_solve"]} main_cus={st["main_stream_cus"]} kmean={st["krylov_mean"]:.1f} cap={st["krylov_cap"]}', flush=True)
    eng.close()
