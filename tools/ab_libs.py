"""Developer script: the headline workload on two builds of the engine, alternating, in one process on one GPU.
    python tools/ab_libs.py tools/libocc_prev.so occuspytial_amd/libocc_gibbs.so"""
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np
import occuspytial_amd._lib as L
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
res = {p: [] for p in sys.argv[1:]}
for rep in range(3):
    for path in sys.argv[1:]:
        L.LIB_PATH, L._lib = path, None
        gens = chain_generators(10, 4)
        eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
        for i, g in enumerate(gens):
            st = default_start(g, prob)
            eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
        eng.run(200, 199)
        t0 = time.perf_counter(); eng.run(2000, 1999); dt = time.perf_counter() - t0
        st = eng.stats()
        res[path].append((1e6 * dt / 2000, st['iter_kernel_mean_us']))
        eng.close()
for p, v in res.items():
    print(p, ' '.join('%.2f/%.2f' % t for t in v))
