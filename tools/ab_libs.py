"""Developer script: one workload on two (or more) builds of the engine, alternating, in one process on one GPU.
    python tools/ab_libs.py [--lattice R C] [--chains N] [--iters K] [--warm W] tools/libocc_prev.so occuspytial_amd/libocc_gibbs.so
Prints, per build and repetition: us per iteration (host clock around occ_run) / mean k_iter (k_tiles) duration by its own clock."""
import argparse, ctypes, sys, time
sys.path.insert(0, '.')
import numpy as np
import occuspytial_amd._lib as L
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
ap = argparse.ArgumentParser()
ap.add_argument('--lattice', type=int, nargs=2, default=[100, 100])
ap.add_argument('--chains', type=int, default=4)
ap.add_argument('--iters', type=int, default=2000)
ap.add_argument('--warm', type=int, default=200)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--kernels', action='store_true', help='also: in-graph launch time of k_omega_a / k_z_ob / k_omega_b / k_noise (occ_profile)')
ap.add_argument('libs', nargs='+')
a = ap.parse_args()
Q, W, X, y, *_ = make_lattice_problem(a.lattice[0], a.lattice[1], visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
res = {p: [] for p in a.libs}
for rep in range(a.reps):
    for path in a.libs:
        L.LIB_PATH, L._lib = path, None
        L.ABI_VERSION = ctypes.CDLL(path).occ_abi_version()   # (an older build: its occ_stats is a prefix of today's)
        gens = chain_generators(10, a.chains)
        eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
        for i, g in enumerate(gens):
            st = default_start(g, prob)
            eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
        eng.run(a.warm, a.warm - 1)
        t0 = time.perf_counter(); eng.run(a.iters, a.iters - 1); dt = time.perf_counter() - t0
        st = eng.stats()
        pr = eng.profile(50) if a.kernels else None
        res[path].append((1e6 * dt / a.iters, st['iter_kernel_mean_us']) + ((pr['omega_a']['avg_us'], pr['z_ob']['avg_us'], pr['omega_b']['avg_us'], pr['noise']['avg_us']) if pr else ()))
        eng.close()
for p, v in res.items():
    print(p, ' '.join('/'.join('%.2f' % x for x in t) for t in v))
