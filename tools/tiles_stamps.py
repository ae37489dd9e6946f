"""Developer script: where a step of k_tiles (occ_tiles.hpp) spends its time.

    make -C occuspytial_amd/csrc stamps && python tools/tiles_stamps.py [rows cols chains]

Loads tools/libocc_gibbs_stamps.so (-DOCC_SOLVE_STAMPS) and prints, for the last solve, the shader-clock deltas between
the stamp points of each step (chain 0, workgroup 0, thread 0)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import occuspytial_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libocc_gibbs_stamps.so')
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem

rows, cols, chains = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (500, 500, 1)
Q, W, X, y, *_ = make_lattice_problem(rows, cols, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, chains)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
for i, g in enumerate(gens):
    st = default_start(g, prob)
    eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
eng.run(100, 99)
print(eng.stats())
lib = C.CDLL(L.LIB_PATH)
STEPS, PTS = 48, 16
buf = (C.c_ulonglong * (STEPS * PTS))()
assert lib.occ_debug_solve_stamps(buf, STEPS * PTS) == STEPS * PTS
t = np.array(buf, dtype=np.int64).reshape(STEPS, PTS)
itn = int(eng.get('minres_itn', 0))
names = ['gather+vectors+store', 'sums+drain+sync', 'group sum+record', 'band poll(leader)', 'pre+bands poll', 'post+bcast', 'sync', 'to next']
pts = [0, 1, 3, 4, 5, 6, 7, 8]
print('last solve of chain 0: %d iterations; shader-clock ticks per segment (workgroup 0, thread 0)' % itn)
print('step ' + ' '.join('%15s' % n for n in names) + '   step total')
tot = np.zeros(len(names)); cnt = 0
for k in range(0, min(itn + 3, STEPS - 1)):
    d = [t[k, pts[j + 1]] - t[k, pts[j]] for j in range(7)] + [t[k + 1, 0] - t[k, 8]]
    print('%4d ' % k + ' '.join('%15d' % v for v in d) + '   %d' % (t[k + 1, 0] - t[k, 0]))
    if k >= 2:
        tot += np.array(d, dtype=float); cnt += 1
print('mean ' + ' '.join('%15.0f' % v for v in tot / max(cnt, 1)) + '   %.0f' % (tot.sum() / max(cnt, 1)))
Lr = STEPS - 1
print('phase A: loads %d, tau %d, rhs+p0 %d, to first step %d' % (t[0, 1] - t[0, 0], t[0, 2] - t[0, 1], t[0, 3] - t[0, 2], t[1, 0] - t[0, 3]))
print('phase C: projection %d, eta+beta partials %d; kernel (phase A start to end) %d' % (t[Lr, 1] - t[Lr, 0], t[Lr, 2] - t[Lr, 1], t[Lr, 2] - t[0, 0]))
eng.close()
