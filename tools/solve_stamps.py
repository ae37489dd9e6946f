"""Developer script: where a step of the persistent eta solve (k_solve) spends its time.

    make -C occuspytial_amd/csrc stamps && python tools/solve_stamps.py

Loads tools/libocc_gibbs_stamps.so (built with -DOCC_SOLVE_STAMPS), runs a few hundred iterations of the
headline workload and prints, for the last solve, the s_memtime deltas between the stamp points of each step
(chain 0, workgroup 0, thread 0)."""
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

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, chains)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
for i, g in enumerate(gens):
    st = default_start(g, prob)
    eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
eng.run(300, 299)
lib = C.CDLL(L.LIB_PATH)
STEPS, PTS = 48, 12
buf = (C.c_ulonglong * (STEPS * PTS))()
assert lib.occ_debug_solve_stamps(buf, STEPS * PTS) == STEPS * PTS
t = np.array(buf, dtype=np.int64).reshape(STEPS, PTS)
itn = int(eng.get('minres_itn', 0))
names = ['top', 'vectors+drain', 'wg barrier', 'record store', 'minres_pre', '-', 'poll', 'gathers issue', 'wave sums+sync', 'minres_post', 'hand-over', 'to next step']
pts = [0, 1, 2, 3, 4, 9, 5, 6, 7, 8, 10, 11]
print('last solve of chain 0: %d iterations; shader-clock ticks per segment' % itn)
print('step ' + ' '.join('%13s' % n for n in names) + '   step total')
tot = np.zeros(len(names))
cnt = 0
for k in range(2, min(itn + 1, STEPS - 1)):
    d = [t[k, pts[j + 1]] - t[k, pts[j]] for j in range(11)] + [t[k + 1, 0] - t[k, 11]]
    print('%4d ' % k + ' '.join('%13d' % v for v in d) + '   %d' % (t[k + 1, 0] - t[k, 0]))
    tot += np.array(d, dtype=float)
    cnt += 1
print('mean ' + ' '.join('%13.0f' % v for v in tot / max(cnt, 1)) + '   %.0f' % (tot.sum() / max(cnt, 1)))
cyc = lambda a, b: int(b - a)
print('phase A: tau %d, rhs+p0 %d, barrier %d, to first step %d' % (cyc(t[0,0], t[0,1]), cyc(t[0,1], t[0,2]), cyc(t[0,2], t[0,3]), cyc(t[0,3], t[1,0])))
L = STEPS - 1
print('phase C: proj sums+barrier %d, eta+beta partials+stats %d; kernel start to end %d' % (cyc(t[L,0], t[L,1]), cyc(t[L,1], t[L,2]), cyc(t[0,0], t[L,2])))
flat = np.array(buf, dtype=np.int64)
arr, rel = flat[480:520], flat[520:560]
if arr.min() > 0:
    a0 = arr.min()
    print('step 6, workgroups 0..39 of chain 0: compute done at (ticks after the first):', ' '.join('%d' % (v - a0) for v in arr))
    print('                                      barrier passed at:', ' '.join('%d' % (v - a0) for v in rel))
eng.close()
