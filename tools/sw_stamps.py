"""Developer script: where a step of k_iter's scalar-wave form spends its time (make -C occuspytial_amd/csrc stamps first).
Scalar wave of workgroup 0 / chain 0: points 0-5; first site wave: points 6-11 (shader-clock ticks)."""
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
STEPS, PTS = 48, 16
buf = (C.c_ulonglong * (STEPS * PTS))()
assert lib.occ_debug_solve_stamps(buf, STEPS * PTS) == STEPS * PTS
t = np.array(buf, dtype=np.int64).reshape(STEPS, PTS)
itn = int(eng.get('minres_itn', 0))
print('last solve of chain 0: %d iterations' % itn)
print('scalar wave: top->post_c+rot | ->B2 | pre+poll | ->B0 | post_ab | ->next top        site wave: top->g | ->B2 | rot+sums | drain+record | ->B0+gathers | ->next top    step')
for k in range(2, min(itn + 2, STEPS - 1)):
    sc = [t[k, 1] - t[k, 0], t[k, 2] - t[k, 1], t[k, 3] - t[k, 2], t[k, 4] - t[k, 3], t[k, 5] - t[k, 4], t[k + 1, 0] - t[k, 5]]
    si = [t[k, 7] - t[k, 6], t[k, 8] - t[k, 7], t[k, 9] - t[k, 8], t[k, 10] - t[k, 9], t[k, 11] - t[k, 10], t[k + 1, 6] - t[k, 11]]
    print('     B2->post_c %d, pre %d, poll %d' % (t[k,12]-t[k,2], t[k,13]-t[k,12], t[k,3]-t[k,13]))
    print('%3d  ' % k + ' '.join('%6d' % v for v in sc) + '     |   ' + ' '.join('%6d' % v for v in si) + '   %6d   site top - scalar top %d' % (t[k + 1, 0] - t[k, 0], t[k, 6] - t[k, 0]))
L_ = STEPS - 1
print('phase A: tau %d, rhs+p0 %d, barrier %d, to first step %d' % (t[0,1]-t[0,0], t[0,2]-t[0,1], t[0,3]-t[0,2], t[1,0]-t[0,3]))
print('phase C: proj sums+barrier %d, eta+beta partials+stats %d; kernel start to end %d' % (t[L_,1]-t[L_,0], t[L_,2]-t[L_,1], t[L_,2]-t[0,0]))
eng.close()
