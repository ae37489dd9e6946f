"""Developer script: where k_rsr_solve spends its time.

    make -C occuspytial_amd/csrc stamps && python tools/rsr_stamps.py [m]

Loads tools/libocc_gibbs_stamps.so (-DOCC_SOLVE_STAMPS) and prints the wall_clock64 (100 MHz) deltas between the
stamp points of the last k_rsr_solve launch (chain 0, thread 0)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import occuspytial_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libocc_gibbs_stamps.so')
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators
from occuspytial_amd.utils import make_lattice_problem

m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
Q, W, X, y, *_ = make_lattice_problem(40, 50, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
prob.enable_rsr(q=m)
gens = chain_generators(10, 4)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
rng = np.random.default_rng(0)
for i in range(4):
    eng.set_start(i, rng.standard_normal(2), rng.standard_normal(2), 1.0, rng.standard_normal(m))
eng.run(100, 99)
lib = C.CDLL(L.LIB_PATH)
n = 48 * 12
buf = (C.c_ulonglong * n)()
assert lib.occ_debug_solve_stamps(buf, n) == n
t = np.array(buf[:8], dtype=np.int64)
names = ['tau (Qr matvec, gamma draw)', 'rhs (noise, E matvec, chunk sums)', 'Cholesky + forward', 'backward']
for i, nm in enumerate(names):
    print('%-36s %8.2f us' % (nm, (t[i + 1] - t[i]) / 100.0))
print('%-36s %8.2f us' % ('total', (t[4] - t[0]) / 100.0))
gs = np.array(buf[16:24], dtype=np.int64)
print('k_rsr_gram tile 0: loop %.2f us, reduction+store %.2f us; K\'u workgroup: u staged %.2f us, whole %.2f us' % (
    (gs[1] - gs[0]) / 100.0, (gs[3] - gs[1]) / 100.0, (gs[6] - gs[4]) / 100.0, (gs[7] - gs[4]) / 100.0))
ws = np.array(buf[24:40], dtype=np.int64)
print('tile 0, loop end of waves 0..15 after kernel start (us):', ' '.join('%.1f' % ((w - gs[0]) / 100.0) for w in ws))
w0 = np.array(buf[40:56], dtype=np.int64)
print('tile 0, start of waves 0..15 after kernel start (us):', ' '.join('%.1f' % ((w - gs[0]) / 100.0) for w in w0))
eng.close()
