"""Developer script: where the first tile row of k_rsrb_step (large-basis reduced-rank model) spends its time.

    make -C occuspytial_amd/csrc stamps && python tools/rsrb_stamps.py

Loads tools/libocc_gibbs_stamps.so (-DOCC_SOLVE_STAMPS) and prints the wall_clock64 (100 MHz) deltas between the stamp
points of panel step 20 of the last iteration (chain 0, second workgroup of the first tile row, thread 0)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import occuspytial_amd._lib as L
L.LIB_PATH = os.environ.get('OCC_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libocc_gibbs_stamps.so'))
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators
from occuspytial_amd.utils import make_lattice_problem

Q, W, X, y, *_ = make_lattice_problem(100, 100, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
prob.enable_rsr(q=1280)
m = prob.rsr['dim']
gens = chain_generators(10, 3)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
rng = np.random.default_rng(0)
for i in range(3):
    eng.set_start(i, rng.standard_normal(2), rng.standard_normal(2), 1.0, 0.1 * rng.standard_normal(m))
eng.run(10, 9)
lib = C.CDLL(L.LIB_PATH)
n = 48 * 16
buf = (C.c_ulonglong * n)()
assert lib.occ_debug_solve_stamps(buf, n) == n
t = np.array(buf[64:88], dtype=np.int64).reshape(6, 4)   # [point][wave]
t0 = t[0].min()
names = ['start (after the right-hand side / exit tests)', 'diagonal tiles in LDS, barrier passed', 'factor done (wave 3) / own tiles done (waves 0-2)',
         'second barrier passed', 'column solve done (wave 0)', 'stores done']
print('m = %d, panel step 20, second workgroup of the first tile row; us after the first wave\'s start, waves 0..3' % m)
for i, nm in enumerate(names):
    print('%-52s %s' % (nm, ' '.join('%7.2f' % ((v - t0) / 100.0) if v else '      -' for v in t[i])))
rp = np.array(buf[88:96], dtype=np.int64)
print('wave 3: block in registers %.2f, rows 0-15 %.2f, A22 on the matrix cores %.2f, rows 16-31 %.2f; then V11, V22 to LDS (or the factor to global memory)' % tuple(
    (v - t0) / 100.0 for v in rp[:4]))
eng.close()
