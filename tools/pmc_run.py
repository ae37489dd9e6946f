"""Developer script: a few eager iterations of one workload for a rocprofv3 --pmc pass, on the product's library or -- OCC_LIB, a
tool-only knob -- on another build of it (the "before" side of a before / after comparison).
    OCC_EAGER_ONLY=1 [OCC_LIB=tools/libocc_gibbs_r3.so] rocprofv3 --pmc ... -- python3 tools/pmc_run.py ROWS COLS CHAINS ITERS"""
import ctypes, os, sys
sys.path.insert(0, '.')
import occuspytial_amd._lib as L
if os.environ.get('OCC_LIB'):
    L.LIB_PATH = os.path.abspath(os.environ['OCC_LIB'])
    L.ABI_VERSION = ctypes.CDLL(L.LIB_PATH).occ_abi_version()
from occuspytial_amd._engine import Engine
from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
from occuspytial_amd.utils import make_lattice_problem
rows, cols, chains, iters = (int(v) for v in sys.argv[1:5])
Q, W, X, y, *_ = make_lattice_problem(rows, cols, visits=5, p=2, q=2, random_state=0)
prob = FlatProblem(Q, W, X, y)
gens = chain_generators(10, chains)
eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
for i, g in enumerate(gens):
    st = default_start(g, prob)
    eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
eng.run(iters, iters - 1)
print('ran', iters, 'iterations of', rows, 'x', cols, 'x', chains, 'on', L.LIB_PATH)
eng.close()
