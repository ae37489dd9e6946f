"""Developer script: do two builds of the library give the same bits?  python tools/compare_libs.py other.so"""
import os, sys, subprocess, pickle
sys.path.insert(0, '.')
if len(sys.argv) > 2 and sys.argv[2] == 'child':
    import numpy as np
    import occuspytial_amd._lib as L
    if sys.argv[1] != 'default':
        L.LIB_PATH = sys.argv[1]
    from occuspytial_amd._engine import Engine
    from occuspytial_amd._problem import FlatProblem, chain_generators, default_start
    from occuspytial_amd.utils import make_lattice_problem
    out = []
    for (r, c, ch, it) in ((60, 60, 2, 60), (100, 100, 4, 40), (23, 31, 3, 80)):
        Q, W, X, y, *_ = make_lattice_problem(r, c, visits=3, p=2, q=2, random_state=1)
        prob = FlatProblem(Q, W, X, y)
        gens = chain_generators(10, ch)
        eng = Engine(prob, [int(g.bit_generator.random_raw()) for g in gens])
        for i, g in enumerate(gens):
            st = default_start(g, prob)
            eng.set_start(i, st['alpha'], st['beta'], st['tau'], st['eta'])
        rec = eng.run(it, 0)
        out.append([np.asarray(a).tobytes() for a in rec] + [eng.get('eta', i).tobytes() for i in range(ch)] + [eng.get('xz', i).tobytes() for i in range(ch)])
        eng.close()
    sys.stdout.buffer.write(pickle.dumps(out))
else:
    a = pickle.loads(subprocess.run([sys.executable, __file__, 'default', 'child'], capture_output=True, check=True).stdout)
    b = pickle.loads(subprocess.run([sys.executable, __file__, sys.argv[1], 'child'], capture_output=True, check=True).stdout)
    print('identical bits' if a == b else 'DIFFERENT bits', [x == y for x, y in zip(a, b)])
