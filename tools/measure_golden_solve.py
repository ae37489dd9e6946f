"""Developer script (GPU box): worst-case deviation of the device's eta solve from the REFERENCE's recorded solves, per
fixture, in the production arithmetic and in scipy's own (OCC_DEBUG_EXACT_DIV=1).  The numbers go into
tests/test_gpu_golden.py (asserted with 2x headroom)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import GOLDEN_CASES, load_golden          # noqa: E402
from tests.test_gpu_parity import _problem_from_golden       # noqa: E402
from tests.test_gpu_golden import _iters, _seat, KEY          # noqa: E402
from occuspytial_amd._engine import Engine                    # noqa: E402

for mode in ('0', '1'):
    os.environ['OCC_DEBUG_EXACT_DIV'] = mode
    for name in GOLDEN_CASES:
        g = load_golden(name)
        prob, start = _problem_from_golden(name)
        eng = Engine(prob, [KEY])
        n = prob.n
        wx = we = 0.0
        for it in _iters(g):
            t = f'it{it}_'
            om, tau = g[t + 'omega_b'], float(g[t + 'tau'])
            eps1 = g[t + 'eta_eps'][:n]
            prior = (g[t + 'eta_rhs'] - g[t + 'eta_b'] - np.sqrt(om) * eps1) / np.sqrt(tau)
            _seat(eng, start, beta=g[t + 'eta_beta'], tau=tau, z=g[t + 'eta_k'] + 0.5, xz=g[t + 'eta_x0'])
            rhs, xz, eta, itn = eng.cond_eta(om, eps1, prior)
            assert itn == int(g[t + 'eta_itn']), (name, it, itn, int(g[t + 'eta_itn']))
            wx = max(wx, np.abs(xz - g[t + 'eta_xz']).max() / np.abs(g[t + 'eta_xz']).max())
            we = max(we, np.abs(eta - g[t + 'eta']).max() / np.abs(g[t + 'eta']).max())
        eng.close()
        print('exact_div=%s %-24s xz %.3e eta %.3e' % (mode, name, wx, we), flush=True)
