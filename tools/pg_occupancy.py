"""Developer script: is a Polya-Gamma wave bound by its own latency or by the SIMD's issue rate?  k_draw (occ_draw 'pg1') on
65 536 x W arguments -- W waves per SIMD on the whole device -- for W = 1, 2, 3, 4, 8; run under
rocprofv3 --kernel-trace --output-format csv and read k_draw's durations (launch order = the order printed here)."""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import occuspytial_amd._lib as L
if os.environ.get('OCC_LIB'):   # another build of the engine (A/B)
    L.LIB_PATH = os.environ['OCC_LIB']
from occuspytial_amd._engine import device_draw
rng = np.random.default_rng(0)
order = []
for rep in range(3):
    for w in (1, 2, 3, 4, 8):
        z = rng.normal(0.0, float(os.environ.get('PG_SD', '1.0')), 65536 * w)
        device_draw('pg1', z, key=5 + rep, it=rep)
        order.append(w)
print('waves per SIMD, in launch order:', order)
