"""Developer script: duration of the Polya-Gamma draw kernel (k_draw via occ_draw) for several z patterns; run under
rocprofv3 --kernel-trace --stats and read k_draw's rows (one launch per pattern, in this order)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from occuspytial_amd._engine import device_draw
rng = np.random.default_rng(0)
n = 40000
pats = {'z=0': np.zeros(n), 'z=1': np.ones(n), 'z=3': np.full(n, 3.0), 'z=8': np.full(n, 8.0),
        'N(0,1.5)': rng.normal(0, 1.5, n), 'N(0,4)': rng.normal(0, 4, n)}
for rep in range(3):
    for name, z in pats.items():
        device_draw('pg1', z, key=5 + rep, it=rep)
print(list(pats))
