import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle module (test infrastructure; built on demand with gcc)."""
    from oracle import occ_oracle
    occ_oracle.lib()
    return occ_oracle


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


GOLDEN_CASES = ['ref_queen150_ragged', 'ref_queen150_hparams', 'ref_rook400_v3', 'ref_queen400_v3',
                'ref_graph300_weighted']
