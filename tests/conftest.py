import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_addoption(parser):
    parser.addoption('--parity-soft', action='store_true', default=False,
                     help='measurement protocol only (tools/tier_round.sh): tests/parity.py PRINTS a bound that does not hold '
                          'instead of raising.  A run with this option is no test result.')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    if os.environ.get('NEFII_PARITY_SOFT', '') not in ('', '0'):
        # rounds 4-5 read this variable in tests/parity.py: set in a shell or CI environment it turned every parity assertion
        # into a print.  It is no longer honoured - and a session that still carries it stops here, loudly.
        raise pytest.UsageError('NEFII_PARITY_SOFT is set in the environment: parity assertions are never softened by an '
                                'environment variable; unset it (a measurement tool passes --parity-soft explicitly)')
    if config.getoption('--parity-soft'):
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import parity
        parity.SOFT = True
        print('\n*** --parity-soft: parity bounds are PRINTED, not asserted - this run is a measurement, not a test result ***')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    import torch

    def load(name):
        z = np.load(os.path.join(GOLDEN, name + '.npz'))
        return {k: torch.from_numpy(z[k]) for k in z.files}
    return load
