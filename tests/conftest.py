import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the package directory has a hyphen: import it once by path name, it registers the alias `neural_imaging_amd`
importlib.import_module('neural-imaging_amd')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_tables.npz'))


@pytest.fixture(autouse=True)
def _parity_mode_between_tests():
    """Every test starts in the float32 parity mode, whatever the previous one selected."""
    yield
    pkg = sys.modules.get('neural_imaging_amd')          # submodules are registered under the hyphenated package name
    ops = getattr(pkg, 'ops', None)
    if ops is not None:
        ops.set_compute('f32')
