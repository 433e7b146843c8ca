import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` through gpurun)')


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must fail loudly (not skip) when there is no GPU or no HIP library."""
    return


@pytest.fixture(scope='session')
def oracle_libs():
    """Build the C oracle (and, when /root/reference is present, oracle/_ref) once."""
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    return True
