import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from tests.helpers import oracle as o
    o.build()
    o.lib()
    return o


@pytest.fixture(scope='session')
def avifdec():
    from tests.helpers import avifdec as d
    if not d.available():
        pytest.skip('bundled libavif/dav1d not available')
    return d
