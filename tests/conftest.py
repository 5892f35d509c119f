import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def states():
    """Seeded synthetic weights, generated once per session."""
    from terran_amd import weights
    cache = {}

    def get(name):
        if name not in cache:
            if name.startswith('wild_'):                  # trained-looking statistics (tests/wild_weights.py)
                from tests import wild_weights
                cache[name] = wild_weights.MAKERS[name[5:]]()
            else:
                cache[name] = getattr(weights, 'make_%s_state' % name)()
        return cache[name]
    return get
