import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')


@pytest.fixture(scope='session')
def chess_bytes():
    with open(os.path.join(GOLDEN, 'chess.maray'), 'rb') as f:
        return f.read()
