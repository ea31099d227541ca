import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle as orc
    orc.build()
    return orc


class Cfg:
    """Attribute bag standing in for the namedtuple parse_ini returns (consumers use getattr(cfg, k, default))."""

    def __init__(self, **kw):
        self.__dict__.update(kw)
