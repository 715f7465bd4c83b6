import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def port():
    from oracle import oracle
    return oracle.port()


@pytest.fixture(scope="session")
def ref():
    from oracle import oracle
    return oracle.reference()


def split_blob(blob, offsets):
    blob = np.asarray(blob, np.uint8).tobytes()
    return [blob[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]
