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
    config.addinivalue_line("markers", "slow: full-size configurations (tens of seconds each on the GPU)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests are skipped (not failed) where there is no HIP device or no built library."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    have_lib = os.path.exists(os.path.join(ROOT, "compression_amd", "libtfc_hip.so"))
    if have_gpu and have_lib:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X) and compression_amd/libtfc_hip.so")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture
def port(request):
    """The checker.  GPU-tier tests compare the HIP path with the reference's own sources compiled in place
    (oracle/_ref, shipped to the GPU box next to the library) where that was built, else with the restatement; the
    CPU tier pins the restatement to the compiled reference and to the golden vectors (tests/test_oracle.py), so it
    takes the restatement itself."""
    from oracle import oracle
    return oracle.best() if request.node.get_closest_marker("gpu") else oracle.port()


@pytest.fixture(scope="session")
def ref():
    from oracle import oracle
    return oracle.reference()


def split_blob(blob, offsets):
    blob = np.asarray(blob, np.uint8).tobytes()
    return [blob[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]
