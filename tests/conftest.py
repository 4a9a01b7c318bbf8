import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")
    config.addinivalue_line("markers", "oracle_as_is: compare with the unmodified oracle (a known divergence is being pinned)")


def has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def alice():
    import synth
    return synth.alice()


# Modules that test the ORACLE against the reference's own pins run it as it is.  Every other module compares the PRODUCT
# with the oracle and gets the oracle in the product's view of one reference behaviour the device path does not model yet
# (H5 bucket entries stored by StoreRangeOptBatch past the first ring revolution, see tests/orc.py); the divergence itself
# is pinned by tests/test_oracle.py::test_h5_store_range_masks_positions and test_emu_parity.py::test_known_divergence_*.
ORACLE_AS_IS = {"test_oracle", "test_oracle_vs_libbrotlienc", "test_synth"}


@pytest.fixture(autouse=True)
def _oracle_view(request):
    import orc
    mod = request.module.__name__.split(".")[-1]
    if mod in ORACLE_AS_IS or request.node.get_closest_marker("oracle_as_is"):
        old = orc.set_h5_absolute_store_range(False)
    else:
        old = orc.set_h5_absolute_store_range(True)
    yield
    orc.set_h5_absolute_store_range(bool(old))
