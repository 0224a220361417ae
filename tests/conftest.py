import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _oracle_built():
    """The oracle's C part (rotated overlap) is built by __graft_entry__.build(); build on demand for CPU runs."""
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_iou3d.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(autouse=True)
def _library_modes_reset(request):
    """The library's arithmetic mode and deterministic switch are process-global: every GPU test starts from the defaults (exact fp32
    products, atomics on) whatever an earlier test left behind."""
    if request.node.get_closest_marker("gpu") is not None:
        from radardistill_amd import kernels as K
        if K.get_conv_math() != "f32":
            K.set_conv_math("f32")
        if K.get_deterministic():
            K.set_deterministic(False)
    yield
