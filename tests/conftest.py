import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        ge._ensure_built()
    return ge.load_oracle()


@pytest.fixture(scope="session")
def pkg():
    """The product package; importing it loads libcugs_hip.so (no GPU needed to load)."""
    import __graft_entry__ as ge
    ge._ensure_built()
    return ge.load_package()


@pytest.fixture(scope="session")
def dev(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _reset_depth_route():
    """render()'s "this stream's views leave the fast depth range" bit is sticky by design (rasterizer._wide_depth);
    a test that provokes it must not steer the sort route of the tests after it."""
    yield
    mod = sys.modules.get("cugs_amd")
    if mod is not None and hasattr(mod, "rasterizer"):
        mod.rasterizer._wide_depth.clear()
