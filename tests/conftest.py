import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # the GPU suite runs under the executor setting the published step time was measured with: an explicit call of the
    # entry point (nothing is set by `import glenet_amd`), before anything initialises the HIP runtime
    from glenet_amd import runtime
    runtime.configure_graph_executor(2)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from glenet_amd import _lib
    _lib.load()  # fail loudly if the HIP library was not built
    return torch.device("cuda", 0)

