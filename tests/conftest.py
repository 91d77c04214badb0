import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import backends
    return backends.oracle()


@pytest.fixture(scope="session")
def ref():
    import backends
    r = backends.ref()
    if r is None:
        pytest.skip("oracle/_ref/libsrsref.so not built (needs /root/reference: make -C oracle ref)")
    return r


@pytest.fixture(scope="session")
def gpu_ctx():
    """A device context of the HIP library.  Fails (does not skip) when the library is missing."""
    import backends
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = backends.pkg.lib.Context(0)
    yield ctx
    ctx.close()
