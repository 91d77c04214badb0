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


@pytest.fixture(scope="session")
def gpu_ctx_for():
    """gpu_ctx_for({"NRPHY_...": "1", ...}) -> a device context created under that environment.  The library reads its A/B
    knobs once, when a context is created (never on a submit path), so a test that wants another setting takes another
    context; one per distinct environment is kept for the session."""
    import contextlib
    import backends
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    made = {}

    @contextlib.contextmanager
    def environment(env):
        knobs = [k for k in os.environ if k.startswith("NRPHY_DECODER_") or k in ("NRPHY_CRC_REGIONS", "NRPHY_SCR_PARTS_BIG")]
        saved = {k: os.environ.get(k) for k in set(knobs) | set(env)}
        for k in knobs:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            yield
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    def get(env):
        key = tuple(sorted(env.items()))
        if key not in made:
            with environment(env):
                made[key] = backends.pkg.lib.Context(0)
        return made[key]

    yield get
    for ctx in made.values():
        ctx.close()
