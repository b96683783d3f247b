import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scene_cache(tmp_path_factory):
    d = tmp_path_factory.mktemp("scenes")
    os.environ["RTR_SCENE_CACHE"] = str(d)
    return str(d)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def gpu_ctx():
    from realtimeraytracer_amd import api
    ctx = api.Context(0)   # raises RtrError(RTR_ERR_NO_DEVICE) without a GPU: no fallback
    yield ctx
    ctx.close()
