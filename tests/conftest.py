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


@pytest.fixture(params=["plain", "binned"])
def queue_mode(request):
    """Runs a GPU test twice: with the shadow-ray queue in emission order and binned by direction octant (k_shadow_gen_oct +
    per-octant batch lists).  By default the library picks by queue size, so small test frames would only ever see the plain
    queue.  The choice is a tunable of the context (rtr_ctx_set_tunable "trace_binned"), which a context reads from the environment
    when it is created: the session's context is set directly, contexts a test makes itself get it from RTR_TRACE_BINNED."""
    v = 1 if request.param == "binned" else 0
    old = os.environ.get("RTR_TRACE_BINNED")
    os.environ["RTR_TRACE_BINNED"] = str(v)
    ctx = request.getfixturevalue("gpu_ctx")
    ctx.set_tunable("trace_binned", v)
    yield request.param
    ctx.set_tunable("trace_binned", 2)
    if old is None:
        os.environ.pop("RTR_TRACE_BINNED", None)
    else:
        os.environ["RTR_TRACE_BINNED"] = old
