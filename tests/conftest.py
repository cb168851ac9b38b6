import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU")
    # every changed-only exchange of the destination partition checks its run offsets two ways under test (hb_api_pass.inc exchange_pack:
    # the ranks' changed counters against the prefix sums over the changed bitmap); inherited by the worker processes the tests launch
    os.environ.setdefault("HB_CHECK_EXCHANGE", "1")
    # native pieces are built in-tree; build them when a test run starts without them
    need = [os.path.join(ROOT, "stract_amd", "lib", "libhyperball.so"),
            os.path.join(ROOT, "stract_amd", "lib", "libhyperball_exp.so"),
            os.path.join(ROOT, "stract_amd", "lib", "libhb_synth.so"),
            os.path.join(ROOT, "oracle", "libhb_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    from stract_amd import _lib

    if _lib.device_count() == 0:
        pytest.fail("GPU test selected but no HIP device is visible (no CPU fallback exists)")
    return _lib.Context
