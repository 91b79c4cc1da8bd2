import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# a crash inside the library or the HIP runtime under it prints its native frames before Python's faulthandler has its say
# (csrc/config.cpp: install_crash_backtrace)
os.environ.setdefault("PIB_CRASH_BACKTRACE", "1")
# The suite never imports torch in the pytest process (tests/test_distributed_gloo.py imports it inside its functions), so the library
# can run on /opt/rocm's own HIP / ROCr -- the runtime a C or C++ application linking libpetibm_amd.so gets -- instead of the older one
# torch bundles, under which runs of this suite died now and then inside the runtime's completion thread (docs/history/round4.md).
# bench.py and __graft_entry__.smoke() import torch themselves and keep its runtime.  PIB_TORCH_FIRST=1 restores the old order.
os.environ.setdefault("PIB_TORCH_FIRST", "0")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    from petibm_amd.build import build_library
    return build_library()
