import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import pta_bootstrap  # noqa: E402

pta_bootstrap.load()
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# PG_FORCE_REDUCER=1 (+ RANK/WORLD_SIZE/MASTER_* in the environment) runs the GPU suite with the data-parallel
# gradient reducer active at world size 1, so the bucketed RCCL all-reduce path is exercised on a single GPU.
if os.environ.get("PG_FORCE_REDUCER") == "1" and "RANK" in os.environ:
    from pose_transfer_amd.runtime import dp as _dp
    _dp.init_from_env()
