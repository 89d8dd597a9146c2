import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


# The library and the package read no PORESEG_* variable on their own (round 6).  The validation scripts (tools/gpu_validate.sh)
# run this suite in other modes by setting such variables: here -- in the test harness, not in the product -- they become the
# options every new engine.Context starts with.
from pypore_amd import engine as _engine  # noqa: E402

_engine.apply_env_defaults()
