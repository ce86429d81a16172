import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The combined (pitch + vocoder) multi-block plan of vp_process_blocks_device only takes groups of eight blocks and more by default (round 6:
# shorter calls measured slower than block by block, vp_capi.hip process_both_blocks).  The suite's calls are short: it lowers the
# threshold so that they keep exercising the plan; test_combined_plan_small_groups_go_block_by_block checks the default.
os.environ.setdefault("VP_BOTH_MB_MIN", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
