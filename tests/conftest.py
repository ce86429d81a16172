import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The multi-block plans of the lane-per-window pipeline (vp_process_blocks_device) follow the batch size by default (round 6: where one
# block's windows fill the chip, short calls measured slower than block by block; vp_capi.hip v2_mb_min_blocks).  The suite's calls are
# short and some of its batches large: it forces the plans so that they stay exercised;
# test_pipeline_multi_block_plans_follow_the_batch_size checks the default.
os.environ.setdefault("VP_BOTH_MB_MIN", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
