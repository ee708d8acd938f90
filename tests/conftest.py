import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session", autouse=True)
def _rank_logs_in_tmp(tmp_path_factory):
    """bench.py's self-spawned ranks log under gpurun_out/ by default (it travels back from the GPU box); under pytest the
    logs go to a session tmp dir so that the suite leaves nothing behind in the tree."""
    old = os.environ.get("APZ_RANK_LOG_DIR")
    os.environ["APZ_RANK_LOG_DIR"] = str(tmp_path_factory.mktemp("rank_logs"))
    yield
    if old is None:
        os.environ.pop("APZ_RANK_LOG_DIR", None)
    else:
        os.environ["APZ_RANK_LOG_DIR"] = old


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
