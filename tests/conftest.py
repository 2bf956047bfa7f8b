import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/pose_estimators")
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    for item in items:
        if "reference" in item.keywords and not have_ref:
            item.add_marker(skip_ref)
