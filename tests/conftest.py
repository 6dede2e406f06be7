import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """no test may wait for ever (a dead-locked worker pool once did): with pytest-timeout installed every test gets a 15-minute ceiling"""
    if config.pluginmanager.hasplugin("timeout"):
        for it in items:
            if it.get_closest_marker("timeout") is None:
                it.add_marker(pytest.mark.timeout(900))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
