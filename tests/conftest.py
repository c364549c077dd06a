import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# property tests (tests/test_properties.py): the same examples on every run, no example database in the tree
try:
    from hypothesis import settings as _hyp_settings
    _hyp_settings.register_profile("repo", derandomize=True, database=None, deadline=None)
    _hyp_settings.load_profile(os.environ.get("HYPOTHESIS_PROFILE", "repo"))
except ImportError:            # hypothesis is in the image; the property tests skip themselves without it
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
