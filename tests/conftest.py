import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_helpers.npz"))


@pytest.fixture(scope="session")
def golden_torch_only():
    """Outputs of the RaDe-GS python files that import with torch alone (loss_utils, general_utils, image_utils): made by
    tests/golden/make_golden.py in the build container, data only."""
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_torch_only.npz"))
