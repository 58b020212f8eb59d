import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model():
    from uhc_amd.sim import load_asset_model
    return load_asset_model()


@pytest.fixture(scope="session")
def ctrl(model):
    from uhc_amd.sim import make_ctrl
    return make_ctrl(model)


@pytest.fixture(scope="session")
def standing():
    z = np.load(os.path.join(ROOT, "uhc_amd", "assets", "standing_neutral.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle.build import build
    build()
