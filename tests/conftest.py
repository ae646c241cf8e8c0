"""pytest configuration: `gpu` marker + repo root on sys.path."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "diag: exercises the experiment ABI of the diagnostics build (include/sdxlstep_diag.h part 2): "
                                       "runs with SDXL_DIAG=1 (libsdxlstep_diag.so), skipped against the product library")


def pytest_collection_modifyitems(config, items):
    import os
    if os.environ.get("SDXL_DIAG", "") == "1":
        return
    skip = pytest.mark.skip(reason="diagnostics build only (SDXL_DIAG=1 + build.py --diag)")
    for it in items:
        if "diag" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(ROOT / "tests" / "golden" / "loss_side.npz", allow_pickle=False)
