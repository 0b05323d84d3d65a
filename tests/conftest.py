import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import lip_amd  # noqa: E402,F401  (registers the package alias `lip_amd`; `src` is its twin)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # On a GPU box the `-m gpu` tests must exercise the HIP path; on a CPU-only box they are
    # deselected by `-m "not gpu"`.  If someone runs them without a GPU, skip instead of failing.
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
