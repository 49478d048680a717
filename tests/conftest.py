import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The golden train-mode fixtures were produced with the frozen towers in eval mode (make_golden.py), so the suite runs the frozen encoder's
# train-mode dropouts OFF unless a test turns them on (tests/test_dropout_gpu.py does, through monkeypatch).
os.environ.setdefault("SC_FROZEN_DROPOUT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
