import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    try:
        from gru4rec_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with `-m gpu`; when collected on a box without a GPU they are skipped
    # (never silently passed on a CPU fallback -- the product has none).
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no MI355X visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
