import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests never silently pass on a CPU box: they are skipped with a visible reason
    # unless selected on a machine that really has a device.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def tuning_lib():
    """The -DDS_TUNING build for the duration of a test: tests that pin a tile / kernel family / channel-block count through
    ds_debug_* run every ops.* wrapper against libds_kernels_tuning.so (same kernel sources); the product library
    (libds_kernels.so, which has no such switch) is restored afterwards."""
    from tumblr_emotions_amd import _lib
    with _lib.tuning_library() as lib:
        yield lib
