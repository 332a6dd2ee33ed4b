"""The C-ABI library builds for gfx950 without a GPU, loads, and exports exactly the entry points
that include/ds_kernels.h declares (no compute calls here: there is no device in this container)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ds_kernels.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from tumblr_emotions_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tumblr_emotions_amd", "csrc"), "-j4"], check=True)
    return _lib


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 25
    dll = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), "libds_kernels.so lacks %s" % n
    assert sorted(lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.load().ds_version() >= 1


def test_errors_are_reported_not_thrown(lib):
    l = lib.load()
    assert l.ds_gather_rows(None, None, None, 1, 1, 1, 1, 1, None) == -1          # DS_ERR_ARG
    assert b"ds_gather_rows" in l.ds_last_error()
    assert l.ds_conv_set_tile(3, 1) == -1 and l.ds_conv_set_tile(0, 0) == 0


def test_product_path_has_no_cpu_fallback(lib):
    import torch
    from tumblr_emotions_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fill(torch.zeros(4), 4, 1.0)
    if not torch.cuda.is_available():
        from tumblr_emotions_amd.net import SentimentNet
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            SentimentNet(mode="text")


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tumblr_emotions_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") or f.endswith(".hip") or f.endswith(".h") or f.endswith(".cpp"):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), os.path.join(dirpath, f)
