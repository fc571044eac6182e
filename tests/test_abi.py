"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, exports
every symbol include/wr_api.h declares, and rejects bad arguments before any
launch (no compute calls here -- this runs without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "wr_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wr_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from wenet_celoss_amd import _lib
    return _lib.load()


def test_header_symbols_all_exported_and_bound(lib):
    from wenet_celoss_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 6
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in wr_api.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.SIGNATURES"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib but not declared in wr_api.h"


def test_version_and_workspace(lib):
    assert lib.wr_api_version() == 1
    assert lib.wr_rnnt_workspace_bytes(0, 10, 10) == 0
    small = lib.wr_rnnt_workspace_bytes(2, 10, 5)
    big = lib.wr_rnnt_workspace_bytes(32, 1000, 151)
    assert 0 < small < big
    # lattice state is a few floats per cell, never logits-sized
    assert big < 32 * 1000 * 151 * 4 * 8


def test_bad_arguments_are_rejected_without_launch(lib):
    null = ctypes.c_void_p(None)
    rc = lib.wr_rnnt_loss_fwd(null, 0, null, null, null, 2, 4, 3, 8, 0, null, null, 0, null)
    assert rc == -1 and b"null" in lib.wr_last_error()
    rc = lib.wr_rnnt_loss_fwd(null, 0, null, null, null, 2, 4, 3, 8, 9, null, null, 0, null)
    assert rc == -1 and b"blank" in lib.wr_last_error()
    rc = lib.wr_rnnt_loss_fwd(null, 0, null, null, null, 2, 4, 600, 8, 0, null, null, 0, null)
    assert rc == -2 and b"512" in lib.wr_last_error()
    rc = lib.wr_rnnt_loss_bwd(null, 0, null, null, null, 0, 4, 3, 8, 0, -1.0, null, null, null, 0, null)
    assert rc == -1


def test_product_has_no_cpu_path():
    import torch
    import wenet_celoss_amd as w
    logits = torch.zeros(1, 2, 2, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        w.rnnt_loss(logits, torch.ones(1, 1, dtype=torch.int32), torch.tensor([2], dtype=torch.int32),
                    torch.tensor([1], dtype=torch.int32), blank=0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "wenet-celoss_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, flags=re.M), f
                assert "oracle/" not in src, f
