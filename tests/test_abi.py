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


def header_api_version():
    text = open(os.path.join(ROOT, "include", "wr_api.h")).read()
    return int(re.search(r"#define\s+WR_API_VERSION\s+(\d+)", text).group(1))


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
    from wenet_celoss_amd import _lib
    assert lib.wr_api_version() == _lib.API_VERSION == header_api_version()
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
    rc = lib.wr_rnnt_loss_fwd(null, 0, null, null, null, 2, 4, 1100, 8, 0, null, null, 0, null)
    assert rc == -2 and b"1024" in lib.wr_last_error()
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


def test_every_entry_point_rejects_bad_arguments_before_launch(lib):
    """No GPU needed: argument validation happens before any HIP call, returns a negative code and a message."""
    import ctypes
    from wenet_celoss_amd import _lib
    null = ctypes.c_void_p(None)
    # CTC: label sequences longer than the sweep supports; vocabulary wider than the gradient kernel's LDS row
    assert lib.wr_ctc_loss_fwd(null, 0, null, null, null, 2, 10, 600, 50, 0, null, null, 0, null) == -2
    assert b"511" in lib.wr_last_error()
    assert lib.wr_ctc_loss_fwd(null, 0, null, null, null, 2, 10, 5, 20000, 0, null, null, 0, null) == -2
    assert lib.wr_ctc_loss_bwd(null, 0, null, null, null, 2, 10, 5, 50, 0, null, null, null, 0, null) == -1
    assert lib.wr_ctc_workspace_bytes(2, 10, 5) > 0 and lib.wr_ctc_workspace_bytes(0, 10, 5) == 0
    # joiner: join_dim limits, missing pointers, half-specified lengths
    assert lib.wr_joint_fwd(null, null, null, null, null, null, 1, 2, 2, 516, 10, 0, null, null, 0, null) == -2
    assert b"join_dim" in lib.wr_last_error()
    assert lib.wr_joint_fwd(null, null, null, null, null, null, 1, 2, 2, 512, 10, 0, null, null, 0, null) == -1
    assert lib.wr_joint_fwd(null, null, null, null, null, null, 1, 2, 2, 512, 10, 6, null, null, 0, null) == -1
    assert b"activation" in lib.wr_last_error()
    assert lib.wr_joint_bwd_dz(null, null, null, null, null, null, 1, 2, 2, 512, 10, 0, null, null, null) == -1
    assert lib.wr_joint_bwd_dw(null, null, null, null, 1, 2, 2, 512, 10, null, null, null, 0, null) == -1
    assert lib.wr_joint_workspace_bytes(512, 5000) >= 512 * 5120 * 4
    assert lib.wr_joint_dw_workspace_bytes(512, 5000) >= 5000 * 512 * 4
    # decoder: null weights, too many lanes / layers
    w = _lib.TransducerWeights()
    assert lib.wr_decoder_workspace_bytes(ctypes.byref(w), 0, 1, 1, 1, 1) == 0
    handle = ctypes.c_void_p()
    assert lib.wr_decoder_create(null, 4, 4, 8, 8, 1, null, 0, null, ctypes.byref(handle)) == -1
    w.vocab_size, w.enc_dim, w.pred_dim, w.embed_dim, w.hidden, w.n_layers, w.join_dim = 50, 8, 8, 8, 8, 9, 16
    assert lib.wr_decoder_create(ctypes.byref(w), 4, 4, 8, 8, 1, null, 0, null, ctypes.byref(handle)) == -2
    assert b"n_layers" in lib.wr_last_error()
    assert lib.wr_decoder_destroy(null) == 0
    assert lib.wr_greedy_search(null, null, null, 1, 1, 1, 0, null, null, null) == -1
    assert lib.wr_prefix_beam_search(null, null, null, null, 1, 1, 1, 0.3, 0.7, 0, null, null, null, null, null) == -1
    assert lib.wr_predictor_step(null, null, null, null, 1, null, null, null, null) == -1
    # CTC decode modes
    assert lib.wr_ctc_greedy_search(null, null, 1, 4, 1, 0, 0, null, null, null, null, 0, null) == -1      # V must be > 1
    assert lib.wr_ctc_prefix_beam_search(null, null, 1, 4, 8, 0, 0, null, null, null, null, null, 0, null) == -1
    assert lib.wr_ctc_forced_align(null, 0, null, null, null, 1, 4, 0, 8, 0, null, null, 0, null) == -1
    assert lib.wr_tune_set(99, 1) == -1
