"""CPU oracle for the RNN-T / CTC loss hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product package (``wenet-celoss_amd/``) never
does: its entry points fail loudly when the HIP library is missing instead of
falling back to anything here.

The arithmetic lives in plain C (``rnnt_oracle.c``, ``ctc_oracle.c``,
``rnnt_baseline.c``; each header cites the reference lines it restates and
states how it is pinned).  This module builds them with gcc into
``oracle/build/liboracle_<cpu-tag>.so`` and wraps them with ctypes/numpy.
``decode_oracle.py`` holds the pure-Python restatements of the reference's
greedy and prefix-beam decoders.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SOURCES = ("rnnt_oracle.c", "ctc_oracle.c", "rnnt_baseline.c")
_lib = None


def _cpu_tag() -> str:
    """The baseline unit is compiled -march=native, so key the build by the
    host's instruction-set flags (the build container and the GPU box differ)."""
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    h = hashlib.sha1(flags.encode())
    for s in _SOURCES + ("Makefile",):
        with open(os.path.join(_HERE, s), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def lib_path() -> str:
    return os.path.join(_HERE, "build", f"liboracle_{_cpu_tag()}.so")


def build(verbose: bool = False) -> str:
    path = lib_path()
    if not os.path.exists(path):
        tag = _cpu_tag()
        r = subprocess.run(["make", "-C", _HERE, f"TAG={tag}"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(path):
            raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(r.stdout)
    return path


def _load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        c_f = ctypes.POINTER(ctypes.c_float)
        c_d = ctypes.POINTER(ctypes.c_double)
        c_i = ctypes.POINTER(ctypes.c_int32)
        lib.wr_oracle_rnnt_f64.argtypes = [c_f, c_i, c_i, c_i] + [ctypes.c_int] * 5 + [ctypes.c_double, c_d, c_f]
        lib.wr_oracle_rnnt_f64.restype = ctypes.c_int
        lib.wr_oracle_rnnt_f32.argtypes = [c_f, c_i, c_i, c_i] + [ctypes.c_int] * 5 + [ctypes.c_float, c_f, c_f, ctypes.c_int]
        lib.wr_oracle_rnnt_f32.restype = ctypes.c_int
        lib.wr_oracle_ctc_f64.argtypes = [c_f, c_i, c_i, c_i] + [ctypes.c_int] * 5 + [c_d, c_f]
        lib.wr_oracle_ctc_f64.restype = ctypes.c_int
        lib.wr_oracle_max_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct)) if a is not None else None


def _prep(logits, targets, a_lens, b_lens):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    a_lens = np.ascontiguousarray(a_lens, dtype=np.int32)
    b_lens = np.ascontiguousarray(b_lens, dtype=np.int32)
    return logits, targets, a_lens, b_lens


def rnnt_loss_f64(logits, targets, logit_lengths, target_lengths, blank=0, clamp=-1.0, want_grad=True):
    """Checker.  logits [B,T,U+1,V] f32 -> (costs [B] f64, grad [B,T,U+1,V] f32 | None)."""
    logits, targets, ll, tl = _prep(logits, targets, logit_lengths, target_lengths)
    B, T, U1, V = logits.shape
    assert targets.shape == (B, U1 - 1) or (U1 == 1 and targets.shape[0] == B), targets.shape
    if targets.size == 0:
        targets = np.zeros((B, 1), dtype=np.int32)
    costs = np.zeros(B, dtype=np.float64)
    grad = np.empty_like(logits) if want_grad else None
    rc = _load().wr_oracle_rnnt_f64(_p(logits, ctypes.c_float), _p(targets, ctypes.c_int32),
                                    _p(ll, ctypes.c_int32), _p(tl, ctypes.c_int32),
                                    B, T, U1, V, int(blank), float(clamp),
                                    _p(costs, ctypes.c_double), _p(grad, ctypes.c_float))
    assert rc == 0
    return costs, grad


def rnnt_loss_f32(logits, targets, logit_lengths, target_lengths, blank=0, clamp=-1.0, want_grad=True,
                  nthreads=0, out_grad=None):
    """Threaded float32 port (the timed CPU baseline)."""
    logits, targets, ll, tl = _prep(logits, targets, logit_lengths, target_lengths)
    B, T, U1, V = logits.shape
    if targets.size == 0:
        targets = np.zeros((B, 1), dtype=np.int32)
    costs = np.zeros(B, dtype=np.float32)
    grad = None
    if want_grad:
        grad = out_grad if out_grad is not None else np.empty_like(logits)
    rc = _load().wr_oracle_rnnt_f32(_p(logits, ctypes.c_float), _p(targets, ctypes.c_int32),
                                    _p(ll, ctypes.c_int32), _p(tl, ctypes.c_int32),
                                    B, T, U1, V, int(blank), float(clamp),
                                    _p(costs, ctypes.c_float), _p(grad, ctypes.c_float), int(nthreads))
    assert rc == 0
    return costs, grad


def ctc_loss_f64(logits, targets, input_lengths, target_lengths, blank=0, want_grad=True):
    """Checker.  logits [B,T,V] f32 (pre-softmax) -> (nll [B] f64, grad [B,T,V] f32 | None)."""
    logits, targets, il, tl = _prep(logits, targets, input_lengths, target_lengths)
    B, T, V = logits.shape
    if targets.ndim == 1:
        targets = targets.reshape(B, -1)
    if targets.shape[1] == 0:
        targets = np.zeros((B, 1), dtype=np.int32)
    targets = np.ascontiguousarray(np.where(targets < 0, 0, targets), dtype=np.int32)
    S = targets.shape[1]
    nll = np.zeros(B, dtype=np.float64)
    grad = np.empty_like(logits) if want_grad else None
    rc = _load().wr_oracle_ctc_f64(_p(logits, ctypes.c_float), _p(targets, ctypes.c_int32),
                                   _p(il, ctypes.c_int32), _p(tl, ctypes.c_int32),
                                   B, T, S, V, int(blank), _p(nll, ctypes.c_double), _p(grad, ctypes.c_float))
    assert rc == 0
    return nll, grad


def max_threads() -> int:
    return int(_load().wr_oracle_max_threads())
