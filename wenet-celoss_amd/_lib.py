"""Loader for libwr_mi355x.so -- the C-ABI HIP library (include/wr_api.h).

There is NO fallback: if the library cannot be built or loaded, every entry
point of this package raises.  The host side only moves pointers; PyTorch is
used for device memory, streams and torch.distributed.
"""
from __future__ import annotations

import ctypes
import glob
import hashlib
import os
import subprocess
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_CSRC = os.path.join(_PKG, "csrc")
_INCLUDE = os.path.join(_ROOT, "include")
LIB_NAME = "libwr_mi355x.so"
LIB_PATH = os.path.join(_PKG, LIB_NAME)
_HASH_PATH = LIB_PATH + ".srchash"

API_VERSION = 3              # include/wr_api.h WR_API_VERSION this binding was written against
WR_F32, WR_F16, WR_BF16 = 0, 1, 2
# wr_activation codes (include/wr_api.h), keyed by the names of wenet/utils/common.py:228-242 get_activation
ACTIVATIONS = {"tanh": 0, "relu": 1, "hardtanh": 2, "selu": 3, "swish": 4, "gelu": 5}

_lock = threading.Lock()
_lib = None


def _sources():
    return sorted(glob.glob(os.path.join(_CSRC, "*.hip")) + glob.glob(os.path.join(_CSRC, "*.cpp")))


def _source_hash() -> str:
    h = hashlib.sha1()
    files = _sources() + sorted(glob.glob(os.path.join(_CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(_INCLUDE, "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH) or not os.path.exists(_HASH_PATH):
        return True
    with open(_HASH_PATH) as f:
        return f.read().strip() != _source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into wenet-celoss_amd/libwr_mi355x.so
    (hipcc cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB_PATH
    build_dir = os.path.join(_PKG, "build")
    os.makedirs(build_dir, exist_ok=True)
    # one builder at a time (several ranks of a torch.distributed launch may find the library stale together)
    import fcntl
    with open(os.path.join(build_dir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not is_stale():
            return LIB_PATH
        return _build_locked(build_dir, verbose)


def _build_locked(build_dir: str, verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + _INCLUDE, "-I" + _CSRC,
             "-Wall", "-Wno-unused-function"] + os.environ.get("WR_EXTRA_HIPCC_FLAGS", "").split()
    procs = []
    for src in _sources():
        obj = os.path.join(build_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([hipcc] + flags + ["-c", src, "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    tmp = LIB_PATH + ".tmp"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    os.replace(tmp, LIB_PATH)
    with open(_HASH_PATH, "w") as f:
        f.write(_source_hash())
    return LIB_PATH


# name -> (restype, argtypes); must list every symbol include/wr_api.h declares.
_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
SIGNATURES = {
    "wr_api_version": (_i, []),
    "wr_last_error": (ctypes.c_char_p, []),
    "wr_tune_set": (_i, [_i, _i]),
    "wr_rnnt_workspace_bytes": (_sz, [_i, _i, _i]),
    "wr_rnnt_loss_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "wr_rnnt_loss_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "wr_rnnt_export_lattice": (_i, [_vp, _sz, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "wr_ctc_workspace_bytes": (_sz, [_i, _i, _i]),
    "wr_ctc_loss_fwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "wr_ctc_loss_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "wr_joint_workspace_bytes": (_sz, [_i, _i]),
    "wr_joint_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "wr_joint_fwd_lse": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _sz, _vp]),
    "wr_joint_fwd_split_lse": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _sz, _vp]),
    "wr_rnnt_loss_fwd_from_lse": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "wr_joint_bwd_dz": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "wr_joint_split_workspace_bytes": (_sz, [_i, _i]),
    "wr_joint_fwd_split": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "wr_joint_dz_split_workspace_bytes": (_sz, [_i, _i]),
    "wr_joint_bwd_dz_split": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "wr_joint_bwd_dz_split_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "wr_joint_db_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "wr_joint_db_bf16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "wr_joint_db_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "wr_joint_dz_act": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "wr_joint_dw_split_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "wr_joint_bwd_dw_split_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "wr_joint_bwd_dw_split": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "wr_joint_dw_workspace_bytes": (_sz, [_i, _i]),
    "wr_joint_bwd_dw": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "wr_decoder_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i, _i]),
    "wr_decoder_create": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp]),
    "wr_decoder_destroy": (_i, [_vp]),
    "wr_decoder_set_graph": (_i, [_vp, _i]),
    "wr_decoder_set_lookahead": (_i, [_vp, _i]),
    "wr_greedy_search": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "wr_greedy_search_chunk": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "wr_prefix_beam_search": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "wr_hotword_workspace_bytes": (_sz, [_vp, _vp, _i]),
    "wr_decoder_attach_hotword": (_i, [_vp, _vp, _i, _vp, _sz, _vp]),
    "wr_greedy_search_hotword": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "wr_predictor_step": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "wr_ctc_align_workspace_bytes": (_sz, [_i, _i, _i]),
    "wr_ctc_forced_align": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "wr_ctc_decode_workspace_bytes": (_sz, [_i, _i, _i]),
    "wr_ctc_greedy_search": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "wr_ctc_prefix_beam_search": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}


class TransducerWeights(ctypes.Structure):
    """ctypes mirror of `wr_transducer_weights` (include/wr_api.h)."""
    _MAXL = 4
    _fields_ = [("vocab_size", ctypes.c_int32), ("enc_dim", ctypes.c_int32), ("pred_dim", ctypes.c_int32),
                ("embed_dim", ctypes.c_int32), ("hidden", ctypes.c_int32), ("n_layers", ctypes.c_int32),
                ("join_dim", ctypes.c_int32), ("activation", ctypes.c_int32),
                ("embed", ctypes.c_void_p),
                ("w_ih", ctypes.c_void_p * 4), ("w_hh", ctypes.c_void_p * 4),
                ("b_ih", ctypes.c_void_p * 4), ("b_hh", ctypes.c_void_p * 4),
                ("proj_w", ctypes.c_void_p), ("proj_b", ctypes.c_void_p),
                ("enc_ffn_w", ctypes.c_void_p), ("enc_ffn_b", ctypes.c_void_p),
                ("pred_ffn_w", ctypes.c_void_p), ("pred_ffn_b", ctypes.c_void_p),
                ("out_w", ctypes.c_void_p), ("out_b", ctypes.c_void_p),
                ("predictor_type", ctypes.c_int32), ("context_size", ctypes.c_int32), ("n_head", ctypes.c_int32),
                ("pred_activation", ctypes.c_int32), ("ln_eps", ctypes.c_float), ("embed_rows", ctypes.c_int32),
                ("pos_w", ctypes.c_void_p), ("ffn_w", ctypes.c_void_p), ("ffn_b", ctypes.c_void_p),
                ("norm_w", ctypes.c_void_p), ("norm_b", ctypes.c_void_p),
                ("conv_w", ctypes.c_void_p), ("conv_b", ctypes.c_void_p)]


class HotwordWeights(ctypes.Structure):
    """ctypes mirror of `wr_hotword_weights` (include/wr_api.h)."""
    _PTRS = ["q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b", "bias_norm_w", "bias_norm_b", "combine_w", "combine_b",
             "out_norm_w", "out_norm_b", "hw_enc_w", "hw_enc_b", "hw_v_w", "hw_v_b", "hw_o_w", "hw_o_b", "hw_norm_w",
             "hw_norm_b", "hw_out_w", "hw_out_b"]
    _fields_ = [("dim", ctypes.c_int32), ("heads", ctypes.c_int32), ("hw_dim", ctypes.c_int32), ("n_labels", ctypes.c_int32)] + \
               [(n, ctypes.c_void_p) for n in _PTRS]


def load():
    """Return the loaded library (building it first if the sources changed)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  -- makes torch's HIP runtime the one resident in the process
        try:
            path = build()
        except Exception as e:  # no silent fallback
            raise RuntimeError(f"wenet_celoss_amd: cannot build {LIB_NAME}: {e}") from e
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:
            raise RuntimeError(f"wenet_celoss_amd: cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.wr_api_version() != API_VERSION:
            raise RuntimeError(f"wenet_celoss_amd: libwr_mi355x.so API version {lib.wr_api_version()} != {API_VERSION} "
                               "(stale library or stale binding)")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().wr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what or 'libwr_mi355x'} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dtype_code(dt) -> int:
    import torch
    if dt == torch.float32:
        return WR_F32
    if dt == torch.float16:
        return WR_F16
    if dt == torch.bfloat16:
        return WR_BF16
    raise RuntimeError(f"unsupported dtype {dt}")
