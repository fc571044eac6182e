"""MI355X-native CTC / RNN-T loss and transducer decode for WeNet.

Host side: thin PyTorch shims that mirror the reference's call sites
(wenet/transducer/transducer.py, wenet/transformer/ctc.py, wenet/transducer/
joint.py, wenet/transducer/search/*.py) and forward device pointers to
libwr_mi355x.so (include/wr_api.h).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .rnnt_loss import rnnt_loss, RNNTLoss  # noqa: F401
from .ctc import CTC, ctc_loss, ctc_greedy_search, ctc_prefix_beam_search, forced_align, forced_align_batch  # noqa: F401
from .joint import TransducerJoint, joint_logits  # noqa: F401
from .fused import joint_rnnt_loss  # noqa: F401
from .predictor import ConvPredictor, EmbeddingPredictor, PredictorBase, RNNPredictor  # noqa: F401
from .search.greedy_search import (basic_greedy_search, basic_greedy_search_both,  # noqa: F401
                                   basic_greedy_search_hw, edit_distance)
from .search.prefix_beam_search import PrefixBeamSearch, Sequence  # noqa: F401
from .transducer import Transducer  # noqa: F401
from .common import IGNORE_ID, add_blank, log_add  # noqa: F401

__all__ = ["rnnt_loss", "RNNTLoss", "CTC", "ctc_loss", "TransducerJoint", "joint_logits", "joint_rnnt_loss", "RNNPredictor",
           "EmbeddingPredictor", "ConvPredictor", "PredictorBase",
           "basic_greedy_search", "PrefixBeamSearch", "Sequence", "Transducer", "IGNORE_ID", "add_blank", "log_add",
           "ctc_greedy_search", "ctc_prefix_beam_search", "forced_align", "forced_align_batch"]
