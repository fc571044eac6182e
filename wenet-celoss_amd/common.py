"""Small host helpers with the reference's names and behaviour
(wenet/utils/common.py): IGNORE_ID :23, add_blank :56-87, log_add :268-276."""
import math
from typing import List

import torch

IGNORE_ID = -1


def add_blank(ys_pad: torch.Tensor, blank: int, ignore_id: int) -> torch.Tensor:
    """Prepend a blank column and map ignore_id to blank: (B, Lmax) -> (B, Lmax + 1)."""
    bs = ys_pad.size(0)
    _blank = torch.full((bs, 1), blank, dtype=torch.long, device=ys_pad.device)
    out = torch.cat([_blank, ys_pad], dim=1)
    return torch.where(out == ignore_id, blank, out)


def log_add(args: List[float]) -> float:
    """Stable log add of Python floats (float64)."""
    if all(a == -float("inf") for a in args):
        return -float("inf")
    a_max = max(args)
    lsp = math.log(sum(math.exp(a - a_max) for a in args))
    return a_max + lsp


def end_blank(ys_pad: torch.Tensor, blank: int, ignore_id: int) -> torch.Tensor:
    """Fork helper (wenet/utils/common.py:89-120): append a blank column, map ignore_id to blank."""
    bs = ys_pad.size(0)
    _blank = torch.full((bs, 1), blank, dtype=torch.long, device=ys_pad.device)
    out = torch.cat([ys_pad, _blank], dim=1)
    return torch.where(out == ignore_id, blank, out)


def pad_list(xs: List[torch.Tensor], pad_value: int) -> torch.Tensor:
    n = len(xs)
    max_len = max(x.size(0) for x in xs)
    pad = torch.full((n, max_len), pad_value, dtype=xs[0].dtype, device=xs[0].device)
    for i, x in enumerate(xs):
        pad[i, :x.size(0)] = x
    return pad


def add_sos_eos(ys_pad: torch.Tensor, sos: int, eos: int, ignore_id: int):
    """wenet/utils/common.py:122-165: (B, Lmax) -> ys_in (sos-prefixed, eos padded), ys_out (eos-suffixed, ignore padded)."""
    _sos = torch.tensor([sos], dtype=torch.long, device=ys_pad.device)
    _eos = torch.tensor([eos], dtype=torch.long, device=ys_pad.device)
    ys = [y[y != ignore_id] for y in ys_pad]
    ys_in = [torch.cat([_sos, y], dim=0) for y in ys]
    ys_out = [torch.cat([y, _eos], dim=0) for y in ys]
    return pad_list(ys_in, eos), pad_list(ys_out, ignore_id)


def reverse_pad_list(ys_pad: torch.Tensor, ys_lens: torch.Tensor, pad_value: float = -1.0) -> torch.Tensor:
    """wenet/utils/common.py:168-190"""
    from torch.nn.utils.rnn import pad_sequence
    return pad_sequence([torch.flip(y.int()[:i], [0]) for y, i in zip(ys_pad, ys_lens)], True, pad_value)


class LabelSmoothingLoss(torch.nn.Module):
    """KL label-smoothing loss with the reference's constructor
    (wenet/transformer/label_smoothing_loss.py); only used when an attention decoder is attached."""

    def __init__(self, size: int, padding_idx: int, smoothing: float, normalize_length: bool = False):
        super().__init__()
        self.size, self.padding_idx = size, padding_idx
        self.confidence, self.smoothing = 1.0 - smoothing, smoothing
        self.normalize_length = normalize_length

    def forward(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        assert x.size(2) == self.size
        batch_size = x.size(0)
        x = x.view(-1, self.size)
        target = target.view(-1)
        true_dist = torch.full_like(x, self.smoothing / (self.size - 1))
        ignore = target == self.padding_idx
        total = len(target) - int(ignore.sum())
        true_dist.scatter_(1, target.masked_fill(ignore, 0).unsqueeze(1), self.confidence)
        kl = torch.nn.functional.kl_div(torch.log_softmax(x, dim=1), true_dist, reduction="none")
        denom = total if self.normalize_length else batch_size
        return kl.masked_fill(ignore.unsqueeze(1), 0).sum() / denom
