"""Test stand-in with the STRUCTURE of the reference's hot-word module (wenet/transformer/context_bias.py::ContextBias
in its shipped configuration: BLSTM context extractor, 'linear' context encoder, MultiHeadedAttention biasing), so that
the weights recorded in tests/golden/greedy_both_real_*.npz (the reference module's own state_dict, data only) can be
loaded on the GPU box, where the reference tree does not exist.  Parameter names and shapes follow the reference so
that `load_state_dict` takes the fixture as is; the arithmetic is restated here (not copied) and pinned by the fixture's
recorded intermediates (bias_hidden, biased encoder outputs, gate logits) in tests/test_hotword_gpu.py and, on the CPU,
tests/test_host_logic.py::test_context_bias_mirror_matches_reference_intermediates."""
import math

import torch
from torch import nn


class Attention(nn.Module):
    """Scaled dot-product attention with h heads, no mask, no dropout (attention.py:35-113,153-186)."""

    def __init__(self, heads, dim):
        super().__init__()
        self.h, self.d_k = heads, dim // heads
        self.linear_q, self.linear_k = nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.linear_v, self.linear_out = nn.Linear(dim, dim), nn.Linear(dim, dim)

    def forward(self, query, key, value):
        B = query.size(0)
        q, k, v = (lin(x).view(B, -1, self.h, self.d_k).transpose(1, 2)
                   for lin, x in ((self.linear_q, query), (self.linear_k, key), (self.linear_v, value)))
        att = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(self.d_k), dim=-1)
        return self.linear_out((att @ v).transpose(1, 2).reshape(B, -1, self.h * self.d_k)), None


class ListEncoder(nn.Module):
    """Bidirectional LSTM over each hot word; final h and c of both directions concatenated (context_bias.py:30-65)."""

    def __init__(self, vocab, dim, layers):
        super().__init__()
        self.word_embedding = nn.Embedding(vocab, dim)
        self.sen_rnn = nn.LSTM(dim, dim, num_layers=layers, batch_first=True, bidirectional=True)

    def forward(self, words, lengths):
        emb = self.word_embedding(words.clamp(min=0))
        packed = nn.utils.rnn.pack_padded_sequence(emb, lengths.cpu().long(), batch_first=True, enforce_sorted=False)
        _, (h, c) = self.sen_rnn(packed)
        return torch.cat([h[-1], h[-2], c[-1], c[-2]], dim=-1)


class ContextBiasMirror(nn.Module):
    def __init__(self, vocab, dim, layers=1, heads=2, hw_dim=8, hw_heads=2, n_labels=2):
        super().__init__()
        self.context_extractor = ListEncoder(vocab, dim, layers)
        self.context_encoder = nn.Sequential(nn.Linear(4 * dim, dim), nn.LayerNorm(dim))
        self.encoder_bias, self.predictor_bias = Attention(heads, dim), Attention(heads, dim)
        self.hw_bias = Attention(hw_heads, hw_dim)
        self.encoder_bias_combine, self.predictor_bias_combine = nn.Linear(2 * dim, dim), nn.Linear(2 * dim, dim)
        # (sic) the reference spells two of its attributes "encdoer"
        self.encdoer_bias_bias_norm, self.encdoer_bias_out_norm = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.predictor_bias_bias_norm, self.predictor_bias_out_norm = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.hw_bias_norm = nn.LayerNorm(hw_dim)
        self.hw_output_layer = nn.Linear(hw_dim, n_labels)
        self.hw_output_layer_enc, self.hw_output_layer_dec = nn.Linear(dim, hw_dim), nn.Linear(dim, hw_dim)

    def forward_bias_hidden(self, context_list, context_lengths):
        dev = self.hw_bias_norm.weight.device
        return self.context_encoder(self.context_extractor(context_list.to(dev), context_lengths).unsqueeze(0))

    def _bias(self, att, norm, combine, out_norm, hidden, x):
        hidden = hidden.expand(x.shape[0], -1, -1)
        feat = norm(att(x, hidden, hidden)[0])
        return out_norm(combine(torch.cat([x, feat], dim=-1))), feat

    def forward_encoder_bias(self, bias_hidden, encoder_out):
        return self._bias(self.encoder_bias, self.encdoer_bias_bias_norm, self.encoder_bias_combine,
                          self.encdoer_bias_out_norm, bias_hidden, encoder_out)

    def forward_predictor_bias(self, bias_hidden, predictor_out):
        return self._bias(self.predictor_bias, self.predictor_bias_bias_norm, self.predictor_bias_combine,
                          self.predictor_bias_out_norm, bias_hidden, predictor_out)

    def forward_hw_pred_both(self, h_enc_bias, h_dec_bias):
        e, d = self.hw_output_layer_enc(h_enc_bias), self.hw_output_layer_dec(h_dec_bias)
        return self.hw_output_layer(self.hw_bias_norm(self.hw_bias(d, e, e)[0]))


def from_fixture(d, device="cpu"):
    """Build the mirror from a greedy_both_real_*.npz fixture (weights under the `cb_` prefix)."""
    sd = {k[3:]: torch.tensor(d[k]) for k in d.files if k.startswith("cb_")}
    vocab, dim = sd["context_extractor.word_embedding.weight"].shape
    layers = sum(1 for k in sd if k.startswith("context_extractor.sen_rnn.weight_ih_l") and not k.endswith("_reverse"))
    m = ContextBiasMirror(vocab, dim, layers, int(d["heads"]), int(d["hw_dim"]), int(d["hw_heads"]),
                          sd["hw_output_layer.weight"].shape[0])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing, missing                 # the reference has more layers (unused here); none of ours may lack weights
    return m.to(device).eval()
