"""A small stand-in for the fork's ContextBias module (wenet/transformer/context_bias.py:159-399) with the same
method interface the greedy loops call.  It is NOT the reference module (that one cannot travel to the GPU box);
it only has to be *some* deterministic hot-word module so that the control flow of the fork's greedy variants
(gate 0/1, go-back re-decoding, which encoder stream feeds the joiner) is exercised identically on both sides:
tests/golden/make_golden.py hands this object to the reference's loops, the GPU tests hand it to ours."""
import torch
from torch import nn


class TinyBias(nn.Module):
    def __init__(self, vocab: int, enc_dim: int, pred_dim: int, ctx_dim: int = 8, seed: int = 0, gate_bias: float = 0.0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)

        def lin(i, o, scale=1.0):
            m = nn.Linear(i, o)
            with torch.no_grad():
                m.weight.copy_(torch.randint(-8, 9, m.weight.shape, generator=g).float() / 16 * scale)
                m.bias.copy_(torch.randint(-8, 9, m.bias.shape, generator=g).float() / 16 * scale)
            return m

        self.embed = nn.Embedding(vocab, ctx_dim)
        with torch.no_grad():
            self.embed.weight.copy_(torch.randint(-8, 9, self.embed.weight.shape, generator=g).float() / 8)
        self.enc_proj = lin(ctx_dim, enc_dim, 0.5)
        self.pred_proj = lin(ctx_dim, pred_dim, 0.5)
        self.enc_feat = lin(enc_dim, ctx_dim)
        self.pred_feat = lin(pred_dim, ctx_dim)
        self.gate = lin(2 * ctx_dim, 2, 2.0)
        self.gate_pred = lin(pred_dim + ctx_dim, 2, 2.0)
        with torch.no_grad():
            self.gate.bias[1] += gate_bias
            self.gate_pred.bias[1] += gate_bias

    def forward_bias_hidden(self, context_list, context_lengths):
        ids = context_list.to(self.embed.weight.device).long().clamp(min=0)
        return self.embed(ids).mean(dim=tuple(range(ids.dim()))).reshape(1, -1)          # (1, C)

    def forward_encoder_bias(self, bias_hidden, enc):
        out = enc + self.enc_proj(bias_hidden)[:, None, :]
        return out, torch.tanh(self.enc_feat(out))

    def forward_predictor_bias(self, bias_hidden, pred):
        out = pred + self.pred_proj(bias_hidden)[:, None, :]
        return out, torch.tanh(self.pred_feat(out))

    def forward_hw_pred_both(self, enc_bias_step, pred_bias_step):
        return self.gate(torch.cat([enc_bias_step, pred_bias_step], -1))                  # (1, 1, 2)

    def forward_hw_pred(self, bias_hidden, pred):
        return self.gate_pred(torch.cat([pred, bias_hidden[:, None, :].expand(-1, pred.size(1), -1)], -1))
