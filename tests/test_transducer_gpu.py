"""End-to-end checks of wenet_celoss_amd.Transducer on the GPU with a tiny stand-in
encoder: the forward dict and every gradient agree with an independent float64
evaluation (torch autograd for the dense parts, CPU oracle for RNN-T / CTC), and
the decode wrappers return the reference's shapes."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class TinySubsampling(torch.nn.Module):
    """Carries the two integers the ASRModel exports read from `encoder.embed` (asr_model.py:542-554); no parameters,
    so the state dict of TinyEncoder -- and every fixture built on it -- is unchanged."""

    def __init__(self):
        super().__init__()
        self.subsampling_rate: int = 4
        self.right_context: int = 6


class TinyEncoder(torch.nn.Module):
    """Linear + frame mask; returns (encoder_out, mask) like wenet encoders (encoder.py forward)."""

    def __init__(self, idim, odim):
        super().__init__()
        self.proj = torch.nn.Linear(idim, odim)
        self.embed = TinySubsampling()

    def forward(self, xs, xs_lens, decoding_chunk_size=0, num_decoding_left_chunks=-1):
        T = xs.size(1)
        mask = (torch.arange(T, device=xs.device)[None, :] < xs_lens[:, None].to(xs.device)).unsqueeze(1)
        return torch.tanh(self.proj(xs)), mask


def build(V=23, E=12, P=10, J=16, H=14, ctc_w=0.3):
    import wenet_celoss_amd as w
    torch.manual_seed(1)
    return w.Transducer(V, 0, TinyEncoder(8, E), w.RNNPredictor(V, P, P, 0.0, H, 2, dropout=0.0),
                        w.TransducerJoint(V, E, P, J), ctc=w.CTC(V, E), ctc_weight=ctc_w, transducer_weight=1.0 - ctc_w,
                        hw_weight=0.0).to(DEV)


def test_forward_dict_and_gradients_match_float64():
    m = build()
    B, Tin, U = 3, 11, 4
    g = torch.Generator().manual_seed(2)
    speech = torch.randn(B, Tin, 8, generator=g).to(DEV)
    slen = torch.tensor([11, 7, 9], dtype=torch.int32, device=DEV)
    text = torch.tensor([[3, 5, 2, 9], [4, 4, -1, -1], [7, 1, 6, -1]], device=DEV)
    tlen = torch.tensor([4, 2, 3], dtype=torch.int32, device=DEV)
    out = m(speech, slen, text, tlen)
    assert set(out.keys()) == {"loss", "loss_att", "loss_ctc", "loss_rnnt", "hw_loss"}
    assert out["loss_att"] is None and out["hw_loss"] is None
    out["loss"].backward()

    # independent float64 evaluation
    md = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
    prm = {k: v.clone().requires_grad_(True) for k, v in md.items()}
    x = speech.double().cpu()
    enc = torch.tanh(x @ prm["encoder.proj.weight"].T + prm["encoder.proj.bias"])
    ys_in = torch.cat([torch.zeros(B, 1, dtype=torch.long), torch.where(text.cpu() < 0, 0, text.cpu())], 1)
    emb = prm["predictor.embed.weight"][ys_in]
    # the LSTM cell by hand (gate order i,f,g,o)
    hs = [torch.zeros(B, 14, dtype=torch.double) for _ in range(2)]
    cs = [torch.zeros(B, 14, dtype=torch.double) for _ in range(2)]
    outs = []
    for t in range(U + 1):
        inp = emb[:, t]
        for l in range(2):
            gates = inp @ prm[f"predictor.rnn.weight_ih_l{l}"].T + prm[f"predictor.rnn.bias_ih_l{l}"] + \
                hs[l] @ prm[f"predictor.rnn.weight_hh_l{l}"].T + prm[f"predictor.rnn.bias_hh_l{l}"]
            i, f, gg, o = gates.chunk(4, 1)
            cs[l] = torch.sigmoid(f) * cs[l] + torch.sigmoid(i) * torch.tanh(gg)
            hs[l] = torch.sigmoid(o) * torch.tanh(cs[l])
            inp = hs[l]
        outs.append(inp)
    pred = torch.stack(outs, 1) @ prm["predictor.projection.weight"].T + prm["predictor.projection.bias"]
    ep = enc @ prm["joint.enc_ffn.weight"].T + prm["joint.enc_ffn.bias"]
    pp = pred @ prm["joint.pred_ffn.weight"].T + prm["joint.pred_ffn.bias"]
    logits = torch.tanh(ep[:, :, None] + pp[:, None]) @ prm["joint.ffn_out.weight"].T + prm["joint.ffn_out.bias"]
    ctc_logits = enc @ prm["ctc.ctc_lo.weight"].T + prm["ctc.ctc_lo.bias"]
    # losses through the CPU oracle (values + gradients w.r.t. logits), chained into autograd
    lnp = logits.detach().float().numpy()
    ytxt = np.where(text.cpu().numpy() < 0, 0, text.cpu().numpy()).astype(np.int32)
    rc, rg = oracle.rnnt_loss_f64(lnp, ytxt, slen.cpu().numpy(), tlen.cpu().numpy())
    cn, cg = oracle.ctc_loss_f64(ctc_logits.detach().float().numpy(), ytxt, slen.cpu().numpy(), tlen.cpu().numpy())
    loss_rnnt, loss_ctc = rc.mean(), cn.sum() / B
    total = 0.7 * loss_rnnt + 0.3 * loss_ctc
    assert out["loss_rnnt"].item() == pytest.approx(loss_rnnt, rel=1e-5)
    assert out["loss_ctc"].item() == pytest.approx(loss_ctc, rel=1e-5)
    assert out["loss"].item() == pytest.approx(total, rel=1e-5)
    surrogate = (logits * torch.tensor(rg, dtype=torch.double) * (0.7 / B)).sum() + \
        (ctc_logits * torch.tensor(cg, dtype=torch.double) * (0.3 / B)).sum()
    surrogate.backward()
    # north_star bar: gradients within 1e-4 relative (fp32).  "Relative" is taken per parameter tensor against its
    # largest float64 gradient entry: single small entries are differences of large terms in fp32 and carry the
    # rounding of the terms, not of the result.
    for name, p in m.named_parameters():
        ref = prm[name].grad
        assert ref is not None, name
        scale = float(ref.abs().max())
        err = float((p.grad.double().cpu() - ref).abs().max())
        assert err <= 1e-4 * scale + 1e-9, (name, err, scale)


@pytest.mark.parametrize("kind,act", [("embedding", "gelu"), ("conv", "relu"), ("conv", "selu")])
def test_stateless_predictors_and_other_activations_train_like_float64(kind, act):
    """Transducer.forward with EmbeddingPredictor / ConvPredictor (predictor.py:203-481) and a non-tanh joiner
    (joint.py:25): the loss and every parameter gradient against a float64 evaluation of the same modules (CPU, library
    ops; the joiner by its formula; the RNN-T loss through the oracle), at the 1e-4 bar."""
    import copy
    import wenet_celoss_amd as w
    torch.manual_seed(5)
    V, E, D, J = 23, 12, 10, 16
    pred = w.EmbeddingPredictor(V, D, 0.0, 2, 2) if kind == "embedding" else w.ConvPredictor(V, D, 0.0, 2, bias=True)
    m = w.Transducer(V, 0, TinyEncoder(8, E), pred, w.TransducerJoint(V, E, D, J, activation=act), ctc=w.CTC(V, E),
                     ctc_weight=0.0, transducer_weight=1.0, hw_weight=0.0).to(DEV)
    B = 3
    g = torch.Generator().manual_seed(2)
    speech = torch.randn(B, 11, 8, generator=g).to(DEV)
    slen = torch.tensor([11, 7, 9], dtype=torch.int32, device=DEV)
    text = torch.tensor([[3, 5, 2, 9], [4, 4, -1, -1], [7, 1, 6, -1]], device=DEV)
    tlen = torch.tensor([4, 2, 3], dtype=torch.int32, device=DEV)
    out = m(speech, slen, text, tlen)
    out["loss"].backward()

    ref = copy.deepcopy(m).cpu().double()
    enc, _ = ref.encoder(speech.double().cpu(), slen.cpu())
    ys_in = torch.cat([torch.zeros(B, 1, dtype=torch.long), torch.where(text.cpu() < 0, 0, text.cpu())], 1)
    po = ref.predictor(ys_in)
    ep, pp = ref.joint.enc_ffn(enc), ref.joint.pred_ffn(po)
    logits = ref.joint.ffn_out(ref.joint.activatoin(ep[:, :, None] + pp[:, None]))
    ytxt = np.where(text.cpu().numpy() < 0, 0, text.cpu().numpy()).astype(np.int32)
    rc, rg = oracle.rnnt_loss_f64(logits.detach().float().numpy(), ytxt, slen.cpu().numpy(), tlen.cpu().numpy())
    assert out["loss"].item() == pytest.approx(rc.mean(), rel=1e-5)
    ref.zero_grad()
    (logits * torch.tensor(rg, dtype=torch.double) / B).sum().backward()
    checked = 0
    for (name, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        if q.grad is None:                       # pos_embed.bias-like parameters the forward never touches
            continue
        scale = float(q.grad.abs().max())
        assert float((p.grad.double().cpu() - q.grad).abs().max()) <= 1e-4 * scale + 1e-9, name
        checked += 1
    assert checked >= 9
    # and the decoders take the same modules
    hyps, _ = m.greedy_search(speech[:1], slen[:1])
    best, _ = m.beam_search(speech[:1], slen[:1], beam_size=3)
    assert isinstance(hyps[0], list) and isinstance(best, list)


def test_decode_wrappers_shapes_and_consistency():
    import wenet_celoss_amd as w
    m = build().eval()
    with torch.no_grad():
        m.joint.ffn_out.weight *= 6
        m.joint.ffn_out.bias[0] += 1.0
    speech = torch.randn(1, 40, 8, device=DEV)
    slen = torch.tensor([40], dtype=torch.int32, device=DEV)
    hyps, dist = m.greedy_search(speech, slen)
    assert isinstance(hyps, list) and len(hyps) == 1 and dist == 0
    hyp, score = m.beam_search(speech, slen, beam_size=4, context_list=None, context_lengths=None)   # extra kwargs ignored
    assert isinstance(hyp, list) and isinstance(score, float)
    # beam 1 with the transducer score only follows the greedy path while at most one token is emitted per frame
    b1, _ = m.beam_search(speech, slen, beam_size=1, ctc_weight=0.0, transducer_weight=1.0)
    g1, _ = m.greedy_search(speech, slen, n_steps=1)
    assert b1 == g1[0]
    # rescoring score helper: -rnnt_loss(reduction='none') per hypothesis
    enc_out, mask = m.encoder(speech, slen)
    hyps_pad = torch.tensor([hyp + [-1] * (max(len(hyp), 1) - len(hyp))] if hyp else [[-1]], device=DEV)
    if hyp:
        td = m._cal_transducer_score(enc_out, mask, torch.tensor([len(hyp)], device=DEV), hyps_pad)
        assert td.shape == (1,) and td.item() < 0      # values: test_cal_transducer_score_values_match_oracle
    # batch extension equals one-by-one
    sp = torch.randn(3, 40, 8, device=DEV)
    sl = torch.tensor([40, 25, 33], dtype=torch.int32, device=DEV)
    batch = m.greedy_search_batch(sp, sl)
    for i in range(3):
        one, _ = m.greedy_search(sp[i:i + 1, :sl[i]], sl[i:i + 1])
        assert one[0] == batch[i]
    with pytest.raises(AssertionError):
        m.greedy_search(sp, sl)                              # reference asserts batch size 1 (transducer.py:545)
    # step exports
    cache = m.forward_predictor_init_state()
    o, c2 = m.forward_predictor_step(torch.zeros(1, 1, dtype=torch.long, device=DEV), cache)
    assert o.shape == (1, 1, 10) and c2[0].shape == cache[0].shape
    js = m.forward_joint_step(enc_out[:, :1], o)
    assert js.shape == (1, 1, 1, 23)


class TinyAttnDecoder(torch.nn.Module):
    """Stand-in for the reference's BiTransformerDecoder with its call signature (decoder.py forward: memory,
    memory_mask, ys_in_pad, ys_in_lens, r_ys_in_pad, reverse_weight -> (l_x, r_x, olens)); scriptable arithmetic:
    logits = out(tanh(embed(ys) + mean_t(memory)))."""

    def __init__(self, V, E):
        super().__init__()
        self.embed = torch.nn.Embedding(V, E)
        self.out = torch.nn.Linear(E, V)
        self.right_decoder = torch.nn.Linear(E, V)       # the rescoring asserts hasattr(decoder, 'right_decoder')

    def forward(self, memory, memory_mask, ys_in_pad, ys_in_lens, r_ys_in_pad, reverse_weight: float = 0.0):
        ctx = memory.mean(1, keepdim=True)
        return (self.out(torch.tanh(self.embed(ys_in_pad) + ctx)),
                self.right_decoder(torch.tanh(self.embed(r_ys_in_pad) + ctx)), ys_in_lens)

    @torch.jit.unused
    def forward_one_step(self, memory, memory_mask, tgt, tgt_mask, cache=None):
        """decoder.py forward_one_step: (memory, memory_mask, tgt (N, i), tgt_mask (N, i, i), cache) ->
        (log-probs of the next token (N, V), cache).  Depends on the last TWO tokens so that beams differ."""
        ctx = memory.mean(1)
        prev = self.embed(tgt[:, -2]) if tgt.size(1) > 1 else torch.zeros_like(ctx)
        h = torch.tanh(self.embed(tgt[:, -1]) + 0.5 * prev + ctx)
        return torch.log_softmax(self.out(h), dim=-1), cache


def _float64_joint_logits(m, enc_out, ys_in):
    """(T,E) encoder frames, blank-prefixed label row -> float64 joiner logits (T, U+1, V), LSTM by hand."""
    prm = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
    H = prm["predictor.rnn.weight_hh_l0"].shape[1]
    L = m.predictor.rnn.num_layers
    hs = [torch.zeros(1, H, dtype=torch.double) for _ in range(L)]
    cs = [torch.zeros(1, H, dtype=torch.double) for _ in range(L)]
    outs = []
    for tok in ys_in:
        inp = prm["predictor.embed.weight"][tok][None]
        for l in range(L):
            g = inp @ prm[f"predictor.rnn.weight_ih_l{l}"].T + prm[f"predictor.rnn.bias_ih_l{l}"] + \
                hs[l] @ prm[f"predictor.rnn.weight_hh_l{l}"].T + prm[f"predictor.rnn.bias_hh_l{l}"]
            i, f, gg, o = g.chunk(4, 1)
            cs[l] = torch.sigmoid(f) * cs[l] + torch.sigmoid(i) * torch.tanh(gg)
            hs[l] = torch.sigmoid(o) * torch.tanh(cs[l])
            inp = hs[l]
        outs.append(inp[0])
    pred = torch.stack(outs) @ prm["predictor.projection.weight"].T + prm["predictor.projection.bias"]
    ep = enc_out.double().cpu() @ prm["joint.enc_ffn.weight"].T + prm["joint.enc_ffn.bias"]
    pp = pred @ prm["joint.pred_ffn.weight"].T + prm["joint.pred_ffn.bias"]
    return torch.tanh(ep[:, None] + pp[None]) @ prm["joint.ffn_out.weight"].T + prm["joint.ffn_out.bias"]


def test_cal_transducer_score_values_match_oracle():
    """transducer.py:277-302: -rnnt_loss(reduction='none') per padded hypothesis == the f64 oracle on float64 logits."""
    m = build().eval()
    torch.manual_seed(11)
    T = 17
    enc_out = torch.tanh(torch.randn(1, T, 12, device=DEV))
    hyps = [[3, 5, 2, 9, 4], [7, 1], [6, 6, 6], [8]]
    L = max(len(h) for h in hyps)
    hyps_pad = torch.tensor([h + [-1] * (L - len(h)) for h in hyps], device=DEV)
    hyps_lens = torch.tensor([len(h) for h in hyps], device=DEV)
    n = len(hyps)
    mask = torch.ones(n, 1, T, dtype=torch.bool, device=DEV)
    with torch.no_grad():
        td = m._cal_transducer_score(enc_out.repeat(n, 1, 1), mask, hyps_lens, hyps_pad)
    assert td.shape == (n,)
    for i, h in enumerate(hyps):
        logits = _float64_joint_logits(m, enc_out[0], [0] + h)
        c, _ = oracle.rnnt_loss_f64(logits.float().numpy()[None], np.array([h], np.int32), np.array([T], np.int32),
                                    np.array([len(h)], np.int32))
        assert td[i].item() == pytest.approx(-float(c[0]), rel=1e-5), (i, h)


@pytest.mark.parametrize("search_type", ["transducer", "ctc"])
@pytest.mark.parametrize("reverse_weight", [0.0, 0.3])
def test_attention_rescoring_matches_float64(search_type, reverse_weight):
    """transducer.py:379-513, both beam_search_type branches: the returned (best hyp, best score) equal a float64
    evaluation of the rescoring formula (:489-513) over the same n-best -- attention scores from a stand-in decoder
    evaluated in float64, transducer scores from the f64 oracle, beam scores as returned by the search."""
    import wenet_celoss_amd as w
    torch.manual_seed(21)
    V, E = 23, 12
    m = build(ctc_w=0.3)
    m.decoder = TinyAttnDecoder(V, E).to(DEV)
    m.reverse_weight = reverse_weight
    m.eval()
    with torch.no_grad():
        m.joint.ffn_out.weight *= 5
        m.joint.ffn_out.bias[0] += 1.5
        m.ctc.ctc_lo.weight *= 5
        m.ctc.ctc_lo.bias[0] += 1.0
    Tin, beam = 26, 4
    speech = torch.randn(1, Tin, 8, device=DEV)
    slen = torch.tensor([Tin], dtype=torch.int32, device=DEV)
    wts = dict(ctc_weight=0.2, attn_weight=0.3, transducer_weight=0.5)
    with torch.no_grad():
        hyp, score = m.transducer_attention_rescoring(speech, slen, beam, reverse_weight=reverse_weight,
                                                      search_ctc_weight=0.3, search_transducer_weight=0.7,
                                                      beam_search_type=search_type, **wts)
        # the n-best the rescoring saw (the searches are deterministic and have their own parity tests)
        if search_type == "transducer":
            nbest, enc_out = m.bs.prefix_beam_search(speech, slen, beam_size=beam, ctc_weight=0.3, transducer_weight=0.7)
            hyps, beam_score = [s.hyp[1:] for s in nbest], [s.score for s in nbest]
        else:
            nb, enc_out = m._ctc_prefix_beam_search(speech, slen, beam_size=beam)
            hyps, beam_score = [list(h[0]) for h in nb], [h[1] for h in nb]
    assert len(hyps) == beam
    dec = {k: v.detach().double().cpu() for k, v in m.decoder.state_dict().items()}
    ctx = enc_out[0].double().cpu().mean(0)
    T = enc_out.size(1)
    totals = []
    for i, h in enumerate(hyps):
        ys_in = [V - 1] + list(h)                                          # add_sos_eos: <sos> + hyp
        lp = torch.log_softmax(torch.tanh(dec["embed.weight"][ys_in] + ctx) @ dec["out.weight"].T + dec["out.bias"], -1)
        s = sum(float(lp[j, wd]) for j, wd in enumerate(h)) + float(lp[len(h), V - 1])
        if reverse_weight > 0:
            r_in = [V - 1] + list(h)[::-1]
            rlp = torch.log_softmax(torch.tanh(dec["embed.weight"][r_in] + ctx) @ dec["right_decoder.weight"].T
                                    + dec["right_decoder.bias"], -1)
            r = sum(float(rlp[len(h) - j - 1, wd]) for j, wd in enumerate(h)) + float(rlp[len(h), V - 1])
            s = s * (1 - reverse_weight) + r * reverse_weight
        logits = _float64_joint_logits(m, enc_out[0], [0] + list(h))
        if len(h) == 0:                                                    # empty hypothesis: the all-blank path
            td = float(torch.log_softmax(logits[:, 0], -1)[:, 0].sum())
        else:
            c, _ = oracle.rnnt_loss_f64(logits.float().numpy()[None], np.array([list(h)], np.int32),
                                        np.array([T], np.int32), np.array([len(h)], np.int32))
            td = -float(c[0])
        totals.append(s * wts["attn_weight"] + beam_score[i] * wts["ctc_weight"] + td * wts["transducer_weight"])
    best = int(np.argmax(totals))
    srt = sorted(totals, reverse=True)
    assert len(srt) < 2 or srt[0] - srt[1] > 1e-3, "ambiguous stand-in scenario"
    assert list(hyp) == list(hyps[best])
    assert float(score) == pytest.approx(totals[best], rel=1e-4)


def test_forward_under_autocast_matches_fp32_within_half_precision():
    """executor.py:91 runs the forward under torch.cuda.amp.autocast when --use_amp is set: the pre-join / ctc_lo
    Linear layers then emit fp16; our kernels take fp32 (joiner, CTC: inputs are cast up) or fp16 natively (RNN-T)."""
    m = build()
    g = torch.Generator().manual_seed(3)
    speech = torch.randn(2, 9, 8, generator=g).to(DEV)
    slen = torch.tensor([9, 6], dtype=torch.int32, device=DEV)
    text = torch.tensor([[3, 5, 2], [4, 1, -1]], device=DEV)
    tlen = torch.tensor([3, 2], dtype=torch.int32, device=DEV)
    ref = m(speech, slen, text, tlen)
    ref["loss"].backward()
    gref = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad()
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        out = m(speech, slen, text, tlen)
    assert torch.isfinite(out["loss"])
    out["loss"].float().backward()
    assert out["loss"].item() == pytest.approx(ref["loss"].item(), rel=5e-3)
    for n, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
        denom = gref[n].abs().max().item() + 1e-6
        assert (p.grad.float() - gref[n]).abs().max().item() / denom < 5e-2, n


def test_ddp_wrapped_training_step_single_rank_rccl():
    """Config 4 plumbing: wenet/bin/train.py:227-240 wraps the model in DistributedDataParallel(find_unused_parameters=True)
    over the nccl (= RCCL) backend.  One rank here (the box has one GPU): the ctypes-backed autograd Functions must
    work under DDP's reducer hooks and give the same gradients as the bare model."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        m = build()
        g = torch.Generator().manual_seed(4)
        speech = torch.randn(2, 10, 8, generator=g).to(DEV)
        slen = torch.tensor([10, 7], dtype=torch.int32, device=DEV)
        text = torch.tensor([[3, 5, 2], [4, 1, -1]], device=DEV)
        tlen = torch.tensor([3, 2], dtype=torch.int32, device=DEV)
        m(speech, slen, text, tlen)["loss"].backward()
        ref = {n: p.grad.clone() for n, p in m.named_parameters()}
        m.zero_grad()
        ddp = torch.nn.parallel.DistributedDataParallel(m, find_unused_parameters=True)
        with ddp.join():                                       # executor.py:48-53 uses the uneven-input join context
            out = ddp(speech, slen, text, tlen)
            out["loss"].backward()
        for n, p in m.named_parameters():
            torch.testing.assert_close(p.grad, ref[n], rtol=1e-5, atol=1e-7, msg=n)
        # gradient accumulation path (executor.py:81-86)
        with ddp.no_sync():
            ddp(speech, slen, text, tlen)["loss"].backward()
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_training_step_on_the_product_path(tmp_path):
    """BASELINE config 4 plumbing with two real ranks (wenet/bin/train.py:227-240, wenet/utils/executor.py:48-53,81-86):
    two freshly started processes wrap the Transducer in DistributedDataParallel(find_unused_parameters=True) and run
    join() / no_sync() steps on uneven shards (3 + 2 utterances) through the HIP joiner / RNN-T / CTC kernels.
    After the synchronising step every rank must hold the mean over ranks of the per-rank accumulated gradients
    (each rank's loss is its own batch mean -- the reference averages rank means); after
    rank 0's extra step, shadowed by rank 1 through join, rank 0's gradient divided by the world size."""
    import subprocess
    import sys
    from ddp_worker import shard
    port = 29700 + os.getpid() % 500
    out = str(tmp_path / "grads")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ddp_worker.py"),
                                       out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    got = [torch.load(f"{out}.rank{r}", weights_only=True) for r in range(2)]

    # single-process reference on this process's GPU
    def grads(rank, scales):
        m = build()
        for sc in scales:
            m(*shard(rank, torch.device(DEV), sc))["loss"].backward()
        return {n: (p.grad.detach().cpu().clone() if p.grad is not None else None) for n, p in m.named_parameters()}
    g0, g1 = grads(0, (1.0, 0.5)), grads(1, (1.0, 0.5))
    for n in g0:
        want = (g0[n] + g1[n]) / 2
        tol = 1e-5 * float(want.abs().max()) + 1e-9
        for r in range(2):
            assert float((got[r]["sync"][n] - want).abs().max()) <= tol, (n, r)
    e0 = grads(0, (0.25,))
    for n in e0:
        want = e0[n] / 2                                  # join() divides by the initial world size
        assert float((got[0]["extra"][n] - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-9, n
    assert "extra" not in got[1]


@pytest.mark.parametrize("amp", [False, True])
def test_two_rank_ddp_at_librispeech_shapes(tmp_path, monkeypatch, amp):
    """BASELINE config 4 rehearsed at shape (tests/ddp_shape_worker.py): two fresh ranks, a 57 M-parameter model (stand-in
    encoder with the real parameter budget + the product's predictor / joiner / CTC head at V = 5000, J = 512), LibriSpeech
    dynamic batches (4 utterances of <= 1500 fbank frames per rank and step), DistributedDataParallel(
    find_unused_parameters=True) inside join(), accum_grad 4 (three no_sync steps + one synchronising step), RCCL when each
    rank has a device of its own, gloo otherwise.  Every parameter's gradient after the synchronising step must equal the
    mean over ranks of the per-rank accumulated gradients computed in this process; the step breakdown is printed.
    amp: the same under the reference's --use_amp step (executor.py:91 autocast, bfloat16) with the joiner's 16-bit mode --
    bf16 logits, the loss's bf16 gradient, the library-GEMM backward."""
    import json
    import subprocess
    import sys
    import ddp_shape_worker as sw
    monkeypatch.setenv("WR_SHAPE_AMP", "1" if amp else "0")
    free, _ = torch.cuda.mem_get_info()
    if free < 40e9:
        pytest.skip("needs ~40 GB of free HBM (two ranks + the reference on one device)")
    port = 29200 + os.getpid() % 500
    out = str(tmp_path / "shape")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ddp_shape_worker.py"),
                                       out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    got = [torch.load(f"{out}.rank{r}", weights_only=True) for r in range(2)]
    infos = [json.loads(g["info"]) for g in got]
    for info in infos:
        print("config-4 rehearsal:", json.dumps(info))
    assert infos[0]["parameters"] > 55e6
    # single-process reference: each rank's accumulated gradient, then the mean over ranks
    dev = torch.device(DEV)
    m = sw.build_model(dev)
    params = list(m.named_parameters())
    per_rank = []
    for r in range(2):
        m.zero_grad()
        per_rank.append(sw.accumulate(m, params, r, dev))
    top = max(float(v.abs().max()) for v in per_rank[0].values())
    checked, worst = 0, 0.0
    for n in per_rank[0]:
        want = (per_rank[0][n] + per_rank[1][n]) / 2
        # two separately started processes against this one: the library GEMMs' split-K / atomics and the CTC kernel's LDS
        # float atomics sum in run-dependent order, and the stand-in encoder is 48 layers deep -- 2e-4 of the tensor's
        # largest entry (a wrong all-reduce, a missed no_sync or a wrong 1 / world_size shows at the 0.5 level)
        # (3e-3 of the tensor's largest entry + 3e-5 of the model's largest gradient entry -- worst seen 0.7 of a third of that: intermediate gradients two
        # orders of magnitude above a small tensor's own set its noise floor)
        tol = 3e-3 * float(want.abs().max()) + 3e-5 * top
        if amp:                                     # bf16 sums inside autocast (bias gradients, the encoder's GEMMs) differ
            tol *= 10                               # from run to run at the bf16 rounding level, 2^-9 of a value
        for r in range(2):
            err = float((got[r]["grads"][n] - want).abs().max())
            worst = max(worst, err / tol)
            assert err <= tol, (n, r, err, tol, top)
        checked += 1
    print("config-4 rehearsal: largest gradient entry", top, "worst error / tolerance", worst)
    assert checked >= 100


def test_amp_two_op_path_buckets_ragged_batches(monkeypatch):
    """The --use_amp configuration (joiner precision "bf16" under autocast: 16-bit logits, two ops) cuts a ragged batch
    into label-length groups as the fused node does; loss and gradients equal the one-call path to summation order."""
    import wenet_celoss_amd as w
    torch.manual_seed(3)
    V, E, P, J, H = 40, 12, 10, 16, 14
    m = w.Transducer(V, 0, TinyEncoder(8, E), w.RNNPredictor(V, P, P, 0.0, H, 2, dropout=0.0),
                     w.TransducerJoint(V, E, P, J, precision="bf16"), ctc=None, ctc_weight=0.0, transducer_weight=1.0,
                     hw_weight=0.0).to(DEV)
    B, Tin, U = 10, 70, 36
    g = torch.Generator().manual_seed(8)
    speech = torch.randn(B, Tin, 8, generator=g).to(DEV)
    slen = torch.tensor([70, 70, 68, 66, 66, 64, 60, 60, 58, 56], dtype=torch.int32, device=DEV)
    tlen = torch.tensor([36, 4, 30, 9, 36, 12, 3, 25, 7, 33], dtype=torch.int32, device=DEV)
    text = torch.randint(1, V, (B, U), generator=g)
    for i in range(B):
        text[i, int(tlen[i]):] = -1
    text = text.to(DEV)
    res = {}
    for nb in ("1", "4"):
        monkeypatch.setenv("WR_FUSED_BUCKETS", nb)
        m.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(speech, slen, text, tlen)
        out["loss"].float().backward()
        res[nb] = (out["loss_rnnt"].item(), {n: p.grad.clone() for n, p in m.named_parameters()})
    # per-utterance costs come back in bfloat16 in both; the one-call path also takes their mean in bfloat16 (ulp = 2 at
    # this magnitude), the grouped path in float32
    assert res["4"][0] == pytest.approx(res["1"][0], rel=1e-2)
    for n, gr in res["1"][1].items():
        assert float((res["4"][1][n].float() - gr.float()).abs().max()) <= 2e-2 * float(gr.float().abs().max()) + 1e-7, n


@pytest.mark.parametrize("fixture", ["transducer_wrappers.npz", "transducer_wrappers_emb.npz"])
def test_wrappers_match_the_reference_class(fixture):
    """tests/golden/transducer_wrappers.npz was produced by the reference's OWN `Transducer` class
    (wenet/transducer/transducer.py) around the stand-in encoder / attention decoder of this file and the reference's
    predictor, joiner, CTC and ContextBias (make_golden.py::gen_transducer_wrappers; its torchaudio.functional.rnnt_loss
    was stubbed with the float64 oracle).  Same weights here: the loss dictionary of `forward` (all five entries),
    `beam_search`, `transducer_attention_rescoring` (both search types, with / without the right-to-left decoder) and
    `greedy_search` with the hot-word module reproduce what the reference class returned.  `..._emb.npz`: the same through
    the reference class built around its EmbeddingPredictor and a gelu joiner."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import GOLDEN
    from context_bias_mirror import ContextBiasMirror
    import wenet_celoss_amd as w
    d = np.load(os.path.join(GOLDEN, fixture))
    sd = {k[2:]: torch.tensor(d[k]) for k in d.files if k.startswith("m_")}
    V, D = sd["predictor.embed.weight"].shape
    J = sd["joint.enc_ffn.weight"].shape[0]
    cb = ContextBiasMirror(V, D, layers=1, heads=int(d["heads"]), hw_dim=int(d["hw_dim"]), hw_heads=int(d["hw_heads"]))
    if "predictor.pos_embed.weight" in sd:
        predictor, joint = w.EmbeddingPredictor(V, D, 0.0, 2, 2, "swish"), w.TransducerJoint(V, D, D, J, activation="gelu")
    else:
        predictor = w.RNNPredictor(V, D, D, 0.0, sd["predictor.rnn.weight_hh_l0"].shape[1], 2, dropout=0.0)
        joint = w.TransducerJoint(V, D, D, J)
    m = w.Transducer(V, 0, TinyEncoder(8, D), predictor, joint,
                     attention_decoder=TinyAttnDecoder(V, D), ctc=w.CTC(V, D), context_bias=cb, ctc_weight=0.1,
                     transducer_weight=0.75, attention_weight=0.15, reverse_weight=0.3, lsm_weight=0.1, hw_weight=0.4,
                     loss_mode="both")
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing, missing                 # (the reference ContextBias has layers the decode / loss path never uses)
    m = m.to(DEV).train()          # MIOpen's LSTM backward needs training mode; every dropout in this model is 0
    T = lambda k, dt=None: torch.tensor(d[k]).to(DEV) if dt is None else torch.tensor(d[k]).to(dt).to(DEV)
    out = m(T("fwd_speech"), T("fwd_slen"), T("fwd_text"), T("fwd_tlen"), torch.tensor(d["fwd_ctx"]),
            torch.tensor(d["fwd_ctx_len"]), T("fwd_hw_label"))
    for k in ("loss", "loss_att", "loss_ctc", "loss_rnnt", "hw_loss"):
        assert out[k].item() == pytest.approx(float(d["fwd_" + k]), rel=2e-5), k
    # ... and loss.backward() through the reference class's autograd graph gave these parameter gradients
    out["loss"].backward()
    checked = 0
    top = max(float(np.abs(d[k]).max()) for k in d.files if k.startswith("grad_"))
    for n, p in m.named_parameters():
        if "grad_" + n not in d.files:
            continue
        ref = torch.tensor(d["grad_" + n])
        assert p.grad is not None, n
        # 1e-4 of the tensor's largest entry; biases in front of a LayerNorm have a zero gradient up to rounding noise,
        # hence the floor at 1e-6 of the model's largest gradient entry
        assert float((p.grad.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-6 * top, n
        checked += 1
    assert checked >= 40
    m.zero_grad()
    m.eval()
    sp, sl = T("dec_speech"), torch.tensor([d["dec_speech"].shape[1]], dtype=torch.int32, device=DEV)
    with torch.no_grad():
        hyp, score = m.beam_search(sp, sl, beam_size=4, ctc_weight=0.3, transducer_weight=0.7)
        assert list(hyp) == d["beam_hyp"].tolist() and float(score) == pytest.approx(float(d["beam_score"]), rel=1e-5)
        for k in range(int(d["n_resc"])):
            h, s = m.transducer_attention_rescoring(sp, sl, 4, reverse_weight=float(d[f"resc_{k}_rw"]), ctc_weight=0.2,
                                                    attn_weight=0.3, transducer_weight=0.5, search_ctc_weight=0.3,
                                                    search_transducer_weight=0.7, beam_search_type=str(d[f"resc_{k}_type"]))
            assert list(h) == d[f"resc_{k}_hyp"].tolist(), k
            assert float(s) == pytest.approx(float(d[f"resc_{k}_score"]), rel=1e-4), k
        gh, gd = m.greedy_search(sp, sl, n_steps=4, context_list=torch.tensor(d["fwd_ctx"]),
                                 context_lengths=torch.tensor(d["fwd_ctx_len"]), context_filter_state="on",
                                 context_decoder_labels_padded=torch.tensor(d["greedy_labels"]))
    assert gh == [d["greedy_hyp"].tolist()] and gd == float(d["greedy_dist"])


def test_asr_model_surface_matches_the_reference_class():
    """The reference's Transducer IS an ASRModel (transducer.py:20): wenet/bin/recognize.py calls `recognize`
    (--mode attention, recognize.py:259) and `attention_rescoring` (--mode attention_rescoring, :351) on it and the C++
    runtime calls the jit exports.  tests/golden/asr_surface.npz holds what the reference's own class returned for all
    of them (make_golden.py::gen_asr_surface); ours, with the same weights, reproduces it: token sequences exactly,
    scores to 1e-5 relative (the n-best of attention_rescoring comes from the HIP CTC prefix beam search)."""
    from conftest import GOLDEN
    import wenet_celoss_amd as w
    d = np.load(os.path.join(GOLDEN, "asr_surface.npz"))
    sd = {k[2:]: torch.tensor(d[k]) for k in d.files if k.startswith("m_")}
    V, D = sd["predictor.embed.weight"].shape
    m = w.Transducer(V, 0, TinyEncoder(8, D), w.RNNPredictor(V, D, D, 0.0, sd["predictor.rnn.weight_hh_l0"].shape[1], 2, dropout=0.0),
                     w.TransducerJoint(V, D, D, sd["joint.enc_ffn.weight"].shape[0]), attention_decoder=TinyAttnDecoder(V, D),
                     ctc=w.CTC(V, D), context_bias=None, ctc_weight=0.1, transducer_weight=0.75, attention_weight=0.15,
                     reverse_weight=0.3, hw_weight=0.0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    T = lambda k: torch.tensor(d[k]).to(DEV)
    with torch.no_grad():
        sp, sl = T("rec_speech"), T("rec_slen")
        for k in range(int(d["n_rec"])):
            hyps, scores = m.recognize(sp, sl, beam_size=int(d[f"rec_{k}_beam"]))
            assert hyps.cpu().tolist() == d[f"rec_{k}_hyps"].tolist(), k
            np.testing.assert_allclose(scores.cpu().numpy(), d[f"rec_{k}_scores"], rtol=1e-5)
        gh, gs = m.ctc_greedy_search(sp, sl)
        assert gh == [row[:n].tolist() for row, n in zip(d["ctcg_hyps"], d["ctcg_len"])]
        np.testing.assert_allclose(torch.as_tensor(gs[0] if isinstance(gs, tuple) else gs).cpu().numpy().reshape(-1),
                                   d["ctcg_scores"].reshape(-1), rtol=1e-5)
        sp1 = T("one_speech")
        sl1 = torch.tensor([sp1.shape[1]], dtype=torch.int32, device=DEV)
        ph, ps = m.ctc_prefix_beam_search(sp1, sl1, 4)
        assert list(ph) == d["cpb_hyp"].tolist() and float(ps) == pytest.approx(float(d["cpb_score"]), rel=1e-5)
        for k in range(int(d["n_ar"])):
            h, s = m.attention_rescoring(sp1, sl1, 4, ctc_weight=float(d[f"ar_{k}_cw"]), reverse_weight=float(d[f"ar_{k}_rw"]))
            assert list(h) == d[f"ar_{k}_hyp"].tolist(), k
            assert float(s) == pytest.approx(float(d[f"ar_{k}_score"]), rel=1e-5), k
        assert m.subsampling_rate() == int(d["subsampling_rate"]) and m.right_context() == int(d["right_context"])
        assert m.sos_symbol() == int(d["sos"]) and m.eos_symbol() == int(d["eos"])
        assert m.is_bidirectional_decoder() == bool(d["bidirectional"])
        np.testing.assert_allclose(m.ctc_activation(T("act_in")).cpu().numpy(), d["act_out"], rtol=1e-5, atol=1e-6)
        for k in range(2):
            a, b = m.forward_attention_decoder(T("fad_hyps"), T("fad_lens"), T("fad_enc"), float(d[f"fad_{k}_rw"]))
            np.testing.assert_allclose(a.cpu().numpy(), d[f"fad_{k}_out"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(b.cpu().numpy(), d[f"fad_{k}_rout"], rtol=1e-5, atol=1e-6)
