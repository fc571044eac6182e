#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE's own Python
modules in the build container.  Nothing of the reference is copied: this
script imports it from the path given on the command line (default
/root/reference, which does not exist on the GPU box) and stores only data --
seeded inputs, small weight tensors and the outputs the reference produced --
as .npz fixtures next to this file.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [/root/reference]

Reference entry points exercised (all importable here with two in-process stubs
for modules that are absent from the image -- `typeguard`, used only for
`assert check_argument_types()`, and `turtle`, an accidental import):
  wenet/transformer/ctc.py            CTC.forward / log_softmax      -> ctc_ref_*.npz
  wenet/transducer/joint.py           TransducerJoint.forward        -> joint_ref_*.npz, joint_var_*.npz (other activations,
                                      prejoin_linear off, postjoin_linear on)
  wenet/transducer/predictor.py       RNNPredictor.forward_step ...  -> predictor_step_*.npz
                                      EmbeddingPredictor / ConvPredictor -> predictor_var_*.npz, decode_var_*.npz (greedy
                                      loop and PrefixBeamSearch over them, with non-tanh joiners)
  wenet/transducer/search/greedy_search copy.py  basic_greedy_search -> greedy_core_*.npz
  wenet/transducer/search/prefix_beam_search.py  PrefixBeamSearch    -> prefix_beam_*.npz
  wenet/utils/common.py               add_blank, log_add             -> common_ref.npz
  wenet/transducer/transducer ref.py  Transducer.reset_cache / forward_greedy_search (the TorchScript streaming
                                      exports, :541-606)             -> greedy_stream_*.npz
                                      (module-level imports of `torchaudio` and `k2.rnnt_loss` are satisfied by empty
                                      in-process stubs; the two methods touch neither)
  wenet/transducer/transducer.py      Transducer.forward / beam_search / transducer_attention_rescoring / greedy_search
                                      through the reference's own class (torchaudio.functional.rnnt_loss stubbed with
                                      this repo's float64 oracle)    -> transducer_wrappers.npz
  wenet/transformer/asr_model.py      ASRModel.recognize / attention_rescoring / ctc_* / the jit exports, as inherited by
                                      the reference Transducer class  -> asr_surface.npz
  wenet/transformer/context_bias.py   ContextBias (the real module) driven by
  wenet/transducer/search/greedy_search.py  basic_greedy_search_both -> greedy_both_real_*.npz
torchaudio.functional.rnnt_loss cannot be imported (torchaudio is absent), so no
RNN-T-loss fixture comes from the reference; rnnt_kat.npz holds the public
known-answer vector instead.
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def install_stubs():
    tg = types.ModuleType("typeguard")
    tg.check_argument_types = lambda *a, **k: True
    sys.modules.setdefault("typeguard", tg)
    tt = types.ModuleType("turtle")
    tt.forward = None
    sys.modules.setdefault("turtle", tt)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)


def sd(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def dyadic(shape, gen, scale=8, lim=1.0):
    """Values on a 2^-k grid so that small dot products are exact in fp32."""
    x = torch.randint(-int(lim * scale), int(lim * scale) + 1, shape, generator=gen).float() / scale
    return x


# ------------------------------------------------------------------- CTC --
def gen_ctc():
    from wenet.transformer.ctc import CTC
    cases = [
        dict(seed=0, B=2, T=12, D=8, V=11, S=4),
        dict(seed=1, B=4, T=40, D=16, V=32, S=8),
        dict(seed=2, B=3, T=9, D=8, V=6, S=4, repeat=True),
        dict(seed=3, B=2, T=49, D=16, V=64, S=6),          # config[0]-like: batch 2, T<=49 frames
        dict(seed=4, B=2, T=5, D=8, V=7, S=4, infeasible=True),
    ]
    for i, c in enumerate(cases):
        torch.manual_seed(c["seed"])
        ctc = CTC(c["V"], c["D"])
        hs = torch.randn(c["B"], c["T"], c["D"], requires_grad=True)
        hlens = torch.randint(min(max(2 * c["S"] + 1, 2), c["T"]), c["T"] + 1, (c["B"],), dtype=torch.int32)
        hlens = torch.clamp(hlens, max=c["T"])
        hlens[0] = c["T"]
        ys_lens = torch.randint(1, c["S"] + 1, (c["B"],), dtype=torch.int32)
        ys_lens[-1] = c["S"]
        ys = torch.randint(1, c["V"], (c["B"], c["S"]))
        if c.get("repeat"):
            ys[:, 1::2] = ys[:, 0::2][:, : ys[:, 1::2].shape[1]]
        if c.get("infeasible"):
            ys[0] = 3                       # "3 3 3 3" needs >= 7 frames
            ys_lens[0] = c["S"]
            hlens[:] = c["T"]
        for b in range(c["B"]):
            ys[b, ys_lens[b]:] = -1         # IGNORE_ID padding, as processor.padding produces
        loss = ctc(hs, hlens, ys, ys_lens)
        loss_val = loss.detach().numpy()
        grads = {}
        if torch.isfinite(loss):
            loss.backward()
            grads = dict(grad_hs=hs.grad.numpy(), grad_w=ctc.ctc_lo.weight.grad.numpy(),
                         grad_b=ctc.ctc_lo.bias.grad.numpy())
        logp = ctc.log_softmax(hs.detach()).detach().numpy()
        amax = ctc.argmax(hs.detach()).detach().numpy()
        save(f"ctc_ref_{i}", hs=hs.detach().numpy(), hlens=hlens.numpy(), ys=ys.numpy(), ys_lens=ys_lens.numpy(),
             loss=loss_val, log_softmax=logp, argmax=amax, **{"w_" + k: v for k, v in sd(ctc).items()}, **grads)


# ----------------------------------------------------------------- joint --
def gen_joint():
    from wenet.transducer.joint import TransducerJoint
    for i, (B, T, U1, E, P, J, V) in enumerate([(2, 7, 4, 16, 16, 32, 50), (1, 5, 3, 8, 12, 20, 33)]):
        torch.manual_seed(10 + i)
        joint = TransducerJoint(V, E, P, J)
        enc = torch.randn(B, T, E, requires_grad=True)
        pred = torch.randn(B, U1, P, requires_grad=True)
        out = joint(enc, pred)
        gout = torch.randn_like(out)
        out.backward(gout)
        save(f"joint_ref_{i}", enc=enc.detach().numpy(), pred=pred.detach().numpy(), out=out.detach().numpy(),
             gout=gout.numpy(), grad_enc=enc.grad.numpy(), grad_pred=pred.grad.numpy(),
             **{"w_" + k: v for k, v in sd(joint).items()},
             **{"g_" + k: p.grad.numpy() for k, p in joint.named_parameters()})


def gen_joint_variants():
    """Non-shipped constructor options of the reference joiner (joint.py:16-43): every activation of get_activation,
    prejoin_linear off, postjoin_linear on."""
    from wenet.transducer.joint import TransducerJoint
    cases = [dict(activation=a) for a in ("relu", "hardtanh", "selu", "swish", "gelu")]
    cases += [dict(activation="swish", prejoin_linear=False), dict(activation="gelu", postjoin_linear=True),
              dict(activation="tanh", prejoin_linear=False, postjoin_linear=True)]
    for i, kw in enumerate(cases):
        torch.manual_seed(40 + i)
        B, T, U1, J, V = 2, 9, 5, 24, 70
        E = P = J if not kw.get("prejoin_linear", True) or kw.get("postjoin_linear", False) else 16
        joint = TransducerJoint(V, E, P, J, **kw)
        enc = (torch.randn(B, T, E) * 1.5).requires_grad_(True)      # spread over the kinks of relu / hardtanh
        pred = (torch.randn(B, U1, P) * 1.5).requires_grad_(True)
        out = joint(enc, pred)
        gout = torch.randn_like(out)
        out.backward(gout)
        save(f"joint_var_{i}", enc=enc.detach().numpy(), pred=pred.detach().numpy(), out=out.detach().numpy(),
             gout=gout.numpy(), grad_enc=enc.grad.numpy(), grad_pred=pred.grad.numpy(),
             activation=np.array(kw["activation"]), prejoin=np.array(kw.get("prejoin_linear", True)),
             postjoin=np.array(kw.get("postjoin_linear", False)),
             **{"w_" + k: v for k, v in sd(joint).items()},
             **{"g_" + k: p.grad.numpy() for k, p in joint.named_parameters()})


# ------------------------------------------------------------- predictor --
def gen_predictor():
    from wenet.transducer.predictor import RNNPredictor
    for i, (V, E, H, O, L, N, steps) in enumerate([(40, 16, 16, 16, 2, 3, 6), (64, 8, 24, 12, 1, 2, 4)]):
        torch.manual_seed(20 + i)
        pred = RNNPredictor(V, E, O, 0.1, H, L).eval()
        toks = torch.randint(0, V, (steps, N))
        cache = pred.init_state(N, device=torch.device("cpu"))
        outs, ms, cs = [], [], []
        padding = torch.zeros(N, 1)
        padding[-1, 0] = 1.0 if i == 1 else 0.0          # one lane keeps its state (ApplyPadding)
        for s in range(steps):
            o, cache = pred.forward_step(toks[s].reshape(N, 1), padding, cache)
            outs.append(o.detach().numpy()); ms.append(cache[0].detach().numpy()); cs.append(cache[1].detach().numpy())
        # training-mode forward over the full sequence (eval => dropout off) for one sequence
        full = pred(toks[:, :1].t().contiguous()).detach().numpy()
        # cache_to_batch / batch_to_cache round trip layout
        split = pred.batch_to_cache(cache)
        back = pred.cache_to_batch(split)
        assert torch.equal(back[0], cache[0]) and torch.equal(back[1], cache[1])
        save(f"predictor_step_{i}", toks=toks.numpy(), padding=padding.numpy(), outs=np.stack(outs), m=np.stack(ms),
             c=np.stack(cs), full_first_lane=full, n_layers=np.array(L), hidden=np.array(H),
             **{"w_" + k: v for k, v in sd(pred).items()})


def gen_predictor_variants():
    """EmbeddingPredictor / ConvPredictor (predictor.py:203-481): forward_step sequences and the training forward."""
    from wenet.transducer.predictor import ConvPredictor, EmbeddingPredictor
    cases = [dict(kind="embedding", V=40, D=16, n_head=4, history=2, act="swish", bias=False, N=3, steps=7),
             dict(kind="embedding", V=30, D=24, n_head=2, history=4, act="gelu", bias=True, N=2, steps=6),
             dict(kind="conv", V=40, D=16, history=2, act="relu", bias=False, N=3, steps=7),
             dict(kind="conv", V=25, D=40, history=1, act="tanh", bias=True, N=1, steps=5)]
    for i, c in enumerate(cases):
        torch.manual_seed(60 + i)
        if c["kind"] == "embedding":
            pred = EmbeddingPredictor(c["V"], c["D"], 0.1, c["n_head"], c["history"], c["act"], c["bias"]).eval()
        else:
            pred = ConvPredictor(c["V"], c["D"], 0.1, c["history"], c["act"], c["bias"]).eval()
        with torch.no_grad():
            pred.norm.weight.add_(torch.randn(c["D"]) * 0.3)
            pred.norm.bias.add_(torch.randn(c["D"]) * 0.3)
        N, steps = c["N"], c["steps"]
        toks = torch.randint(0, c["V"], (steps, N))
        cache = pred.init_state(N, device=torch.device("cpu"))
        outs, hist = [], []
        with torch.no_grad():
            for s_ in range(steps):
                o, cache = pred.forward_step(toks[s_].reshape(N, 1), torch.zeros(N, 1), cache)
                outs.append(o.numpy()); hist.append(cache[0].numpy())
            full = pred(toks.t().contiguous()).numpy()                  # (N, steps, D): equals the step outputs
        split = pred.batch_to_cache(cache)
        assert torch.equal(pred.cache_to_batch(split)[0], cache[0])
        save(f"predictor_var_{i}", kind=np.array(c["kind"]), history=np.array(c["history"]), n_head=np.array(c.get("n_head", 0)),
             act=np.array(c["act"]), bias=np.array(c["bias"]), toks=toks.numpy(), outs=np.stack(outs), hist=np.stack(hist),
             full=full, **{"w_" + k: v for k, v in sd(pred).items()})


def build_variant_modules(seed, kind, V, E, D, J, joint_act, blank_bias, weight_scale=2.0, postjoin=False):
    from wenet.transducer.joint import TransducerJoint
    from wenet.transducer.predictor import ConvPredictor, EmbeddingPredictor, RNNPredictor
    from wenet.transformer.ctc import CTC
    g = torch.Generator().manual_seed(seed)
    if kind == "lstm_nobias":
        pred = RNNPredictor(V, D, D, 0.1, D, 2, bias=False).eval()
    else:
        pred = (EmbeddingPredictor(V, D, 0.1, 4, 2, "swish") if kind == "embedding" else ConvPredictor(V, D, 0.1, 2, "relu", True)).eval()
    joint = TransducerJoint(V, E, D, J, activation=joint_act, postjoin_linear=postjoin).eval()
    ctc = CTC(V, E).eval()
    with torch.no_grad():
        for m in (pred, joint, ctc):
            for n_, p_ in m.named_parameters():
                if n_.startswith("norm."):
                    continue                                   # LayerNorm keeps weight 1 / bias 0
                p_.copy_(dyadic(p_.shape, g, scale=16, lim=0.5) * weight_scale)
        joint.ffn_out.bias[0] += blank_bias
    return pred, joint, ctc


def gen_decode_variants():
    """The reference's greedy loop and PrefixBeamSearch over the stateless predictors and a non-tanh joiner."""
    path = os.path.join(REF, "wenet", "transducer", "search", "greedy_search copy.py")
    spec = importlib.util.spec_from_file_location("ref_greedy_core", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from wenet.transducer.search.prefix_beam_search import PrefixBeamSearch
    V, E, D, J = 64, 16, 16, 32
    cases = [dict(kind="embedding", joint_act="relu", seed=700, T=30, n_steps=64, blank_bias=6.0, beam=4),
             dict(kind="conv", joint_act="swish", seed=701, T=40, n_steps=3, blank_bias=4.0, beam=5),
             dict(kind="conv", joint_act="tanh", seed=702, T=25, n_steps=64, blank_bias=5.0, beam=8),
             # LSTM without biases, post-join Linear (needs enc_output_size == join_dim, joint.py:41), hardtanh
             dict(kind="lstm_nobias", joint_act="hardtanh", seed=703, T=30, n_steps=64, blank_bias=5.0, beam=4, postjoin=True, E=32)]
    for i, c in enumerate(cases):
        E = c.get("E", 16)
        for attempt in range(300):
            seed = c["seed"] + 1000 * attempt
            pred, joint, ctc = build_variant_modules(seed, c["kind"], V, E, D, J, c["joint_act"], c["blank_bias"],
                                                     postjoin=c.get("postjoin", False))
            model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint, context_bias=PassThroughBias())
            g = torch.Generator().manual_seed(seed + 7)
            enc = dyadic((1, c["T"], E), g, scale=8, lim=2.0)
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                hyps = mod.basic_greedy_search(model, enc, torch.tensor(c["T"]), n_steps=c["n_steps"])
            min_margin = replay_margin(pred, joint, enc, c["T"], c["n_steps"], hyps[0])
            if not (margins_ok(min_margin) and len(hyps[0]) >= 4):
                continue
            bs = PrefixBeamSearch(GivenEncoder(enc), pred, joint, ctc, 0)
            with torch.no_grad():
                beam, _ = bs.prefix_beam_search(torch.zeros(1, c["T"], 80), torch.tensor([c["T"]]), beam_size=c["beam"])
            scores = np.array([s_.score for s_ in beam], dtype=np.float64)
            if len(scores) > 1 and np.abs(np.diff(scores)).min() < 1e-3:
                continue
            break
        else:
            raise AssertionError(f"no seed with clear margins for variant case {i}")
        maxlen = max(len(s_.hyp) for s_ in beam)
        bh = np.full((len(beam), maxlen), -1, dtype=np.int64)
        for k, s_ in enumerate(beam):
            bh[k, :len(s_.hyp)] = s_.hyp
        print(f"  variant case {i}: seed {seed}, greedy {len(hyps[0])} tokens (margin {min_margin:.4f}), beam lens "
              f"{[len(s_.hyp) for s_ in beam]}")
        save(f"decode_var_{i}", kind=np.array(c["kind"]), joint_act=np.array(c["joint_act"]), enc=enc.numpy(), T=np.array(c["T"]),
             n_steps=np.array(c["n_steps"]), hyp=np.array(hyps[0], dtype=np.int64), min_margin=np.array(min_margin),
             beam=np.array(c["beam"]), beam_hyps=bh, beam_lens=np.array([len(s_.hyp) for s_ in beam]), beam_scores=scores,
             beam_hist=np.stack([s_.cache[0].numpy() for s_ in beam]), postjoin=np.array(c.get("postjoin", False)),
             **{"pred_" + k: v for k, v in sd(pred).items()}, **{"joint_" + k: v for k, v in sd(joint).items()},
             **{"ctc_" + k: v for k, v in sd(ctc).items()})


# --------------------------------------------------------- decode models --
class PassThroughBias(torch.nn.Module):
    """Stands in for ContextBias with no hot words: the upstream greedy loop calls
    forward_encoder_bias / forward_predictor_bias and expects a single tensor back
    ('greedy_search copy.py':31,39)."""

    def forward_bias_hidden(self, context_list, context_lengths):
        return None

    def forward_encoder_bias(self, bias_hidden, x):
        return x

    def forward_predictor_bias(self, bias_hidden, x):
        return x


class GivenEncoder(torch.nn.Module):
    """Returns the tensor it was constructed with, like an encoder would."""

    def __init__(self, out):
        super().__init__()
        self.out = out

    def forward(self, speech, speech_lengths, decoding_chunk_size=-1, num_decoding_left_chunks=-1):
        T = self.out.size(1)
        return self.out, torch.ones(1, 1, T, dtype=torch.bool)


def build_decode_modules(seed, V, E, P, J, H, L, blank_bias=0.0, weight_scale=1.0):
    from wenet.transducer.joint import TransducerJoint
    from wenet.transducer.predictor import RNNPredictor
    from wenet.transformer.ctc import CTC
    g = torch.Generator().manual_seed(seed)
    pred = RNNPredictor(V, P, P, 0.1, H, L).eval()
    joint = TransducerJoint(V, E, P, J).eval()
    ctc = CTC(V, E).eval()
    with torch.no_grad():
        for m in (pred, joint, ctc):
            for p in m.parameters():
                p.copy_(dyadic(p.shape, g, scale=16, lim=0.5) * weight_scale)
        joint.ffn_out.bias[0] += blank_bias
    return pred, joint, ctc


def margins_ok(min_margin):
    return min_margin > 2e-3


def gen_greedy():
    path = os.path.join(REF, "wenet", "transducer", "search", "greedy_search copy.py")
    spec = importlib.util.spec_from_file_location("ref_greedy_core", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases = [
        dict(seed=100, T=30, n_steps=64, blank_bias=12.0),                  # realistic: mostly blanks, some emissions
        dict(seed=101, T=60, n_steps=64, blank_bias=11.0),
        dict(seed=102, T=20, n_steps=1, blank_bias=-2.0),                   # emission cap reached on most frames
        dict(seed=103, T=25, n_steps=2, blank_bias=6.0),
        dict(seed=104, T=40, n_steps=64, blank_bias=12.0, two_layer=False),
        dict(seed=105, T=16, n_steps=3, blank_bias=-30.0),                  # never blank: every frame runs into the cap
        dict(seed=106, T=30, n_steps=64, blank_bias=8.0),                   # several emissions per frame
    ]
    V, E, P, J, H = 64, 16, 16, 32, 16
    for i, c in enumerate(cases):
        L = 1 if c.get("two_layer") is False else 2
        # deterministic seed search: keep the first seed whose every decision has a clear top-1/top-2 margin,
        # so that the token-exact check does not hinge on last-bit GEMM rounding
        for attempt in range(200):
            seed = c["seed"] + 1000 * attempt
            pred, joint, ctc = build_decode_modules(seed, V, E, P, J, H, L, blank_bias=c["blank_bias"], weight_scale=2.0)
            model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint, context_bias=PassThroughBias())
            g = torch.Generator().manual_seed(seed + 7)
            enc = dyadic((1, c["T"], E), g, scale=8, lim=2.0)
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                hyps = mod.basic_greedy_search(model, enc, torch.tensor(c["T"]), n_steps=c["n_steps"])
            min_margin = replay_margin(pred, joint, enc, c["T"], c["n_steps"], hyps[0])
            if margins_ok(min_margin) and len(hyps[0]) >= c.get("min_len", 3):
                break
        else:
            raise AssertionError(f"no seed with a clear margin for case {i}")
        print(f"  greedy case {i}: seed {seed}, {len(hyps[0])} tokens, min margin {min_margin:.4f}")
        save(f"greedy_core_{i}", enc=enc.numpy(), T=np.array(c["T"]), n_steps=np.array(c["n_steps"]),
             hyp=np.array(hyps[0], dtype=np.int64), min_margin=np.array(min_margin), n_layers=np.array(L),
             hidden=np.array(H), **{"pred_" + k: v for k, v in sd(pred).items()},
             **{"joint_" + k: v for k, v in sd(joint).items()})


def replay_margin(pred, joint, enc, T, n_steps, hyp):
    """Re-run the decision sequence independently and return the smallest gap
    between the best and second-best log-prob over all steps."""
    with torch.no_grad():
        cache = pred.init_state(1, device=torch.device("cpu"))
        tok = torch.zeros(1, 1, dtype=torch.long)
        padding = torch.zeros(1, 1)
        t, nblk, k, prev = 0, 0, 0, True
        out, new_cache, mm = None, None, 1e9
        while t < T:
            if prev:
                out, new_cache = pred.forward_step(tok, padding, cache)
            lp = joint(enc[:, t:t + 1], out).log_softmax(-1).flatten()
            top = lp.topk(2)
            mm = min(mm, float(top.values[0] - top.values[1]))
            a = int(top.indices[0])
            if a != 0:
                if k >= len(hyp) or hyp[k] != a:
                    return 0.0                     # an exact tie resolved differently by topk and argmax: not a usable seed
                k += 1; prev = True; nblk += 1; tok = torch.tensor([[a]]); cache = new_cache
            if a == 0 or nblk >= n_steps:
                if a == 0:
                    prev = False
                t += 1; nblk = 0
        if k != len(hyp):
            return 0.0
    return mm


def gen_beam():
    from wenet.transducer.search.prefix_beam_search import PrefixBeamSearch
    V, E, P, J, H, L = 64, 16, 16, 32, 16, 2
    cases = [
        dict(seed=200, T=25, beam=4, cw=0.3, tw=0.7, blank_bias=10.0),
        dict(seed=201, T=40, beam=8, cw=0.3, tw=0.7, blank_bias=11.0),
        dict(seed=202, T=12, beam=1, cw=0.3, tw=0.7, blank_bias=8.0),
        dict(seed=203, T=20, beam=4, cw=0.0, tw=1.0, blank_bias=10.0),
        dict(seed=204, T=20, beam=4, cw=1.0, tw=0.0001, blank_bias=10.0),
        dict(seed=205, T=30, beam=5, cw=0.3, tw=0.7, blank_bias=12.0),
        dict(seed=206, T=30, beam=8, cw=0.3, tw=0.7, blank_bias=4.0),
    ]
    for i, c in enumerate(cases):
        pred, joint, ctc = build_decode_modules(c["seed"], V, E, P, J, H, L, blank_bias=c["blank_bias"], weight_scale=2.0)
        g = torch.Generator().manual_seed(c["seed"] + 7)
        enc = dyadic((1, c["T"], E), g, scale=8, lim=2.0)
        bs = PrefixBeamSearch(GivenEncoder(enc), pred, joint, ctc, 0)
        with torch.no_grad():
            beam, enc_out = bs.prefix_beam_search(torch.zeros(1, c["T"], 80), torch.tensor([c["T"]]),
                                                  beam_size=c["beam"], ctc_weight=c["cw"], transducer_weight=c["tw"])
        maxlen = max(len(s.hyp) for s in beam)
        hyps = np.full((len(beam), maxlen), -1, dtype=np.int64)
        for k, s in enumerate(beam):
            hyps[k, :len(s.hyp)] = s.hyp
        scores = np.array([s.score for s in beam], dtype=np.float64)
        caches_m = np.stack([s.cache[0].numpy() for s in beam])
        caches_c = np.stack([s.cache[1].numpy() for s in beam])
        gaps = np.diff(scores)
        print(f"  beam case {i}: lens {[len(s.hyp) for s in beam]}, scores {np.round(scores, 3)}")
        save(f"prefix_beam_{i}", enc=enc.numpy(), T=np.array(c["T"]), beam=np.array(c["beam"]),
             ctc_weight=np.array(c["cw"]), transducer_weight=np.array(c["tw"]), hyps=hyps,
             hyp_lens=np.array([len(s.hyp) for s in beam]), scores=scores, min_score_gap=np.array(np.abs(gaps).min() if len(gaps) else 1.0),
             cache_m=caches_m, cache_c=caches_c, n_layers=np.array(L), hidden=np.array(H),
             **{"pred_" + k: v for k, v in sd(pred).items()}, **{"joint_" + k: v for k, v in sd(joint).items()},
             **{"ctc_" + k: v for k, v in sd(ctc).items()})


def gen_ctc_decode():
    """ASRModel.ctc_greedy_search / _ctc_prefix_beam_search (wenet/transformer/asr_model.py:281-409), called
    unbound with a stand-in `self` that supplies the encoder output and the reference CTC module."""
    from wenet.transformer.asr_model import ASRModel
    from wenet.transformer.ctc import CTC
    cases = [dict(seed=300, B=1, T=30, V=20, E=8, beam=4), dict(seed=301, B=1, T=60, V=64, E=16, beam=10),
             dict(seed=302, B=3, T=25, V=12, E=8, beam=3), dict(seed=303, B=1, T=12, V=5, E=4, beam=5)]
    for i, c in enumerate(cases):
        torch.manual_seed(c["seed"])
        ctc = CTC(c["V"], c["E"]).eval()
        with torch.no_grad():
            ctc.ctc_lo.weight.mul_(6.0)
            ctc.ctc_lo.bias[0] += 1.0
        enc = torch.randn(c["B"], c["T"], c["E"])
        lens = torch.randint(c["T"] // 2, c["T"] + 1, (c["B"],))
        lens[0] = c["T"]
        mask = (torch.arange(c["T"])[None, :] < lens[:, None]).unsqueeze(1)

        class Self:
            eos = c["V"] - 1

            def __init__(self, enc, mask):
                self.enc, self.mask, self.ctc = enc, mask, ctc

            def _forward_encoder(self, speech, speech_lengths, a=-1, b=-1, c=False):
                return self.enc, self.mask

        with torch.no_grad():
            st = Self(enc, mask)
            ghyps, gscores = ASRModel.ctc_greedy_search(st, torch.zeros(c["B"], c["T"], 1), lens)
            nbest = []
            for b in range(c["B"]):                      # the prefix beam search asserts batch size 1
                sb = Self(enc[b:b + 1, :int(lens[b])], mask[b:b + 1, :, :int(lens[b])])
                hyps, _ = ASRModel._ctc_prefix_beam_search(sb, torch.zeros(1, int(lens[b]), 1), lens[b:b + 1], c["beam"])
                nbest.append(hyps)
        gl = max(len(h) for h in ghyps)
        garr = np.full((c["B"], max(gl, 1)), -1, np.int64)
        for b, h in enumerate(ghyps):
            garr[b, :len(h)] = h
        nb = max(len(h) for h in nbest)
        ml = max(max((len(p) for p, _ in h), default=0) for h in nbest)
        parr = np.full((c["B"], nb, max(ml, 1)), -1, np.int64)
        plen = np.zeros((c["B"], nb), np.int64)
        psc = np.full((c["B"], nb), -np.inf)
        for b, h in enumerate(nbest):
            for k, (pref, sc) in enumerate(h):
                parr[b, k, :len(pref)] = pref
                plen[b, k] = len(pref)
                psc[b, k] = sc
        logits = (enc @ ctc.ctc_lo.weight.T + ctc.ctc_lo.bias).detach().numpy()
        save(f"ctc_decode_{i}", logits=logits, lens=lens.numpy(), beam=np.array(c["beam"]),
             greedy=garr, greedy_lens=np.array([len(h) for h in ghyps]), greedy_scores=gscores.values.squeeze(-1).numpy(),
             nbest=parr, nbest_lens=plen, nbest_scores=psc, nbest_n=np.array([len(h) for h in nbest]))
    # the reference's own known-answer test (runtime/core/test/ctc_prefix_beam_search_test.cc:30-73): data only
    probs = np.array([[0.25, 0.40, 0.35], [0.40, 0.35, 0.25], [0.10, 0.50, 0.40]], np.float32)
    save("ctc_prefix_kat", probs=probs, beam=np.array(3), nbest=np.array([[2, 1], [1, 2], [1, -1]]),
         nbest_lens=np.array([2, 2, 1]), likelihood=np.array([0.2185, 0.1550, 0.1525], np.float32))


def gen_ctc_align():
    """forced_align (wenet/utils/ctc_util.py:27-83) on seeded log-posteriors."""
    from wenet.utils.ctc_util import forced_align
    for i, (seed, T, V, L, rep) in enumerate([(400, 12, 6, 3, False), (401, 40, 20, 9, True), (402, 25, 8, 12, False),
                                              (403, 60, 30, 20, True)]):
        g = torch.Generator().manual_seed(seed)
        lp = torch.log_softmax(torch.randn(T, V, generator=g) * 2, -1)
        y = torch.randint(1, V, (L,), generator=g)
        if rep:
            y[1::3] = y[0::3][: len(y[1::3])]               # repeated neighbours: the s-2 skip must be blocked
        ali = forced_align(lp, y)
        save(f"ctc_align_{i}", ctc_probs=lp.numpy(), y=y.numpy(), alignment=np.array([int(a) for a in ali], np.int64))


def gen_greedy_fork():
    """The fork's hot-word greedy variants, wenet/transducer/search/greedy_search.py:34-176 (`basic_greedy_search`,
    loss_mode 'pred') and :297-430 (`basic_greedy_search_both`, the default loss_mode 'both'), context filter on
    and off, run with the reference's predictor / joiner and the stand-in hot-word module tests/bias_stub.py."""
    sys.path.insert(0, os.path.dirname(HERE))
    from bias_stub import TinyBias
    from wenet.transducer.search import greedy_search as gs
    V, E, P, J, H = 64, 16, 16, 32, 16
    cases = [dict(seed=500, T=30, mode="both", filt="on", gb=0.0), dict(seed=501, T=40, mode="both", filt="off", gb=0.0),
             dict(seed=502, T=30, mode="pred", filt="on", gb=0.0), dict(seed=503, T=25, mode="pred", filt="off", gb=0.0),
             dict(seed=504, T=50, mode="both", filt="on", gb=-0.5), dict(seed=505, T=50, mode="pred", filt="on", gb=0.5)]
    for i, c in enumerate(cases):
        for attempt in range(400):
            seed = c["seed"] + 1000 * attempt
            pred, joint, _ = build_decode_modules(seed, V, E, P, J, H, 2, blank_bias=10.0, weight_scale=2.0)
            bias = TinyBias(V, E, P, seed=seed, gate_bias=c["gb"]).eval()
            model = types.SimpleNamespace(blank=0, predictor=pred, joint=joint, context_bias=bias)
            g = torch.Generator().manual_seed(seed + 7)
            enc = dyadic((1, c["T"], E), g, scale=8, lim=2.0)
            ctx = torch.randint(1, V, (3, 4), generator=g)
            ctx_len = torch.tensor([4, 4, 4], dtype=torch.int32)
            labels = torch.randint(0, 2, (1, 12), generator=g)
            fn = gs.basic_greedy_search_both if c["mode"] == "both" else gs.basic_greedy_search
            try:
                with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                    out = fn(model, enc, torch.tensor(c["T"]), ctx, ctx_len, n_steps=64,
                             context_filter_state=c["filt"], context_decoder_labels_padded=labels)
            except IndexError:
                continue                    # the reference's own go-back bookkeeping can pop from an empty list
            hyps, dist = out[0], out[1]
            trace = out[2] if len(out) > 2 else None
            if len(hyps[0]) >= 4 and (c["filt"] == "off" or attempt > 50 or dist != len(labels[0])):
                break
        else:
            raise AssertionError(f"no usable seed for fork greedy case {i}")
        print(f"  fork greedy case {i}: seed {seed}, {len(hyps[0])} tokens, dist {dist}")
        save(f"greedy_fork_{i}", enc=enc.numpy(), T=np.array(c["T"]), mode=np.array(c["mode"]), filt=np.array(c["filt"]),
             gate_bias=np.array(c["gb"]), seed=np.array(seed), ctx=ctx.numpy(), ctx_len=ctx_len.numpy(), labels=labels.numpy(),
             hyp=np.array(hyps[0], np.int64), dist=np.array(float(dist)),
             trace=np.array(trace if trace is not None else [], np.int64), n_layers=np.array(2), hidden=np.array(H),
             **{"pred_" + k: v for k, v in sd(pred).items()}, **{"joint_" + k: v for k, v in sd(joint).items()})


class MarginJoint(torch.nn.Module):
    """Wraps a reference joiner; records every decision (argmax) and the smallest top-1 / top-2 log-prob gap."""

    def __init__(self, joint):
        super().__init__()
        self.joint = joint
        self.min_margin = 1e9
        self.decisions = []

    def forward(self, enc, pred):
        out = self.joint(enc, pred)
        top = out.log_softmax(-1).flatten().topk(2)
        self.min_margin = min(self.min_margin, float(top.values[0] - top.values[1]))
        self.decisions.append(int(top.indices[0]))
        return out


def per_frame_counts(decisions, n_steps, blank=0):
    """Tokens emitted on each frame, derived from the decision sequence of the loop (a frame ends on a blank or
    when n_steps tokens were emitted on it)."""
    counts, n = [], 0
    for k in decisions:
        if k != blank:
            n += 1
        if k == blank or n >= n_steps:
            counts.append(n)
            n = 0
    return counts


def gen_greedy_stream():
    """The stateful streaming exports of "wenet/transducer/transducer ref.py":541-606 (reset_cache /
    forward_greedy_search, consumed by runtime/core/decoder/torch_asr_model.cc:126,313), called unbound on a
    stand-in `self` that carries the reference predictor / joiner.  The file imports torchaudio and k2.rnnt_loss at
    module level; neither is used by these two methods, so empty stub modules satisfy the import.
    reset_cache reads `self.predictor.output_size`, which this fork's RNNPredictor does not define (predictor.py:58-90);
    the stand-in sets it to the projection width.  The reference's per-chunk token buffer holds one token per chunk
    frame (:565) and overflows (IndexError) beyond that; the cases stay inside it."""
    for name in ("torchaudio", "k2", "k2.rnnt_loss"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["k2.rnnt_loss"].rnnt_loss_simple = None
    sys.modules["k2"].rnnt_loss = sys.modules["k2.rnnt_loss"]
    path = os.path.join(REF, "wenet", "transducer", "transducer ref.py")
    spec = importlib.util.spec_from_file_location("ref_transducer_ref", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    T_cls = mod.Transducer
    V, E, P, J, H = 64, 16, 16, 32, 16

    def run(pred, joint, enc, sizes, n_steps):
        mj = MarginJoint(joint)
        st = types.SimpleNamespace(predictor=pred, joint=mj, blank=0)
        outs, a = [], 0
        with torch.no_grad():
            T_cls.reset_cache(st)
            for n in sizes:
                outs.append(T_cls.forward_greedy_search(st, enc[:, a:a + n], torch.tensor(n), n_steps))
                a += n
        return outs, mj

    cases = [
        dict(seed=600, T=48, n_steps=64, blank_bias=11.0, chunks="even16"),
        dict(seed=601, T=40, n_steps=64, blank_bias=9.0, chunks="after_nonblank"),   # every boundary right after an emission
        dict(seed=602, T=24, n_steps=2, blank_bias=6.0, chunks=[5, 7, 12], want_cap=True),   # cap reached inside a chunk
        dict(seed=603, T=36, n_steps=64, blank_bias=9.0, chunks=[1, 1, 2, 16, 16]),     # tiny chunks
        dict(seed=604, T=30, n_steps=1, blank_bias=-30.0, chunks=[10, 10, 10]),         # never blank: cap on every frame
    ]
    for i, c in enumerate(cases):
        for attempt in range(600):
            seed = c["seed"] + 1000 * attempt
            pred, joint, _ = build_decode_modules(seed, V, E, P, J, H, 2, blank_bias=c["blank_bias"], weight_scale=2.0)
            pred.output_size = P
            g = torch.Generator().manual_seed(seed + 7)
            enc = dyadic((1, c["T"], E), g, scale=8, lim=2.0)
            try:
                _, off = run(pred, joint, enc, [c["T"]], c["n_steps"])          # offline pass: one chunk
            except IndexError:
                continue
            pf = per_frame_counts(off.decisions, c["n_steps"])
            assert len(pf) == c["T"]
            if c["chunks"] == "even16":
                sizes = [16] * (c["T"] // 16)
            elif c["chunks"] == "after_nonblank":
                cuts = [t + 1 for t in range(c["T"] - 1) if pf[t] > 0][:4]
                if len(cuts) < 3:
                    continue
                edges = [0] + cuts + [c["T"]]
                sizes = [b - a for a, b in zip(edges[:-1], edges[1:])]
            else:
                sizes = list(c["chunks"])
            assert sum(sizes) == c["T"]
            if c.get("want_cap") and not any(n >= c["n_steps"] for n in pf):
                continue
            try:
                outs, mj = run(pred, joint, enc, sizes, c["n_steps"])
            except IndexError:
                continue
            nonempty = sum(1 for o in outs if o)
            if margins_ok(mj.min_margin) and nonempty >= min(3, len(sizes)):
                break
        else:
            raise AssertionError(f"no usable seed for streaming case {i}")
        print(f"  streaming case {i}: seed {seed}, chunks {sizes}, tokens per chunk {[len(o) for o in outs]}, "
              f"min margin {mj.min_margin:.4f}")
        flat = np.array([t for o in outs for t in o], np.int64)
        save(f"greedy_stream_{i}", enc=enc.numpy(), T=np.array(c["T"]), n_steps=np.array(c["n_steps"]),
             chunk_sizes=np.array(sizes, np.int64), chunk_tokens=flat,
             chunk_token_counts=np.array([len(o) for o in outs], np.int64), min_margin=np.array(mj.min_margin),
             n_layers=np.array(2), hidden=np.array(H), **{"pred_" + k: v for k, v in sd(pred).items()},
             **{"joint_" + k: v for k, v in sd(joint).items()})


def gen_greedy_both_real():
    """The fork's default decode path with the REAL hot-word module: wenet/transducer/search/greedy_search.py:297-430
    (`basic_greedy_search_both`) driving wenet/transformer/context_bias.py::ContextBias (BLSTM extractor, 'linear'
    context encoder, MultiHeadedAttention biasing, hot-word classifier), context filter on and off.  The module's
    weights travel as data; so do the loop-invariant tensors it computes before the loop (bias_hidden for the real and
    the empty list, the two biased encoder outputs and the encoder-side bias feature), which pin the GPU tests'
    mirror module (tests/context_bias_mirror.py).  The gate trace is the list the reference prints last
    (greedy_search.py:428-429)."""
    import ast
    from wenet.transformer.context_bias import ContextBias
    from wenet.transducer.search import greedy_search as gs
    V, E, P, J, H = 64, 16, 16, 32, 16
    HW = 8
    cases = [dict(seed=700, T=40, filt="on", n_ctx=3, n_steps=64), dict(seed=701, T=40, filt="off", n_ctx=3, n_steps=64),
             dict(seed=702, T=60, filt="on", n_ctx=5, n_steps=64), dict(seed=703, T=50, filt="on", n_ctx=2, n_steps=2, blank_bias=4.0),
             dict(seed=704, T=70, filt="on", n_ctx=4, n_steps=64), dict(seed=705, T=30, filt="on", n_ctx=2, n_steps=64)]
    for i, c in enumerate(cases):
        for attempt in range(300):
            seed = c["seed"] + 1000 * attempt
            pred, joint, _ = build_decode_modules(seed, V, E, P, J, H, 2, blank_bias=c.get("blank_bias", 10.0), weight_scale=2.0)
            torch.manual_seed(seed)
            cb = ContextBias(input_size=E, output_size=E, vocab_size=V, embedding_size=E, num_layers=1, attention_heads=2,
                             bias_encoder_type="linear", context_extractor="BLSTM", unified_hw_odim=HW,
                             unified_hw_heads=2).eval()
            with torch.no_grad():           # make the 2-class gate flip between frames: spread its input projection
                cb.hw_output_layer_enc.weight.mul_(6.0)
                cb.hw_output_layer.weight.mul_(4.0)
            mj = MarginJoint(joint)
            model = types.SimpleNamespace(blank=0, predictor=pred, joint=mj, context_bias=cb)
            g = torch.Generator().manual_seed(seed + 7)
            enc = dyadic((1, c["T"], E), g, scale=8, lim=2.0)
            ctx = torch.randint(1, V, (c["n_ctx"], 4), generator=g)
            ctx_len = torch.randint(2, 5, (c["n_ctx"],), generator=g).to(torch.int32)
            ctx[0, 0], ctx_len[0] = 0, 1                               # row 0 = [0]: the "no hot word" entry the data
            for r in range(c["n_ctx"]):                                # pipeline always puts first (processor.py:763-804)
                ctx[r, ctx_len[r]:] = -1                               # IGNORE_ID padding as processor.padding produces
            labels = torch.randint(0, 2, (1, 12), generator=g)
            with torch.no_grad():           # centre the gate: shift its bias so that about half of the frames answer 1
                hid = cb.forward_bias_hidden(ctx, ctx_len)
                _, feat = cb.forward_encoder_bias(hid, enc)
                gl = cb.forward_hw_pred_both(feat.transpose(0, 1), torch.zeros(c["T"], 1, E))[:, 0, :]
                dsort = (gl[:, 0] - gl[:, 1]).sort().values
                shift = float((dsort[c["T"] // 2 - 1] + dsort[c["T"] // 2]) / 2)
                cb.hw_output_layer.bias[1] += shift
                if float((dsort - shift).abs().min()) < 1e-3:
                    continue                # a frame whose two gate logits nearly tie
            buf = io.StringIO()
            try:
                with torch.no_grad(), contextlib.redirect_stdout(buf):
                    hyps, dist = gs.basic_greedy_search_both(model, enc, torch.tensor(c["T"]), ctx, ctx_len,
                                                             n_steps=c["n_steps"], context_filter_state=c["filt"],
                                                             context_decoder_labels_padded=labels)
            except IndexError:
                continue
            trace = ast.literal_eval(buf.getvalue().strip().splitlines()[-1])
            n0, n1 = trace.count(0), trace.count(1)
            # gate margins: the reference prints the 2-class gate logits of every predictor step
            want_both = c["filt"] == "on"
            if len(hyps[0]) < 5 or len(hyps[0]) > 4 * c["T"] or not margins_ok(mj.min_margin) or \
                    (want_both and (n0 < 2 or n1 < 2)):
                continue
            if c["n_steps"] == 64 and len(hyps[0]) > 2 * c["T"]:
                continue                    # a frame ran into the 64-token cap: keep these cases speech-like
            # re-run to count go-backs (a 1 right after a 0 in the reference's own trace cannot occur with the filter on:
            # the 0 is popped) -- use the number of joiner decisions beyond tokens + frames as the witness
            break
        else:
            raise AssertionError(f"no usable seed for real-ContextBias case {i}")
        with torch.no_grad():
            hidden = cb.forward_bias_hidden(ctx, ctx_len)
            hidden_empty = cb.forward_bias_hidden(torch.zeros((1, 1), dtype=torch.int), ctx_len[0].unsqueeze(0))
            enc_hot, enc_hot_feat = cb.forward_encoder_bias(hidden, enc)
            enc_cold, _ = cb.forward_encoder_bias(hidden_empty, enc.clone())
            gate_logits = cb.forward_hw_pred_both(enc_hot_feat.transpose(0, 1), torch.zeros(c["T"], 1, E))[:, 0, :]
        gmargin = float((gate_logits[:, 0] - gate_logits[:, 1]).abs().min())
        print(f"  real-bias case {i}: seed {seed}, {len(hyps[0])} tokens, dist {dist}, trace zeros/ones {n0}/{n1}, "
              f"{len(mj.decisions)} joiner decisions for {c['T']} frames, min margin {mj.min_margin:.4f}, gate margin {gmargin:.4f}")
        save(f"greedy_both_real_{i}", enc=enc.numpy(), T=np.array(c["T"]), filt=np.array(c["filt"]),
             n_steps=np.array(c["n_steps"]), seed=np.array(seed), ctx=ctx.numpy(), ctx_len=ctx_len.numpy(), labels=labels.numpy(),
             hyp=np.array(hyps[0], np.int64), dist=np.array(float(dist)), trace=np.array(trace, np.int64),
             n_decisions=np.array(len(mj.decisions)), min_margin=np.array(mj.min_margin), gate_margin=np.array(gmargin),
             hidden=hidden.numpy(), hidden_empty=hidden_empty.numpy(), enc_hot=enc_hot.numpy(),
             enc_hot_feat=enc_hot_feat.numpy(), enc_cold=enc_cold.numpy(), gate_logits=gate_logits.numpy(),
             n_layers=np.array(2), hidden_size=np.array(H), heads=np.array(2), hw_dim=np.array(HW), hw_heads=np.array(2),
             **{"pred_" + k: v for k, v in sd(pred).items()}, **{"joint_" + k: v for k, v in sd(joint).items()},
             **{"cb_" + k: v for k, v in sd(cb).items()})


def gen_transducer_wrappers():
    """The reference's own `Transducer` class (wenet/transducer/transducer.py:20-629), instantiated around stand-in
    encoder / attention-decoder modules (tests/test_transducer_gpu.py: TinyEncoder, TinyAttnDecoder) and the reference's
    own RNNPredictor, TransducerJoint, CTC and ContextBias, and driven through `forward` (:79-270), `beam_search`
    (:332-377), `transducer_attention_rescoring` (:379-513, both beam_search_type branches, with and without the
    right-to-left decoder) and `greedy_search` (:515-598).  The module imports torchaudio (:4), which the image lacks: an
    in-process stub provides `torchaudio.functional.rnnt_loss` from THIS repo's float64 oracle (oracle/rnnt_oracle.c),
    so the RNN-T numbers inside these fixtures are the oracle's (that leg stays "parity unpinned"); everything around
    them -- label preparation, the loss dictionary and its weights, hot-word loss, n-best padding, score combination,
    arg-max -- is the reference's own code."""
    env = _reference_transducer_env()
    for variant in ("", "_emb"):
        _gen_transducer_wrappers_variant(variant, env)


def _reference_transducer_env():
    """Import the reference's Transducer class (see gen_transducer_wrappers for the torchaudio stub) and the stand-in
    encoder / attention decoder; returns the names the generators need."""
    root = os.path.dirname(os.path.dirname(HERE))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.dirname(HERE))
    import oracle
    from test_transducer_gpu import TinyAttnDecoder, TinyEncoder

    class OracleRnnt(torch.autograd.Function):
        @staticmethod
        def forward(ctx, logits, targets, logit_lengths, target_lengths, blank):
            c, gr = oracle.rnnt_loss_f64(logits.detach().float().numpy(), targets.numpy().astype(np.int32),
                                         logit_lengths.numpy().astype(np.int32), target_lengths.numpy().astype(np.int32),
                                         blank=blank)
            ctx.save_for_backward(torch.tensor(gr, dtype=torch.float32))
            return torch.tensor(c, dtype=torch.float32)

        @staticmethod
        def backward(ctx, g):
            return ctx.saved_tensors[0] * g[:, None, None, None], None, None, None, None

    def rnnt_loss(logits, targets, logit_lengths, target_lengths, blank=-1, clamp=-1, reduction="mean"):
        c = OracleRnnt.apply(logits, targets, logit_lengths, target_lengths, blank if blank >= 0 else logits.shape[-1] + blank)
        return c.mean() if reduction == "mean" else c.sum() if reduction == "sum" else c
    ta = sys.modules.setdefault("torchaudio", types.ModuleType("torchaudio"))
    ta.functional = types.ModuleType("torchaudio.functional")
    ta.functional.rnnt_loss = rnnt_loss
    sys.modules["torchaudio.functional"] = ta.functional
    from wenet.transducer.transducer import Transducer
    from wenet.transducer.joint import TransducerJoint
    from wenet.transducer.predictor import EmbeddingPredictor, RNNPredictor
    from wenet.transformer.context_bias import ContextBias
    from wenet.transformer.ctc import CTC
    return dict(locals())


def gen_asr_surface():
    """The entry points the reference's Transducer INHERITS from ASRModel (wenet/transformer/asr_model.py) and that
    wenet/bin/recognize.py / the C++ runtime call on it: `recognize` (:175-279, --mode attention), `attention_rescoring`
    (:443-540, --mode attention_rescoring), `ctc_greedy_search` (:281-324), `ctc_prefix_beam_search` (:411-440) and the
    exports `subsampling_rate`, `right_context`, `sos_symbol`, `eos_symbol`, `ctc_activation`,
    `is_bidirectional_decoder`, `forward_attention_decoder` (:542-720) -- run on the reference's own class built around
    the stand-in encoder / attention decoder of tests/test_transducer_gpu.py -> asr_surface.npz."""
    env = _reference_transducer_env()
    Transducer, TransducerJoint, RNNPredictor, CTC = (env[k] for k in ("Transducer", "TransducerJoint", "RNNPredictor", "CTC"))
    TinyEncoder, TinyAttnDecoder = env["TinyEncoder"], env["TinyAttnDecoder"]
    V, D, J, H = 23, 12, 16, 14
    torch.manual_seed(51)
    m = Transducer(V, 0, TinyEncoder(8, D), RNNPredictor(V, D, D, 0.0, H, 2, dropout=0.0), TransducerJoint(V, D, D, J),
                   attention_decoder=TinyAttnDecoder(V, D), ctc=CTC(V, D), context_bias=None, ctc_weight=0.1,
                   transducer_weight=0.75, attention_weight=0.15, reverse_weight=0.3, hw_weight=0.0).eval()
    with torch.no_grad():
        m.ctc.ctc_lo.weight *= 5
        m.ctc.ctc_lo.bias[0] += 1.0
        m.decoder.out.weight *= 4
        m.decoder.out.bias[V - 1] += 1.0            # <eos> likely enough that some beams finish early
    g = torch.Generator().manual_seed(52)
    out = {}
    with torch.no_grad():
        # recognize: batch of 3 with ragged lengths, two beam sizes
        sp = torch.randn(3, 14, 8, generator=g)
        sl = torch.tensor([14, 9, 12], dtype=torch.int32)
        out.update(rec_speech=sp.numpy(), rec_slen=sl.numpy())
        for k, beam in enumerate((1, 4)):
            hyps, scores = m.recognize(sp, sl, beam_size=beam)
            out[f"rec_{k}_beam"] = np.array(beam); out[f"rec_{k}_hyps"] = hyps.numpy(); out[f"rec_{k}_scores"] = scores.numpy()
            print(f"  recognize beam {beam}:", hyps.tolist(), scores.tolist())
        out["n_rec"] = np.array(2)
        # ctc_greedy_search (batched) / ctc_prefix_beam_search / attention_rescoring (batch 1)
        gh, gs = m.ctc_greedy_search(sp, sl)
        out["ctcg_len"] = np.array([len(h) for h in gh]); out["ctcg_hyps"] = np.array([h + [-1] * (14 - len(h)) for h in gh])
        out["ctcg_scores"] = gs.values.numpy() if hasattr(gs, "values") else np.asarray(gs[0])
        sp1 = torch.randn(1, 21, 8, generator=g)
        sl1 = torch.tensor([21], dtype=torch.int32)
        out.update(one_speech=sp1.numpy())
        ph, ps = m.ctc_prefix_beam_search(sp1, sl1, 4)
        out.update(cpb_hyp=np.array(list(ph), np.int64), cpb_score=np.array(float(ps)))
        print("  ctc_prefix_beam_search:", list(ph), float(ps))
        k = 0
        for rw in (0.0, 0.3):
            for cw in (0.0, 0.5):
                h, s = m.attention_rescoring(sp1, sl1, 4, ctc_weight=cw, reverse_weight=rw)
                out[f"ar_{k}_rw"] = np.array(rw); out[f"ar_{k}_cw"] = np.array(cw)
                out[f"ar_{k}_hyp"] = np.array(list(h), np.int64); out[f"ar_{k}_score"] = np.array(float(s))
                print(f"  attention_rescoring rw={rw} ctc_weight={cw}:", list(h), float(s))
                k += 1
        out["n_ar"] = np.array(k)
        # exports
        out.update(subsampling_rate=np.array(m.subsampling_rate()), right_context=np.array(m.right_context()),
                   sos=np.array(m.sos_symbol()), eos=np.array(m.eos_symbol()),
                   bidirectional=np.array(m.is_bidirectional_decoder()))
        xs = torch.randn(2, 5, D, generator=g)
        out.update(act_in=xs.numpy(), act_out=m.ctc_activation(xs).numpy())
        hy = torch.tensor([[V - 1, 3, 4, 5, 9], [V - 1, 7, 2, V - 1, V - 1], [V - 1, 6, V - 1, V - 1, V - 1]])
        hl = torch.tensor([5, 3, 2])
        eo = torch.randn(1, 6, D, generator=g)
        for k, rw in enumerate((0.0, 0.3)):
            a, b = m.forward_attention_decoder(hy, hl, eo, rw)
            out[f"fad_{k}_rw"] = np.array(rw); out[f"fad_{k}_out"] = a.numpy(); out[f"fad_{k}_rout"] = b.numpy()
        out.update(fad_hyps=hy.numpy(), fad_lens=hl.numpy(), fad_enc=eo.numpy())
    save("asr_surface", **out, **{"m_" + k: v for k, v in sd(m).items()})


def _gen_transducer_wrappers_variant(variant, env):
    """variant "": the shipped module types (RNNPredictor, tanh joiner); "_emb": EmbeddingPredictor + gelu joiner."""
    Transducer, TransducerJoint, RNNPredictor, EmbeddingPredictor = (env[k] for k in
                                                                     ("Transducer", "TransducerJoint", "RNNPredictor", "EmbeddingPredictor"))
    ContextBias, CTC, TinyEncoder, TinyAttnDecoder = (env[k] for k in ("ContextBias", "CTC", "TinyEncoder", "TinyAttnDecoder"))
    V, D, J, H, HW = 23, 12, 16, 14, 8
    torch.manual_seed(41 if not variant else 43)
    enc = TinyEncoder(8, D)
    if variant == "_emb":
        pred = EmbeddingPredictor(V, D, 0.0, 2, 2, "swish")
        joint = TransducerJoint(V, D, D, J, activation="gelu")
    else:
        pred = RNNPredictor(V, D, D, 0.0, H, 2, dropout=0.0)
        joint = TransducerJoint(V, D, D, J)
    ctc = CTC(V, D)
    dec = TinyAttnDecoder(V, D)
    cb = ContextBias(input_size=D, output_size=D, vocab_size=V, embedding_size=D, num_layers=1, attention_heads=2,
                     bias_encoder_type="linear", context_extractor="BLSTM", unified_hw_odim=HW, unified_hw_heads=2)
    m = Transducer(V, 0, enc, pred, joint, attention_decoder=dec, ctc=ctc, context_bias=cb, ctc_weight=0.1,
                   transducer_weight=0.75, attention_weight=0.15, reverse_weight=0.3, lsm_weight=0.1, hw_weight=0.4,
                   loss_mode="both").eval()
    with torch.no_grad():
        m.joint.ffn_out.weight *= 5
        m.joint.ffn_out.bias[0] += 1.5
        m.ctc.ctc_lo.weight *= 5
        m.ctc.ctc_lo.bias[0] += 1.0
    g = torch.Generator().manual_seed(42 if not variant else 44)
    out = {}
    # ---- forward (the loss dictionary)
    B, Tin, U = 3, 13, 4
    speech = torch.randn(B, Tin, 8, generator=g)
    slen = torch.tensor([13, 9, 11], dtype=torch.int32)
    text = torch.tensor([[3, 5, 2, 9], [4, 4, -1, -1], [7, 1, 6, -1]])
    tlen = torch.tensor([4, 2, 3], dtype=torch.int32)
    ctx = torch.tensor([[0, -1, -1], [3, 5, -1], [7, 1, 6]])
    ctx_len = torch.tensor([1, 2, 3], dtype=torch.int32)
    hw_label = torch.tensor([[1, 1, 0, 0], [0, 0, -1, -1], [1, 1, 1, -1]])
    res = m(speech, slen, text, tlen, ctx, ctx_len, hw_label)
    res["loss"].backward()                 # the reference's own autograd graph (RNN-T gradient from the oracle stub)
    out.update({"grad_" + k: p.grad.numpy().copy() for k, p in m.named_parameters() if p.grad is not None})
    res = {k: v.detach() for k, v in res.items()}
    out.update(fwd_speech=speech.numpy(), fwd_slen=slen.numpy(), fwd_text=text.numpy(), fwd_tlen=tlen.numpy(),
               fwd_ctx=ctx.numpy(), fwd_ctx_len=ctx_len.numpy(), fwd_hw_label=hw_label.numpy(),
               **{"fwd_" + k: np.array(float(v)) for k, v in res.items()})
    print("  forward dict:", {k: round(float(v), 5) for k, v in res.items()})
    # ---- decode wrappers (batch 1)
    Td = 26
    sp1 = torch.randn(1, Td, 8, generator=g)
    sl1 = torch.tensor([Td], dtype=torch.int32)
    out.update(dec_speech=sp1.numpy())
    with torch.no_grad():
        hyp, score = m.beam_search(sp1, sl1, beam_size=4, ctc_weight=0.3, transducer_weight=0.7)
        out.update(beam_hyp=np.array(hyp, np.int64), beam_score=np.array(float(score)))
        print("  beam_search:", hyp, float(score))
        k = 0
        for typ in ("transducer", "ctc"):
            for rw in (0.0, 0.3):
                h, s = m.transducer_attention_rescoring(sp1, sl1, 4, reverse_weight=rw, ctc_weight=0.2, attn_weight=0.3,
                                                        transducer_weight=0.5, search_ctc_weight=0.3,
                                                        search_transducer_weight=0.7, beam_search_type=typ)
                out[f"resc_{k}_type"] = np.array(typ); out[f"resc_{k}_rw"] = np.array(rw)
                out[f"resc_{k}_hyp"] = np.array(list(h), np.int64); out[f"resc_{k}_score"] = np.array(float(s))
                print(f"  rescoring {typ} rw={rw}:", list(h), float(s))
                k += 1
        out["n_resc"] = np.array(k)
        labels = torch.randint(0, 2, (1, 8), generator=g)
        with contextlib.redirect_stdout(io.StringIO()):
            gh, gd = m.greedy_search(sp1, sl1, n_steps=4, context_list=ctx, context_lengths=ctx_len,
                                     context_filter_state="on", context_decoder_labels_padded=labels)
        out.update(greedy_hyp=np.array(gh[0], np.int64), greedy_dist=np.array(float(gd)), greedy_labels=labels.numpy())
        print("  greedy_search:", gh, gd)
    save("transducer_wrappers" + variant, **out, **{"m_" + k: v for k, v in sd(m).items()}, heads=np.array(2),
         hw_dim=np.array(HW), hw_heads=np.array(2))


def gen_common():
    from wenet.utils.common import add_blank, log_add
    ys = torch.tensor([[1, 2, 3, 4, 5], [4, 5, 6, -1, -1], [7, 8, 9, -1, -1]])
    ab = add_blank(ys, 0, -1).numpy()
    pairs = np.array([[-1.5, -2.25], [-100.0, -0.5], [-3.0, -3.0], [-float("inf"), -2.0]])
    la = np.array([log_add(list(p)) for p in pairs])
    save("common_ref", ys=ys.numpy(), add_blank=ab, log_add_in=pairs, log_add_out=la,
         log_add_all_inf=np.array(log_add([-float("inf"), -float("inf")])))


def gen_rnnt_kat():
    # Public warp-transducer / torchaudio unit-test vector; not produced by the reference.
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from test_oracle_rnnt import KAT_COST, KAT_GRAD, KAT_LOGITS
    save("rnnt_kat", logits=KAT_LOGITS, targets=np.array([[1, 2]], np.int32), cost=np.array(KAT_COST), grad=KAT_GRAD)


if __name__ == "__main__":
    install_stubs()
    torch.set_num_threads(1)
    only = os.environ.get("GOLDEN_ONLY")          # e.g. GOLDEN_ONLY=gen_joint_variants,gen_predictor: just these
    if only:
        for name in only.split(","):
            globals()[name]()
        sys.exit(0)
    gen_common()
    gen_rnnt_kat()
    gen_ctc()
    gen_joint()
    gen_joint_variants()
    gen_predictor()
    gen_predictor_variants()
    gen_greedy()
    gen_beam()
    gen_decode_variants()
    gen_ctc_decode()
    gen_ctc_align()
    gen_greedy_fork()
    gen_greedy_stream()
    gen_greedy_both_real()
    gen_transducer_wrappers()
    gen_asr_surface()
