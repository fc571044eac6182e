"""The drop-in seam (wenet_celoss_amd.patch) and BASELINE config 1 (CTC on CPU tensors), without a GPU.

* a stub `wenet` package tree whose `utils/init_model.py` binds its classes by the same import statements as the
  reference's (wenet/utils/init_model.py:16-22) picks up the replacements after `patch.install()`; the stub's own
  `wenet/transducer/transducer.py` raises on import (as the reference's does here: `import torchaudio`), which
  proves the swap happened before the import;
* with the real reference tree present (this container only) the reference's unedited `init_model()` builds a
  `wenet_celoss_amd.Transducer` around the reference's own ConformerEncoder;
* `CTC.forward` on CPU tensors runs the reference's statements on stock torch.nn.CTCLoss and reproduces the
  fixtures generated from the reference module."""
import glob
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

REF = "/root/reference"


def _make_stub_tree(root):
    for d in ("wenet", "wenet/utils", "wenet/transducer", "wenet/transformer"):
        os.makedirs(os.path.join(root, d))
        open(os.path.join(root, d, "__init__.py"), "w").close()
    with open(os.path.join(root, "wenet/transducer/transducer.py"), "w") as f:
        f.write("import torchaudio_that_does_not_exist\n")
    for name in ("wenet/transducer/joint.py", "wenet/transducer/predictor.py", "wenet/transformer/ctc.py"):
        with open(os.path.join(root, name), "w") as f:
            f.write("raise ImportError('the original module must not be imported after patch.install()')\n")
    with open(os.path.join(root, "wenet/utils/init_model.py"), "w") as f:
        f.write(textwrap.dedent("""
            import torch
            from wenet.transducer.joint import TransducerJoint
            from wenet.transducer.predictor import (ConvPredictor, EmbeddingPredictor, RNNPredictor)
            from wenet.transducer.transducer import Transducer
            from wenet.transformer.ctc import CTC

            def init_model(configs):
                v = configs['output_dim']
                enc = torch.nn.Identity()
                ctc = CTC(v, configs['enc'])
                pred = RNNPredictor(v, **configs['predictor_conf'])
                joint = TransducerJoint(v, enc_output_size=configs['enc'],
                                        pred_output_size=configs['predictor_conf']['output_size'], **configs['joint_conf'])
                return Transducer(vocab_size=v, blank=0, predictor=pred, encoder=enc, context_bias=None,
                                  attention_decoder=None, joint=joint, ctc=ctc, **configs['model_conf'])
            """))


def _run(code, extra_path):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, extra_path]), PYTHONDONTWRITEBYTECODE="1")
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True,
                          timeout=300)


def test_install_swaps_classes_in_stub_tree(tmp_path):
    _make_stub_tree(str(tmp_path))
    r = _run("""
        import wenet_celoss_amd as w
        import wenet_celoss_amd.patch as patch
        assert not patch.installed()
        patch.install()
        assert patch.installed()
        from wenet.utils.init_model import init_model
        m = init_model(dict(output_dim=11, enc=6,
                            predictor_conf=dict(embed_size=6, output_size=6, embed_dropout=0.1, hidden_size=8, num_layers=2),
                            joint_conf=dict(join_dim=12),
                            model_conf=dict(ctc_weight=0.25, transducer_weight=0.75, attention_weight=0.0)))
        assert type(m) is w.Transducer and type(m.joint) is w.TransducerJoint and type(m.ctc) is w.CTC
        assert type(m.predictor) is w.RNNPredictor
        import wenet.transducer.predictor as p
        assert p.ConvPredictor is w.ConvPredictor and p.EmbeddingPredictor is w.EmbeddingPredictor
        assert issubclass(p.RNNPredictor, p.PredictorBase)
        c = p.ConvPredictor(7, 4, 0.1)
        assert c.context_size == 3 and c.conv.weight.shape == (4, 1, 3)
        patch.uninstall()
        assert not patch.installed()
        print("OK")
        """, str(tmp_path))
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_without_install_the_stub_tree_fails_like_the_reference(tmp_path):
    _make_stub_tree(str(tmp_path))
    r = _run("from wenet.utils.init_model import init_model", str(tmp_path))
    assert r.returncode != 0 and "must not be imported" in r.stderr


def test_install_refuses_after_init_model_import(tmp_path):
    r = _run("""
        import sys, types
        im = types.ModuleType("wenet.utils.init_model")
        class T: pass
        T.__module__ = "wenet.transducer.transducer"
        im.Transducer = T
        sys.modules["wenet.utils.init_model"] = im
        import wenet_celoss_amd.patch as patch
        try:
            patch.install()
        except RuntimeError as e:
            assert "before install" in str(e); print("OK")
        """, str(tmp_path))
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "wenet")), reason="reference tree only exists in the build container")
def test_reference_init_model_unedited_builds_our_transducer():
    """wenet/utils/init_model.py:29-109 of the real reference, imported after patch.install() (plus the two
    in-process stubs for modules the image lacks), builds wenet_celoss_amd.Transducer around the
    reference's own ConformerEncoder / BiTransformerDecoder / ContextBias."""
    r = _run("""
        import sys, types
        tg = types.ModuleType("typeguard"); tg.check_argument_types = lambda *a, **k: True
        sys.modules["typeguard"] = tg
        tt = types.ModuleType("turtle"); tt.forward = None
        sys.modules["turtle"] = tt
        import wenet_celoss_amd as w
        import wenet_celoss_amd.patch as patch
        patch.install()
        from wenet.utils.init_model import init_model
        cfg = dict(cmvn_file=None, is_json_cmvn=True, input_dim=80, output_dim=50, encoder='conformer',
                   decoder='bitransformer', context='bias',
                   encoder_conf=dict(output_size=16, attention_heads=2, linear_units=32, num_blocks=1, dropout_rate=0.0,
                                     input_layer='conv2d', normalize_before=True),
                   decoder_conf=dict(attention_heads=2, linear_units=32, num_blocks=1, r_num_blocks=1, dropout_rate=0.0),
                   context_conf=dict(embedding_size=16, num_layers=1, attention_heads=2, bias_encoder_type='linear',
                                     context_extractor='BLSTM', unified_hw_odim=8, unified_hw_heads=2),
                   predictor='rnn',
                   predictor_conf=dict(embed_size=16, output_size=16, embed_dropout=0.1, hidden_size=16, num_layers=2),
                   joint_conf=dict(join_dim=32, prejoin_linear=True, postjoin_linear=False, joint_mode='add', activation='tanh'),
                   model_conf=dict(ctc_weight=0.1, attention_weight=0.15, transducer_weight=0.75, reverse_weight=0.3,
                                   lsm_weight=0.1, length_normalized_loss=False, loss_mode='both'))
        m = init_model(cfg)
        assert type(m) is w.Transducer, type(m)
        assert type(m.encoder).__module__ == "wenet.transformer.encoder"
        assert type(m.context_bias).__module__ == "wenet.transformer.context_bias"
        assert type(m.ctc) is w.CTC and type(m.joint) is w.TransducerJoint
        from wenet_celoss_amd.hotword import device_capable
        assert device_capable(m.context_bias)      # the reference's own ContextBias takes the device hot-word path
        keys = set(m.state_dict())
        for k in ("joint.ffn_out.weight", "ctc.ctc_lo.bias", "predictor.rnn.weight_hh_l1", "encoder.after_norm.weight"):
            assert k in keys, k
        print("OK")
        """, REF)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "ctc_ref_*.npz"))))
def test_ctc_module_on_cpu_is_the_reference_call(path):
    """BASELINE config 1: CTC-only plumbing on CPU.  wenet_celoss_amd.CTC on CPU tensors == the reference module's
    recorded loss and gradients (ctc.py:46-64 on stock torch.nn.CTCLoss)."""
    import wenet_celoss_amd as w
    d = np.load(path)
    V, D = d["w_ctc_lo.weight"].shape
    ctc = w.CTC(V, D)
    ctc.load_state_dict({"ctc_lo.weight": torch.tensor(d["w_ctc_lo.weight"]), "ctc_lo.bias": torch.tensor(d["w_ctc_lo.bias"])})
    hs = torch.tensor(d["hs"], requires_grad=True)
    loss = ctc(hs, torch.tensor(d["hlens"]), torch.tensor(d["ys"]), torch.tensor(d["ys_lens"]))
    if np.isfinite(d["loss"]):
        assert loss.item() == pytest.approx(float(d["loss"]), rel=1e-6)
        loss.backward()
        np.testing.assert_allclose(hs.grad.numpy(), d["grad_hs"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ctc.ctc_lo.weight.grad.numpy(), d["grad_w"], rtol=1e-5, atol=1e-6)
    else:
        assert not torch.isfinite(loss)
    np.testing.assert_allclose(ctc.log_softmax(hs.detach()).detach().numpy(), d["log_softmax"], rtol=1e-6, atol=1e-6)


def test_config1_ctc_only_batch2_runs_on_cpu():
    """configs[0]: batch 2, <= 200 input frames -> <= 49 encoder frames, V = 5000, CTC only, no GPU."""
    import wenet_celoss_amd as w
    torch.manual_seed(0)
    ctc = w.CTC(5000, 256)
    hs = torch.randn(2, 49, 256, requires_grad=True)
    ys = torch.randint(1, 5000, (2, 12)); ys[1, 9:] = -1
    loss = ctc(hs, torch.tensor([49, 40]), ys, torch.tensor([12, 9]))
    ref = torch.nn.functional.ctc_loss(ctc.ctc_lo(hs).transpose(0, 1).log_softmax(2), ys, torch.tensor([49, 40]),
                                       torch.tensor([12, 9]), reduction="sum") / 2
    assert torch.isfinite(loss) and loss.item() == pytest.approx(ref.item(), rel=1e-6)
    loss.backward()
    assert torch.isfinite(hs.grad).all() and hs.grad[1, 40:].abs().max() == 0
    # the functional fused op and the RNN-T / joiner paths still refuse CPU tensors
    with pytest.raises(RuntimeError, match="HIP device"):
        w.ctc_loss(torch.zeros(1, 4, 5), torch.ones(1, 1, dtype=torch.long), torch.tensor([4]), torch.tensor([1]))


# wenet/bin/recognize.py:259-362: args.mode -> (method called on the model, keyword arguments it passes besides
# (feats, feats_lengths)).  Restated as data; checked against the reference file's text when the tree is present.
_RECOGNIZE_MODES = {
    "attention": ("recognize", dict(beam_size=4, decoding_chunk_size=-1, num_decoding_left_chunks=-1, simulate_streaming=False)),
    "ctc_greedy_search": ("ctc_greedy_search", dict(decoding_chunk_size=-1, num_decoding_left_chunks=-1, simulate_streaming=False)),
    "rnnt_greedy_search": ("greedy_search", dict(decoding_chunk_size=-1, num_decoding_left_chunks=-1, simulate_streaming=False,
                                                 context_list=torch.IntTensor([[0]]), context_lengths=torch.IntTensor([1]),
                                                 context_filter_state="on",
                                                 context_decoder_labels_padded=torch.IntTensor([[0]]))),
    "rnnt_beam_search": ("beam_search", dict(decoding_chunk_size=-1, beam_size=4, num_decoding_left_chunks=-1,
                                             simulate_streaming=False, ctc_weight=0.3, transducer_weight=0.7,
                                             context_list=torch.IntTensor([[0]]), context_lengths=torch.IntTensor([1]))),
    "rnnt_beam_attn_rescoring": ("transducer_attention_rescoring",
                                 dict(decoding_chunk_size=-1, beam_size=4, num_decoding_left_chunks=-1, simulate_streaming=False,
                                      ctc_weight=0.2, transducer_weight=0.5, attn_weight=0.3, reverse_weight=0.3,
                                      search_ctc_weight=0.3, search_transducer_weight=0.7)),
    "ctc_beam_td_attn_rescoring": ("transducer_attention_rescoring",
                                   dict(decoding_chunk_size=-1, beam_size=4, num_decoding_left_chunks=-1, simulate_streaming=False,
                                        ctc_weight=0.2, transducer_weight=0.5, attn_weight=0.3, reverse_weight=0.3,
                                        search_ctc_weight=0.3, search_transducer_weight=0.7, beam_search_type="ctc")),
    "ctc_prefix_beam_search": ("ctc_prefix_beam_search", dict(beam_size=4, decoding_chunk_size=-1, num_decoding_left_chunks=-1,
                                                              simulate_streaming=False)),
    "attention_rescoring": ("attention_rescoring", dict(beam_size=4, decoding_chunk_size=-1, num_decoding_left_chunks=-1,
                                                        ctc_weight=0.5, simulate_streaming=False, reverse_weight=0.3)),
}


def test_every_recognize_mode_reaches_a_method_of_the_replacement_class():
    """north_star: bin/recognize.py runs unchanged.  Its eight `args.mode` branches (recognize.py:259-362) call eight
    methods on the model with fixed keyword sets; two of them (`recognize`, `attention_rescoring`) the reference's
    Transducer only inherits from ASRModel.  Each must exist on wenet_celoss_amd.Transducer, accept exactly that call,
    and -- run on CPU tensors here -- either complete (the pure host-logic mode 'attention') or stop at the product's
    "tensors must live on a HIP device" check: never AttributeError / TypeError."""
    import inspect
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_transducer_gpu import TinyAttnDecoder, TinyEncoder
    import wenet_celoss_amd as w
    torch.manual_seed(0)
    V, D = 23, 12
    m = w.Transducer(V, 0, TinyEncoder(8, D), w.RNNPredictor(V, D, D, 0.0, 14, 2, dropout=0.0), w.TransducerJoint(V, D, D, 16),
                     attention_decoder=TinyAttnDecoder(V, D), ctc=w.CTC(V, D), context_bias=None, ctc_weight=0.1,
                     transducer_weight=0.75, attention_weight=0.15, reverse_weight=0.3, hw_weight=0.0).eval()
    feats, lens = torch.randn(1, 17, 8), torch.tensor([17], dtype=torch.int32)
    completed = []
    for mode, (name, kwargs) in _RECOGNIZE_MODES.items():
        fn = getattr(m, name, None)
        assert callable(fn), f"--mode {mode}: Transducer has no method {name}"
        inspect.signature(fn).bind(feats, feats_lengths := lens, **kwargs)          # the call recognize.py makes binds
        try:
            with torch.no_grad():
                fn(feats, feats_lengths, **kwargs)
            completed.append(mode)
        except RuntimeError as e:                                                     # the HIP path refusing CPU tensors
            assert "HIP" in str(e) or "hip" in str(e), (mode, e)
    assert "attention" in completed
    # the exports torch_asr_model.cc reads at load time
    assert (m.subsampling_rate(), m.right_context(), m.sos_symbol(), m.eos_symbol()) == (4, 6, V - 1, V - 1)
    assert m.is_bidirectional_decoder() is True
    if os.path.isfile(os.path.join(REF, "wenet/bin/recognize.py")):
        import re
        text = open(os.path.join(REF, "wenet/bin/recognize.py")).read()
        called = set(re.findall(r"model\.([a-z_]+)\(", text)) - {"eval", "to", "load_state_dict"}
        assert called == {name for name, _ in _RECOGNIZE_MODES.values()}, called
        for mode in _RECOGNIZE_MODES:
            assert f"'{mode}'" in text, mode
