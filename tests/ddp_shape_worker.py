"""BASELINE config 4 rehearsed AT SHAPE on one box (child process of
tests/test_transducer_gpu.py::test_two_rank_ddp_at_librispeech_shapes).

Config 4 is "U2++ conformer CTC+RNN-T joint training, LibriSpeech-960 shapes, DP batch shard across 8 x MI355X with
RCCL/xGMI all-reduce" (conf/encoder_bias_conformer_rnnt_4_head_bi_1_layer_2_Labels_both.yaml: 58.4 M parameters,
dynamic batches of <= 6000 fbank frames per rank, accum_grad 4, V = 5000, join_dim 512).  No 8-GPU node is available to
the builder, so this worker runs ONE rank of a two-rank job with everything that does not need eight devices at its
real size:
  * a stand-in encoder with the real parameter budget (12 residual blocks of four 256 <-> 2048 feed-forward pairs,
    50.4 M fp32 parameters; the conformer itself is stock PyTorch and out of scope) + the product's predictor
    (LSTM 2 x 256), joiner (256/256 -> 512 -> 5000) and CTC head: 57.1 M parameters, a 228 MB gradient all-reduce in
    DDP's 25 MB buckets;
  * a LibriSpeech dynamic batch per rank: 4 utterances of <= 1500 fbank frames (375 encoder frames after the 4x
    subsampling), 30-50 labels, different on every rank and accumulation step;
  * the step structure of wenet/bin/train.py:227-240 and wenet/utils/executor.py:48-53,81-86:
    DistributedDataParallel(find_unused_parameters=True), `join()`, accum_grad 4 = three `no_sync()` steps and one
    synchronising step.
Backend: nccl (RCCL) when every rank has its own device, gloo when the ranks share the box's single GPU.
Writes the accumulated gradients after the synchronising step and a timing breakdown (encoder / loss block / backward
with and without the all-reduce) for the parent to compare and print."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

V, D, J, H = 5000, 256, 512, 256
ACCUM = 4


def amp_on() -> bool:
    """WR_SHAPE_AMP=1: the reference's --use_amp step (executor.py:91 autocast, here bfloat16) with the joiner's 16-bit
    single-term mode: bf16 logits, the loss's bf16 gradient, the library-GEMM backward."""
    return os.environ.get("WR_SHAPE_AMP", "0") == "1"


class FFBlock(torch.nn.Module):
    def __init__(self, d, hidden, pairs):
        super().__init__()
        self.up = torch.nn.ModuleList(torch.nn.Linear(d, hidden) for _ in range(pairs))
        self.down = torch.nn.ModuleList(torch.nn.Linear(hidden, d) for _ in range(pairs))
        self.norm = torch.nn.LayerNorm(d)

    def forward(self, x):
        for u, dn in zip(self.up, self.down):
            x = x + 0.5 * dn(torch.relu(u(x)))
        return self.norm(x)


class BudgetEncoder(torch.nn.Module):
    """(B, Tin, 80) fbank -> (B, Tin // 4, 256), frame mask: the reference encoder's interface (encoder.py forward) with
    its parameter count, not its architecture."""

    def __init__(self, idim=80, d=D, blocks=12, hidden=2048, pairs=4):
        super().__init__()
        self.inp = torch.nn.Linear(4 * idim, d)
        self.blocks = torch.nn.ModuleList(FFBlock(d, hidden, pairs) for _ in range(blocks))

    def forward(self, xs, xs_lens, decoding_chunk_size=0, num_decoding_left_chunks=-1):
        B, Tin, F = xs.shape
        T = Tin // 4
        x = self.inp(xs[:, :T * 4].reshape(B, T, 4 * F))
        for blk in self.blocks:
            x = blk(x)
        lens = torch.div(xs_lens.to(xs.device), 4, rounding_mode="floor")
        mask = (torch.arange(T, device=xs.device)[None, :] < lens[:, None]).unsqueeze(1)
        return x, mask


def build_model(dev):
    import wenet_celoss_amd as w
    torch.manual_seed(7)
    m = w.Transducer(V, 0, BudgetEncoder(), w.RNNPredictor(V, D, D, 0.0, H, 2, dropout=0.0),
                     w.TransducerJoint(V, D, D, J, precision="bf16" if amp_on() else None),
                     ctc=w.CTC(V, D), ctc_weight=0.25, transducer_weight=0.75, hw_weight=0.0)
    return m.to(dev)


def micro_batch(rank, step, dev):
    """A dynamic batch of <= 6000 fbank frames: 4 utterances sorted by length (processor.py:704), 30-50 labels."""
    g = torch.Generator().manual_seed(1000 + 17 * rank + step)
    B = 4
    slen = torch.sort(torch.randint(1100, 1501, (B,), generator=g), descending=True).values
    slen[0] = 1500
    assert int(slen.sum()) <= 6000
    speech = torch.randn(B, 1500, 80, generator=g) * 0.5
    tlen = torch.randint(30, 51, (B,), generator=g)
    U = int(tlen.max())
    text = torch.randint(1, V - 1, (B, U), generator=g)
    for b in range(B):
        text[b, tlen[b]:] = -1
    return speech.to(dev), slen.to(torch.int32).to(dev), text.to(dev), tlen.to(torch.int32).to(dev)


def accumulate(model_call, params, rank, dev, sync_ctx=None, timings=None):
    """ACCUM micro-batches; `sync_ctx(i)` returns the context manager of step i (no_sync for all but the last)."""
    import contextlib
    for i in range(ACCUM):
        ctx = sync_ctx(i) if sync_ctx else contextlib.nullcontext()
        with ctx:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp_on()):  # executor.py:91
                loss = model_call(*micro_batch(rank, i, dev))["loss"] / ACCUM     # executor.py:101 loss / accum_grad
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            loss.backward()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        if timings is not None:
            timings.append({"step": i, "forward_ms": round((t1 - t0) * 1e3, 2), "backward_ms": round((t2 - t1) * 1e3, 2),
                            "loss": float(loss) * ACCUM})
    return {n: p.grad.detach().cpu().clone() for n, p in params if p.grad is not None}


def main():
    rank, world, out_path = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
    ndev = torch.cuda.device_count()
    own_device = ndev >= world
    dev = torch.device("cuda", rank if own_device else 0)
    torch.cuda.set_device(dev)
    backend = "nccl" if own_device else "gloo"
    dist.init_process_group(backend, rank=rank, world_size=world)
    m = build_model(dev)
    nparam = sum(p.numel() for p in m.parameters())
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[dev.index], find_unused_parameters=True)
    timings = []
    with ddp.join():                                              # executor.py:48-53
        # warm-up accumulation cycle (kernels built, allocator warm, DDP buckets rebuilt), then the measured one
        accumulate(ddp, list(m.named_parameters()), rank, dev, lambda i: ddp.no_sync() if i < ACCUM - 1 else _null())
        m.zero_grad()
        grads = accumulate(ddp, list(m.named_parameters()), rank, dev,
                           lambda i: ddp.no_sync() if i < ACCUM - 1 else _null(), timings)
    # the loss block alone (joiner + RNN-T loss + CTC, forward + backward) on the last micro-batch, for the breakdown
    speech, slen, text, tlen = micro_batch(rank, ACCUM - 1, dev)
    with torch.no_grad():
        enc, mask = m.encoder(speech, slen)
    enc = enc.detach().requires_grad_(True)
    from wenet_celoss_amd.common import add_blank
    pred = m.predictor(add_blank(text, 0, -1)).detach().requires_grad_(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp_on()):
        _, lr = m.compute_loss(enc, mask.squeeze(1).sum(1), pred, text, tlen)
    lr.backward()
    torch.cuda.synchronize()
    loss_block_ms = (time.perf_counter() - t0) * 1e3
    m.zero_grad()
    no_sync_bwd = sorted(t["backward_ms"] for t in timings[:-1])[len(timings[:-1]) // 2]
    info = {"rank": rank, "backend": backend, "amp": amp_on(), "parameters": nparam, "gradient_MB": round(nparam * 4 / 1e6, 1),
            "steps": timings, "loss_block_fwd_bwd_ms": round(loss_block_ms, 2),
            "all_reduce_ms(sync backward - median no_sync backward)": round(timings[-1]["backward_ms"] - no_sync_bwd, 2)}
    torch.save({"grads": grads, "info": json.dumps(info)}, f"{out_path}.rank{rank}")
    dist.destroy_process_group()


def _null():
    import contextlib
    return contextlib.nullcontext()


if __name__ == "__main__":
    main()
