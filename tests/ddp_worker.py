"""Child process of tests/test_transducer_gpu.py::test_two_rank_ddp_training_step_on_the_product_path.

One rank of a 2-rank data-parallel job, started fresh (no GPU state inherited): builds the tiny Transducer of the
GPU tests on cuda:0, wraps it in DistributedDataParallel(find_unused_parameters=True) exactly as
wenet/bin/train.py:227-240 does, and runs the step structure of wenet/utils/executor.py:48-53,81-86 on its own shard
of a 5-utterance batch (3 + 2): inside `model.join()`, one accumulation step under `no_sync()`, one synchronising
step, and -- on rank 0 only -- one extra step that rank 1 shadows through join.  Gradients after each synchronising
step are written to `out_path` for the parent to compare.  Backend gloo (both ranks share the box's single GPU; RCCL
needs one device per rank); the collective still runs on the HIP tensors DDP hands it."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def batch():
    g = torch.Generator().manual_seed(31)
    speech = torch.randn(5, 12, 8, generator=g)
    slen = torch.tensor([12, 10, 9, 12, 7], dtype=torch.int32)
    text = torch.tensor([[3, 5, 2, 9], [4, 4, -1, -1], [7, 1, 6, -1], [2, 8, 8, 1], [5, -1, -1, -1]])
    tlen = torch.tensor([4, 2, 3, 4, 1], dtype=torch.int32)
    return speech, slen, text, tlen


SHARDS = [(0, 3), (3, 5)]                      # uneven: 3 + 2 utterances


def shard(rank, dev, scale=1.0):
    speech, slen, text, tlen = batch()
    lo, hi = SHARDS[rank]
    T = int(slen[lo:hi].max())
    U = int(tlen[lo:hi].max())
    return ((speech[lo:hi, :T] * scale).to(dev), slen[lo:hi].to(dev), text[lo:hi, :U].to(dev), tlen[lo:hi].to(dev))


def main():
    rank, world, out_path = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_transducer_gpu import build
    m = build()                                        # same seed on both ranks -> same initial weights
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], find_unused_parameters=True)
    saved = {}
    with ddp.join():                                   # executor.py:48-53
        with ddp.no_sync():                            # executor.py:81-86: accumulation step, no all-reduce
            ddp(*shard(rank, dev, 1.0))["loss"].backward()
        ddp(*shard(rank, dev, 0.5))["loss"].backward()  # synchronising step: all-reduce of the accumulated gradients
        saved["sync"] = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
        m.zero_grad()
        if rank == 0:                                  # rank 1 has run out of data: join shadows this all-reduce
            ddp(*shard(rank, dev, 0.25))["loss"].backward()
            saved["extra"] = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    torch.save(saved, f"{out_path}.rank{rank}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
