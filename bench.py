#!/usr/bin/env python3
"""bench.py -- RNN-T loss + gradient throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic joiner
logits already resident in HBM: wr_rnnt_loss_fwd (row log-sum-exp + alpha/beta
sweeps -> costs) followed by wr_rnnt_loss_bwd (gradient w.r.t. the logits with
reduction="mean" folded in), through the C-ABI of libwr_mi355x.so.

Workload at N=1: BASELINE.json configs[1] -- B=32, T=1000, U=150, V=5000, fp32,
full-length utterances.  N>1, one process per GPU (started by torch.distributed.run,
or by this script itself when no launcher environment is set); the loss has no
data-path collective, RCCL only carries the barrier and the
max-over-ranks time:
  --scaling weak   (default) every rank runs its own batch of B utterances;
  --scaling strong the literal BASELINE metric: ONE global batch of B utterances,
                   dealt to the ranks by lattice size (dist.balanced_shards), each
                   shard padded to its own maxima.
value = utterances of the whole job per step / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- for the dominant kernel (rnnt_grad_kernel: 2/3 of the bytes),
                  algorithmic bytes per launch / its mean duration measured with
                  HIP events on the launch stream inside the timed region;
  cpu_baseline -- the threaded fp32 CPU port (oracle/rnnt_baseline.c, "port")
                  timed on this box's host cores on a bounded sample of the same
                  (T,U,V) utterances (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--B", type=int, default=32, help="utterances per GPU (weak) / in the whole job (strong)")
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--U", type=int, default=150)
    ap.add_argument("--V", type=int, default=5000)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every GPU runs its own batch of --B utterances; strong: ONE global batch of --B "
                         "utterances dealt to the GPUs by lattice size (wenet_celoss_amd.dist.balanced_shards)")
    ap.add_argument("--ragged", action="store_true", help="T_b~U{T/2..T}, U_b~U{U/3..U} (maxima pinned)")
    ap.add_argument("--inplace", action="store_true", help="write the gradient over the logits storage")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: let the N ranks share the visible devices (rank %% devices) and use the gloo "
                         "backend for the barrier -- exercises the N>1 code path on a 1-GPU box; the line is marked")
    ap.add_argument("--launch-timeout", type=float, default=3000.0,
                    help="self-launch only: seconds after which ranks that are still running are stopped (exit code 3)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="utterances timed on the CPU (-1: auto, 0: skip)")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="target CPU time budget for the baseline")
    ap.add_argument("--extra-seconds", type=float, default=90.0,
                    help="N=1 only: time budget of the secondary measurements appended as \"extra\" (tools/secondary.py: "
                         "joiner TFLOP/s, loss-block step, CTC, greedy, beam, hot-word); 0 skips them")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start one fresh child process per GPU
    (rank environment set per child, rendezvous on 127.0.0.1) BEFORE this process makes any GPU call, pass rank 0's
    JSON line through, and fail if any rank fails.  Nothing is exec'ed; the parent never initialises the GPU
    (counting devices does not)."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()
    if have < n and not (args.rehearse_on_one_gpu and have >= 1):
        missing = ", ".join(f"cuda:{i}" for i in range(have, n))
        print(f"bench.py: --gpus {n} needs {n} HIP devices but {have} {'is' if have == 1 else 'are'} visible "
              f"(missing: {missing})", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    failed = None
    pending = set(range(n))
    deadline = time.monotonic() + args.launch_timeout
    while pending and failed is None:
        time.sleep(0.2)
        if time.monotonic() > deadline:
            failed = (min(pending), "timeout")
            break
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0 and failed is None:
                    failed = (r, rc)
    if failed is not None:
        for r in pending:
            procs[r].kill()                       # exact PIDs of our own children
        for p in procs:
            p.wait()
        if failed[1] == "timeout":
            print(f"bench.py: ranks {sorted(pending)} still running after {args.launch_timeout:.0f} s; stopped", file=sys.stderr)
            return 3
        print(f"bench.py: rank {failed[0]} exited with code {failed[1]}", file=sys.stderr)
        return 1
    out = procs[0].stdout.read()
    sys.stdout.write(out)
    sys.stdout.flush()
    return 0 if out.strip() else 1


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path in the product)")
    if args.rehearse_on_one_gpu:
        local_rank %= torch.cuda.device_count()
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local_rank} but only {torch.cuda.device_count()} "
                         f"HIP device(s) are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev                                        # where the few scalars that cross ranks live
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
            red_dev = torch.device("cpu")
        else:
            dist.init_process_group("nccl", device_id=dev)   # RCCL; only the barrier / max-reduce of timings use it

    from wenet_celoss_amd import _lib
    lib = _lib.load()

    from wenet_celoss_amd.dist import balanced_shards
    Bjob, T, U, V = args.B, args.T, args.U, args.V
    strong = args.scaling == "strong"
    if strong and Bjob < world:
        raise SystemExit(f"bench.py: --scaling strong needs at least one utterance per GPU (B={Bjob}, gpus={world})")
    # lengths of the batch this rank draws from: the ONE global batch (strong; same seed on every rank) or the
    # rank's own batch (weak)
    len_seed = 20260 if strong else 20260 + rank
    if args.ragged:
        cpu_gen = torch.Generator().manual_seed(len_seed)
        tl = torch.randint(T // 2, T + 1, (Bjob,), generator=cpu_gen)
        ul = torch.randint(max(U // 3, 1), U + 1, (Bjob,), generator=cpu_gen)
        tl[0], ul[0] = T, U
        order = torch.argsort(tl, descending=True)      # processor.py:704 sorts by feats length
        tl, ul = tl[order], ul[order]
        ul[ul.argmax()] = U
    else:
        tl = torch.full((Bjob,), T)
        ul = torch.full((Bjob,), U)
    if strong:
        mine = balanced_shards([int(a) * (int(b) + 1) for a, b in zip(tl, ul)], world)[rank]
        tl, ul = tl[mine], ul[mine]
        T, U = int(tl.max()), int(ul.max())              # a DP shard is padded to its own maxima
    B = int(tl.numel())
    U1 = U + 1
    gen = torch.Generator(device=dev)
    gen.manual_seed(20260 + rank)
    logits = torch.empty(B, T, U1, V, dtype=torch.float32, device=dev)
    for b in range(B):                       # per-utterance fill: no 96 GB temporary
        logits[b].normal_(generator=gen)
    targets = torch.randint(1, V, (B, U), dtype=torch.int32, device=dev, generator=gen)
    llens = tl.to(torch.int32).to(dev)
    tlens = ul.to(torch.int32).to(dev)
    valid_cells = int((tl * (ul + 1)).sum())
    pad_cells = B * T * U1 - valid_cells

    inplace = args.inplace
    grads = None
    if not inplace:
        try:
            grads = torch.empty_like(logits)
        except torch.OutOfMemoryError:
            inplace = True
    if inplace:
        grads = logits
    ws_bytes = lib.wr_rnnt_workspace_bytes(B, T, U1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    costs = torch.empty(B, dtype=torch.float32, device=dev)
    # reduction="mean": weak = each rank's own batch mean (what DDP then averages); strong = mean over the global batch
    gcosts = torch.full((B,), 1.0 / (Bjob if strong else B), dtype=torch.float32, device=dev)
    stream = _lib.current_stream(dev)
    P = _lib.ptr

    def fwd():
        _lib.check(lib.wr_rnnt_loss_fwd(P(logits), 0, P(targets), P(llens), P(tlens), B, T, U1, V, 0, P(costs),
                                        P(ws), ws_bytes, stream), "fwd")

    def bwd():
        _lib.check(lib.wr_rnnt_loss_bwd(P(logits), 0, P(targets), P(llens), P(tlens), B, T, U1, V, 0, -1.0,
                                        P(gcosts), P(grads), P(ws), ws_bytes, stream), "bwd")

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        fwd(); bwd()
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        fwd()
        ev[k][1].record()
        bwd()
        ev[k][2].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # same-process calibration outside the timed region: the runtime's own device-to-device copy of the
    # logits-sized buffer (one read + one write per byte, like the gradient pass).  MI355X parts differ
    # by several per cent in sustained HBM rate; this says what this device gives a plain copy.
    copy_gbs = None
    if not inplace:
        cms = []
        for _ in range(3):
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record(); grads.copy_(logits); c1.record(); torch.cuda.synchronize()
            cms.append(c0.elapsed_time(c1))
        copy_gbs = 2 * 4.0 * B * T * U1 * V / (sorted(cms)[1] * 1e-3) / 1e9
        bwd()                                            # restore the gradient buffer's contents

    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    bwd_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    ms_per_step = elapsed * 1e3 / args.steps
    job_utts = Bjob if strong else Bjob * world
    value = job_utts * args.steps / elapsed
    if world > 1:                            # rank 0 reports its own kernel times; the job total needs every rank's bytes
        import torch.distributed as dist
        cells = torch.tensor([valid_cells, pad_cells], dtype=torch.float64, device=red_dev)
        dist.all_reduce(cells)
        job_valid, job_pad = float(cells[0]), float(cells[1])
    else:
        job_valid, job_pad = float(valid_cells), float(pad_cells)

    # algorithmic bytes: 4*V per valid cell per pass; pass 3 also zero-fills padded cells
    bytes_fwd = 4.0 * V * valid_cells
    bytes_bwd = 2 * 4.0 * V * valid_cells + 4.0 * V * pad_cells
    grad_gbs = bytes_bwd / (bwd_ms * 1e-3) / 1e9
    lse_gbs = bytes_fwd / (fwd_ms * 1e-3) / 1e9

    # HBM traffic of the dominant kernel from the committed PMC profile of this same command/shape (bench.py
    # cannot collect counters itself); only reported for the default configuration it was measured on.
    traffic, traffic_source = None, None
    try:
        if (B, T, U, V) == (32, 1000, 150, 5000) and not args.ragged and not inplace:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                tj = json.load(f)
            traffic = float(tj["kernels"]["rnnt_grad_kernel"]["hbm_bytes"])
            traffic_source = ("committed constant from profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                              "passes of this command and shape, " + str(tj.get("source", "see profiles/")) + "); "
                              "not re-measured in this run")
    except (OSError, KeyError, ValueError):
        traffic, traffic_source = None, None

    out = {
        "metric": f"utterances/sec RNN-T loss+grad (B={Bjob},T={args.T},U={args.U},V={V})",
        "value": round(value, 3),
        "unit": "utterances/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"rnnt_loss+grad B={Bjob} T={args.T} U={args.U} V={V} fp32 "
                               f"{'ragged' if args.ragged else 'full-length'} logits resident in HBM"
                               f"{' (gradient written in place)' if inplace else ''}"
                               + (f"; one global batch dealt to {world} GPUs by lattice size" if strong and world > 1 else ""),
                   "per_gpu_batch": B if not strong else f"{B} on rank 0 (global {Bjob} dealt by balanced_shards)",
                   "global_batch": job_utts,
                   "parallelism": f"dp{world} (utterance shards, no data-path collective)"},
        "roofline": {"bound": "hbm", "kernel": "rnnt_grad_kernel", "achieved": round(grad_gbs, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(grad_gbs / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_source,
                     "bytes_per_launch": bytes_bwd, "avg_ms": round(bwd_ms, 4),
                     "other": {"rnnt_lse+sweep": {"achieved": round(lse_gbs, 1), "avg_ms": round(fwd_ms, 4),
                                                  "bytes_per_launch": bytes_fwd}},
                     "whole_step_GBps": round((3 * 4.0 * V * job_valid + 4.0 * V * job_pad) / (ms_per_step * 1e-3) / 1e9, 1),
                     "device_copy_GBps": None if copy_gbs is None else round(copy_gbs, 1)},
    }

    if args.rehearse_on_one_gpu:
        out["rehearsal"] = "ranks share one device (development run, not a scaling measurement)"
    if rank == 0 and world == 1 and args.cpu_sample != 0:
        out["cpu_baseline"] = cpu_baseline(args, logits if not inplace else None, targets, llens, tlens, costs, gen, dev)
    if rank == 0 and world == 1 and args.extra_seconds > 0:
        # secondary metrics (SURVEY.md 8d "Secondary"), after the timed region and the CPU baseline, every leg guarded
        # on its own; the 193 GB of headline buffers are released first
        fwd = bwd = None
        del logits, grads, ws
        torch.cuda.empty_cache()
        try:
            from tools import secondary
            out["extra"] = secondary.collect(dev, budget_s=args.extra_seconds)
        except Exception as e:                  # never lose the headline line
            out["extra"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def host_cores() -> int:
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, int(q / int(f.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(args, logits, targets, llens, tlens, gpu_costs, gen, dev):
    """Time the threaded fp32 CPU port on a bounded sample of the same workload."""
    import numpy as np
    import oracle
    B, T, U, V = args.B, args.T, args.U, args.V
    cores = host_cores()
    per_utt_gb = T * (U + 1) * V * 4 / 1e9
    bs = args.cpu_sample
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") / 1e9
    except (ValueError, OSError):
        avail = 32.0
    avail = min(avail, 200.0)                  # gpurun caps one command at ~270 GiB of host memory
    # SURVEY.md 8d protocol: B' = min(B, floor(0.4 * RAM / (2 * 3.02 GB))) utterances of the same (T,U,V), 3 warm-ups,
    # >= 5 timed repetitions, median.  Work is exactly linear in B' (utterances are independent), so when 3 + 5 runs of
    # the RAM-sized sample would not fit --cpu-seconds (bench.py has to finish within minutes) B' is cut down to what
    # does, never below 1; the sample actually used and every count are reported in "sample".
    ram_bs = int(max(1, min(B, avail * 0.4 // (2 * per_utt_gb))))
    if bs < 0:
        bs = ram_bs
    bs = int(max(1, min(bs, B)))
    if logits is None:      # in-place run destroyed the logits: regenerate the sample
        logits = torch.empty(bs, T, U + 1, V, dtype=torch.float32, device=dev)
        logits.normal_(generator=gen)
    y = targets[:bs].cpu().numpy()
    # probe: one utterance, to size the sample to the time budget
    x1 = logits[:1].cpu().numpy()
    g1 = np.empty_like(x1)
    t0 = time.perf_counter()
    oracle.rnnt_loss_f32(x1, y[:1], np.array([T], np.int32), np.array([U], np.int32), nthreads=cores, out_grad=g1)
    oracle.rnnt_loss_f32(x1, y[:1], np.array([T], np.int32), np.array([U], np.int32), nthreads=cores, out_grad=g1)
    per_utt_s = (time.perf_counter() - t0) / 2
    del x1, g1
    WARM, REPS = 3, 5
    if args.cpu_sample < 0:
        bs = int(max(1, min(bs, args.cpu_seconds // ((WARM + REPS) * per_utt_s))))
    x = logits[:bs].cpu().numpy()
    y = y[:bs]
    ll = llens[:bs].cpu().numpy().copy()
    tl = tlens[:bs].cpu().numpy().copy()
    ll[0], tl[0] = T, U               # keep the maxima pinned for the sliced batch
    g = np.empty_like(x)
    for _ in range(WARM):             # page faults, thread pool, caches
        c, _ = oracle.rnnt_loss_f32(x, y, ll, tl, nthreads=cores, out_grad=g)
    times = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        c, _ = oracle.rnnt_loss_f32(x, y, ll, tl, nthreads=cores, out_grad=g)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    res = {"value": round(bs / med, 3), "unit": "utterances/s", "cores": cores, "kind": "port",
           "sample": f"{bs} of {B} utterances at the same (T={T},U={U},V={V}), fp32, loss+grad, "
                     f"median of {REPS} runs after {WARM} warm-ups (SURVEY 8d protocol; RAM-sized sample would be {ram_bs}, "
                     f"cut to fit {args.cpu_seconds:.0f} s of CPU time); oracle/rnnt_baseline.c with {cores} OpenMP threads "
                     f"(torchaudio is not installed on this image)"}
    try:
        gc = gpu_costs[:bs].cpu().numpy()
        same_lens = bool((llens[:bs].cpu().numpy() == ll).all() and (tlens[:bs].cpu().numpy() == tl).all())
        if same_lens:
            res["max_rel_cost_diff_vs_gpu"] = float(np.max(np.abs(gc - c) / np.maximum(1.0, np.abs(c))))
    except Exception:   # the check is informational
        pass
    return res


if __name__ == "__main__":
    main()
