#!/usr/bin/env python3
"""bench.py -- sequences/sec (fwd+bwd) of the ExplaiNN hot path on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  300-unit ExplaiNN, kernel 19, 200 bp one-hot, 1 binary task, batch 1024 PER GPU, fp32.
A step = one pass of the hot path over one synthetic batch already resident in HBM:
train-mode forward (dropout 0.3 active) + BCE-with-logits loss + backward into a flat gradient
buffer, plus -- for N > 1 -- one RCCL all-reduce of that buffer (weak scaling, batch sharding).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`)

Prints ONE JSON line on rank 0 (see DESIGN.md section 6 for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

U, K, L, T, B_PER_GPU = 300, 19, 200, 1, 1024
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_step(B, P):
    """per sequence 16*L (fp32 one-hot in) + 8*T (targets in, logits out);
    per step 2*4*P (parameters read, gradients written)."""
    return B * (16 * L + 8 * T) + 2 * 4 * P


def measured_traffic():
    """HBM bytes per step from the PMC passes committed under profiles/ (rocprofv3 cannot run inside
    this process); None if the file is absent or was taken on another workload."""
    path = os.path.join(ROOT, "profiles", "r01_final_traffic.json")
    try:
        with open(path) as fh:
            return int(json.load(fh)["traffic_bytes_per_step"])
    except Exception:
        return None


def measured_copy_peak(device):
    """Device-to-device copy rate on this box (read + write bytes / time), the practical HBM roof
    next to the 8 TB/s spec (SURVEY.md 8d asks for both)."""
    n = 1 << 28                                        # 1 GiB of fp32 per buffer, beyond the 256 MB cache
    a = torch.empty(n, device=device, dtype=torch.float32).fill_(1.0)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    gbps = 5 * 2 * n * 4 / (e0.elapsed_time(e1) / 1e3) / 1e9
    del a, b
    torch.cuda.empty_cache()
    return round(gbps, 1)


def synthetic_batch(B, seed, device, n_frac=0.0):
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, 4, (B, L), generator=g)
    x = torch.zeros(B, 4, L).scatter_(1, idx[:, None, :], 1.0)
    if n_frac > 0:                                    # N bases = all-zero columns
        x = x * (torch.rand(B, 1, L, generator=g) >= n_frac)
    y = (torch.rand(B, T, generator=torch.Generator().manual_seed(seed + 1)) > 0.5).float()
    return x.to(device), y.to(device)


def cpu_baseline(budget_s=20.0):
    """The stock-PyTorch CPU restatement of the reference (oracle/torch_ref.py, pinned to the
    reference's outputs by tests/test_torch_ref_golden.py), timed on this box's host cores on a
    bounded sample of the same workload: whole C2 batches of 1024 until ~budget_s is used."""
    from oracle import torch_ref
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a 1-GPU box gives this job a 16-core share of the host (more threads only oversubscribe:
    # 256 threads measured 72 seq/s against 16 threads' several hundred)
    cores = min(cores, int(os.environ.get("EXPLAINN_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    sd = torch_ref.init_state(U, K, L, T, seed=0)
    x, y = synthetic_batch(B_PER_GPU, 1, "cpu")
    torch_ref.train_step(sd, x, y)                      # warm-up (allocations, oneDNN primitives)
    t0 = time.perf_counter()
    steps = 0
    while True:
        torch_ref.train_step(sd, x, y)
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 20:
            break
    return {"value": round(steps * B_PER_GPU / el, 1), "unit": "sequences/s", "cores": cores,
            "kind": "port",
            "sample": "%d train steps (fwd+BCE+bwd, dropout 0.3) of the 300-unit/200bp/batch-1024 "
                      "workload, stock-PyTorch CPU ops, %d threads, %.1f s" % (steps, cores, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--n-frac", type=float, default=0.0,
                    help="fraction of N bases in the synthetic batch (robustness variant; the "
                         "headline is 0)")
    ap.add_argument("--skip-optimizer", action="store_true",
                    help="skip the secondary fwd+bwd+Adam timing (profiling runs: one leg only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # EXPLAINN_BENCH_FORCE_SYNC=1: rehearse the N > 1 code path (process group, overlapped RCCL
    # all-reduce, barriers) in a one-rank group on a single GPU; the collectives are identities
    force_sync = world == 1 and os.environ.get("EXPLAINN_BENCH_FORCE_SYNC") == "1"
    if world > 1 or force_sync:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from explainn_amd import ExplaiNN
    from explainn_amd.engine import StepEngine
    from explainn_amd.parallel import broadcast_parameters, GradAllReduce

    torch.manual_seed(0)
    model = ExplaiNN(U, K, L, T).to(dev).train()
    model.validate_input = False          # flags are checked once, after the timed region
    if world > 1:
        broadcast_parameters(model)
    eng = StepEngine(model, B_PER_GPU, loss="binary")
    sync = (GradAllReduce(eng.flat_grad, split=eng.conv_grad_elements, force=force_sync)
            if dist is not None else None)
    # Default: ONE all-reduce of the flat buffer after the step.  EXPLAINN_BENCH_OVERLAP=1 switches to
    # the two-part reduction (FC/head gradients reduced under the filter-bank backward); in the
    # one-rank rehearsal its second collective cost more than the overlap saved
    overlap = os.environ.get("EXPLAINN_BENCH_OVERLAP", "0") == "1"
    P = eng.flat_grad.numel()
    x, y = synthetic_batch(B_PER_GPU, 1000 + rank, dev, args.n_frac)

    def one_step(i):
        if overlap:
            eng.step(x, y, seed=(rank << 40) + i + 1, grad_sync=sync)
        else:
            eng.step(x, y, seed=(rank << 40) + i + 1)
            if sync is not None:
                sync()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        one_step(args.warmup + i)
    ev1.record()
    fence()
    wall = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)       # HIP events on the launch stream (torch current stream)
    # secondary figure ("report with and without the optimiser step"): the same step
    # followed by the fused Adam update, timed the same way; not part of `value`
    wall_opt = 0.0
    if not args.skip_optimizer:
        from explainn_amd import get_optimizer
        opt = get_optimizer(model.parameters(), lr=0.003)
        eng.attach_grads()
        for i in range(3):
            one_step(i)
            opt.step()
        fence()
        t1 = time.perf_counter()
        for i in range(args.steps):
            one_step(args.warmup + args.steps + i)
            opt.step()
        fence()
        wall_opt = time.perf_counter() - t1
    flags = model.input_flags()
    assert flags == 0, "synthetic input flagged as not one-hot"
    assert torch.isfinite(eng.loss).all() and torch.isfinite(eng.flat_grad).all()

    t = torch.tensor([wall, wall_opt], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall, wall_opt = float(t[0].item()), float(t[1].item())
    if rank == 0:
        seqs = B_PER_GPU * world * args.steps
        step_gpu_s = gpu_ms / 1e3 / args.steps
        alg = algorithmic_bytes_per_step(B_PER_GPU, P)
        achieved = alg / step_gpu_s / 1e9
        traffic = measured_traffic() if world == 1 else None
        copy_peak = measured_copy_peak(dev) if world == 1 else None
        out = {
            "metric": "sequences/sec (fwd+bwd), 200bp one-hot, 300 units, batch 1024",
            "value": round(seqs / wall, 1), "unit": "sequences/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: 300-unit ExplaiNN, k=19, 200 bp one-hot, 1 binary task, "
                                   "batch 1024 per GPU, train fwd + BCE + bwd (dropout 0.3)"
                                   + (", RCCL all-reduce of the flat gradient" if dist is not None else ""),
                       "cnn_units": U, "kernel_size": K, "sequence_length": L, "n_features": T,
                       "batch_per_gpu": B_PER_GPU, "global_batch": B_PER_GPU * world,
                       "parameters": P, "parallelism": "dp%d" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "traffic": traffic,
                         # what the memory side actually moved (SURVEY 8d "hbm_measured_GBps"):
                         # measured bytes / this run's GPU time; includes Infinity-Cache hits
                         "measured_GBps": round(traffic / step_gpu_s / 1e9, 1) if traffic else None,
                         "measured_frac": round(traffic / step_gpu_s / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
                         "traffic_note": "bytes per step, FETCH_SIZE(x2)+WRITE_SIZE from profiles/"
                                         "r01_final_traffic.json (separate rocprofv3 --pmc passes)",
                         "copy_peak_GBps_measured_here": copy_peak,
                         "kernel": "train_step pipeline (all launches of one step)",
                         "algorithmic_bytes_per_step": alg,
                         "gpu_ms_per_step_hip_events": round(step_gpu_s * 1e3, 4)},
        }
        out["with_optimizer"] = None if args.skip_optimizer else {
            "ms_per_step": round(wall_opt / args.steps * 1e3, 4),
            "value": round(seqs / wall_opt, 1), "unit": "sequences/s",
            "optimizer": "Adam(lr=0.003), one fused launch (csrc/adam.hip)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        if args.n_frac > 0:
            out["data"] = "synthetic, %.3g N bases" % args.n_frac
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
