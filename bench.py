#!/usr/bin/env python3
"""bench.py -- sequences/sec (fwd+bwd) of the ExplaiNN hot path on MI355X.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  C2: 300-unit ExplaiNN, kernel 19, 200 bp one-hot, 1 binary task, batch 1024 PER GPU, fp32.
A step = one pass of the hot path over one synthetic batch already resident in HBM:
train-mode forward (dropout 0.3 active) + BCE-with-logits loss + backward into a flat gradient
buffer, plus -- for N > 1 -- one RCCL all-reduce of that buffer.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C3|C4|C5] [--scaling weak|strong]
  N > 1 runs one process per GPU over RCCL: either launched by the driver as
  `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, or -- when WORLD_SIZE is
  not set -- by this script itself: the parent (which never touches the GPU) starts that very command
  as a child process, relays rank 0's JSON line and exits with the child's status.

--scaling weak (default): every GPU takes the workload's per-GPU batch.  --scaling strong: the
workload's GLOBAL batch (C4: 8192 x 1000 bp x 50 tasks) is split over the N ranks.

Prints ONE JSON line on rank 0 (see DESIGN.md section 6 for every field).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

K = 19
# BASELINE.json configs[1..4]: units, length, tasks, per-GPU batch of the
# weak-scaling run, global batch of the strong-scaling run
WORKLOADS = {
    "C2": dict(U=300, L=200, T=1, B=1024, G=8192,
               text="C2: 300-unit ExplaiNN, k=19, 200 bp one-hot, 1 binary task, batch 1024 per GPU"),
    "C3": dict(U=300, L=200, T=50, B=4096, G=4096,
               text="C3: 300-unit, 200 bp, 50 binary tasks, batch 4096 per GPU"),
    "C4": dict(U=300, L=1000, T=50, B=1024, G=8192,
               text="C4: 300-unit, 1000 bp, 50 binary tasks, global batch 8192 (1024 per GPU at 8 GPUs)"),
    "C5": dict(U=2000, L=600, T=164, B=1024, G=8192,
               text="C5: 2000-unit, 600 bp, 164 binary tasks, global batch 8192 (1024 per GPU at 8 GPUs), fp32"),
}
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3      # same guide: fp32 MFMA = fp32 vector rate
LDS_PEAK_GBPS = 150000.0          # ds_read_b64/b128 aggregate
CLOCK_HZ, SIMDS = 2.4e9, 1024
MFMA_BF16_16x16x32_CLK = 16       # same guide: v_mfma_f32_16x16x32_bf16, cycles per SIMD
MFMA_BF16_32x32x16_CLK = 32       # v_mfma_f32_32x32x16_bf16 (tools/mfma_probe.hip: 32.0 back to back, 1 or 2 chains)
# tools/mfma_probe.hip, every SIMD issuing bf16 MFMAs back to back: the shader clock settles at
# 1.7-1.9 GHz (2.1 with constant operands), not the nominal 2.4 the roofs below are priced at --
# a kernel that keeps the matrix pipe busy is power-limited before it is issue-limited.
MFMA_SUSTAINED_CLOCK_GHZ = (1.7, 2.1)
# Measured here (tools/issue_rate.hip -> profiles/r03_issue_rate.txt), every SIMD of the chip holding
# 4 waves: shader cycles one SIMD needs per wave-instruction.  Round 2 priced every vector
# instruction at 4; the guide's 2-cycle figure holds for v_fma_f32 / v_xor_b32 only.
ISSUE_CLK_PER_INST = {"v_fma_f32, v_xor_b32": 2.4, "other VALU (shifts, bfe, perm, and_or, add3, max, mul_u24, packed fp32)": 4.2,
                      "v_cmp + v_cndmask pair (per instruction)": 3.3, "v_exp_f32": 8.1,
                      "ds_read_b128": 16.4, "vector instruction behind an MFMA (tools/mfma_probe.hip)": 5.0,
                      "global/buffer store behind an MFMA": 30.0}


def launcher_command(argv, gpus, port=None, python=None):
    """The command a plain `python bench.py --gpus N` re-issues itself as: one rank per GPU of this
    node under torch.distributed.run, rendezvous on 127.0.0.1 (the container's hostname may not
    resolve).  `argv` = this script's own arguments, passed through unchanged."""
    if port is None:
        port = 29500 + (os.getpid() % 2000)
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            "--nproc-per-node", str(int(gpus)), "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, gpus):
    """Parent of a multi-GPU run started without a launcher.  Nothing here initialises HIP (no torch
    import, no device query): the ranks are CHILD processes, never an exec of a process that has
    touched the GPU.  Rank 0's JSON line is the only thing written to stdout."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it here
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = launcher_command(argv, gpus)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for raw in proc.stdout:
        if raw.startswith("{") and '"metric"' in raw:
            line = raw.strip()
        else:
            sys.stderr.write(raw)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
        rc = 1
    return rc


def algorithmic_bytes_per_step(B, L, T, P):
    """per sequence 16*L (fp32 one-hot in) + 8*T (targets in, logits out);
    per step 2*4*P (parameters read, gradients written)."""
    return B * (16 * L + 8 * T) + 2 * 4 * P


def csrc_sha():
    """Hash of the kernel sources: ties the committed PMC summaries to the code they were taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "explainn_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def committed_counters(workload):
    """Per-kernel PMC counters of the training step (profiles/*_counters.json, written by
    tools/final_profile.sh from separate rocprofv3 --pmc passes; rocprofv3 cannot run inside this
    process).  Used only when they were taken on exactly these kernel sources and this workload:
    otherwise every figure derived from them is reported as null."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    try:
        for name in sorted(os.listdir(pdir)):
            if name.endswith("_counters.json"):
                with open(os.path.join(pdir, name)) as fh:
                    d = json.load(fh)
                if d.get("csrc_sha") == csrc_sha() and d.get("workload") == workload:
                    best = dict(d, file="profiles/" + name)
    except Exception:
        return None
    return best


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


def synthetic_batch(B, L, T, seed, device, n_frac=0.0):
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, 4, (B, L), generator=g)
    x = torch.zeros(B, 4, L).scatter_(1, idx[:, None, :], 1.0)
    if n_frac > 0:                                    # N bases = all-zero columns
        x = x * (torch.rand(B, 1, L, generator=g) >= n_frac)
    y = (torch.rand(B, T, generator=torch.Generator().manual_seed(seed + 1)) > 0.5).float()
    return x.to(device), y.to(device)


def cpu_baseline(w, budget_s=24.0):
    """The stock-PyTorch CPU restatement of the reference (oracle/torch_ref.py, pinned to the
    reference's outputs by tests/test_torch_ref_golden.py), timed on this box's host cores on a
    bounded sample of the same workload: whole batches until about half the budget is used with
    this job's core share, then the same with ONE thread -- the reference's own default
    (`--cpu-threads 1`, train.py:43-48)."""
    from oracle import torch_ref
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a 1-GPU box gives this job a 16-core share of the host (more threads only oversubscribe:
    # 256 threads measured 72 seq/s against 16 threads' several hundred)
    cores = min(cores, int(os.environ.get("EXPLAINN_CPU_THREADS", "16")))
    sd = torch_ref.init_state(w["U"], K, w["L"], w["T"], seed=0)
    x, y = synthetic_batch(w["B"], w["L"], w["T"], 1, "cpu")

    def timed(threads, budget, max_steps):
        torch.set_num_threads(threads)
        torch_ref.train_step(sd, x, y)                  # warm-up (allocations, oneDNN primitives)
        t0 = time.perf_counter()
        steps = 0
        while True:
            torch_ref.train_step(sd, x, y)
            steps += 1
            el = time.perf_counter() - t0
            if el >= budget or steps >= max_steps:
                break
        return steps, el

    steps, el = timed(cores, budget_s / 2, 20)
    out = {"value": round(steps * w["B"] / el, 1), "unit": "sequences/s", "cores": cores, "kind": "port",
           "sample": "%d train steps (fwd+BCE+bwd, dropout 0.3) of the %s workload, stock-PyTorch CPU "
                     "ops, %d threads, %.1f s" % (steps, w["text"].split(":")[0], cores, el)}
    s1, e1 = timed(1, budget_s / 2, 2)
    out["one_thread"] = {"value": round(s1 * w["B"] / e1, 1), "unit": "sequences/s", "cores": 1,
                         "sample": "%d train step(s), 1 thread (the reference's default -c 1), %.1f s" % (s1, e1)}
    torch.set_num_threads(cores)
    return out


def kernel_model(w, B):
    """Work of the heavy kernels of one training step, from the shapes alone:
    the resource that binds each and the amount of it, so that a live duration turns into a
    fraction of that resource's peak."""
    U, L, T = w["U"], w["L"], w["T"]
    Lo = L - K + 1
    n = Lo // 7
    nt = (K + 1) // 2
    return {
        # FLOPs of the UNPADDED problem (100 hidden channels, n pooled positions), priced against
        # the fp32 MFMA peak ("mfma_f32" = the yardstick, not the instruction).  The contractions run
        # on the bf16 MFMA with fp32 operands split exactly into three bf16 pieces: passA and the T.bit
        # part of passB have a bit matrix as one operand (3 piece products, 3/16 of the fp32 cost),
        # fc_fwd has two real operands (9 piece products, 9/16) -- fp32-equivalent results, so the
        # fraction of the fp32 peak may exceed what an fp32-MFMA kernel could reach
        "fc_fwd": ("mfma_f32", 2.0 * 100 * n * B * U),
        "passA": ("mfma_f32", 2.0 * 100 * n * B * U),
        "passB": ("mfma_f32", 2.0 * (100 + n) * n * B * U),
        # the filter bank as a GEMM on the bf16 matrix core (csrc/convpool.hip): 3 exact pieces x
        # ceil(k/4) k-steps of v_mfma_f32_32x32x16_bf16 per (32 sequences, 32 units, position); the
        # unit tiles are padded to whole groups of two (k <= 20)
        "conv_pool": ("mfma_bf16", 3.0 * ((K + 3) // 4) * ((B + 31) // 32) * (2 * ((U + 63) // 64)) * n * 7,
                      MFMA_BF16_32x32x16_CLK),
        # the filter gradient as a GEMM on the bf16 matrix core (csrc/bwd.hip): 3 exact pieces x
        # ceil(k/4) column tiles per (32 sequences, 16 units, window, position)
        "conv_bwd": ("mfma_bf16", 3.0 * ((K + 3) // 4) * ((B + 31) // 32) * ((U + 15) // 16) * n * 7,
                     MFMA_BF16_16x16x32_CLK),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="C2")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--n-frac", type=float, default=0.0,
                    help="fraction of N bases in the synthetic batch (robustness variant; the "
                         "headline is 0)")
    ap.add_argument("--skip-optimizer", action="store_true",
                    help="skip the secondary fwd+bwd+Adam timing (profiling runs: one leg only)")
    ap.add_argument("--skip-stage-times", action="store_true",
                    help="skip the per-kernel HIP-event leg (profiling runs)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (before torch is even imported)
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    global torch
    import torch
    w = WORKLOADS[args.workload]
    U, L, T = w["U"], w["L"], w["T"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if args.scaling == "strong":
        if w["G"] % world:
            raise SystemExit("global batch %d does not split over %d ranks" % (w["G"], world))
        B = w["G"] // world
    else:
        B = w["B"]
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
    eng = StepEngine(model, B, loss="binary")
    sync = (GradAllReduce(eng.flat_grad, split=eng.conv_grad_elements, force=force_sync)
            if dist is not None else None)
    # Default: ONE all-reduce of the flat buffer after the step.  EXPLAINN_BENCH_OVERLAP=1 switches to
    # the two-part reduction (FC/head gradients reduced under the filter-bank backward); in the
    # one-rank rehearsal its second collective cost more than the overlap saved
    overlap = os.environ.get("EXPLAINN_BENCH_OVERLAP", "0") == "1"
    P = eng.flat_grad.numel()
    x, y = synthetic_batch(B, L, T, 1000 + rank, dev, args.n_frac)

    def one_step(i, overlap=overlap):
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

    def timed_region(ov):
        """W untimed + exactly K timed steps between barrier + synchronize fences; host wall time
        and the HIP-event time on the launch stream."""
        for i in range(args.warmup):
            one_step(i, ov)
        fence()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_start = time.perf_counter()
        e0.record()
        for i in range(args.steps):
            one_step(args.warmup + i, ov)
        e1.record()
        fence()
        return time.perf_counter() - t_start, e0.elapsed_time(e1)

    # per-kernel durations, live: HIP events recorded by the library on the launch stream around
    # every stage of the step (explainn_stage_timing), averaged over a few steps.  A separate leg (the
    # events themselves cost time), run BEFORE the headline region: the device then enters the W
    # warm-up steps with its clocks and caches already up (with `--steps 20 --warmup 5` straight
    # after start-up the same step measured 0.2166 ms against 0.2024 after 200 warm-up steps)
    stage_us = None
    if not args.skip_stage_times and rank == 0 and world == 1:
        eng.ctx.stage_timing(True)
        acc, reps, skip = {}, 10, 20          # (the first steps after start-up run at ramping clocks)
        for i in range(reps + skip):
            one_step(10 ** 6 + i)
            t = eng.ctx.stage_times()
            if i >= skip:
                for k_, v in t.items():
                    acc[k_] = acc.get(k_, 0.0) + v
        eng.ctx.stage_timing(False)
        stage_us = {k_: round(v / reps, 2) for k_, v in acc.items()}
    wall, gpu_ms = timed_region(overlap)
    # N > 1: the other all-reduce schedule is timed as well (same K steps, after the headline
    # region) so that the choice between them rests on data from the node that ran this
    wall_alt = 0.0
    if sync is not None:
        wall_alt, _ = timed_region(not overlap)
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

    t = torch.tensor([wall, wall_opt, wall_alt], device=dev, dtype=torch.float64)
    n_ranks_seen = 1
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        n_ranks_seen = dist.get_world_size()
    wall, wall_opt, wall_alt = float(t[0].item()), float(t[1].item()), float(t[2].item())
    if rank == 0:
        seqs = B * world * args.steps
        step_gpu_s = gpu_ms / 1e3 / args.steps
        alg = algorithmic_bytes_per_step(B, L, T, P)
        achieved = alg / step_gpu_s / 1e9
        counters = committed_counters(args.workload) if (world == 1 and B == w["B"]) else None
        traffic = counters["traffic_bytes_per_step"] if counters else None
        copy_peak = measured_copy_peak(dev) if world == 1 else None
        if args.workload == "C2":
            metric = "sequences/sec (fwd+bwd), 200bp one-hot, 300 units, batch 1024"
        else:
            metric = "sequences/sec (fwd+bwd), %d bp one-hot, %d units, %d tasks" % (L, U, T)
        kernels = None
        dominant = None
        if stage_us:
            model_k = kernel_model(w, B)
            kernels = {}
            for name, us in sorted(stage_us.items(), key=lambda kv: -kv[1]):
                ent = {"us": us}
                if name in model_k:
                    kind, work = model_k[name][:2]
                    if kind == "mfma_f32":
                        ent.update(bound="mfma_f32", unpadded_gflop=round(work / 1e9, 3),
                                   mfma_frac=round(work / (us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4))
                    elif kind == "mfma_bf16":
                        # matrix-pipe time of the kernel's own MFMAs at the nominal clock / its duration
                        roof_us = work * model_k[name][2] / SIMDS / CLOCK_HZ * 1e6
                        ent.update(bound="mfma_bf16", mfma_instructions=int(work),
                                   mfma_roof_us=round(roof_us, 2), mfma_frac=round(roof_us / us, 4))
                    else:
                        ent.update(bound="lds+valu", lds_gbytes=round(work / 1e9, 3),
                                   lds_frac=round(work / (us * 1e-6) / 1e9 / LDS_PEAK_GBPS, 4))
                pc = (counters or {}).get("per_kernel", {}).get(name)
                if pc:
                    # vector pipe: measured busy time (SQ_ACTIVE_INST_VALU, quad-cycles summed over the
                    # waves) per SIMD at the nominal clock / the kernel's duration; the instruction count
                    # beside it gives the kernel's own cycles per vector instruction
                    if pc.get("SQ_ACTIVE_INST_VALU"):
                        ent["valu_frac"] = round(pc["SQ_ACTIVE_INST_VALU"] * 4 / SIMDS / CLOCK_HZ / (us * 1e-6), 4)
                        if pc.get("SQ_INSTS_VALU"):
                            ent["valu_clk_per_inst"] = round(pc["SQ_ACTIVE_INST_VALU"] * 4 / pc["SQ_INSTS_VALU"], 2)
                    if pc.get("SQ_INSTS_LDS"):
                        # LDS pipe: 4 array cycles per ds_read_b128 wave-instruction and CU (16 per SIMD)
                        ent["lds_issue_frac"] = round(pc["SQ_INSTS_LDS"] * 16.4 / SIMDS / CLOCK_HZ / (us * 1e-6), 4)
                    if pc.get("hbm_bytes"):
                        ent["hbm_MB"] = round(pc["hbm_bytes"] / 1e6, 1)
                        ent["hbm_frac"] = round(pc["hbm_bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
                kernels[name] = ent
            dominant = max(stage_us, key=stage_us.get)
        out = {
            "metric": metric,
            "value": round(seqs / wall, 1), "unit": "sequences/s", "n_gpus": world,
            "n_ranks_seen": n_ranks_seen,
            "steps": args.steps, "warmup": args.warmup,
            "legs": "per-kernel stage times (30 steps, the last 10 averaged), then W warm-up + K timed steps, then the optimizer leg",
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["text"] + ", train fwd + BCE + bwd (dropout 0.3)"
                                   + (", RCCL all-reduce of the flat gradient" if dist is not None else ""),
                       "cnn_units": U, "kernel_size": K, "sequence_length": L, "n_features": T,
                       "batch_per_gpu": B, "global_batch": B * world,
                       "parameters": P, "parallelism": "dp%d" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "traffic": traffic,
                         # what the memory side actually moved (SURVEY 8d "hbm_measured_GBps"):
                         # measured bytes / this run's GPU time; includes Infinity-Cache hits
                         "measured_GBps": round(traffic / step_gpu_s / 1e9, 1) if traffic else None,
                         "measured_frac": round(traffic / step_gpu_s / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
                         "traffic_note": ("bytes per step, FETCH_SIZE(x2)+WRITE_SIZE from %s (separate "
                                          "rocprofv3 --pmc passes on these kernel sources, csrc_sha %s)"
                                          % (counters["file"], counters["csrc_sha"])) if counters else
                                         "no committed PMC summary matches these kernel sources / this workload",
                         "copy_peak_GBps_measured_here": copy_peak,
                         "kernel": "train_step pipeline (all launches of one step; BatchNorm's batch "
                                   "statistics force the launches apart)",
                         "algorithmic_bytes_per_step": alg,
                         "gpu_ms_per_step_hip_events": round(step_gpu_s * 1e3, 4),
                         # every stage of the step, HIP events on the launch stream, microseconds;
                         # with the fraction of the resource that binds it (this path is
                         # ~600-900 FLOP/B: no kernel is HBM-bound on algorithmic bytes)
                         "dominant_kernel": dominant,
                         "issue_clk_per_inst": ISSUE_CLK_PER_INST,
                         "clock_note": "per-kernel roofs are priced at %.1f GHz; with the matrix pipe saturated "
                                       "the shader clock measured here is %.1f-%.1f GHz (tools/mfma_probe.hip)"
                                       % (CLOCK_HZ / 1e9, MFMA_SUSTAINED_CLOCK_GHZ[0], MFMA_SUSTAINED_CLOCK_GHZ[1]),
                         "kernels": kernels,
                         "kernel_sum_us": round(sum(stage_us.values()), 1) if stage_us else None},
        }
        out["with_optimizer"] = None if args.skip_optimizer else {
            "ms_per_step": round(wall_opt / args.steps * 1e3, 4),
            "value": round(seqs / wall_opt, 1), "unit": "sequences/s",
            "optimizer": "Adam(lr=0.003), one fused launch (csrc/adam.hip)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        if dist is not None:
            names = ("one all-reduce of the flat gradient after the step",
                     "FC/head gradients reduced under the filter-bank backward + a second reduce of the rest")
            out["allreduce"] = {
                "backend": "nccl (RCCL)", "bytes": 4 * P,
                "schedule": names[1 if overlap else 0], "ms_per_step": round(wall / args.steps * 1e3, 4),
                "other_schedule": names[0 if overlap else 1],
                "other_ms_per_step": round(wall_alt / args.steps * 1e3, 4),
                "select": "EXPLAINN_BENCH_OVERLAP=0|1",
                "rccl_debug": "run with NCCL_DEBUG=INFO to see the algorithm / protocol RCCL picked"}
        if args.n_frac > 0:
            out["data"] = "synthetic, %.3g N bases" % args.n_frac
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
