"""Eval-mode (predict) throughput at the C2 shape: forward only, fp32 one-hot and base-code input."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from explainn_amd import ExplaiNN
from explainn_amd.architectures import BaseCodes
dev = torch.device("cuda", 0)
U, L, T = 300, 200, 1
torch.manual_seed(0)
m = ExplaiNN(U, 19, L, T).to(dev).eval(); m.validate_input = False
for B in (100, 1024, 4096):
    idx = torch.randint(0, 4, (B, L))
    x = torch.zeros(B, 4, L).scatter_(1, idx[:, None, :], 1.0).to(dev)
    codes = idx.to(torch.uint8).to(dev)
    for name, inp in (("one-hot", x), ("codes", codes), ("codes rc", BaseCodes(codes, True))):
        # the folded tables are reused across calls only inside an eval_cache() scope (as predict() does)
        with torch.no_grad(), m.eval_cache():
            for _ in range(100): m(inp)          # (long enough for the clock to settle after the idle gap)
            torch.cuda.synchronize(); t0 = time.perf_counter(); K = 1000
            for _ in range(K): m(inp)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        print("eval B=%-5d %-9s %.3f ms/batch  %.2f M seq/s" % (B, name, dt * 1e3, B / dt / 1e6), flush=True)

# two independent batches in flight: the model and its eval_replica() (same tensors, own context)
# on two streams -- what predict() does with the two strands of a chunk
rep = m.eval_replica()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for B in (1024, 4096):
    codes = torch.randint(0, 4, (B, L)).to(torch.uint8).to(dev)
    with torch.no_grad(), m.eval_cache(), rep.eval_cache():
        for _ in range(50):
            with torch.cuda.stream(s1): m(codes)
            with torch.cuda.stream(s2): rep(codes)
        torch.cuda.synchronize(); t0 = time.perf_counter(); K = 1000
        for _ in range(K // 2):
            with torch.cuda.stream(s1): m(codes)
            with torch.cuda.stream(s2): rep(codes)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print("eval B=%-5d codes, two contexts on two streams  %.3f ms/batch  %.2f M seq/s" % (B, dt * 1e3, B / dt / 1e6), flush=True)

# the predict entry point on host data (N = 20000), reference default batch size
import numpy as np
from explainn_amd.predict import predict
from explainn_amd import sequence as sq
N = 20000
idx = np.random.default_rng(0).integers(0, 4, size=(N, L)).astype(np.uint8)
xh = sq.codes_to_one_hot(idx)
for name, inp in (("one-hot", xh), ("codes", idx)):
    predict(m, inp[:512], batch_size=100)
    t0 = time.perf_counter(); predict(m, inp, batch_size=100); dt = time.perf_counter() - t0
    print("predict() %-8s N=%d batch_size=100: %.1f ms, %.2f M seq/s (both strands)" % (name, N, dt * 1e3, N / dt / 1e6), flush=True)
