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
        with torch.no_grad():
            for _ in range(10): m(inp)
            torch.cuda.synchronize(); t0 = time.perf_counter(); K = 200
            for _ in range(K): m(inp)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        print("eval B=%-5d %-9s %.3f ms/batch  %.2f M seq/s" % (B, name, dt * 1e3, B / dt / 1e6), flush=True)
