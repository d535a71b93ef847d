"""Per-step wall times (synchronised each step) at one configuration: looks for sporadic stalls."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from explainn_amd import ExplaiNN
from explainn_amd.engine import StepEngine
U, L, T, B = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (300, 200, 50, 4096))]
N = int(sys.argv[5]) if len(sys.argv) > 5 else 300
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = ExplaiNN(U, 19, L, T).to(dev).train(); m.validate_input = False
eng = StepEngine(m, B)
idx = torch.randint(0, 4, (B, L))
x = torch.zeros(B, 4, L).scatter_(1, idx[:, None, :], 1.0).to(dev)
y = (torch.rand(B, T) > 0.5).float().to(dev)
for _ in range(5): eng.step(x, y)
torch.cuda.synchronize()
ts = []
for _ in range(N):
    t0 = time.perf_counter(); eng.step(x, y); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts_sorted = sorted(ts)
print("U=%d L=%d T=%d B=%d: median %.3f ms, p99 %.3f, max %.3f at step %d; steps > 3x median: %s" % (
    U, L, T, B, ts_sorted[len(ts) // 2], ts_sorted[int(len(ts) * 0.99)], max(ts), ts.index(max(ts)),
    [(i, round(t, 2)) for i, t in enumerate(ts) if t > 3 * ts_sorted[len(ts) // 2]][:10]))
