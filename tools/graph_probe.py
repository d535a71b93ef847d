"""Does hipGraph replay of the step (EXPLAINN_GRAPH, non-default stream) beat direct launches?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from explainn_amd import ExplaiNN
from explainn_amd.engine import StepEngine
U, L, T, B = [int(v) for v in sys.argv[1:5]]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = ExplaiNN(U, 19, L, T).to(dev).train(); m.validate_input = False
eng = StepEngine(m, B)
idx = torch.randint(0, 4, (B, L))
x = torch.zeros(B, 4, L).scatter_(1, idx[:, None, :], 1.0).to(dev)
y = (torch.rand(B, T) > 0.5).float().to(dev)
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
for _ in range(20): eng.step(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter(); K = 500
for _ in range(K): eng.step(x, y)
torch.cuda.synchronize()
print("U=%d L=%d T=%d B=%d EXPLAINN_GRAPH=%s: %.4f ms/step" % (U, L, T, B, os.environ.get("EXPLAINN_GRAPH", "1"), (time.perf_counter() - t0) / K * 1e3))
