import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from explainn_amd import ExplaiNN
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m = ExplaiNN(300, 19, 200, 1).to(dev).eval(); m.validate_input = False
idx = torch.randint(0, 4, (B, 200))
x = torch.zeros(B, 4, 200).scatter_(1, idx[:, None, :], 1.0).to(dev)
with torch.no_grad():
    for _ in range(50): m(x)
torch.cuda.synchronize()
