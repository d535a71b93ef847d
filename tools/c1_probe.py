#!/usr/bin/env python3
"""Small-batch step time (the reference's default shape: 100 units, 200 bp, batch 64 / 100)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from explainn_amd import ExplaiNN
from explainn_amd.engine import StepEngine
for U, B in ((100, 64), (100, 100), (300, 128)):
    m = ExplaiNN(U, 19, 200, 1).cuda().train(); m.validate_input = False
    eng = StepEngine(m, B)
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, 4, (B, 200), generator=g)
    x = torch.zeros(B, 4, 200).scatter_(1, idx[:, None, :], 1.0).cuda()
    y = (torch.rand(B, 1, generator=g) > 0.5).float().cuda()
    for i in range(20): eng.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(300): eng.step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    print("U=%d B=%d: %.4f ms/step" % (U, B, dt * 1e3))
