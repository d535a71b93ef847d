"""Run a few train steps at the BASELINE.json secondary configurations (per-GPU shapes) and report
time per step; parity at these shapes is covered by tests/ at small unit counts (same n)."""
import gc, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from explainn_amd import ExplaiNN
from explainn_amd.engine import StepEngine
dev = torch.device("cuda", 0)
CONFIGS = [("C2", 300, 200, 1, 1024), ("C3", 300, 200, 50, 4096), ("C4/8", 300, 1000, 50, 1024),
           ("C5/8 fp32", 2000, 600, 164, 1024)]
ONLY = os.environ.get("ONLY")
for name, U, L, T, B in CONFIGS:
    if ONLY and not name.startswith(ONLY): continue
    torch.manual_seed(0)
    m = ExplaiNN(U, 19, L, T).to(dev).train(); m.validate_input = False
    eng = StepEngine(m, B)
    idx = torch.randint(0, 4, (B, L))
    x = torch.zeros(B, 4, L).scatter_(1, idx[:, None, :], 1.0).to(dev)
    y = (torch.rand(B, T) > 0.5).float().to(dev)
    gc.collect()            # a previous configuration's context is freed here, not inside the timed loop
    for _ in range(3): eng.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter(); K = 10
    for _ in range(K): eng.step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    ok = bool(torch.isfinite(eng.flat_grad).all() and torch.isfinite(eng.loss).all())
    print("%-10s U=%d L=%d T=%d B=%d n=%d: %.3f ms/step, %.0f seq/s, loss %.4f, finite=%s, scratch %.0f MB" % (
        name, U, L, T, B, m._n, dt * 1e3, B / dt, eng.loss.item(), ok, eng.ctx.scratch_bytes() / 2**20), flush=True)
    del eng, m; torch.cuda.empty_cache()
