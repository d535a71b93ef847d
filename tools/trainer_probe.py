"""Steps per second of the drop-in Trainer loop (selene.Trainer.train) at the reference's default
shape -- 100 units, batch 100 -- including the DataLoader, the host-to-device copies, the fused step,
Adam and the per-step loss.item() the reference's loop does."""
import sys, os, time, tempfile, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.utils.data import DataLoader, TensorDataset
from explainn_amd import ExplaiNN, get_loss, get_metrics, get_optimizer
from explainn_amd.selene import Trainer
from explainn_amd.train import _get_data_loader
U, L, B, N = 100, 200, 100, 20000
rng = np.random.default_rng(0)
idx = rng.integers(0, 4, size=(N, L))
x = np.zeros((N, 4, L), dtype=np.float32); x[np.arange(N)[:, None], idx, np.arange(L)[None, :]] = 1
y = (rng.random((N, 1)) > 0.5).astype(np.float32)
if os.environ.get("STOCK_LOADER") == "1":        # the reference's loader, for comparison
    loaders = {"train": DataLoader(TensorDataset(torch.Tensor(x), torch.Tensor(y)), B, shuffle=True),
               "validation": DataLoader(TensorDataset(torch.Tensor(x[:1000]), torch.Tensor(y[:1000])), B)}
else:
    loaders = {"train": _get_data_loader(x, y, B, shuffle=True),
               "validation": _get_data_loader(x[:1000], y[:1000], B)}
m = ExplaiNN(U, 19, L, 1)
out = tempfile.mkdtemp()
tr = Trainer(m, loaders, get_loss("binary"), get_metrics("binary"), get_optimizer(m.parameters(), 0.003),
             max_steps=10 ** 9, patience=10 ** 9, report_stats_every_n_steps=10 ** 9, output_dir=out,
             use_cuda=True, logging_verbosity=0)
for s in range(1, 51): tr.step = s; tr.train()
torch.cuda.synchronize(); t0 = time.perf_counter(); K = 1000
pr = cProfile.Profile() if len(sys.argv) > 1 else None
if pr: pr.enable()
for s in range(51, 51 + K): tr.step = s; tr.train()
if pr: pr.disable()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print("Trainer.train: %.3f ms/step, %.0f steps/s, %.0f seq/s (U=%d, batch %d)" % (dt * 1e3, 1 / dt, B / dt, U, B))
if pr: pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
