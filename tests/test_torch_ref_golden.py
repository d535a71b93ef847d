"""Pin the stock-torch CPU restatement (the cpu_baseline of bench.py) to the golden vectors."""
import numpy as np
import torch

from oracle import torch_ref


def test_torch_ref_matches_reference(golden):
    g = golden
    sd = {k: torch.from_numpy(np.array(v)) for k, v in g.sd().items()}
    x = torch.from_numpy(g.onehot())
    y = torch.from_numpy(g.targets().astype(np.float32))
    with torch.no_grad():
        ev = torch_ref.forward(sd, x, False)
    assert np.abs(ev.numpy() - g.z["eval/logits"]).max() < 1e-5
    loss, logits, grads = torch_ref.train_step(sd, x, y, loss=g.loss_kind, p=0.0)
    assert np.abs(logits.numpy() - g.z["train0/logits"]).max() < 1e-5
    ref = g.group("train0/grad/")
    gw = grads["linears.0.weight"].numpy()
    assert np.abs(gw - ref["linears.0.weight"]).max() < 1e-5 * max(1, np.abs(gw).max())
