import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import explainn_oracle as orc
import test_gpu_properties as P
m = P._c2_model()
x = P._batch()
y = (torch.rand(P.B, P.T, generator=torch.Generator().manual_seed(3)) > 0.5).float().cuda()
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
lg1, g1 = P._grads(m, x, y)
m.load_state_dict(sd0)
perm = torch.randperm(P.B, generator=torch.Generator().manual_seed(4)).cuda()
lg2, g2 = P._grads(m, x[perm], y[perm])
names = [n for n, _ in m.named_parameters()]
t0 = time.time()
sdn = {k: v.cpu().numpy() for k, v in sd0.items()}
ref_logits, cache, _ = orc.forward(sdn, x.cpu().numpy(), training=True, return_cache=True, dtype=np.float64)
_, dl = orc.bce_with_logits(ref_logits, y.cpu().numpy().astype(np.float64))
gr = orc.backward(cache, dl)
print("oracle fp64 took %.1fs; logits err %.2e / permuted %.2e" % (time.time() - t0, np.abs(lg1.cpu().numpy() - ref_logits).max(), np.abs(lg2.cpu().numpy() - ref_logits[perm.cpu().numpy()]).max()))
for n, a, b in zip(names, g1, g2):
    r = gr[n].reshape(a.shape); sc = np.abs(r).max() + 1e-30
    ea = np.abs(a.cpu().numpy() - r); eb = np.abs(b.cpu().numpy() - r)
    print("%-20s scale %.2e  err(orig) %.2e  err(perm) %.2e" % (n, sc, ea.max() / sc, eb.max() / sc))
n = "linears.1.weight"; i = names.index(n)
r = gr[n]; ea = np.abs(g1[i].cpu().numpy() - r); eb = np.abs(g2[i].cpu().numpy() - r)
for arr, tag in ((ea, "orig"), (eb, "perm")):
    u = int(arr.argmax()); print(tag, "worst unit", u, "ours", (g1 if tag == "orig" else g2)[i][u].item(), "oracle", r[u], "gamma1", sdn["linears.1.weight"][u])
y2 = cache["y2"]; print("min |y2| %.2e  count(|y2|<1e-6)=%d" % (np.abs(y2).min(), (np.abs(y2) < 1e-6).sum()))
