"""Adam for the hot loop (reference: architectures/__init__.py:463-464 builds torch.optim.Adam,
selene/__init__.py:291 steps it): the same update, all parameter tensors in ONE HIP launch
(csrc/adam.hip) instead of torch's string of foreach kernels.

`FusedAdam` IS a torch.optim.Adam: same constructor, same `state` / `param_groups` /
`state_dict()` layout (so the reference's checkpoints load and save unchanged); only `step()` is
replaced.  Option combinations the kernel does not implement (amsgrad, weight decay, maximize,
non-fp32 or non-HIP parameters) go through torch's own Adam step."""
import ctypes as C

import torch

from . import _lib


class FusedAdam(torch.optim.Adam):

    def _fusable(self, group, params):
        if group.get("amsgrad") or group.get("weight_decay", 0) != 0 or group.get("maximize"):
            return False
        if group.get("capturable") or group.get("differentiable"):
            return False
        if torch.is_tensor(group["lr"]):
            return False
        for p in params:
            if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                return False
            if p.grad.is_sparse or p.grad.dtype != torch.float32 or p.grad.device != p.device:
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = [(g, [p for p in g["params"] if p.grad is not None]) for g in self.param_groups]
        if not all(self._fusable(g, ps) for g, ps in groups):
            super().step()
            return loss
        lib = _lib.load()
        for group, params in groups:
            if not params:
                continue
            # one launch per distinct step count (all equal unless parameters were added later)
            by_step = {}
            for p in params:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st["step"].device.type != "cpu":
                    st["step"] = st["step"].cpu()
                st["step"] += 1
                by_step.setdefault(int(st["step"].item()), []).append(p)
            beta1, beta2 = group["betas"]
            for step, ps in by_step.items():
                n = len(ps)
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]

                def arr(ts):
                    return (C.c_void_p * n)(*[t.data_ptr() for t in ts])

                dev = ps[0].device
                with torch.cuda.device(dev):
                    _lib.check(lib.explainn_adam_step(
                        n, arr(ps), arr(grads), arr([self.state[p]["exp_avg"] for p in ps]),
                        arr([self.state[p]["exp_avg_sq"] for p in ps]),
                        (C.c_int64 * n)(*[p.numel() for p in ps]), step, float(group["lr"]),
                        float(beta1), float(beta2), float(group["eps"]),
                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return loss
