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

    def _plan(self, gi, group, params):
        """Everything about a parameter group that does not change from step to step -- the ctypes
        pointer tables of parameters / gradients / moments, the sizes, the step tensors -- built
        once and reused while the same tensor objects are in place (a training loop keeps them)."""
        plan = self.__dict__.setdefault("_plans", {}).get(gi)
        if plan is not None and len(plan["params"]) == len(params) and \
                all(a is b for a, b in zip(plan["params"], params)) and \
                all(p.grad is g for p, g in zip(params, plan["grads"])):
            return plan
        steps, ms, vs = [], [], []
        for p in params:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif st["step"].device.type != "cpu":
                st["step"] = st["step"].cpu()
            steps.append(st["step"]); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
        n = len(params)
        grads = [p.grad for p in params]

        def arr(ts):
            return (C.c_void_p * n)(*[t.data_ptr() for t in ts])

        plan = {"params": list(params), "grads": grads, "steps": steps, "ms": ms, "vs": vs, "n": n,
                "p_arr": arr(params), "g_arr": arr(grads), "m_arr": arr(ms), "v_arr": arr(vs),
                "sizes": (C.c_int64 * n)(*[p.numel() for p in params]),
                "uniform": len({float(t) for t in steps}) == 1, "count": int(steps[0].item())}
        self._sync_steps()                # an older plan of this group may hold a newer count
        plan["count"] = int(steps[0].item())
        self._plans[gi] = plan
        return plan

    def _sync_steps(self):
        """Write the Python-side step counts into the state's `step` tensors."""
        for plan in self.__dict__.get("_plans", {}).values():
            if plan["uniform"]:
                for t in plan["steps"]:
                    t.fill_(float(plan["count"]))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = [(g, [p for p in g["params"] if p.grad is not None]) for g in self.param_groups]
        if not all(self._fusable(g, ps) for g, ps in groups):
            self._sync_steps()
            self.__dict__.pop("_plans", None)
            super().step()
            return loss
        lib = _lib.load()
        for gi, (group, params) in enumerate(groups):
            if not params:
                continue
            plan = self._plan(gi, group, params)
            # state tensors can be swapped under us (load_state_dict): the plan must still match
            st0 = self.state[params[0]]
            if st0["exp_avg"] is not plan["ms"][0] or st0["step"] is not plan["steps"][0] or \
                    any(not g.is_contiguous() for g in plan["grads"]):
                self._sync_steps()
                self._plans.pop(gi, None)
                plan = self._plan(gi, group, params)
                if any(not g.is_contiguous() for g in plan["grads"]):
                    self._sync_steps()
                    self.__dict__.pop("_plans", None)
                    super().step()
                    return loss
            # the step count lives in a Python int between steps; the per-parameter `step` tensors
            # (torch's state layout) are brought up to date when the state is read (_sync_steps)
            if plan["uniform"]:
                plan["count"] += 1
            else:
                torch._foreach_add_(plan["steps"], 1)
            beta1, beta2 = group["betas"]
            dev = params[0].device
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            with torch.cuda.device(dev):
                if plan["uniform"]:
                    _lib.check(lib.explainn_adam_step(
                        plan["n"], plan["p_arr"], plan["g_arr"], plan["m_arr"], plan["v_arr"],
                        plan["sizes"], plan["count"], float(group["lr"]), float(beta1),
                        float(beta2), float(group["eps"]), stream))
                    # the kernel wrote through raw pointers: move torch's version counters as an
                    # in-place torch update would (the model's eval-table cache keys on them)
                    torch.autograd.graph.increment_version(plan["params"])
                else:
                    # parameters added later carry their own step count: one launch per tensor
                    for i in range(plan["n"]):
                        one = lambda a: (C.c_void_p * 1)(a[i])   # noqa: E731
                        _lib.check(lib.explainn_adam_step(
                            1, one(plan["p_arr"]), one(plan["g_arr"]), one(plan["m_arr"]),
                            one(plan["v_arr"]), (C.c_int64 * 1)(plan["sizes"][i]),
                            int(plan["steps"][i].item()), float(group["lr"]), float(beta1),
                            float(beta2), float(group["eps"]), stream))
                    torch.autograd.graph.increment_version(plan["params"])
        return loss

    def load_state_dict(self, state_dict):
        self.__dict__.pop("_plans", None)
        return super().load_state_dict(state_dict)

    def add_param_group(self, param_group):
        if "_plans" in self.__dict__:
            self._sync_steps()
            self.__dict__.pop("_plans", None)
        return super().add_param_group(param_group)
