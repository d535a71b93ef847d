"""Fused training step of the hot loop (reference: selene/__init__.py:288-291) without Python in
the middle: one C-ABI call enqueues train forward + loss + backward on the current stream and
leaves the 14 gradients in ONE flat fp32 buffer (the buffer a multi-GPU run all-reduces).

`StepEngine` is what `bench.py` times and what `selene.Trainer` uses when its criterion is one of
the two losses `get_loss` can return; `model(x)` + `loss.backward()` (autograd path) computes the
same thing through separate forward/backward calls.
"""
import ctypes as C

import torch

from . import _lib


class StepEngine:
    def __init__(self, model, max_batch, loss="binary"):
        self.model = model
        self.dev = model._device()
        self.loss_kind = _lib.LOSS_BCE_WITH_LOGITS if loss == "binary" else _lib.LOSS_MSE
        self.max_batch = max_batch
        self.ctx = model._context(max_batch, self.dev)
        self.params = list(model.parameters())
        n = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(n, device=self.dev, dtype=torch.float32)
        self.views, off = [], 0
        self.gs = _lib.Grads()
        for field, p in zip(_lib.GRAD_FIELDS, self.params):
            v = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
            self.views.append(v)
            setattr(self.gs, field, v.data_ptr())
        T = model._options["n_features"]
        self.logits = torch.empty(max_batch, T, device=self.dev, dtype=torch.float32)
        self.loss = torch.zeros(1, device=self.dev, dtype=torch.float32)
        self.ps, self._keep = model._params_struct(self.dev)
        self.step_no = 0

    def refresh_params(self):
        """Call after parameters were re-assigned (not needed after in-place optimiser steps)."""
        self.ps, self._keep = self.model._params_struct(self.dev)

    def attach_grads(self):
        """Point every parameter's .grad at its slice of the flat buffer (for torch optimisers)."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    @property
    def conv_grad_elements(self):
        """Leading elements of flat_grad that the last part of the backward writes (conv_w, conv_b,
        bn1_w, bn1_b): the `split` of parallel.GradAllReduce."""
        return sum(p.numel() for p in self.params[:4])

    def step(self, x, y, seed=None, freeze_top_n_filters=0, grad_sync=None):
        """x (B,4,L) fp32 one-hot -- or base codes (uint8 (B,L) / architectures.BaseCodes) -- and
        y (B,T) fp32, both resident on the device.  Enqueues one train-mode forward + loss +
        backward; returns (logits view, loss tensor) without syncing.  With grad_sync (a
        parallel.GradAllReduce over flat_grad) the step also averages the gradients across ranks:
        the all-reduce of the FC/head gradients is enqueued as soon as they are final and runs
        under the filter-bank backward (explainn_train_step_fc / _conv)."""
        B = x.shape[0]
        if seed is None:
            self.step_no += 1
            seed = self.step_no * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF
        m = self.model
        # The context is re-resolved every step: an eval forward with a larger batch in between
        # (validation batches larger than the train batch, selene/__init__.py:334) makes the model
        # replace its context by a bigger one, and the one cached here would be closed.
        if B > self.max_batch:
            self.max_batch = B
            self.logits = torch.empty(B, self.logits.shape[1], device=self.dev, dtype=torch.float32)
        self.ctx = m._context(self.max_batch, self.dev)
        if not (torch.is_tensor(x) and x.dtype == torch.float32):
            x = m._prep_input(x, self.dev)
        stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        # the model's own validation schedule (architectures.ExplaiNN.validate_input): the first
        # steps read the flag before computing and route a soft batch to the dense kernels, as
        # forward() does; later steps enqueue without a host sync and the Trainer reads the sticky
        # flag periodically; dense_input=True goes straight to the dense kernels
        xp = m._x_ptr(self.ctx, x, self.dev)
        if grad_sync is None:
            _lib.check(self.ctx.lib.explainn_train_step(
                self.ctx.handle, xp, y.data_ptr(), B, C.byref(self.ps), C.byref(self.gs),
                self.loss_kind, float(m.dropout_p), C.c_uint64(seed), int(freeze_top_n_filters),
                self.logits.data_ptr(), self.loss.data_ptr(), stream))
        else:
            _lib.check(self.ctx.lib.explainn_train_step_fc(
                self.ctx.handle, xp, y.data_ptr(), B, C.byref(self.ps), C.byref(self.gs),
                self.loss_kind, float(m.dropout_p), C.c_uint64(seed), self.logits.data_ptr(),
                self.loss.data_ptr(), stream))
            work = grad_sync.start_tail()
            _lib.check(self.ctx.lib.explainn_train_step_conv(
                self.ctx.handle, B, C.byref(self.ps), C.byref(self.gs), int(freeze_top_n_filters),
                stream))
            grad_sync.finish(work)
        m._touched()
        m._rt.token += 1
        return self.logits[:B], self.loss
