"""Drop-in `ExplaiNN` nn.Module whose forward/backward run as hand-written gfx950 kernels.

Mirrors the reference's module surface for this path (reference: explainn/architectures/__init__.py):
  * constructor `ExplaiNN(cnn_units, kernel_size, sequence_length, n_features=1, weights_file=None)`
    and `_options`                                                         (:44-67)
  * `state_dict()` keys/shapes `linears.{0,1,6,7,10,11}.*`, `final.*`      (:72-104)
  * `forward(x)`: x (B,4,L) fp32 one-hot -> logits (B,T); train mode uses batch statistics,
    updates the BatchNorm buffers and applies Dropout(0.3)                 (:109-114)
  * `model.linears[0].weight`, `model.linears(x_rep)`, `model.linears[:3](x_rep)`,
    `model.final(outs)` as test.py:148-160 / train.py:318-324 / selene/__init__.py:257 use them
  * `get_loss`, `get_metrics`, `get_optimizer`                             (:446-464)

The compute is in libexplainn_hip.so (include/explainn_hip.h); PyTorch only owns the device
memory, the stream and autograd bookkeeping.  There is no CPU or eager fallback: a model that is
not on a HIP device raises.
"""
import contextlib
import copy
import ctypes as C
import math
import weakref
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import _lib

FC_HIDDEN = 100
POOL = 7
DROPOUT_P = 0.3


# ----------------------------------------------------------------------------------------------
# parameter holders (positions 0,1,6,7,10,11 of `linears`; the other positions hold no state)
# ----------------------------------------------------------------------------------------------
def _uniform_fan_in_(weight, bias, fan_in):
    """torch's default Conv1d/Linear init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for both."""
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
    with torch.no_grad():
        weight.uniform_(-bound, bound)
        bias.uniform_(-bound, bound)


class _GroupedTaps(nn.Module):
    """weight/bias of a grouped Conv1d; holds state only (the kernels do the math)."""

    def __init__(self, out_channels, in_per_group, kernel_size):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_per_group, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        _uniform_fan_in_(self.weight, self.bias, in_per_group * kernel_size)

    def forward(self, *a, **k):
        raise RuntimeError("sub-layers are fused; call model(x), model.linears(x_rep) or "
                           "model.linears[:3](x_rep)")


class _BatchStats(nn.Module):
    """BatchNorm1d state: affine parameters and running statistics."""

    def __init__(self, channels):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    forward = _GroupedTaps.forward


class _Fused(nn.Module):
    """Stateless stage (exp, max-pool, flatten, ReLU, dropout) -- fused into the kernels."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what

    forward = _GroupedTaps.forward


class _UnitPrefix:
    """`model.linears[:3]` -- conv + BatchNorm + exp per position (test.py:159-160)."""

    def __init__(self, owner):
        self._owner = owner

    def __call__(self, x_rep):
        return self._owner._unit_activations(x_rep)


class _UnitStack(nn.Sequential):
    """The `linears` container: real parameters at the reference's indices, fused forward."""

    def _bind(self, owner):
        self.__dict__["_owner_ref"] = weakref.ref(owner)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_owner_ref", None)           # re-bound by the owner's __setstate__/__deepcopy__
        return state

    def _owner(self):
        owner = self.__dict__["_owner_ref"]()
        if owner is None:
            raise RuntimeError("owning ExplaiNN module is gone")
        return owner

    def forward(self, x_rep):
        return self._owner()._unit_outputs(x_rep)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            if idx == slice(None, 3, None) or idx == slice(0, 3, None):
                return _UnitPrefix(self._owner())
            raise NotImplementedError("only linears[:3] (conv+BN+exp) is exposed as a sub-stack")
        return super().__getitem__(idx)


class BaseCodes:
    """A batch handed over as base codes instead of an fp32 one-hot: `codes` is a
    (B,L) uint8 tensor on the model's device, 0..3 = A,C,G,T, 4 = N (sequence.encode_codes_many);
    reverse_complement=True makes the kernel read it as its reverse complement, so the strand
    augmentation of train.py:275-278 / predict.py:78-79 needs no second copy.  Accepted wherever
    the model takes `x`; a bare uint8 (B,L) tensor means BaseCodes(codes, False)."""

    def __init__(self, codes, reverse_complement=False):
        self.codes = codes
        self.reverse_complement = bool(reverse_complement)

    @property
    def shape(self):
        return self.codes.shape


VALIDATE_EVERY = 64      # deferred input validation: the sticky device flag is read every this many calls


class _Runtime:
    """Per-model device context; never copied or pickled with the module."""

    def __init__(self):
        self.ctx = None
        self.token = 0
        self.pending = None
        self.x_keep = None
        self.calls = 0           # forwards since the model was built (input validation schedule)
        self.soft_seen = False   # a validating call met a batch that was not one-hot
        self.cache_depth = 0     # nesting depth of eval_cache() scopes
        self.replica = None      # eval_replica(): second handle on the same tensors, own context
        self.side_stream = None  # the stream predict() runs the replica on

    def __deepcopy__(self, memo):
        return _Runtime()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()


class _TrainStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        ctx.model = model
        logits, ctx.token = model._launch_train(x)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        grads = ctx.model._launch_backward(dlogits, ctx.token)
        return (None, None) + tuple(grads)


class _Model(nn.Module):

    def load_weights(self, weight_file):
        """architectures/__init__.py:27-39: positional remap of a saved tensor list onto this
        module's keys.  Accepts both the (100U,n,1)/(U,100,1) layout of this class's own
        state_dict and the squeezed (100U,n)/(U,100) layout of the older Linear-based variant."""
        sd = torch.load(weight_file, map_location="cpu", weights_only=True)
        own = self.state_dict()
        keys = list(own.keys())
        remapped = OrderedDict()
        for key, v in zip(keys, sd.values()):
            if v.dim() == own[key].dim() - 1:
                v = v.unsqueeze(-1)
            elif v.dim() == own[key].dim() + 1 and v.shape[-1] == 1:
                v = v.squeeze(-1)
            remapped[key] = v
        self.load_state_dict(remapped)


class ExplaiNN(_Model):
    """ExplaiNN: explainable neural networks (MI355X-native forward/backward)."""

    def __init__(self, cnn_units, kernel_size, sequence_length, n_features=1, weights_file=None):
        super().__init__()
        self._options = {
            "cnn_units": cnn_units,
            "kernel_size": kernel_size,
            "sequence_length": sequence_length,
            "n_features": n_features,
            "weights_file": weights_file,
        }
        if cnn_units < 1 or n_features < 1 or kernel_size < 1:
            # torch's Conv1d/Linear constructors reject these in the reference
            raise ValueError("cnn_units, kernel_size and n_features must be positive")
        n = int(math.floor((sequence_length - kernel_size + 1) / float(POOL)))
        if n < 1:
            raise ValueError("sequence_length too short for kernel_size and MaxPool1d(7, 7)")
        self._n = n
        U = cnn_units
        self.linears = _UnitStack(
            _GroupedTaps(U, 4, kernel_size),                 # 0  Conv1d(4U->U, k, groups=U)
            _BatchStats(U),                                  # 1  BatchNorm1d(U)
            _Fused("exp"),                                   # 2
            _Fused("max_pool1d(7, 7)"),                      # 3
            _Fused("flatten"),                               # 4
            _Fused("unsqueeze(-1)"),                         # 5
            _GroupedTaps(FC_HIDDEN * U, n, 1),               # 6  per-unit Linear(n->100)
            _BatchStats(FC_HIDDEN * U),                      # 7  BatchNorm1d(100U)
            _Fused("relu"),                                  # 8
            _Fused("dropout(p=%g)" % DROPOUT_P),             # 9
            _GroupedTaps(U, FC_HIDDEN, 1),                   # 10 per-unit Linear(100->1)
            _BatchStats(U),                                  # 11 BatchNorm1d(U)
            _Fused("relu"),                                  # 12
            _Fused("flatten"),                               # 13
        )
        self.linears._bind(self)
        self.final = nn.Linear(U, n_features)
        self.dropout_p = DROPOUT_P
        # Input validation (is every column of x one-hot or all-zero?) happens on the device while
        # the batch is packed and raises a sticky flag.  True: the first two forwards read the flag
        # before computing (so that a soft batch is routed to the dense kernels), later ones
        # enqueue and return without touching the host; the flag is then read every
        # VALIDATE_EVERY-th call, by check_input(), and at the end of predict() / a validation pass.
        # "always": every call validates before computing (one host sync per forward).
        # False: never.
        self.validate_input = True
        # None: a batch that fails validation takes the dense kernels (the reference accepts any
        # float input); True: always dense; False: a batch that is not one-hot is an error
        self.dense_input = None
        self.grad_sync = None          # optional callable(flat_grad_tensor): multi-GPU all-reduce
        # rows [0:n) of the filter gradient are zeroed inside the backward kernel (what the hook of
        # selene/__init__.py:254-257, 509-515 does to the reference's gradient)
        self.freeze_top_n_filters = 0
        self._rt = _Runtime()
        if weights_file is not None:
            self.load_weights(weights_file)

    # -- module protocol ------------------------------------------------------------------
    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for key, val in self.__dict__.items():
            if key in ("_slots", "_ps_cache", "_vkey", "_bufs"):   # resolved against THIS object's modules
                continue
            new.__dict__[key] = _Runtime() if key == "_rt" else copy.deepcopy(val, memo)
        new.linears._bind(new)
        return new

    def __getstate__(self):
        state = self.__dict__.copy()
        for key in ("_slots", "_ps_cache", "_vkey", "_bufs"):     # ctypes tables: rebuilt on demand
            state.pop(key, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__["_rt"] = _Runtime()
        self.__dict__.pop("_slots", None); self.__dict__.pop("_ps_cache", None)
        self.linears._bind(self)

    def eval_replica(self):
        """A second handle on the SAME parameters and buffers (the very tensor objects) with its own
        device context, for eval-mode forwards that overlap this model's on another stream: a
        forward is four dependent launches, and two independent batches in flight fill the gaps
        between them (batch 1024: 18 -> 25 M sequences/s).  predict() runs the two strands of a chunk
        this way.  Forward only; the replica follows this model's mode and input settings at the
        time of the call."""
        r = self._rt.replica
        if r is None:
            r = self.__class__.__new__(self.__class__)
            for key, val in self.__dict__.items():
                if key in ("_slots", "_ps_cache", "_vkey", "_bufs", "_pver"):   # resolved per handle
                    continue
                r.__dict__[key] = val
            r.__dict__["_rt"] = _Runtime()
            self._rt.replica = r
        r.__dict__["training"] = self.training
        r.__dict__["validate_input"] = self.validate_input
        r.__dict__["dense_input"] = self.dense_input
        return r

    # -- plumbing -------------------------------------------------------------------------
    def _device(self):
        dev = self.final.weight.device
        if dev.type != "cuda":
            raise RuntimeError(
                "explainn_amd.ExplaiNN runs only on a HIP device (model is on %s): call "
                ".cuda()/.to('cuda'). There is no CPU fallback." % dev)
        return dev

    def _context(self, B, dev):
        o = self._options
        geom = (o["cnn_units"], o["kernel_size"], o["sequence_length"], o["n_features"])
        ctx = self._rt.ctx
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        if ctx is None or ctx.geom != geom or ctx.device != index or ctx.max_batch < B:
            if ctx is not None:
                torch.cuda.synchronize(dev)
                ctx.close()
            cap = max(B, ctx.max_batch if ctx is not None and ctx.geom == geom else 0)
            self._rt.ctx = _lib.Context(*geom, max_batch=cap, device=index)
        return self._rt.ctx

    def _tensors(self):
        sd = {k: v for k, v in self.named_parameters()}
        sd.update({k: v for k, v in self.named_buffers()})
        return sd

    def __setattr__(self, name, value):
        if name in ("final", "linears"):
            self.__dict__.pop("_slots", None)          # sub-module replaced: re-resolve the slots
        super().__setattr__(name, value)

    def _param_slots(self):
        """(C field, the owning module's parameter/buffer dict, name, dtype) for the 23 tensors of
        explainn_params, resolved once: the per-call check is then 23 dict lookups instead of
        walking named_parameters() or going through nn.Module.__getattr__."""
        slots = self.__dict__.get("_slots")
        if slots is None:
            slots = []
            for field in _lib.PARAM_FIELDS:
                path = _lib.PARAM_KEYS[field].split(".")
                mod = self
                for part in path[:-1]:
                    mod = mod[int(part)] if part.isdigit() else getattr(mod, part)
                store = mod._parameters if path[-1] in mod._parameters else mod._buffers
                slots.append((field, store, path[-1],
                              torch.int64 if field.endswith("nbt") else torch.float32))
            self.__dict__["_slots"] = slots
            self.__dict__.pop("_ps_cache", None)
        return slots

    def _params_struct(self, dev):
        """The explainn_params table (+ the tensors it points into, kept alive with it).  Cached
        until a parameter object is replaced (train.py:324 re-assigns the filter bank) or its
        storage moves (`.cuda()`, `.data = ...`); in-place optimiser updates keep it valid."""
        slots = self._param_slots()
        cache = self.__dict__.get("_ps_cache")
        if cache is not None and cache[0] == dev:
            _, ps, keep, ptrs = cache
            for (field, store, attr, want), t, ptr in zip(slots, keep, ptrs):
                cur = store[attr]
                if cur is not t or cur.data_ptr() != ptr:
                    break
            else:
                self._stamp_version(ps, keep)
                return ps, keep
        ps = _lib.Params()
        keep, ptrs = [], []
        for field, store, attr, want in slots:
            t = store[attr]
            if t.device != dev or t.dtype != want:
                raise RuntimeError("parameter %s must be %s on %s (is %s on %s)" % (
                    _lib.PARAM_KEYS[field], want, dev, t.dtype, t.device))
            if not t.is_contiguous():
                raise RuntimeError("parameter %s must be contiguous" % _lib.PARAM_KEYS[field])
            keep.append(t)
            ptrs.append(t.data_ptr())
            setattr(ps, field, ptrs[-1])
        self.__dict__["_ps_cache"] = (dev, ps, keep, ptrs)
        self.__dict__["_bufs"] = [t for (field, _, _, _), t in zip(slots, keep)
                                  if field.endswith(("_rm", "_rv", "_nbt"))]
        self.__dict__.pop("_vkey", None)
        self._stamp_version(ps, keep)
        return ps, keep

    def _stamp_version(self, ps, keep):
        """explainn_params.version: 0 ("unknown": the eval entry points rebuild their folded
        tables on every call) unless an eval_cache() scope is open; inside one, a counter that
        moves whenever any of the 23 tensors changed value as far as torch can tell (its per-tensor
        version counters; this package's own kernels, which write through raw pointers, bump
        them explicitly -- _touched()).  Writes through `.data` do NOT move torch's counters
        (`p.data.clamp_(0)`, selene/__init__.py:294), which is why caching is opt-in and scoped:
        the scope's owner promises not to do that inside it, or calls invalidate()."""
        if self._rt.cache_depth <= 0:
            ps.version = 0
            return
        vkey = tuple(t._version for t in keep)
        if vkey != self.__dict__.get("_vkey"):
            self.__dict__["_vkey"] = vkey
            self.__dict__["_pver"] = self.__dict__.get("_pver", 0) + 1
        ps.version = self.__dict__["_pver"]

    def invalidate(self):
        """Forget the cached eval-mode tables (after writing parameters or buffers through `.data`
        or raw pointers inside an eval_cache() scope)."""
        self.__dict__["_pver"] = self.__dict__.get("_pver", 0) + 1
        self.__dict__.pop("_vkey", None)

    @contextlib.contextmanager
    def eval_cache(self):
        """Scope in which eval-mode forwards reuse the folded tables (filter LUTs, BatchNorm folds,
        FC1 fragments) of the previous call instead of rebuilding them per batch -- predict(), a
        validation pass and the filter export open one around their loops.  Contract: inside the
        scope parameters and buffers change only through torch ops that bump tensor versions
        (optimiser steps, in-place ops on the tensor itself, load_state_dict, re-assignment) or
        through this package's kernels; after a `.data` write call invalidate()."""
        self._rt.cache_depth += 1
        if self._rt.cache_depth == 1:
            self.invalidate()            # whatever happened outside the scope is unknown
        try:
            yield self
        finally:
            self._rt.cache_depth -= 1

    def _touched(self):
        """The train-mode kernels updated the BatchNorm buffers in place (as torch does)."""
        torch.autograd.graph.increment_version(self.__dict__.get("_bufs") or list(self.buffers()))

    def _prep_input(self, x, dev):
        o = self._options
        if isinstance(x, BaseCodes) or (torch.is_tensor(x) and x.dtype == torch.uint8 and x.dim() == 2):
            bc = x if isinstance(x, BaseCodes) else BaseCodes(x)
            c = bc.codes
            if not torch.is_tensor(c) or c.dtype != torch.uint8 or c.dim() != 2 or \
                    c.shape[1] != o["sequence_length"]:
                raise RuntimeError("base codes must be a uint8 tensor of shape (B, %d)" % o["sequence_length"])
            if c.device != dev:
                raise RuntimeError("input is on %s but the model is on %s" % (c.device, dev))
            return BaseCodes(c.detach().contiguous(), bc.reverse_complement)
        if x.dim() != 3 or x.shape[1] != 4 or x.shape[2] != o["sequence_length"]:
            raise RuntimeError("expected input of shape (B, 4, %d), got %s" % (
                o["sequence_length"], tuple(x.shape)))
        if x.device != dev:
            raise RuntimeError("input is on %s but the model is on %s" % (x.device, dev))
        return x.detach().to(torch.float32).contiguous()

    def _validate_now(self):
        """Does THIS forward read the validation flag before computing?  (validate_input above.)"""
        v = self.validate_input
        if not v:
            return False
        # a model that has met a soft batch once keeps validating (and routing) per call
        return v == "always" or self._rt.soft_seen or self._rt.calls <= 2

    def _x_ptr(self, ctx, x, dev, validate=None):
        """Device pointer of the batch as the C ABI wants it: base codes are staged in the context
        and NULL ("the staged batch", include/explainn_hip.h) is returned.  An fp32 batch is, when
        this call validates (validate_input), staged too and its validation flag read BEFORE
        anything is computed from it: a batch that is not one-hot (the reference accepts any float
        tensor, architectures/__init__.py:111) goes through the dense kernels instead of being
        run as if its soft columns were N -- unless dense_input is False, which keeps the strict
        error.  Otherwise the batch is packed inside the forward itself and the flag stays sticky
        on the device for a later read (no host sync in the call)."""
        lib, h, stream = ctx.lib, ctx.handle, self._stream(dev)
        self._rt.calls += 1
        if isinstance(x, BaseCodes):
            _lib.check(lib.explainn_dense_input(h, 0))
            _lib.check(lib.explainn_stage_codes(h, x.codes.data_ptr(), x.codes.shape[0],
                                                int(x.reverse_complement), stream))
            return None
        if self.dense_input:
            _lib.check(lib.explainn_dense_input(h, 1))
            self._rt.x_keep = x
            return x.data_ptr()
        if self._validate_now() if validate is None else validate:
            _lib.check(lib.explainn_stage_onehot(h, x.data_ptr(), x.shape[0], stream))
            flags = C.c_int(0)
            _lib.check(lib.explainn_input_flags(h, C.byref(flags), stream))
            if flags.value & 1:
                self._rt.soft_seen = True
                if self.dense_input is False:
                    raise ValueError(
                        "input is not one-hot: every column of x must be one-hot (A,C,G,T) or all-zero "
                        "(N) as sequence.one_hot_encode produces (dense_input=False forbids the dense path)")
                _lib.check(lib.explainn_dense_input(h, 1))
                self._rt.x_keep = x            # the backward of a train forward reads x again
                return x.data_ptr()
            _lib.check(lib.explainn_dense_input(h, 0))
            return None
        _lib.check(lib.explainn_dense_input(h, 0))
        return x.data_ptr()

    def _stream(self, dev):
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def _check_flags(self, ctx, dev, x=None):
        """After a forward was enqueued: read the sticky flag when this call is one that reads it
        (a validating call already did for an fp32 batch -- base codes are only checked here -- and
        every VALIDATE_EVERY-th call settles the deferred ones)."""
        if not self.validate_input:
            return
        codes = isinstance(x, BaseCodes)
        if self.dense_input and not codes:
            return                                    # dense kernels: nothing is packed or flagged
        now = self._validate_now()
        if now and not codes:
            return                                    # _x_ptr read (and cleared) the flag already
        if now or self._rt.calls % VALIDATE_EVERY == 0:
            self.check_input()

    def check_input(self):
        """Read (one host sync) and clear the sticky validation flag; raise if any batch since the
        last read was not one-hot.  predict(), Trainer validation passes and the filter export call
        it after their loops."""
        if self.input_flags() & 1:
            raise ValueError(
                "input is not one-hot: base codes must be 0..4, and every column of an fp32 x must "
                "be one-hot (A,C,G,T) or all-zero (N) as sequence.one_hot_encode produces.  A batch "
                "since the last check was run with such columns read as N; for soft (real-valued) "
                "input set model.dense_input = True, or validate_input = 'always' to route per batch")

    def input_flags(self):
        """Synchronise and return (then clear) the device-side input validation flags."""
        ctx = self._rt.ctx
        if ctx is None:
            return 0
        dev = self._device()
        flags = C.c_int(0)
        _lib.check(ctx.lib.explainn_input_flags(ctx.handle, C.byref(flags), self._stream(dev)))
        return flags.value

    # -- forward / backward ---------------------------------------------------------------
    def forward(self, x):
        """Forward propagation of a batch: (B,4,L) one-hot -> (B,T) logits."""
        dev = self._device()
        if self.training:
            if torch.is_grad_enabled():
                return _TrainStep.apply(self, x, *self.parameters())
            return self._launch_train(x)[0]
        x = self._prep_input(x, dev)
        B = x.shape[0]
        if B == 0:                                   # torch returns an empty (0, T) tensor in eval
            return torch.empty(0, self._options["n_features"], device=dev, dtype=torch.float32)
        self._rt.token += 1        # eval overwrites the scratch of a train forward still awaiting backward
        ctx = self._context(B, dev)
        ps, keep = self._params_struct(dev)
        logits = torch.empty(B, self._options["n_features"], device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(ctx.lib.explainn_forward_eval(ctx.handle, self._x_ptr(ctx, x, dev), B, C.byref(ps),
                                                     logits.data_ptr(), self._stream(dev)))
            self._check_flags(ctx, dev, x)
        return logits

    def _launch_train(self, x, keep_mask=None):
        dev = self._device()
        x = self._prep_input(x, dev)
        B = x.shape[0]
        if B == 0:
            # what torch's BatchNorm1d raises for an empty batch in train mode
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             "[0, %d, 1]" % (FC_HIDDEN * self._options["cnn_units"]))
        ctx = self._context(B, dev)
        ps, keep = self._params_struct(dev)
        logits = torch.empty(B, self._options["n_features"], device=dev, dtype=torch.float32)
        mask = self._rt.pending if keep_mask is None else keep_mask
        self._rt.pending = None
        mask_ptr = None
        if mask is not None:
            mask = mask.to(device=dev, dtype=torch.uint8).contiguous()
            if mask.numel() != B * FC_HIDDEN * self._options["cnn_units"]:
                raise RuntimeError("keep-mask must have shape (B, 100*cnn_units)")
            mask_ptr = mask.data_ptr()
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if self.dropout_p > 0 else 0
        with torch.cuda.device(dev):
            _lib.check(ctx.lib.explainn_forward_train(
                ctx.handle, self._x_ptr(ctx, x, dev), B, C.byref(ps), mask_ptr, float(self.dropout_p),
                C.c_uint64(seed), logits.data_ptr(), self._stream(dev)))
            self._check_flags(ctx, dev, x)
        self._touched()
        self._rt.token += 1
        return logits, self._rt.token

    def set_dropout_mask(self, keep_mask):
        """Use `keep_mask` ((B,100U), nonzero = keep) instead of the generator for the NEXT
        train-mode forward (parity testing against a recorded reference mask)."""
        self._rt.pending = keep_mask

    def _launch_backward(self, dlogits, token):
        if token != self._rt.token:
            raise RuntimeError("backward of a stale forward: the fused kernels keep one training "
                               "step in flight per model (call backward before the next forward)")
        dev = self._device()
        ctx = self._rt.ctx
        ps, keep = self._params_struct(dev)
        params = list(self.parameters())
        flat = torch.empty(sum(p.numel() for p in params), device=dev, dtype=torch.float32)
        views, off = [], 0
        gs = _lib.Grads()
        for field, p in zip(_lib.GRAD_FIELDS, params):
            v = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            views.append(v)
            setattr(gs, field, v.data_ptr())
        dl = dlogits.to(torch.float32).contiguous()
        B = dl.shape[0]
        with torch.cuda.device(dev):
            _lib.check(ctx.lib.explainn_backward(ctx.handle, dl.data_ptr(), B, C.byref(ps),
                                                 C.byref(gs), int(self.freeze_top_n_filters),
                                                 self._stream(dev)))
        if self.grad_sync is not None:
            self.grad_sync(flat)
        return views

    # -- the façade test.py / interpret.py use -----------------------------------------------
    def _first_four_rows(self, x_rep):
        U = self._options["cnn_units"]
        if isinstance(x_rep, BaseCodes) or (torch.is_tensor(x_rep) and x_rep.dtype == torch.uint8):
            return x_rep
        if x_rep.dim() != 3 or x_rep.shape[1] not in (4, 4 * U):
            raise RuntimeError("expected the repeated input (B, 4*cnn_units, L) or (B, 4, L)")
        return x_rep[:, :4, :]

    def _unit_outputs(self, x_rep):
        """`model.linears(x.repeat(1,U,1))` -> per-unit outputs (B,U) (test.py:151)."""
        if self.training:
            raise NotImplementedError("linears(x) is an eval-mode export path; call model.eval()")
        dev = self._device()
        x = self._prep_input(self._first_four_rows(x_rep), dev)
        B = x.shape[0]
        self._rt.token += 1
        ctx = self._context(B, dev)
        ps, keep = self._params_struct(dev)
        outs = torch.empty(B, self._options["cnn_units"], device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(ctx.lib.explainn_unit_outputs(ctx.handle, self._x_ptr(ctx, x, dev), B, C.byref(ps),
                                                     outs.data_ptr(), self._stream(dev)))
            self._check_flags(ctx, dev, x)
        return outs

    def _unit_activations(self, x_rep):
        """`model.linears[:3](x.repeat(1,U,1))` -> exp(BN(conv)) (B,U,L-k+1) (test.py:159-160)."""
        if self.training:
            raise NotImplementedError("linears[:3](x) is an eval-mode export path; call model.eval()")
        dev = self._device()
        x = self._prep_input(self._first_four_rows(x_rep), dev)
        B = x.shape[0]
        o = self._options
        self._rt.token += 1
        ctx = self._context(B, dev)
        ps, keep = self._params_struct(dev)
        acts = torch.empty(B, o["cnn_units"], o["sequence_length"] - o["kernel_size"] + 1,
                           device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(ctx.lib.explainn_unit_activations(ctx.handle, self._x_ptr(ctx, x, dev), B, C.byref(ps),
                                                         acts.data_ptr(), self._stream(dev)))
            self._check_flags(ctx, dev, x)
        return acts


    # -- filter -> PWM export (interpret.py:363-459 on the device; see explainn_amd/interpret.py) --
    def _export_args(self, x, select):
        if self.training:
            raise NotImplementedError("the filter export runs in eval mode; call model.eval()")
        dev = self._device()
        x = self._prep_input(x, dev)
        B = x.shape[0]
        if select is not None:
            if select.shape != (B,) or select.device != dev:
                raise RuntimeError("select must be a (B,) tensor on the model's device")
            select = select.to(torch.uint8).contiguous()
        self._rt.token += 1
        return dev, x, B, select

    def filter_act_max(self, x, unit_max, select=None):
        """unit_max[u] (float32 [U], zeroed by the caller before the first batch) <- running max of
        the float16-rounded eval-mode activations of the selected sequences of this batch."""
        dev, x, B, select = self._export_args(x, select)
        ctx = self._context(B, dev)
        ps, keep = self._params_struct(dev)
        with torch.cuda.device(dev):
            _lib.check(ctx.lib.explainn_filter_act_max(
                ctx.handle, self._x_ptr(ctx, x, dev), B, C.byref(ps),
                select.data_ptr() if select is not None else None, unit_max.data_ptr(),
                self._stream(dev)))
            self._check_flags(ctx, dev, x)
        return unit_max

    def filter_sites(self, x, thresholds, site_total, pfm, select=None, site_cap=1000000,
                     want_hit=False):
        """Accumulate this batch's sites into pfm (int32 (U,k,4)) / site_total (int32 [U]); returns
        the (B,U) uint8 "has a site" matrix when want_hit."""
        dev, x, B, select = self._export_args(x, select)
        o = self._options
        U, k = o["cnn_units"], o["kernel_size"]
        for name, t, shape, dt in (("thresholds", thresholds, (U,), torch.float32),
                                   ("site_total", site_total, (U,), torch.int32),
                                   ("pfm", pfm, (U, k, 4), torch.int32)):
            if tuple(t.shape) != shape or t.dtype != dt or t.device != dev or not t.is_contiguous():
                raise RuntimeError("%s must be a contiguous %s tensor of shape %s on %s" % (
                    name, dt, shape, dev))
        ctx = self._context(B, dev)
        ps, keep = self._params_struct(dev)
        hit = torch.empty(B, U, device=dev, dtype=torch.uint8) if want_hit else None
        with torch.cuda.device(dev):
            _lib.check(ctx.lib.explainn_filter_sites(
                ctx.handle, self._x_ptr(ctx, x, dev), B, C.byref(ps),
                select.data_ptr() if select is not None else None, thresholds.data_ptr(),
                int(site_cap), site_total.data_ptr(), pfm.data_ptr(),
                hit.data_ptr() if hit is not None else None, self._stream(dev)))
            self._check_flags(ctx, dev, x)
        return hit


class PWM(nn.Module):
    """Frozen position-weight-matrix scanner, the reference's `PWM` (architectures/__init__.py:
    116-170): `pwms` (G,4,k) in ACGT row order, both strands, per-sequence max or sum of the window
    scores -> (B,G).  `conv1d.weight` / `conv1d.bias` keep the reference's state_dict layout; the
    scan itself is one HIP launch (csrc/pwm.hip), there is no CPU fallback."""

    def __init__(self, pwms, sequence_length, scoring="sum"):
        super().__init__()
        pwms = np.asarray(pwms, dtype=np.float32)
        groups, four, kernel_size = pwms.shape
        if four != 4:
            raise ValueError("pwms must have shape (n, 4, length)")
        self._options = {"groups": groups, "kernel_size": kernel_size,
                         "sequence_length": sequence_length, "scoring": scoring}
        self.conv1d = _GroupedTaps(groups, 4, kernel_size)
        self.conv1d.weight.data = torch.from_numpy(pwms.copy())
        self.conv1d.bias.data = torch.zeros(groups)
        for p in self.conv1d.parameters():
            p.requires_grad = False

    def forward(self, x):
        o = self._options
        w = self.conv1d.weight
        if w.device.type != "cuda":
            raise RuntimeError("explainn_amd.PWM runs only on a HIP device; call .cuda()")
        if x.dim() != 3 or x.shape[1] != 4 or x.shape[2] != o["sequence_length"] or x.device != w.device:
            raise RuntimeError("expected input of shape (B, 4, %d) on %s" % (o["sequence_length"], w.device))
        x = x.detach().to(torch.float32).contiguous()
        scores = torch.empty(x.shape[0], o["groups"], device=w.device, dtype=torch.float32)
        lib = _lib.load()
        with torch.cuda.device(w.device):
            _lib.check(lib.explainn_pwm_scan(
                x.data_ptr(), x.shape[0], o["sequence_length"], w.detach().contiguous().data_ptr(),
                o["groups"], o["kernel_size"],
                _lib.PWM_MAX if o["scoring"] == "max" else _lib.PWM_SUM, scores.data_ptr(),
                C.c_void_p(torch.cuda.current_stream(w.device).cuda_stream)))
        # the reference adds conv1d.bias (zeros by construction) to every window score
        windows = 1 if o["scoring"] == "max" else 2 * (o["sequence_length"] - o["kernel_size"] + 1)
        return scores.add_(self.conv1d.bias.detach() * windows)


# ----------------------------------------------------------------------------------------------
def get_loss(input_data="binary"):
    """architectures/__init__.py:446-456."""
    if input_data == "binary":
        return nn.BCEWithLogitsLoss()
    return nn.MSELoss()


def get_metrics(input_data="binary"):
    """architectures/__init__.py:458-461."""
    if input_data == "binary":
        from sklearn.metrics import average_precision_score, roc_auc_score
        return dict(aucROC=roc_auc_score, aucPR=average_precision_score)
    from scipy.stats import pearsonr, spearmanr
    return dict(Pearson=pearsonr, Spearman=spearmanr)


def get_optimizer(params, lr=1e-03):
    """architectures/__init__.py:463-464: Adam with torch's defaults.  The object returned is a
    torch.optim.Adam whose step() runs as one HIP launch (optim.FusedAdam)."""
    from .optim import FusedAdam
    return FusedAdam(params, lr=lr)
