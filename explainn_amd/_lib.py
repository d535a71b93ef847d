"""ctypes binding of libexplainn_hip.so (C ABI: include/explainn_hip.h).

The library is the product; there is no CPU or eager-PyTorch fallback.  `load()` raises if the
shared object is missing (run `python -c "import __graft_entry__ as g; g.build()"` or
`make -C explainn_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EXPLAINN_HIP_LIB", os.path.join(_HERE, "libexplainn_hip.so"))

OK, E_ARG, E_HIP, E_BATCH1, E_STATE, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
LOSS_BCE_WITH_LOGITS, LOSS_MSE = 0, 1
PWM_SUM, PWM_MAX = 0, 1

_fp = C.c_void_p          # device pointers travel as integers (tensor.data_ptr())

PARAM_FIELDS = (
    "conv_w", "conv_b", "bn1_w", "bn1_b", "bn1_rm", "bn1_rv",
    "fc1_w", "fc1_b", "bn2_w", "bn2_b", "bn2_rm", "bn2_rv",
    "fc2_w", "fc2_b", "bn3_w", "bn3_b", "bn3_rm", "bn3_rv",
    "final_w", "final_b", "bn1_nbt", "bn2_nbt", "bn3_nbt",
)
# C field -> reference state_dict key (architectures/__init__.py:72-104)
PARAM_KEYS = {
    "conv_w": "linears.0.weight", "conv_b": "linears.0.bias",
    "bn1_w": "linears.1.weight", "bn1_b": "linears.1.bias",
    "bn1_rm": "linears.1.running_mean", "bn1_rv": "linears.1.running_var",
    "fc1_w": "linears.6.weight", "fc1_b": "linears.6.bias",
    "bn2_w": "linears.7.weight", "bn2_b": "linears.7.bias",
    "bn2_rm": "linears.7.running_mean", "bn2_rv": "linears.7.running_var",
    "fc2_w": "linears.10.weight", "fc2_b": "linears.10.bias",
    "bn3_w": "linears.11.weight", "bn3_b": "linears.11.bias",
    "bn3_rm": "linears.11.running_mean", "bn3_rv": "linears.11.running_var",
    "final_w": "final.weight", "final_b": "final.bias",
    "bn1_nbt": "linears.1.num_batches_tracked", "bn2_nbt": "linears.7.num_batches_tracked",
    "bn3_nbt": "linears.11.num_batches_tracked",
}
GRAD_FIELDS = ("conv_w", "conv_b", "bn1_w", "bn1_b", "fc1_w", "fc1_b", "bn2_w", "bn2_b",
               "fc2_w", "fc2_b", "bn3_w", "bn3_b", "final_w", "final_b")

EXPORTS = (
    "explainn_create", "explainn_destroy", "explainn_last_error", "explainn_scratch_bytes",
    "explainn_forward_eval", "explainn_forward_train", "explainn_backward", "explainn_loss_grad",
    "explainn_train_step", "explainn_unit_outputs", "explainn_unit_activations",
    "explainn_input_flags", "explainn_filter_act_max", "explainn_filter_sites",
    "explainn_pwm_scan", "explainn_stage_codes", "explainn_adam_step",
    "explainn_train_step_fc", "explainn_train_step_conv",
    "explainn_stage_onehot", "explainn_dense_input",
    "explainn_stage_timing", "explainn_stage_count", "explainn_stage_name", "explainn_stage_times",
    "explainn_debug_keep_bits",
)


class Params(C.Structure):
    _fields_ = [(f, _fp) for f in PARAM_FIELDS] + [("version", C.c_uint64)]


class Grads(C.Structure):
    _fields_ = [(f, _fp) for f in GRAD_FIELDS]


class ExplainnError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExplainnError(
            "libexplainn_hip.so is not built (%s). explainn_amd has no CPU fallback: build it "
            "with `make -C explainn_amd/csrc` (hipcc, gfx950)." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    ctx = C.c_void_p
    pp, gp = C.POINTER(Params), C.POINTER(Grads)
    lib.explainn_create.argtypes = [C.POINTER(ctx), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.explainn_create.restype = C.c_int
    lib.explainn_destroy.argtypes = [ctx]
    lib.explainn_destroy.restype = None
    lib.explainn_last_error.argtypes = []
    lib.explainn_last_error.restype = C.c_char_p
    lib.explainn_scratch_bytes.argtypes = [ctx]
    lib.explainn_scratch_bytes.restype = C.c_int64
    lib.explainn_forward_eval.argtypes = [ctx, _fp, C.c_int, pp, _fp, _fp]
    lib.explainn_forward_eval.restype = C.c_int
    lib.explainn_forward_train.argtypes = [ctx, _fp, C.c_int, pp, _fp, C.c_float, C.c_uint64, _fp, _fp]
    lib.explainn_forward_train.restype = C.c_int
    lib.explainn_backward.argtypes = [ctx, _fp, C.c_int, pp, gp, C.c_int, _fp]
    lib.explainn_backward.restype = C.c_int
    lib.explainn_loss_grad.argtypes = [ctx, C.c_int, _fp, _fp, C.c_int, _fp, _fp, _fp]
    lib.explainn_loss_grad.restype = C.c_int
    lib.explainn_train_step.argtypes = [ctx, _fp, _fp, C.c_int, pp, gp, C.c_int, C.c_float,
                                        C.c_uint64, C.c_int, _fp, _fp, _fp]
    lib.explainn_train_step.restype = C.c_int
    lib.explainn_train_step_fc.argtypes = [ctx, _fp, _fp, C.c_int, pp, gp, C.c_int, C.c_float,
                                           C.c_uint64, _fp, _fp, _fp]
    lib.explainn_train_step_fc.restype = C.c_int
    lib.explainn_train_step_conv.argtypes = [ctx, C.c_int, pp, gp, C.c_int, _fp]
    lib.explainn_train_step_conv.restype = C.c_int
    lib.explainn_unit_outputs.argtypes = [ctx, _fp, C.c_int, pp, _fp, _fp]
    lib.explainn_unit_outputs.restype = C.c_int
    lib.explainn_unit_activations.argtypes = [ctx, _fp, C.c_int, pp, _fp, _fp]
    lib.explainn_unit_activations.restype = C.c_int
    lib.explainn_filter_act_max.argtypes = [ctx, _fp, C.c_int, pp, _fp, _fp, _fp]
    lib.explainn_filter_act_max.restype = C.c_int
    lib.explainn_filter_sites.argtypes = [ctx, _fp, C.c_int, pp, _fp, _fp, C.c_int, _fp, _fp, _fp, _fp]
    lib.explainn_filter_sites.restype = C.c_int
    lib.explainn_stage_codes.argtypes = [ctx, _fp, C.c_int, C.c_int, _fp]
    lib.explainn_stage_codes.restype = C.c_int
    lib.explainn_adam_step.argtypes = [C.c_int, C.POINTER(_fp), C.POINTER(_fp), C.POINTER(_fp),
                                       C.POINTER(_fp), C.POINTER(C.c_int64), C.c_int64, C.c_double,
                                       C.c_double, C.c_double, C.c_double, _fp]
    lib.explainn_adam_step.restype = C.c_int
    lib.explainn_pwm_scan.argtypes = [_fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]
    lib.explainn_pwm_scan.restype = C.c_int
    lib.explainn_stage_onehot.argtypes = [ctx, _fp, C.c_int, _fp]
    lib.explainn_stage_onehot.restype = C.c_int
    lib.explainn_dense_input.argtypes = [ctx, C.c_int]
    lib.explainn_dense_input.restype = C.c_int
    lib.explainn_stage_timing.argtypes = [ctx, C.c_int]
    lib.explainn_stage_timing.restype = C.c_int
    lib.explainn_stage_count.argtypes = []
    lib.explainn_stage_count.restype = C.c_int
    lib.explainn_stage_name.argtypes = [C.c_int]
    lib.explainn_stage_name.restype = C.c_char_p
    lib.explainn_stage_times.argtypes = [ctx, C.POINTER(C.c_float), C.c_int]
    lib.explainn_stage_times.restype = C.c_int
    lib.explainn_debug_keep_bits.argtypes = [ctx, C.c_int, _fp, _fp]
    lib.explainn_debug_keep_bits.restype = C.c_int
    lib.explainn_input_flags.argtypes = [ctx, C.POINTER(C.c_int), _fp]
    lib.explainn_input_flags.restype = C.c_int
    _lib = lib
    return lib


def check(rc):
    """Map a C return code to the exception the reference path would raise."""
    if rc == OK:
        return
    msg = load().explainn_last_error().decode("utf-8", "replace")
    if rc == E_BATCH1:
        # torch.nn.BatchNorm1d raises ValueError for a single value per channel in train mode
        raise ValueError(msg)
    raise ExplainnError("libexplainn_hip: %s (code %d)" % (msg, rc))


class Context:
    """Owner of one explainn_ctx (device scratch for a fixed model geometry and max batch)."""

    def __init__(self, cnn_units, kernel_size, sequence_length, n_features, max_batch, device):
        self.lib = load()
        self.geom = (cnn_units, kernel_size, sequence_length, n_features)
        self.max_batch = max_batch
        self.device = device
        h = C.c_void_p()
        check(self.lib.explainn_create(C.byref(h), cnn_units, kernel_size, sequence_length,
                                       n_features, max_batch, device))
        self.handle = h

    def stage_timing(self, enable=True):
        check(self.lib.explainn_stage_timing(self.handle, int(bool(enable))))

    def stage_times(self):
        """{stage name: microseconds} of the last training step (device-synchronising)."""
        n = self.lib.explainn_stage_count()
        buf = (C.c_float * n)()
        check(self.lib.explainn_stage_times(self.handle, buf, n))
        return {self.lib.explainn_stage_name(i).decode(): float(buf[i]) for i in range(n) if buf[i] >= 0}

    def scratch_bytes(self):
        return int(self.lib.explainn_scratch_bytes(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.explainn_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
