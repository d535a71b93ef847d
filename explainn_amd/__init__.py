"""explainn_amd -- MI355X-native ExplaiNN forward/backward over one-hot DNA.

Drop-in for the hot path of oriolfornes/ExplaiNN: `ExplaiNN` (explainn/architectures/__init__.py),
the `Trainer` step loop (explainn/selene/__init__.py) and the `train._train` / `predict` entry
points.  The compute lives in `libexplainn_hip.so` (hand-written HIP for gfx950, C ABI in
include/explainn_hip.h); importing this package never falls back to a CPU implementation.
"""
from .architectures import ExplaiNN, PWM, get_loss, get_metrics, get_optimizer  # noqa: F401

__all__ = ["ExplaiNN", "PWM", "get_loss", "get_metrics", "get_optimizer"]
