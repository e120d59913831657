"""vbnn_amd -- MI355X-native VBLinear hot path (louissmit/VBNN's VBLinear.lua / mlp.lua),
hand-written HIP kernels behind the C ABI of include/vbnn_hip.h. See DESIGN.md."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
