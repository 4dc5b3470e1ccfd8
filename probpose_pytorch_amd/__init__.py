"""probpose_pytorch_amd -- MI355X-native ProbPose forward + decode path.

Python host mirroring the reference's ``probpose.model`` / ``backbone`` /
``head`` / ``codec`` / ``heatmap`` / ``util`` API (same class names,
constructor signatures, parameter names and return structures); all
arithmetic on the hot path runs in hand-written HIP kernels for gfx950 behind
the C ABI of ``include/probpose_hip.h`` (``lib/libprobpose_hip.so``).  There
is no CPU fallback: without the built extension or without a gfx950 device
the hot-path calls raise.
"""
from .util import to_numpy  # noqa: F401
from .codec import ArgMaxProbMap, Codec, ProbMap  # noqa: F401
from .heatmap import get_heatmap_expected_value  # noqa: F401

__all__ = ["to_numpy", "Codec", "ProbMap", "ArgMaxProbMap", "get_heatmap_expected_value"]


def __getattr__(name):  # lazy: the nn.Module side pulls in the engine
    if name in ("ProbPoseModel",):
        from .model import ProbPoseModel
        return ProbPoseModel
    if name in ("ScratchViTBackbone", "RadioBackbone"):
        from . import backbone
        return getattr(backbone, name)
    if name == "ProbMapHead":
        from .head import ProbMapHead
        return ProbMapHead
    raise AttributeError(name)
