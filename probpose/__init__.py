"""Drop-in import path: ``probpose.model`` / ``backbone`` / ``head`` / ``codec`` /
``heatmap`` / ``util`` resolve to the MI355X-native implementations in
``probpose_pytorch_amd`` (same class names, constructor signatures, parameter
names and return structures as zir-vision/ProbPose_pytorch)."""
import importlib
import sys

for _name in ("model", "backbone", "head", "codec", "heatmap", "util", "inference", "frontend", "metrics"):
    _mod = importlib.import_module("probpose_pytorch_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod

# the reference keeps its evaluation metrics in probpose/loss.py (compute_oks, pose_pck_accuracy,
# keypoint_pck_accuracy: loss.py:715-866); the training losses of that file are out of scope, the metrics resolve here
sys.modules[__name__ + ".loss"] = globals()["metrics"]
loss = globals()["metrics"]
