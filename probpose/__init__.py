"""Drop-in import path: ``probpose.model`` / ``backbone`` / ``head`` / ``codec`` /
``heatmap`` / ``util`` resolve to the MI355X-native implementations in
``probpose_pytorch_amd`` (same class names, constructor signatures, parameter
names and return structures as zir-vision/ProbPose_pytorch)."""
import importlib
import sys

for _name in ("model", "backbone", "head", "codec", "heatmap", "util", "inference", "frontend"):
    _mod = importlib.import_module("probpose_pytorch_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
