"""Mirror of the reference's ``probpose/util.py``."""
from typing import TypedDict

import numpy as np
from torch import Tensor


def to_numpy(tensor) -> np.ndarray:
    """Device -> host crossing (reference util.py:6-12).  ndarray passes through."""
    if isinstance(tensor, np.ndarray):
        return tensor
    return tensor.detach().cpu().numpy()


class ProbPoseGroundTruth(TypedDict):
    """Reference util.py:15-21 (training-side record; kept for import compatibility)."""
    heatmaps: np.ndarray
    in_image: np.ndarray
    keypoints_visible: np.ndarray
    keypoints_visibility: np.ndarray
