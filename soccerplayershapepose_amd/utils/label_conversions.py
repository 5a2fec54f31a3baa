"""Mirror of `PR/utils/label_conversions.py` (the two functions the proxy builder uses).  Host-side numpy:
this is input synthesis in front of the hot path, not part of the timed path."""
import numpy as np

from ..synthetic import gaussian_heatmaps


def convert_2Djoints_to_gaussian_heatmaps(joints2D, img_wh, std=4):
    """`PR/utils/label_conversions.py:58-87`."""
    return gaussian_heatmaps(np.asarray(joints2D), img_wh, std)


def convert_multiclass_to_binary_labels(multiclass_labels):
    """`PR/utils/label_conversions.py:39-46`."""
    return (np.asarray(multiclass_labels) != 0).astype(np.asarray(multiclass_labels).dtype)
