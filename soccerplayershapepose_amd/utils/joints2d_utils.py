"""Mirror of `PR/utils/joints2d_utils.py`."""


def undo_keypoint_normalisation(normalised_keypoints, img_wh):
    """[-1, 1] -> pixel space (`PR/utils/joints2d_utils.py:5-10`); works on tensors and arrays."""
    return (normalised_keypoints + 1) * (img_wh / 2.0)
