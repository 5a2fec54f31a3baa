"""CPU: the index math of csrc/conv_s2q.hip -- layer2.0's stride-2 pair with all 128 output channels in one workgroup
(tests/emul/conv_s2q_model.py mirrors its weight fragments, the parity-plane LDS fill with pitches 16 / 24, the fragment addresses
and the epilogue map) against a plain 3x3 stride-2 pad-1 convolution + ReLU and a 1x1 stride-2 convolution on integer data."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul'))
from conv_s2q_model import pack_weight_fragments, run_tile  # noqa: E402


@pytest.mark.parametrize('tile', [0, 1, 16 + 14, 2 * 16 + 15, 16 + 7])
def test_stride2_pair_tile_against_plain_convolutions(tile):
    rng = np.random.default_rng(100 + tile)
    B = 3
    x = rng.integers(-2, 3, size=(B, 64, 64, 64)).astype(np.float64)
    xb = np.zeros((B, 66, 66, 64))
    xb[:, 1:-1, 1:-1] = x
    w = rng.integers(-1, 2, size=(128, 3, 3, 64)).astype(np.float64)
    w2 = rng.integers(-1, 2, size=(128, 64)).astype(np.float64)
    b1 = rng.integers(-3, 4, size=128).astype(np.float64)
    b2 = rng.integers(-3, 4, size=128).astype(np.float64)
    img, y0, x0, main, ds = run_tile(xb, pack_weight_fragments(w.reshape(128, 576), w2), b1, b2, tile)
    for j in range(4):
        for c in range(16):
            oy, ox = y0 + j, x0 + c
            patch = xb[img, 2 * oy:2 * oy + 3, 2 * ox:2 * ox + 3]
            ref = np.maximum(np.einsum('nyxc,yxc->n', w, patch) + b1, 0.0)
            ref2 = w2 @ xb[img, 2 * oy + 1, 2 * ox + 1] + b2
            assert np.array_equal(main[j, c], ref), (img, oy, ox)
            assert np.array_equal(ds[j, c], ref2), (img, oy, ox)
