"""CPU: the index math of csrc/conv_rs.hip -- layer2's 3x3 stride-1 convolutions with a 64-channel block's weights in registers
(tests/emul/conv_rs_model.py mirrors its fragment packing, the two-sub-image LDS fill, the fragment addresses and the epilogue map)
against a plain 3x3 pad-1 convolution on integer data."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul'))
from conv_rs_model import pack_weight_fragments, run_tile  # noqa: E402


@pytest.mark.parametrize('tile,nb', [(0, 0), (7, 1), (8 + 3, 0), (2 * 8 + 6, 1)])
def test_register_stationary_3x3_tile_against_a_plain_convolution(tile, nb):
    rng = np.random.default_rng(200 + tile)
    B = 3
    x = rng.integers(-2, 3, size=(B, 32, 32, 128)).astype(np.float64)
    xb = np.zeros((B, 34, 34, 128))
    xb[:, 1:-1, 1:-1] = x
    w = rng.integers(-1, 2, size=(128, 3, 3, 128)).astype(np.float64)
    bias = rng.integers(-3, 4, size=128).astype(np.float64)
    img, y0, x0, got = run_tile(xb, pack_weight_fragments(w.reshape(128, 1152)), bias, tile, nb)
    for oy in range(8):
        for ox in range(16):
            patch = xb[img, y0 + oy:y0 + oy + 3, x0 + ox:x0 + ox + 3]
            ref = np.einsum('nyxc,yxc->n', w[64 * nb:64 * nb + 64], patch) + bias[64 * nb:64 * nb + 64]
            assert np.array_equal(got[oy, ox], ref), (img, oy, ox)
