"""CPU: the index math of csrc/conv_fb.hip -- layer1's BasicBlock in one launch (tests/emul/conv_fb_model.py mirrors its weight
fragments, LDS fill, fragment addresses, halo zeroing and epilogue maps) against a plain conv3x3 -> ReLU -> conv3x3 -> + x -> ReLU
on integer data (exact in float64), for a corner tile, an edge tile and an interior tile.  A check of the kernel's design, not a
product code path."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul'))
from conv_fb_model import pack_weight_fragments, run_tile  # noqa: E402


def _conv3x3(xp, w, b):
    """xp [H+2][W+2][C] zero-padded, w [N][3][3][C] -> [H][W][N]"""
    H, W = xp.shape[0] - 2, xp.shape[1] - 2
    out = np.zeros((H, W, w.shape[0])) + b
    for ky in range(3):
        for kx in range(3):
            out += xp[ky:ky + H, kx:kx + W] @ w[:, ky, kx].T
    return out


@pytest.mark.parametrize('tile', [0, 16 + 3, 2 * 16 + 5, 16 + 15, 13])
def test_fused_basic_block_tile_against_plain_convolutions(tile):
    rng = np.random.default_rng(tile)
    B = 3
    x = rng.integers(-2, 3, size=(B, 64, 64, 64)).astype(np.float64)
    xb = np.zeros((B, 66, 66, 64))
    xb[:, 1:-1, 1:-1] = x
    w1 = rng.integers(-1, 2, size=(64, 3, 3, 64)).astype(np.float64)
    w2 = rng.integers(-1, 2, size=(64, 3, 3, 64)).astype(np.float64)
    b1 = rng.integers(-3, 4, size=64).astype(np.float64)
    b2 = rng.integers(-3, 4, size=64).astype(np.float64)
    img, y0, x0, got = run_tile(xb, pack_weight_fragments(w1.reshape(64, 576)), b1, pack_weight_fragments(w2.reshape(64, 576)), b2, tile)
    t = np.maximum(_conv3x3(xb[img], w1, b1), 0.0)
    tp = np.zeros((66, 66, 64))
    tp[1:-1, 1:-1] = t
    ref = np.maximum(_conv3x3(tp, w2, b2) + x[img], 0.0)
    assert np.array_equal(got, ref[y0:y0 + 16, x0:x0 + 16]), (img, y0, x0)
