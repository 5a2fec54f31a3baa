"""CPU: the index math of csrc/conv_s2.hip (tests/emul/conv_s2_model.py mirrors its LDS fill, fragment addresses, K order and
tile mapping) against a plain 3x3 stride-2 pad-1 convolution + 1x1 stride-2 downsample, for the three geometries the
regressor runs it in.  A check of the kernel's design, not a product code path."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emul'))
from conv_s2_model import run_tile  # noqa: E402


@pytest.mark.parametrize('So,Cin,m_tile,n_blk', [(32, 64, 5, 64), (16, 128, 3, 128), (8, 128, 1, 0)])
def test_parity_plane_decomposition_and_lds_addressing(So, Cin, m_tile, n_blk):
    rng = np.random.default_rng(So)
    B, Si, N = 4, 2 * So, 2 * Cin
    x = rng.integers(-3, 4, size=(B, Si, Si, Cin)).astype(np.float64)
    xp = np.zeros((B, Si + 2, Si + 2, Cin))
    xp[:, 1:-1, 1:-1] = x
    w = rng.integers(-2, 3, size=(N, 3, 3, Cin)).astype(np.float64)        # [n][ky][kx][c] = the kernel's K order
    w2 = rng.integers(-2, 3, size=(N, Cin)).astype(np.float64)
    out = run_tile(xp, w.reshape(N, 9 * Cin), w2, m_tile, n_blk, So, So == 8)
    assert len(out) == 128
    for (img, oy, ox), (main, ds) in out.items():
        patch = xp[img, 2 * oy:2 * oy + 3, 2 * ox:2 * ox + 3]               # padded rows 2 oy + ky
        ref = np.einsum('nyxc,yxc->n', w[n_blk:n_blk + 64], patch)
        ref2 = w2[n_blk:n_blk + 64] @ xp[img, 2 * oy + 1, 2 * ox + 1]
        assert np.array_equal(main, ref), (img, oy, ox)
        assert np.array_equal(ds, ref2), (img, oy, ox)
    imgs = {k[0] for k in out}
    if So == 8:
        assert imgs == {2 * m_tile, 2 * m_tile + 1} and {(k[1], k[2]) for k in out} == {(a, b) for a in range(8) for b in range(8)}
    else:
        assert len(imgs) == 1
