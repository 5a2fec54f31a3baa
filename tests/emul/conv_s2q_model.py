"""Index-math model of csrc/conv_s2q.hip (test infrastructure) -- layer2.0's 3x3 / stride-2 convolution + its 1x1 / stride-2
downsample (PR/models/resnet.py:177-199: 64 -> 128 channels, 64 x 64 -> 32 x 32) with ALL 128 output channels in one workgroup and
the weights stationary in registers: the host-packed weight fragments (wave q = channels 32 q .. 32 q + 31, twenty K-steps: nine
taps x two 32-channel halves + the downsample's two), the LDS-DMA fill of the four parity planes of a 4 x 16-pixel output tile
(pitch 16 / 24 so that the swizzle key is the column's low bits), fragment addresses = `per-lane base [dx][kh] + immediate`, the
MFMA operand layout and the lane -> channel map of the epilogue.  Mirrors the kernel statement by statement; no timing."""
import numpy as np

from conv_fb_model import mfma_16x16x32

PIXB = 128
# planes in LDS order: (py, px), rows, pitch (pixels)
PLANES = [((1, 1), 4, 16), ((1, 0), 4, 24), ((0, 1), 5, 16), ((0, 0), 5, 24)]
PBASE = [0, 64, 64 + 96, 64 + 96 + 80]                     # first pixel of each plane
TILE_PIX = 360                                             # 45 LDS-DMA pieces
PLANE_OF = {(1, 1): 0, (1, 0): 1, (0, 1): 2, (0, 0): 3}


def pack_weight_fragments(w, w2):
    """w [128][9*64] (K = tap-major, channel-minor), w2 [128][64] -> [q 4][s 20][i 2][lane 64][8]"""
    out = np.zeros((4, 20, 2, 64, 8), dtype=w.dtype)
    for q in range(4):
        for s in range(20):
            for i in range(2):
                for lane in range(64):
                    rho, fq = lane & 15, lane >> 4
                    n = 32 * q + 8 * (rho >> 2) + 4 * i + (rho & 3)
                    if s < 18:
                        k0 = (s >> 1) * 64 + (s & 1) * 32 + 8 * fq
                        out[q, s, i, lane] = w[n, k0:k0 + 8]
                    else:
                        k0 = (s & 1) * 32 + 8 * fq
                        out[q, s, i, lane] = w2[n, k0:k0 + 8]
    return out


def run_tile(x_buf, wf, b1, b2, tile):
    """x_buf [B][66][66][64] (zero border), wf = pack_weight_fragments(w, w2); tile over (image, 8 x 2 tiles of 4 rows x 16 columns).
    Returns (img, y0, x0, main [4][16][128] = relu(conv3x3s2 + b1), ds [4][16][128] = conv1x1s2 + b2)."""
    img, r = tile // 16, tile % 16
    y0, x0 = (r // 2) * 4, (r % 2) * 16
    lds = np.zeros(TILE_PIX * PIXB // 2)
    lanes = np.arange(64)
    frow, fq = lanes & 15, lanes >> 4
    # ---- LDS-DMA: piece j = pixels 8 j .. 8 j + 7 of the tile image; lane l -> pixel p = 8 j + l / 8 of plane P, (row, col) by the
    #      plane's pitch; source = buffer pixel (2 (y0 + row) + py, 2 (x0 + col) + px), column clamped (pad columns of the pitch)
    for wave in range(4):
        for i in range(12):
            j = wave + 4 * i
            if j >= 45:
                continue
            for l in range(64):
                p = 8 * j + (l >> 3)
                pl = 0 if p < PBASE[1] else (1 if p < PBASE[2] else (2 if p < PBASE[3] else 3))
                (py, px), rows, pitch = PLANES[pl]
                q = p - PBASE[pl]
                row, col = q // pitch, q % pitch
                brow = 2 * (y0 + row) + py
                bcol = min(2 * (x0 + col) + px, 65)
                assert brow <= 65
                c8 = ((l & 7) ^ (p & 7)) * 8
                dst = (j * 1024 + l * 16) // 2
                lds[dst:dst + 8] = x_buf[img, brow, bcol, c8:c8 + 8]

    def rd128(byte_addr):
        return np.stack([lds[a // 2:a // 2 + 8] for a in byte_addr])

    A = {dx: (frow + dx) * PIXB + ((fq ^ ((frow + dx) & 7)) * 16) for dx in (0, 1)}
    main = np.zeros((4, 16, 128)); ds = np.zeros((4, 16, 128))
    for q in range(4):
        for phase, (s_lo, s_hi, bias, dst) in enumerate(((0, 18, b1, main), (18, 20, b2, ds))):
            for j in range(4):
                acc = [np.array([bias[32 * q + 8 * f + 4 * i:32 * q + 8 * f + 4 * i + 4] for f in fq]) for i in range(2)]
                for s in range(s_lo, s_hi):
                    kh = s & 1
                    ky, kx = ((s >> 1) // 3, (s >> 1) % 3) if s < 18 else (1, 1)
                    pl = PLANE_OF[(ky & 1, kx & 1)]
                    _, rows, pitch = PLANES[pl]
                    row, dx = j + (ky >> 1), kx >> 1
                    assert row < rows
                    imm = (PBASE[pl] + row * pitch) * PIXB
                    xf = rd128((A[dx] ^ (kh * 64)) + imm)
                    for i in range(2):
                        acc[i] = mfma_16x16x32(wf[q, s, i], xf, acc[i])
                v = np.concatenate([acc[0], acc[1]], axis=1)
                if phase == 0:
                    v = np.maximum(v, 0.0)
                for l in range(64):
                    dst[j, frow[l], 32 * q + 8 * fq[l]:32 * q + 8 * fq[l] + 8] = v[l]
    return img, y0, x0, main, ds
