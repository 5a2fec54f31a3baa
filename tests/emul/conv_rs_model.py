"""Index-math model of csrc/conv_rs.hip (test infrastructure) -- layer2's 3x3 / stride-1 convolutions (128 -> 128 channels, 32 x 32
maps) with a 64-channel block's weights stationary in registers: host-packed fragments (wave (ch, ph): channels 64 nb + 32 ch .. + 31;
K-step s = 4 tap + 2 chunk + kh), the LDS-DMA fill of the 10 x 18 input patch of an 8 x 16 output tile as two 64-channel sub-images
(23 pieces each, chunk swizzle), fragment addresses `per-lane base [delta][kh] + immediate` with delta = (2 (row + ky) + kx) & 7, the
MFMA operand layout and the epilogue's lane -> channel map.  Mirrors the kernel statement by statement; no timing."""
import numpy as np

from conv_fb_model import mfma_16x16x32

PIXB = 128
PITCH = 18
CHUNK_B = 23 * 1024            # one 64-channel sub-image: 10 x 18 = 180 pixels + 4 pad = 23 LDS-DMA pieces
TILE_B = 2 * CHUNK_B


def pack_weight_fragments(w):
    """w [128][9*128] (K = tap-major, channel-minor) -> [nb 2][ch 2][s 36][i 2][lane 64][8]"""
    out = np.zeros((2, 2, 36, 2, 64, 8), dtype=w.dtype)
    for nb in range(2):
        for ch in range(2):
            for s in range(36):
                tap, c, kh = s >> 2, (s >> 1) & 1, s & 1
                for i in range(2):
                    for lane in range(64):
                        rho, fq = lane & 15, lane >> 4
                        n = 64 * nb + 32 * ch + 8 * (rho >> 2) + 4 * i + (rho & 3)
                        k0 = tap * 128 + c * 64 + kh * 32 + 8 * fq
                        out[nb, ch, s, i, lane] = w[n, k0:k0 + 8]
    return out


def run_tile(x_buf, wf, bias, tile, nb):
    """x_buf [B][34][34][128] (zero border); tile over (image, 4 x 2 tiles of 8 rows x 16 columns); returns the conv + bias of the
    tile for channels 64 nb .. 64 nb + 63: (img, y0, x0, out [8][16][64])."""
    img, r = tile // 8, tile % 8
    y0, x0 = (r // 2) * 8, (r % 2) * 16
    lds = np.zeros(TILE_B // 2)
    lanes = np.arange(64)
    frow, fq = lanes & 15, lanes >> 4
    for wave in range(4):
        for i in range(12):
            j = wave + 4 * i
            if j >= 46:
                continue
            c, jj = j // 23, j % 23
            for l in range(64):
                p = min(8 * jj + (l >> 3), 179)
                py, px = p // 18, p % 18
                pp = 8 * jj + (l >> 3)                    # the LDS pixel slot (pad slots 180..183 hold a copy of pixel 179: never read)
                c8 = ((l & 7) ^ (pp & 7)) * 8
                dst = (c * CHUNK_B + jj * 1024 + l * 16) // 2
                lds[dst:dst + 8] = x_buf[img, y0 + py, x0 + px, 64 * c + c8:64 * c + c8 + 8]

    def rd128(byte_addr):
        return np.stack([lds[a // 2:a // 2 + 8] for a in byte_addr])

    A = {d: (frow + d) * PIXB + ((fq ^ ((frow + d) & 7)) * 16) for d in range(8)}
    out = np.zeros((8, 16, 64))
    for wave in range(4):
        ch, ph = wave & 1, wave >> 1
        for j in range(4):
            oy = 4 * ph + j
            acc = [np.array([bias[64 * nb + 32 * ch + 8 * f + 4 * i:64 * nb + 32 * ch + 8 * f + 4 * i + 4] for f in fq]) for i in range(2)]
            for s in range(36):
                tap, c, kh = s >> 2, (s >> 1) & 1, s & 1
                ky, kx = tap // 3, tap % 3
                d = (2 * (j + ky) + kx) & 7
                imm = c * CHUNK_B + ((j + ky) * PITCH + kx - d) * PIXB
                assert 0 <= imm < 65536
                xf = rd128(ph * 4 * PITCH * PIXB + (A[d] ^ (kh * 64)) + imm)
                for i in range(2):
                    acc[i] = mfma_16x16x32(wf[nb, ch, s, i], xf, acc[i])
            v = np.concatenate([acc[0], acc[1]], axis=1)
            for l in range(64):
                out[oy, frow[l], 32 * ch + 8 * fq[l]:32 * ch + 8 * fq[l] + 8] = v[l]
    return img, y0, x0, out
