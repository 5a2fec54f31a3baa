"""Index-math model of csrc/conv_fb.hip (test infrastructure) -- ONE launch per ResNet BasicBlock of layer1
(PR/models/resnet.py:61-77: relu(bn2(conv2(relu(bn1(conv1(x))))) + x), 64 -> 64 channels, 64 x 64 maps).

Mirrors the kernel statement by statement at the level where its bugs would live: the register-resident weight FRAGMENTS in the
order the host packs them, the LDS-DMA fill of the 20 x 20 input patch (per-lane sources, clamped at the buffer's edges, chunk
swizzle), the fragment addresses as `per-lane base + immediate offset` with the six swizzle variants, the v_mfma_f32_16x16x32
operand layout, conv1 on the 23 (+1 idle) linearly indexed pixel tiles of the pitch-20 intermediate grid with the out-of-image
positions zeroed, conv2 on the sixteen output rows, the residual out of the patch's centre, and the lane -> channel map of the
epilogue.  No timing, no barriers."""
import numpy as np

PITCH = 20                       # pixels per row of the input patch AND of the intermediate grid in LDS
PIXB = 128                       # bytes per pixel (64 channels, 2-byte elements)
PATCH_B = PITCH * PITCH * PIXB   # 51 200
INTER_OFF = 2 * PATCH_B          # two patch buffers, then the intermediate
INTER_PIX = 368                  # 23 tiles of 16
LDS_BYTES = INTER_OFF + INTER_PIX * PIXB
DELTAS = [0, 1, 2, 4, 5, 6]      # (4 ky + kx) & 7 over the nine taps


def pack_weight_fragments(w):
    """w [64][9*64] (K = tap-major, channel-minor: (ky*3+kx)*64 + c) -> [ch 2][s 18][i 2][lane 64][8]: the A operand of k-step
    s = 2 tap + kh for the wave that owns output channels 32 ch .. 32 ch + 31; MFMA row rho of tile i is channel
    32 ch + 8 (rho >> 2) + 4 i + (rho & 3), so that after the MFMAs lane (frow, fq) holds 8 CONSECUTIVE channels of its pixel."""
    out = np.zeros((2, 18, 2, 64, 8), dtype=w.dtype)
    for ch in range(2):
        for s in range(18):
            tap, kh = s >> 1, s & 1
            for i in range(2):
                for lane in range(64):
                    rho, fq = lane & 15, lane >> 4
                    n = 32 * ch + 8 * (rho >> 2) + 4 * i + (rho & 3)
                    k0 = tap * 64 + kh * 32 + 8 * fq
                    out[ch, s, i, lane] = w[n, k0:k0 + 8]
    return out


def mfma_16x16x32(a, b, c):
    """a, b [64 lanes][8], c [64 lanes][4] -> c + A.B with A[row = l & 15][k = 8 (l >> 4) + j], B[k][col = l & 15],
    D[row = 4 (l >> 4) + e][col = l & 15]."""
    A = np.zeros((16, 32)); Bm = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a[l]
        Bm[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b[l]
    D = A @ Bm
    out = c.copy()
    for l in range(64):
        for e in range(4):
            out[l, e] += D[4 * (l >> 4) + e, l & 15]
    return out


def run_tile(x_buf, wf1, b1, wf2, b2, tile):
    """x_buf [B][66][66][64] (zero border baked in), wf1 / wf2 = pack_weight_fragments(...), b1 / b2 [64]; tile = index over
    (image, 4 x 4 tiles of 16 x 16).  Returns out [16][16][64] = relu(conv2(relu(conv1(x) + b1)) + b2 + x) of that tile."""
    img, r = tile // 16, tile % 16
    y0, x0 = (r // 4) * 16, (r % 4) * 16
    lds = np.zeros(LDS_BYTES // 2)                       # one float per 2-byte element
    lanes = np.arange(64)
    frow, fq = lanes & 15, lanes >> 4
    cur = 0
    patch_base = cur * PATCH_B

    # ---- LDS-DMA of the patch: piece j (1 KiB) = patch pixels 8 j .. 8 j + 7; lane l -> pixel p = 8 j + l / 8, physical chunk l % 8
    #      holds logical chunk (l % 8) ^ (p & 7); source row / column clamped to the buffer (those pixels only feed zeroed outputs)
    for wave in range(4):
        for i in range(13):
            j = wave + 4 * i
            if j >= 50:
                continue
            for l in range(64):
                p = 8 * j + (l >> 3)
                py, px = (p * 205) >> 12, p - 20 * ((p * 205) >> 12)
                assert (py, px) == divmod(p, 20)
                row = min(max(y0 - 1 + py, 0), 65)
                col = min(max(x0 - 1 + px, 0), 65)
                c8 = ((l & 7) ^ (p & 7)) * 8
                dst = (patch_base + j * 1024 + l * 16) // 2
                lds[dst:dst + 8] = x_buf[img, row, col, c8:c8 + 8]

    def rd128(byte_addr):                                # ds_read_b128, per-lane byte address
        return np.stack([lds[a // 2:a // 2 + 8] for a in byte_addr])

    # per-lane swizzled bases, one per delta: pixel (frow + d), logical chunk fq (kh = 0; kh = 1: ^ 64)
    A = {d: (frow + d) * PIXB + ((fq ^ ((frow + d) & 7)) * 16) for d in DELTAS}

    # ---- conv1 on the linear pitch-20 grid: wave (ch, ph) takes tiles T = 12 ph + T', T' = 0..11 (T = 23: idle, nothing stored)
    for wave in range(4):
        ch, ph = wave & 1, wave >> 1
        for Tp in range(12):
            T = 12 * ph + Tp
            acc = [np.tile(np.array([b1[32 * ch + 8 * q + 4 * i:32 * ch + 8 * q + 4 * i + 4] for q in fq]), 1) for i in range(2)]
            if T < 23:
                for s in range(18):
                    tap, kh = s >> 1, s & 1
                    ky, kx = tap // 3, tap % 3
                    d = (4 * ky + kx) & 7
                    imm = (16 * Tp + 20 * ky + kx - d) * PIXB
                    assert 0 <= imm < 65536
                    xf = rd128(patch_base + ph * 12 * 2048 + (A[d] ^ (kh * 64)) + imm)
                    for i in range(2):
                        acc[i] = mfma_16x16x32(wf1[ch, s, i], xf, acc[i])
                # epilogue: ReLU, zero outside the image, 16 bytes = channels 32 ch + 8 fq .. + 7 of pixel m = 16 T + frow
                m = 16 * T + frow
                rr, cc = m // 20, m % 20
                inside = ((y0 - 1 + rr >= 0) & (y0 - 1 + rr < 64) & (x0 - 1 + cc >= 0) & (x0 - 1 + cc < 64))
                v = np.maximum(np.concatenate([acc[0], acc[1]], axis=1), 0.0) * inside[:, None]
                E0 = frow * PIXB + (((ch * 4 + fq) ^ (frow & 7)) * 16)
                for l in range(64):
                    a = (INTER_OFF + E0[l] + T * 2048) // 2
                    lds[a:a + 8] = v[l]

    # ---- conv2: wave (ch, ph) takes output rows oy = 8 ph + oy', 16 pixels each
    out = np.zeros((16, 16, 64))
    for wave in range(4):
        ch, ph = wave & 1, wave >> 1
        for oyp in range(8):
            oy = 8 * ph + oyp
            acc = [np.array([b2[32 * ch + 8 * q + 4 * i:32 * ch + 8 * q + 4 * i + 4] for q in fq]) for i in range(2)]
            for s in range(18):
                tap, kh = s >> 1, s & 1
                ky, kx = tap // 3, tap % 3
                d = (4 * (oyp + ky) + kx) & 7
                imm = (20 * (oyp + ky) + kx - d) * PIXB
                assert 0 <= imm < 65536
                xf = rd128(INTER_OFF + ph * 8 * 20 * PIXB + (A[d] ^ (kh * 64)) + imm)
                for i in range(2):
                    acc[i] = mfma_16x16x32(wf2[ch, s, i], xf, acc[i])
            # residual: pixel (oy + 2, frow + 2) of the patch, logical chunk 4 ch + fq
            par = oyp & 1
            key = (frow + 2 + 4 * par) & 7
            Rr = (frow + 42) * PIXB + (((ch * 4 + fq) ^ key) * 16)
            res = rd128(patch_base + ph * 8 * 20 * PIXB + Rr + 20 * oyp * PIXB)
            v = np.maximum(np.concatenate([acc[0], acc[1]], axis=1) + res, 0.0)
            for l in range(64):
                out[oy, frow[l], 32 * ch + 8 * fq[l]:32 * ch + 8 * fq[l] + 8] = v[l]
    return img, y0, x0, out
