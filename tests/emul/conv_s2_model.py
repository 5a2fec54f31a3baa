"""Index-math model of csrc/conv_s2.hip (test infrastructure): fills a byte-addressed LDS image exactly as the kernel's
LDS-DMA pieces do (plane layout, per-lane source offsets, chunk swizzle), reads the MFMA fragments from the addresses the
kernel computes, accumulates with the v_mfma_f32_16x16x32 operand layout, and returns the tile's outputs -- so that the
CPU suite checks the parity-plane decomposition, the K order, the swizzle and the tile / image mapping against a plain
stride-2 convolution before a GPU ever runs it.  Mirrors the kernel statement by statement; no timing, no barriers."""
import numpy as np

K_SLAB, K_STEPS = 64 * 128, 10
TAPS = [4, -1, 3, 5, 1, 7, 0, 6, 2, 8]
# step -> (combo, dy); combo -> (plane, dx)
STEP_X = [(0, 0), (0, 0), (1, 0), (2, 0), (3, 0), (3, 1), (4, 0), (4, 1), (5, 0), (5, 1)]
CPLANE, CDX = [0, 1, 1, 2, 3, 3], [0, 0, 1, 0, 0, 1]


def acc_channel(i, r):
    return 32 * (i >> 1) + 8 * (r >> 2) + 4 * (i & 1) + (r & 3)


class Geo:
    def __init__(self, w8):
        self.w8 = w8
        self.py = [1, 1, 0, 0]
        self.px = [1, 0, 1, 0]
        self.rows = [8 + (1 if self.py[p] == 0 else 0) for p in range(4)]
        self.cw = [(2 * (8 + (1 if self.px[p] == 0 else 0))) if w8 else 16 + (1 if self.px[p] == 0 else 0) for p in range(4)]
        self.npix = [self.rows[p] * self.cw[p] for p in range(4)]
        self.npiece = [(n + 7) // 8 for n in self.npix]
        self.dpw = [(n + 3) // 4 for n in self.npiece]
        self.base_piece = [sum(self.npiece[:p]) for p in range(4)]
        self.patch = sum(self.npiece) * 1024
        self.dxs = 2 if w8 else 1


def run_tile(x_pad, w, w2, m_tile, n_blk, So, w8):
    """x_pad [B, Si+2, Si+2, Cin] (zero border), w [N, 9*Cin] (tap-major, chunk... = (ky*3+kx)*Cin + c), w2 [N, Cin].
    Returns {(img, oy, ox): (main[64], ds[64])} for the tile's 128 pixels, channels n_blk .. n_blk + 63 (no bias)."""
    g = Geo(w8)
    B, Hp, Wp, Cin = x_pad.shape
    in_pix, in_row, in_img = Cin, Wp * Cin, Hp * Wp * Cin
    xin = x_pad.reshape(-1)
    nchunks = Cin // 64
    Ktot, Ktot2 = 9 * Cin, Cin
    wflat, w2flat = w.reshape(-1), w2.reshape(-1)
    if w8:
        img0, y0, x0 = 2 * m_tile, 0, 0
    else:
        tpr = So // 16
        tpi = tpr * (So // 8)
        img0 = m_tile // tpi
        t = m_tile - img0 * tpi
        y0, x0 = (t // tpr) * 8, (t % tpr) * 16
    lds = np.zeros(K_STEPS * K_SLAB + g.patch, dtype=np.float64).reshape(-1)          # one float per 2-byte element
    LE = lambda byte: byte // 2
    acc = np.zeros((4, 2, 4, 4, 64, 4))       # [wave][set][i][j][lane][e]
    lanes = np.arange(64)
    drow, dchunk = lanes >> 3, lanes & 7
    frow, fq = lanes & 15, lanes >> 4

    def dma(src_elem, lds_byte):              # 64 lanes x 16 bytes, lane-linear destination
        for l in range(64):
            lds[LE(lds_byte) + 8 * l: LE(lds_byte) + 8 * l + 8] = xin_or_w[src_elem[l]: src_elem[l] + 8]

    for ch in range(nchunks):
        # ---- fill: all four groups of this chunk (the kernel's refill order does not change the bytes)
        for wave_id in range(4):
            for p in range(4):
                for i in range(g.dpw[p]):
                    piece = min(wave_id + 4 * i, g.npiece[p] - 1)
                    q = np.minimum(piece * 8 + drow, g.npix[p] - 1)
                    r, c = q // g.cw[p], q % g.cw[p]
                    if w8:
                        off = (img0 + (c & 1)) * in_img + (2 * r + g.py[p]) * in_row + (2 * (c >> 1) + g.px[p]) * in_pix
                    else:
                        off = img0 * in_img + (2 * (y0 + r) + g.py[p]) * in_row + (2 * (x0 + c) + g.px[p]) * in_pix
                    src = off + ((dchunk ^ (q & 7)) * 8) + ch * 64
                    xin_or_w = xin
                    dma(src, K_STEPS * K_SLAB + (g.base_piece[p] + piece) * 1024)
            for s in range(K_STEPS):
                for i in range(2):
                    row = (wave_id * 2 + i) * 8 + drow
                    chn = n_blk + np.array([acc_channel(rr >> 4, rr & 15) for rr in row])
                    if TAPS[s] < 0:
                        src = chn * Ktot2 + ((dchunk ^ (row & 7)) * 8) + ch * 64
                        xin_or_w = w2flat
                    else:
                        src = chn * Ktot + ((dchunk ^ (row & 7)) * 8) + (TAPS[s] * nchunks + ch) * 64
                        xin_or_w = wflat
                    dma(src, s * K_SLAB + (wave_id * 2 + i) * 1024)
        # ---- compute: every wave, every step
        for wave_id in range(4):
            pg, kw = wave_id & 1, wave_id >> 1
            w_lane_k = frow * 128 + (((kw * 4 + fq) ^ (frow & 7)) * 16)
            for s in range(K_STEPS):
                cb, dy = STEP_X[s]
                p = CPLANE[cb]
                A = np.zeros((4, 64, 8))
                Bf = np.zeros((4, 64, 8))
                for i in range(4):
                    ad = w_lane_k + s * K_SLAB + i * 2048
                    for l in range(64):
                        A[i, l] = lds[LE(ad[l]): LE(ad[l]) + 8]
                for j in range(4):
                    row = min(4 * pg + j + dy, g.rows[p] - 1)
                    q = row * g.cw[p] + frow + CDX[cb] * g.dxs
                    ad = K_STEPS * K_SLAB + g.base_piece[p] * 1024 + q * 128 + (((kw * 4 + fq) ^ (q & 7)) * 16)
                    for l in range(64):
                        Bf[j, l] = lds[LE(ad[l]): LE(ad[l]) + 8]
                st = 1 if s == 1 else 0
                for i in range(4):
                    # A operand: lane l = row l % 16, k = 8 (l / 16) .. + 7 ;  B operand: lane l = column l % 16, same k
                    Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
                    for l in range(64):
                        Am[l & 15, 8 * (l >> 4): 8 * (l >> 4) + 8] = A[i, l]
                    for j in range(4):
                        for l in range(64):
                            Bm[8 * (l >> 4): 8 * (l >> 4) + 8, l & 15] = Bf[j, l]
                        D = Am @ Bm                                    # D[row][col]: lane l holds rows 4 (l / 16) + e, column l % 16
                        for l in range(64):
                            acc[wave_id, st, i, j, l] += D[4 * (l >> 4): 4 * (l >> 4) + 4, l & 15]
    # ---- K-half combine + output mapping (wave (kw, pg) keeps tiles 2 kw, 2 kw + 1)
    out = {}
    for pg in range(2):
        tot = acc[pg] + acc[pg + 2]
        for j in range(4):
            for l in range(64):
                fr, q4 = l & 15, l >> 4
                if w8:
                    key = (img0 + (fr & 1), 4 * pg + j, fr >> 1)
                else:
                    key = (img0, y0 + 4 * pg + j, x0 + fr)
                main, ds = out.setdefault(key, (np.zeros(64), np.zeros(64)))
                for i in range(4):
                    for e in range(4):
                        chn = acc_channel(i, 4 * q4 + e)
                        main[chn] = tot[0, i, j, l, e]
                        ds[chn] = tot[1, i, j, l, e]
    return out
