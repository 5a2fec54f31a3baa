// 3x3 / stride-1 convolution with an LDS-RESIDENT INPUT PATCH (third generation).
//
// Why: the implicit-GEMM kernels (conv.hip, conv_ring.hip) re-stage every input pixel once per filter tap, so a
// 3x3 conv pulls 9x its input through the L2 -> LDS path, and that path (not HBM: the L2 hit rate is ~85 %, not
// MFMA) is what they saturate at 32 crops.  Here a workgroup loads the (tile + halo) pixels of one 64-channel
// chunk ONCE, keeps them in LDS, and feeds all 9 taps from there; only the [64 x 64] weight slab of each tap is
// streamed (LDS-DMA ring).  Per workgroup: 256 output pixels x 64 output channels, wave tile 64 x 64,
// v_mfma_f32_16x16x32_bf16, weights = A operand, pixels = B operand (lane = pixel -> 8-byte NHWC stores).
//
// L2 -> LDS bytes per MFMA cycle drop from 64-128 (implicit GEMM, 128x128..64x64 tiles) to ~16 (weights) + ~10
// (patch), inside what one CU can pull from L2.
//
// Patch image: pixel-major [pixels][64 ch] bf16, 128 B per pixel, 16-byte chunk index XOR-swizzled by f(q) of the patch
// pixel index q: f = q & 7 for the 16-pixel-row tiles -- conflict-free for ds_read_b128, whose four lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... each touch 16 DISTINCT 16-byte bank slots ((a / 16) mod 16) for any
// 16 consecutive pixels (the first version used (q >> 1) & 7: 2-way conflicts at 3 of 4 tap positions, 8 LDS cycles
// per fragment read instead of 4; model: tools/lds_bank_check.py).  The 8 x 8 maps get the same reads by INTERLEAVING two
// images along the columns of the patch (patch column c' = 2 col + image): an MFMA tile is then 16 consecutive patch pixels
// = one row of both images, a one-column tap shift is a shift by 2.  (Rounds 1-2 read two 8-pixel rows of ONE image per
// fragment: no linear swizzle is conflict-free for that shape -- 32 % of the kernel's LDS cycles were bank conflicts.)
// The image is written by LDS-DMA (lane-linear: 8 pixels x 8 chunks per wave-instruction, swizzle applied on the
// SOURCE chunk) and read with ds_read_b128; the 36 per-lane fragment offsets (9 taps x 4 tiles) live in registers.
//
// Tile geometries (W = feature-map width = height):
//   W >= 16 : tile = 16 x 16 pixels of one image, patch 18 x 18; wave w owns rows 4w..4w+3, MFMA tile j = one row.
//   W == 8  : tile = NPG whole 8 x 8 images = NPG / 2 interleaved pairs, patch (NPG / 2) x (10 x 20); pixel group w owns
//             rows 4 (w & 1) .. + 3 of BOTH images of pair w >> 1, MFMA tile j = row 4 (w & 1) + j, lane = 2 col + image.
//
// One structure (the per-tap-ring forms of the first round are gone): 2 NPG waves = NPG 64-pixel groups x 2 K-halves; wave
// pairs (w, w + NPG) split every 64-deep K-step in halves and exchange accumulators through LDS at the end; GROUPED weight
// ring -- the three slabs of a kernel row land together in one third of a 3 x 3-slab ring, one counted vmcnt + barrier
// per three taps, and the slabs of group g+2 are requested at the top of group g: a group computes for 768 (128-pixel
// tile) or 1536 cycles, an L2 -> LDS fetch takes ~2k under load, and with the two-group ring of the first version every
// group waited for its own weights.  With DB (template flag, the default for layer3 / layer4) TWO patch buffers, so the
// next 64-channel chunk's patch lands under the nine taps of the current one.
// Inside a tap the 8 fragment reads of the NEXT tap are issued between the first MFMA pairs of the current one (two
// register sets, every position pinned by sched_barrier): hipcc otherwise gathers the reads in front and sinks the
// register-only MFMAs below the lgkmcnt wait, so that a wave waits for reads it has just issued (round 1 ran that way:
// ~65 % matrix-pipe issue inside the loop).
// Tile order: the 8 XCDs form an XM x XN grid over (pixel tiles) x (64-channel blocks), every XCD walks its rectangle
// channel-block-fastest -- the channel blocks that share an input patch run on ONE XCD at the same time (one L2 fill
// instead of N/64), and each XCD's L2 holds 1/XN of the weights.  The host picks XN per layer (fewest fabric bytes).
#include <cstdlib>
#include <vector>
#include <type_traits>

#include "conv.hpp"
#include "conv_elem.hpp"

// (the step macros index fragment arrays with literal constants inside `if (<literal false>)` branches)
#pragma clang diagnostic ignored "-Warray-bounds"

namespace {
using namespace spsp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_b128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void dma16(const uint16_t* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


#ifdef SPSP_CONV_STAMPS
#define CSTAMP(i) if (a.dbg && blockIdx.x == 8 && tid == 0) a.dbg[i] = __builtin_amdgcn_s_memtime()
// cycles this wave spends at the top of the weight groups: in the counted vmcnt wait, and in the barrier behind it
#define CS_DECL unsigned long long cs_t0_ = 0, cs_t1_ = 0, cs_wait_ = 0, cs_bar_ = 0;
#define CS_BEGIN cs_t0_ = __builtin_amdgcn_s_memtime();
#define CS_MID cs_t1_ = __builtin_amdgcn_s_memtime(); cs_wait_ += cs_t1_ - cs_t0_;
#define CS_END cs_bar_ += __builtin_amdgcn_s_memtime() - cs_t1_;
#define CS_STORE if (a.dbg && blockIdx.x == 8 && tid == 0) { a.dbg[8] = cs_wait_; a.dbg[9] = cs_bar_; }
// diagnosis: splitk == -1 in the stamped build = no weight / patch DMA inside the loop (results are garbage; read the K-loop cycles)
#define CS_NODMA (a.splitk == -1)
// every workgroup: start / end on the 100 MHz constant clock (s_memtime is per XCD)
#define CS_WG(i) if (a.dbg && tid == 0) a.dbg[16 + 2 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define CSTAMP(i)
#define CS_DECL
#define CS_BEGIN
#define CS_MID
#define CS_END
#define CS_STORE
#define CS_NODMA false
#define CS_WG(i)
#endif
constexpr int kPixB = 128;                   // bytes per patch pixel (64 ch)

// Output channel (within the 64-channel block) of accumulator tile i, MFMA row r = 4 fq + e: the weight rows are loaded in
// this order so that a lane ends up with 8 CONSECUTIVE channels of each of its pixels in tiles (2h, 2h+1) -- the epilogue
// is then plain 16-byte loads and stores, four lanes covering an aligned 64-byte half line (no LDS staging).
__host__ __device__ constexpr int acc_channel(int i, int r) { return 32 * (i >> 1) + 8 * (r >> 2) + 4 * (i & 1) + (r & 3); }

template <bool W8>
__device__ __forceinline__ int patch_swz(int q) { return q & 7; }

// NPG = 64-pixel groups per tile: 4 (256-pixel tile, 8 waves) or 2 (128-pixel tile, 4 waves -- twice the workgroups, for
// the layers whose 256-pixel tiles leave most of the chip idle: layer3 has 128 of them, layer4 64)
template <bool W8, int NPG>
struct Geo {
    static constexpr int NW = 2 * NPG;                            // waves per workgroup (pixel groups x 2 K-halves)
    static constexpr int TPIX = 64 * NPG;                         // output pixels per tile
    static constexpr int PW = W8 ? 20 : 18;                       // patch row pitch in pixels (8 x 8 maps: 10 columns x 2 interleaved images)
    static constexpr int DXS = W8 ? 2 : 1;                        // patch-pixel shift of a one-column tap shift
    static constexpr int NPIX = W8 ? 100 * NPG : (4 * NPG + 2) * 18;   // patch pixels
    static constexpr int NDMA = (NPIX + 7) / 8;                   // 8 pixels per wave-instruction
    static constexpr int PATCH = NDMA * 8 * kPixB;                // bytes (padded to whole DMA pieces)
    static constexpr int DPW = (NDMA + NW - 1) / NW;              // patch DMA instructions per wave
};

// One workgroup: TPIX pixels x (16 NCT) channels.  grid = (M/TPIX) * (N / (16 NCT)).
// NCT = channel tiles (of 16) per wave: 4 (64-channel blocks) or 2 (32-channel blocks -- twice the workgroups, for layer4 whose
// 128 blocks of 128 pixels x 64 channels fill half the chip and whose 776 KB of staging per workgroup, not its MFMAs, set its time).
//
// PROD (round 3): FOUR MORE wavefronts, one per SIMD, that do nothing but issue the LDS-DMA pieces (patches and weight slabs).  A
// global_load_lds occupies its wave's instruction stream for ~60-180 cycles (the address path takes the 64 lanes in quarters),
// and with ONE wave per SIMD the matrix pipe drains meanwhile: ~100 pieces per wave and tile = a quarter of a workgroup's
// life.  The producers wait (counted vmcnt) for their own pieces in front of the very barriers the consumers already have, so
// the consumers' group top is the bare barrier; they end after the K loop (a barrier counts the surviving waves only).
// The accumulating waves then share their SIMD's register file with a producer: <= 256 VGPRs.
template <bool W8, bool F16, bool DB, int NPG, int NG, int NCT = 4, bool PROD = false>
__global__ __launch_bounds__(128 * NPG + (PROD ? 256 : 0)) void conv3x3_patch_kernel(const ConvArgs a) {
    using G = Geo<W8, NPG>;
    static_assert(!PROD || NPG == 2, "producer waves: four accumulating waves, the pieces are split four ways as without them");
    constexpr int PW = G::PW, kNS = 3 * NG, NW = G::NW, KS = 2;   // NG = ring depth in groups of three slabs (2 or 3)
    constexpr int NCH = 16 * NCT, kWStage = NCH * 128;            // (shadows the file-level constant)
    static_assert(NG == 2 || NG == 3, "ring depth");
    static_assert(NCT == 4 || (NCT == 2 && NW == 4), "32-channel blocks: four waves (one slab piece each)");
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const ring = smem;                              // kNS x 8 KB weight stages
    unsigned char* const patch0 = smem + kNS * kWStage;            // the patch

    const int tid = threadIdx.x, lane = tid & 63;
    CS_WG(0)
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = PROD && wave_id >= NW;                   // (wave-uniform)
    const int dma_id = PROD ? wave_id - NW : wave_id;              // which share of the DMA pieces this wave issues
    const int wave = wave_id & (NPG - 1);                          // which 64 pixels of the tile
    const int kw = (wave_id / NPG) & 1;                            // KS == 2: which 32-deep half of every K-step
    const int W = a.Wo;                                            // == Ho, == input W (stride 1)
    const int mt = a.M / G::TPIX, nb = a.N / NCH;
    int m_tile, n_blk;
    {
        // blocks b, b+8, b+16 ... share an XCD (round-robin dispatch; a speed assumption only): XCD x = b & 7 owns the
        // rectangle (x / XN, x % XN) of the XM x XN grid and walks it channel-block-fastest
        const int XN = a.xcd_n, XM = 8 / XN;
        const int bid = blockIdx.x;
        if (XN > 0 && mt % XM == 0 && nb % XN == 0) {
            const int x = bid & 7, li = bid >> 3, nbl = nb / XN;
            m_tile = (x / XN) * (mt / XM) + li / nbl;
            n_blk = ((x % XN) * nbl + li % nbl) * NCH;
        } else {
            m_tile = bid / nb;
            n_blk = (bid % nb) * NCH;
        }
    }

    // tile origin in the padded input (element offset of patch pixel (0,0), channel 0)
    int img0, y0 = 0, x0 = 0;
    if constexpr (W8) { img0 = m_tile * NPG; }
    else {
        constexpr int TH = 4 * NPG;                                // tile height (its width is 16)
        const int tpr = W / 16, tpi = tpr * (W / TH);              // tiles per row / per image
        img0 = m_tile / tpi;
        const int t = m_tile - img0 * tpi;
        y0 = (t / tpr) * TH; x0 = (t % tpr) * 16;
    }
    const int nch_tot = a.chunks, nchunks = a.chunks, ch0 = 0;     // Cin / 64
    const uint16_t* const in_ptr = a.in;                           // scalar copies (keep the kernarg struct out of lambdas)
    const uint16_t* const w_ptr = a.w;
    const int Ktot = a.nsteps * 64;                                // 9 * Cin

    // ---- patch LDS-DMA sources: lane l of piece p feeds patch pixel q = 8p + l/8, physical chunk l%8 ----
    const int drow = lane >> 3, dchunk = lane & 7;
    int p_src[G::DPW];                                             // element offsets into a.in (chunk 0)
#pragma unroll
    for (int i = 0; i < G::DPW; ++i) {
        int p = dma_id + NW * i;
        p = p < G::NDMA ? p : G::NDMA - 1;
        int q = p * 8 + drow;
        q = q < G::NPIX ? q : G::NPIX - 1;                         // tail lanes re-read the last pixel (harmless)
        int off;
        if constexpr (W8) {
            const int pair = q / 200, r = q - pair * 200, py = r / 20, cc = r - py * 20;      // cc = 2 px + image of the pair
            off = (img0 + 2 * pair + (cc & 1)) * a.in_img + py * a.in_row + (cc >> 1) * a.in_pix;
        } else {
            const int py = q / 18, px = q - py * 18;
            off = img0 * a.in_img + (y0 + py) * a.in_row + (x0 + px) * a.in_pix;
        }
        p_src[i] = off + ((dchunk ^ patch_swz<W8>(q)) * 8);
    }
    auto issue_patch = [&](int ch, int pbuf = 0) {
#pragma unroll
        for (int i = 0; i < G::DPW; ++i) {
            int p = dma_id + NW * i;
            p = p < G::NDMA ? p : G::NDMA - 1;       // every wave issues exactly DPW pieces (counted vmcnt waits): the surplus repeats the last one
            dma16(in_ptr + p_src[i] + (ch0 + ch) * 64, patch0 + pbuf * G::PATCH + p * 1024);
        }
    };
    // ---- weight ring sources: stage = [64 n][64 k] of K-step g (tap-major, chunk-minor: g = tap*nchunks + ch) ----
    constexpr int LPS = (2 * NCT) / NW;                            // weight pieces per wave per stage (2 NCT pieces of 8 rows per stage)
    const uint16_t* w_src[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int row = (dma_id * LPS + i) * 8 + drow;
        w_src[i] = w_ptr + (size_t)(n_blk + acc_channel(row >> 4, row & 15)) * Ktot + ((dchunk ^ (row & 7)) * 8);   // LDS row 'row' holds that output channel
    }
    auto issue_w = [&](int g, int buf) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) dma16(w_src[i] + (size_t)g * 64, ring + buf * kWStage + (dma_id * LPS + i) * 1024);
    };

    f32x4 acc[NCT][4];
#pragma unroll
    for (int i = 0; i < NCT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    // weight fragment reads: row t*16 + frow, chunk (ks*4 + fq) ^ (frow & 7)
    const uint32_t w_lane = frow * 128 + ((fq ^ (frow & 7)) * 16);
    // patch pixel index of this lane for MFMA tile j = qb + j * PW (tap (ky, dx) adds ky * PW + dx * DXS)
    int qb;
    if constexpr (W8) qb = (wave >> 1) * 200 + (4 * (wave & 1)) * PW + frow;
    else qb = (wave * 4) * 18 + frow;

    // LDS byte offsets of this lane's pixel fragments (chunk-independent: registers for the whole kernel).
    //   W >= 16 : MFMA tile j of tap (ky, dx) is patch row j + ky shifted by dx pixels, so the three taps of a kernel COLUMN
    //             read only 6 different fragments (rows 0..5): 18 addresses xo[dx][row], and a tap needs ONE new pixel
    //             fragment beside its four weight fragments (the first tap of a column: four).  LDS reads per workgroup drop
    //             by a quarter (24 -> 18 per three taps) -- they, not the MFMAs, set the pace of the loop: 8 waves x 8
    //             ds_read_b128 x 8 cycles = the 512 cycles 16 MFMAs per wave take, before the LDS-DMA writes.
    //   W == 8  : a fragment is two 8-pixel rows, a one-row shift is not a fragment: all 36 (tap, j) addresses.
    //   (since round 3 the 8 x 8 maps are interleaved pairs and take the same form with a column shift of 2)
    constexpr bool ROWS = true;
    uint32_t xoff[3][6];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int q = qb + r * PW + dx * G::DXS;
            xoff[dx][r] = smem_base + kNS * kWStage + q * kPixB + ((((KS == 2 ? kw * 4 : 0) + fq) ^ patch_swz<W8>(q)) * 16);
        }

    // K order inside the kernel: chunk-major, then the kernel's columns (ROWS) or rows (W8), then the other tap index; the
    // weight matrix is stored tap-major/chunk-minor like the implicit-GEMM kernels, so step (ch, tap) reads K-step
    // g = tap * nchunks + ch.
    const int nsteps = 9 * nchunks;

    // ---- software pipeline: the fragment reads of step s+1 are issued between the MFMAs of step s (two weight register
    // sets; a step's set is static because the step loop is fully unrolled and 9 is odd: set = (9 ch + s) & 1).
    // Ring = 3 groups x 3 slabs.  Group g = (chunk, kernel column or row) sits in ring third g % 3; at its top every wave
    // waits (counted) for its own DMA pieces of group g, and after the barrier the slabs of group g+2 are issued into the
    // third group g-1 was read from, then the three taps run back to back -- 48 MFMAs per wave between barriers.
    // Pixel fragments: xw[group parity][row 0..5] (ROWS) -- tap T of a group multiplies rows T..T+3 -- or xw[set][j] (W8).
    // (macros, not lambdas: the fragment arrays must be indexed by literal constants to stay in registers)
    bf16x8 wfr[2][NCT], xw[2][6];
    const uint32_t w_lane_k = smem_base + (w_lane ^ (kw * 64));
#define P_SB __builtin_amdgcn_sched_barrier(0);
#ifdef SPSP_PATCH_NOREADS      // diagnosis only: the loop without its fragment reads (garbage results)
#define P_RW(SET, I, SLOT) asm volatile("" : "=v"(wfr[SET][I]) : "v"(w_lane_k));
#define P_RX(XP, XR, A0, A1) asm volatile("" : "=v"(xw[XP][XR]) : "v"(xoff[A0][A1] + pboff_));
#else
#define P_RW(SET, I, SLOT) wfr[SET][I] = lds_read_b128<(I) * 2048>(w_lane_k + (SLOT) * kWStage);
#define P_RX(XP, XR, A0, A1) xw[XP][XR] = lds_read_b128<0>(xoff[A0][A1] + pboff_);
#endif
    // two MFMAs of the step whose weights are in set S1 and whose pixel fragments are xw[XP][XT + j]
#define P_M2(S1, XP, XT, I, J, ON)                                                                        \
        P_SB                                                                                              \
        if (ON) {                                                                                         \
            acc[I][J] = Elem<F16>::mfma(wfr[S1][I], xw[XP][(XT) + (J)], acc[I][J]);                       \
            acc[I][(J) + 1] = Elem<F16>::mfma(wfr[S1][I], xw[XP][(XT) + (J) + 1], acc[I][(J) + 1]);       \
        }                                                                                                 \
        P_SB
#define SPSP_MFMA_SET(S1, XP, XT)                                                                         \
    {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        P_M2(S1, XP, XT, 0, 0, true) P_M2(S1, XP, XT, 0, 2, true) P_M2(S1, XP, XT, 1, 0, true) P_M2(S1, XP, XT, 1, 2, true) \
        if constexpr (NCT == 4) { P_M2(S1, XP, XT, 2, 0, true) P_M2(S1, XP, XT, 2, 2, true) P_M2(S1, XP, XT, 3, 0, true) P_M2(S1, XP, XT, 3, 2, true) } \
    }
    // One step: its reads (weights -> set SET from ring slab SLOT; pixels: X4 = four fragments, else one) inside the MFMAs of
    // the previous step (weights in set SET^1, pixels xw[MP][MT + j]; none if FIRST).
    //   X4: xw[XP][XR..XR+3] <- xoff[A0][A1..A1+3];   one: xw[XP][XR] <- xoff[A0][A1]
    // A step is only 16 MFMAs (256 cycles) long and every wave issues its reads at the same moment, so a step's last
    // fragments are not back when its MFMAs are through (stamps: 31 cycles per MFMA with one wave per SIMD).  The reads go
    // out in the order the NEXT step uses them -- pixels, W0, W1, W2, W3 -- and the waits are counted (LDS returns in
    // order): a step ends when W0 has landed (ENDW = 3 reads may stay in flight) and waits for W1 / W2 / W3 right before
    // the MFMAs of channel tile 1 / 2 / 3, by which time its own reads are in flight too (the counts below).
    // The last step of a group ends with ENDW = 0: after the next barrier other waves overwrite the slab it read.
#define P_WAIT(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#define SPSP_GTAP(SET, SLOT, X4, XP, XR, A0, A1, MP, MT, FIRST, ENDW)                                     \
    if constexpr (NCT == 2) {                                                                             \
        /* 32-channel blocks: 8 MFMAs per step, reads X.., W0, W1; a step ends when W0 has landed (ENDW = 3 -> 1 here) and */ \
        /* waits for the previous W1 before channel tile 1, by which time all of its own reads are in flight            */ \
        P_RX(XP, XR, A0, A1)                                                                              \
        if (X4) { P_RX(XP, (XR) + 1, A0, (A1) + 1) } else { P_RW(SET, 0, SLOT) }                          \
        P_M2((SET) ^ 1, MP, MT, 0, 0, !(FIRST))                                                           \
        if (X4) { P_RX(XP, (XR) + 2, A0, (A1) + 2) P_RX(XP, (XR) + 3, A0, (A1) + 3) } else { P_RW(SET, 1, SLOT) } \
        P_M2((SET) ^ 1, MP, MT, 0, 2, !(FIRST))                                                           \
        if (X4) { P_RW(SET, 0, SLOT) P_RW(SET, 1, SLOT) }                                                 \
        if (X4) { P_WAIT(6) } else { P_WAIT(3) }              /* previous W1 (behind it: 6 or 3 new) */   \
        P_M2((SET) ^ 1, MP, MT, 1, 0, !(FIRST)) P_M2((SET) ^ 1, MP, MT, 1, 2, !(FIRST))                   \
        if ((ENDW) == 0) { P_WAIT(0) } else { P_WAIT(1) }                                                 \
        P_SB                                                                                              \
    } else {                                                                                              \
        P_RX(XP, XR, A0, A1)                                                                              \
        if (X4) { P_RX(XP, (XR) + 1, A0, (A1) + 1) } else { P_RW(SET, 0, SLOT) }                          \
        P_M2((SET) ^ 1, MP, MT, 0, 0, !(FIRST))                                                           \
        if (X4) { P_RX(XP, (XR) + 2, A0, (A1) + 2) P_RX(XP, (XR) + 3, A0, (A1) + 3) } else { P_RW(SET, 1, SLOT) P_RW(SET, 2, SLOT) } \
        P_M2((SET) ^ 1, MP, MT, 0, 2, !(FIRST))                                                           \
        P_WAIT(6)                                             /* previous W1 (behind it: W2, W3 + 4 new) */ \
        if (X4) { P_RW(SET, 0, SLOT) P_RW(SET, 1, SLOT) } else { P_RW(SET, 3, SLOT) }                     \
        P_M2((SET) ^ 1, MP, MT, 1, 0, !(FIRST))                                                           \
        if (X4) { P_RW(SET, 2, SLOT) P_RW(SET, 3, SLOT) }                                                 \
        P_M2((SET) ^ 1, MP, MT, 1, 2, !(FIRST))                                                           \
        if (X4) { P_WAIT(9) } else { P_WAIT(6) }              /* previous W2 (behind it: W3 + 8 or 5 new) */ \
        P_M2((SET) ^ 1, MP, MT, 2, 0, !(FIRST)) P_M2((SET) ^ 1, MP, MT, 2, 2, !(FIRST))                   \
        if (X4) { P_WAIT(8) } else { P_WAIT(5) }              /* previous W3 */                            \
        P_M2((SET) ^ 1, MP, MT, 3, 0, !(FIRST)) P_M2((SET) ^ 1, MP, MT, 3, 2, !(FIRST))                   \
        P_WAIT(ENDW)                                                                                      \
        P_SB                                                                                              \
    }
    // Step T (0..2) of group R of a chunk with parity P0; s = 3 R + T is its place in the chunk, set = (s + P0) & 1.
    //   group = kernel column R, tap (ky = T, dx = R): pixels xw[GP][T..T+3], GP = (P0 + R) & 1; the step before it
    //   used xw[GP][T-1 ..] (T > 0) or the previous group's xw[GP ^ 1][2 ..].
    // ring third of group R of a chunk with parity P0: g % NG with g = 3 ch + R
#define P_GSLOT(R, P0) ((NG == 3 ? (R) : (((P0) + (R)) & 1)) * 3)
#define SPSP_STEP(R, T, P0, FIRST)                                                                        \
    {                                                                                                     \
        if constexpr ((T) == 0) SPSP_GTAP((3 * (R) + (P0)) & 1, P_GSLOT(R, P0), true, ((P0) + (R)) & 1, 0, R, 0, (((P0) + (R)) & 1) ^ 1, 2, FIRST, 3) \
        else SPSP_GTAP((3 * (R) + (T) + (P0)) & 1, P_GSLOT(R, P0) + (T), false, ((P0) + (R)) & 1, (T) + 3, R, (T) + 3, ((P0) + (R)) & 1, (T) - 1, FIRST, (T) == 2 ? 0 : 3) \
    }

    CSTAMP(0);
    CS_DECL
    {
        const int ngroups = 3 * nchunks;
        auto issue_group = [&](int g) {
            const int gch = g / 3, gr = g - gch * 3;
#pragma unroll
            for (int t = 0; t < 3; ++t) issue_w((ROWS ? t * 3 + gr : gr * 3 + t) * nch_tot + ch0 + gch, (g % NG) * 3 + t);   // ROWS: taps (ky = t, dx = gr)
        };
        if (!PROD || producer) {
            issue_patch(0);
            issue_group(0);
            if (NG == 3 && ngroups > 1) issue_group(1);
        }
        if constexpr (PROD) {
            if (producer) {
                // the DMA side of SPSP_GROUP below, group by group: the same counted waits in front of the same barriers
                for (int g = 0; g < ngroups; ++g) {
                    const int ch = g / 3, R = g - ch * 3;
                    if (NG == 2 || g + 1 >= ngroups) wait_vmcnt<0>();
                    else if (DB && R == 1 && ch + 1 < nchunks) wait_vmcnt<3 * LPS + G::DPW>();
                    else wait_vmcnt<3 * LPS>();
                    __builtin_amdgcn_s_barrier();
                    if (!DB && R == 0 && ch > 0) {
                        issue_patch(ch);
                        if (g + NG - 1 < ngroups) { issue_group(g + NG - 1); wait_vmcnt<3 * LPS>(); } else wait_vmcnt<0>();
                        __builtin_amdgcn_s_barrier();
                    } else {
                        if (DB && R == 0 && ch + 1 < nchunks) issue_patch(ch + 1, (ch + 1) & 1);
                        if (g + NG - 1 < ngroups) issue_group(g + NG - 1);
                    }
                }
                return;
            }
        }
        // At the top of group g its slabs (requested two groups ago) must have landed; what this wave issued after them may
        // stay in flight: the slabs of group g+1 and, with two patch buffers, the next chunk's patch when that was
        // requested one group ago (at a chunk's first group).
#define SPSP_GROUP(R, P0, CH)                                                                             \
    {                                                                                                     \
        const int g_ = (CH) * 3 + (R);                                                                    \
        const uint32_t pboff_ = DB ? ((CH) & 1) * G::PATCH : 0;   /* this chunk's patch buffer */          \
        CS_BEGIN                                                                                          \
        if (PROD) {}                                              /* (the producers wait for the pieces) */ \
        else if (NG == 2 || g_ + 1 >= ngroups) wait_vmcnt<0>();                                           \
        else if (DB && (R) == 1 && (CH) + 1 < nchunks) wait_vmcnt<3 * LPS + G::DPW>();                    \
        else wait_vmcnt<3 * LPS>();                                                                       \
        CS_MID                                                                                            \
        __builtin_amdgcn_s_barrier();                                                                     \
        CS_END                                                                                            \
        if (!DB && (R) == 0 && (CH) > 0) {                                                                \
            /* one patch buffer, chunk boundary: every wave is past its last read of the old patch; refill it, run */ \
            /* the pending MFMAs of the previous tap underneath, then wait for the patch (older than the slabs) */ \
            if (!PROD) issue_patch(CH);                                                                   \
            if (!PROD && g_ + NG - 1 < ngroups && !CS_NODMA) issue_group(g_ + NG - 1);                    \
            SPSP_MFMA_SET(((P0) & 1) ^ 1, ROWS ? ((P0) & 1) ^ 1 : ((P0) & 1) ^ 1, ROWS ? 2 : 0)           \
            if (PROD) {} else if (g_ + NG - 1 < ngroups) wait_vmcnt<3 * LPS>(); else wait_vmcnt<0>();     \
            __builtin_amdgcn_s_barrier();                                                                 \
            SPSP_STEP(0, 0, P0, true)                                                                     \
        } else {                                                                                          \
            /* two patch buffers, first group of a chunk: every wave is past its reads of chunk CH-1, whose buffer */ \
            /* now takes chunk CH+1 -- it has the nine taps of this chunk to land (no wait at chunk boundaries)    */ \
            if (!PROD && DB && (R) == 0 && (CH) + 1 < nchunks && !CS_NODMA) issue_patch((CH) + 1, ((CH) + 1) & 1); \
            if (!PROD && g_ + NG - 1 < ngroups && !CS_NODMA) issue_group(g_ + NG - 1);                    \
            SPSP_STEP(R, 0, P0, (R) == 0 && (CH) == 0)                                                    \
        }                                                                                                 \
        if ((R) == 1 && (CH) == 0) { CSTAMP(10); }                                                        \
        SPSP_STEP(R, 1, P0, false)                                                                        \
        if ((R) == 1 && (CH) == 0) { CSTAMP(11); }                                                        \
        SPSP_STEP(R, 2, P0, false)                                                                        \
        if ((R) == 1 && (CH) == 0) { CSTAMP(12); }                                                        \
    }
        // ring slot of group g = 3 ch + r is r (three groups of three slabs); the register-set parity of a tap flips with ch
        for (int ch = 0; ch < nchunks; ch += 2) {
            SPSP_GROUP(0, 0, ch) SPSP_GROUP(1, 0, ch) SPSP_GROUP(2, 0, ch)
            if (ch + 1 < nchunks) { SPSP_GROUP(0, 1, ch + 1) SPSP_GROUP(1, 1, ch + 1) SPSP_GROUP(2, 1, ch + 1) }
        }
#undef SPSP_GROUP
    }
    // the last step: set = (nsteps - 1) & 1; ROWS: the last group has parity (3 nchunks - 1) & 1 and its tap 2 uses rows 2..5
    if ((nsteps - 1) & 1) { if (ROWS && !((3 * nchunks - 1) & 1)) SPSP_MFMA_SET(1, 0, 2) else SPSP_MFMA_SET(1, 1, ROWS ? 2 : 0) }
    else { if (ROWS && ((3 * nchunks - 1) & 1)) SPSP_MFMA_SET(0, 1, 2) else SPSP_MFMA_SET(0, 0, ROWS ? 2 : 0) }
#undef SPSP_STEP
#undef P_GSLOT
#undef SPSP_GTAP
#undef P_WAIT
#undef SPSP_MFMA_SET
#undef P_M2
#undef P_RX
#undef P_RW
#undef P_SB

    CSTAMP(4);
    CS_STORE
    // ---- epilogue.  After the K-half exchange wave (kw, w) owns channel tiles 2 kw, 2 kw + 1 of its 64 pixels: with the
    // permuted weight rows that is channels n_blk + 32 kw + 8 fq .. + 7 in this lane, 16 bytes per pixel.  The residual is
    // loaded straight into registers before the exchange (its latency sits under the two barriers), the result is stored
    // straight from registers: no staging tile, no barrier after the exchange.  tile pixel tp = wave*64 + j*16 + frow.
    auto pix_off = [&](int tp) -> size_t {
        // 8 x 8 maps: pixel group w = tp >> 6 holds rows 4 (w & 1) + j of the two images of pair w >> 1, lane = 2 col + image
        if constexpr (W8) return (size_t)a.out_off + (size_t)(img0 + 2 * (tp >> 7) + (tp & 1)) * a.out_img + (size_t)(4 * ((tp >> 6) & 1) + ((tp >> 4) & 3)) * a.out_row + (size_t)((tp & 15) >> 1) * a.out_pix;
        else return (size_t)a.out_off + (size_t)img0 * a.out_img + (size_t)(y0 + (tp >> 4)) * a.out_row + (size_t)(x0 + (tp & 15)) * a.out_pix;
    };
    // What a wave owns after the K-half exchange, with static register indices:
    //   NCT = 4: channel tiles I0 = 2 kw, I0 + 1 of all four pixel tiles  (give: the other two channel tiles)
    //   NCT = 2: both channel tiles of pixel tiles J0 = 2 kw, J0 + 1      (give: the other two pixel tiles)
    // either way a lane holds 8 consecutive channels, cbase .. cbase + 7, of each of its pixels.
    constexpr int NJ = NCT == 4 ? 4 : 2;                            // pixel tiles kept
    const int cbase = n_blk + (NCT == 4 ? 32 * kw : 0) + 8 * fq;
    const int jbase = NCT == 4 ? 0 : 2 * kw;
    size_t poff[NJ];
    uint4 resv[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        poff[jj] = pix_off(wave * 64 + (jbase + jj) * 16 + frow) + cbase;
        resv[jj] = make_uint4(0, 0, 0, 0);
        if (a.res) resv[jj] = *reinterpret_cast<const uint4*>(a.res + poff[jj]);
    }
    const float4 bias0 = *reinterpret_cast<const float4*>(a.bias + cbase), bias1 = *reinterpret_cast<const float4*>(a.bias + cbase + 4);
    {
        // Combine the two K-halves through LDS (ring + patch are dead by now; NW waves x 8 KB).  Static register indices only.
        // (raw barriers: __syncthreads() would also wait for the residual loads just issued)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* xch = reinterpret_cast<float*>(smem);
        auto slot = [&](int w, int n) { return xch + ((size_t)(w * 8 + n) * 64 + lane) * 4; };
        auto put = [&](auto GIVE) {
            constexpr int g = decltype(GIVE)::value;
            if constexpr (NCT == 4) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(slot(wave_id, ii * 4 + j)) = acc[g + ii][j];
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) *reinterpret_cast<f32x4*>(slot(wave_id, i * 2 + jj)) = acc[i][g + jj];
            }
        };
        if (kw == 0) put(std::integral_constant<int, 2>{}); else put(std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int partner = wave_id ^ NPG;
        auto get = [&](auto KEEP) {
            constexpr int k = decltype(KEEP)::value;
            if constexpr (NCT == 4) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[k + ii][j] += *reinterpret_cast<const f32x4*>(slot(partner, ii * 4 + j));
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) acc[i][k + jj] += *reinterpret_cast<const f32x4*>(slot(partner, i * 2 + jj));
            }
        };
        if (kw == 0) get(std::integral_constant<int, 0>{}); else get(std::integral_constant<int, 2>{});
    }
    CSTAMP(5);
    auto epilogue = [&](auto K0c) {
        constexpr int K0 = decltype(K0c)::value;          // NCT = 4: first channel tile kept; NCT = 2: first pixel tile kept
        constexpr int I0 = NCT == 4 ? K0 : 0, J0 = NCT == 4 ? 0 : K0;
        float psum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) psum[e] = 0.f;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const f32x4 lo4 = acc[I0][J0 + jj], hi4 = acc[I0 + 1][J0 + jj];
            float v[8] = {lo4[0] + bias0.x, lo4[1] + bias0.y, lo4[2] + bias0.z, lo4[3] + bias0.w,
                          hi4[0] + bias1.x, hi4[1] + bias1.y, hi4[2] + bias1.z, hi4[3] + bias1.w};
            if (a.res) {
                const uint4 r = resv[jj];
                v[0] += Elem<F16>::lo(r.x); v[1] += Elem<F16>::hi(r.x); v[2] += Elem<F16>::lo(r.y); v[3] += Elem<F16>::hi(r.y);
                v[4] += Elem<F16>::lo(r.z); v[5] += Elem<F16>::hi(r.z); v[6] += Elem<F16>::lo(r.w); v[7] += Elem<F16>::hi(r.w);
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) psum[e] += v[e];
            if (a.out)
                *reinterpret_cast<uint4*>(a.out + poff[jj]) =
                    make_uint4(Elem<F16>::pack2(v[0], v[1]), Elem<F16>::pack2(v[2], v[3]), Elem<F16>::pack2(v[4], v[5]), Elem<F16>::pack2(v[6], v[7]));
        }
        if constexpr (W8) {
            if (a.pool_out) {
                // fused global average pool (resnet.py:157).  Lane parity = image of the pair.  A wave holds, of each image, the
                // pixels of its NJ rows: sum over the 8 columns by shuffles, then the waves that share the image pair and the
                // channels hand their sums over through LDS (behind the K-half exchange area) and ONE of them adds in a fixed order:
                //   NCT = 4: waves w, w ^ 1 (rows 4 (w & 1) .. + 3 each);  NCT = 2 (NPG = 2): all four waves (2 rows each).
                float* const px_ = reinterpret_cast<float*>(smem + NW * 8192);          // [wave][image 2][fq 4][8]
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = psum[e];
                    v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                    psum[e] = v;
                    if (frow < 2) px_[((wave_id * 2 + frow) * 4 + fq) * 8 + e] = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if constexpr (NCT == 4) {
                    if ((wave & 1) == 0 && frow < 2) {
                        const int b = img0 + 2 * (wave >> 1) + frow;
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            a.pool_out[(size_t)b * a.N + cbase + e] = (psum[e] + px_[(((wave_id ^ 1) * 2 + frow) * 4 + fq) * 8 + e]) * (1.0f / 64.0f);
                    }
                } else {
                    if (wave_id == 0 && frow < 2) {
                        const int b = img0 + frow;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float s01 = px_[((0 * 2 + frow) * 4 + fq) * 8 + e] + px_[((1 * 2 + frow) * 4 + fq) * 8 + e];
                            const float s23 = px_[((2 * 2 + frow) * 4 + fq) * 8 + e] + px_[((3 * 2 + frow) * 4 + fq) * 8 + e];
                            a.pool_out[(size_t)b * a.N + cbase + e] = (s01 + s23) * (1.0f / 64.0f);
                        }
                    }
                }
            }
        }
    };
    if (kw == 0) epilogue(std::integral_constant<int, 0>{}); else epilogue(std::integral_constant<int, 2>{});
    CSTAMP(6);
    CS_WG(1)
}

template <bool W8, bool F16, bool DB, int NPG, int NG, int NCT = 4, bool PROD = false>
int launch_t(const ConvArgs& a, hipStream_t st) {
    using G = Geo<W8, NPG>;
    constexpr size_t lds_need = (size_t)3 * NG * (NCT * 16 * 128) + (DB ? 2 : 1) * G::PATCH;
    constexpr size_t lds_xch = (size_t)G::NW * 8192 + (size_t)G::NW * 256;     // K-half exchange + pooled-sum hand-over reuse the space
    constexpr size_t lds = lds_need > lds_xch ? lds_need : lds_xch;
    static_assert(lds <= 160 * 1024, "one CU has 160 KB of LDS");
    static_assert(!W8 || NPG % 2 == 0, "8 x 8 maps are tiled in interleaved pairs of images");
    SPSP_TRY((ensure_dynamic_lds<&conv3x3_patch_kernel<W8, F16, DB, NPG, NG, NCT, PROD>>((int)lds)));
    const int grid = (a.M / G::TPIX) * (a.N / (16 * NCT));
#ifdef SPSP_CONV_STAMPS
    if (getenv("SPSP_CONV_DEBUG")) {
        ConvArgs b = a;
        if (getenv("SPSP_PATCH_NODMA")) b.splitk = -1;
        const size_t nd = 16 + 2 * (size_t)grid;
        (void)hipMalloc(&b.dbg, nd * sizeof(unsigned long long));
        (void)hipMemset(b.dbg, 0, nd * sizeof(unsigned long long));
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipExtLaunchKernelGGL((conv3x3_patch_kernel<W8, F16, DB, NPG, NG, NCT, PROD>), dim3(grid), dim3(64 * (G::NW + (PROD ? 4 : 0))), (unsigned)lds, st, e0, e1, 0, b);
        (void)hipDeviceSynchronize();
        float kms = 0.f;
        (void)hipEventElapsedTime(&kms, e0, e1);
        std::vector<unsigned long long> hv(nd);
        (void)hipMemcpy(hv.data(), b.dbg, nd * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const unsigned long long* h = hv.data();
        fprintf(stderr, "patch<W8=%d,DB=%d,NPG=%d,NG=%d%s> chunks=%d wgs=%d: K loop %llu (of which weight wait %llu, barrier %llu) | residual loads + K-half exchange %llu | epilogue + stores %llu | total %llu cycles; one step (16 MFMAs) %llu, the next %llu\n",
                (int)W8, (int)DB, NPG, NG, PROD ? ",PROD" : "", a.chunks, grid, h[4] - h[0], h[8], h[9], h[5] - h[4], h[6] - h[5], h[6] - h[0], h[11] - h[10], h[12] - h[11]);
        {
            unsigned long long t0 = ~0ull, t1 = 0, smax = 0;
            double dur = 0;
            for (int g = 0; g < grid; ++g) {
                const unsigned long long s0 = h[16 + 2 * g], e = h[16 + 2 * g + 1];
                if (s0 < t0) t0 = s0;
                if (s0 > smax) smax = s0;
                if (e > t1) t1 = e;
                dur += (double)(e - s0);
            }
            fprintf(stderr, "  kernel %.2f us by its own timestamps; workgroups on the 100 MHz clock: first start -> last start %.2f us, first start -> last end %.2f us, mean residency %.2f us\n",
                    kms * 1e3, (smax - t0) * 0.01, (t1 - t0) * 0.01, dur / grid * 0.01);
        }
        (void)hipFree(b.dbg);
        return SPSP_OK;
    }
#endif
    launch_kernel(&conv3x3_patch_kernel<W8, F16, DB, NPG, NG, NCT, PROD>, dim3(grid), dim3(64 * (G::NW + (PROD ? 4 : 0))), lds, st, a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

template <bool W8, bool DB, int NPG, int NG, int NCT = 4, bool PROD = false>
int launch(const ConvArgs& a, hipStream_t st) {
    return a.f16 ? launch_t<W8, true, DB, NPG, NG, NCT, PROD>(a, st) : launch_t<W8, false, DB, NPG, NG, NCT, PROD>(a, st);
}

}  // namespace

namespace spsp {

bool conv_patch_supported(const ConvArgs& a) {
    return a.stride == 1 && a.kw_steps == 3 && a.nsteps == 9 * a.chunks && a.in_off == 0 && a.M % 128 == 0 && a.N % 64 == 0 &&
           a.Ho == a.Wo && (a.Wo == 8 || a.Wo % 16 == 0) && a.splitk <= 1;
}

// How the 8 XCDs are laid over the (pixel tiles) x (channel blocks) grid: XN of them along the channel blocks.  An XCD
// pulls its share of the weights (1 / XN) and of the input (XN / 8) through the fabric into its own L2; the split with
// the fewest bytes wins (layer2: 8 x 1, layer4: 2 x 4).
int conv_patch_xcd_n(const ConvArgs& a, int tile_pixels, int block_channels) {
    const int mt = a.M / tile_pixels, nb = a.N / block_channels;
    const double wbytes = (double)a.N * a.nsteps * 64 * 2, xbytes = (double)a.M * a.chunks * 64 * 2;
    int best = 0;
    double best_b = 0;
    for (int xn = 1; xn <= 8; xn *= 2) {
        const int xm = 8 / xn;
        if (nb % xn || mt % xm) continue;
        const double bytes = 8.0 * (wbytes / xn + xbytes / xm);
        if (!best || bytes < best_b) { best = xn; best_b = bytes; }
    }
    return best;      // 0: no split fits (the kernel falls back to channel-block-fastest over the whole grid)
}

int launch_conv_patch(const ConvArgs& a0, bool two_patch_buffers, int tile_pixels, int ring_groups, int block_channels, bool producer_waves, hipStream_t stream) {
    SPSP_REQUIRE(conv_patch_supported(a0), "conv_patch: unsupported convolution (need 3x3 stride-1 pad-1, M%%128==0, N%%64==0)");
    SPSP_REQUIRE(!a0.pool_out || a0.Wo == 8, "conv_patch: fused avg-pool only for 8x8 maps");
    SPSP_REQUIRE(tile_pixels == 256 || tile_pixels == 128, "conv_patch: tile of %d pixels", tile_pixels);
    SPSP_REQUIRE(a0.M % tile_pixels == 0, "conv_patch: %d pixels are no whole number of %d-pixel tiles", a0.M, tile_pixels);
    SPSP_REQUIRE(ring_groups == 2 || ring_groups == 3, "conv_patch: weight ring of %d groups", ring_groups);
    SPSP_REQUIRE(block_channels == 64 || (block_channels == 32 && a0.Wo == 8 && tile_pixels == 128 && ring_groups == 3 && a0.N % 32 == 0),
                 "conv_patch: %d-channel blocks (64; 32 for 8 x 8 maps in 128-pixel tiles with a three-group ring)", block_channels);
    ConvArgs a = a0;
    a.xcd_n = conv_patch_xcd_n(a, tile_pixels, block_channels);
    const bool db = two_patch_buffers, w8 = a.Wo == 8;
    if (producer_waves) {
        SPSP_REQUIRE(tile_pixels == 128 && ring_groups == 3 && db, "conv_patch: producer waves come with 128-pixel tiles, a three-group ring and two patch buffers");
        if (block_channels == 32) return launch<true, true, 2, 3, 2, true>(a, stream);
        return w8 ? launch<true, true, 2, 3, 4, true>(a, stream) : launch<false, true, 2, 3, 4, true>(a, stream);
    }
    if (block_channels == 32) return db ? launch<true, true, 2, 3, 2>(a, stream) : launch<true, false, 2, 3, 2>(a, stream);
    // (8 x 8 maps, 256-pixel tiles: two 50 KB patches do not fit beside a three-group ring -> one patch buffer)
#define SPSP_PICK(W8, DB, NPG, NG) return launch<W8, DB, NPG, NG>(a, stream)
    if (tile_pixels == 256) {
        if (ring_groups == 3) { if (w8) SPSP_PICK(true, false, 4, 3); if (db) SPSP_PICK(false, true, 4, 3); SPSP_PICK(false, false, 4, 3); }
        if (w8) { if (db) SPSP_PICK(true, true, 4, 2); SPSP_PICK(true, false, 4, 2); }
        if (db) SPSP_PICK(false, true, 4, 2);
        SPSP_PICK(false, false, 4, 2);
    }
    if (ring_groups == 3) {
        if (w8) { if (db) SPSP_PICK(true, true, 2, 3); SPSP_PICK(true, false, 2, 3); }
        if (db) SPSP_PICK(false, true, 2, 3);
        SPSP_PICK(false, false, 2, 3);
    }
    if (w8) { if (db) SPSP_PICK(true, true, 2, 2); SPSP_PICK(true, false, 2, 2); }
    if (db) SPSP_PICK(false, true, 2, 2);
    SPSP_PICK(false, false, 2, 2);
#undef SPSP_PICK
}

}  // namespace spsp
