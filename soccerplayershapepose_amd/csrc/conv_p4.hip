// 3x3 / stride-1 convolution of the wide maps (layer2: 32 x 32 x 128) on 256-pixel tiles WITHOUT the split of K over wave pairs:
// FOUR accumulating waves, each over the whole K of its 64 pixels x 64 channels (conv_wres.hip's wave tile and tap loop), and
// FOUR producer waves that issue every LDS-DMA piece (conv_patch.hip's PROD).
//
// Why (round 4).  conv_patch.hip runs these launches with 8 accumulating waves = 4 pixel groups x 2 K-halves.  Stamps: of ~20 k
// cycles per workgroup the K loop takes 15 k -- 4.5 k of them in the six group-top barriers (two waves per SIMD arrive a step
// apart, and nobody issues an MFMA while the last one is awaited) -- and the exchange of the K-halves' accumulators through LDS
// another 3.3 k; 9.2 k are MFMAs.  There is no room for producer waves beside eight accumulating ones (three waves per SIMD:
// <= 168 registers), so the accumulating waves also issue ~230 DMA pieces per tile.  Here a wave owns its 16 accumulator tiles
// from the first product to the store (no exchange, epilogue straight from registers as in conv_wres.hip), the DMA pieces are
// somebody else's instruction stream, and there is ONE barrier per group of three taps, placed where conv_s2.hip puts it: right
// behind the reads of the group's LAST tap -- by then every wave holds the group's last fragments in registers, so the barrier
// releases the group's ring third (the producers refill it at once with the group three ahead) and publishes the next group,
// whose first reads go out beside the pending MFMAs.
//
// Round 4, measured (profiles/r04_p4_stamps.txt, r04_ab_p4.txt): K loop 15.5 k -> 11.2 k cycles per tile, layer2's launch 13.1 -> 11.7 us;
// layer3 as one image per workgroup (128 workgroups at B = 32) is slower alone (12.0 -> 14.9 us) and 40 % less CU time: plan
// "throughput" only.  Launched with fewer workgroups than tiles the kernel is persistent (tile loop below).
//
// LDS: ring of 9 weight slabs (three groups of three taps; the slab of tap t of ANY chunk sits in slot t) 72 KB + two 18 x 18
// patches (one per chunk parity) 82 KB.  Patch image, swizzles, permuted weight rows: conv_patch.hip / conv_wres.hip.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "conv.hpp"
#include "conv_elem.hpp"

namespace {
using namespace spsp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// byte address ^ 64 = the K-half + 1 fragment of the same row (swizzled 16-byte chunk index + 4), as an instruction that stays where
// it is written: the 18 pixel addresses ^ 64 do not fit beside the 18 (the kernel lives in the 256 registers two waves per SIMD leave
// it), and formed next to their reads (8 v_xor in the steps' first halves, whose issue slots the 16 ds_reads fill) they cost ~2 k
// cycles per tile -- the four the NEXT tap needs are formed in the bare second half of each step
__device__ __forceinline__ uint32_t xor64_here(uint32_t addr) {
    uint32_t t;
    asm volatile("v_xor_b32 %0, 64, %1" : "=v"(t) : "v"(addr));
    return t;
}
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

constexpr int kSlab = 64 * 128;                  // one tap: [64 n][64 k] 16-bit
constexpr int kRing = 9 * kSlab;                 // 73 728 B
constexpr int kNPix = 324, kNDma = 41;           // 18 x 18 patch, 8 pixels per 1-KiB DMA piece
constexpr int kPatch = kNDma * 1024;             // 41 984 B
constexpr int kLds = kRing + 2 * kPatch;         // 157 696 B
static_assert(kLds <= 160 * 1024, "one CU has 160 KB of LDS");

#ifdef SPSP_CONV_STAMPS
#define P4STAMP(i) if (a.dbg && blockIdx.x == 8 && tid == 0) a.dbg[i] = __builtin_amdgcn_s_memtime()
#define P4_BAR { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_barrier(); p4_bar_ += __builtin_amdgcn_s_memtime() - t_; }
#define P4_WG(i) if (a.dbg && tid == 0) a.dbg[16 + 2 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime();
#else
#define P4STAMP(i)
#define P4_BAR __builtin_amdgcn_s_barrier();
#define P4_WG(i)
#endif

// output channel held by accumulator tile i, MFMA row r (= 4 fq + e): conv_wres.hip / conv_patch.hip
__host__ __device__ constexpr int acc_channel(int i, int r) { return 32 * (i >> 1) + 8 * (r >> 2) + 4 * (i & 1) + (r & 3); }

template <bool F16>
__global__ __launch_bounds__(512) void conv3x3_p4_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const patch0 = smem + kRing;

    const int tid = threadIdx.x, lane = tid & 63;
    P4_WG(0)
    P4STAMP(0);
#ifdef SPSP_CONV_STAMPS
    unsigned long long p4_bar_ = 0;
#endif
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave_id >= 4;
    const int w = wave_id & 3;                                     // consumer: pixel rows 4 w .. 4 w + 3; producer: its share of the pieces
    const int W = a.Wo;
    const int mt = a.M / 256, nb = a.N / 64;
    const int tpr = W / 16, tpi = tpr * tpr;
    const int nchunks = a.chunks, ngroups = 3 * nchunks;
    const int Ktot = a.nsteps * 64;
    const int drow = lane >> 3, dchunk = lane & 7;
    // PERSISTENT over tiles blockIdx.x + i * gridDim.x (plan "throughput" launches fewer workgroups than tiles: the DMA stream
    // runs on across the tile boundary -- the next tile's first patch and groups land under this tile's last taps and epilogue,
    // so only a workgroup's FIRST tile pays the ~5 k cycles of L2 -> LDS time before the first MFMA)
    const int nwg = gridDim.x;
    const int ntiles = (mt * nb - (int)blockIdx.x + nwg - 1) / nwg;
    struct Tile { int img0, y0, x0, n_blk; };
    auto tile_of = [&](int i) {
        const int v = blockIdx.x + i * nwg;
        int m_tile, n_blk;
        // XCD-aware tile order, as conv_patch.hip (gridDim.x is a multiple of 8 whenever it is below the tile count: v & 7 = the XCD)
        const int XN = a.xcd_n, XM = XN > 0 ? 8 / XN : 0;
        if (XN > 0 && mt % XM == 0 && nb % XN == 0) {
            const int x = v & 7, li = v >> 3, nbl = nb / XN;
            m_tile = (x / XN) * (mt / XM) + li / nbl;
            n_blk = ((x % XN) * nbl + li % nbl) * 64;
        } else {
            m_tile = v / nb;
            n_blk = (v % nb) * 64;
        }
        const int img0 = m_tile / tpi, tt = m_tile - img0 * tpi;
        return Tile{img0, (tt / tpr) * 16, (tt % tpr) * 16, n_blk};
    };

    if (producer) {
        // ---- the DMA side.  Patch: lane l of piece p feeds patch pixel q = 8 p + l / 8, physical chunk l % 8 (logical ^ q & 7);
        //      producer w issues pieces w, w + 4, .. (11 or 10).  Weights: a group = 3 slabs = 24 pieces of 8 rows, six per producer;
        //      LDS row R of a slab holds output channel acc_channel(R / 16, R % 16).
        //      Groups and chunks are numbered through ALL the workgroup's tiles: G = tile * ngroups + g, C = tile * nchunks + ch
        //      (ngroups is a multiple of 3 and nchunks even: ring third G % 3 and patch buffer C & 1 carry on across tiles).
        const int Gtot = ntiles * ngroups, Ctot = ntiles * nchunks;
        const uint16_t* const in_ptr = a.in;
        int p_src[11];
        int p_tile = -1;
        auto issue_patch = [&](int C) {
            const int i = C / nchunks, ch = C - i * nchunks;
            if (i != p_tile) {
                p_tile = i;
                const Tile t = tile_of(i);
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    int q = (w + 4 * k) * 8 + drow;
                    q = q < kNPix ? q : kNPix - 1;
                    const int py = q / 18, px = q - py * 18;
                    p_src[k] = t.img0 * a.in_img + (t.y0 + py) * a.in_row + (t.x0 + px) * a.in_pix + ((dchunk ^ (q & 7)) * 8);
                }
            }
            unsigned char* const buf = patch0 + (ch & 1) * kPatch;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const int p = w + 4 * k;
                if (p < kNDma) dma16(in_ptr + p_src[k] + ch * 64, buf + p * 1024);
            }
        };
        const uint16_t* w_src[6];
        int w_slot[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int j = w + 4 * k;
            w_slot[k] = (j >> 3) * kSlab + (j & 7) * 1024;
        }
        int w_tile = -1;
        auto issue_group = [&](int G) {          // group g = (chunk g / 3, kernel row g % 3): taps 3 r + t, K-step (tap * nchunks + chunk)
            const int i = G / ngroups, g = G - i * ngroups;
            if (i != w_tile) {
                w_tile = i;
                const int n_blk = tile_of(i).n_blk;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int row = ((w + 4 * k) & 7) * 8 + drow;
                    w_src[k] = a.w + (size_t)(n_blk + acc_channel(row >> 4, row & 15)) * Ktot + ((dchunk ^ (row & 7)) * 8);
                }
            }
            const int ch = g / 3, r = g - 3 * ch;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int t = (w + 4 * k) >> 3;
                dma16(w_src[k] + (size_t)((3 * r + t) * nchunks + ch) * 64, ring + r * 3 * kSlab + w_slot[k]);
            }
        };
        issue_patch(0);
        issue_group(0);
        issue_group(1);
        issue_group(2);
        // a producer's six pieces of a group are two per tap, in tap order: the consumers start on the first patch + tap 0's slab
        // (49 of the 65 KB that patch + group 0 come to: the prologue is L2 -> LDS time, ~12 B / cycle with every CU in it at once)
        wait_vmcnt<16>();                        // the first patch and tap 0 (behind them: taps 1, 2, groups 1 and 2)
        __builtin_amdgcn_s_barrier();
        wait_vmcnt<14>();                        // tap 1
        __builtin_amdgcn_s_barrier();
        wait_vmcnt<12>();                        // tap 2
        __builtin_amdgcn_s_barrier();
        // barrier B_G (the consumers' barrier behind the reads of group G's last tap): group G + 1 (and, at a chunk's end, the next
        // chunk's patch) has landed -- what was issued at B_(G-1) may stay in flight (counts below: a wave's MINIMUM, wave 0 issues
        // 11 patch pieces: a count below the truth only waits for more); then the ring third of group G is refilled with group
        // G + 3 and, at a chunk's first group, the other patch buffer with the next chunk's patch
        for (int G = 0; G + 1 < Gtot; ++G) {
            const bool more = G + 2 < Gtot;
            const bool with_patch = G >= 1 && (G - 1) % 3 == 0 && (G - 1) / 3 + 1 < Ctot;
            if (!more) wait_vmcnt<0>();
            else if (with_patch) wait_vmcnt<16>();
            else wait_vmcnt<6>();
            __builtin_amdgcn_s_barrier();
            if (G + 3 < Gtot) issue_group(G + 3);
            if (G % 3 == 0 && G / 3 + 1 < Ctot) issue_patch(G / 3 + 1);
        }
        return;
    }

    // ---- the accumulating side (conv_wres.hip's tap loop) ----
    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    const uint32_t w_lane = smem_base + frow * 128 + ((fq ^ (frow & 7)) * 16);       // K-half 0; K-half 1: the chunk index + 4 = ^ 64
    const uint32_t w_lane1 = w_lane ^ 64u;
    const uint32_t w_lane8 = w_lane + 8 * kSlab, w_lane81 = w_lane1 + 8 * kSlab;
    const int qb = (w * 4) * 18 + frow;
    uint32_t xo[6][3];                      // patch pixel (4 w + r, frow + dx), r = j + ky: 18 addresses serve the 9 x 4 (tap, j) pairs
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int q = qb + r * 18 + dx;
            xo[r][dx] = smem_base + kRing + q * 128 + ((fq ^ (q & 7)) * 16);
        }
    f32x4 acc[4][4];
    bf16x8 wfr[2][2][4], xfr[2][2][4];
    uint32_t xa[4];                         // ^ 64 of the four pixel addresses of the tap being read next
#define P4_SB __builtin_amdgcn_sched_barrier(0);
    // (the ds offset field has 16 bits: tap 8's slab is addressed from its own base register)
#define P4_WB(KK, TAP) ((TAP) == 8 ? ((KK) ? w_lane81 : w_lane8) : ((KK) ? w_lane1 : w_lane))
#define P4_WO(I, TAP) ((I) * 2048 + ((TAP) == 8 ? 0 : (TAP) * kSlab))
#define P4_RW(SET, KK, I, TAP) wfr[SET][KK][I] = lds_read_b128<P4_WO(I, TAP)>(P4_WB(KK, TAP));
#define P4_RX(SET, KK, J, TAP, POFF) xfr[SET][KK][J] = lds_read_b128<POFF>((KK) ? xa[J] : xo[(J) + (TAP) / 3][(TAP) % 3]);
#define P4_XA(J, TAPN) xa[J] = xor64_here(xo[(J) + (TAPN) / 3][(TAPN) % 3]);
#define P4_M2(S1, KK, I, J)                                                                    \
        P4_SB                                                                                  \
        acc[I][J] = Elem<F16>::mfma(wfr[S1][KK][I], xfr[S1][KK][J], acc[I][J]);                \
        acc[I][(J) + 1] = Elem<F16>::mfma(wfr[S1][KK][I], xfr[S1][KK][(J) + 1], acc[I][(J) + 1]); \
        P4_SB
    // One tap: the 16 fragment reads of tap TAP of the patch at POFF (set SET) between the first 16 MFMAs of the tap before it
    // (set SET ^ 1), one read pair per MFMA pair, pinned; its last 16 MFMAs run bare.
#define P4_STEP(TAP, SET, POFF)                                                                \
    {                                                                                          \
        P4_RW(SET, 0, 0, TAP) P4_RW(SET, 0, 1, TAP) P4_M2((SET) ^ 1, 0, 0, 0)                   \
        P4_RW(SET, 0, 2, TAP) P4_RW(SET, 0, 3, TAP) P4_M2((SET) ^ 1, 0, 0, 2)                   \
        P4_RX(SET, 0, 0, TAP, POFF) P4_RX(SET, 0, 1, TAP, POFF) P4_M2((SET) ^ 1, 0, 1, 0)       \
        P4_RX(SET, 0, 2, TAP, POFF) P4_RX(SET, 0, 3, TAP, POFF) P4_M2((SET) ^ 1, 0, 1, 2)       \
        P4_RW(SET, 1, 0, TAP) P4_RW(SET, 1, 1, TAP) P4_M2((SET) ^ 1, 0, 2, 0)                   \
        P4_RW(SET, 1, 2, TAP) P4_RW(SET, 1, 3, TAP) P4_M2((SET) ^ 1, 0, 2, 2)                   \
        P4_RX(SET, 1, 0, TAP, POFF) P4_RX(SET, 1, 1, TAP, POFF) P4_M2((SET) ^ 1, 0, 3, 0)       \
        P4_RX(SET, 1, 2, TAP, POFF) P4_RX(SET, 1, 3, TAP, POFF) P4_M2((SET) ^ 1, 0, 3, 2)       \
        P4_M2((SET) ^ 1, 1, 0, 0) P4_XA(0, ((TAP) + 1) % 9) P4_M2((SET) ^ 1, 1, 0, 2)           \
        P4_XA(1, ((TAP) + 1) % 9) P4_M2((SET) ^ 1, 1, 1, 0) P4_XA(2, ((TAP) + 1) % 9) P4_M2((SET) ^ 1, 1, 1, 2) \
        P4_XA(3, ((TAP) + 1) % 9) P4_M2((SET) ^ 1, 1, 2, 0) P4_M2((SET) ^ 1, 1, 2, 2) P4_M2((SET) ^ 1, 1, 3, 0) P4_M2((SET) ^ 1, 1, 3, 2) \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
        P4_SB                                                                                  \
    }
    // the 32 MFMAs of a tile's last tap (set SET)
#define P4_LAST(SET)                                                                           \
    {                                                                                          \
        P4_M2(SET, 0, 0, 0) P4_M2(SET, 0, 0, 2) P4_M2(SET, 0, 1, 0) P4_M2(SET, 0, 1, 2)         \
        P4_M2(SET, 0, 2, 0) P4_M2(SET, 0, 2, 2) P4_M2(SET, 0, 3, 0) P4_M2(SET, 0, 3, 2)         \
        P4_M2(SET, 1, 0, 0) P4_M2(SET, 1, 0, 2) P4_M2(SET, 1, 1, 0) P4_M2(SET, 1, 1, 2)         \
        P4_M2(SET, 1, 2, 0) P4_M2(SET, 1, 2, 2) P4_M2(SET, 1, 3, 0) P4_M2(SET, 1, 3, 2)         \
    }
    // taps 1..8 of a chunk whose tap 0 sits in set S0 (= chunk parity: nine taps per chunk) and whose patch is at POFF; the group
    // barriers behind the reads of taps 2, 5 and (unless it is the last chunk of the workgroup's last tile) 8
#define P4_CHUNK(S0, POFF, VERYLAST, FIRSTCH)                                                  \
    {                                                                                          \
        if (FIRSTCH) P4_BAR                                                                    \
        P4_STEP(1, (S0) ^ 1, POFF)                                                             \
        if (FIRSTCH) P4_BAR                                                                    \
        P4_STEP(2, S0, POFF)                                                                   \
        P4_BAR                                                                                 \
        P4_STEP(3, (S0) ^ 1, POFF) P4_STEP(4, S0, POFF) P4_STEP(5, (S0) ^ 1, POFF)             \
        P4_BAR                                                                                 \
        P4_STEP(6, S0, POFF) P4_STEP(7, (S0) ^ 1, POFF) P4_STEP(8, S0, POFF)                   \
        if (!(VERYLAST)) P4_BAR                                                                \
    }
    for (int it = 0; it < ntiles; ++it) {
        const Tile tl = tile_of(it);
        // the accumulators start at the folded-BN bias of the tile's rows (channels acc_channel(i, 4 fq + e))
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 b = *reinterpret_cast<const float4*>(a.bias + tl.n_blk + 32 * (i >> 1) + 8 * fq + 4 * (i & 1));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{b.x, b.y, b.z, b.w};
        }
        // first tile: the first patch and tap 0's slab have landed (taps 1 and 2: a barrier each, first chunk only); later tiles:
        // the barrier behind the previous tile's last reads has published this tile's first patch and group
        if (it == 0) __builtin_amdgcn_s_barrier();
        P4STAMP(1);
        // tap 0 of chunk 0 -> set 0
        P4_XA(0, 0) P4_XA(1, 0) P4_XA(2, 0) P4_XA(3, 0)
        P4_RW(0, 0, 0, 0) P4_RW(0, 0, 1, 0) P4_RW(0, 0, 2, 0) P4_RW(0, 0, 3, 0)
        P4_RX(0, 0, 0, 0, 0) P4_RX(0, 0, 1, 0, 0) P4_RX(0, 0, 2, 0, 0) P4_RX(0, 0, 3, 0, 0)
        P4_RW(0, 1, 0, 0) P4_RW(0, 1, 1, 0) P4_RW(0, 1, 2, 0) P4_RW(0, 1, 3, 0)
        P4_RX(0, 1, 0, 0, 0) P4_RX(0, 1, 1, 0, 0) P4_RX(0, 1, 2, 0, 0) P4_RX(0, 1, 3, 0, 0)
        P4_XA(0, 1) P4_XA(1, 1) P4_XA(2, 1) P4_XA(3, 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        P4_SB
        for (int ch = 0; ch < nchunks; ch += 2) {          // (nchunks is even: two chunks per round, sets and patch buffers alternate)
            P4_CHUNK(0, 0, false, it == 0 && ch == 0)
            P4_STEP(0, 1, kPatch)                          // tap 0 of the odd chunk (patch buffer 1) beside tap 8 of the even one
            const bool last = ch + 2 >= nchunks;
            P4_CHUNK(1, kPatch, last && it + 1 == ntiles, false)
            if (!last) P4_STEP(0, 0, 0)                    // tap 0 of the next even chunk beside tap 8 of the odd one
        }
        P4STAMP(2);
        // the epilogue's loads go out before the last tap's MFMAs (one fragment set is dead by now: its registers take them)
        const size_t org = (size_t)a.out_off + (size_t)tl.img0 * a.out_img + (size_t)(tl.y0 + 4 * w) * a.out_row + (size_t)(tl.x0 + frow) * a.out_pix + tl.n_blk + 8 * fq;
        uint4 resv[8];
        if (a.res) {
#pragma unroll
            for (int c = 0; c < 8; ++c) resv[c] = *reinterpret_cast<const uint4*>(a.res + org + (size_t)(c >> 1) * a.out_row + 32 * (c & 1));
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) resv[c] = make_uint4(0u, 0u, 0u, 0u);
        }
        P4_SB
        P4_LAST(1)                                         // tap 8 of the last (odd) chunk: set (1 + 8) & 1 = 1
        P4STAMP(3);
        // ---- epilogue straight from registers (conv_wres.hip): accumulator tiles (2 h, 2 h + 1) of pixel row j hold channels
        //      n_blk + 32 h + 8 fq .. + 7 of pixel (4 w + j, frow): residual, ReLU, pack, one 16-byte store per piece
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int j = c >> 1, h = c & 1;
            const size_t off = org + (size_t)j * a.out_row + 32 * h;
            const f32x4 lo4 = acc[2 * h][j], hi4 = acc[2 * h + 1][j];
            const uint4 r = resv[c];            // (zeros without a residual: +0.0 in either element type)
            float v[8] = {lo4[0] + Elem<F16>::lo(r.x), lo4[1] + Elem<F16>::hi(r.x), lo4[2] + Elem<F16>::lo(r.y), lo4[3] + Elem<F16>::hi(r.y),
                          hi4[0] + Elem<F16>::lo(r.z), hi4[1] + Elem<F16>::hi(r.z), hi4[2] + Elem<F16>::lo(r.w), hi4[3] + Elem<F16>::hi(r.w)};
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            *reinterpret_cast<uint4*>(a.out + off) =
                make_uint4(Elem<F16>::pack2(v[0], v[1]), Elem<F16>::pack2(v[2], v[3]), Elem<F16>::pack2(v[4], v[5]), Elem<F16>::pack2(v[6], v[7]));
        }
        P4_SB
        P4STAMP(4);
    }
#undef P4_CHUNK
#undef P4_LAST
#undef P4_STEP
#undef P4_M2
#undef P4_XA
#undef P4_RX
#undef P4_RW
#undef P4_WO
#undef P4_WB
#undef P4_SB
#ifdef SPSP_CONV_STAMPS
    if (a.dbg && blockIdx.x == 8 && tid == 0) a.dbg[8] = p4_bar_;
#endif
    P4_WG(1)
}

}  // namespace

namespace spsp {

// a 3x3 stride-1 pad-1 convolution on maps at least 16 wide, whole 256-pixel tiles, an even number of 64-channel input chunks
bool conv_p4_supported(const ConvArgs& a) {
    return a.stride == 1 && a.kw_steps == 3 && a.nsteps == 9 * a.chunks && a.chunks >= 2 && a.chunks % 2 == 0 && a.in_off == 0 &&
           a.M % 256 == 0 && a.N % 64 == 0 && a.Ho == a.Wo && a.Wo % 16 == 0 && !a.pool_out && a.out && a.in_pix == a.chunks * 64;
}

int launch_conv_p4(const ConvArgs& a0, int max_workgroups, hipStream_t stream) {
    SPSP_REQUIRE(conv_p4_supported(a0), "conv_p4: needs a 3x3 stride-1 convolution on maps >= 16 wide, whole 256-pixel tiles, an even number of input chunks");
    ConvArgs a = a0;
    a.xcd_n = conv_patch_xcd_n(a, 256, 64);
    // fewer workgroups than tiles: persistent, tiles blockIdx.x + i * grid (a multiple of 8: a workgroup's tiles stay on its XCD)
    const int tiles = (a.M / 256) * (a.N / 64);
    const int cap = max_workgroups >= 8 ? max_workgroups / 8 * 8 : 8;
    const int grid = tiles <= cap ? tiles : cap;
    SPSP_TRY((ensure_dynamic_lds<&conv3x3_p4_kernel<true>>(kLds)));
    SPSP_TRY((ensure_dynamic_lds<&conv3x3_p4_kernel<false>>(kLds)));
#ifdef SPSP_CONV_STAMPS
    if (getenv("SPSP_CONV_DEBUG")) {
        const size_t nd = 16 + 2 * (size_t)grid;
        (void)hipMalloc(&a.dbg, nd * sizeof(unsigned long long));
        (void)hipMemset(a.dbg, 0, nd * sizeof(unsigned long long));
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        if (a.f16) hipExtLaunchKernelGGL((conv3x3_p4_kernel<true>), dim3(grid), dim3(512), (unsigned)kLds, stream, e0, e1, 0, a);
        else hipExtLaunchKernelGGL((conv3x3_p4_kernel<false>), dim3(grid), dim3(512), (unsigned)kLds, stream, e0, e1, 0, a);
        (void)hipDeviceSynchronize();
        float kms = 0.f;
        (void)hipEventElapsedTime(&kms, e0, e1);
        std::vector<unsigned long long> hv(nd);
        (void)hipMemcpy(hv.data(), a.dbg, nd * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const unsigned long long* h = hv.data();
        fprintf(stderr, "p4 chunks=%d wgs=%d: prologue (first patch + group 0) %llu | K loop %llu (of which in barriers %llu) | last tap %llu | epilogue %llu | total %llu cycles\n",
                a.chunks, grid, h[1] - h[0], h[2] - h[1], h[8], h[3] - h[2], h[4] - h[3], h[4] - h[0]);
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
        (void)hipFree(a.dbg);
        return SPSP_OK;
    }
#endif
    if (a.f16) launch_kernel(&conv3x3_p4_kernel<true>, dim3(grid), dim3(512), kLds, stream, a);
    else launch_kernel(&conv3x3_p4_kernel<false>, dim3(grid), dim3(512), kLds, stream, a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp
