// The 3x3 / STRIDE-2 convolution + fused 1x1 / stride-2 downsample of layer2's first BasicBlock (PR/models/resnet.py:177-199:
// 64 -> 128 channels, 64 x 64 -> 32 x 32) with the WEIGHTS STATIONARY IN REGISTERS.
//
// Why a second form of conv_s2.hip for this one layer.  With C_in = 64 a 64-channel block of both convolutions has only
// 64 x (9 + 1) x 64 weights = 80 KB, and a wave that takes one 32-deep half of every K-step needs 40 MFMA A-fragments of them:
// 160 registers of the 512 a lone wave per SIMD owns.  conv_s2.hip streams those 80 KB through a ten-slot LDS ring for EVERY
// 128-pixel tile and a workgroup lives for one tile: 2.6 k cycles of MFMAs inside ~20 k cycles of waiting for 152 KB (0.13 of
// the MFMA peak).  Here (the stem's scheme, conv_stem.hip)
//   * a workgroup fetches its block's weights ONCE (LDS-DMA into the part of LDS the first tile does not use, 40 conflict-free
//     ds_read_b128 per wave) and keeps them as A-fragments for all its tiles: no weight ring, no weight reads and no group
//     barriers in the K loop -- a K-step is 4 pixel-fragment reads beside 16 MFMAs;
//   * LDS then holds nothing but pixels, so BOTH tiles of a workgroup are requested up front (2 x 71 KB of parity planes): the
//     second tile's planes land under the first tile's MFMAs and epilogue;
//   * everything else is conv_s2.hip's: the four parity planes (every tap reads 16 consecutive plane pixels: conflict-free),
//     the K order by plane, waves = 2 pixel groups x 2 K-halves with the accumulator exchange at the end of a tile (through the
//     first tile's dead plane buffer), permuted weight rows so that a lane stores 8 consecutive channels.
// Workgroup = 2 tiles (1 for small batches; 4 or 8 in plan "throughput": tile t + 2 is requested into the buffer tile t has left, the
// K-half exchange goes through the finished tile's own buffer) of 128 output pixels x one 64-channel block.
#include <type_traits>

#include "conv.hpp"
#include "conv_elem.hpp"

namespace {
using namespace spsp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 lds_read_b128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void dma16(const uint16_t* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int kSlab = 64 * 128;              // [64 n][64 k] 16-bit
constexpr int kSteps = 10;                   // K-steps: nine taps + the downsample
constexpr int kPixB = 128;                   // bytes per patch pixel (64 channels)

// output channel (within the 64-channel block) held by accumulator tile i, MFMA row r -- as conv_patch.hip / conv_s2.hip
__host__ __device__ constexpr int acc_channel(int i, int r) { return 32 * (i >> 1) + 8 * (r >> 2) + 4 * (i & 1) + (r & 3); }

// planes in consumption order: 0 = (1,1), 1 = (1,0), 2 = (0,1), 3 = (0,0); tile = 16 x 8 output pixels (conv_s2.hip, S2Geo<false>)
struct Geo {
    static constexpr int py(int p) { return p < 2 ? 1 : 0; }
    static constexpr int px(int p) { return (p == 0 || p == 2) ? 1 : 0; }
    static constexpr int rows(int p) { return 8 + (py(p) == 0 ? 1 : 0); }
    static constexpr int cw(int p) { return 16 + (px(p) == 0 ? 1 : 0); }
    static constexpr int npix(int p) { return rows(p) * cw(p); }
    static constexpr int npiece(int p) { return (npix(p) + 7) / 8; }
    static constexpr int dpw(int p) { return (npiece(p) + 3) / 4; }
    static constexpr int base_piece(int p) { return p == 0 ? 0 : base_piece(p - 1) + npiece(p - 1); }
};
constexpr int kPatch = (Geo::base_piece(3) + Geo::npiece(3)) * 1024;                  // one tile's four planes: 72 704 B
constexpr int kPieces = Geo::dpw(0) + Geo::dpw(1) + Geo::dpw(2) + Geo::dpw(3);        // LDS-DMA pieces one wave issues per tile: 19
constexpr int kLds = kPatch + kSteps * kSlab;           // tile 0 | tile 1 (weight staging over it and the 9 KB behind): 154 624
static_assert(2 * kPatch <= kLds && kLds <= 160 * 1024, "LDS plan");
static_assert(kPatch >= 2 * 4 * 8 * 1024, "K-half exchange of two accumulator sets in the first tile's buffer");

struct RsTileMap {      // which tiles / channel block a workgroup takes (host and device agree through these three numbers)
    int tiles_per_wg, paired;
};

template <bool F16>
__global__ __launch_bounds__(256) void conv3x3s2_kernel(const ConvArgs a, const RsTileMap tm) {
    using G = Geo;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const stage = smem + kPatch;             // ten 8 KB weight slabs, only during the prologue

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave_id & 1;                                // which 64 pixels of the tile
    const int kw = wave_id >> 1;                               // which 32-deep half of every K-step
    // Workgroups b and b + 8 share an XCD (round-robin placement; speed only): they take the two channel blocks of the SAME
    // pixel tiles, so the planes come out of one L2.
    int n_blk, unit;
    {
        const int bid = blockIdx.x;
        if (tm.paired) { n_blk = ((bid >> 3) & 1) * 64; unit = (bid & 7) + 8 * (bid >> 4); }
        else { n_blk = (bid & 1) * 64; unit = bid >> 1; }
    }
    const int ntiles = tm.tiles_per_wg;
    const int So = a.Wo, tpr = So / 16, tpi = tpr * (So / 8);
    const int Ktot = a.nsteps * 64;                            // 576
    const uint16_t* const in_ptr = a.in;

    // ---- patch LDS-DMA sources, relative to the tile's first input pixel: lane l of piece p of a plane feeds plane pixel
    //      q = 8 p + l / 8, physical 16-byte chunk l % 8 (logical chunk XOR q & 7)
    const int drow = lane >> 3, dchunk = lane & 7;
    int p_rel[4][5];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            p_rel[p][i] = 0;
            if (i < G::dpw(p)) {
                int piece = wave_id + 4 * i;
                piece = piece < G::npiece(p) ? piece : G::npiece(p) - 1;
                int q = piece * 8 + drow;
                q = q < G::npix(p) ? q : G::npix(p) - 1;          // tail lanes re-read the last pixel (harmless)
                const int r = q / G::cw(p), c = q - r * G::cw(p);
                p_rel[p][i] = (2 * r + G::py(p)) * a.in_row + (2 * c + G::px(p)) * a.in_pix + ((dchunk ^ (q & 7)) * 8);
            }
        }
    auto tile_origin = [&](int t, int& img0, int& y0, int& x0) {
        const int m_tile = unit * ntiles + t;
        img0 = m_tile / tpi;
        const int tt = m_tile - img0 * tpi;
        y0 = (tt / tpr) * 8; x0 = (tt % tpr) * 16;
    };
    auto issue_tile = [&](int t) {                              // all four planes of tile t into buffer t, in consumption order
        int img0, y0, x0;
        tile_origin(t, img0, y0, x0);
        const uint16_t* const base = in_ptr + (size_t)img0 * a.in_img + (size_t)(2 * y0) * a.in_row + (size_t)(2 * x0) * a.in_pix;
        unsigned char* const buf = smem + (t & 1) * kPatch;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 5; ++i)
                if (i < G::dpw(p)) {
                    int piece = wave_id + 4 * i;
                    piece = piece < G::npiece(p) ? piece : G::npiece(p) - 1;   // every wave issues exactly dpw pieces (counted waits)
                    dma16(base + p_rel[p][i], buf + (G::base_piece(p) + piece) * 1024);
                }
    };

    // ---- prologue: the block's weights (ten slabs, conv_s2.hip's ring image: LDS row R = output channel acc_channel(R / 16, R % 16),
    //      chunk XOR row & 7) and the first tile's planes
    {
        constexpr int taps[kSteps] = {4, -1, 3, 5, 1, 7, 0, 6, 2, 8};     // step s = filter tap ky * 3 + kx; -1 = the downsample
#pragma unroll
        for (int s = 0; s < kSteps; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wave_id * 2 + i) * 8 + drow;
                const int chn = n_blk + acc_channel(row >> 4, row & 15);
                const uint16_t* src = taps[s] < 0 ? a.w2 + (size_t)chn * 64 + ((dchunk ^ (row & 7)) * 8)
                                                  : a.w + (size_t)chn * Ktot + (size_t)taps[s] * 64 + ((dchunk ^ (row & 7)) * 8);
                dma16(src, stage + s * kSlab + (wave_id * 2 + i) * 1024);
            }
    }
    issue_tile(0);
    wait_vmcnt<kPieces>();                                   // this wave's slab pieces have landed (LDS-DMA completes in order)
    __builtin_amdgcn_s_barrier();

    const int frow = lane & 15, fq = lane >> 4;
    // this wave's 40 weight fragments, resident for the whole kernel: slab row i * 16 + frow, chunk (kw * 4 + fq) ^ (frow & 7).
    // Plain LDS loads: the compiler places their waits itself (an asm read followed by a separate asm wait lets it copy a
    // long-lived fragment to another register before the data is there -- conv_stem.hip).
    bf16x8 wreg[kSteps][4];
    {
        const unsigned char* const wl = stage + frow * 128 + ((((kw * 4 + fq) ^ (frow & 7))) * 16);
#pragma unroll
        for (int s = 0; s < kSteps; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) wreg[s][i] = *reinterpret_cast<const bf16x8*>(wl + s * kSlab + i * 2048);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // every wave has its fragments: the staging area is free
    if (ntiles > 1) issue_tile(1);

    // pixel fragment addresses inside a plane buffer: combo = (plane, dx), row = 4 pg + (j + dy); MFMA tile j of tap (dy, dx) reads
    // plane pixels q = (4 pg + j + dy) * cw + frow + dx .. (16 consecutive), chunk (kw * 4 + fq) ^ (q & 7)
    //   combo 0: plane (1,1)   1: plane (1,0) dx 0   2: plane (1,0) dx 1   3: plane (0,1)   4: plane (0,0) dx 0   5: plane (0,0) dx 1
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    uint32_t xrel[6][5];
    {
        constexpr int cplane[6] = {0, 1, 1, 2, 3, 3}, cdx[6] = {0, 0, 1, 0, 0, 1};
#pragma unroll
        for (int cb = 0; cb < 6; ++cb)
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int p = cplane[cb];
                int row = 4 * pg + r;
                row = row < G::rows(p) ? row : G::rows(p) - 1;       // (row 4 of an 8-row plane is never read)
                const int q = row * G::cw(p) + frow + cdx[cb];
                xrel[cb][r] = smem_base + G::base_piece(p) * 1024 + q * kPixB + ((((kw * 4 + fq) ^ (q & 7))) * 16);
            }
    }
    const int cbase = n_blk + 32 * kw + 8 * fq;
    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + cbase), b1 = *reinterpret_cast<const float4*>(a.bias + cbase + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(a.bias2 + cbase), c1 = *reinterpret_cast<const float4*>(a.bias2 + cbase + 4);

    for (int t = 0; t < ntiles; ++t) {
        // ---- this tile's planes: everything this wave has issued for it has landed, then everybody's
        if (t + 1 < ntiles) wait_vmcnt<kPieces>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const uint32_t bufo = (uint32_t)((t & 1) * kPatch);

        f32x4 acc[4][4], acc2[4][4];

        bf16x8 xfr[2][4];
#define RS_SB __builtin_amdgcn_sched_barrier(0);
#define RS_RX(SET, J, CB, DY) xfr[SET][J] = lds_read_b128(xrel[CB][(J) + (DY)] + bufo);
#define RS_M2(S, SET, ACC, I, J)                                                                        \
        RS_SB                                                                                           \
        if constexpr (RS_FIRST) {    /* a chain's first K-step: C = 0 in the instruction, no VALU zeroing (conv_elem.hpp) */ \
            Elem<F16>::mfma_agpr_zero(ACC[I][J], wreg[S][I], xfr[SET][J]);                              \
            Elem<F16>::mfma_agpr_zero(ACC[I][(J) + 1], wreg[S][I], xfr[SET][(J) + 1]);                  \
        } else {                                                                                        \
            Elem<F16>::mfma_agpr(ACC[I][J], wreg[S][I], xfr[SET][J]);                                   \
            Elem<F16>::mfma_agpr(ACC[I][(J) + 1], wreg[S][I], xfr[SET][(J) + 1]);                       \
        }                                                                                               \
        RS_SB
        // One K-step S: its 16 MFMAs (pixel fragments in set SET, into ACC) with the four pixel-fragment reads of the step
        // after next-in-the-other-set (combo NCB, row shift NDY; skipped unless HASN) beside the first four MFMA pairs.
#define RS_STEP(S, SET, ACC, HASN, NSET, NCB, NDY)                                                      \
    {                                                                                                   \
        constexpr bool RS_FIRST = (S) == 0 || (S) == 1;      /* ring step 0 opens the main chain, 1 (the downsample) the second */ \
        if (HASN) { RS_RX(NSET, 0, NCB, NDY) }                                                          \
        RS_M2(S, SET, ACC, 0, 0)                                                                        \
        if (HASN) { RS_RX(NSET, 1, NCB, NDY) }                                                          \
        RS_M2(S, SET, ACC, 0, 2)                                                                        \
        if (HASN) { RS_RX(NSET, 2, NCB, NDY) }                                                          \
        RS_M2(S, SET, ACC, 1, 0)                                                                        \
        if (HASN) { RS_RX(NSET, 3, NCB, NDY) }                                                          \
        RS_M2(S, SET, ACC, 1, 2)                                                                        \
        RS_M2(S, SET, ACC, 2, 0) RS_M2(S, SET, ACC, 2, 2) RS_M2(S, SET, ACC, 3, 0) RS_M2(S, SET, ACC, 3, 2) \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
        RS_SB                                                                                           \
    }
        RS_RX(0, 0, 0, 0) RS_RX(0, 1, 0, 0) RS_RX(0, 2, 0, 0) RS_RX(0, 3, 0, 0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RS_SB
        // the downsample (ring step 1, the centre tap's pixels again) runs LAST: its accumulators then live from here to the
        // epilogue only, not across the nine taps (with them in the loop hipcc moved accumulator tiles around behind s_nop 6)
        RS_STEP(0, 0, acc, true, 1, 1, 0)        // tap (1,1)                         -> reads tap (1,0)
        RS_STEP(2, 1, acc, true, 0, 2, 0)        // tap (1,0)                         -> reads tap (1,2)
        RS_STEP(3, 0, acc, true, 1, 3, 0)        // tap (1,2)                         -> reads tap (0,1)
        RS_STEP(4, 1, acc, true, 0, 3, 1)        // tap (0,1)                         -> reads tap (2,1)
        RS_STEP(5, 0, acc, true, 1, 4, 0)        // tap (2,1)                         -> reads tap (0,0)
        RS_STEP(6, 1, acc, true, 0, 4, 1)        // tap (0,0)                         -> reads tap (2,0)
        RS_STEP(7, 0, acc, true, 1, 5, 0)        // tap (2,0)                         -> reads tap (0,2)
        RS_STEP(8, 1, acc, true, 0, 5, 1)        // tap (0,2)                         -> reads tap (2,2)
        RS_STEP(9, 0, acc, true, 1, 0, 0)        // tap (2,2)                         -> reads the centre tap's pixels again
        RS_STEP(1, 1, acc2, false, 0, 0, 0)      // downsample
        // round 5: the MFMAs are inline asm with the weight fragment read straight from the accumulation register file (hipcc copied
        // AGPR-resident fragments back in front of every MFMA: 578 v_accvgpr_read per 320 MFMAs); the wait between the last MFMA and
        // the first read of its result is spelled out
        mfma_drain8(acc2[0][0], acc2[0][1], acc2[0][2], acc2[0][3], acc2[1][0], acc2[1][1], acc2[1][2], acc2[1][3]);
        asm volatile("" : "+v"(acc2[2][0]), "+v"(acc2[2][1]), "+v"(acc2[2][2]), "+v"(acc2[2][3]), "+v"(acc2[3][0]), "+v"(acc2[3][1]), "+v"(acc2[3][2]), "+v"(acc2[3][3]));
        __builtin_amdgcn_sched_barrier(0);
#undef RS_STEP
#undef RS_M2
#undef RS_RX
#undef RS_SB

        // ---- epilogue: K-half exchange of both accumulator sets through this tile's own plane buffer (dead; the next tile
        // sits in the other one; with more than two tiles: through the buffer of the tile itself), then bias (+ ReLU) -> 16-bit, 16-byte stores straight from registers: wave (kw, pg) ends with
        // channel tiles 2 kw, 2 kw + 1 = channels n_blk + 32 kw + 8 fq .. + 7 of its 64 pixels.
        int img0, y0, x0;
        tile_origin(t, img0, y0, x0);
        size_t poff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            poff[j] = (size_t)a.out_off + (size_t)img0 * a.out_img + (size_t)(y0 + 4 * pg + j) * a.out_row + (size_t)(x0 + frow) * a.out_pix + cbase;
        {
            __builtin_amdgcn_s_barrier();                          // every wave is done with this tile's planes
            float* xch = reinterpret_cast<float*>(smem + (t & 1) * kPatch);      // [set 2][wave 4][8 tiles][64 lanes] f32x4 = 64 KB
            using I0 = std::integral_constant<int, 0>;
            using I2 = std::integral_constant<int, 2>;
            auto put = [&](auto GIVE) {
                constexpr int g = decltype(GIVE)::value;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        *reinterpret_cast<f32x4*>(xch + ((size_t)((0 * 4 + wave_id) * 8 + ii * 4 + j) * 64 + lane) * 4) = acc[g + ii][j];
                        *reinterpret_cast<f32x4*>(xch + ((size_t)((1 * 4 + wave_id) * 8 + ii * 4 + j) * 64 + lane) * 4) = acc2[g + ii][j];
                    }
            };
            if (kw == 0) put(I2{}); else put(I0{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int partner = wave_id ^ 2;
            // one accumulator set at a time (partner's halves in, bias, pack, store), pinned: all 32 partner tiles in flight at
            // once next to the 160 weight registers overflowed the 256 architectural VGPRs (24 spilled)
            auto finish = [&](auto KEEP, auto SETC) {
                constexpr int k = decltype(KEEP)::value;
                constexpr int set = decltype(SETC)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 lo = set == 0 ? acc[k][j] : acc2[k][j], hi = set == 0 ? acc[k + 1][j] : acc2[k + 1][j];
                    lo += *reinterpret_cast<const f32x4*>(xch + ((size_t)((set * 4 + partner) * 8 + 0 * 4 + j) * 64 + lane) * 4);
                    hi += *reinterpret_cast<const f32x4*>(xch + ((size_t)((set * 4 + partner) * 8 + 1 * 4 + j) * 64 + lane) * 4);
                    if constexpr (set == 0) {
                        float v[8] = {lo[0] + b0.x, lo[1] + b0.y, lo[2] + b0.z, lo[3] + b0.w, hi[0] + b1.x, hi[1] + b1.y, hi[2] + b1.z, hi[3] + b1.w};
                        if (a.relu) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        *reinterpret_cast<uint4*>(a.out + poff[j]) =
                            make_uint4(Elem<F16>::pack2(v[0], v[1]), Elem<F16>::pack2(v[2], v[3]), Elem<F16>::pack2(v[4], v[5]), Elem<F16>::pack2(v[6], v[7]));
                    } else {
                        *reinterpret_cast<uint4*>(a.out2 + poff[j]) =
                            make_uint4(Elem<F16>::pack2(lo[0] + c0.x, lo[1] + c0.y), Elem<F16>::pack2(lo[2] + c0.z, lo[3] + c0.w),
                                       Elem<F16>::pack2(hi[0] + c1.x, hi[1] + c1.y), Elem<F16>::pack2(hi[2] + c1.z, hi[3] + c1.w));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            using S0 = std::integral_constant<int, 0>;
            using S1 = std::integral_constant<int, 1>;
            if (kw == 0) { finish(I0{}, S0{}); finish(I0{}, S1{}); } else { finish(I2{}, S0{}); finish(I2{}, S1{}); }
        }
        // more than two tiles per workgroup (plan "throughput": the 80 KB of weights are fetched once per workgroup): the tile after
        // next goes into the buffer this tile has just left -- once every wave has read its partner's halves out of it
        if (t + 2 < ntiles) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue_tile(t + 2);
        }
    }
}

}  // namespace

namespace spsp {

// conv_s2_supported() AND one 64-channel chunk of input, 32 x 32 output maps (layer2.0), an even number of 128-pixel tiles
bool conv_s2rs_supported(const ConvArgs& a) {
    return conv_s2_supported(a) && a.chunks == 1 && a.Wo == 32 && a.N == 128 && (a.M / 128) % 2 == 0;
}

int launch_conv_s2rs(const ConvArgs& a, int num_cus, hipStream_t stream) {
    SPSP_REQUIRE(conv_s2rs_supported(a), "conv_s2rs: needs layer2.0's shape (64 -> 128 channels, 32 x 32 output maps, whole pairs of 128-pixel tiles)");
    const int mt = a.M / 128, units = mt * (a.N / 64);
    RsTileMap tm;
    // two tiles per workgroup once one tile each would need a second round of workgroups; `num_cus` below the chip's (plan
    // "throughput": a quarter of it) -> 4 or 8 tiles per resident weight set, as long as they divide the tiles of a block
    tm.tiles_per_wg = units > num_cus ? 2 : 1;
    while (tm.tiles_per_wg < 8 && units > tm.tiles_per_wg * num_cus && mt % (2 * tm.tiles_per_wg) == 0) tm.tiles_per_wg *= 2;
    const int grid = units / tm.tiles_per_wg;
    tm.paired = grid % 16 == 0 ? 1 : 0;
    SPSP_TRY((ensure_dynamic_lds<&conv3x3s2_kernel<true>>(kLds)));
    SPSP_TRY((ensure_dynamic_lds<&conv3x3s2_kernel<false>>(kLds)));
    if (a.f16) launch_kernel(&conv3x3s2_kernel<true>, dim3(grid), dim3(256), kLds, stream, a, tm);
    else launch_kernel(&conv3x3s2_kernel<false>, dim3(grid), dim3(256), kLds, stream, a, tm);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp
