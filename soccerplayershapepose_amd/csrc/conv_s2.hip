// 3x3 / STRIDE-2 convolution of a residual stage (PR/models/resnet.py:177-199: the first BasicBlock of layer2-4) with the
// block's 1x1 / stride-2 downsample fused in, on an LDS-RESIDENT PATCH SPLIT INTO THE FOUR PARITY PLANES of the input.
//
// Why: the implicit-GEMM ring kernel (conv_ring.hip) stages every output pixel's input once per filter tap -- 126-168 MB
// through the L2 -> LDS path per launch for 5.4 GFLOP, which is what its 17 us were (round 2: 0.12 of the MFMA peak).  A
// stride-2 3x3 is nine STRIDE-1 taps on the four phase-subsampled copies of the (zero-bordered) input:
//     padded input pixel (2 oy + ky, 2 ox + kx)  =  plane (ky & 1, kx & 1), plane pixel (oy + (ky >> 1), ox + (kx >> 1))
// so a workgroup loads the (tile + halo) pixels of one 64-channel chunk ONCE, plane by plane, and every tap reads 16
// CONSECUTIVE plane pixels per MFMA tile -- the conflict-free fragment read of conv_patch.hip (chunk index XOR q & 7).  The
// 1x1 downsample reads the centre tap's pixels (plane (1,1)): its K-step reuses that tap's pixel fragments with its own
// weight slab and a second accumulator set.  Staged bytes per launch: 78 MB (patch 72 KB + ten 8 KB weight slabs per chunk
// and workgroup) instead of 126-168.
//
// Workgroup = 128 output pixels x 64 output channels (of both convolutions), 4 waves = 2 pixel groups x 2 K-halves (wave
// pairs exchange accumulators through LDS at the end, as conv_patch.hip).  Weights = MFMA A operand (rows permuted so that a
// lane ends up with 8 consecutive channels: 16-byte stores), pixels = B operand.
//   So >= 16 : tile = 16 x 8 output pixels of one image; plane (py, px) holds (8 + !py) x (16 + !px) pixels.
//   So == 8  : tile = TWO whole 8 x 8 images INTERLEAVED along the columns (plane column c' = 2 col + image): an MFMA tile
//              is then one 16-"pixel" row, a one-column shift is a shift by 2 -- the same conflict-free reads as the wide
//              maps (the 8 x 8 form of conv_patch.hip, two 8-pixel rows per fragment, has 2-way conflicts).
//
// K order inside a 64-channel chunk -- by plane, so that the planes can be waited for (first chunk) and refilled (later
// chunks) one at a time:   G0 = plane (1,1): tap (1,1), downsample | G1 = plane (1,0): taps (1,0), (1,2) |
//                          G2 = plane (0,1): taps (0,1), (2,1)     | G3 = plane (0,0): taps (0,0), (2,0), (0,2), (2,2)
// = ten K-steps, each with its own 8 KB slot in the weight ring.  Everything of chunk 0 is requested in consumption order in
// the prologue and waited for with counted vmcnt (LDS-DMA completes in order): the MFMAs start when plane (1,1) and two slabs
// are there, the rest lands underneath.  ONE barrier per group, placed before the group's LAST step: by then every wave has
// all fragments of the group in registers (reads run one step ahead), so the barrier releases the group's plane and slots --
// refilled at once with the next chunk's data, a whole chunk (2.5 k MFMA cycles) before it is needed -- and, because every
// wave waits for its own pieces of the NEXT group before it, it also publishes the next group: its first fragment reads go
// out beside the last step's MFMAs, no bubble behind the barrier.
#include <type_traits>

#include "conv.hpp"
#include "conv_elem.hpp"

namespace {
using namespace spsp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

constexpr int kSlab = 64 * 128;              // [64 n][64 k] 16-bit
constexpr int kSteps = 10;                   // K-steps per 64-channel chunk: nine taps + the downsample
constexpr int kPixB = 128;                   // bytes per patch pixel (64 channels)

// output channel (within the 64-channel block) held by accumulator tile i, MFMA row r -- as conv_patch.hip
__host__ __device__ constexpr int acc_channel(int i, int r) { return 32 * (i >> 1) + 8 * (r >> 2) + 4 * (i & 1) + (r & 3); }

// planes in consumption order: 0 = (1,1), 1 = (1,0), 2 = (0,1), 3 = (0,0)
template <bool W8>
struct S2Geo {
    static constexpr int py(int p) { return p < 2 ? 1 : 0; }
    static constexpr int px(int p) { return (p == 0 || p == 2) ? 1 : 0; }
    static constexpr int rows(int p) { return 8 + (py(p) == 0 ? 1 : 0); }
    static constexpr int cw(int p) { return W8 ? 2 * (8 + (px(p) == 0 ? 1 : 0)) : 16 + (px(p) == 0 ? 1 : 0); }   // plane row pitch in pixels
    static constexpr int npix(int p) { return rows(p) * cw(p); }
    static constexpr int npiece(int p) { return (npix(p) + 7) / 8; }         // LDS-DMA pieces: 8 pixels per wave-instruction
    static constexpr int dpw(int p) { return (npiece(p) + 3) / 4; }          // pieces per wave (4 waves)
    static constexpr int base_piece(int p) { return p == 0 ? 0 : base_piece(p - 1) + npiece(p - 1); }
    static constexpr int kPatch = (base_piece(3) + npiece(3)) * 1024;
    static constexpr int DXS = W8 ? 2 : 1;                                   // plane-column shift of a one-pixel tap shift
    // vmcnt units a wave issues per group: its plane pieces + 2 per weight slab
    static constexpr int nG(int g) { return dpw(g) + 2 * (g == 3 ? 4 : 2); }
};

template <bool W8, bool F16>
__global__ __launch_bounds__(256) void conv3x3s2_kernel(const ConvArgs a) {
    using G = S2Geo<W8>;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const ring = smem;                          // ten 8 KB weight slots
    unsigned char* const patch = smem + kSteps * kSlab;        // the four planes

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave_id & 1;                                // which 64 pixels of the tile
    const int kw = wave_id >> 1;                               // which 32-deep half of every K-step
    const int mt = a.M / 128, nb = a.N / 64;
    int m_tile, n_blk;
    {   // XCD-aware tile order, as conv_patch.hip: XCD x = b & 7 owns a rectangle of the (pixel tiles) x (channel blocks) grid
        const int XN = a.xcd_n, XM = XN > 0 ? 8 / XN : 0;
        const int bid = blockIdx.x;
        if (XN > 0 && mt % XM == 0 && nb % XN == 0) {
            const int x = bid & 7, li = bid >> 3, nbl = nb / XN;
            m_tile = (x / XN) * (mt / XM) + li / nbl;
            n_blk = ((x % XN) * nbl + li % nbl) * 64;
        } else {
            m_tile = bid / nb;
            n_blk = (bid % nb) * 64;
        }
    }
    int img0, y0 = 0, x0 = 0;
    if constexpr (W8) { img0 = 2 * m_tile; }
    else {
        const int So = a.Wo, tpr = So / 16, tpi = tpr * (So / 8);
        img0 = m_tile / tpi;
        const int t = m_tile - img0 * tpi;
        y0 = (t / tpr) * 8; x0 = (t % tpr) * 16;
    }
    const int nchunks = a.chunks;
    const int Ktot = a.nsteps * 64, Ktot2 = nchunks * 64;
    const uint16_t* const in_ptr = a.in;

    // ---- patch LDS-DMA sources: lane l of piece p of a plane feeds plane pixel q = 8 p + l / 8, physical chunk l % 8 ----
    const int drow = lane >> 3, dchunk = lane & 7;
    int p_src[4][6];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            p_src[p][i] = 0;
            if (i < G::dpw(p)) {
                int piece = wave_id + 4 * i;
                piece = piece < G::npiece(p) ? piece : G::npiece(p) - 1;
                int q = piece * 8 + drow;
                q = q < G::npix(p) ? q : G::npix(p) - 1;          // tail lanes re-read the last pixel (harmless)
                const int r = q / G::cw(p), c = q - r * G::cw(p);
                int off;
                if constexpr (W8) off = (img0 + (c & 1)) * a.in_img + (2 * r + G::py(p)) * a.in_row + (2 * (c >> 1) + G::px(p)) * a.in_pix;
                else off = img0 * a.in_img + (2 * (y0 + r) + G::py(p)) * a.in_row + (2 * (x0 + c) + G::px(p)) * a.in_pix;
                p_src[p][i] = off + ((dchunk ^ (q & 7)) * 8);
            }
        }
    auto issue_plane = [&](auto pc, int ch) {
        constexpr int p = decltype(pc)::value;
#pragma unroll
        for (int i = 0; i < G::dpw(p); ++i) {
            int piece = wave_id + 4 * i;
            piece = piece < G::npiece(p) ? piece : G::npiece(p) - 1;   // every wave issues exactly dpw pieces (counted waits)
            dma16(in_ptr + p_src[p][i] + ch * 64, patch + (G::base_piece(p) + piece) * 1024);
        }
    };
    // ---- weight slab sources: LDS row R of a slab holds output channel acc_channel(R / 16, R % 16); a wave owns rows 16 w .. + 15
    const uint16_t *w_src[2], *w2_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave_id * 2 + i) * 8 + drow;
        const int chn = n_blk + acc_channel(row >> 4, row & 15);
        w_src[i] = a.w + (size_t)chn * Ktot + ((dchunk ^ (row & 7)) * 8);
        w2_src[i] = a.w2 + (size_t)chn * Ktot2 + ((dchunk ^ (row & 7)) * 8);
    }
    // step s of a chunk: filter tap ky * 3 + kx (the weight matrix is stored tap-major, chunk-minor), -1 = the downsample
    auto issue_slab = [&](auto sc, int ch) {
        constexpr int s = decltype(sc)::value;
        constexpr int taps[kSteps] = {4, -1, 3, 5, 1, 7, 0, 6, 2, 8};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint16_t* src = taps[s] < 0 ? w2_src[i] + (size_t)ch * 64 : w_src[i] + (size_t)(taps[s] * nchunks + ch) * 64;
            dma16(src, ring + s * kSlab + (wave_id * 2 + i) * 1024);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>; using I8 = std::integral_constant<int, 8>;
    using I9 = std::integral_constant<int, 9>;
    auto issue_group = [&](auto gc, int ch) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g == 0) { issue_slab(I0{}, ch); issue_slab(I1{}, ch); issue_plane(I0{}, ch); }
        if constexpr (g == 1) { issue_slab(I2{}, ch); issue_slab(I3{}, ch); issue_plane(I1{}, ch); }
        if constexpr (g == 2) { issue_slab(I4{}, ch); issue_slab(I5{}, ch); issue_plane(I2{}, ch); }
        if constexpr (g == 3) { issue_slab(I6{}, ch); issue_slab(I7{}, ch); issue_slab(I8{}, ch); issue_slab(I9{}, ch); issue_plane(I3{}, ch); }
    };

    f32x4 acc[4][4], acc2[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    // weight fragment reads: slab row i * 16 + frow, 16-byte chunk (kw * 4 + fq) ^ (frow & 7)
    const uint32_t w_lane_k = smem_base + frow * 128 + ((((kw * 4 + fq) ^ (frow & 7))) * 16);
    // pixel fragment addresses: combo = (plane, dx), row = 4 pg + (j + dy); MFMA tile j of tap (dy, dx) reads plane pixels
    // q = (4 pg + j + dy) * cw + frow + dx * DXS .. (16 consecutive), chunk (kw * 4 + fq) ^ (q & 7)
    //   combo 0: plane (1,1)   1: plane (1,0) dx 0   2: plane (1,0) dx 1   3: plane (0,1)   4: plane (0,0) dx 0   5: plane (0,0) dx 1
    uint32_t xaddr[6][5];
    {
        constexpr int cplane[6] = {0, 1, 1, 2, 3, 3}, cdx[6] = {0, 0, 1, 0, 0, 1};
#pragma unroll
        for (int cb = 0; cb < 6; ++cb)
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int p = cplane[cb];
                int row = 4 * pg + r;
                row = row < G::rows(p) ? row : G::rows(p) - 1;       // (row 4 of an 8-row plane is never read)
                const int q = row * G::cw(p) + frow + cdx[cb] * G::DXS;
                xaddr[cb][r] = smem_base + kSteps * kSlab + G::base_piece(p) * 1024 + q * kPixB + ((((kw * 4 + fq) ^ (q & 7))) * 16);
            }
    }

    bf16x8 wfr[2][4], xfr[2][4];
#define S2_SB __builtin_amdgcn_sched_barrier(0);
#define S2_RW(SET, I, SLOT) wfr[SET][I] = lds_read_b128<(I) * 2048>(w_lane_k + (SLOT) * kSlab);
#define S2_RX(SET, J, CB, DY) xfr[SET][J] = lds_read_b128<0>(xaddr[CB][(J) + (DY)]);
#define S2_M2(SET, ACC, I, J)                                                                           \
        S2_SB                                                                                           \
        ACC[I][J] = Elem<F16>::mfma(wfr[SET][I], xfr[SET][J], ACC[I][J]);                               \
        ACC[I][(J) + 1] = Elem<F16>::mfma(wfr[SET][I], xfr[SET][(J) + 1], ACC[I][(J) + 1]);             \
        S2_SB
    // One K-step: its 16 MFMAs (fragments in set SET, into ACC) with the 8 fragment reads of the NEXT step (ring slot NS, pixel
    // combo NCB, row shift NDY; skipped unless HASN) beside the first eight, two per MFMA pair, pinned by sched_barriers (hipcc
    // otherwise gathers the reads in front and sinks the register-only MFMAs below the wait).
#define S2_STEP(SET, ACC, HASN, NS, NCB, NDY)                                                           \
    {                                                                                                   \
        if (HASN) { S2_RW((SET) ^ 1, 0, NS) S2_RW((SET) ^ 1, 1, NS) }                                   \
        S2_M2(SET, ACC, 0, 0)                                                                           \
        if (HASN) { S2_RW((SET) ^ 1, 2, NS) S2_RW((SET) ^ 1, 3, NS) }                                   \
        S2_M2(SET, ACC, 0, 2)                                                                           \
        if (HASN) { S2_RX((SET) ^ 1, 0, NCB, NDY) S2_RX((SET) ^ 1, 1, NCB, NDY) }                       \
        S2_M2(SET, ACC, 1, 0)                                                                           \
        if (HASN) { S2_RX((SET) ^ 1, 2, NCB, NDY) S2_RX((SET) ^ 1, 3, NCB, NDY) }                       \
        S2_M2(SET, ACC, 1, 2)                                                                           \
        S2_M2(SET, ACC, 2, 0) S2_M2(SET, ACC, 2, 2) S2_M2(SET, ACC, 3, 0) S2_M2(SET, ACC, 3, 2)         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
        S2_SB                                                                                           \
    }
    // The barrier of group GG, in front of its last step.  WAITM / WAITL = what this wave may still have in flight when the
    // pieces of the group that is read next have landed (middle chunk / last chunk).
#define S2_TOP(GC, WAITM, WAITL, NEED_LAST)                                                             \
    {                                                                                                   \
        if (!last) wait_vmcnt<WAITM>(); else if (NEED_LAST) wait_vmcnt<WAITL>();                        \
        if (!last || (NEED_LAST)) __builtin_amdgcn_s_barrier();                                         \
        if (!last) issue_group(GC{}, ch + 1);                                                           \
    }

    // ---- prologue: chunk 0, everything, in consumption order
    issue_group(I0{}, 0); issue_group(I1{}, 0); issue_group(I2{}, 0); issue_group(I3{}, 0);
    wait_vmcnt<G::nG(1) + G::nG(2) + G::nG(3)>();
    __builtin_amdgcn_s_barrier();
    S2_RW(0, 0, 0) S2_RW(0, 1, 0) S2_RW(0, 2, 0) S2_RW(0, 3, 0)
    S2_RX(0, 0, 0, 0) S2_RX(0, 1, 0, 0) S2_RX(0, 2, 0, 0) S2_RX(0, 3, 0, 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    S2_SB

    for (int ch = 0; ch < nchunks; ++ch) {
        const bool last = ch + 1 == nchunks;
        S2_STEP(0, acc, true, 1, 0, 0)                                     // s0: tap (1,1)         -> reads s1 (downsample: same pixels)
        S2_TOP(I0, G::nG(2) + G::nG(3), G::nG(2) + G::nG(3), true)          //     needs G1 (chunk ch)
        S2_STEP(1, acc2, true, 2, 1, 0)                                    // s1: downsample        -> reads s2 = tap (1,0)
        S2_STEP(0, acc, true, 3, 2, 0)                                     // s2: tap (1,0)         -> reads s3 = tap (1,2)
        S2_TOP(I1, G::nG(3) + G::nG(0), G::nG(3), true)                    //     needs G2
        S2_STEP(1, acc, true, 4, 3, 0)                                     // s3: tap (1,2)         -> reads s4 = tap (0,1)
        S2_STEP(0, acc, true, 5, 3, 1)                                     // s4: tap (0,1)         -> reads s5 = tap (2,1)
        S2_TOP(I2, G::nG(0) + G::nG(1), 0, true)                           //     needs G3
        S2_STEP(1, acc, true, 6, 4, 0)                                     // s5: tap (2,1)         -> reads s6 = tap (0,0)
        S2_STEP(0, acc, true, 7, 4, 1)                                     // s6: tap (0,0)         -> reads s7 = tap (2,0)
        S2_STEP(1, acc, true, 8, 5, 0)                                     // s7: tap (2,0)         -> reads s8 = tap (0,2)
        S2_STEP(0, acc, true, 9, 5, 1)                                     // s8: tap (0,2)         -> reads s9 = tap (2,2)
        S2_TOP(I3, G::nG(1) + G::nG(2), 0, false)                          //     needs G0 of the next chunk
        S2_STEP(1, acc, !last, 0, 0, 0)                                    // s9: tap (2,2)         -> reads s0 of the next chunk
    }
#undef S2_TOP
#undef S2_STEP
#undef S2_M2
#undef S2_RX
#undef S2_RW
#undef S2_SB

    // ---- epilogue: K-half exchange of both accumulator sets through LDS (ring + patch are dead), then bias (+ ReLU) -> 16-bit,
    // 16-byte stores straight from registers: wave (kw, pg) ends with channel tiles 2 kw, 2 kw + 1 = channels
    // n_blk + 32 kw + 8 fq .. + 7 of its 64 pixels.
    const int cbase = n_blk + 32 * kw + 8 * fq;
    size_t poff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (W8) poff[j] = (size_t)a.out_off + (size_t)(img0 + (frow & 1)) * a.out_img + (size_t)(4 * pg + j) * a.out_row + (size_t)(frow >> 1) * a.out_pix + cbase;
        else poff[j] = (size_t)a.out_off + (size_t)img0 * a.out_img + (size_t)(y0 + 4 * pg + j) * a.out_row + (size_t)(x0 + frow) * a.out_pix + cbase;
    }
    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + cbase), b1 = *reinterpret_cast<const float4*>(a.bias + cbase + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(a.bias2 + cbase), c1 = *reinterpret_cast<const float4*>(a.bias2 + cbase + 4);
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* xch = reinterpret_cast<float*>(smem);               // [set 2][wave 4][8 tiles][64 lanes] f32x4 = 64 KB
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
        auto get = [&](auto KEEP) {
            constexpr int k = decltype(KEEP)::value;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[k + ii][j] += *reinterpret_cast<const f32x4*>(xch + ((size_t)((0 * 4 + partner) * 8 + ii * 4 + j) * 64 + lane) * 4);
                    acc2[k + ii][j] += *reinterpret_cast<const f32x4*>(xch + ((size_t)((1 * 4 + partner) * 8 + ii * 4 + j) * 64 + lane) * 4);
                }
        };
        if (kw == 0) get(I0{}); else get(I2{});
    }
    auto store = [&](auto I0c) {
        constexpr int T0 = decltype(I0c)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 lo = acc[T0][j], hi = acc[T0 + 1][j];
            float v[8] = {lo[0] + b0.x, lo[1] + b0.y, lo[2] + b0.z, lo[3] + b0.w, hi[0] + b1.x, hi[1] + b1.y, hi[2] + b1.z, hi[3] + b1.w};
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            *reinterpret_cast<uint4*>(a.out + poff[j]) =
                make_uint4(Elem<F16>::pack2(v[0], v[1]), Elem<F16>::pack2(v[2], v[3]), Elem<F16>::pack2(v[4], v[5]), Elem<F16>::pack2(v[6], v[7]));
            const f32x4 lo2 = acc2[T0][j], hi2 = acc2[T0 + 1][j];
            *reinterpret_cast<uint4*>(a.out2 + poff[j]) =
                make_uint4(Elem<F16>::pack2(lo2[0] + c0.x, lo2[1] + c0.y), Elem<F16>::pack2(lo2[2] + c0.z, lo2[3] + c0.w),
                           Elem<F16>::pack2(hi2[0] + c1.x, hi2[1] + c1.y), Elem<F16>::pack2(hi2[2] + c1.z, hi2[3] + c1.w));
        }
    };
    if (kw == 0) store(I0{}); else store(I2{});
}

template <bool W8, bool F16>
int launch_t(const ConvArgs& a, hipStream_t st) {
    using G = S2Geo<W8>;
    constexpr size_t lds = (size_t)kSteps * kSlab + G::kPatch;
    static_assert(lds <= 160 * 1024, "one CU has 160 KB of LDS");
    static_assert(kSteps * kSlab >= 2 * 4 * 8 * 1024, "K-half exchange of two accumulator sets");
    SPSP_TRY((ensure_dynamic_lds<&conv3x3s2_kernel<W8, F16>>((int)lds)));
    const int grid = (a.M / 128) * (a.N / 64);
    launch_kernel(&conv3x3s2_kernel<W8, F16>, dim3(grid), dim3(256), lds, st, a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace

namespace spsp {

// a 3x3 stride-2 pad-1 convolution with its 1x1 stride-2 companion on zero-bordered maps, in whole 128-pixel tiles
bool conv_s2_supported(const ConvArgs& a) {
    return a.stride == 2 && a.kw_steps == 3 && a.nsteps == 9 * a.chunks && a.in_off == 0 && a.n2 == a.N && a.w2 && a.bias2 && a.out2 &&
           a.out && !a.res && !a.pool_out && a.N % 64 == 0 && a.Ho == a.Wo && (a.Wo == 8 || a.Wo == 16 || a.Wo == 32) && a.M % 128 == 0 &&
           a.in_pix == a.chunks * 64;
}

int launch_conv_s2(const ConvArgs& a0, hipStream_t stream) {
    SPSP_REQUIRE(conv_s2_supported(a0), "conv_s2: needs a 3x3 stride-2 convolution with its 1x1 downsample, output maps of 8 / 16 / 32 pixels, M %% 128 == 0");
    ConvArgs a = a0;
    a.xcd_n = conv_patch_xcd_n(a, 128);
    if (a.Wo == 8) return a.f16 ? launch_t<true, true>(a, stream) : launch_t<true, false>(a, stream);
    return a.f16 ? launch_t<false, true>(a, stream) : launch_t<false, false>(a, stream);
}

}  // namespace spsp
