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
// per fragment read instead of 4; model: tools/lds_bank_check.py).  The 8 x 8-map tiles (two 8-pixel rows per
// fragment) keep (q >> 1) & 7: no linear swizzle is conflict-free for that shape.
// The image is written by LDS-DMA (lane-linear: 8 pixels x 8 chunks per wave-instruction, swizzle applied on the
// SOURCE chunk) and read with ds_read_b128; the 36 per-lane fragment offsets (9 taps x 4 tiles) live in registers.
//
// Tile geometries (W = feature-map width = height):
//   W >= 16 : tile = 16 x 16 pixels of one image, patch 18 x 18; wave w owns rows 4w..4w+3, MFMA tile j = one row.
//   W == 8  : tile = 4 whole 8 x 8 images, patch 4 x (10 x 10); wave w owns image w, MFMA tile j = rows 2j, 2j+1.
//
// Variants (template <W8, KS, NS>; regressor.hip picks per stage, SPSP_PATCH_VARIANT overrides):
//   <., 1, 3>  4 waves, per-tap ring of 3 slabs, one patch buffer: 65-74 KB, two workgroups per CU.  The first form;
//              at 32 crops a layer has no more tiles than the chip has CUs, so the partner workgroup rarely exists.
//   <., 2, 5>  8 waves: wave pairs (w, w+4 -- the two waves of one SIMD) split every K-step in halves and exchange
//              accumulators through LDS at the end; per-tap ring of 5 slabs.
//   <., 2, 6>  (default) as above with a GROUPED ring -- the three slabs of a kernel row land together in one half
//              of a 2 x 3-slab ring, one vmcnt(0) + barrier per three taps; with DB (template flag, the default for
//              layer3 / layer4) TWO patch buffers, so the next 64-channel chunk's patch lands under the nine taps of
//              the current one (131 / 150 KB of LDS instead of 89 / 99 KB).
// Fragment reads are software-pipelined one tap ahead of the MFMAs in every variant (two register sets).
// What the counters say about <false, 2, 6> alone on the chip (rocprofv3 --pmc, layer2 shape): the matrix pipe is busy
// 25 % of the kernel; a wave is parked in s_waitcnt / s_barrier 46 % of its cycles, stalled at issue (matrix pipe taken
// by its SIMD partner, MFMA dependencies) 33 %, issuing 21 %; LDS bank conflicts < 5 % of LDS cycles.
#include <cstdlib>
#include <type_traits>

#include "conv.hpp"
#include "conv_elem.hpp"

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

constexpr int kWStage = 64 * 128;            // [64 n][64 k] bf16

#ifdef SPSP_CONV_STAMPS
#define CSTAMP(i) if (a.dbg && blockIdx.x == 8 && tid == 0) a.dbg[i] = __builtin_amdgcn_s_memtime()
#else
#define CSTAMP(i)
#endif
constexpr int kPixB = 128;                   // bytes per patch pixel (64 ch)

template <bool W8>
__device__ __forceinline__ int patch_swz(int q) { return W8 ? ((q >> 1) & 7) : (q & 7); }

template <bool W8, int KS, int NS>
struct Geo {
    static constexpr int NW = 4 * KS;                             // waves per workgroup
    static constexpr int PW = W8 ? 10 : 18;                       // patch row pitch in pixels
    static constexpr int NPIX = W8 ? 400 : 324;                   // patch pixels
    static constexpr int NDMA = (NPIX + 7) / 8;                   // 8 pixels per wave-instruction
    static constexpr int PATCH = NDMA * 8 * kPixB;                // bytes (padded to whole DMA pieces)
    static constexpr int DPW = (NDMA + NW - 1) / NW;              // patch DMA instructions per wave
};

// One workgroup: 256 pixels x 64 channels.  grid = (M/256) * (N/64) (XCD-contiguous order).
template <bool W8, int KS, int NS, bool F16, bool DB>
__global__ __launch_bounds__(256 * KS) void conv3x3_patch_kernel(const ConvArgs a) {
    using G = Geo<W8, KS, NS>;
    constexpr int PW = G::PW, kNS = NS, NW = G::NW;
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const ring = smem;                              // kNS x 8 KB weight stages
    unsigned char* const patch0 = smem + kNS * kWStage;            // the patch

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_id & 3;                                  // which 64 pixels of the tile
    const int kw = wave_id >> 2;                                   // KS == 2: which 32-deep half of every K-step
    const int W = a.Wo;                                            // == Ho, == input W (stride 1)
    const int mt = a.M / 256;
    const int ntiles = mt * (a.N / 64);
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
    const int m_tile = tile % mt, n_blk = (tile / mt) * 64;

    // tile origin in the padded input (element offset of patch pixel (0,0), channel 0)
    int img0, y0 = 0, x0 = 0;
    if constexpr (W8) { img0 = m_tile * 4; }
    else {
        const int tpr = W / 16, tpi = tpr * tpr;                   // tiles per row / per image
        img0 = m_tile / tpi;
        const int t = m_tile - img0 * tpi;
        y0 = (t / tpr) * 16; x0 = (t % tpr) * 16;
    }
    // split-K over workgroups (deep, tile-starved layers): blockIdx.y takes chunks [ch0, ch0 + nchunks) of a.chunks and
    // leaves its fp32 partial tile in slab blockIdx.y of a.part; splitk_epilogue_kernel (conv_ring.hip) finishes.
    const int nch_tot = a.chunks;                                  // Cin / 64
    const int nchunks = a.splitk > 1 ? nch_tot / a.splitk : nch_tot;
    const int ch0 = a.splitk > 1 ? (int)blockIdx.y * nchunks : 0;
    const uint16_t* const in_ptr = a.in;                           // scalar copies (keep the kernarg struct out of lambdas)
    const uint16_t* const w_ptr = a.w;
    const int Ktot = a.nsteps * 64;                                // 9 * Cin

    // ---- patch LDS-DMA sources: lane l of piece p feeds patch pixel q = 8p + l/8, physical chunk l%8 ----
    const int drow = lane >> 3, dchunk = lane & 7;
    int p_src[G::DPW];                                             // element offsets into a.in (chunk 0)
#pragma unroll
    for (int i = 0; i < G::DPW; ++i) {
        const int p = wave_id + NW * i;
        int q = p * 8 + drow;
        q = q < G::NPIX ? q : G::NPIX - 1;                         // tail lanes re-read the last pixel (harmless)
        int off;
        if constexpr (W8) {
            const int im = q / 100, r = q - im * 100, py = r / 10, px = r - py * 10;
            off = (img0 + im) * a.in_img + py * a.in_row + px * a.in_pix;
        } else {
            const int py = q / 18, px = q - py * 18;
            off = img0 * a.in_img + (y0 + py) * a.in_row + (x0 + px) * a.in_pix;
        }
        p_src[i] = off + ((dchunk ^ patch_swz<W8>(q)) * 8);
    }
    auto issue_patch = [&](int ch, int pbuf = 0) {
#pragma unroll
        for (int i = 0; i < G::DPW; ++i) {
            const int p = wave_id + NW * i;
            if (p < G::NDMA) dma16(in_ptr + p_src[i] + (ch0 + ch) * 64, patch0 + pbuf * G::PATCH + p * 1024);
        }
    };
    // ---- weight ring sources: stage = [64 n][64 k] of K-step g (tap-major, chunk-minor: g = tap*nchunks + ch) ----
    constexpr int LPS = 2 / KS;                                    // weight pieces per wave per stage (8 per stage)
    const uint16_t* w_src[LPS];
#pragma unroll
    for (int i = 0; i < LPS; ++i) {
        const int row = (wave_id * LPS + i) * 8 + drow;
        w_src[i] = w_ptr + (size_t)(n_blk + row) * Ktot + ((dchunk ^ (row & 7)) * 8);
    }
    auto issue_w = [&](int g, int buf) {
#pragma unroll
        for (int i = 0; i < LPS; ++i) dma16(w_src[i] + (size_t)g * 64, ring + buf * kWStage + (wave_id * LPS + i) * 1024);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    // weight fragment reads: row t*16 + frow, chunk (ks*4 + fq) ^ (frow & 7)
    const uint32_t w_lane = frow * 128 + ((fq ^ (frow & 7)) * 16);
    // patch pixel index of this lane for MFMA tile j = qb + j * QJ (tap adds r*PW + s)
    int qb;
    if constexpr (W8) qb = wave * 100 + (frow >> 3) * 10 + (frow & 7);
    else qb = (wave * 4) * 18 + frow;
    constexpr int QJ = W8 ? 20 : 18;

    // LDS byte offsets of this lane's pixel fragments for every (tap, tile j), ks = 0 (ks = 1: chunk bit 2 flips).
    // They do not depend on the chunk, so the 36 values live in registers for the whole kernel.
    uint32_t xoff[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = qb + j * QJ + (tap / 3) * PW + (tap % 3);
            xoff[tap][j] = smem_base + kNS * kWStage + q * kPixB + ((((KS == 2 ? kw * 4 : 0) + fq) ^ patch_swz<W8>(q)) * 16);
        }

    // K order inside the kernel: chunk-major, tap-minor (patch of chunk ch serves 9 consecutive steps); the weight
    // matrix is stored tap-major/chunk-minor like the implicit-GEMM kernels, so step (ch, tap) reads K-step
    // g = tap * nchunks + ch.
    const int nsteps = 9 * nchunks;
    constexpr int NK = 2 / KS;                          // 32-deep k-substeps this wave does per step

    // ---- software pipeline: the fragment reads of step s are issued BEFORE the MFMAs of step s-1, so the LDS
    // latency (measured: ~380 cycles per step, as much as the MFMAs themselves) hides under the matrix pipe.  Two
    // register sets alternate; a step's set is static because the tap loop is fully unrolled and 9 is odd
    // (set = (9 ch + tap) & 1).  Per step s:
    //   lgkmcnt(0)            fragments of step s-1 have arrived (which also frees the ring slot / patch they came from)
    //   vmcnt(..) + barrier   weight stage s has landed, for every wave
    //   issue LDS-DMA         weight stage s + NS-1 into the slot of stage s-1
    //   ds_read x (8 NK)      fragments of step s     -> set s & 1
    //   MFMA x (16 NK)        step s-1                <- the other set
    // (macros, not lambdas: the fragment arrays must be indexed by literal constants to stay in registers)
    bf16x8 wfr[2][NK][4], xfr[2][NK][4];
#define SPSP_READ_FRAGS(TAP, SET, SLOT)                                                                   \
    {                                                                                                     \
        const uint32_t wbase_ = smem_base + (SLOT) * kWStage;                                             \
        _Pragma("unroll") for (int kk = 0; kk < NK; ++kk) {                                               \
            const int ks_ = KS == 2 ? kw : kk;                                                            \
            const uint32_t wa_ = wbase_ + (w_lane ^ (ks_ * 64));                                          \
            wfr[SET][kk][0] = lds_read_b128<0>(wa_);                                                      \
            wfr[SET][kk][1] = lds_read_b128<2048>(wa_);                                                   \
            wfr[SET][kk][2] = lds_read_b128<4096>(wa_);                                                   \
            wfr[SET][kk][3] = lds_read_b128<6144>(wa_);                                                   \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
                xfr[SET][kk][j] = lds_read_b128<0>((KS == 2 ? xoff[TAP][j] : (xoff[TAP][j] ^ (kk * 64))) + pboff_); \
        }                                                                                                 \
    }
#define SPSP_MFMA_SET(SET)                                                                                \
    {                                                                                                     \
        _Pragma("unroll") for (int kk = 0; kk < NK; ++kk)                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
                    acc[i][j] = Elem<F16>::mfma(wfr[SET][kk][i], xfr[SET][kk][j], acc[i][j]); \
    }
    // wait until weight stage s has landed: `ahead` younger stages (LPS pieces each) may stay in flight
    auto wait_stage = [&](int ahead) {
        if (ahead >= 3 && kNS >= 5) wait_vmcnt<3 * LPS>();
        else if (ahead == 2 && kNS >= 4) wait_vmcnt<2 * LPS>();
        else if (ahead >= 1) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
    };

    CSTAMP(0);
    issue_patch(0);
    if constexpr (kNS == 6) {
        // ---- grouped ring: the three slabs of one kernel row land together, ONE barrier per three K-steps ----------
        // Ring = 2 halves x 3 slabs.  Group g = (chunk, kernel row): at its start every wave has the fragments of the
        // previous group in registers (lgkmcnt(0)) and its own DMA pieces of group g landed (vmcnt(0): nothing younger
        // is in flight yet); after the barrier the slabs of group g+1 are issued into the half group g-1 was read
        // from, then the three taps run back to back -- 48 MFMAs per wave between barriers instead of 16.
        const int ngroups = 3 * nchunks;
        auto issue_group = [&](int g) {
            const int gch = g / 3, gr = g - gch * 3;
#pragma unroll
            for (int t = 0; t < 3; ++t) issue_w((gr * 3 + t) * nch_tot + ch0 + gch, (g & 1) * 3 + t);
        };
        issue_group(0);
#define SPSP_GTAP(TAP, SET, SLOT, FIRST)                                                                  \
    {                                                                                                     \
        SPSP_READ_FRAGS(TAP, SET, SLOT)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if (!(FIRST)) SPSP_MFMA_SET((SET) ^ 1)                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    }
#define SPSP_GROUP(R, P0, CH, HALF)                                                                       \
    {                                                                                                     \
        const int g_ = (CH) * 3 + (R);                                                                    \
        const uint32_t pboff_ = DB ? ((CH) & 1) * G::PATCH : 0;   /* this chunk's patch buffer */          \
        wait_vmcnt<0>();                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                     \
        if (!DB && (R) == 0 && (CH) > 0) {                                                                \
            /* one patch buffer, chunk boundary: every wave is past its last read of the old patch; refill it, run */ \
            /* the pending MFMAs of the previous tap underneath, then wait for the patch (older than the slabs) */ \
            issue_patch(CH);                                                                              \
            if (g_ + 1 < ngroups) issue_group(g_ + 1);                                                    \
            SPSP_MFMA_SET(((P0) & 1) ^ 1)                                                                 \
            if (g_ + 1 < ngroups) wait_vmcnt<3 * LPS>(); else wait_vmcnt<0>();                            \
            __builtin_amdgcn_s_barrier();                                                                 \
            SPSP_GTAP(0, (P0) & 1, (HALF) * 3 + 0, true)                                                  \
        } else {                                                                                          \
            /* two patch buffers, first group of a chunk: every wave is past its reads of chunk CH-1, whose buffer */ \
            /* now takes chunk CH+1 -- it has the nine taps of this chunk to land (no wait at chunk boundaries)    */ \
            if (DB && (R) == 0 && (CH) + 1 < nchunks) issue_patch((CH) + 1, ((CH) + 1) & 1);              \
            if (g_ + 1 < ngroups) issue_group(g_ + 1);                                                    \
            SPSP_GTAP((R) * 3 + 0, ((R) * 3 + 0 + (P0)) & 1, (HALF) * 3 + 0, (R) == 0 && (CH) == 0)       \
        }                                                                                                 \
        SPSP_GTAP((R) * 3 + 1, ((R) * 3 + 1 + (P0)) & 1, (HALF) * 3 + 1, false)                           \
        SPSP_GTAP((R) * 3 + 2, ((R) * 3 + 2 + (P0)) & 1, (HALF) * 3 + 2, false)                           \
    }
        // group parity: g = 3 ch + r -> half = g & 1; a chunk has 3 groups, so the half of row r flips with ch
        for (int ch = 0; ch < nchunks; ch += 2) {
            SPSP_GROUP(0, 0, ch, 0) SPSP_GROUP(1, 0, ch, 1) SPSP_GROUP(2, 0, ch, 0)
            if (ch + 1 < nchunks) { SPSP_GROUP(0, 1, ch + 1, 1) SPSP_GROUP(1, 1, ch + 1, 0) SPSP_GROUP(2, 1, ch + 1, 1) }
        }
#undef SPSP_GROUP
#undef SPSP_GTAP
    } else {
    constexpr uint32_t pboff_ = 0;
    int issued = 0;
    for (; issued < kNS - 1 && issued < nsteps; ++issued) issue_w((issued % 9) * nch_tot + ch0 + issued / 9, issued);
    auto issue_next = [&]() {
        if (issued < nsteps) {
            const int ich = issued / 9, itap = issued - ich * 9;
            issue_w(itap * nch_tot + ch0 + ich, issued % kNS);
            ++issued;
        }
    };

#define SPSP_STEP(TAP, SET, CH, S)                                                                        \
    {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        wait_stage(min(issued - ((S) + 1), kNS - 2));                                                     \
        __builtin_amdgcn_s_barrier();                                                                     \
        if ((TAP) == 0 && (CH) > 0) {                                                                     \
            /* chunk boundary: every wave is past its last read of the old patch.  Refill it; the MFMAs of the */ \
            /* previous step run while it is in flight (patch pieces are older than the stage issued after).  */ \
            issue_patch(CH);                                                                              \
            issue_next();                                                                                 \
            SPSP_MFMA_SET((SET) ^ 1)                                                                      \
            if (issued - ((S) + 1) >= 1) wait_vmcnt<LPS>(); else wait_vmcnt<0>();                         \
            __builtin_amdgcn_s_barrier();                                                                 \
            SPSP_READ_FRAGS(TAP, SET, (S) % kNS)                                                          \
        } else {                                                                                          \
            issue_next();                                                                                 \
            SPSP_READ_FRAGS(TAP, SET, (S) % kNS)                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            if ((TAP) > 0 || (S) > 0) SPSP_MFMA_SET((SET) ^ 1)                                            \
        }                                                                                                 \
    }
#define SPSP_CHUNK(P0, CH)                                                                                \
    {                                                                                                     \
        const int s0_ = (CH) * 9;                                                                         \
        SPSP_STEP(0, (0 + P0) & 1, CH, s0_ + 0) SPSP_STEP(1, (1 + P0) & 1, CH, s0_ + 1)                    \
        SPSP_STEP(2, (2 + P0) & 1, CH, s0_ + 2) SPSP_STEP(3, (3 + P0) & 1, CH, s0_ + 3)                    \
        SPSP_STEP(4, (4 + P0) & 1, CH, s0_ + 4) SPSP_STEP(5, (5 + P0) & 1, CH, s0_ + 5)                    \
        SPSP_STEP(6, (6 + P0) & 1, CH, s0_ + 6) SPSP_STEP(7, (7 + P0) & 1, CH, s0_ + 7)                    \
        SPSP_STEP(8, (8 + P0) & 1, CH, s0_ + 8)                                                            \
    }
    for (int ch = 0; ch < nchunks; ch += 2) {
        SPSP_CHUNK(0, ch)
        if (ch + 1 < nchunks) SPSP_CHUNK(1, ch + 1)
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if ((nsteps - 1) & 1) SPSP_MFMA_SET(1) else SPSP_MFMA_SET(0)
#undef SPSP_CHUNK
#undef SPSP_STEP
#undef SPSP_MFMA_SET
#undef SPSP_READ_FRAGS

    CSTAMP(4);
    if constexpr (KS == 2) {
        // Combine the two K-halves: wave (kw, w) keeps channel tiles i = 2kw, 2kw+1 and hands the other two to its
        // partner through LDS (ring + patch are dead by now; 8 waves x 8 KB).  Static register indices only.
        __syncthreads();
        float* xch = reinterpret_cast<float*>(smem);
        auto put = [&](auto GIVE) {
            constexpr int g = decltype(GIVE)::value;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<f32x4*>(xch + ((size_t)(wave_id * 8 + ii * 4 + j) * 64 + lane) * 4) = acc[g + ii][j];
        };
        if (kw == 0) put(std::integral_constant<int, 2>{}); else put(std::integral_constant<int, 0>{});
        __syncthreads();
        const int partner = wave_id ^ 4;
        auto get = [&](auto KEEP) {
            constexpr int k = decltype(KEEP)::value;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[k + ii][j] += *reinterpret_cast<const f32x4*>(xch + ((size_t)(partner * 8 + ii * 4 + j) * 64 + lane) * 4);
        };
        if (kw == 0) get(std::integral_constant<int, 0>{}); else get(std::integral_constant<int, 2>{});
    }

    // ---- epilogue, staged through LDS so that HBM sees whole 128-byte lines ----
    // In the accumulator a lane owns 4 channels (8 B) of a pixel: stored directly, one wave-instruction touches a
    // quarter of 16 different lines (measured: 10-20 k cycles per workgroup, more than the whole K loop).  Instead:
    //   A  residual tile  global -> LDS   (16 B per lane, 8 lanes = one pixel's 128-B channel block)
    //   C  per-lane math, residual from LDS, packed bf16 into an LDS output tile
    //   E  output tile    LDS -> global   (16 B per lane, whole lines)
    // LDS tiles are [256 px][128 B] with the 16-byte chunk XOR-swizzled by the pixel (bank-conflict-free both ways).
    // tile pixel tp = wave*64 + j*16 + frow.
    if (a.splitk > 1) {
        // partial tile -> slab [blockIdx.y][m][n] (m = linear pixel index); a lane's 4 channels are one 16-byte store
        float* const slab = a.part + (size_t)blockIdx.y * a.M * a.N;
        auto put_slab = [&](auto I0c) {
            constexpr int I0 = decltype(I0c)::value, NI = 4 / KS;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tp = wave * 64 + j * 16 + frow;
                size_t m;
                if constexpr (W8) m = (size_t)m_tile * 256 + tp;
                else m = ((size_t)img0 * W + (y0 + (tp >> 4))) * W + (x0 + (tp & 15));
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    *reinterpret_cast<f32x4*>(slab + m * a.N + n_blk + (I0 + i) * 16 + fq * 4) = acc[I0 + i][j];
            }
        };
        if constexpr (KS == 1) put_slab(std::integral_constant<int, 0>{});
        else { if (kw == 0) put_slab(std::integral_constant<int, 0>{}); else put_slab(std::integral_constant<int, 2>{}); }
        return;
    }
    __syncthreads();                                   // ring / patch / exchange buffers are dead from here on
    unsigned char* const resT = smem;
    unsigned char* const outT = smem + 256 * 128;
    auto pix_off = [&](int tp) -> size_t {
        if constexpr (W8) return (size_t)a.out_off + (size_t)(img0 + (tp >> 6)) * a.out_img + (size_t)((tp >> 3) & 7) * a.out_row + (size_t)(tp & 7) * a.out_pix;
        else return (size_t)a.out_off + (size_t)img0 * a.out_img + (size_t)(y0 + (tp >> 4)) * a.out_row + (size_t)(x0 + (tp & 15)) * a.out_pix;
    };
    constexpr int NTHR = 256 * KS;
    if (a.res) {
#pragma unroll
        for (int it = 0; it < 2048 / NTHR; ++it) {
            const int idx = tid + it * NTHR, tp = idx >> 3, c = idx & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(a.res + pix_off(tp) + n_blk + c * 8);
            *reinterpret_cast<uint4*>(resT + tp * 128 + ((c ^ (tp & 7)) * 16)) = v;
        }
        __syncthreads();
    }
    auto epilogue = [&](auto I0c) {
        constexpr int I0 = decltype(I0c)::value, NI = 4 / KS;
        float psum[NI][4];
#pragma unroll
        for (int i = 0; i < NI; ++i) psum[i][0] = psum[i][1] = psum[i][2] = psum[i][3] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tp = wave * 64 + j * 16 + frow;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int nl = (I0 + i) * 16 + fq * 4;                 // channel within the 64-channel block
                const int loff = tp * 128 + ((((I0 + i) * 2 + (fq >> 1)) ^ (tp & 7)) * 16) + (fq & 1) * 8;
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + n_blk + nl);
                const f32x4 c = acc[I0 + i][j];
                float v0 = c[0] + bv.x, v1 = c[1] + bv.y, v2 = c[2] + bv.z, v3 = c[3] + bv.w;
                if (a.res) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(resT + loff);
                    v0 += Elem<F16>::lo(rv.x); v1 += Elem<F16>::hi(rv.x); v2 += Elem<F16>::lo(rv.y); v3 += Elem<F16>::hi(rv.y);
                }
                if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                *reinterpret_cast<uint2*>(outT + loff) = make_uint2(Elem<F16>::pack2(v0, v1), Elem<F16>::pack2(v2, v3));
                psum[i][0] += v0; psum[i][1] += v1; psum[i][2] += v2; psum[i][3] += v3;
            }
        }
        if constexpr (W8) {
            if (a.pool_out) {
                // fused global average pool (resnet.py:157): this wave holds all 64 pixels of image img0 + wave
                const int b = img0 + wave;
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = psum[i][r];
                        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                        if (frow == 0) a.pool_out[(size_t)b * a.N + n_blk + (I0 + i) * 16 + fq * 4 + r] = v * (1.0f / 64.0f);
                    }
            }
        }
    };
    CSTAMP(5);
    if constexpr (KS == 1) epilogue(std::integral_constant<int, 0>{});
    else { if (kw == 0) epilogue(std::integral_constant<int, 0>{}); else epilogue(std::integral_constant<int, 2>{}); }
    if (a.out) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2048 / NTHR; ++it) {
            const int idx = tid + it * NTHR, tp = idx >> 3, c = idx & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(outT + tp * 128 + ((c ^ (tp & 7)) * 16));
            *reinterpret_cast<uint4*>(a.out + pix_off(tp) + n_blk + c * 8) = v;
        }
    }
    CSTAMP(6);
}

template <bool W8, int KS, int NS, bool F16, bool DB>
int launch_t(const ConvArgs& a, hipStream_t st) {
    using G = Geo<W8, KS, NS>;
    constexpr size_t lds = (size_t)NS * kWStage + (DB ? 2 : 1) * G::PATCH;
    static_assert(lds >= 2 * 256 * 128, "epilogue staging (residual + output tile) / K-half exchange need 64 KB");
    SPSP_TRY((ensure_dynamic_lds<&conv3x3_patch_kernel<W8, KS, NS, F16, DB>>((int)lds)));
#ifdef SPSP_CONV_STAMPS
    if (getenv("SPSP_CONV_DEBUG")) {
        ConvArgs b = a;
        (void)hipMalloc(&b.dbg, 16 * sizeof(unsigned long long));
        conv3x3_patch_kernel<W8, KS, NS, F16, DB><<<dim3((a.M / 256) * (a.N / 64)), 256 * KS, lds, st>>>(b);
        (void)hipDeviceSynchronize();
        unsigned long long h[16];
        (void)hipMemcpy(h, b.dbg, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "patch<W8=%d,KS=%d,NS=%d> chunks=%d wgs=%d: K loop %llu | K-half exchange + residual staging %llu | epilogue + stores %llu | total %llu cycles\n",
                (int)W8, KS, NS, a.chunks, (a.M / 256) * (a.N / 64), h[4] - h[0], h[5] - h[4], h[6] - h[5], h[6] - h[0]);
        (void)hipFree(b.dbg);
        return SPSP_OK;
    }
#endif
    conv3x3_patch_kernel<W8, KS, NS, F16, DB><<<dim3((a.M / 256) * (a.N / 64), a.splitk > 1 ? a.splitk : 1), 256 * KS, lds, st>>>(a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

template <bool W8, int KS, int NS, bool DB = false>
int launch(const ConvArgs& a, hipStream_t st) {
    return a.f16 ? launch_t<W8, KS, NS, true, DB>(a, st) : launch_t<W8, KS, NS, false, DB>(a, st);
}

}  // namespace

namespace spsp {

bool conv_patch_supported(const ConvArgs& a) {
    return a.stride == 1 && a.kw_steps == 3 && a.nsteps == 9 * a.chunks && a.in_off == 0 && a.M % 256 == 0 && a.N % 64 == 0 &&
           a.Ho == a.Wo && (a.Wo == 8 || a.Wo % 16 == 0) && (a.splitk <= 1 || (a.part && a.chunks % a.splitk == 0));
}

int launch_conv_patch(const ConvArgs& a, int variant, hipStream_t stream) {
    SPSP_REQUIRE(conv_patch_supported(a), "conv_patch: unsupported convolution (need 3x3 stride-1 pad-1, M%%256==0, N%%64==0)");
    SPSP_REQUIRE(!a.pool_out || a.Wo == 8, "conv_patch: fused avg-pool only for 8x8 maps");
    // variant 0: 4 waves, ring 3 (65-74 KB LDS: two workgroups per CU) -- many-workgroup layers
    // variant 1: 8 waves (wave pairs split each K-step), ring 5 -- few-workgroup deep layers, one workgroup per CU
    // variant 3: 8 waves, grouped ring of 2 x 3 slabs -- one barrier per kernel row (three K-steps)
    // variant 4: 3 with TWO patch buffers (the next chunk's patch lands under the current chunk's taps; 131 / 150 KB)
    if (a.Wo == 8) return variant == 0 ? launch<true, 1, 3>(a, stream) : variant == 3 ? launch<true, 2, 6>(a, stream) :
                          variant == 4 ? launch<true, 2, 6, true>(a, stream) : launch<true, 2, 5>(a, stream);
    return variant == 0 ? launch<false, 1, 3>(a, stream) : variant == 3 ? launch<false, 2, 6>(a, stream) :
           variant == 4 ? launch<false, 2, 6, true>(a, stream) : launch<false, 2, 5>(a, stream);
}

}  // namespace spsp
