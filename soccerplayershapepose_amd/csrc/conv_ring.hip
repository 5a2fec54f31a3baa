// Implicit-GEMM convolution, second generation: NS-stage LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: HBM/L2 -> LDS with no VGPR round trip), counted vmcnt waits, ONE raw s_barrier per
// K-step, fragment reads by inline-asm ds_read_b128 (hipcc would otherwise drain vmcnt(0) in front of every
// ds_read that might alias an in-flight LDS-DMA).
//
// Why: at 32 crops the deep layers have only 256 workgroups (one per CU, one wave per SIMD); with a single
// prefetch stage every K-step exposes a full L2/MALL round trip.  A ring keeps NS-1 K-steps of loads in flight
// per workgroup, and split-K turns 64 fat 128x128 tiles into 256 workgroups without shrinking the tile (the
// L2 -> LDS traffic of an implicit GEMM falls with the tile edge).
//
// Geometry / operand roles / swizzle are those of conv.hip: weights = MFMA A operand, pixels = B operand,
// [rows][64] bf16 LDS tiles with chunk ^= row & 7.  With LDS-DMA the LDS image is lane-linear (one
// wave-instruction writes 8 rows x 128 B), so the swizzle is applied to the per-lane SOURCE address and again on
// the fragment read (guide rule 21: both sides or neither).
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

template <int BM, int BN, int NS, bool F16>
__global__ __launch_bounds__(256) void conv_ring_kernel(const ConvArgs a) {
    constexpr int BK = 64, ROWB = 128;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int IA = BM / 32, IB = BN / 32;          // LDS-DMA instructions per wave per stage (1 KiB each)
    constexpr int LPS = IA + IB;
    constexpr int MT = BM / 32, NT = BN / 32;          // 16x16 tiles per wave (2x2 waves)
    static_assert(NS >= 2 && NS <= 4, "ring depth");

    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int HoWo = a.Ho * a.Wo;

    // tile order: XCD-contiguous
    const int mt = a.M / BM;
    const int ntiles = mt * (a.N / BN);
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
    const int m_blk = (tile % mt) * BM, n_blk = (tile / mt) * BN;
    // Optional SECOND convolution on the same pixels (a.n2 == a.N channels): the 1x1 stride-2 downsample of a residual stage
    // rides with the stage's 3x3 stride-2 conv.  Its input pixels are the 3x3's centre tap, so the workgroup simply runs
    // `chunks` more K-steps behind the 9 * chunks of the 3x3 -- pixel rows of the centre tap again, weight slabs from w2,
    // a second accumulator set -- through the same ring (the prefetch runs across the seam), and stores two tiles.
    const bool dual = a.n2 > 0;
    const int n1 = a.nsteps, nsteps = n1 + (dual ? a.chunks : 0);

    // ---- LDS-DMA sources: lane l of instruction i feeds LDS row (wave*ROWS/4 + i*8 + l/8), physical chunk l%8,
    //      i.e. logical chunk (l%8) ^ (row%8) of that row ----
    const int drow = lane >> 3, dchunk = lane & 7;
    const uint16_t* a_src[IA];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int row = wave * (BM / 4) + i * 8 + drow;
        const int m = m_blk + row;
        const int b = m / HoWo, rem = m - b * HoWo, ho = rem / a.Wo, wo = rem - ho * a.Wo;
        a_src[i] = a.in + a.in_off + (size_t)b * a.in_img + ho * a.stride * a.in_row + wo * a.stride * a.in_pix +
                   ((dchunk ^ (row & 7)) * 8);
    }
    const int Ktot = n1 * BK, Ktot2 = a.chunks * BK;
    const uint16_t* b_src[IB];
    const uint16_t* b2_src[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int row = wave * (BN / 4) + i * 8 + drow;
        b_src[i] = a.w + (size_t)(n_blk + row) * Ktot + ((dchunk ^ (row & 7)) * 8);
        b2_src[i] = dual ? a.w2 + (size_t)(n_blk + row) * Ktot2 + ((dchunk ^ (row & 7)) * 8) : b_src[i];
    }
    const int in_row = a.in_row, in_pix = a.in_pix, kw_steps = a.kw_steps, chunks = a.chunks;

    auto issue = [&](int gstep, int buf) {
        const bool second = gstep >= n1;                       // centre tap (r = c = 1 of the 3x3) + chunk gstep - n1
        const int g = second ? 4 * chunks + (gstep - n1) : gstep;
        const int per_r = kw_steps * chunks;
        const int r = g / per_r, rem = g - r * per_r, c = rem / chunks, ch = rem - c * chunks;
        const int soff = r * in_row + c * in_pix + ch * BK;
        unsigned char* sa = smem + buf * STAGE + wave * (BM / 4) * ROWB;
        unsigned char* sb = smem + buf * STAGE + BM * ROWB + wave * (BN / 4) * ROWB;
#pragma unroll
        for (int i = 0; i < IA; ++i) dma16(a_src[i] + soff, sa + i * 8 * ROWB);
#pragma unroll
        for (int i = 0; i < IB; ++i) dma16(second ? b2_src[i] + (size_t)(gstep - n1) * BK : b_src[i] + (size_t)gstep * BK, sb + i * 8 * ROWB);
    };

    f32x4 acc[NT][MT], acc2[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // fragment read addresses: row = region + t*16 + frow, chunk (ks*4 + fq) ^ (frow & 7)  (region bases are
    // multiples of 16 rows, so row & 7 == frow & 7); ks = 1 flips byte-offset bit 6.
    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t lane_off = frow * ROWB + ((fq ^ (frow & 7)) * 16);
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;      // LDS addresses are 32-bit offsets
    const uint32_t xa0 = smem_base + wm * (BM / 2) * ROWB;               // this wave's pixel rows
    const uint32_t wb0 = smem_base + BM * ROWB + wn * (BN / 2) * ROWB;   // this wave's weight rows

    int issued = 0;
    for (; issued < NS - 1 && issued < nsteps; ++issued) issue(issued, issued);

    // one K-step: wait for its stage, refill the ring, fragments, 2 x NT x MT MFMAs into ACC
#define RING_STEP(S, ACC)                                                                              \
    {                                                                                                  \
        const int ahead = issued - ((S) + 1);          /* stages in flight behind the one about to be read */ \
        if (ahead >= 2) wait_vmcnt<2 * LPS>();                                                         \
        else if (ahead == 1) wait_vmcnt<LPS>();                                                        \
        else wait_vmcnt<0>();                                                                          \
        __builtin_amdgcn_s_barrier();                                                                  \
        if (issued < nsteps) { issue(issued, issued % NS); ++issued; }                                 \
        const uint32_t so = (uint32_t)(((S) % NS) * STAGE);                                            \
        bf16x8 wf[2][NT], xf[2][MT];                                                                   \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                             \
            const uint32_t lo = lane_off ^ (ks * 64);                                                  \
            const uint32_t wa = wb0 + so + lo, pa = xa0 + so + lo;                                     \
            wf[ks][0] = lds_read_b128<0>(wa);                                                          \
            if constexpr (NT > 1) wf[ks][1] = lds_read_b128<2048>(wa);                                 \
            if constexpr (NT > 2) { wf[ks][2] = lds_read_b128<4096>(wa); wf[ks][3] = lds_read_b128<6144>(wa); } \
            xf[ks][0] = lds_read_b128<0>(pa);                                                          \
            if constexpr (MT > 1) xf[ks][1] = lds_read_b128<2048>(pa);                                 \
            if constexpr (MT > 2) { xf[ks][2] = lds_read_b128<4096>(pa); xf[ks][3] = lds_read_b128<6144>(pa); } \
        }                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NT + MT) : "memory");                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        _Pragma("unroll") for (int i = 0; i < NT; ++i)                                                 \
            _Pragma("unroll") for (int j = 0; j < MT; ++j)                                             \
                ACC[i][j] = Elem<F16>::mfma(wf[0][i], xf[0][j], ACC[i][j]);                            \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        _Pragma("unroll") for (int i = 0; i < NT; ++i)                                                 \
            _Pragma("unroll") for (int j = 0; j < MT; ++j)                                             \
                ACC[i][j] = Elem<F16>::mfma(wf[1][i], xf[1][j], ACC[i][j]);                            \
    }
    for (int s = 0; s < n1; ++s) RING_STEP(s, acc)
    for (int s = n1; s < nsteps; ++s) RING_STEP(s, acc2)
#undef RING_STEP

    // ---- epilogue(s): bias (+ residual) (+ ReLU) -> element type; a lane owns 4 channels of a pixel ----
    auto store_tile = [&](f32x4 (&A)[NT][MT], const float* bias_ptr, const uint16_t* res_ptr, uint16_t* out_ptr, int relu, float* pool_out) {
        float psum[NT][4];
#pragma unroll
        for (int i = 0; i < NT; ++i) psum[i][0] = psum[i][1] = psum[i][2] = psum[i][3] = 0.f;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m_blk + wm * (BM / 2) + j * 16 + frow;
            const int b = m / HoWo, rem = m - b * HoWo, ho = rem / a.Wo, wo = rem - ho * a.Wo;
            const size_t obase = (size_t)a.out_off + (size_t)b * a.out_img + (size_t)ho * a.out_row + (size_t)wo * a.out_pix;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int n = n_blk + wn * (BN / 2) + i * 16 + fq * 4;
                const float4 bv = *reinterpret_cast<const float4*>(bias_ptr + n);
                float v0 = A[i][j][0] + bv.x, v1 = A[i][j][1] + bv.y, v2 = A[i][j][2] + bv.z, v3 = A[i][j][3] + bv.w;
                if (res_ptr) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(res_ptr + obase + n);
                    v0 += Elem<F16>::lo(rv.x); v1 += Elem<F16>::hi(rv.x); v2 += Elem<F16>::lo(rv.y); v3 += Elem<F16>::hi(rv.y);
                }
                if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                if (out_ptr) *reinterpret_cast<uint2*>(out_ptr + obase + n) = make_uint2(Elem<F16>::pack2(v0, v1), Elem<F16>::pack2(v2, v3));
                psum[i][0] += v0; psum[i][1] += v1; psum[i][2] += v2; psum[i][3] += v3;
            }
        }
        if (pool_out) {
            __syncthreads();
            float* red = reinterpret_cast<float*>(smem);          // [2 (wm)][BN]
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = psum[i][r];
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                    if (frow == 0) red[wm * BN + wn * (BN / 2) + i * 16 + fq * 4 + r] = v;
                }
            __syncthreads();
            if (tid < BN) {
                const int b = m_blk / HoWo;
                pool_out[(size_t)b * a.N + n_blk + tid] = (red[tid] + red[BN + tid]) / (float)HoWo;
            }
        }
    };
    store_tile(acc, a.bias, a.res, a.out, a.relu, a.pool_out);
    if (dual) store_tile(acc2, a.bias2, nullptr, a.out2, 0, nullptr);
}

template <int BM, int BN, int NS, bool F16>
int launch_t(const ConvArgs& a, hipStream_t st) {
    SPSP_REQUIRE(a.N % BN == 0 && a.M % BM == 0, "conv_ring: M=%d N=%d not multiples of the %dx%d tile", a.M, a.N, BM, BN);
    SPSP_REQUIRE(a.in_pix % 8 == 0 && a.in_off % 8 == 0 && a.in_row % 8 == 0, "conv_ring: input not 16-byte aligned");
    constexpr size_t lds = (size_t)NS * (BM + BN) * 128;
    SPSP_REQUIRE(a.n2 == 0 || (a.n2 == a.N && a.w2 && a.bias2 && a.out2 && a.kw_steps == 3 && a.nsteps == 9 * a.chunks && !a.pool_out),
                 "conv_ring: the second convolution must be a 1x1 with the 3x3's channel count riding on it");
    const int tiles = (a.M / BM) * (a.N / BN);
    SPSP_REQUIRE(!a.pool_out || BM == a.Ho * a.Wo, "conv_ring: fused avg-pool needs one image per M tile");
    SPSP_TRY((ensure_dynamic_lds<&conv_ring_kernel<BM, BN, NS, F16>>((int)lds)));
    launch_kernel(&conv_ring_kernel<BM, BN, NS, F16>, dim3(tiles), dim3(256), lds, st, a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

template <int BM, int BN, int NS>
int launch(const ConvArgs& a, hipStream_t st) {
    return a.f16 ? launch_t<BM, BN, NS, true>(a, st) : launch_t<BM, BN, NS, false>(a, st);
}

}  // namespace

namespace spsp {

int launch_conv_ring(const ConvArgs& a, RingTile tile, hipStream_t stream) {
    switch (tile) {
        case kRing128x128s3: return launch<128, 128, 3>(a, stream);
        case kRing128x128s2: return launch<128, 128, 2>(a, stream);
        case kRing128x64s3:  return launch<128, 64, 3>(a, stream);
        case kRing128x64s4:  return launch<128, 64, 4>(a, stream);
        case kRing64x128s3:  return launch<64, 128, 3>(a, stream);
        case kRing64x64s4:   return launch<64, 64, 4>(a, stream);
        case kRingNone:      break;
    }
    set_error("conv_ring: unknown tile %d", (int)tile);
    return SPSP_EINVAL;
}

}  // namespace spsp
