// Implicit-GEMM convolution for gfx950: bf16 NHWC activations x bf16 [Cout][K] weights on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation, fused (folded-BN bias + residual + ReLU [+ global avg-pool])
// epilogue.  Written for CDNA4 only: 64-lane wavefronts, 4 waves (2x2) per workgroup, LDS tiles
// double-buffered with register-staged prefetch (global loads of K-step s+1 are issued before the MFMAs of
// step s and written to the other LDS buffer after them; one barrier per K-step).
//
// GEMM view (see conv.hpp for the gather):  D[n][m] = sum_k Wt[n][k] * X[m][k]
//   MFMA "A" operand = weight rows (n), "B" operand = output pixels (m): the accumulator then holds, per
//   lane, 4 CONSECUTIVE CHANNELS of one pixel -> 8-byte NHWC stores / residual loads in the epilogue.
//
// LDS image: row-major [rows][BK] bf16 with the 16-byte chunk index XOR-swizzled by the row
// (chunk ^ (row & (BK/8-1))): a ds_read_b128 fragment read (16 rows x 4 k-chunks per 16-lane group) then
// touches 16 distinct 16-byte slots of the 256-byte bank row -> conflict-free (guide: LDS, T2).
#include "conv.hpp"
#include "conv_elem.hpp"

namespace {
using namespace spsp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


// BM x BN output tile (pixels x channels), BK-deep K-steps, LOADB bytes per lane per activation load.
template <int BM, int BN, int BK, int LOADB, bool F16>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int kThreads = 256;
    constexpr int CH16 = BK / 8;                  // 16-byte chunks per LDS row
    constexpr int ROWB = BK * 2;                  // bytes per LDS row
    constexpr int CPR_A = ROWB / LOADB;           // activation loads per row
    constexpr int RPP_A = kThreads / CPR_A;       // rows per pass
    constexpr int PA = BM / RPP_A;                // passes
    constexpr int DW_A = LOADB / 4;
    constexpr int CPR_B = CH16;
    constexpr int RPP_B = kThreads / CPR_B;
    constexpr int PB = BN / RPP_B;
    constexpr int MT = BM / 32, NT = BN / 32;     // 16x16 tiles per wave (2x2 waves)
    static_assert(PA >= 1 && PB >= 1 && BM % RPP_A == 0 && BN % RPP_B == 0, "tile/loader mismatch");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                             // [2][BM][ROWB]
    unsigned char* Bs = smem + 2 * BM * ROWB;             // [2][BN][ROWB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (guide T1): workgroups are dealt round-robin over the 8 XCDs, each with a private
    // 4 MB L2.  Give XCD x the x-th CONTIGUOUS eighth of the tile list (n-tile major, m-tile minor), so the
    // workgroups sharing an L2 work on one weight panel and on neighbouring pixels (halo + 9-tap re-reads
    // become L2 hits).  Placement only affects speed, never results.
    const int mt = a.M / BM;
    const int ntiles = mt * (a.N / BN);
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
    const int m_blk = (tile % mt) * BM, n_blk = (tile / mt) * BN;
    const int HoWo = a.Ho * a.Wo;

    // ---- loader set-up: this thread's activation rows (fixed for the whole K loop) ----
    const int a_chunk = tid % CPR_A, a_row0 = tid / CPR_A;
    int a_base[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        int m = m_blk + a_row0 + p * RPP_A;
        m = m < a.M ? m : a.M - 1;
        const int b = m / HoWo, rem = m - b * HoWo, ho = rem / a.Wo, wo = rem - ho * a.Wo;
        a_base[p] = a.in_off + b * a.in_img + ho * a.stride * a.in_row + wo * a.stride * a.in_pix + a_chunk * (LOADB / 2);
    }
    const int b_chunk = tid % CPR_B, b_row0 = tid / CPR_B;
    const int Ktot = a.nsteps * BK;
    const uint16_t* w_ptr = a.w + (size_t)(n_blk + b_row0) * Ktot + b_chunk * 8;

    uint32_t ra[PA][DW_A];
    uint32_t rb[PB][4];

    // scalar copies: capturing the kernarg struct by reference in the lambdas would spill it to scratch
    const uint16_t* const in_ptr = a.in;
    const int in_row = a.in_row, in_pix = a.in_pix, kw_steps = a.kw_steps, chunks = a.chunks, nsteps = a.nsteps;

    auto load_step = [&](int step) {
        const int per_r = kw_steps * chunks;
        const int r = step / per_r, rem = step - r * per_r, c = rem / chunks, ch = rem - c * chunks;
        const int soff = r * in_row + c * in_pix + ch * BK;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const uint16_t* src = in_ptr + a_base[p] + soff;
            if constexpr (LOADB == 16) {
                const uint4 v = *reinterpret_cast<const uint4*>(src);
                ra[p][0] = v.x; ra[p][1] = v.y; ra[p][2] = v.z; ra[p][3] = v.w;
            } else {
                const uint2 v = *reinterpret_cast<const uint2*>(src);
                ra[p][0] = v.x; ra[p][1] = v.y;
            }
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const uint4 v = *reinterpret_cast<const uint4*>(w_ptr + (size_t)p * RPP_B * Ktot + (size_t)step * BK);
            rb[p][0] = v.x; rb[p][1] = v.y; rb[p][2] = v.z; rb[p][3] = v.w;
        }
    };
    auto store_step = [&](int buf) {
        unsigned char* Ad = As + buf * BM * ROWB;
        unsigned char* Bd = Bs + buf * BN * ROWB;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int row = a_row0 + p * RPP_A;
            if constexpr (LOADB == 16) {
                const int phys = a_chunk ^ (row & (CH16 - 1));
                *reinterpret_cast<uint4*>(Ad + row * ROWB + phys * 16) = make_uint4(ra[p][0], ra[p][1], ra[p][2], ra[p][3]);
            } else {
                const int phys = (a_chunk >> 1) ^ (row & (CH16 - 1));
                *reinterpret_cast<uint2*>(Ad + row * ROWB + phys * 16 + (a_chunk & 1) * 8) = make_uint2(ra[p][0], ra[p][1]);
            }
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int row = b_row0 + p * RPP_B;
            const int phys = b_chunk ^ (row & (CH16 - 1));
            *reinterpret_cast<uint4*>(Bd + row * ROWB + phys * 16) = make_uint4(rb[p][0], rb[p][1], rb[p][2], rb[p][3]);
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    load_step(0);
    store_step(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) load_step(step + 1);
        const unsigned char* Ab = As + buf * BM * ROWB + (wm * (BM / 2)) * ROWB;
        const unsigned char* Bb = Bs + buf * BN * ROWB + (wn * (BN / 2)) * ROWB;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 wf[NT], xf[MT];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int row = i * 16 + frow;
                const int phys = (ks * 4 + fq) ^ ((wn * (BN / 2) + row) & (CH16 - 1));
                wf[i] = *reinterpret_cast<const bf16x8*>(Bb + row * ROWB + phys * 16);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int row = j * 16 + frow;
                const int phys = (ks * 4 + fq) ^ ((wm * (BM / 2) + row) & (CH16 - 1));
                xf[j] = *reinterpret_cast<const bf16x8*>(Ab + row * ROWB + phys * 16);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = Elem<F16>::mfma(wf[i], xf[j], acc[i][j]);
        }
        if (step + 1 < nsteps) store_step(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds channels n0..n0+3 of pixel (lane & 15) of each 16x16 tile ----
    float psum[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) psum[i][0] = psum[i][1] = psum[i][2] = psum[i][3] = 0.f;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m_blk + wm * (BM / 2) + j * 16 + frow;
        const bool valid = m < a.M;
        const int mm = valid ? m : a.M - 1;
        const int b = mm / HoWo, rem = mm - b * HoWo, ho = rem / a.Wo, wo = rem - ho * a.Wo;
        const size_t obase = (size_t)a.out_off + (size_t)b * a.out_img + (size_t)ho * a.out_row + (size_t)wo * a.out_pix;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int n = n_blk + wn * (BN / 2) + i * 16 + fq * 4;
            const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (a.res) {
                const uint2 rv = *reinterpret_cast<const uint2*>(a.res + obase + n);
                v0 += Elem<F16>::lo(rv.x); v1 += Elem<F16>::hi(rv.x); v2 += Elem<F16>::lo(rv.y); v3 += Elem<F16>::hi(rv.y);
            }
            if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            if (valid) {
                if (a.out) *reinterpret_cast<uint2*>(a.out + obase + n) = make_uint2(Elem<F16>::pack2(v0, v1), Elem<F16>::pack2(v2, v3));
                psum[i][0] += v0; psum[i][1] += v1; psum[i][2] += v2; psum[i][3] += v3;
            }
        }
    }
    if (a.pool_out) {
        // fused global average pool (resnet.py:157,213): the tile's BM rows are exactly one image.
        // Sum the 16 pixel-lanes of each quad group, then the two M-waves through LDS (fixed order).
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
            a.pool_out[(size_t)b * a.N + n_blk + tid] = (red[tid] + red[BN + tid]) / (float)HoWo;
        }
    }
}

template <int BM, int BN, int BK, int LOADB, bool F16>
int launch_t(const ConvArgs& a, hipStream_t st) {
    SPSP_REQUIRE(a.N % BN == 0, "conv: N=%d not a multiple of the %d-channel tile", a.N, BN);
    SPSP_REQUIRE(a.M % BM == 0, "conv: M=%d not a multiple of the %d-pixel tile", a.M, BM);
    SPSP_REQUIRE(!a.pool_out || BM == a.Ho * a.Wo, "conv: fused avg-pool needs one image per M tile");
    constexpr size_t lds = 2 * (size_t)(BM + BN) * BK * 2;
    static bool attr_set = false;
    if (!attr_set) {
        SPSP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, BK, LOADB, F16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid((a.M / BM) * (a.N / BN));
    conv_igemm_kernel<BM, BN, BK, LOADB, F16><<<grid, 256, lds, st>>>(a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

template <int BM, int BN, int BK, int LOADB>
int launch(const ConvArgs& a, hipStream_t st) {
    return a.f16 ? launch_t<BM, BN, BK, LOADB, true>(a, st) : launch_t<BM, BN, BK, LOADB, false>(a, st);
}

}  // namespace

namespace spsp {

int launch_conv(const ConvArgs& a, ConvTile tile, hipStream_t stream) {
    switch (tile) {
        case kTile128x64:  return launch<128, 64, 64, 16>(a, stream);
        case kTile128x128: return launch<128, 128, 64, 16>(a, stream);
        case kTile64x128:  return launch<64, 128, 64, 16>(a, stream);
        case kTile64x64:   return launch<64, 64, 64, 16>(a, stream);
        case kTileConv1:   return launch<128, 64, 128, 8>(a, stream);
    }
    set_error("conv: unknown tile %d", (int)tile);
    return SPSP_EINVAL;
}

}  // namespace spsp
