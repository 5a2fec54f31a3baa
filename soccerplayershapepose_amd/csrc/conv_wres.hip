// 3x3 / stride-1 convolution, 64 -> 64 channels (ResNet-18 layer1: 4 convs, 12 % of the FLOPs but the largest share of
// GPU time of the patch kernels) with the WHOLE WEIGHT TENSOR RESIDENT IN LDS.
//
// conv_patch.hip streams one [64 x 64] weight slab per filter tap through an LDS-DMA ring and therefore pays a
// counted-vmcnt wait + workgroup barrier per tap (measured ~600 cycles of overhead next to 512 cycles of MFMA).  With
// Cin = Cout = 64 all nine slabs are 73.7 KB: this kernel loads them ONCE per workgroup, makes the workgroups
// persistent over the 16 x 16-pixel tiles, and runs the 9 taps x 32 MFMAs of a tile back to back with no barrier at
// all; fragment reads are software-pipelined one tap ahead (two register sets), the next tile's input patch is
// fetched by LDS-DMA while the current tile's epilogue runs.
//
// Workgroup = 4 waves (one per SIMD), wave tile 64 px x 64 ch on v_mfma_f32_16x16x32_bf16 (weights = A operand,
// pixels = B operand).  LDS: weights 73.7 KB + patch 41 KB + epilogue staging 32 KB = 147 KB.
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

constexpr int kWSlab = 64 * 128;                 // one tap: [64 n][64 k] bf16
constexpr int kWAll = 9 * kWSlab;                // 73 728 B
constexpr int kNPix = 324, kNDma = 41;           // 18 x 18 patch, 8 pixels per 1-KiB DMA piece
constexpr int kPatch = kNDma * 1024;             // 41 984 B
constexpr int kStage = 256 * 128;                // epilogue staging tile [256 px][128 B]
constexpr int kLds = kWAll + kPatch + kStage + 16; // 148 496 B (last 16 B: tile-id broadcast slot)

#ifdef SPSP_CONV_STAMPS
#define WSTAMP(i) if (a.dbg && blockIdx.x == 8 && tid == 0) a.dbg[i] = __builtin_amdgcn_s_memtime()
#else
#define WSTAMP(i)
#endif

template <bool F16>
__global__ __launch_bounds__(256) void conv3x3_wres_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const wlds = smem;
    unsigned char* const patch = smem + kWAll;
    unsigned char* const stage = smem + kWAll + kPatch;
    volatile int* const s_tile = reinterpret_cast<volatile int*>(smem + kWAll + kPatch + kStage);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.Wo, tpr = W / 16, tpi = tpr * tpr;
    const int ntiles = a.M / 256;
    const uint16_t* const in_ptr = a.in;

    // first tile id (dynamic scheduler, see below) BEFORE any DMA is in flight: the atomic's return is then the only
    // thing its s_waitcnt waits for
    if (tid == 0) s_tile[0] = (int)atomicAdd(a.counter, 1u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s_tile[0] >= ntiles) return;                        // a late starter: the resident workgroups took every tile

    // ---- all nine weight slabs, once: piece p = 8 rows of one tap; lane l -> row l/8, physical chunk l%8 ----
    const int drow = lane >> 3, dchunk = lane & 7;
    {
        const int Ktot = 9 * 64;
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int p = wave + 4 * i, tap = p >> 3, row = (p & 7) * 8 + drow;
            dma16(a.w + (size_t)row * Ktot + tap * 64 + ((dchunk ^ (row & 7)) * 8), wlds + p * 1024);
        }
    }
    // ---- patch sources (relative to the tile origin): lane l of piece p feeds pixel q = 8p + l/8, chunk l%8 ^ f(q) ----
    int p_rel[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        int q = (wave + 4 * i) * 8 + drow;
        q = q < kNPix ? q : kNPix - 1;
        const int py = q / 18, px = q - py * 18;
        p_rel[i] = py * a.in_row + px * a.in_pix + ((dchunk ^ (q & 7)) * 8);        // chunk swizzle q & 7: conflict-free ds_read_b128 (see conv_patch.hip)
    }
    auto tile_origin = [&](int t, int& img, int& y0, int& x0) {
        img = t / tpi;
        const int r = t - img * tpi;
        y0 = (r / tpr) * 16; x0 = (r % tpr) * 16;
    };
    auto issue_patch = [&](int t) {
        int img, y0, x0;
        tile_origin(t, img, y0, x0);
        const uint16_t* base = in_ptr + (size_t)img * a.in_img + (size_t)y0 * a.in_row + (size_t)x0 * a.in_pix;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const int p = wave + 4 * i;
            if (p < kNDma) dma16(base + p_rel[i], patch + p * 1024);
        }
    };

    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    const uint32_t w_lane = smem_base + frow * 128 + ((fq ^ (frow & 7)) * 16);       // ks = 0; ks = 1: ^ 64 on the low part
    const uint32_t w_lane1 = smem_base + frow * 128 + (((4 + fq) ^ (frow & 7)) * 16);
    const int qb = (wave * 4) * 18 + frow;
    uint32_t xoff[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = qb + j * 18 + (tap / 3) * 18 + (tap % 3);
            xoff[tap][j] = smem_base + kWAll + q * 128 + ((fq ^ (q & 7)) * 16);
        }

    // bias of this lane's 4 x 4 channels, once
    float4 bias_r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias_r[i] = *reinterpret_cast<const float4*>(a.bias + i * 16 + fq * 4);
    // residual tile by LDS-DMA straight into the staging tile (same swizzled image the epilogue uses): piece p =
    // pixels 8p..8p+7, lane l -> pixel 8p + l/8, physical chunk l%8 = logical chunk ^ (pixel & 7)
    const bool has_res = a.res != nullptr;
    auto issue_res = [&](int tt) {
        int img, y0, x0;
        tile_origin(tt, img, y0, x0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = wave + 4 * i, tp = p * 8 + drow;
            const size_t o = (size_t)a.out_off + (size_t)img * a.out_img + (size_t)(y0 + (tp >> 4)) * a.out_row + (size_t)(x0 + (tp & 15)) * a.out_pix;
            dma16(a.res + o + ((dchunk ^ (tp & 7)) * 8), stage + p * 1024);
        }
    };
    WSTAMP(0);
    // Dynamic tile scheduler: workgroups take tiles from a device counter (zero before the launch) instead of the
    // static b, b + grid, ...  The grid is one workgroup per CU, but in the multi-stream pipeline some CUs are held by
    // other kernels' workgroups (a fit workgroup keeps its CU's LDS for 230 us): with static tiles the workgroups of
    // those CUs start late and the launch takes two rounds (measured 43 us under contention vs 21 us alone); with the
    // counter the workgroups that ARE resident take the work and late starters find none.
    int t = s_tile[0];                                      // fetched before the weight DMA was issued (top of the kernel)
    if (t < ntiles) issue_patch(t);
    bool first = true;
    while (t < ntiles) {
        // weights (first tile) + this tile's patch have landed.  On later tiles the patch DMA was issued BEFORE the
        // previous tile's 8 output stores per lane, and vmcnt counts stores too: leave those 8 in flight.
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        // (the staging tile may still be read by the previous tile's stores of OTHER waves: the barrier below orders
        //  their LDS reads -- already complete, they waited lgkmcnt before storing -- against this tile's residual DMA)
        __builtin_amdgcn_s_barrier();
        if (first) { WSTAMP(1); } else { WSTAMP(5); }
        first = false;
        // next tile id: issued now, read after the K loop.  Inline asm: hipcc's atomicAdd waits vmcnt(0) right behind the
        // instruction (wave-aggregated lowering), which would park this wave on the previous tile's stores as well.
        unsigned next_raw = 0;
        if (tid == 0) {
            const unsigned one = 1u;
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(next_raw) : "v"(a.counter), "v"(one) : "memory");
        }
        if (has_res) issue_res(t);                            // lands during the K loop

        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- 9 taps, no barrier: fragments of tap+1 are read while the 32 MFMAs of tap run ----
        bf16x8 wfr[2][2][4], xfr[2][2][4];
#define WRES_READ(TAP, SET)                                                                    \
    {                                                                                          \
        const uint32_t w0_ = w_lane + (TAP) * kWSlab, w1_ = w_lane1 + (TAP) * kWSlab;          \
        wfr[SET][0][0] = lds_read_b128<0>(w0_);                                                \
        wfr[SET][0][1] = lds_read_b128<2048>(w0_);                                             \
        wfr[SET][0][2] = lds_read_b128<4096>(w0_);                                             \
        wfr[SET][0][3] = lds_read_b128<6144>(w0_);                                             \
        wfr[SET][1][0] = lds_read_b128<0>(w1_);                                                \
        wfr[SET][1][1] = lds_read_b128<2048>(w1_);                                             \
        wfr[SET][1][2] = lds_read_b128<4096>(w1_);                                             \
        wfr[SET][1][3] = lds_read_b128<6144>(w1_);                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
            xfr[SET][0][j] = lds_read_b128<0>(xoff[TAP][j]);                                   \
            xfr[SET][1][j] = lds_read_b128<0>(xoff[TAP][j] ^ 64);                              \
        }                                                                                      \
    }
#define WRES_MFMA(SET)                                                                         \
    {                                                                                          \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                       \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                      \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                  \
                    acc[i][j] = Elem<F16>::mfma(wfr[SET][kk][i], xfr[SET][kk][j], acc[i][j]); \
    }
        // One tap: the 16 fragment reads of tap TAP (set SET) go out between the 32 MFMAs of the previous tap (set SET^1),
        // one read per MFMA pair and all of them within the first half, so that their issue slots and their latency sit
        // under the matrix pipe.  Every sched_barrier pins that order: hipcc otherwise gathers the reads in front and sinks
        // the register-only MFMAs BELOW the wait (measured: the wave then waits for reads it has just issued).
#define WRES_SB __builtin_amdgcn_sched_barrier(0);
#define WRES_RW(SET, KK, I, TAP) wfr[SET][KK][I] = lds_read_b128<(I) * 2048>(((KK) ? w_lane1 : w_lane) + (TAP) * kWSlab);
#define WRES_RX(SET, KK, J, TAP) xfr[SET][KK][J] = lds_read_b128<0>(xoff[TAP][J] ^ ((KK) * 64));
#define WRES_M2(S1, KK, I, J)                                                                  \
        WRES_SB                                                                                \
        acc[I][J] = Elem<F16>::mfma(wfr[S1][KK][I], xfr[S1][KK][J], acc[I][J]);                \
        acc[I][(J) + 1] = Elem<F16>::mfma(wfr[S1][KK][I], xfr[S1][KK][(J) + 1], acc[I][(J) + 1]); \
        WRES_SB
#define WRES_STEP(TAP, SET)                                                                    \
    {                                                                                          \
        WRES_RW(SET, 0, 0, TAP) WRES_RW(SET, 0, 1, TAP) WRES_M2((SET) ^ 1, 0, 0, 0)             \
        WRES_RW(SET, 0, 2, TAP) WRES_RW(SET, 0, 3, TAP) WRES_M2((SET) ^ 1, 0, 0, 2)             \
        WRES_RX(SET, 0, 0, TAP) WRES_RX(SET, 0, 1, TAP) WRES_M2((SET) ^ 1, 0, 1, 0)             \
        WRES_RX(SET, 0, 2, TAP) WRES_RX(SET, 0, 3, TAP) WRES_M2((SET) ^ 1, 0, 1, 2)             \
        WRES_RW(SET, 1, 0, TAP) WRES_RW(SET, 1, 1, TAP) WRES_M2((SET) ^ 1, 0, 2, 0)             \
        WRES_RW(SET, 1, 2, TAP) WRES_RW(SET, 1, 3, TAP) WRES_M2((SET) ^ 1, 0, 2, 2)             \
        WRES_RX(SET, 1, 0, TAP) WRES_RX(SET, 1, 1, TAP) WRES_M2((SET) ^ 1, 0, 3, 0)             \
        WRES_RX(SET, 1, 2, TAP) WRES_RX(SET, 1, 3, TAP) WRES_M2((SET) ^ 1, 0, 3, 2)             \
        WRES_M2((SET) ^ 1, 1, 0, 0) WRES_M2((SET) ^ 1, 1, 0, 2) WRES_M2((SET) ^ 1, 1, 1, 0) WRES_M2((SET) ^ 1, 1, 1, 2) \
        WRES_M2((SET) ^ 1, 1, 2, 0) WRES_M2((SET) ^ 1, 1, 2, 2) WRES_M2((SET) ^ 1, 1, 3, 0) WRES_M2((SET) ^ 1, 1, 3, 2) \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
        WRES_SB                                                                                \
    }
        WRES_READ(0, 0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        WRES_STEP(1, 1) WRES_STEP(2, 0) WRES_STEP(3, 1) WRES_STEP(4, 0)
        WRES_STEP(5, 1) WRES_STEP(6, 0) WRES_STEP(7, 1) WRES_STEP(8, 0)
        WRES_MFMA(0)
#undef WRES_STEP
#undef WRES_M2
#undef WRES_RX
#undef WRES_RW
#undef WRES_SB
#undef WRES_MFMA
#undef WRES_READ

        if (t == (int)blockIdx.x) { WSTAMP(2); } else { WSTAMP(6); }
        if (tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(next_raw) : : "memory");     // the atomic's return (and the residual DMA, long landed)
            s_tile[0] = (int)next_raw;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // every wave is done with the patch; next tile id published
        const int tn = s_tile[0];
        if (tn < ntiles) issue_patch(tn);                     // next tile's patch streams in under the epilogue

        // ---- epilogue through the LDS staging tile (see conv_patch.hip): whole 128-byte lines to and from HBM ----
        int img, y0, x0;
        tile_origin(t, img, y0, x0);
        auto pix_off = [&](int tp) -> size_t {
            return (size_t)a.out_off + (size_t)img * a.out_img + (size_t)(y0 + (tp >> 4)) * a.out_row + (size_t)(x0 + (tp & 15)) * a.out_pix;
        };
        if (has_res) {
            // residual DMA (8 pieces) is older than the next-patch DMA (<= 11 pieces) just issued
            if (tn < ntiles) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tp = wave * 64 + j * 16 + frow;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int loff = tp * 128 + (((i * 2 + (fq >> 1)) ^ (tp & 7)) * 16) + (fq & 1) * 8;
                const float4 bv = bias_r[i];
                const f32x4 c = acc[i][j];
                float v0 = c[0] + bv.x, v1 = c[1] + bv.y, v2 = c[2] + bv.z, v3 = c[3] + bv.w;
                if (has_res) {
                    const uint2 rv = *reinterpret_cast<const uint2*>(stage + loff);      // own slot: read, then overwrite
                    v0 += Elem<F16>::lo(rv.x); v1 += Elem<F16>::hi(rv.x); v2 += Elem<F16>::lo(rv.y); v3 += Elem<F16>::hi(rv.y);
                }
                if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                *reinterpret_cast<uint2*>(stage + loff) = make_uint2(Elem<F16>::pack2(v0, v1), Elem<F16>::pack2(v2, v3));
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = tid + it * 256, tp = idx >> 3, c = idx & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(stage + tp * 128 + ((c ^ (tp & 7)) * 16));
            *reinterpret_cast<uint4*>(a.out + pix_off(tp) + c * 8) = v;
        }
        if (t == (int)blockIdx.x) { WSTAMP(3); } else { WSTAMP(7); }
        // the staging tile is rewritten only after the next tile's first barrier
        t = tn;
    }
}

}  // namespace

namespace spsp {

bool conv_wres_supported(const ConvArgs& a) {
    return a.stride == 1 && a.kw_steps == 3 && a.chunks == 1 && a.nsteps == 9 && a.N == 64 && a.in_pix == 64 && a.in_off == 0 &&
           a.M % 256 == 0 && a.Ho == a.Wo && a.Wo % 16 == 0 && a.splitk <= 1 && !a.pool_out && a.out && a.counter;
}

int launch_conv_wres(const ConvArgs& a, int num_cus, hipStream_t st) {
    SPSP_REQUIRE(conv_wres_supported(a), "conv_wres: needs a 3x3 stride-1 64->64 convolution on a map that is a multiple of 16 wide");
    SPSP_TRY(ensure_dynamic_lds<&conv3x3_wres_kernel<false>>(kLds));
    SPSP_TRY(ensure_dynamic_lds<&conv3x3_wres_kernel<true>>(kLds));
    const int ntiles = a.M / 256;
    const int grid = ntiles < num_cus ? ntiles : num_cus;       // persistent: one workgroup per CU (147 KB of LDS each)
#ifdef SPSP_CONV_STAMPS
    if (getenv("SPSP_CONV_DEBUG")) {
        ConvArgs b = a;
        (void)hipMalloc(&b.dbg, 16 * sizeof(unsigned long long));
        conv3x3_wres_kernel<false><<<dim3(grid), 256, kLds, st>>>(b);
        (void)hipDeviceSynchronize();
        unsigned long long h[16];
        (void)hipMemcpy(h, b.dbg, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "wres grid=%d tiles=%d: weights+patch landed %llu | K loop %llu | epilogue %llu | tile2 wait %llu | K loop %llu | epilogue %llu | total %llu\n",
                grid, ntiles, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[5] - h[3], h[6] - h[5], h[7] - h[6], h[7] - h[0]);
        (void)hipFree(b.dbg);
        return SPSP_OK;
    }
#endif
    if (a.f16) conv3x3_wres_kernel<true><<<dim3(grid), 256, kLds, st>>>(a);
    else conv3x3_wres_kernel<false><<<dim3(grid), 256, kLds, st>>>(a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp
