// conv1 (7x7 / stride 2 / pad 3, 18 -> 64 channels, 256x256 -> 128x128) with an LDS-RESIDENT INPUT PATCH.
//
// The first-generation conv1 (conv.hip, <128,64,128,8>) re-gathers every input row 3.5x (7 kernel rows / stride
// 2) and every pixel 3.5x within a row: 940 MB of L2 -> LDS traffic per 32 crops for 79 MB of input; it is bound
// by that path (262 TFLOP/s, 10 % of the MFMA peak).  Here a workgroup copies the 13 input rows that 4 output
// rows need into LDS ONCE -- one contiguous 121 KB block of the native layout, so the LDS-DMA is a linear copy
// -- and reads every MFMA pixel fragment straight from it:
//   K-step = kernel row r (7 steps); the 7 taps x 18 channels of an output pixel are 126 contiguous bf16 of the
//   input row (padded to 128 with two zero weights), starting 72*ox bytes into the row -> 8-byte aligned, so a
//   16-byte fragment is two ds_read_b64.
// Workgroup: 4 output rows x 128 pixels x 64 channels, 8 waves (two per SIMD), wave = (row, 64-pixel half),
// wave tile 64 px x 64 ch on v_mfma_f32_16x16x32_bf16; the [64 x 128] weight slab of each kernel row streams
// through a 2-stage LDS-DMA ring.  All fragment addresses are ONE per-lane base + immediates.
#include <cstdlib>
#include <type_traits>

#include "conv.hpp"
#include "conv_elem.hpp"

namespace {
using namespace spsp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_b128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// 16-byte fragment from an 8-byte aligned address: two ds_read_b64 into the halves of one 128-bit register group
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_2xb64(uint32_t addr) {
    bf16x4 lo, hi;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 8) : "memory");
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void dma16(const uint16_t* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int kRowB = 264 * 18 * 2;                 // 9504 bytes per padded input row
constexpr int kPatchRows = 13;                      // 4 output rows: input rows 2*oy0 .. 2*oy0 + 12
constexpr int kPatchPieces = (kPatchRows * kRowB + 1023) / 1024;   // 121 x 1 KiB
constexpr int kPatchB = kPatchPieces * 1024;
constexpr int kWStage = 64 * 256;                   // [64 n][128 k] bf16
constexpr int kLds = kPatchB + 2 * kWStage;

template <bool F16>
__global__ __launch_bounds__(512) void conv1_patch_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const ring = smem + kPatchB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = gridDim.x;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
    const int b = tile >> 5, oy0 = (tile & 31) * 4;            // 32 row-quads per image

    const uint16_t* const in_ptr = a.in;
    const uint16_t* const w_ptr = a.w;
    // ---- patch: linear copy of 13 contiguous rows ----
    const uint16_t* psrc = in_ptr + (size_t)b * a.in_img + (size_t)(2 * oy0) * a.in_row + lane * 8;
#pragma unroll
    for (int i = 0; i < (kPatchPieces + 7) / 8; ++i) {
        const int p = wave + 8 * i;
        if (p < kPatchPieces) dma16(psrc + (size_t)p * 512, smem + p * 1024);
    }
    // ---- weight ring: stage r = [64 n][128 k]; piece = 4 rows x 256 B; lane l -> row l/16, physical chunk l%16 ----
    const int drow = lane >> 4, dchunk = lane & 15;
    const uint16_t* w_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 4 + drow;
        w_src[i] = w_ptr + (size_t)row * (7 * 128) + ((dchunk ^ (row & 15)) * 8);
    }
    auto issue_w = [&](int r, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(w_src[i] + r * 128, ring + buf * kWStage + (wave * 2 + i) * 1024);
    };
    issue_w(0, 0);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const int ry = wave >> 1, half = wave & 1;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    // weights: row i*16 + frow, chunk (ks*4 + fq) ^ frow  ->  base ^ (ks * 64)
    const uint32_t w_lane = frow * 256 + ((fq ^ frow) * 16);
    // pixels: input row 2*ry + r, byte 72*ox + 16*(ks*4 + fq), ox = half*64 + j*16 + frow
    const uint32_t x_lane = smem_base + (2 * ry) * kRowB + (half * 64 + frow) * 72 + fq * 16;

#pragma unroll
    for (int r = 0; r < 7; ++r) {
        wait_vmcnt<0>();                                   // patch (first step) + weight stage r
        __builtin_amdgcn_s_barrier();
        if (r + 1 < 7) issue_w(r + 1, (r + 1) & 1);
        const uint32_t wbase = smem_base + kPatchB + (r & 1) * kWStage;
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {                   // two 32-deep k-substeps at a time
            bf16x8 wf[2][4], xf[2][4];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ks = kp * 2 + kk;
                const uint32_t wa = wbase + (w_lane ^ (ks * 64));
                wf[kk][0] = lds_read_b128<0>(wa);
                wf[kk][1] = lds_read_b128<4096>(wa);
                wf[kk][2] = lds_read_b128<8192>(wa);
                wf[kk][3] = lds_read_b128<12288>(wa);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                constexpr int J = 16 * 72;
                const uint32_t xa = x_lane + r * kRowB + (kp * 2 + kk) * 64;
                xf[kk][0] = lds_read_2xb64<0>(xa);
                xf[kk][1] = lds_read_2xb64<J>(xa);
                xf[kk][2] = lds_read_2xb64<2 * J>(xa);
                xf[kk][3] = lds_read_2xb64<3 * J>(xa);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = Elem<F16>::mfma(wf[kk][i], xf[kk][j], acc[i][j]);
        }
    }

    // ---- epilogue: bias + ReLU -> bf16, staged through LDS so every global store is a whole 128-byte line
    //      (a lane owns 4 channels = 8 B of a pixel; stored directly a wave-instruction touches a quarter of 16 lines).
    //      Output tile [512 px][128 B], 16-byte chunk XOR-swizzled by the pixel; tile pixel tp = wave*64 + j*16 + frow
    //      = (row tp / 128, column tp % 128).
    __syncthreads();                                       // patch and ring are dead
    unsigned char* const outT = smem;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tp = wave * 64 + j * 16 + frow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = i * 16 + fq * 4;
            const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
            const f32x4 c = acc[i][j];
            float v0 = c[0] + bv.x, v1 = c[1] + bv.y, v2 = c[2] + bv.z, v3 = c[3] + bv.w;
            if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            *reinterpret_cast<uint2*>(outT + tp * 128 + (((i * 2 + (fq >> 1)) ^ (tp & 7)) * 16) + (fq & 1) * 8) =
                make_uint2(Elem<F16>::pack2(v0, v1), Elem<F16>::pack2(v2, v3));
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {                       // 512 px x 8 chunks / 512 threads
        const int idx = tid + it * 512, tp = idx >> 3, c = idx & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(outT + tp * 128 + ((c ^ (tp & 7)) * 16));
        const size_t o = (size_t)a.out_off + (size_t)b * a.out_img + (size_t)(oy0 + (tp >> 7)) * a.out_row + (size_t)(tp & 127) * a.out_pix;
        *reinterpret_cast<uint4*>(a.out + o + c * 8) = v;
    }
}


// ---- second variant: HALF-WIDTH tiles, two workgroups per CU ------------------------------------------------------
// The kernel above owns its CU alone (153 KB of LDS): its 121 KB patch load, its K loop and its 64 KB of output stores
// run one after the other, four times per CU.  Here a workgroup takes 4 output rows x 64 pixels x 64 channels with
// 4 waves (one output row each): the 13 x 133-pixel input patch is 61 KB (rows are gathered with per-lane source
// addresses; the LDS image keeps the 36-byte pixel pitch, row pitch 4800 B), the weights stream as fourteen
// [64 x 64] half-row slabs through a 2 x 8 KB ring, 77 KB in all -- so TWO workgroups share a CU and one loads or
// stores while the other multiplies (2048 workgroups = 8 per CU).  Fragment reads are software-pipelined one
// half-step ahead of the MFMAs (two register sets), as in conv_patch.hip.
constexpr int kHRowB = 4800;                               // 133.3 px x 36 B, a multiple of 16
constexpr int kHPieces = (kPatchRows * kHRowB + 1023) / 1024;   // 61
constexpr int kHPatchB = kHPieces * 1024;                  // 62 464
constexpr int kHWStage = 64 * 128;                         // [64 n][64 k]
constexpr int kHLds = kHPatchB + 2 * kHWStage;             // 78 848

template <bool F16>
__global__ __launch_bounds__(256) void conv1_half_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const ring = smem + kHPatchB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = gridDim.x;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
    const int b = tile >> 6, oy0 = ((tile >> 1) & 31) * 4, ox0 = (tile & 1) * 64;
    const uint16_t* const w_ptr = a.w;

    // ---- patch: 13 rows x 4800 B; byte o of the LDS image comes from input row o / 4800, byte o % 4800 of the run
    //      that starts at input pixel 2*ox0 (4608 B into the row for the right half: 16-byte aligned) ----
    const unsigned char* const pbase = reinterpret_cast<const unsigned char*>(a.in) +
                                       ((size_t)b * a.in_img + (size_t)(2 * oy0) * a.in_row) * 2 + (size_t)(2 * ox0) * 36;
#pragma unroll
    for (int i = 0; i < (kHPieces + 3) / 4; ++i) {
        const int p = wave + 4 * i;
        if (p < kHPieces) {
            const int o = (p * 64 + lane) * 16, row = o / kHRowB, col = o - row * kHRowB;
            dma16(reinterpret_cast<const uint16_t*>(pbase + (size_t)row * kRowB + col), smem + p * 1024);
        }
    }
    // ---- weight ring: stage g = K-elements [64 g, 64 g + 64) of all 64 rows; piece = 8 rows x 128 B ----
    const int drow = lane >> 3, dchunk = lane & 7;
    const uint16_t* w_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + drow;
        w_src[i] = w_ptr + (size_t)row * (7 * 128) + ((dchunk ^ (row & 7)) * 8);
    }
    auto issue_w = [&](int g, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(w_src[i] + g * 64, ring + buf * kHWStage + (wave * 2 + i) * 1024);
    };
    issue_w(0, 0);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    const uint32_t w_lane = frow * 128 + ((fq ^ (frow & 7)) * 16);            // ks = 1: ^ 64
    const uint32_t x_lane = smem_base + (2 * wave) * kHRowB + frow * 72 + fq * 16;

    bf16x8 wfr[2][2][4], xfr[2][2][4];
#define SPSP_C1_READ(G, SET)                                                                              \
    {                                                                                                     \
        const uint32_t wbase_ = smem_base + kHPatchB + ((G) & 1) * kHWStage;                              \
        const uint32_t xa_ = x_lane + ((G) >> 1) * kHRowB + ((G) & 1) * 128;                              \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                \
            const uint32_t wa_ = wbase_ + (w_lane ^ (kk * 64));                                           \
            wfr[SET][kk][0] = lds_read_b128<0>(wa_);                                                      \
            wfr[SET][kk][1] = lds_read_b128<2048>(wa_);                                                   \
            wfr[SET][kk][2] = lds_read_b128<4096>(wa_);                                                   \
            wfr[SET][kk][3] = lds_read_b128<6144>(wa_);                                                   \
            xfr[SET][kk][0] = lds_read_2xb64<0>(xa_ + kk * 64);                                           \
            xfr[SET][kk][1] = lds_read_2xb64<16 * 72>(xa_ + kk * 64);                                     \
            xfr[SET][kk][2] = lds_read_2xb64<32 * 72>(xa_ + kk * 64);                                     \
            xfr[SET][kk][3] = lds_read_2xb64<48 * 72>(xa_ + kk * 64);                                     \
        }                                                                                                 \
    }
#define SPSP_C1_MFMA(SET)                                                                                 \
    {                                                                                                     \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                  \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                 \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
                    acc[i][j] = Elem<F16>::mfma(wfr[SET][kk][i], xfr[SET][kk][j], acc[i][j]);             \
    }
#define SPSP_C1_STEP(G, SET)                                                                              \
    {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    /* fragments of step G-1 are in registers */ \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        wait_vmcnt<0>();                                       /* stage G (and, at G = 0, the patch) landed */ \
        __builtin_amdgcn_s_barrier();                                                                     \
        if ((G) + 1 < 14) issue_w((G) + 1, ((G) + 1) & 1);     /* into the slot step G-1 was read from */  \
        SPSP_C1_READ(G, SET)                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        if ((G) > 0) SPSP_C1_MFMA((SET) ^ 1)                                                              \
    }
    SPSP_C1_STEP(0, 0) SPSP_C1_STEP(1, 1) SPSP_C1_STEP(2, 0) SPSP_C1_STEP(3, 1) SPSP_C1_STEP(4, 0) SPSP_C1_STEP(5, 1)
    SPSP_C1_STEP(6, 0) SPSP_C1_STEP(7, 1) SPSP_C1_STEP(8, 0) SPSP_C1_STEP(9, 1) SPSP_C1_STEP(10, 0) SPSP_C1_STEP(11, 1)
    SPSP_C1_STEP(12, 0) SPSP_C1_STEP(13, 1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    SPSP_C1_MFMA(1)
#undef SPSP_C1_STEP
#undef SPSP_C1_MFMA
#undef SPSP_C1_READ

    // ---- epilogue: bias + ReLU, staged through LDS (tile [256 px][128 B], chunk ^= pixel & 7), whole-line stores ----
    float4 bias_r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias_r[i] = *reinterpret_cast<const float4*>(a.bias + i * 16 + fq * 4);
    __syncthreads();                                       // patch and ring are dead
    unsigned char* const outT = smem;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tp = wave * 64 + j * 16 + frow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 bv = bias_r[i];
            const f32x4 c = acc[i][j];
            float v0 = c[0] + bv.x, v1 = c[1] + bv.y, v2 = c[2] + bv.z, v3 = c[3] + bv.w;
            if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            *reinterpret_cast<uint2*>(outT + tp * 128 + (((i * 2 + (fq >> 1)) ^ (tp & 7)) * 16) + (fq & 1) * 8) =
                make_uint2(Elem<F16>::pack2(v0, v1), Elem<F16>::pack2(v2, v3));
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {                       // 256 px x 8 chunks / 256 threads
        const int idx = tid + it * 256, tp = idx >> 3, c = idx & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(outT + tp * 128 + ((c ^ (tp & 7)) * 16));
        const size_t o = (size_t)a.out_off + (size_t)b * a.out_img + (size_t)(oy0 + (tp >> 6)) * a.out_row +
                         (size_t)(ox0 + (tp & 63)) * a.out_pix;
        *reinterpret_cast<uint4*>(a.out + o + c * 8) = v;
    }
}

}  // namespace

namespace spsp {

bool conv1_uses_half_tiles() { return true; }

int launch_conv1_patch(const ConvArgs& a, hipStream_t st) {
    SPSP_REQUIRE(a.N == 64 && a.Ho == 128 && a.Wo == 128 && a.nsteps == 7 && a.stride == 2 && a.in_off == 0 &&
                     a.in_pix == 18 && a.in_row == 264 * 18,
                 "conv1_patch: geometry is fixed to the 7x7/2 stem on the [262,264,18] native layout");
    SPSP_TRY(ensure_dynamic_lds<&conv1_patch_kernel<false>>(kLds));
    SPSP_TRY(ensure_dynamic_lds<&conv1_patch_kernel<true>>(kLds));
    const int B = a.M / (128 * 128);
    if (conv1_uses_half_tiles()) {
        SPSP_TRY(ensure_dynamic_lds<&conv1_half_kernel<false>>(kHLds));
        SPSP_TRY(ensure_dynamic_lds<&conv1_half_kernel<true>>(kHLds));
        if (a.f16) conv1_half_kernel<true><<<dim3(B * 64), 256, kHLds, st>>>(a);
        else conv1_half_kernel<false><<<dim3(B * 64), 256, kHLds, st>>>(a);
        SPSP_HIP_CHECK(hipGetLastError());
        return SPSP_OK;
    }
    if (a.f16) conv1_patch_kernel<true><<<dim3(B * 32), 512, kLds, st>>>(a);
    else conv1_patch_kernel<false><<<dim3(B * 32), 512, kLds, st>>>(a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp
