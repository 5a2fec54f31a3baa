// layer2.0's 3x3 / STRIDE-2 convolution + its 1x1 / stride-2 downsample (PR/models/resnet.py:177-199: 64 -> 128 channels, 64 x 64 ->
// 32 x 32) with ALL 128 OUTPUT CHANNELS IN ONE WORKGROUP and the weights stationary in registers -- round 5.
//
// conv_s2rs.hip (round 4) keeps a 64-channel block's weights in registers, so the four parity planes of a tile are staged TWICE (once
// per block: 37 MB per launch where 19 MB would do), a wave takes one 32-deep half of every K-step, and the halves meet through LDS at
// the end of a tile (two barriers + 64 KB of accumulator traffic).  Here
//   * wave q owns output channels 32 q .. 32 q + 31 over the whole K of both convolutions: 2 channel tiles x 20 K-steps (nine taps x two
//     32-channel halves + the downsample's two) = 40 MFMA A-fragments = 160 registers, read straight from the accumulation file by
//     inline-asm MFMAs (conv_elem.hpp); the host packs them in register order: a wave's prologue is 40 coalesced 1-KiB loads;
//   * a workgroup walks tiles of 4 x 16 output pixels (two tile buffers: the next tile's planes land under this tile's MFMAs); every
//     wave reads all four pixel rows: a K-step = 4 fragment reads beside 8 MFMAs, no K split, no exchange, ONE barrier per tile;
//   * the planes with 17 columns are stored with a pitch of 24 pixels (those with 16: 16), so the bank swizzle's key -- pixel & 7 -- is
//     the column's low bits on every row: a fragment address is `per-lane base [dx][kh] + immediate`, four address registers in all;
//   * the downsample runs after the main convolution's epilogue on the same accumulators (its two K-steps read the centre tap's pixels).
// tests/emul/conv_s2q_model.py mirrors the fragment packing, the LDS fill, the addresses and the epilogue map.
#include <type_traits>

#include "conv.hpp"
#include "conv_elem.hpp"

namespace {
using namespace spsp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_relu16(uint32_t x) {      // max(x, +0) on both 16-bit floats of the pair (conv_fb.hip)
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, x), i16x2{0, 0}));
}

constexpr int kPixB = 128;
// planes in LDS order: (1,1) 4 rows x pitch 16 | (1,0) 4 x 24 | (0,1) 5 x 16 | (0,0) 5 x 24   (17 columns of the 24 are read)
__host__ __device__ constexpr int q_plane(int ky, int kx) { return (ky & 1) ? ((kx & 1) ? 0 : 1) : ((kx & 1) ? 2 : 3); }
__host__ __device__ constexpr int q_pitch(int p) { return (p & 1) ? 24 : 16; }
__host__ __device__ constexpr int q_base(int p) { return p == 0 ? 0 : (p == 1 ? 64 : (p == 2 ? 160 : 240)); }
constexpr int kTilePix = 360, kPieces = 45, kPiecesPerWave = 12;
constexpr int kTileB = kTilePix * kPixB;                 // 46 080
constexpr int kLds = 2 * kTileB;                         // 92 160 B
constexpr int kSteps = 20;

struct S2qArgs {
    const uint16_t* in;      // [B][66][66][64], zero border
    const uint16_t* wf;      // [q 4][K-step 20][tile 2][lane 64][8] (pack_s2q_fragments)
    const float* bias;       // [128] main convolution
    const float* bias2;      // [128] downsample
    uint16_t* out;           // [B][34][34][128]: relu(conv3x3s2 + bias), interior
    uint16_t* out2;          // [B][34][34][128]: conv1x1s2 + bias2
    int ntiles;              // 16 per crop
    int tiles_per_wg;
    int relu;
    unsigned long long* dbg;
};

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

// K-step S: taps 0..8 (S = 2 tap + kh), then the downsample (S = 18, 19: the centre tap's pixels)
__host__ __device__ constexpr int q_ky(int S) { return S < 18 ? (S >> 1) / 3 : 1; }
__host__ __device__ constexpr int q_kx(int S) { return S < 18 ? (S >> 1) % 3 : 1; }
// fragment of output row J: plane pixel (J + (ky >> 1), frow + (kx >> 1))
template <int S, int J>
__device__ __forceinline__ void q_read(bf16x8 (&xf)[2][4], const uint32_t (&ad)[2][2]) {
    constexpr int ky = q_ky(S), kx = q_kx(S), p = q_plane(ky, kx);
    constexpr int imm = (q_base(p) + (J + (ky >> 1)) * q_pitch(p)) * kPixB;
    xf[S & 1][J] = lds_read_b128<imm>(ad[kx >> 1][S & 1]);
}
// K-step S: its 8 MFMAs with the four fragment reads of step S + 1 beside them (conv_fb.hip's scheme: a pair waits only for ITS
// fragment, issued a step ago, with a counted lgkmcnt).  FIRST: the accumulation chains start here, from the bias registers.
template <bool F16, int S, bool FIRST, bool NEXT>
__device__ __forceinline__ void q_step(f32x4 (&acc)[2][4], bf16x8 (&xf)[2][4], const bf16x8 (&w)[kSteps][2], const uint32_t (&ad)[2][2],
                                       const f32x4 (&bias)[2]) {
    constexpr int set = S & 1;
#define Q_PAIR(J, CNT)                                                                                   \
    if constexpr (NEXT) q_read<S + 1, J>(xf, ad);                                                        \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NEXT ? 4 : CNT) : "memory");                              \
    if constexpr (FIRST) {                                                                               \
        Elem<F16>::mfma_agpr_from(acc[0][J], w[S][0], xf[set][J], bias[0]);                              \
        Elem<F16>::mfma_agpr_from(acc[1][J], w[S][1], xf[set][J], bias[1]);                              \
    } else {                                                                                             \
        Elem<F16>::mfma_agpr(acc[0][J], w[S][0], xf[set][J]);                                            \
        Elem<F16>::mfma_agpr(acc[1][J], w[S][1], xf[set][J]);                                            \
    }
    Q_PAIR(0, 3) Q_PAIR(1, 2) Q_PAIR(2, 1) Q_PAIR(3, 0)
#undef Q_PAIR
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef SPSP_CONV_STAMPS
#define QSTAMP(i) if (a.dbg && blockIdx.x == 8 && tid == 0 && ntile_done == 1) a.dbg[i] = __builtin_amdgcn_s_memtime()
#define QSTAMPK(i) if (a.dbg && blockIdx.x == 8 && tid == 0) a.dbg[i] = __builtin_amdgcn_s_memtime()
#else
#define QSTAMP(i)
#define QSTAMPK(i)
#endif

template <bool F16>
__global__ __launch_bounds__(256, 1) void conv3x3s2q_kernel(const S2qArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);       // this wave's channel quarter
    const int frow = lane & 15, fq = lane >> 4;
    QSTAMPK(0);
    // Workgroups b, b + 8, ... share an XCD (round-robin dispatch; speed only): consecutive tile ranges per XCD
    const int G = (int)gridDim.x;
    int L = (int)blockIdx.x;
    if ((G & 7) == 0) L = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    int t = L * a.tiles_per_wg;
    const int t_end = t + a.tiles_per_wg < a.ntiles ? t + a.tiles_per_wg : a.ntiles;
    if (t >= t_end) return;
    int ntile_done = 0;

    // ---- LDS-DMA of a tile: piece j = tile-image pixels 8 j .. 8 j + 7; lane l feeds pixel p = 8 j + l / 8 (plane, row, column by the
    //      plane's pitch), physical chunk l % 8 = logical chunk (l % 8) ^ (p & 7); source = buffer pixel (2 (y0 + row) + py,
    //      2 (x0 + col) + px), the column clamped for the pad columns of the pitch-24 planes (never read)
    const int drow = lane >> 3, dchunk = lane & 7;
    auto issue_piece = [&](int i, int img, int y0, int x0, int buf) {
        const int j = q + 4 * i;
        if (j >= kPieces) return;
        const int p = 8 * j + drow;
        const int pl = p < q_base(1) ? 0 : (p < q_base(2) ? 1 : (p < q_base(3) ? 2 : 3));
        const int pq = p - (pl == 0 ? q_base(0) : (pl == 1 ? q_base(1) : (pl == 2 ? q_base(2) : q_base(3))));
        const int wide = pl & 1;                                   // pitch 24
        const int row = wide ? (pq * 171) >> 12 : pq >> 4;         // pq / 24 (pq < 120), pq / 16
        const int col = wide ? pq - 24 * row : pq & 15;
        const int brow = 2 * (y0 + row) + (pl < 2 ? 1 : 0);
        int bcol = 2 * (x0 + col) + ((pl & 1) ? 0 : 1);
        bcol = bcol > 65 ? 65 : bcol;
        dma16(a.in + ((size_t)(img * 66 + brow) * 66 + bcol) * 64 + ((dchunk ^ (p & 7)) * 8), smem + buf * kTileB + j * 1024);
    };
    // ---- prologue: this wave's 40 weight fragments (resident for the whole kernel) and the first tile's planes
    bf16x8 w[kSteps][2];
    {
        const bf16x8* const wsrc = reinterpret_cast<const bf16x8*>(a.wf) + (size_t)(q * kSteps * 2) * 64 + lane;
#pragma unroll
        for (int s = 0; s < kSteps; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) w[s][i] = wsrc[(s * 2 + i) * 64];
    }
    {
        const int img = t >> 4, r = t & 15;
#pragma unroll
        for (int i = 0; i < kPiecesPerWave; ++i) issue_piece(i, img, (r >> 1) * 4, (r & 1) * 16, 0);
    }
    // accumulator element e of tile (i, j) = channel 32 q + 8 fq + 4 i + e
    f32x4 bias_m[2], bias_d[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 u = *reinterpret_cast<const float4*>(a.bias + 32 * q + 8 * fq + 4 * i);
        const float4 v = *reinterpret_cast<const float4*>(a.bias2 + 32 * q + 8 * fq + 4 * i);
        bias_m[i] = f32x4{u.x, u.y, u.z, u.w};
        bias_d[i] = f32x4{v.x, v.y, v.z, v.w};
    }
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    uint32_t a0[2][2];                 // pixel (frow + dx) of a plane row, logical chunk kh * 4 + fq
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
        a0[dx][0] = (uint32_t)((frow + dx) * kPixB + ((fq ^ ((frow + dx) & 7)) * 16));
        a0[dx][1] = a0[dx][0] ^ 64u;
    }
    const int lane_out = (34 + frow + 1) * 128 + 32 * q + 8 * fq;      // (elements; + the tile's origin: row y0 + 1, column x0 + 1)

    int cur = 0;
    for (;; cur ^= 1) {
        const int img = t >> 4, r = t & 15;
        const int y0 = (r >> 1) * 4, x0 = (r & 1) * 16;
        // this tile's planes: this wave's pieces have landed (older than the eight stores of the previous tile), then everybody's; every
        // wave has left the other buffer
        QSTAMP(1);
        if (ntile_done == 0) wait_vmcnt<0>(); else wait_vmcnt<8>();
        __builtin_amdgcn_s_barrier();
        QSTAMP(2);
        const bool has_next = t + 1 < t_end;
        const int nimg = (t + 1) >> 4, nr = (t + 1) & 15, ny0 = (nr >> 1) * 4, nx0 = (nr & 1) * 16;
        uint32_t ad[2][2];
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            ad[dx][0] = smem_base + cur * kTileB + a0[dx][0];
            ad[dx][1] = smem_base + cur * kTileB + a0[dx][1];
        }
        f32x4 acc[2][4];
        bf16x8 xf[2][4];
        uint16_t* const out_t = a.out + ((size_t)(img * 34 + y0) * 34 + x0) * 128 + lane_out;
        uint16_t* const out2_t = a.out2 + ((size_t)(img * 34 + y0) * 34 + x0) * 128 + lane_out;

        // ---- main convolution: 18 K-steps; the next tile's twelve LDS-DMA pieces ride behind the first twelve
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (counted waits below: nothing else may be in flight)
        q_read<0, 0>(xf, ad); q_read<0, 1>(xf, ad); q_read<0, 2>(xf, ad); q_read<0, 3>(xf, ad);
        __builtin_amdgcn_sched_barrier(0);
#define Q_S(S, FIRST, NEXT, B) q_step<F16, S, FIRST, NEXT>(acc, xf, w, ad, B);
#define Q_S_DMA(S, FIRST) Q_S(S, FIRST, true, bias_m) if (has_next) issue_piece(S, nimg, ny0, nx0, cur ^ 1);
        Q_S_DMA(0, true) Q_S_DMA(1, false) Q_S_DMA(2, false) Q_S_DMA(3, false) Q_S_DMA(4, false) Q_S_DMA(5, false)
        Q_S_DMA(6, false) Q_S_DMA(7, false) Q_S_DMA(8, false) Q_S_DMA(9, false) Q_S_DMA(10, false) Q_S_DMA(11, false)
        Q_S(12, false, true, bias_m) Q_S(13, false, true, bias_m) Q_S(14, false, true, bias_m) Q_S(15, false, true, bias_m)
        Q_S(16, false, true, bias_m)
        Q_S(17, false, true, bias_m)                             // (its reads: the downsample's first fragments)
        QSTAMP(3);
        // accumulators (the bias went in with the first K-step) -> ReLU -> 16-bit -> 16-byte stores: 8 consecutive channels of a pixel
        mfma_drain8(acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 lo = acc[0][j], hi = acc[1][j];
            uint32_t p0 = Elem<F16>::pack2(lo[0], lo[1]), p1 = Elem<F16>::pack2(lo[2], lo[3]);
            uint32_t p2 = Elem<F16>::pack2(hi[0], hi[1]), p3 = Elem<F16>::pack2(hi[2], hi[3]);
            if (a.relu) { p0 = pk_relu16(p0); p1 = pk_relu16(p1); p2 = pk_relu16(p2); p3 = pk_relu16(p3); }
            *reinterpret_cast<uint4*>(out_t + (size_t)j * (34 * 128)) = make_uint4(p0, p1, p2, p3);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the downsample on the same accumulators: two K-steps on the centre tap's pixels, no ReLU
        Q_S(18, true, true, bias_d)
        Q_S(19, false, false, bias_d)
        mfma_drain8(acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 lo = acc[0][j], hi = acc[1][j];
            *reinterpret_cast<uint4*>(out2_t + (size_t)j * (34 * 128)) =
                make_uint4(Elem<F16>::pack2(lo[0], lo[1]), Elem<F16>::pack2(lo[2], lo[3]), Elem<F16>::pack2(hi[0], hi[1]), Elem<F16>::pack2(hi[2], hi[3]));
        }
#undef Q_S_DMA
#undef Q_S
        QSTAMP(4);
        ++ntile_done;
        if (!has_next) break;
        ++t;
    }
    QSTAMPK(5);
}

}  // namespace

namespace spsp {

// w [128][576] (K = tap-major, channel-minor; BN folded), w2 [128][64] -> conv3x3s2q_kernel's register-order fragment image
// [q][K-step s][tile i][lane][8]: MFMA row rho of tile i = output channel 32 q + 8 (rho >> 2) + 4 i + (rho & 3); lane (rho, fq) holds
// k = tap * 64 + kh * 32 + 8 fq .. + 7 (s = 2 tap + kh < 18) or, for the downsample (s = 18 + kh), kh * 32 + 8 fq .. + 7
void pack_s2q_fragments(const uint16_t* w, const uint16_t* w2, std::vector<uint16_t>& out) {
    out.assign((size_t)4 * kSteps * 2 * 64 * 8, 0);
    for (int q = 0; q < 4; ++q)
        for (int s = 0; s < kSteps; ++s)
            for (int i = 0; i < 2; ++i)
                for (int lane = 0; lane < 64; ++lane) {
                    const int rho = lane & 15, fq = lane >> 4;
                    const int n = 32 * q + 8 * (rho >> 2) + 4 * i + (rho & 3);
                    uint16_t* d = &out[((((size_t)q * kSteps + s) * 2 + i) * 64 + lane) * 8];
                    const uint16_t* src = s < 18 ? w + (size_t)n * 576 + (s >> 1) * 64 + (s & 1) * 32 + 8 * fq
                                                 : w2 + (size_t)n * 64 + (s & 1) * 32 + 8 * fq;
                    for (int e = 0; e < 8; ++e) d[e] = src[e];
                }
}

// in [B][66][66][64]; out / out2 [B][34][34][128]; wf: pack_s2q_fragments' image on the device
int launch_conv_s2q(const uint16_t* in, const uint16_t* wf, const float* bias, const float* bias2, uint16_t* out, uint16_t* out2, int B,
                    int f16, int relu, int max_workgroups, hipStream_t st) {
    SPSP_REQUIRE(in && wf && bias && bias2 && out && out2 && B > 0 && max_workgroups > 0, "conv_s2q: bad arguments");
    S2qArgs a;
    a.in = in; a.wf = wf; a.bias = bias; a.bias2 = bias2; a.out = out; a.out2 = out2; a.relu = relu;
    a.ntiles = 16 * B;
    a.tiles_per_wg = (a.ntiles + max_workgroups - 1) / max_workgroups;
    const int grid = (a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    a.dbg = nullptr;
    SPSP_TRY((ensure_dynamic_lds<&conv3x3s2q_kernel<true>>(kLds)));
    SPSP_TRY((ensure_dynamic_lds<&conv3x3s2q_kernel<false>>(kLds)));
#ifdef SPSP_CONV_STAMPS
    if (getenv("SPSP_CONV_DEBUG")) {
        (void)hipMalloc(&a.dbg, 16 * sizeof(unsigned long long));
        (void)hipMemset(a.dbg, 0, 16 * sizeof(unsigned long long));
        if (f16) conv3x3s2q_kernel<true><<<dim3(grid), 256, kLds, st>>>(a); else conv3x3s2q_kernel<false><<<dim3(grid), 256, kLds, st>>>(a);
        (void)hipDeviceSynchronize();
        unsigned long long h[16];
        (void)hipMemcpy(h, a.dbg, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "conv_s2q grid=%d tiles/wg=%d: prologue + tile 1 %llu | tile 2: wait+barrier %llu | main K loop %llu | epilogue + downsample %llu | whole kernel %llu cycles\n",
                grid, a.tiles_per_wg, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[0]);
        (void)hipFree(a.dbg);
        return SPSP_OK;
    }
#endif
    if (f16) launch_kernel(&conv3x3s2q_kernel<true>, dim3(grid), dim3(256), kLds, st, a);
    else launch_kernel(&conv3x3s2q_kernel<false>, dim3(grid), dim3(256), kLds, st, a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp
