// layer2's 3x3 / stride-1 convolutions (PR/models/resnet.py:61-77 inside layer2: 128 -> 128 channels on 32 x 32 maps) with a 64-channel
// block's WEIGHTS STATIONARY IN REGISTERS -- round 5, plan "throughput".
//
// conv_p4.hip streams the nine [64 x 128] weight slabs of a block through an LDS ring for every 256-pixel tile and lives for one tile:
// of ~19 k cycles per tile 9.2 k are MFMAs (prologue 4.5-5 k: first patch + first slab; group barriers; a tail).  With several batches
// in flight what a launch costs is its CU time, so here a workgroup
//   * keeps its block's weights as MFMA A-fragments for ALL its tiles: wave (ch, ph) owns output channels 64 nb + 32 ch .. + 31 over the
//     whole K = 1152 = 72 fragments = 288 registers, 248 of them in the accumulation file, read there by inline-asm MFMAs
//     (conv_elem.hpp; conv_fb.hip's scheme) -- the host packs them in register order: 72 coalesced 1-KiB loads per wave, once;
//   * walks tiles of 8 x 16 output pixels: the 10 x 18 input patch as two 64-channel sub-images (46 LDS-DMA pieces), two buffers -- the
//     next tile's patch lands under this tile's MFMAs; LDS carries pixels only; ONE barrier per tile, no ring, no group barriers;
//   * a fragment address is `per-lane base [delta][kh] + immediate`, delta = (2 (row + ky) + kx) & 7 (the patch pitch is 18 pixels: the
//     bank swizzle's key, pixel & 7, moves by 2 per row): sixteen address registers.
// A K-step = 4 pixel-fragment reads beside 8 MFMAs, fragment reads a whole step ahead with counted lgkmcnt waits (conv_fb.hip).
// The prologue (288 KB of fragment loads per workgroup) pays from about four tiles per workgroup on: plan "throughput" launches 64
// workgroups (eight tiles each at B = 32); plan "latency" keeps conv_p4.hip.  tests/emul/conv_rs_model.py mirrors the index math.
#include <type_traits>

#include "conv.hpp"
#include "conv_elem.hpp"

namespace {
using namespace spsp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pk_relu16(uint32_t x) {      // max(x, +0) on both 16-bit floats of the pair (conv_fb.hip)
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, x), i16x2{0, 0}));
}

constexpr int kPixB = 128, kPitch = 18;
constexpr int kChunkPieces = 23;                         // 10 x 18 = 180 pixels of 64 channels + 4 pad pixels
constexpr int kChunkB = kChunkPieces * 1024;             // 23 552
constexpr int kTileB = 2 * kChunkB;                      // 47 104
constexpr int kLds = 2 * kTileB;                         // 94 208 B
constexpr int kPieces = 2 * kChunkPieces, kPiecesPerWave = 12;
constexpr int kSteps = 36;                               // K-step s = 4 tap + 2 chunk + kh
constexpr int kAgprSteps = 31;                           // the fragments of K-steps 0..30 in AGPRs (248 registers), the last five in VGPRs

struct RsArgs {
    const uint16_t* in;      // [B][34][34][128], zero border
    const uint16_t* wf;      // [nb 2][ch 2][K-step 36][tile 2][lane 64][8] (pack_rs_fragments)
    const float* bias;       // [128]
    const uint16_t* res;     // residual, laid out like out (may be NULL)
    uint16_t* out;           // [B][34][34][128], interior written
    int ntiles;              // 8 per crop
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

__host__ __device__ constexpr int rs_ky(int S) { return (S >> 2) / 3; }
__host__ __device__ constexpr int rs_kx(int S) { return (S >> 2) % 3; }
// fragment of output row J (of this wave's four), K-step S
template <int S, int J>
__device__ __forceinline__ void rs_read(bf16x8 (&xf)[2][4], const uint32_t (&ad)[8][2]) {
    constexpr int ky = rs_ky(S), kx = rs_kx(S), c = (S >> 1) & 1, kh = S & 1;
    constexpr int d = (2 * (J + ky) + kx) & 7;
    constexpr int imm = c * kChunkB + ((J + ky) * kPitch + kx - d) * kPixB;
    static_assert(imm >= 0 && imm < 65536, "ds offset");
    xf[S & 1][J] = lds_read_b128<imm>(ad[d][kh]);
}
// K-step S: its 8 MFMAs with the four fragment reads of step S + 1 beside them; a pair waits only for ITS fragment, issued a step
// ago (counted lgkmcnt: conv_fb.hip).  Step 0 starts the accumulation chains from the bias registers.
template <bool F16, int S>
__device__ __forceinline__ void rs_step(f32x4 (&acc)[2][4], bf16x8 (&xf)[2][4], const bf16x8 (&w)[kSteps][2], const uint32_t (&ad)[8][2],
                                        const f32x4 (&bias)[2]) {
    constexpr int set = S & 1;
    constexpr bool NEXT = S + 1 < kSteps, AG = S < kAgprSteps;
#define RS_PAIR(J, CNT)                                                                                  \
    if constexpr (NEXT) rs_read<(NEXT ? S + 1 : S), J>(xf, ad);                                          \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NEXT ? 4 : CNT) : "memory");                              \
    if constexpr (S == 0) {                                                                              \
        Elem<F16>::mfma_agpr_from(acc[0][J], w[S][0], xf[set][J], bias[0]);                              \
        Elem<F16>::mfma_agpr_from(acc[1][J], w[S][1], xf[set][J], bias[1]);                              \
    } else if constexpr (AG) {                                                                           \
        Elem<F16>::mfma_agpr(acc[0][J], w[S][0], xf[set][J]);                                            \
        Elem<F16>::mfma_agpr(acc[1][J], w[S][1], xf[set][J]);                                            \
    } else {                                                                                             \
        Elem<F16>::mfma_vgpr(acc[0][J], w[S][0], xf[set][J]);                                            \
        Elem<F16>::mfma_vgpr(acc[1][J], w[S][1], xf[set][J]);                                            \
    }
    RS_PAIR(0, 3) RS_PAIR(1, 2) RS_PAIR(2, 1) RS_PAIR(3, 0)
#undef RS_PAIR
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef SPSP_CONV_STAMPS
#define RSTAMP(i) if (a.dbg && blockIdx.x == 8 && tid == 0 && ntile_done == 1) a.dbg[i] = __builtin_amdgcn_s_memtime()
#define RSTAMPK(i) if (a.dbg && blockIdx.x == 8 && tid == 0) a.dbg[i] = __builtin_amdgcn_s_memtime()
#else
#define RSTAMP(i)
#define RSTAMPK(i)
#endif

template <bool F16, bool RES>
__global__ __launch_bounds__(256, 1) void conv3x3rs_kernel(const RsArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave & 1, ph = wave >> 1;
    const int frow = lane & 15, fq = lane >> 4;
    RSTAMPK(0);
    // Workgroups b and b + 8 share an XCD (round-robin dispatch; speed only): they take the two channel blocks of the SAME tile range, so
    // that the patches come out of one L2; consecutive ranges per XCD
    const int G = (int)gridDim.x;                           // even: half per channel block
    int nb, u;
    if ((G & 15) == 0) { nb = ((int)blockIdx.x >> 3) & 1; const int v = ((int)blockIdx.x & 7) + 8 * ((int)blockIdx.x >> 4); u = (v & 7) * (G >> 4) + (v >> 3); }
    else { nb = (int)blockIdx.x & 1; u = (int)blockIdx.x >> 1; }
    int t = u * a.tiles_per_wg;
    const int t_end = t + a.tiles_per_wg < a.ntiles ? t + a.tiles_per_wg : a.ntiles;
    if (t >= t_end) return;
    int ntile_done = 0;

    // ---- LDS-DMA of a tile's patch: piece j = sub-image c = j / 23, pixels 8 jj .. 8 jj + 7 (jj = j % 23) of the 10 x 18 patch (the
    //      last four slots of a sub-image are pad: they repeat pixel 179 and are never read); lane l feeds slot 8 jj + l / 8, physical
    //      chunk l % 8 = logical chunk (l % 8) ^ (slot & 7)
    const int drow = lane >> 3, dchunk = lane & 7;
    auto issue_piece = [&](int i, int img, int y0, int x0, int buf) {
        const int j = wave + 4 * i;
        if (j >= kPieces) return;
        const int c = j >= kChunkPieces ? 1 : 0, jj = j - kChunkPieces * c;
        const int slot = 8 * jj + drow;
        const int p = slot < 179 ? slot : 179;
        const int py = (p * 57) >> 10, px = p - kPitch * py;      // p / 18 for p < 180
        dma16(a.in + ((size_t)(img * 34 + y0 + py) * 34 + x0 + px) * 128 + 64 * c + ((dchunk ^ (slot & 7)) * 8),
              smem + buf * kTileB + c * kChunkB + jj * 1024);
    };
    // ---- prologue: this wave's 72 weight fragments (resident for the whole kernel) and the first tile's patch
    bf16x8 w[kSteps][2];
    {
        const bf16x8* const wsrc = reinterpret_cast<const bf16x8*>(a.wf) + (size_t)((nb * 2 + ch) * kSteps * 2) * 64 + lane;
#pragma unroll
        for (int s = 0; s < kSteps; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) w[s][i] = wsrc[(s * 2 + i) * 64];
    }
    {
        const int img = t >> 3, r = t & 7;
#pragma unroll
        for (int i = 0; i < kPiecesPerWave; ++i) issue_piece(i, img, (r >> 1) * 8, (r & 1) * 16, 0);
    }
    // accumulator element e of tile (i, j) = channel 64 nb + 32 ch + 8 fq + 4 i + e
    f32x4 bias_v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 u4 = *reinterpret_cast<const float4*>(a.bias + 64 * nb + 32 * ch + 8 * fq + 4 * i);
        bias_v[i] = f32x4{u4.x, u4.y, u4.z, u4.w};
    }
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    uint32_t a0[8][2];                 // pixel (frow + d) of a patch row, logical chunk kh * 4 + fq
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        a0[d][0] = (uint32_t)(ph * 4 * kPitch * kPixB + (frow + d) * kPixB + ((fq ^ ((frow + d) & 7)) * 16));
        a0[d][1] = a0[d][0] ^ 64u;
    }
    const int lane_out = ((4 * ph + 1) * 34 + frow + 1) * 128 + 64 * nb + 32 * ch + 8 * fq;     // (elements; + the tile's origin)

    int cur = 0;
    for (;; cur ^= 1) {
        const int img = t >> 3, r = t & 7;
        const int y0 = (r >> 1) * 8, x0 = (r & 1) * 16;
        // this tile's patch: this wave's pieces have landed (older than the four stores of the previous tile), then everybody's; every wave
        // has left the other buffer
        RSTAMP(1);
        if (ntile_done == 0) wait_vmcnt<0>(); else wait_vmcnt<4>();
        __builtin_amdgcn_s_barrier();
        RSTAMP(2);
        const bool has_next = t + 1 < t_end;
        const int nimg = (t + 1) >> 3, nr = (t + 1) & 7, ny0 = (nr >> 1) * 8, nx0 = (nr & 1) * 16;
        const size_t tile_off = ((size_t)(img * 34 + y0) * 34 + x0) * 128 + lane_out;
        uint16_t* const out_t = a.out + tile_off;
        // the residual of this tile's four rows: requested now, used in the epilogue (in front of the next tile's patch pieces in the
        // vmcnt order, so their wait does not wait for those)
        // (inline asm: behind loads hipcc can see it puts an s_waitcnt vmcnt(0) in front of the tile's first MFMA -- the whole latency
        //  of the four loads exposed per tile; the wait is spelled out in front of the epilogue instead)
        u32x4 resv[4];
        if constexpr (RES) {
            const uint16_t* const rp = a.res + tile_off;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                         "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off"
                         : "=&v"(resv[0]), "=&v"(resv[1]), "=&v"(resv[2]), "=&v"(resv[3])
                         : "v"(rp), "v"(rp + 34 * 128), "v"(rp + 2 * 34 * 128), "v"(rp + 3 * 34 * 128) : "memory");
        }
        uint32_t ad[8][2];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            ad[d][0] = smem_base + cur * kTileB + a0[d][0];
            ad[d][1] = smem_base + cur * kTileB + a0[d][1];
        }
        f32x4 acc[2][4];
        bf16x8 xf[2][4];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (counted waits below: nothing else may be in flight)
        rs_read<0, 0>(xf, ad); rs_read<0, 1>(xf, ad); rs_read<0, 2>(xf, ad); rs_read<0, 3>(xf, ad);
        __builtin_amdgcn_sched_barrier(0);
#define RS_S(S) rs_step<F16, S>(acc, xf, w, ad, bias_v);
#define RS_S_DMA(S) RS_S(S) if (has_next) issue_piece(S, nimg, ny0, nx0, cur ^ 1);
        RS_S_DMA(0) RS_S_DMA(1) RS_S_DMA(2) RS_S_DMA(3) RS_S_DMA(4) RS_S_DMA(5) RS_S_DMA(6) RS_S_DMA(7) RS_S_DMA(8) RS_S_DMA(9)
        RS_S_DMA(10) RS_S_DMA(11)
        RS_S(12) RS_S(13) RS_S(14) RS_S(15) RS_S(16) RS_S(17) RS_S(18) RS_S(19) RS_S(20) RS_S(21) RS_S(22) RS_S(23)
        RS_S(24) RS_S(25) RS_S(26) RS_S(27) RS_S(28) RS_S(29) RS_S(30) RS_S(31) RS_S(32) RS_S(33) RS_S(34) RS_S(35)
#undef RS_S_DMA
#undef RS_S
        RSTAMP(3);
        if constexpr (RES) {     // the residual loads are older than the (at most twelve) patch pieces issued behind them
            if (has_next) asm volatile("s_waitcnt vmcnt(12)" : "+v"(resv[0]), "+v"(resv[1]), "+v"(resv[2]), "+v"(resv[3]));
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(resv[0]), "+v"(resv[1]), "+v"(resv[2]), "+v"(resv[3]));
        }
        mfma_drain8(acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
        __builtin_amdgcn_sched_barrier(0);
        // accumulators (the bias went in with the first K-step) [+ residual] -> 16-bit -> ReLU on the packed pairs -> 16-byte stores
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 lo = acc[0][j], hi = acc[1][j];
            float v0 = lo[0], v1 = lo[1], v2 = lo[2], v3 = lo[3], v4 = hi[0], v5 = hi[1], v6 = hi[2], v7 = hi[3];
            if constexpr (RES) {
                const u32x4 rq = resv[j];
                v0 += Elem<F16>::lo(rq[0]); v1 += Elem<F16>::hi(rq[0]); v2 += Elem<F16>::lo(rq[1]); v3 += Elem<F16>::hi(rq[1]);
                v4 += Elem<F16>::lo(rq[2]); v5 += Elem<F16>::hi(rq[2]); v6 += Elem<F16>::lo(rq[3]); v7 += Elem<F16>::hi(rq[3]);
            }
            uint32_t p0 = Elem<F16>::pack2(v0, v1), p1 = Elem<F16>::pack2(v2, v3), p2 = Elem<F16>::pack2(v4, v5), p3 = Elem<F16>::pack2(v6, v7);
            if (a.relu) { p0 = pk_relu16(p0); p1 = pk_relu16(p1); p2 = pk_relu16(p2); p3 = pk_relu16(p3); }
            *reinterpret_cast<uint4*>(out_t + (size_t)j * (34 * 128)) = make_uint4(p0, p1, p2, p3);
        }
        RSTAMP(4);
        ++ntile_done;
        if (!has_next) break;
        ++t;
    }
    RSTAMPK(5);
}

}  // namespace

namespace spsp {

// w [128][1152] (K = tap-major, channel-minor; BN folded) -> conv3x3rs_kernel's register-order fragment image
// [nb][ch][K-step s = 4 tap + 2 chunk + kh][tile i][lane][8]: MFMA row rho of tile i = output channel 64 nb + 32 ch + 8 (rho >> 2) + 4 i +
// (rho & 3); lane (rho, fq) holds k = tap * 128 + chunk * 64 + kh * 32 + 8 fq .. + 7
void pack_rs_fragments(const uint16_t* w, std::vector<uint16_t>& out) {
    out.assign((size_t)2 * 2 * kSteps * 2 * 64 * 8, 0);
    for (int nb = 0; nb < 2; ++nb)
        for (int ch = 0; ch < 2; ++ch)
            for (int s = 0; s < kSteps; ++s)
                for (int i = 0; i < 2; ++i)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int rho = lane & 15, fq = lane >> 4;
                        const int n = 64 * nb + 32 * ch + 8 * (rho >> 2) + 4 * i + (rho & 3);
                        const int k0 = (s >> 2) * 128 + ((s >> 1) & 1) * 64 + (s & 1) * 32 + 8 * fq;
                        uint16_t* d = &out[(((((size_t)nb * 2 + ch) * kSteps + s) * 2 + i) * 64 + lane) * 8];
                        for (int e = 0; e < 8; ++e) d[e] = w[(size_t)n * 1152 + k0 + e];
                    }
}

// in / out / res: [B][34][34][128] maps; wf: pack_rs_fragments' image on the device; max_workgroups: even, the CUs to spread over
int launch_conv_rs(const uint16_t* in, const uint16_t* wf, const float* bias, const uint16_t* res, uint16_t* out, int B, int f16, int relu,
                   int max_workgroups, hipStream_t st) {
    SPSP_REQUIRE(in && wf && bias && out && in != out && B > 0 && max_workgroups >= 2, "conv_rs: bad arguments");
    RsArgs a;
    a.in = in; a.wf = wf; a.bias = bias; a.res = res; a.out = out; a.relu = relu;
    a.ntiles = 8 * B;
    const int per_block = max_workgroups / 2;
    a.tiles_per_wg = (a.ntiles + per_block - 1) / per_block;
    const int grid = 2 * ((a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg);
    a.dbg = nullptr;
    SPSP_TRY((ensure_dynamic_lds<&conv3x3rs_kernel<true, true>>(kLds)));
    SPSP_TRY((ensure_dynamic_lds<&conv3x3rs_kernel<true, false>>(kLds)));
    SPSP_TRY((ensure_dynamic_lds<&conv3x3rs_kernel<false, true>>(kLds)));
    SPSP_TRY((ensure_dynamic_lds<&conv3x3rs_kernel<false, false>>(kLds)));
#ifdef SPSP_CONV_STAMPS
    if (getenv("SPSP_CONV_DEBUG")) {
        (void)hipMalloc(&a.dbg, 16 * sizeof(unsigned long long));
        (void)hipMemset(a.dbg, 0, 16 * sizeof(unsigned long long));
        if (res) conv3x3rs_kernel<true, true><<<dim3(grid), 256, kLds, st>>>(a); else conv3x3rs_kernel<true, false><<<dim3(grid), 256, kLds, st>>>(a);
        (void)hipDeviceSynchronize();
        unsigned long long h[16];
        (void)hipMemcpy(h, a.dbg, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "conv_rs grid=%d tiles/wg=%d: prologue + tile 1 %llu | tile 2: wait+barrier %llu | K loop %llu | epilogue %llu | whole kernel %llu cycles\n",
                grid, a.tiles_per_wg, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[0]);
        (void)hipFree(a.dbg);
        return SPSP_OK;
    }
#endif
    auto go = [&](auto f16_c, auto res_c) {
        launch_kernel(&conv3x3rs_kernel<decltype(f16_c)::value, decltype(res_c)::value>, dim3(grid), dim3(256), kLds, st, a);
    };
    if (f16) { if (res) go(std::true_type{}, std::true_type{}); else go(std::true_type{}, std::false_type{}); }
    else { if (res) go(std::false_type{}, std::true_type{}); else go(std::false_type{}, std::false_type{}); }
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp
