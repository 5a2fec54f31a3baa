// ONE launch per BasicBlock of layer1 (PR/models/resnet.py:61-77: out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x), 64 -> 64
// channels on 64 x 64 maps) -- round 5.  Until now a block was two launches of conv_wres.hip: the 16.8 MB intermediate map went to
// HBM and came back, the block's input was read twice (patch + residual), and each launch paid its own prologue (72 KB of weights
// through L2 -> LDS per workgroup) and launch boundary.
//
//   * A workgroup walks 16 x 16 output tiles.  Per tile it fetches ONE 20 x 20 input patch (LDS-DMA, two buffers: the next tile's
//     patch lands under this tile's MFMAs), computes conv1 + bias + ReLU for the 18 x 18 pixels conv2 needs INTO LDS (positions
//     outside the image are conv2's zero padding and are stored as zeros), then conv2 + bias + residual + ReLU from LDS; the
//     residual is the patch's centre.  Nothing but the input patch is read and nothing but the output tile is written.
//   * BOTH convolutions' weights are stationary in REGISTERS (the stem's / conv_s2rs's scheme): wave (ch, ph) owns output channels
//     32 ch .. 32 ch + 31 over the whole K = 576 of both convolutions = 72 MFMA A-fragments = 288 registers of the 512 a lone
//     wave per SIMD has.  hipcc keeps long-lived fragments in accumulation registers and copies them back for every MFMA
//     (v_accvgpr_read: 578 of them for 320 MFMAs in conv_s2rs) -- here the MFMAs are inline asm whose A operand is constrained to
//     the accumulation file, which the matrix core reads directly (248 of the 256 AGPRs; the last ten fragments stay in VGPRs).
//     The host packs the fragments in register order, so a wave's prologue is 72 fully coalesced 1-KiB loads and LDS carries
//     pixels only.
//   * conv1 runs on the LINEAR index m = 20 r + c of a pitch-20 grid (the patch's own pitch): 16 consecutive m are 16 consecutive
//     patch pixels for every filter tap, so a fragment address is `per-lane base + immediate` with the bank swizzle (16-byte chunk
//     XOR pixel & 7) folded into six per-lane bases, one per (4 ky + kx) & 7 -- 23 tiles of 16 for the 18 rows (two garbage columns
//     per row: +10 % over the 20.25 a dense 18 x 18 would need; a dense index would need an address register per tile and tap).
//     conv2's sixteen output rows read the intermediate grid the same way.  tests/emul/conv_fb_model.py mirrors all of it.
// Workgroup = 4 waves = (channel half ch) x (pixel half ph); a K-step = 4 pixel-fragment reads beside 8 MFMAs.
#include <type_traits>

#include "conv.hpp"
#include "conv_elem.hpp"

namespace {
using namespace spsp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

typedef short i16x2 __attribute__((ext_vector_type(2)));
// max(x, +0) on both 16-bit floats of a pair, either element type: as SIGNED 16-bit integers every negative value (-0 included) is
// below +0 = 0.  Rounding to the element type and ReLU commute (both monotone), so ReLU runs on the packed pair: one instruction
// where the fp32 form takes two (conv_stem.hip does the same).
__device__ __forceinline__ uint32_t pk_relu16(uint32_t x) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, x), i16x2{0, 0}));
}

constexpr int kPitch = 20, kPixB = 128;
constexpr int kPatchB = kPitch * kPitch * kPixB;       // 51 200: 50 LDS-DMA pieces of 8 pixels
constexpr int kInterOff = 2 * kPatchB;                 // two patch buffers, then the intermediate grid
constexpr int kInterPix = 368;                         // 23 tiles of 16 (m = 20 r + c <= 357 is ever read)
constexpr int kLds = kInterOff + kInterPix * kPixB;    // 149 504 B
constexpr int kPieces = 50, kPiecesPerWave = 13;
constexpr int kAgprSteps2 = 13;                        // conv2: the fragments of K-steps 0..12 in AGPRs (with conv1's: 248 registers)
static_assert(kLds <= 160 * 1024, "LDS plan");

struct FbArgs {
    const uint16_t* in;      // [B][66][66][64], zero border
    const uint16_t* wf;      // [conv 2][ch 2][K-step 18][tile 2][lane 64][8]: MFMA A-fragments in register order (pack_block_fragments)
    const float* bias1;      // [64] folded BN biases of the two convolutions
    const float* bias2;
    uint16_t* out;           // [B][66][66][64]: interior written
    int ntiles;              // 16 per crop
    int tiles_per_wg;
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

// acc += W . X on the matrix core; AG: the weight fragment is read from the accumulation register file (conv_elem.hpp).  FIRST: the
// chain starts here, from the bias registers (no VALU initialisation of the accumulators: conv_elem.hpp)
template <bool F16, bool AG, bool FIRST>
__device__ __forceinline__ void fb_mfma(f32x4& acc, const bf16x8& w, const bf16x8& x, const f32x4& bias) {
    if constexpr (FIRST) { static_assert(AG, "a chain's first K-step is an AGPR step"); Elem<F16>::mfma_agpr_from(acc, w, x, bias); }
    else if constexpr (AG) Elem<F16>::mfma_agpr(acc, w, x);
    else Elem<F16>::mfma_vgpr(acc, w, x);
}

// the six swizzle variants: (4 ky + kx) & 7 over the nine taps (and, for conv2, over 4 (row + ky) + kx)
__host__ __device__ constexpr int fb_delta_index(int d) { return d < 4 ? d : d - 1; }      // 0 1 2 4 5 6 -> 0..5
// fragment of pixel tile / output row U (within this wave's twelve / eight), K-step S = 2 tap + kh
template <int CONV>
__host__ __device__ constexpr int fb_delta(int U, int S) {
    const int tap = S >> 1, ky = tap / 3, kx = tap % 3;
    return CONV == 0 ? ((4 * ky + kx) & 7) : ((4 * (U + ky) + kx) & 7);
}
template <int CONV>
__host__ __device__ constexpr int fb_imm(int U, int S) {
    const int tap = S >> 1, ky = tap / 3, kx = tap % 3;
    return CONV == 0 ? (16 * U + 20 * ky + kx - fb_delta<0>(U, S)) * kPixB : (20 * (U + ky) + kx - fb_delta<1>(U, S)) * kPixB;
}

// the four pixel fragments of (pass PASS, K-step S) into xf[S & 1]
template <int CONV, int PASS, int S, int J>
__device__ __forceinline__ void fb_read(bf16x8 (&xf)[2][4], const uint32_t (&ad)[6][2]) {
    constexpr int U = 4 * PASS + J;
    xf[S & 1][J] = lds_read_b128<fb_imm<CONV>(U, S)>(ad[fb_delta_index(fb_delta<CONV>(U, S))][S & 1]);
}

// K-step S of pass PASS: its 8 MFMAs, with the four fragment reads of the following step (of this pass or the next) beside them
template <bool F16, int CONV, int PASS, int NPASS, int S>
__device__ __forceinline__ void fb_step(f32x4 (&acc)[2][4], bf16x8 (&xf)[2][4], const bf16x8 (&w)[18][2], const uint32_t (&ad)[6][2],
                                        const f32x4 (&bias)[2]) {
    constexpr bool AG = CONV == 0 || S < kAgprSteps2;
    constexpr int NP = S == 17 ? PASS + 1 : PASS, NS = S == 17 ? 0 : S + 1;
    constexpr bool NEXT = NP < NPASS;
    constexpr int set = S & 1;
    constexpr bool F = S == 0;
    // Fragment reads run a whole K-step ahead: the read of the NEXT step's fragment j is issued in front of this step's MFMA pair j,
    // and a pair waits only for ITS fragment -- issued a step ago -- with a counted lgkmcnt (LDS reads return in order; at that point
    // the reads issued after it are the three of the previous step behind it and the one just issued: 4; a convolution's last step
    // has no reads behind it: 3 - j).  The loop holds no scalar load (they would make the count meaningless).  The first version waited
    // lgkmcnt(0) at the end of every step, ~30 cycles behind its last read: 215 cycles per step for 128 of MFMA (stamps).
    if constexpr (NEXT) fb_read<CONV, NP, NS, 0>(xf, ad);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NEXT ? 4 : 3) : "memory");
    fb_mfma<F16, AG, F>(acc[0][0], w[S][0], xf[set][0], bias[0]);
    fb_mfma<F16, AG, F>(acc[1][0], w[S][1], xf[set][0], bias[1]);
    if constexpr (NEXT) fb_read<CONV, NP, NS, 1>(xf, ad);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NEXT ? 4 : 2) : "memory");
    fb_mfma<F16, AG, F>(acc[0][1], w[S][0], xf[set][1], bias[0]);
    fb_mfma<F16, AG, F>(acc[1][1], w[S][1], xf[set][1], bias[1]);
    if constexpr (NEXT) fb_read<CONV, NP, NS, 2>(xf, ad);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NEXT ? 4 : 1) : "memory");
    fb_mfma<F16, AG, F>(acc[0][2], w[S][0], xf[set][2], bias[0]);
    fb_mfma<F16, AG, F>(acc[1][2], w[S][1], xf[set][2], bias[1]);
    if constexpr (NEXT) fb_read<CONV, NP, NS, 3>(xf, ad);
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NEXT ? 4 : 0) : "memory");
    fb_mfma<F16, AG, F>(acc[0][3], w[S][0], xf[set][3], bias[0]);
    fb_mfma<F16, AG, F>(acc[1][3], w[S][1], xf[set][3], bias[1]);
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef SPSP_CONV_STAMPS
#define FSTAMP(i) if (a.dbg && blockIdx.x == 8 && tid == 0 && ntile_done == 1) a.dbg[i] = __builtin_amdgcn_s_memtime()
#define FSTAMPK(i) if (a.dbg && blockIdx.x == 8 && tid == 0) a.dbg[i] = __builtin_amdgcn_s_memtime()
#else
#define FSTAMP(i)
#define FSTAMPK(i)
#endif

template <bool F16>
__global__ __launch_bounds__(256, 1) void conv_basic_block_kernel(const FbArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = wave & 1, ph = wave >> 1;
    const int frow = lane & 15, fq = lane >> 4;
    FSTAMPK(0);
    // Workgroups b, b + 8, ... share an XCD (round-robin dispatch; speed only): they take CONSECUTIVE tile ranges, so that the halos
    // neighbouring tiles share come out of one L2.  (Tiles from per-XCD device counters, conv_wres.hip's dynamic scheduler, were built
    // and measured in round 5: 130.7 k players/s in the four-slot pipeline with either, +0.5 us per launch alone -- fixed ranges stay.)
    const int G = (int)gridDim.x;
    int L = (int)blockIdx.x;
    if ((G & 7) == 0) L = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
    int t = L * a.tiles_per_wg;
    const int t_end = t + a.tiles_per_wg < a.ntiles ? t + a.tiles_per_wg : a.ntiles;
    if (t >= t_end) return;
    int ntile_done = 0;

    // ---- LDS-DMA of a tile's patch: piece j = patch pixels 8 j .. 8 j + 7 (wave w: j = w, w + 4, ...); lane l feeds pixel
    //      p = 8 j + l / 8, physical 16-byte chunk l % 8 = logical chunk (l % 8) ^ (p & 7).  Rows / columns beyond the buffer's
    //      one-pixel border are clamped: they only feed intermediate positions outside the image, which are stored as zeros.
    const int drow = lane >> 3, dchunk = lane & 7;
    auto issue_piece = [&](int i, int img, int y0, int x0, int buf) {
        const int j = wave + 4 * i;
        if (j >= kPieces) return;
        const int p = 8 * j + drow;
        const int py = (p * 205) >> 12, px = p - kPitch * py;
        int row = y0 - 1 + py, col = x0 - 1 + px;
        row = row < 0 ? 0 : (row > 65 ? 65 : row);
        col = col < 0 ? 0 : (col > 65 ? 65 : col);
        dma16(a.in + ((size_t)(img * 66 + row) * 66 + col) * 64 + ((dchunk ^ (p & 7)) * 8), smem + buf * kPatchB + j * 1024);
    };
    // ---- prologue.  This wave's 72 weight fragments, resident for the whole kernel: 72 fully coalesced 1-KiB loads (the host packed
    // them in register order), and the first tile's patch pieces between the two convolutions' halves.
    bf16x8 wA[18][2], wB[18][2];
    const bf16x8* const w1 = reinterpret_cast<const bf16x8*>(a.wf) + (size_t)(ch * 36) * 64 + lane;
    const bf16x8* const w2 = reinterpret_cast<const bf16x8*>(a.wf) + (size_t)((2 + ch) * 36) * 64 + lane;
#pragma unroll
    for (int s = 0; s < 18; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) wA[s][i] = w1[(s * 2 + i) * 64];
    __builtin_amdgcn_sched_barrier(0);
    {
        const int img = t >> 4, r = t & 15;
#pragma unroll
        for (int i = 0; i < kPiecesPerWave; ++i) issue_piece(i, img, (r >> 2) * 16, (r & 3) * 16, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 18; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) wB[s][i] = w2[(s * 2 + i) * 64];
    // (hipcc waits for every load it knows of -- vmcnt(0) -- before it enters the tile loop.  Round 5 tried conv2's fragments as
    //  inline-asm loads inside the first tile, under conv1's MFMAs: the conditionally loaded registers cost 42 VGPRs and spilled.)
    // accumulator element e of tile (i, j) = channel 32 ch + 8 fq + 4 i + e
    f32x4 bias1[2], bias2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 u = *reinterpret_cast<const float4*>(a.bias1 + 32 * ch + 8 * fq + 4 * i);
        const float4 v = *reinterpret_cast<const float4*>(a.bias2 + 32 * ch + 8 * fq + 4 * i);
        bias1[i] = f32x4{u.x, u.y, u.z, u.w};
        bias2[i] = f32x4{v.x, v.y, v.z, v.w};
    }
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    // fragment bases: pixel (frow + d) of a grid, logical chunk kh * 4 + fq
    uint32_t a0[6][2], ad2[6][2];
#pragma unroll
    for (int di = 0; di < 6; ++di) {
        const int d = di < 3 ? di : di + 1;
        a0[di][0] = (uint32_t)((frow + d) * kPixB + ((fq ^ ((frow + d) & 7)) * 16));
        a0[di][1] = a0[di][0] ^ 64u;
        ad2[di][0] = smem_base + kInterOff + ph * 8 * kPitch * kPixB + a0[di][0];
        ad2[di][1] = smem_base + kInterOff + ph * 8 * kPitch * kPixB + a0[di][1];
    }
    // intermediate pixel m = 16 (12 ph + T') + frow: its 16 bytes = channels 32 ch + 8 fq .. + 7 (logical chunk 4 ch + fq)
    const uint32_t e0 = smem_base + kInterOff + ph * 12 * 2048 + frow * kPixB + (((ch * 4 + fq) ^ (frow & 7)) * 16);
    // ... zeros where it lies outside the image: grid row 0 / 17 at the image's top / bottom border, column 0 / 17 at its left / right
    // one.  Four flag bits per tile T' (bit 0 row 0, 1 row 17, 2 column 0, 3 column 17) of this lane's pixel, twelve tiles in two words.
    uint32_t edge_lo = 0u, edge_hi = 0u;
#pragma unroll
    for (int tp = 0; tp < 12; ++tp) {
        const int m = 192 * ph + frow + 16 * tp;
        const int rr = (m * 205) >> 12, cc = m - kPitch * rr;
        const uint32_t f = (rr == 0 ? 1u : 0u) | (rr == 17 ? 2u : 0u) | (cc == 0 ? 4u : 0u) | (cc == 17 ? 8u : 0u);
        if (tp < 8) edge_lo |= f << (4 * tp); else edge_hi |= f << (4 * (tp - 8));
    }
    // residual of output pixel (8 ph + oy', frow) = patch pixel (oy + 2, frow + 2), the same 16 bytes; two swizzle keys by row parity
    uint32_t rr0[2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
        rr0[par] = (uint32_t)(ph * 8 * kPitch * kPixB + (frow + 42) * kPixB + (((ch * 4 + fq) ^ ((frow + 2 + 4 * par) & 7)) * 16));
    const int lane_out = ((8 * ph + 1) * 66 + frow + 1) * 64 + 32 * ch + 8 * fq;      // (elements; + tile origin)

    int cur = 0;
    for (;; cur ^= 1) {
        const int img = t >> 4, r = t & 15;
        const int y0 = (r >> 2) * 16, x0 = (r & 3) * 16;
        // this tile's patch: this wave's pieces have landed (they are older than the eight stores of the previous tile), then everybody's;
        // every wave has left the other patch buffer and the intermediate grid
        FSTAMP(1);
        if (ntile_done == 0) wait_vmcnt<0>(); else wait_vmcnt<8>();
        __builtin_amdgcn_s_barrier();
        FSTAMP(2);
        const bool has_next = t + 1 < t_end;
        const int nimg = (t + 1) >> 4, nr = (t + 1) & 15, ny0 = (nr >> 2) * 16, nx0 = (nr & 3) * 16;
        const uint32_t border = (y0 == 0 ? 1u : 0u) | (y0 == 48 ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 == 48 ? 8u : 0u);
        uint32_t ad1[6][2];
#pragma unroll
        for (int di = 0; di < 6; ++di) {
            ad1[di][0] = smem_base + cur * kPatchB + ph * 12 * 2048 + a0[di][0];
            ad1[di][1] = smem_base + cur * kPatchB + ph * 12 * 2048 + a0[di][1];
        }
        f32x4 acc[2][4];
        bf16x8 xf[2][4];

        // ================= conv1: three passes of four pixel tiles, K = 18 steps each =================
#define FB_BIAS_0 bias1
#define FB_BIAS_1 bias2
#define FB_S(CONV, PASS, NPASS, S, W, AD) fb_step<F16, CONV, PASS, NPASS, S>(acc, xf, W, AD, FB_BIAS_##CONV);
#define FB_PASS_TAIL(CONV, PASS, NPASS, W, AD)                                                                           \
        FB_S(CONV, PASS, NPASS, 13, W, AD) FB_S(CONV, PASS, NPASS, 14, W, AD) FB_S(CONV, PASS, NPASS, 15, W, AD)       \
        FB_S(CONV, PASS, NPASS, 16, W, AD) FB_S(CONV, PASS, NPASS, 17, W, AD)
#define FB_PASS(CONV, PASS, NPASS, W, AD)                                                                                \
        FB_S(CONV, PASS, NPASS, 0, W, AD) FB_S(CONV, PASS, NPASS, 1, W, AD) FB_S(CONV, PASS, NPASS, 2, W, AD)          \
        FB_S(CONV, PASS, NPASS, 3, W, AD) FB_S(CONV, PASS, NPASS, 4, W, AD) FB_S(CONV, PASS, NPASS, 5, W, AD)          \
        FB_S(CONV, PASS, NPASS, 6, W, AD) FB_S(CONV, PASS, NPASS, 7, W, AD) FB_S(CONV, PASS, NPASS, 8, W, AD)          \
        FB_S(CONV, PASS, NPASS, 9, W, AD) FB_S(CONV, PASS, NPASS, 10, W, AD) FB_S(CONV, PASS, NPASS, 11, W, AD)        \
        FB_S(CONV, PASS, NPASS, 12, W, AD) FB_PASS_TAIL(CONV, PASS, NPASS, W, AD)
        // the first fragments of a convolution (later passes' arrive beside the previous pass's last K-step)
#define FB_FIRST(CONV, AD)                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* (counted waits below: nothing else may be in flight) */      \
        fb_read<CONV, 0, 0, 0>(xf, AD); fb_read<CONV, 0, 0, 1>(xf, AD); fb_read<CONV, 0, 0, 2>(xf, AD); fb_read<CONV, 0, 0, 3>(xf, AD); \
        __builtin_amdgcn_sched_barrier(0);
        // accumulators -> ReLU -> 16-bit -> the intermediate grid (zeros outside the image); T' = 4 PASS + j
#define FB_EPI1_J(PASS, J)                                                                                               \
        if (!(ph == 1 && 4 * (PASS) + (J) == 11)) {          /* (tile 23: beyond the grid) */                            \
            /* outside the image = on an edge row / column of the grid (this lane's four flag bits of tile T') at an image border */ \
            const bool out_ = (((4 * (PASS) + (J)) < 8 ? edge_lo >> (4 * ((4 * (PASS) + (J)) & 7)) : edge_hi >> (4 * ((4 * (PASS) + (J)) & 7))) & border) != 0u; \
            const f32x4 lo_ = acc[0][J], hi_ = acc[1][J];                                                                \
            u32x4 v_ = u32x4{pk_relu16(Elem<F16>::pack2(lo_[0], lo_[1])), pk_relu16(Elem<F16>::pack2(lo_[2], lo_[3])),  \
                             pk_relu16(Elem<F16>::pack2(hi_[0], hi_[1])), pk_relu16(Elem<F16>::pack2(hi_[2], hi_[3]))}; \
            if (out_) v_ = u32x4{0u, 0u, 0u, 0u};                                                                        \
            /* (inline asm: behind a plain LDS store hipcc waits vmcnt(0) -- for the next tile's LDS-DMA, which it cannot tell apart) */ \
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(e0), "v"(v_), "n"((4 * (PASS) + (J)) * 2048) : "memory");    \
        }
#define FB_EPI1(PASS)                                                                                                    \
        mfma_drain8(acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]);   /* the last MFMAs' results -> VALU */ \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        FB_EPI1_J(PASS, 0) FB_EPI1_J(PASS, 1) FB_EPI1_J(PASS, 2) FB_EPI1_J(PASS, 3)

        FB_FIRST(0, ad1)
        // pass 0 carries the next tile's thirteen LDS-DMA pieces, one behind each of its first K-steps
#define FB_S_DMA(S) FB_S(0, 0, 3, S, wA, ad1) if (has_next) issue_piece(S, nimg, ny0, nx0, cur ^ 1);
        FB_S_DMA(0) FB_S_DMA(1) FB_S_DMA(2) FB_S_DMA(3) FB_S_DMA(4) FB_S_DMA(5) FB_S_DMA(6)
        FB_S_DMA(7) FB_S_DMA(8) FB_S_DMA(9) FB_S_DMA(10) FB_S_DMA(11) FB_S_DMA(12)
        FB_PASS_TAIL(0, 0, 3, wA, ad1)
        FB_EPI1(0)
        FB_PASS(0, 1, 3, wA, ad1)
        FB_EPI1(1)
        FB_PASS(0, 2, 3, wA, ad1)
        FB_EPI1(2)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's part of the intermediate grid is in LDS
        FSTAMP(3);
        __builtin_amdgcn_s_barrier();
        FSTAMP(4);

        // ================= conv2: two passes of four output rows =================
        uint16_t* const out_t = a.out + ((size_t)(img * 66 + y0) * 66 + x0) * 64 + lane_out;
        const uint32_t rbase = smem_base + cur * kPatchB;
        // residual reads first (their latency under the drain of the last MFMAs), then bias is already in, + residual, ReLU, store
#define FB_EPI2(PASS)                                                                                                    \
        {                                                                                                                \
            bf16x8 res_[4];                                                                                              \
            res_[0] = lds_read_b128<(4 * (PASS) + 0) * kPitch * kPixB>(rbase + rr0[0]);                                   \
            res_[1] = lds_read_b128<(4 * (PASS) + 1) * kPitch * kPixB>(rbase + rr0[1]);                                   \
            res_[2] = lds_read_b128<(4 * (PASS) + 2) * kPitch * kPixB>(rbase + rr0[0]);                                   \
            res_[3] = lds_read_b128<(4 * (PASS) + 3) * kPitch * kPixB>(rbase + rr0[1]);                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
            mfma_drain8(acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]);         \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                           \
                const uint4 rq_ = __builtin_bit_cast(uint4, res_[j_]);                                                   \
                const f32x4 lo_ = acc[0][j_], hi_ = acc[1][j_];                                                          \
                const float v0_ = lo_[0] + Elem<F16>::lo(rq_.x), v1_ = lo_[1] + Elem<F16>::hi(rq_.x);                    \
                const float v2_ = lo_[2] + Elem<F16>::lo(rq_.y), v3_ = lo_[3] + Elem<F16>::hi(rq_.y);                    \
                const float v4_ = hi_[0] + Elem<F16>::lo(rq_.z), v5_ = hi_[1] + Elem<F16>::hi(rq_.z);                    \
                const float v6_ = hi_[2] + Elem<F16>::lo(rq_.w), v7_ = hi_[3] + Elem<F16>::hi(rq_.w);                    \
                *reinterpret_cast<uint4*>(out_t + (size_t)(4 * (PASS) + j_) * (66 * 64)) =                                \
                    make_uint4(pk_relu16(Elem<F16>::pack2(v0_, v1_)), pk_relu16(Elem<F16>::pack2(v2_, v3_)),            \
                               pk_relu16(Elem<F16>::pack2(v4_, v5_)), pk_relu16(Elem<F16>::pack2(v6_, v7_)));           \
            }                                                                                                            \
        }
        FB_FIRST(1, ad2)
        FB_PASS(1, 0, 2, wB, ad2)
        FB_EPI2(0)
        FB_PASS(1, 1, 2, wB, ad2)
        FB_EPI2(1)
        FSTAMP(5);
        ++ntile_done;
        if (!has_next) break;
        ++t;
#undef FB_EPI2
#undef FB_S_DMA
#undef FB_EPI1
#undef FB_EPI1_J
#undef FB_FIRST
#undef FB_PASS
#undef FB_PASS_TAIL
#undef FB_S
#undef FB_BIAS_0
#undef FB_BIAS_1
    }
    FSTAMPK(6);
}

}  // namespace

namespace spsp {

// [64][576] element weights of the block's two convolutions (K = tap-major, channel-minor; BN folded) -> the register-order
// fragment image conv_basic_block_kernel loads: [conv][ch][K-step s = 2 tap + kh][tile i][lane][8]; MFMA row rho of tile i = output
// channel 32 ch + 8 (rho >> 2) + 4 i + (rho & 3), lane (rho, fq) holds k = tap * 64 + kh * 32 + 8 fq .. + 7
void pack_block_fragments(const uint16_t* w1, const uint16_t* w2, std::vector<uint16_t>& out) {
    out.assign((size_t)2 * 2 * 36 * 64 * 8, 0);
    for (int conv = 0; conv < 2; ++conv) {
        const uint16_t* w = conv ? w2 : w1;
        for (int ch = 0; ch < 2; ++ch)
            for (int s = 0; s < 18; ++s)
                for (int i = 0; i < 2; ++i)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int rho = lane & 15, fq = lane >> 4;
                        const int n = 32 * ch + 8 * (rho >> 2) + 4 * i + (rho & 3);
                        const int k0 = (s >> 1) * 64 + (s & 1) * 32 + 8 * fq;
                        uint16_t* d = &out[((((size_t)(conv * 2 + ch) * 18 + s) * 2 + i) * 64 + lane) * 8];
                        for (int e = 0; e < 8; ++e) d[e] = w[(size_t)n * 576 + k0 + e];
                    }
    }
}

// in / out: [B][66][66][64] maps (distinct buffers); wf: pack_block_fragments' image on the device; max_workgroups: the CUs to
// spread over (plan "throughput": a fraction of the chip -- more tiles per resident weight set)
int launch_conv_fb(const uint16_t* in, const uint16_t* wf, const float* bias1, const float* bias2, uint16_t* out, int B, int f16,
                   int max_workgroups, hipStream_t st) {
    SPSP_REQUIRE(in && wf && bias1 && bias2 && out && in != out && B > 0 && max_workgroups > 0, "conv_fb: bad arguments");
    FbArgs a;
    a.in = in; a.wf = wf; a.bias1 = bias1; a.bias2 = bias2; a.out = out;
    a.ntiles = 16 * B;
    a.tiles_per_wg = (a.ntiles + max_workgroups - 1) / max_workgroups;
    const int grid = (a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    a.dbg = nullptr;
    SPSP_TRY((ensure_dynamic_lds<&conv_basic_block_kernel<true>>(kLds)));
    SPSP_TRY((ensure_dynamic_lds<&conv_basic_block_kernel<false>>(kLds)));
#ifdef SPSP_CONV_STAMPS
    if (getenv("SPSP_CONV_DEBUG")) {
        (void)hipMalloc(&a.dbg, 16 * sizeof(unsigned long long));
        (void)hipMemset(a.dbg, 0, 16 * sizeof(unsigned long long));
        if (f16) conv_basic_block_kernel<true><<<dim3(grid), 256, kLds, st>>>(a); else conv_basic_block_kernel<false><<<dim3(grid), 256, kLds, st>>>(a);
        (void)hipDeviceSynchronize();
        unsigned long long h[16];
        (void)hipMemcpy(h, a.dbg, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "conv_fb grid=%d tiles/wg=%d: prologue %llu | tile 2: wait+barrier %llu | conv1 %llu | barrier %llu | conv2 %llu | whole kernel %llu cycles\n",
                grid, a.tiles_per_wg, h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[0]);
        (void)hipFree(a.dbg);
        return SPSP_OK;
    }
#endif
    if (f16) launch_kernel(&conv_basic_block_kernel<true>, dim3(grid), dim3(256), kLds, st, a);
    else launch_kernel(&conv_basic_block_kernel<false>, dim3(grid), dim3(256), kLds, st, a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp
