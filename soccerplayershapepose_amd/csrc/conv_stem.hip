// The stem of the backbone in ONE kernel: conv1 (7x7 / stride 2 / pad 3, 18 -> 64 channels, 256x256 -> 128x128)
// + folded BN bias + ReLU + max-pool 3x3 / stride 2 / pad 1 (-> 64x64)   (PR/models/resnet.py:145-149, 203-207).
//
// Why a new structure.  The previous stem (conv1_half_kernel + maxpool_kernel, 64 + 21 us at 32 crops) was bound by
// the L2 -> LDS fill path: every workgroup re-streamed the 114 KB weight matrix next to its input patch (362 MB of
// LDS-DMA per launch against 80 MB of input), then wrote the 128x128x64 map (69 MB) only for the pool kernel to read it
// back (102 MB with halos) and write 17.8 MB.  Here
//   * the WEIGHTS LIVE IN REGISTERS: a wave owns 32 output channels and keeps their whole [32 x 896] weight slice as MFMA
//     A-fragments (2 x 28 fragments = 224 VGPRs, loaded once; one wave per SIMD has 512 registers).  No weight traffic
//     and no weight LDS reads in the loop: a 32-deep sub-step reads 8 pixel fragments for 16 MFMAs;
//   * the INPUT IS A ROLLING WINDOW: a persistent workgroup walks a band of conv rows two at a time; a step needs 9 input
//     rows of which 5 are already in LDS, the 4 new ones (38 KB) arrive by LDS-DMA under the previous step's MFMAs
//     (13-row ring, 130 KB).  L2 -> LDS traffic = the input, once (+ one halo step per band);
//   * the POOL IS FUSED: horizontal 3 -> 1 max in registers (DPP row shifts: the 16 pixels of an MFMA tile sit in the 16
//     lanes of a DPP row), rows exchanged through 8 KB LDS buffers, vertical max over (row 2p-1 kept from the previous
//     step, 2p, 2p+1), whole 128-byte pixel lines to HBM.  The 128x128 map never exists.
// Rounding: bias + ReLU + round-to-element commute with max (all monotone), so the result is bit-identical to
// pool(round(relu(conv + bias))) -- what the unfused kernels and the oracle compute.
//
// Workgroup = 4 waves = (channel half h) x (conv row r of the step's two rows); wave tile = 32 ch x 128 px (one whole
// conv row): acc[2][8] x f32x4.  K-step = kernel row (7 taps x 18 ch = 126 contiguous input values, padded to 128 by
// two zero weights), 4 sub-steps of 32; pixel fragments are 8-byte aligned (pixel pitch 36 B): 2 x ds_read_b64 each.
#include <cstdlib>
#include <utility>

#include "conv.hpp"
#include "conv_elem.hpp"

namespace {
using namespace spsp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_2xb64(uint32_t addr) {
    bf16x4 lo, hi;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 8) : "memory");
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_b128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void dma16(const unsigned char* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                     (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

constexpr int kRowB = 264 * 18 * 2;        // 9504 bytes of one native input row
constexpr int kRowPieces = 10;             // 1-KiB LDS-DMA pieces per row (the last one is 18 lanes wide: EXEC-masked)
constexpr int kPitch = kRowB;              // LDS row pitch = the row itself (16-byte aligned): 148 KB in all, so that a fit
                                           // workgroup (13 KB) of another stream can share the CU
constexpr int kRing = 13;                  // rows in the window: 9 of the current step + 4 arriving for the next
constexpr int kRingB = kRing * kPitch;     // 123 552
constexpr int kHRowB = 64 * 128;           // one horizontally pooled row: [64 pool px][64 ch] elements
constexpr int kLds = kRingB + 3 * kHRowB;  // + A, B[2]  = 148 128
constexpr int kWBytes = 64 * 896 * 2;      // the weight matrix, 114 688 B = 112 LDS-DMA pieces
constexpr int kWOff = 28 * 1024;           // prologue: it passes through LDS behind ring rows 0..2 (28 512 B), 1-KiB aligned
static_assert(kWOff >= 3 * kPitch && kWOff + kWBytes <= kLds, "weight staging must fit behind the first three ring rows");

// DPP row controls (a DPP row = 16 lanes = the 16 pixels of one MFMA tile)
constexpr int kDppShl1 = 0x101;            // lane l <- lane l+1
constexpr int kDppShr1 = 0x111;            // lane l <- lane l-1
constexpr int kDppRor1 = 0x121;            // lane l <- lane (l-1) mod 16
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t old, uint32_t src) {      // lanes without a source lane keep `old`
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, 0xf, false);
}
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_relu16(uint32_t x) {      // max(x, +0) on both 16-bit floats of the pair (bf16 or fp16)
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, x), i16x2{0, 0}));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, x), __builtin_bit_cast(u16x2, y)));
}

#ifdef SPSP_CONV_STAMPS
#define SSTAMP(i) if (a.dbg && blockIdx.x == 9 && tid == 0 && st == 3) a.dbg[i] = __builtin_amdgcn_s_memtime()
#define SSTAMPK(i) if (a.dbg && blockIdx.x == 9 && tid == 0) a.dbg[i] = __builtin_amdgcn_s_memtime()
#else
#define SSTAMP(i)
#define SSTAMPK(i)
#endif

struct StemArgs {
    const uint16_t* in;      // native [B][262][264][18]
    const uint16_t* w;       // [64][7*128], K = kernel row * 128 + tap * 18 + channel (+ 2 zero pads per row)
    const float* bias;       // [64]
    uint16_t* out;           // [B][66][66][64]: pooled map with a 1-px zero border (interior written here)
    int bands;               // bands per image
    int rows_per_band;       // conv rows per band: 16, 8 or 4 (a band = rows_per_band / 2 pool rows)
    unsigned int* zero;      // words the FIRST workgroup zeroes: the tile-scheduler / hand-off counters of the forward's later kernels
    int nzero;               //   (this is the forward's first kernel; a memset node in front of it cost ~5 us of every forward)
    uint4* zero2;            // ... and 16-byte words EVERY workgroup zeroes a share of: the polling IEF's hand-off buffers (regressor.hip)
    int nzero2;
    unsigned long long* dbg; // diagnostic builds (-DSPSP_CONV_STAMPS): s_memtime stamps of one workgroup
};

// DIAG (diagnostic builds only, results invalid): 1 = no fragment reads in the K loop, 2 = no MFMAs -- what each stream costs alone
template <bool F16, int DIAG = 0>
__global__ __launch_bounds__(256, 1) void conv1_pool_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const hA = smem + kRingB;                  // pooled conv row 2s   (wave row 0)
    unsigned char* const hB = smem + kRingB + kHRowB;         // pooled conv row 2s+1 (wave row 1), two parities

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x == 0)
        for (int i = tid; i < a.nzero; i += 256) a.zero[i] = 0u;
    for (int i = blockIdx.x * 256 + tid; i < a.nzero2; i += gridDim.x * 256) a.zero2[i] = make_uint4(0u, 0u, 0u, 0u);
    SSTAMPK(0);
    const int h = wave & 1, r = wave >> 1;                    // channel half, conv row within the step
    const int b = blockIdx.x / a.bands, band = blockIdx.x - b * a.bands;
    const int RB = a.rows_per_band;
    const int s0 = band > 0 ? -1 : 0;                         // the halo step: conv rows band*RB-2, band*RB-1 (only the carry row is kept)
    const int nsteps = RB / 2 - s0;
    const int first_row = 2 * (band * RB + 2 * s0);           // native input row of ring row 0
    const unsigned char* const in_b = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * (262 * kRowB) + (size_t)first_row * kRowB;

    // ---- LDS-DMA of one input row (relative index n) into its ring slot: 9 whole pieces + one of 18 lanes (288 bytes):
    //      the slots are packed, a full tenth piece would run into the next slot's row
    const int lane_src = lane * 16;
    auto issue_row = [&](int n) {
        const unsigned char* src = in_b + (size_t)n * kRowB;
        unsigned char* dst = smem + (n % kRing) * kPitch;
#pragma unroll
        for (int p = 0; p < kRowPieces - 1; ++p) dma16(src + p * 1024 + lane_src, dst + p * 1024);
        if (lane < (kRowB - (kRowPieces - 1) * 1024) / 16) dma16(src + (kRowPieces - 1) * 1024 + lane_src, dst + (kRowPieces - 1) * 1024);
    };
    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    // ---- prologue.  The weights reach the registers THROUGH LDS: fetched straight from global as fragments they are 64-byte
    // segments of 16 rows per instruction, twice per workgroup (measured: 21.8 k cycles of prologue with every CU doing it at
    // once).  Here the 114 KB matrix is ONE linear LDS-DMA copy (112 whole-line pieces) into the part of LDS the first four
    // input rows do not use, every wave then picks its 56 fragments with conflict-free ds_read_b128 (rows are 1792 B = 7 bank
    // rows apart, so the 16-byte chunk index is XORed with the row's low 4 bits -- on the DMA source, and on the read), and
    // the remaining six rows of the first step are fetched behind it.
    if (wave < 3) issue_row(wave);                            // ring rows 0..2
    {
        const unsigned char* const wb = reinterpret_cast<const unsigned char*>(a.w);
        for (int p = wave; p < kWBytes / 1024; p += 4) {
            const int L = p * 64 + lane, row = L / 112, c = L - row * 112;
            dma16(wb + (size_t)(row * 112 + (c ^ (row & 15))) * 16, smem + kWOff + p * 1024);
        }
    }
    f32x4 bias_v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 b4 = *reinterpret_cast<const float4*>(a.bias + 32 * h + 16 * i + 4 * fq);
        bias_v[i] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- this wave's weights as MFMA A-fragments, resident for the whole kernel: row 32h + 16i + frow, chunk 4g + fq ----
    bf16x8 wreg[2][28];
    {
        // chunk' = (4g + fq) ^ frow: bits 0-1 fq ^ (frow & 3), bits 2-3 (g & 3) ^ (frow >> 2), the rest g >> 2 -> one base per g & 3
        uint32_t wa[4];
#pragma unroll
        for (int g3 = 0; g3 < 4; ++g3)
            wa[g3] = smem_base + kWOff + (32 * h + frow) * 1792 + 64 * (g3 ^ (frow >> 2)) + 16 * (fq ^ (frow & 3));
        // four fragments AND their wait in one asm statement (hipcc takes an asm output as written when the statement ends), read
        // STRAIGHT INTO accumulation registers (round 5: the MFMAs take them from there; defined in VGPRs hipcc kept half of them
        // there and copied a fragment over in front of every use -- 228 v_accvgpr_write per step)
#define STEM_WR4(I, G)                                                                                    \
        asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%8\n\t"                 \
                     "ds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8\n\ts_waitcnt lgkmcnt(0)" \
                     : "=&a"(wreg[I][G]), "=&a"(wreg[I][(G) + 1]), "=&a"(wreg[I][(G) + 2]), "=&a"(wreg[I][(G) + 3]) \
                     : "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "n"((I) * 16 * 1792 + ((G) >> 2) * 256) : "memory");
        STEM_WR4(0, 0) STEM_WR4(0, 4) STEM_WR4(0, 8) STEM_WR4(0, 12) STEM_WR4(0, 16) STEM_WR4(0, 20) STEM_WR4(0, 24)
        STEM_WR4(1, 0) STEM_WR4(1, 4) STEM_WR4(1, 8) STEM_WR4(1, 12) STEM_WR4(1, 16) STEM_WR4(1, 20) STEM_WR4(1, 24)
#undef STEM_WR4
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                          // every wave has its fragments: the staging area is free
    for (int n = 3 + wave; n < 9; n += 4) issue_row(n);       // ring rows 3..8
    // the row above the image is padding: the carry of the first step of the top band is all zeros (post-ReLU values are >= 0)
    if (s0 == 0)
        for (int i = tid; i < kHRowB / 16; i += 256) reinterpret_cast<uint4*>(hB + kHRowB)[i] = make_uint4(0u, 0u, 0u, 0u);

    // MFMA tile j = (span q = j >> 1, type t = j & 1) holds the conv columns 32q + 2t + {0,1, 4,5, 8,9, ...}: lane frow ->
    // column 32q + 2t + 4 (frow >> 1) + (frow & 1).  With consecutive columns in consecutive lanes (72-byte pixel pitch, 8-byte
    // bank slots s = 9 frow + 2 fq mod 32) lanes (0, fq=0) and (14, fq=1) of a half-wave always meet on a bank: every
    // ds_read_b64 took two passes (measured: SQ_LDS_BANK_CONFLICT = 48 % of the LDS cycles, the K loop LDS-bound).  Taking
    // the columns = 0,1 mod 4 in one tile and = 2,3 mod 4 in the other makes the 32 slots of a half-wave a permutation
    // (the slot set {r : r mod 4 in {0,1}} and its shift by 2 partition Z32), for free: the tile's columns are a per-lane
    // constant plus an immediate.  It also leaves every lane with a pooled column to produce (see the epilogue).
    const uint32_t lane_off = 72 * (4 * (frow >> 1) + (frow & 1)) + 16 * fq;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SSTAMPK(1);

    for (int st = 0; st < nsteps; ++st) {
        SSTAMP(2);
        const int s = s0 + st;                                // conv rows band*RB + 2s, + 2s + 1  ->  pool row band*RB/2 + s
        const int n0 = 4 * st;                                // ring row index of this step's first input row
        if (st + 1 < nsteps) issue_row(n0 + 9 + wave);        // the 4 new rows of the next step, one per wave

        // the accumulators START at the folded BN bias (accumulator element e of tile (i, j) = channel 32 h + 16 i + 4 fq + e):
        // 64 additions less in the pooling phase, which a lone wave issues at 4-5 cycles each
        // (since round 5 not by moves: sub-step 0's MFMAs take the bias registers as their C operand -- conv_elem.hpp, mfma_agpr_from)
        f32x4 acc[2][8];

        // fragment base address per kernel row: input row 2 (conv row) + kr = ring row n0 + 2r + kr
        uint32_t xa[7];
#pragma unroll
        for (int kr = 0; kr < 7; ++kr) xa[kr] = smem_base + (uint32_t)(((n0 + 2 * r + kr) % kRing) * kPitch) + lane_off;

        bf16x8 xf[8];
        // The K loop as ONE stream of 224 fragment / MFMA-pair slots (sub-step G = 4 KR + KS, tile J; slot number 8 G + J).
        // Slot n: issue the two ds_read_b64 of the fragment SEVEN slots ahead into the register quad the previous slot has
        // just consumed (xf[(J + 7) & 7]), wait until all but the 14 youngest LDS reads have returned -- exactly the
        // fragment of slot n -- and issue its two MFMAs.  224 cycles of look-ahead at <= 14 reads in flight (lgkmcnt counts
        // to 15); the per-sub-step form (all reads of G+1 during G, one lgkmcnt(0) at its end) left ~85 cycles of LDS
        // latency exposed in every 256-cycle sub-step (K loop 9.5 k cycles for 7.2 k of MFMA).  Counted lgkmcnt waits need
        // the counter to see LDS reads only: no scalar load may be in flight (they return out of order) -- the lgkmcnt(0)
        // in front drains any, and the loop body issues none (checked in the ISA).
        // Every sched_barrier pins the order (hipcc otherwise gathers the reads and sinks the register-only MFMAs).
#define STEM_SB __builtin_amdgcn_sched_barrier(0);
#define STEM_RD(J, KR, KS) xf[J] = lds_read_2xb64<64 * (KS) + 2304 * ((J) >> 1) + 144 * ((J) & 1)>(xa[KR]);
#define STEM_SLOT(G, J, KRA, KSA, AHEAD, WAITN)                                                            \
        if (AHEAD && DIAG != 1) { STEM_RD(((J) + 7) & 7, KRA, KSA) }                                       \
        if (DIAG != 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(WAITN) : "memory");                      \
        STEM_SB                                                                                           \
        if (DIAG != 2) {                                                                                  \
            if constexpr ((G) == 0) {                                                                     \
                Elem<F16>::mfma_agpr_from(acc[0][J], wreg[0][G], xf[J], bias_v[0]);                       \
                Elem<F16>::mfma_agpr_from(acc[1][J], wreg[1][G], xf[J], bias_v[1]);                       \
            } else {                                                                                      \
                Elem<F16>::mfma_agpr(acc[0][J], wreg[0][G], xf[J]);                                       \
                Elem<F16>::mfma_agpr(acc[1][J], wreg[1][G], xf[J]);                                       \
            }                                                                                             \
        } else { if constexpr ((G) == 0) { acc[0][J] = bias_v[0]; acc[1][J] = bias_v[1]; } asm volatile("" : "+v"(xf[J])); }  \
        STEM_SB
        // sub-step G; (KR1, KS1) = coordinates of sub-step G+1 (slots J >= 1 read ahead into it, slot 0 still into G)
#define STEM_STEP(G, KR0, KS0, KR1, KS1)                                                                  \
        STEM_SLOT(G, 0, KR0, KS0, true, 14) STEM_SLOT(G, 1, KR1, KS1, true, 14) STEM_SLOT(G, 2, KR1, KS1, true, 14) \
        STEM_SLOT(G, 3, KR1, KS1, true, 14) STEM_SLOT(G, 4, KR1, KS1, true, 14) STEM_SLOT(G, 5, KR1, KS1, true, 14) \
        STEM_SLOT(G, 6, KR1, KS1, true, 14) STEM_SLOT(G, 7, KR1, KS1, true, 14)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STEM_SB
        STEM_RD(0, 0, 0) STEM_RD(1, 0, 0) STEM_RD(2, 0, 0) STEM_RD(3, 0, 0) STEM_RD(4, 0, 0) STEM_RD(5, 0, 0) STEM_RD(6, 0, 0)
        STEM_SB
        STEM_STEP(0, 0, 0, 0, 1)
        STEM_STEP(1, 0, 1, 0, 2)
        STEM_STEP(2, 0, 2, 0, 3)
        STEM_STEP(3, 0, 3, 1, 0)
        STEM_STEP(4, 1, 0, 1, 1)
        STEM_STEP(5, 1, 1, 1, 2)
        STEM_STEP(6, 1, 2, 1, 3)
        STEM_STEP(7, 1, 3, 2, 0)
        STEM_STEP(8, 2, 0, 2, 1)
        STEM_STEP(9, 2, 1, 2, 2)
        STEM_STEP(10, 2, 2, 2, 3)
        STEM_STEP(11, 2, 3, 3, 0)
        STEM_STEP(12, 3, 0, 3, 1)
        STEM_STEP(13, 3, 1, 3, 2)
        STEM_STEP(14, 3, 2, 3, 3)
        STEM_STEP(15, 3, 3, 4, 0)
        STEM_STEP(16, 4, 0, 4, 1)
        STEM_STEP(17, 4, 1, 4, 2)
        STEM_STEP(18, 4, 2, 4, 3)
        STEM_STEP(19, 4, 3, 5, 0)
        STEM_STEP(20, 5, 0, 5, 1)
        STEM_STEP(21, 5, 1, 5, 2)
        STEM_STEP(22, 5, 2, 5, 3)
        STEM_STEP(23, 5, 3, 6, 0)
        STEM_STEP(24, 6, 0, 6, 1)
        STEM_STEP(25, 6, 1, 6, 2)
        STEM_STEP(26, 6, 2, 6, 3)
        STEM_SLOT(27, 0, 6, 3, true, 14)
        STEM_SLOT(27, 1, 6, 3, false, 12) STEM_SLOT(27, 2, 6, 3, false, 10) STEM_SLOT(27, 3, 6, 3, false, 8)
        STEM_SLOT(27, 4, 6, 3, false, 6)  STEM_SLOT(27, 5, 6, 3, false, 4)  STEM_SLOT(27, 6, 6, 3, false, 2)
        STEM_SLOT(27, 7, 6, 3, false, 0)
#undef STEM_SLOT
#undef STEM_STEP
#undef STEM_RD
#undef STEM_SB

        // (round 5: the MFMAs are inline asm with the weight fragment read straight from the accumulation file -- hipcc copied 144
        //  AGPR-resident fragments back per step, into a loop that is bound by its issue slots; nothing inside an asm is padded, so the
        //  wait between the last MFMA and the first VALU read of its result is spelled out)
        mfma_drain8(acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[0][4], acc[0][5], acc[0][6], acc[0][7]);
        asm volatile("" : "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[1][4]), "+v"(acc[1][5]), "+v"(acc[1][6]), "+v"(acc[1][7]));
        __builtin_amdgcn_sched_barrier(0);
        SSTAMP(3);
        // ---- horizontal 3 -> 1 max in registers.  Round to the element type FIRST, then ReLU on the packed pairs (all monotone:
        // max commutes with them; as SIGNED 16-bit integers every negative value, -0 included, is below +0 = 0, so
        // v_pk_max_i16 against 0 is the ReLU of both element types at half the instructions), so that the pool runs on packed
        // registers with v_pk_max_u16 -- post-ReLU values are >= +0, and for bf16 and fp16 alike the numeric order of
        // non-negative values is the order of their 16-bit patterns.
        // Span q: tile a = type 0 (lane n: column 4 (n>>1) + (n&1)), tile b = type 1 (the same + 2).  Pool column 16q + n:
        //   even n = 2m:  max(col 4m-1, 4m, 4m+1) = max(b[n-1], a[n], a[n+1])      (b[-1] = lane 15 of the previous span's b;
        //   odd  n = 2m+1: max(col 4m+1, 4m+2, 4m+3) = max(a[n], b[n-1], b[n])      column -1 is padding: a[0] stands in)
        // so EVERY lane ends with one pooled column: three DPP moves, one select and two packed max per register.
        unsigned char* const hrow = r == 0 ? hA : hB + (s & 1) * kHRowB;
        const bool odd = frow & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t P[8][2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 c = acc[i][j];
                P[j][0] = pk_relu16(Elem<F16>::pack2(c[0], c[1]));
                P[j][1] = pk_relu16(Elem<F16>::pack2(c[2], c[3]));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t m[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const uint32_t va = P[2 * q][k], vb = P[2 * q + 1][k];
                    const uint32_t wrap = q > 0 ? dpp<kDppRor1>(va, P[q > 0 ? 2 * q - 1 : 0][k]) : va;   // lane 0 <- lane 15 of the previous span's b
                    const uint32_t bl = dpp<kDppShr1>(wrap, vb);                                         // b[n-1]; lane 0 keeps `wrap`
                    const uint32_t ar = dpp<kDppShl1>(va, va);                                           // a[n+1]
                    m[k] = pk_max_u16(pk_max_u16(va, bl), odd ? vb : ar);
                }
                const int px = 16 * q + frow;
                // [px][64 ch] with the 8-byte slot index XORed by 2 (px & 7)
                const int slot = (8 * h + 4 * i + fq) ^ (2 * (px & 7));
                *reinterpret_cast<uint2*>(hrow + px * 128 + slot * 8) = make_uint2(m[0], m[1]);
            }
        }
        SSTAMP(4);
        // next step's rows have landed (each wave waits for its own piece), pooled rows are written: meet
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        SSTAMP(5);
        __syncthreads();
        SSTAMP(6);
        if (s >= 0) {
            // ---- vertical max over conv rows 2p-1 (carry), 2p, 2p+1 and the store of pool row p: thread -> pixel tid/4, 32 bytes
            const int px = tid >> 2, q = tid & 3;
            const int prow = band * (RB / 2) + s;
            uint16_t* const orow = a.out + (size_t)b * (66 * 66 * 64) + (size_t)(prow + 1) * (66 * 64) + (size_t)(px + 1) * 64;
            const unsigned char* const carry = hB + ((s + 1) & 1) * kHRowB;
            const unsigned char* const cur1 = hB + (s & 1) * kHRowB;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int c = 2 * q + c2;
                const int off = px * 128 + (((2 * c) ^ (2 * (px & 7))) * 8);
                const uint4 va = *reinterpret_cast<const uint4*>(hA + off);
                const uint4 vb = *reinterpret_cast<const uint4*>(cur1 + off);
                const uint4 vc = *reinterpret_cast<const uint4*>(carry + off);
                // post-ReLU values are >= +0: for bf16 and fp16 alike the numeric order is the order of the 16-bit patterns
                auto mx = [](uint32_t x, uint32_t y, uint32_t z) -> uint32_t { return pk_max_u16(pk_max_u16(x, y), z); };
                *reinterpret_cast<uint4*>(orow + c * 8) = make_uint4(mx(va.x, vb.x, vc.x), mx(va.y, vb.y, vc.y), mx(va.z, vb.z, vc.z), mx(va.w, vb.w, vc.w));
            }
        }
        SSTAMP(7);
        __syncthreads();          // hA and the carry buffer are free for the next step
        SSTAMP(8);
    }
    SSTAMPK(9);
}


// ==== round 5: the same kernel on v_mfma_f32_32x32x16 ==========================================================================
// The K loop above is bound by its ISSUE slots: with one wave per SIMD an MFMA holds the vector issue for 8 of its 16 cycles, and a
// slot's two ds_read_b64, its counted wait and its two MFMAs take ~41 cycles where the matrix pipe needs 32 (stamps: 9.3 k cycles per
// step for 7.2 k of MFMA; round 4 showed that nothing can ride in that loop).  The 32 x 32 x 16 shape does the same FLOPs with ONE
// MFMA per 32 cycles: half the MFMA issue, the same two reads -- the slot fits.  What changes:
//   * wave tile = 32 channels x 128 conv columns = four 32 x 32 tiles (16 accumulator registers each); a K sub-step is 16 deep
//     (56 per conv row), the 56 weight fragments are 224 AGPRs as before;
//   * lane l of a pixel tile holds conv column 2 (l & 15) + ((l >> 4) & 1): even columns in the even 16-lane rows, odd columns in the
//     odd rows (any bijection of 32 lanes onto 32 consecutive columns is conflict-free for the ds_read_b64 pair: 8-byte bank slot
//     9 * column mod 32) -- so that the horizontal 3 -> 1 pool, which needs columns 2p - 1, 2p, 2p + 1 for pooled column p, finds
//     2p in the even row, and 2p + 1 / 2p - 1 in the partner row: one v_permlane16_swap per register PAIR + one DPP row shift;
//   * the weight rows are permuted so that lane (column, hh = l >> 5) ends with channels 32 h + 16 hh + 0..15 in its 16 registers;
//   * the pooled rows go to the same LDS row image ([pooled column][64 channels], 16-byte chunk XOR column & 7) from the even
//     rows' lanes; the vertical pool and the stores are the kernel above's.
// The sums are the same products in another order (56 MFMAs of depth 16 per output instead of 28 of depth 32): not bit-identical to
// the 16 x 16 x 32 form; against the rounding-emulating oracle in the stage tests.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kSlots32 = 224;              // 56 sub-steps x 4 pixel tiles per conv row

// slot N = (sub-step G = N >> 2, tile T = N & 3): issue the fragment of slot N + 7 into the quad slot N - 1 has consumed, wait for
// slot N's own fragment (two reads per fragment, seven fragments behind it), issue its MFMA
template <bool F16, int N>
__device__ __forceinline__ void stem32_slot(f32x16 (&acc)[4], bf16x8 (&xf)[8], const bf16x8 (&wreg)[56], const uint32_t (&xa)[7], const f32x16& bias) {
    constexpr int G = N >> 2, T = N & 3;
    if constexpr (N + 7 < kSlots32) {
        constexpr int M = N + 7, G2 = M >> 2, T2 = M & 3;
        xf[M & 7] = lds_read_2xb64<32 * (G2 & 7) + 2304 * T2>(xa[G2 >> 3]);
    }
    // a lone wave issues an instruction every ~8 cycles whatever its kind, so a slot of two reads, a wait and an MFMA is 32 cycles of
    // issue for 32 cycles of matrix pipe: ONE wait per two slots (for both slots' fragments: six fragments stay in flight) makes room
    constexpr int behind = kSlots32 - 1 - N < 7 ? kSlots32 - 1 - N : 7;
    if constexpr ((N & 1) == 0 || behind < 7) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(behind > 0 ? 2 * (behind - 1) : 0) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G == 0) Elem<F16>::mfma32_agpr_from(acc[T], wreg[0], xf[N & 7], bias);     // the chain starts from the bias
    else Elem<F16>::mfma32_agpr(acc[T], wreg[G], xf[N & 7]);
    __builtin_amdgcn_sched_barrier(0);
}
template <bool F16, int... Is>
__device__ __forceinline__ void stem32_slots(f32x16 (&acc)[4], bf16x8 (&xf)[8], const bf16x8 (&wreg)[56], const uint32_t (&xa)[7], const f32x16& bias,
                                             std::integer_sequence<int, Is...>) {
    (stem32_slot<F16, Is>(acc, xf, wreg, xa, bias), ...);
}

template <bool F16>
__global__ __launch_bounds__(256, 1) void conv1_pool32_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const hA = smem + kRingB;                  // pooled conv row 2s   (wave row 0)
    unsigned char* const hB = smem + kRingB + kHRowB;         // pooled conv row 2s+1 (wave row 1), two parities

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x == 0)
        for (int i = tid; i < a.nzero; i += 256) a.zero[i] = 0u;
    for (int i = blockIdx.x * 256 + tid; i < a.nzero2; i += gridDim.x * 256) a.zero2[i] = make_uint4(0u, 0u, 0u, 0u);
    SSTAMPK(0);
    const int h = wave & 1, r = wave >> 1;                    // channel half, conv row within the step
    const int b = blockIdx.x / a.bands, band = blockIdx.x - b * a.bands;
    const int RB = a.rows_per_band;
    const int s0 = band > 0 ? -1 : 0;
    const int nsteps = RB / 2 - s0;
    const int first_row = 2 * (band * RB + 2 * s0);
    const unsigned char* const in_b = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * (262 * kRowB) + (size_t)first_row * kRowB;
    const int lane_src = lane * 16;
    auto issue_row = [&](int n) {
        const unsigned char* src = in_b + (size_t)n * kRowB;
        unsigned char* dst = smem + (n % kRing) * kPitch;
#pragma unroll
        for (int p = 0; p < kRowPieces - 1; ++p) dma16(src + p * 1024 + lane_src, dst + p * 1024);
        if (lane < (kRowB - (kRowPieces - 1) * 1024) / 16) dma16(src + (kRowPieces - 1) * 1024 + lane_src, dst + (kRowPieces - 1) * 1024);
    };
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    const int rho = lane & 31, kc = lane >> 5;                // MFMA row of this lane's weight fragments; which 8 of a sub-step's 16 k
    const int hh = kc;                                        // ... and, in the results, which 16 of the wave's 32 channels
    const int lcol = 2 * (lane & 15) + ((lane >> 4) & 1);     // this lane's conv column within a 32-column tile
    // ---- prologue: the weight matrix through LDS (one linear LDS-DMA copy, 16-byte chunk index XOR row & 15), as above
    if (wave < 3) issue_row(wave);
    {
        const unsigned char* const wb = reinterpret_cast<const unsigned char*>(a.w);
        for (int p = wave; p < kWBytes / 1024; p += 4) {
            const int L = p * 64 + lane, row = L / 112, c = L - row * 112;
            dma16(wb + (size_t)(row * 112 + (c ^ (row & 15))) * 16, smem + kWOff + p * 1024);
        }
    }
    // accumulator register e of lane (column, hh) = channel 32 h + 16 hh + e
    f32x16 bias_v;
    {
        const float4* bp = reinterpret_cast<const float4*>(a.bias + 32 * h + 16 * hh);
        const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        bias_v = f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- this wave's 56 weight fragments, straight into accumulation registers: MFMA row rho is output channel
    //      32 h + 16 ((rho >> 2) & 1) + 4 (rho >> 3) + (rho & 3) (the 32 x 32 result layout puts row (e & 3) + 8 (e >> 2) + 4 hh in register e);
    //      sub-step g: logical 16-byte chunk 2 g + kc of that row = physical chunk 16 (g >> 3) + ((2 (g & 7) + kc) ^ (row & 15))
    bf16x8 wreg[56];
    {
        const int n = 32 * h + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
        uint32_t wa[8];
#pragma unroll
        for (int v8 = 0; v8 < 8; ++v8) wa[v8] = smem_base + kWOff + n * 1792 + (((2 * v8 + kc) ^ (n & 15)) * 16);
#define STEM32_WR4(G)                                                                                              \
        asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%9\n\t"                          \
                     "ds_read_b128 %2, %6 offset:%10\n\tds_read_b128 %3, %7 offset:%11\n\ts_waitcnt lgkmcnt(0)"    \
                     : "=&a"(wreg[G]), "=&a"(wreg[(G) + 1]), "=&a"(wreg[(G) + 2]), "=&a"(wreg[(G) + 3])            \
                     : "v"(wa[(G) & 7]), "v"(wa[((G) + 1) & 7]), "v"(wa[((G) + 2) & 7]), "v"(wa[((G) + 3) & 7]),   \
                       "n"(256 * ((G) >> 3)), "n"(256 * (((G) + 1) >> 3)), "n"(256 * (((G) + 2) >> 3)), "n"(256 * (((G) + 3) >> 3)) : "memory");
        STEM32_WR4(0) STEM32_WR4(4) STEM32_WR4(8) STEM32_WR4(12) STEM32_WR4(16) STEM32_WR4(20) STEM32_WR4(24)
        STEM32_WR4(28) STEM32_WR4(32) STEM32_WR4(36) STEM32_WR4(40) STEM32_WR4(44) STEM32_WR4(48) STEM32_WR4(52)
#undef STEM32_WR4
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                          // every wave has its fragments: the staging area is free
    for (int n = 3 + wave; n < 9; n += 4) issue_row(n);       // ring rows 3..8
    if (s0 == 0)
        for (int i = tid; i < kHRowB / 16; i += 256) reinterpret_cast<uint4*>(hB + kHRowB)[i] = make_uint4(0u, 0u, 0u, 0u);

    // pixel fragment of tile T, sub-step (kernel row kr, gs): the lane's column, 16 bytes at k = 16 gs + 8 kc of the kernel row's
    // 126 (+2) contiguous values: two ds_read_b64 (the 72-byte column pitch keeps 8-byte alignment only)
    const uint32_t lane_off = 72 * lcol + 16 * kc;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SSTAMPK(1);

    for (int st = 0; st < nsteps; ++st) {
        SSTAMP(2);
        const int s = s0 + st;
        const int n0 = 4 * st;
        if (st + 1 < nsteps) issue_row(n0 + 9 + wave);

        f32x16 acc[4];
        uint32_t xa[7];
#pragma unroll
        for (int kr = 0; kr < 7; ++kr) xa[kr] = smem_base + (uint32_t)(((n0 + 2 * r + kr) % kRing) * kPitch) + lane_off;
        bf16x8 xf[8];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (counted waits below: LDS reads only may be in flight)
        __builtin_amdgcn_sched_barrier(0);
        // slots 0..6: (g 0, tiles 0..3), (g 1, tiles 0..2)
        xf[0] = lds_read_2xb64<0>(xa[0]); xf[1] = lds_read_2xb64<2304>(xa[0]); xf[2] = lds_read_2xb64<4608>(xa[0]); xf[3] = lds_read_2xb64<6912>(xa[0]);
        xf[4] = lds_read_2xb64<32>(xa[0]); xf[5] = lds_read_2xb64<32 + 2304>(xa[0]); xf[6] = lds_read_2xb64<32 + 4608>(xa[0]);
        __builtin_amdgcn_sched_barrier(0);
        stem32_slots<F16>(acc, xf, wreg, xa, bias_v, std::make_integer_sequence<int, kSlots32>{});
        // (the wait between the last MFMA and the first VALU read of its result: nothing inside an asm is padded)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        __builtin_amdgcn_sched_barrier(0);
        SSTAMP(3);
        // ---- round to the element type, ReLU on the packed pairs, horizontal 3 -> 1 max.  Pooled column 16 T + p (p = lane & 15) =
        // max(col 2p - 1, 2p, 2p + 1) = max(O[p - 1], E[p], O[p]) with E / O = the tile's even / odd columns, which sit in the even / odd
        // 16-lane rows.  ONE v_permlane16_swap of the packed registers k and k + 4 (channel pairs 2k.. and 2k + 8..) puts E of both
        // into one register and O of both into the other -- rows 0 / 2 for pair k, rows 1 / 3 for pair k + 4 -- so that every lane pools
        // (the first form swapped a register with itself: half the lanes idle, two copies per swap, 2.2 k cycles); O[p - 1] by a DPP
        // row shift, lane 0 of a row taking O[15] of the previous tile (column -1 is padding: E[0] stands in; post-ReLU values are
        // >= +0 and ordered like their bit patterns).  A lane ends with 8 consecutive channels of its pooled column: one 16-byte store.
        unsigned char* const hrow = r == 0 ? hA : hB + (s & 1) * kHRowB;
        uint32_t oprev[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            uint32_t m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t xa_ = pk_relu16(Elem<F16>::pack2(acc[T][2 * k], acc[T][2 * k + 1]));
                const uint32_t xb_ = pk_relu16(Elem<F16>::pack2(acc[T][2 * k + 8], acc[T][2 * k + 9]));
                const auto sw = __builtin_amdgcn_permlane16_swap(xa_, xb_, false, false);
                const uint32_t e = sw[0], o = sw[1];                                 // E / O of pair k (rows 0, 2) and k + 4 (rows 1, 3)
                const uint32_t wrap = T > 0 ? dpp<kDppRor1>(e, oprev[k]) : e;       // lane 0 of a row <- lane 15 of the previous tile's O
                const uint32_t bl = dpp<kDppShr1>(wrap, o);                          // O[p - 1]; lane 0 keeps `wrap`
                m[k] = pk_max_u16(pk_max_u16(e, o), bl);
                oprev[k] = o;
            }
            const int px = 16 * T + (lane & 15);
            const int c = 4 * h + 2 * hh + ((lane >> 4) & 1);                        // 16-byte chunk = 8 channels
            *reinterpret_cast<uint4*>(hrow + px * 128 + ((c ^ (px & 7)) * 16)) = make_uint4(m[0], m[1], m[2], m[3]);
        }
        SSTAMP(4);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        SSTAMP(5);
        __syncthreads();
        SSTAMP(6);
        if (s >= 0) {
            const int px = tid >> 2, q = tid & 3;
            const int prow = band * (RB / 2) + s;
            uint16_t* const orow = a.out + (size_t)b * (66 * 66 * 64) + (size_t)(prow + 1) * (66 * 64) + (size_t)(px + 1) * 64;
            const unsigned char* const carry = hB + ((s + 1) & 1) * kHRowB;
            const unsigned char* const cur1 = hB + (s & 1) * kHRowB;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int c = 2 * q + c2;
                const int off = px * 128 + (((2 * c) ^ (2 * (px & 7))) * 8);
                const uint4 va = *reinterpret_cast<const uint4*>(hA + off);
                const uint4 vb = *reinterpret_cast<const uint4*>(cur1 + off);
                const uint4 vc = *reinterpret_cast<const uint4*>(carry + off);
                auto mx = [](uint32_t x, uint32_t y, uint32_t z) -> uint32_t { return pk_max_u16(pk_max_u16(x, y), z); };
                *reinterpret_cast<uint4*>(orow + c * 8) = make_uint4(mx(va.x, vb.x, vc.x), mx(va.y, vb.y, vc.y), mx(va.z, vb.z, vc.z), mx(va.w, vb.w, vc.w));
            }
        }
        SSTAMP(7);
        __syncthreads();
        SSTAMP(8);
    }
    SSTAMPK(9);
}


// ==== round 5, second form: the pooling RIDES in the K loop ======================================================================
// conv1_pool32_kernel spends 1.9 k of a step's 10 k cycles outside the MFMAs (stamps: K loop 8.2 k | horizontal pool 0.95 k | LDS wait
// 0.34 k | barrier | vertical pool + stores 0.4 k | barrier).  A 32 x 32 x 16 MFMA occupies the matrix pipe for 32 cycles and the
// wave's issue for 8, so -- unlike the 16 x 16 x 32 loop, where round 4 measured this zero-sum -- other instructions fit beside it.
// What makes it possible without a second accumulator set:
//   * the four pixel tiles run as two PAIRS (tiles 0, 1 over all 56 sub-steps in slots 0..111, then tiles 2, 3): pair A's sums are
//     final at slot 112 and not needed before the next step's slot 0, pair B's from the step's end to the next step's slot 112; two
//     chains interleaved keep dependent MFMAs 64 cycles apart;
//   * step s' K loop carries: H_B(s - 1) -- the horizontal pool of pair B of the previous step -- in slots 2..44, H_A(s) in slots
//     118..160, V_lo(s - 1) -- the vertical max and the store of pooled columns 0..31 -- at slots 52 / 64 and V_hi(s - 2) (columns
//     32..63) at slots 70 / 82: every LDS hand-off (H writes -> V reads) then crosses exactly ONE barrier, the step's own (which
//     also publishes the input ring rows), with the pooled-row buffers two deep by step parity; the row above (conv row 2 s' - 1),
//     which the first form kept in a third LDS row, is carried in registers: it is what the same thread read one step before;
//   * the extra LDS instructions are inline asm like the fragment reads: LDS returns in order, so the slots' counted lgkmcnt waits
//     (at most 14 outstanding) cover anything issued eight slots earlier; an extra instruction inside the window only makes a wait
//     stricter.  hipcc never sees an LDS access here and adds no wait of its own.
// Same products in the same order per output as conv1_pool32_kernel: bit-identical (test_stem_...).
constexpr int kLds32r = kRingB + 4 * kHRowB;                  // hA[2], hB[2] by step parity: 156 320 B
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N> struct Slot32r {                             // slot N -> (sub-step G, tile T), pair-major
    static constexpr int P = N / 112, J = N % 112, G = J >> 1, T = 2 * P + (J & 1);
};

template <bool F16, int N, typename Ride>
__device__ __forceinline__ void stem32r_slot(f32x16 (&acc)[4], bf16x8 (&xf)[8], const bf16x8 (&wreg)[56], const uint32_t (&xa)[7], const f32x16& bias, Ride& ride) {
    constexpr int G = Slot32r<N>::G, T = Slot32r<N>::T;
    if constexpr (N + 7 < kSlots32) {
        constexpr int M = N + 7, G2 = Slot32r<M>::G, T2 = Slot32r<M>::T;
        // (two ds_read_b64: as ONE ds_read2_b64 per fragment -- an address register per (kernel row, tile), the sub-step in the 8-bit
        // offsets -- the K loop took 15.1 k cycles instead of 8.8 k: the pair is served as two conflicting passes)
        xf[M & 7] = lds_read_2xb64<32 * (G2 & 7) + 2304 * T2>(xa[G2 >> 3]);
    }
    constexpr int behind = kSlots32 - 1 - N < 7 ? kSlots32 - 1 - N : 7;                      // (one wait per two slots: conv1_pool32_kernel)
    if constexpr ((N & 1) == 0 || behind < 7) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(behind > 0 ? 2 * (behind - 1) : 0) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (G == 0) Elem<F16>::mfma32_agpr_from(acc[T], wreg[0], xf[N & 7], bias);
    else Elem<F16>::mfma32_agpr(acc[T], wreg[G], xf[N & 7]);
    __builtin_amdgcn_sched_barrier(0);
    ride(std::integral_constant<int, N>{});
    __builtin_amdgcn_sched_barrier(0);
}
template <bool F16, typename Ride, int... Is>
__device__ __forceinline__ void stem32r_slots(f32x16 (&acc)[4], bf16x8 (&xf)[8], const bf16x8 (&wreg)[56], const uint32_t (&xa)[7], const f32x16& bias, Ride& ride,
                                              std::integer_sequence<int, Is...>) {
    (stem32r_slot<F16, Is>(acc, xf, wreg, xa, bias, ride), ...);
}

template <bool F16>
__global__ __launch_bounds__(256, 1) void conv1_pool32r_kernel(const StemArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.x == 0)
        for (int i = tid; i < a.nzero; i += 256) a.zero[i] = 0u;
    for (int i = blockIdx.x * 256 + tid; i < a.nzero2; i += gridDim.x * 256) a.zero2[i] = make_uint4(0u, 0u, 0u, 0u);
    SSTAMPK(0);
    const int h = wave & 1, r = wave >> 1;
    const int b = blockIdx.x / a.bands, band = blockIdx.x - b * a.bands;
    const int RB = a.rows_per_band;
    const int s0 = band > 0 ? -1 : 0;
    const int nsteps = RB / 2 - s0;
    const int first_row = 2 * (band * RB + 2 * s0);
    const unsigned char* const in_b = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * (262 * kRowB) + (size_t)first_row * kRowB;
    const int lane_src = lane * 16;
    auto issue_row = [&](int n) {
        const unsigned char* src = in_b + (size_t)n * kRowB;
        unsigned char* dst = smem + (n % kRing) * kPitch;
#pragma unroll
        for (int p = 0; p < kRowPieces - 1; ++p) dma16(src + p * 1024 + lane_src, dst + p * 1024);
        if (lane < (kRowB - (kRowPieces - 1) * 1024) / 16) dma16(src + (kRowPieces - 1) * 1024 + lane_src, dst + (kRowPieces - 1) * 1024);
    };
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    const int rho = lane & 31, kc = lane >> 5;
    const int hh = kc;
    const int lcol = 2 * (lane & 15) + ((lane >> 4) & 1);
    // ---- prologue: conv1_pool32_kernel's
    if (wave < 3) issue_row(wave);
    {
        const unsigned char* const wb = reinterpret_cast<const unsigned char*>(a.w);
        for (int p = wave; p < kWBytes / 1024; p += 4) {
            const int L = p * 64 + lane, row = L / 112, c = L - row * 112;
            dma16(wb + (size_t)(row * 112 + (c ^ (row & 15))) * 16, smem + kWOff + p * 1024);
        }
    }
    f32x16 bias_v;
    {
        const float4* bp = reinterpret_cast<const float4*>(a.bias + 32 * h + 16 * hh);
        const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
        bias_v = f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bf16x8 wreg[56];
    {
        const int n = 32 * h + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
        uint32_t wa[8];
#pragma unroll
        for (int v8 = 0; v8 < 8; ++v8) wa[v8] = smem_base + kWOff + n * 1792 + (((2 * v8 + kc) ^ (n & 15)) * 16);
#define STEM32_WR4(G)                                                                                              \
        asm volatile("ds_read_b128 %0, %4 offset:%8\n\tds_read_b128 %1, %5 offset:%9\n\t"                          \
                     "ds_read_b128 %2, %6 offset:%10\n\tds_read_b128 %3, %7 offset:%11\n\ts_waitcnt lgkmcnt(0)"    \
                     : "=&a"(wreg[G]), "=&a"(wreg[(G) + 1]), "=&a"(wreg[(G) + 2]), "=&a"(wreg[(G) + 3])            \
                     : "v"(wa[(G) & 7]), "v"(wa[((G) + 1) & 7]), "v"(wa[((G) + 2) & 7]), "v"(wa[((G) + 3) & 7]),   \
                       "n"(256 * ((G) >> 3)), "n"(256 * (((G) + 1) >> 3)), "n"(256 * (((G) + 2) >> 3)), "n"(256 * (((G) + 3) >> 3)) : "memory");
        STEM32_WR4(0) STEM32_WR4(4) STEM32_WR4(8) STEM32_WR4(12) STEM32_WR4(16) STEM32_WR4(20) STEM32_WR4(24)
        STEM32_WR4(28) STEM32_WR4(32) STEM32_WR4(36) STEM32_WR4(40) STEM32_WR4(44) STEM32_WR4(48) STEM32_WR4(52)
#undef STEM32_WR4
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    for (int n = 3 + wave; n < 9; n += 4) issue_row(n);
    const uint32_t lane_off = 72 * lcol + 16 * kc;

    // ---- what rides.  Pooled rows in LDS: row image [64 pooled columns][64 channels], 16-byte chunk c of column px at px * 128 +
    //      ((c ^ (px & 7)) * 16); hA[parity] at kRingB + parity * kHRowB (conv row 2 s', waves r = 0), hB[parity] two rows further (2 s' + 1)
    // H: this lane's store address for tile T of the step of parity p: + p * kHRowB + T * 2048
    const uint32_t hw0 = smem_base + kRingB + (r ? 2 * kHRowB : 0) + (lane & 15) * 128 + (((4 * h + 2 * hh + ((lane >> 4) & 1)) ^ (lane & 7)) * 16);
    // V: thread -> pooled column 32 * half + tid / 8, chunk tid & 7 (8 channels): one 16-byte read per row, one 16-byte store
    const int vpx = tid >> 3, vc = tid & 7;
    const uint32_t vr0 = smem_base + kRingB + vpx * 128 + ((vc ^ (vpx & 7)) * 16);            // hA[0], columns 0..31; hB: + 2 * kHRowB; columns 32..63: + 4096
    uint16_t* const vout = a.out + (size_t)b * (66 * 66 * 64) + (size_t)(band * (RB / 2) + 1) * (66 * 64) + (size_t)(vpx + 1) * 64 + vc * 8;
    uint32_t oprev[4] = {0u, 0u, 0u, 0u};
    u32x4 carry_lo = {0u, 0u, 0u, 0u}, carry_hi = {0u, 0u, 0u, 0u};     // conv row 2 s' - 1, pooled horizontally (image top: zeros -- post-ReLU values are >= 0)
    u32x4 va, vb;
    f32x16 acc[4];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[2][e] = 0.0f; acc[3][e] = 0.0f; }
    int st = 0, s = s0;
    for (int i = tid; i < 4 * kHRowB / 16; i += 256)           // (the weight staging area is free: every wave passed the barrier behind its reads)
        reinterpret_cast<uint4*>(smem + kRingB)[i] = make_uint4(0u, 0u, 0u, 0u);

    // horizontal pool of (tile T, register pair k / k + 4): conv1_pool32_kernel's statements, in five micro-steps of two instructions --
    // a wave issues in order, so whatever rides in a slot delays that slot's MFMA by its issue time: a whole piece (ten instructions)
    // in one slot cost as much as it saved (first form: K loop 8.2 k -> 9.5 k cycles), two instructions fit beside the 32 cycles
    uint32_t m[4], hx_a = 0u, hx_b = 0u, hx_e = 0u, hx_o = 0u, hx_w = 0u;
    auto hmicro = [&](auto T_, auto k_, auto ph_) {
        constexpr int T = decltype(T_)::value, k = decltype(k_)::value, ph = decltype(ph_)::value;
        if constexpr (ph == 0) { hx_a = Elem<F16>::pack2(acc[T][2 * k], acc[T][2 * k + 1]); hx_b = Elem<F16>::pack2(acc[T][2 * k + 8], acc[T][2 * k + 9]); }
        if constexpr (ph == 1) { hx_a = pk_relu16(hx_a); hx_b = pk_relu16(hx_b); }
        if constexpr (ph == 2) { const auto sw = __builtin_amdgcn_permlane16_swap(hx_a, hx_b, false, false); hx_e = sw[0]; hx_o = sw[1]; }
        if constexpr (ph == 3) { hx_w = T > 0 ? dpp<kDppRor1>(hx_e, oprev[k]) : hx_e; hx_e = pk_max_u16(hx_e, hx_o); }
        if constexpr (ph == 4) { const uint32_t bl = dpp<kDppShr1>(hx_w, hx_o); m[k] = pk_max_u16(hx_e, bl); oprev[k] = hx_o; }
    };
    auto hpiece = [&](auto T_, auto k_) {
        using std::integral_constant;
        hmicro(T_, k_, integral_constant<int, 0>{}); hmicro(T_, k_, integral_constant<int, 1>{}); hmicro(T_, k_, integral_constant<int, 2>{});
        hmicro(T_, k_, integral_constant<int, 3>{}); hmicro(T_, k_, integral_constant<int, 4>{});
    };
    auto hstore = [&](auto T_, int parity) {
        constexpr int T = decltype(T_)::value;
        const uint32_t ad = hw0 + (uint32_t)parity * kHRowB;
        const u32x4 v = {m[0], m[1], m[2], m[3]};
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ad), "v"(v), "n"(T * 2048) : "memory");
    };
    auto vread = [&](auto HALF_, int parity) {                 // pooled rows 2 s', 2 s' + 1 of the step of this parity
        constexpr int HALF = decltype(HALF_)::value;
        const uint32_t ad = vr0 + (uint32_t)parity * kHRowB;
        u32x4 ra, rb;                                          // (asm operands cannot name a capture of a generic lambda)
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                     : "=&v"(ra), "=&v"(rb) : "v"(ad), "n"(HALF * 4096), "n"(HALF * 4096 + 2 * kHRowB) : "memory");
        va = ra; vb = rb;
    };
    u32x4 vo4;
    auto vmicro = [&](auto HALF_, auto ph_, int sp) {          // sp = the step the rows belong to (relative to the band; < 0: the halo step)
        constexpr int HALF = decltype(HALF_)::value, ph = decltype(ph_)::value;
        u32x4& carry = HALF ? carry_hi : carry_lo;
        if constexpr (ph < 4) vo4[ph] = pk_max_u16(pk_max_u16(va[ph], vb[ph]), carry[ph]);
        if constexpr (ph == 4) {
            carry = vb;
            if (sp >= 0) *reinterpret_cast<u32x4*>(vout + (size_t)sp * (66 * 64) + HALF * (32 * 64)) = vo4;
        }
    };
    auto vfinish = [&](auto HALF_, int sp) {
        using std::integral_constant;
        vmicro(HALF_, integral_constant<int, 0>{}, sp); vmicro(HALF_, integral_constant<int, 1>{}, sp); vmicro(HALF_, integral_constant<int, 2>{}, sp);
        vmicro(HALF_, integral_constant<int, 3>{}, sp); vmicro(HALF_, integral_constant<int, 4>{}, sp);
    };
    using std::integral_constant;
    // the K loop's riders, one micro-step per slot: H_B(s - 1) in slots 2..41 (tiles 2, 3), V_lo(s - 1) read at 46 / finished in 58..62,
    // V_hi(s - 2) read at 64 / finished in 76..80, H_A(s) in slots 116..155 (tiles 0, 1: final since slot 112).  No guards for the
    // band's first steps (a compare and a branch per slot would ride too): the riders that run before their step exists pool ZEROS --
    // pair B's accumulators and the four pooled-row buffers start as zeros -- into rows that are never stored (step < 0) and leave
    // the carried row zero, which is what the image's top wants and what the halo step of a lower band overwrites before it is used
    auto ride = [&](auto n_) {
        constexpr int N = decltype(n_)::value;
        if constexpr (N >= 2 && N < 42) {
            constexpr int i = (N - 2) / 5, T = 2 + i / 4, k = i % 4, ph = (N - 2) % 5;
            hmicro(integral_constant<int, T>{}, integral_constant<int, k>{}, integral_constant<int, ph>{});
            if constexpr (k == 3 && ph == 4) hstore(integral_constant<int, T>{}, (s - 1) & 1);
        }
        if constexpr (N == 46) vread(integral_constant<int, 0>{}, (s - 1) & 1);
        if constexpr (N >= 58 && N < 63) vmicro(integral_constant<int, 0>{}, integral_constant<int, N - 58>{}, s - 1);
        if constexpr (N == 64) vread(integral_constant<int, 1>{}, (s - 2) & 1);
        if constexpr (N >= 76 && N < 81) vmicro(integral_constant<int, 1>{}, integral_constant<int, N - 76>{}, s - 2);
        if constexpr (N >= 116 && N < 156) {
            constexpr int i = (N - 116) / 5, T = i / 4, k = i % 4, ph = (N - 116) % 5;
            hmicro(integral_constant<int, T>{}, integral_constant<int, k>{}, integral_constant<int, ph>{});
            if constexpr (k == 3 && ph == 4) hstore(integral_constant<int, T>{}, s & 1);
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SSTAMPK(1);

    for (; st < nsteps; ++st, ++s) {
        SSTAMP(2);
        const int n0 = 4 * st;
        if (st + 1 < nsteps) issue_row(n0 + 9 + wave);
        uint32_t xa[7];
#pragma unroll
        for (int kr = 0; kr < 7; ++kr) xa[kr] = smem_base + (uint32_t)(((n0 + 2 * r + kr) % kRing) * kPitch) + lane_off;
        bf16x8 xf[8];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // slots 0..6 = (g 0: tiles 0, 1), (g 1: 0, 1), (g 2: 0, 1), (g 3: 0)
        xf[0] = lds_read_2xb64<0>(xa[0]); xf[1] = lds_read_2xb64<2304>(xa[0]); xf[2] = lds_read_2xb64<32>(xa[0]); xf[3] = lds_read_2xb64<32 + 2304>(xa[0]);
        xf[4] = lds_read_2xb64<64>(xa[0]); xf[5] = lds_read_2xb64<64 + 2304>(xa[0]); xf[6] = lds_read_2xb64<96>(xa[0]);
        __builtin_amdgcn_sched_barrier(0);
        stem32r_slots<F16>(acc, xf, wreg, xa, bias_v, ride, std::make_integer_sequence<int, kSlots32>{});
        SSTAMP(3);
        SSTAMP(4);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        SSTAMP(5);
        __builtin_amdgcn_s_barrier();       // the step's ONE barrier: ring rows of the next step landed and published, this step's ring rows and H_A / H_B buffers handed over
        SSTAMP(6);
        SSTAMP(7);
        SSTAMP(8);
    }
    // ---- what has not ridden yet: H_B of the last step, then V_hi(last - 1), V_lo(last), V_hi(last)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[2]), "+v"(acc[3]));
    {
        using I0 = integral_constant<int, 0>; using I1 = integral_constant<int, 1>; using I2 = integral_constant<int, 2>; using I3 = integral_constant<int, 3>;
        const int sl = s - 1;                                  // the last step
        hpiece(I2{}, I0{}); hpiece(I2{}, I1{}); hpiece(I2{}, I2{}); hpiece(I2{}, I3{}); hstore(I2{}, sl & 1);
        hpiece(I3{}, I0{}); hpiece(I3{}, I1{}); hpiece(I3{}, I2{}); hpiece(I3{}, I3{}); hstore(I3{}, sl & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        vread(I1{}, (sl - 1) & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va), "+v"(vb)::"memory");
        vfinish(I1{}, sl - 1);
        vread(I0{}, sl & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va), "+v"(vb)::"memory");
        vfinish(I0{}, sl);
        vread(I1{}, sl & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va), "+v"(vb)::"memory");
        vfinish(I1{}, sl);
    }
    SSTAMPK(9);
}

}  // namespace

namespace spsp {

// conv1 + BN + ReLU + max-pool for B crops: in = native [B][262][264][18], out = [B][66][66][64] (interior).
int launch_conv_stem(const uint16_t* in, const uint16_t* w, const float* bias, uint16_t* out, int B, int f16, int num_cus,
                     unsigned int* zero_words, int nzero, hipStream_t st, int rows_per_band, int mfma32, float* zero_wide, int nzero_wide16) {
    SPSP_REQUIRE(in && w && bias && out && B > 0, "conv_stem: bad arguments");
    SPSP_REQUIRE(rows_per_band == 0 || (rows_per_band >= 4 && rows_per_band <= 128 && (rows_per_band & (rows_per_band - 1)) == 0),
                 "conv_stem: bands of %d conv rows (a power of two, 4 .. 128)", rows_per_band);
    StemArgs a;
    a.in = in; a.w = w; a.bias = bias; a.out = out;
    a.zero = zero_words; a.nzero = zero_words ? nzero : 0;
    a.zero2 = reinterpret_cast<uint4*>(zero_wide); a.nzero2 = zero_wide ? nzero_wide16 : 0;
    // bands of 16 conv rows (8 per image) fill 256 CUs at 32 crops with one halo step in nine; smaller batches take
    // finer bands so that the grid still covers the chip
    // (cost of a band height = rounds of workgroups x a workgroup's life, prologue 13 k + 12 k cycles per step of two conv rows --
    // the stamps below: 22 crops are 176 workgroups of 16 rows in ONE round, not 352 of 8 rows in two)
    {
        long best = 0;
        for (int rb = 16; rb >= 4; rb /= 2) {
            const long wgs = (long)B * (128 / rb), rounds = (wgs + num_cus - 1) / num_cus;
            const long cost = rounds * (13 + 12 * (rb / 2 + 1));
            if (best == 0 || cost < best) { best = cost; a.rows_per_band = rb; }
        }
    }
    // (plan "throughput": coarser bands = fewer, longer-lived workgroups -- the 114 KB of weights and the halo step are paid per band)
    // -- as long as a quarter of the chip still has a workgroup each
    while (a.rows_per_band < rows_per_band && B * (128 / (2 * a.rows_per_band)) * 4 >= num_cus) a.rows_per_band *= 2;
    a.bands = 128 / a.rows_per_band;
    SPSP_TRY(ensure_dynamic_lds<&conv1_pool_kernel<false>>(kLds));
    SPSP_TRY(ensure_dynamic_lds<&conv1_pool_kernel<true>>(kLds));
    SPSP_TRY(ensure_dynamic_lds<&conv1_pool32_kernel<false>>(kLds));
    SPSP_TRY(ensure_dynamic_lds<&conv1_pool32_kernel<true>>(kLds));
    SPSP_TRY(ensure_dynamic_lds<&conv1_pool32r_kernel<false>>(kLds32r));
    SPSP_TRY(ensure_dynamic_lds<&conv1_pool32r_kernel<true>>(kLds32r));
    a.dbg = nullptr;
#ifdef SPSP_CONV_STAMPS
    if (getenv("SPSP_CONV_DEBUG")) {
        (void)hipMalloc(&a.dbg, 16 * sizeof(unsigned long long));
        const int diag = getenv("SPSP_STEM_DIAG") ? atoi(getenv("SPSP_STEM_DIAG")) : 0;
        (void)ensure_dynamic_lds<&conv1_pool_kernel<true, 1>>(kLds);
        (void)ensure_dynamic_lds<&conv1_pool_kernel<true, 2>>(kLds);
        if (mfma32 == 1 && diag == 0) { if (f16) conv1_pool32r_kernel<true><<<dim3(B * a.bands), 256, kLds32r, st>>>(a); else conv1_pool32r_kernel<false><<<dim3(B * a.bands), 256, kLds32r, st>>>(a); }
        else if (mfma32 && diag == 0) { if (f16) conv1_pool32_kernel<true><<<dim3(B * a.bands), 256, kLds, st>>>(a); else conv1_pool32_kernel<false><<<dim3(B * a.bands), 256, kLds, st>>>(a); }
        else if (diag == 1) conv1_pool_kernel<true, 1><<<dim3(B * a.bands), 256, kLds, st>>>(a);
        else if (diag == 2) conv1_pool_kernel<true, 2><<<dim3(B * a.bands), 256, kLds, st>>>(a);
        else if (f16) conv1_pool_kernel<true><<<dim3(B * a.bands), 256, kLds, st>>>(a);
        else conv1_pool_kernel<false><<<dim3(B * a.bands), 256, kLds, st>>>(a);
        (void)hipDeviceSynchronize();
        unsigned long long hh[16];
        (void)hipMemcpy(hh, a.dbg, sizeof hh, hipMemcpyDeviceToHost);
        fprintf(stderr, "stem[%s diag %d] wg9 (band 1, 9 steps): prologue %llu | step3: K loop %llu | h-pool + LDS writes %llu | vmcnt/lgkm wait %llu | barrier %llu | vertical + stores %llu | barrier %llu | whole kernel %llu cycles\n",
                mfma32 == 1 ? "32x32x16, pooling in the K loop" : mfma32 ? "32x32x16" : "16x16x32", diag, hh[1] - hh[0], hh[3] - hh[2], hh[4] - hh[3], hh[5] - hh[4], hh[6] - hh[5], hh[7] - hh[6], hh[8] - hh[7], hh[9] - hh[0]);
        (void)hipFree(a.dbg);
        return SPSP_OK;
    }
#endif
    if (mfma32 == 1) {
        if (f16) launch_kernel(&conv1_pool32r_kernel<true>, dim3(B * a.bands), dim3(256), kLds32r, st, a);
        else launch_kernel(&conv1_pool32r_kernel<false>, dim3(B * a.bands), dim3(256), kLds32r, st, a);
    } else if (mfma32) {
        if (f16) launch_kernel(&conv1_pool32_kernel<true>, dim3(B * a.bands), dim3(256), kLds, st, a);
        else launch_kernel(&conv1_pool32_kernel<false>, dim3(B * a.bands), dim3(256), kLds, st, a);
    } else if (f16) launch_kernel(&conv1_pool_kernel<true, 0>, dim3(B * a.bands), dim3(256), kLds, st, a);
    else launch_kernel(&conv1_pool_kernel<false, 0>, dim3(B * a.bands), dim3(256), kLds, st, a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp
