// 3x3 / stride-1 convolution, 64 -> 64 channels (ResNet-18 layer1: 4 convs, 12 % of the FLOPs but the largest share of
// GPU time of the 3x3 kernels) with the WHOLE WEIGHT TENSOR RESIDENT IN LDS.
//
// conv_patch.hip streams one [64 x 64] weight slab per filter tap through an LDS-DMA ring and therefore pays a
// counted-vmcnt wait + workgroup barrier per tap.  With Cin = Cout = 64 all nine slabs are 73.7 KB: this kernel loads
// them ONCE per workgroup, makes the workgroups persistent over the 16 x 16-pixel tiles, and runs the 9 taps x 32 MFMAs
// of a tile back to back with no barrier at all; fragment reads are software-pipelined one tap ahead (two register sets).
//
// Round 2: the phases of consecutive tiles overlap.  At 64 channels a tile moves 32 KB in, 32 KB out and 32 KB of
// residual for 4.6k cycles of MFMA, and the first version ran load -> K loop -> epilogue one after the other (stamps:
// K loop 5.4k of ~13k cycles per tile).  Now
//   * TWO patch buffers: the next tile's input patch is fetched by LDS-DMA under the current tile's K loop;
//   * the epilogue needs no LDS and no barrier: the weight rows are permuted at load time so that, after the MFMAs, a lane
//     holds 2 x 8 CONSECUTIVE output channels of each of its pixels (accumulator tile i, register e of lane group fq is
//     channel 32 (i >> 1) + 8 fq + 4 (i & 1) + e) -- residual loads and output stores are plain 16-byte accesses, four
//     lanes covering an aligned 64-byte half line;
//   * the previous tile's accumulators stay in registers (two accumulator sets, the tile loop is unrolled by two) and its
//     bias + residual + ReLU + pack + store is issued in eight pieces between the MFMAs of the current tile's taps; the
//     current tile's residual is loaded into the registers those pieces free.
// One s_waitcnt vmcnt(0) + one workgroup barrier per tile remain (patch landed / every wave has left the other buffer).
//
// Workgroup = 4 waves (one per SIMD), wave tile 64 px x 64 ch on v_mfma_f32_16x16x32 (weights = A operand, pixels = B
// operand).  LDS: weights 73.7 KB + 2 x patch 41 KB = 155.7 KB.
#include <cstdlib>
#include <vector>
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
constexpr int kBiasOff = kWAll + 2 * kPatch;     // 64 fp32 biases in accumulator order (row R of the permuted weights)
constexpr int kLds = kBiasOff + 256 + 16;        // 157 968 B (last 16 B: tile-id slots)

#ifdef SPSP_CONV_STAMPS
#define WSTAMP(i) if (a.dbg && blockIdx.x == 8 && tid == 0) a.dbg[i] = __builtin_amdgcn_s_memtime()
#define WSTAMP_WG(i) if (a.dbg && tid == 0) a.dbg[16 + 4 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime()
#else
#define WSTAMP(i)
#define WSTAMP_WG(i)
#endif

// output channel held by accumulator tile i, MFMA row r (= 4 fq + e) -- see the header
__host__ __device__ constexpr int wres_channel(int i, int r) { return 32 * (i >> 1) + 8 * (r >> 2) + 4 * (i & 1) + (r & 3); }

template <bool F16, bool RES, bool RELU>
__global__ __launch_bounds__(256) void conv3x3_wres_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    unsigned char* const wlds = smem;
    unsigned char* const patch = smem + kWAll;               // two buffers of kPatch
    volatile int* const s_tile = reinterpret_cast<volatile int*>(smem + kBiasOff + 256);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.Wo, tpr = W / 16, tpi = tpr * tpr;
    const int ntiles = a.M / 256;
    const uint16_t* const in_ptr = a.in;

    // Dynamic tile scheduler: workgroups take tiles from a device counter (zero before the launch) instead of the static
    // b, b + grid, ...: in the multi-stream pipeline some CUs are held by other kernels' workgroups, the workgroups that
    // ARE resident take the work and late starters find none.  Ids are fetched two tiles ahead (the next tile's patch is
    // requested at the top of the current one).  The first two BEFORE any DMA is in flight: the atomics' returns are then
    // the only thing their s_waitcnt waits for.
    WSTAMP_WG(0);
    // Workgroups b, b + 8, ... (one XCD under round-robin dispatch -- a speed assumption only) share ONE of eight counters,
    // each on its own 128-byte line, and own the tiles = class (mod 8): with a single counter the 256 workgroups' first
    // fetches serialised at one L2 line (stamps: ~12k cycles mean between kernel entry and the first DMA).
    const int nclass = gridDim.x < 8 ? (int)gridDim.x : 8, cls = (int)blockIdx.x % nclass;
    unsigned int* const ctr = a.counter + 32 * cls;
    if (tid == 0) {
        const int r0 = (int)atomicAdd(ctr, 2u);             // the first two ids in one fetch
        s_tile[2] = r0 * nclass + cls;
        s_tile[0] = (r0 + 1) * nclass + cls;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int t = __builtin_amdgcn_readfirstlane(s_tile[2]);
    if ((unsigned)t >= (unsigned)ntiles) return;
    if (tid < 64) reinterpret_cast<float*>(smem + kBiasOff)[tid] = a.bias[wres_channel(tid >> 4, tid & 15)];                                // a late starter: the resident workgroups took every tile

    // ---- all nine weight slabs, once: piece p = 8 LDS rows of one tap; lane l -> LDS row R = 8 (p & 7) + l/8 (which
    //      holds output channel wres_channel(R / 16, R % 16)), physical chunk l%8 ----
    const int drow = lane >> 3, dchunk = lane & 7;
    {
        const int Ktot = 9 * 64;
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int p = wave + 4 * i, tap = p >> 3, row = (p & 7) * 8 + drow;
            const int ch = wres_channel(row >> 4, row & 15);
            dma16(a.w + (size_t)ch * Ktot + tap * 64 + ((dchunk ^ (row & 7)) * 8), wlds + p * 1024);
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
    auto tile_origin = [&](int tt, int& img, int& y0, int& x0) {
        img = tt / tpi;
        const int r = tt - img * tpi;
        y0 = (r / tpr) * 16; x0 = (r % tpr) * 16;
    };
    auto issue_patch = [&](int tt, unsigned char* buf) {
        int img, y0, x0;
        tile_origin(tt, img, y0, x0);
        const uint16_t* base = in_ptr + (size_t)img * a.in_img + (size_t)y0 * a.in_row + (size_t)x0 * a.in_pix;
#pragma unroll
        for (int i = 0; i < 11; ++i) {
            const int p = wave + 4 * i;
            if (p < kNDma) dma16(base + p_rel[i], buf + p * 1024);
        }
    };
    auto out_origin = [&](int tt) -> size_t {
        int img, y0, x0;
        tile_origin(tt, img, y0, x0);
        return 2 * ((size_t)a.out_off + (size_t)img * a.out_img + (size_t)y0 * a.out_row + (size_t)x0 * a.out_pix);   // bytes
    };

    const int frow = lane & 15, fq = lane >> 4;
    const uint32_t smem_base = (uint32_t)(uintptr_t)smem;
    const uint32_t w_lane = smem_base + frow * 128 + ((fq ^ (frow & 7)) * 16);       // ks = 0; ks = 1: ^ 64 on the low part
    const uint32_t w_lane1 = smem_base + frow * 128 + (((4 + fq) ^ (frow & 7)) * 16);
    const int qb = (wave * 4) * 18 + frow;
    uint32_t xo[6][3];                      // patch pixel (4 wave + r, frow + dx), r = j + ky: 18 addresses serve the 9 x 4 (tap, j) pairs
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int q = qb + r * 18 + dx;
            xo[r][dx] = smem_base + kWAll + q * 128 + ((fq ^ (q & 7)) * 16);
        }
    // this lane's pixel (row 4 wave + j, column frow of the tile) and channel group: byte offset from the tile origin
    const uint32_t lane_boff = (uint32_t)(((wave * 4) * a.out_row + frow * a.out_pix + fq * 8) * 2);
    const uint32_t bias_lane = smem_base + kBiasOff + fq * 16;
    constexpr bool has_res = RES, relu = RELU;
    unsigned char* const out_ptr = reinterpret_cast<unsigned char*>(a.out);
    const unsigned char* const res_ptr = reinterpret_cast<const unsigned char*>(a.res);
    const int out_row_b = a.out_row * 2;

    uint4 resr[8];                          // residual of piece c = 2 j + h: channels 32 h + 8 fq .. + 7 of pixel row j
#pragma unroll
    for (int c = 0; c < 8; ++c) resr[c] = make_uint4(0, 0, 0, 0);
    // piece c of a finished tile (the bias went in as the accumulators' initial value): residual + ReLU, pack, one 16-byte
    // store.  org = byte offset of the tile origin, wave-uniform: the address is scalar base + 32-bit lane offset.
    auto epi_piece = [&](const f32x4 (&ap)[4][4], size_t org, int c) {
        const int j = c >> 1, h = c & 1;
        const f32x4 lo4 = ap[2 * h][j], hi4 = ap[2 * h + 1][j];
        float v0 = lo4[0], v1 = lo4[1], v2 = lo4[2], v3 = lo4[3], v4 = hi4[0], v5 = hi4[1], v6 = hi4[2], v7 = hi4[3];
        if constexpr (has_res) {
            const uint4 r = resr[c];
            v0 += Elem<F16>::lo(r.x); v1 += Elem<F16>::hi(r.x); v2 += Elem<F16>::lo(r.y); v3 += Elem<F16>::hi(r.y);
            v4 += Elem<F16>::lo(r.z); v5 += Elem<F16>::hi(r.z); v6 += Elem<F16>::lo(r.w); v7 += Elem<F16>::hi(r.w);
        }
        if constexpr (relu) {
            v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
            v4 = fmaxf(v4, 0.f); v5 = fmaxf(v5, 0.f); v6 = fmaxf(v6, 0.f); v7 = fmaxf(v7, 0.f);
        }
        unsigned char* const base = out_ptr + org + (size_t)(j * out_row_b + h * 64);
        *reinterpret_cast<uint4*>(base + lane_boff) =
            make_uint4(Elem<F16>::pack2(v0, v1), Elem<F16>::pack2(v2, v3), Elem<F16>::pack2(v4, v5), Elem<F16>::pack2(v6, v7));
    };
    auto load_res = [&](size_t org, int c) {
        const int j = c >> 1, h = c & 1;
        const unsigned char* const base = res_ptr + org + (size_t)(j * out_row_b + h * 64);
        resr[c] = *reinterpret_cast<const uint4*>(base + lane_boff);
    };

    WSTAMP(0);
    issue_patch(t, patch);
    size_t prev_org = 0;
    unsigned next_raw = 0;
    int ntile_done = 0;

    // One tile on patch buffer BUF: acc receives its products, ap holds the previous tile's (stored piece by piece here).
#define WRES_SB __builtin_amdgcn_sched_barrier(0);
#define WRES_RW(SET, KK, I, TAP) wfr[SET][KK][I] = lds_read_b128<(I) * 2048>(((KK) ? w_lane1 : w_lane) + (TAP) * kWSlab);
#define WRES_RX(SET, KK, J, TAP) xfr[SET][KK][J] = lds_read_b128<POFF>(xo[(J) + (TAP) / 3][(TAP) % 3] ^ ((KK) * 64));
#define WRES_M2(S1, KK, I, J)                                                                  \
        WRES_SB                                                                                \
        acc[I][J] = Elem<F16>::mfma(wfr[S1][KK][I], xfr[S1][KK][J], acc[I][J]);                \
        acc[I][(J) + 1] = Elem<F16>::mfma(wfr[S1][KK][I], xfr[S1][KK][(J) + 1], acc[I][(J) + 1]); \
        WRES_SB
    // One tap: the 16 fragment reads of tap TAP (set SET) go out between the first 16 MFMAs of the previous tap (set
    // SET^1), one read pair per MFMA pair, pinned by sched_barriers (hipcc otherwise gathers the reads in front and sinks
    // the register-only MFMAs BELOW the wait).  The last 16 MFMAs have no reads beside them: the previous tile's epilogue
    // piece PIECE and the residual load that refills its registers are left to the scheduler inside that region.
#define WRES_STEP(TAP, SET, PIECE)                                                             \
    {                                                                                          \
        WRES_RW(SET, 0, 0, TAP) WRES_RW(SET, 0, 1, TAP) WRES_M2((SET) ^ 1, 0, 0, 0)             \
        WRES_RW(SET, 0, 2, TAP) WRES_RW(SET, 0, 3, TAP) WRES_M2((SET) ^ 1, 0, 0, 2)             \
        WRES_RX(SET, 0, 0, TAP) WRES_RX(SET, 0, 1, TAP) WRES_M2((SET) ^ 1, 0, 1, 0)             \
        WRES_RX(SET, 0, 2, TAP) WRES_RX(SET, 0, 3, TAP) WRES_M2((SET) ^ 1, 0, 1, 2)             \
        WRES_RW(SET, 1, 0, TAP) WRES_RW(SET, 1, 1, TAP) WRES_M2((SET) ^ 1, 0, 2, 0)             \
        WRES_RW(SET, 1, 2, TAP) WRES_RW(SET, 1, 3, TAP) WRES_M2((SET) ^ 1, 0, 2, 2)             \
        WRES_RX(SET, 1, 0, TAP) WRES_RX(SET, 1, 1, TAP) WRES_M2((SET) ^ 1, 0, 3, 0)             \
        WRES_RX(SET, 1, 2, TAP) WRES_RX(SET, 1, 3, TAP) WRES_M2((SET) ^ 1, 0, 3, 2)             \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                       \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                   \
                acc[i_][j_] = Elem<F16>::mfma(wfr[(SET) ^ 1][1][i_], xfr[(SET) ^ 1][1][j_], acc[i_][j_]); \
        if constexpr (PREV) epi_piece(ap, prev_org, PIECE);                                    \
        if constexpr (has_res) load_res(org, PIECE);                                           \
        _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* one MFMA ... */          \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      /* ... three VALU under it */ \
        }                                                                                      \
        WRES_SB                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                     \
        WRES_SB                                                                                \
    }
    auto tile = [&](auto buf_c, auto prev_c, f32x4 (&acc)[4][4], const f32x4 (&ap)[4][4]) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr bool PREV = decltype(prev_c)::value;
        constexpr int POFF = BUF * kPatch;
        // this tile's patch (requested a tile ago), its residual registers and the id of the tile after next have landed;
        // the previous tile's stores have left
        // (the id of the tile after next was requested a tile ago into an ACCUMULATION register, see below)
        if constexpr (PREV) {
            unsigned got;
            asm volatile("s_waitcnt vmcnt(0)\n\tv_accvgpr_read_b32 %0, %1" : "=v"(got) : "a"(next_raw) : "memory");
            if (tid == 0) s_tile[BUF] = (int)got * nclass + cls;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // ... for every wave, and every wave has left the other buffer
        if (ntile_done == 0) { WSTAMP(1); } else if (ntile_done == 1) { WSTAMP(4); }
        const int tn = __builtin_amdgcn_readfirstlane(s_tile[BUF]);
        // (unsigned compares: a negative or garbage id -- a stale register, see below -- must never become an address)
        const bool tn_ok = (unsigned)tn < (unsigned)ntiles;
        if (tn_ok) issue_patch(tn, patch + (BUF ^ 1) * kPatch);
        {
            // Next id: a returning atomic whose result is not needed for a whole tile.  Inline asm, because hipcc's own
            // atomic waits for the return at once (wave-aggregated lowering, or a copy of the result), which would park the
            // wave on the patch DMA just issued.  hipcc treats an asm output as written at ASMEND and may move it before
            // the data has arrived (it did: a stale tile id, found as a memory fault at 512 tiles) -- so the result goes
            // to an AGPR, which the register allocator has no reason to touch before the asm above reads it, and the
            // statement has no branch around it (lane 0 of wave 0 is selected through EXEC; EXEC = 0 issues nothing).
            // "No reason to" is not a guarantee: tests/test_abi_and_host.py disassembles this kernel and fails if any
            // instruction between the atomic and the `s_waitcnt vmcnt(0); v_accvgpr_read` names that AGPR, and every use of
            // an id is range-checked unsigned, so that a stale id could cost a wrong tile, never a wild address.
            const int only = __builtin_amdgcn_readfirstlane((wave == 0 && tn_ok) ? 1 : 0);
            const unsigned one = 1u;
            unsigned long long saved;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, 0\n\t"
                         "global_atomic_add %0, %3, %4, off sc0\n\ts_mov_b64 exec, %1"
                         : "=a"(next_raw), "=&s"(saved) : "s"(only), "v"(ctr), "a"(one) : "memory");
        }
        const size_t org = out_origin(t);
        // accumulators start at the bias (tile i, register e of this lane = channel row 16 i + 4 fq + e of the bias block)
        {
            f32x4 b0, b1, b2, b3;       // loads and their wait in ONE statement: hipcc treats asm outputs as written at ASMEND
            asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
                         "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(bias_lane) : "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[0][j] = b0; acc[1][j] = b1; acc[2][j] = b2; acc[3][j] = b3; }
        }
        bf16x8 wfr[2][2][4], xfr[2][2][4];
        // tap 0 fragments
        WRES_RW(0, 0, 0, 0) WRES_RW(0, 0, 1, 0) WRES_RW(0, 0, 2, 0) WRES_RW(0, 0, 3, 0)
        WRES_RX(0, 0, 0, 0) WRES_RX(0, 0, 1, 0) WRES_RX(0, 0, 2, 0) WRES_RX(0, 0, 3, 0)
        WRES_RW(0, 1, 0, 0) WRES_RW(0, 1, 1, 0) WRES_RW(0, 1, 2, 0) WRES_RW(0, 1, 3, 0)
        WRES_RX(0, 1, 0, 0) WRES_RX(0, 1, 1, 0) WRES_RX(0, 1, 2, 0) WRES_RX(0, 1, 3, 0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        WRES_STEP(1, 1, 0) WRES_STEP(2, 0, 1) WRES_STEP(3, 1, 2) WRES_STEP(4, 0, 3)
        WRES_STEP(5, 1, 4) WRES_STEP(6, 0, 5) WRES_STEP(7, 1, 6) WRES_STEP(8, 0, 7)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = Elem<F16>::mfma(wfr[0][kk][i], xfr[0][kk][j], acc[i][j]);
        if (ntile_done == 0) { WSTAMP(2); } else if (ntile_done == 1) { WSTAMP(5); }
        ++ntile_done;
        prev_org = org;
        t = tn;
    };
    f32x4 accA[4][4], accB[4][4];
    // the last tile's epilogue has no K loop to hide under
    auto flush = [&](const f32x4 (&ap)[4][4]) {
        if constexpr (has_res) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int c = 0; c < 8; ++c) epi_piece(ap, prev_org, c);
        WSTAMP(7);
        WSTAMP_WG(1);
#ifdef SPSP_CONV_STAMPS
        if (a.dbg && tid == 0) a.dbg[16 + 4 * blockIdx.x + 2] = (unsigned long long)ntile_done;
#endif
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    tile(B0{}, std::false_type{}, accA, accB);              // the first tile has no predecessor to store
    if ((unsigned)t >= (unsigned)ntiles) flush(accA);
    else
        for (;;) {
            tile(B1{}, std::true_type{}, accB, accA);
            if ((unsigned)t >= (unsigned)ntiles) { flush(accB); break; }
            tile(B0{}, std::true_type{}, accA, accB);
            if ((unsigned)t >= (unsigned)ntiles) { flush(accA); break; }
        }
#undef WRES_STEP
#undef WRES_M2
#undef WRES_RX
#undef WRES_RW
#undef WRES_SB
}

}  // namespace

namespace spsp {

bool conv_wres_supported(const ConvArgs& a) {
    return a.stride == 1 && a.kw_steps == 3 && a.chunks == 1 && a.nsteps == 9 && a.N == 64 && a.in_pix == 64 && a.in_off == 0 &&
           a.M % 256 == 0 && a.Ho == a.Wo && a.Wo % 16 == 0 && a.splitk <= 1 && !a.pool_out && a.out && a.counter;
}

int launch_conv_wres(const ConvArgs& a, int num_cus, hipStream_t st) {
    SPSP_REQUIRE(conv_wres_supported(a), "conv_wres: needs a 3x3 stride-1 64->64 convolution on a map that is a multiple of 16 wide");
    const int ntiles = a.M / 256;
    const int grid = ntiles < num_cus ? ntiles : num_cus;       // persistent: one workgroup per CU (154 KB of LDS each)
#ifdef SPSP_CONV_STAMPS
    if (getenv("SPSP_CONV_DEBUG")) {
        ConvArgs b = a;
        const size_t nd = 16 + 4 * (size_t)grid;
        (void)hipMalloc(&b.dbg, nd * sizeof(unsigned long long));
        (void)hipMemset(b.dbg, 0, nd * sizeof(unsigned long long));
        SPSP_TRY((ensure_dynamic_lds<&conv3x3_wres_kernel<true, true, true>>(kLds)));
        SPSP_TRY((ensure_dynamic_lds<&conv3x3_wres_kernel<true, false, true>>(kLds)));
        if (b.res) conv3x3_wres_kernel<true, true, true><<<dim3(grid), 256, kLds, st>>>(b);
        else conv3x3_wres_kernel<true, false, true><<<dim3(grid), 256, kLds, st>>>(b);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> hv(nd);
        (void)hipMemcpy(hv.data(), b.dbg, nd * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const unsigned long long* h = hv.data();
        fprintf(stderr, "wres grid=%d tiles=%d: weights+patch landed %llu | K loop %llu | tile2 top %llu | K loop (+ epilogue 1) %llu | last epilogue %llu | total %llu\n",
                grid, ntiles, h[1] - h[0], h[2] - h[1], h[4] - h[2], h[5] - h[4], h[7] - h[5], h[7] - h[0]);
        {   // every workgroup: start / end against the earliest start, tiles done
            unsigned long long t0 = ~0ull, t1 = 0, smax = 0;
            int hist[8] = {0};
            for (int g = 0; g < grid; ++g) { const unsigned long long s0 = h[16 + 4 * g]; if (s0 && s0 < t0) t0 = s0; }
            double dur = 0;
            for (int g = 0; g < grid; ++g) {
                const unsigned long long s0 = h[16 + 4 * g], e0 = h[16 + 4 * g + 1];
                if (e0 > t1) t1 = e0;
                if (s0 > smax) smax = s0;
                dur += (double)(e0 - s0);
                const int nt = (int)h[16 + 4 * g + 2];
                hist[nt < 7 ? nt : 7]++;
            }
            fprintf(stderr, "  all workgroups: first start -> last start %llu | first start -> last end %llu | mean residency %.0f | tiles/wg histogram 0:%d 1:%d 2:%d 3:%d 4+:%d\n",
                    smax - t0, t1 - t0, dur / grid, hist[0], hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
        }
        (void)hipFree(b.dbg);
        return SPSP_OK;
    }
#endif
    auto go = [&](auto f16_c, auto res_c, auto relu_c) -> int {
        constexpr auto kern = &conv3x3_wres_kernel<decltype(f16_c)::value, decltype(res_c)::value, decltype(relu_c)::value>;
        SPSP_TRY(ensure_dynamic_lds<kern>(kLds));
        launch_kernel(kern, dim3(grid), dim3(256), kLds, st, a);
        return SPSP_OK;
    };
    auto pick_relu = [&](auto f16_c, auto res_c) -> int {
        return a.relu ? go(f16_c, res_c, std::true_type{}) : go(f16_c, res_c, std::false_type{});
    };
    auto pick_res = [&](auto f16_c) -> int {
        return a.res ? pick_relu(f16_c, std::true_type{}) : pick_relu(f16_c, std::false_type{});
    };
    SPSP_TRY(a.f16 ? pick_res(std::true_type{}) : pick_res(std::false_type{}));
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp
