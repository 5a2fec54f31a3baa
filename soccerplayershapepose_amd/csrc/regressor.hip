// SingleInputRegressor(18, 18, 3) for gfx950: ResNet-18 encoder (PR/models/resnet.py:124-216) as a plan of
// MFMA convolutions (conv_stem / conv_wres / conv_patch / conv_ring .hip) over zero-bordered NHWC 16-bit feature maps, + the IEF MLP
// (PR/models/ief_module.py:48-64) in fp32.  Eval-mode BatchNorm is folded into weights/bias at load.
#include <cmath>
#include <cstdlib>
#include <string>

#include "conv.hpp"
#include "conv_elem.hpp"

namespace {
using namespace spsp;

constexpr int kWH = SPSP_PROXY_WH, kCin = SPSP_PROXY_C, kPad = SPSP_PROXY_PAD;
constexpr int kWHp = SPSP_PROXY_NATIVE_H, kWWp = SPSP_PROXY_NATIVE_W;   // 262 rows x 264 columns
constexpr size_t kNativeElems = (size_t)kWHp * kWWp * kCin;
constexpr int kFeat = 512, kNPar = 157, kHid = 512;
// per-forward counter block: 128-byte lines 0..31 = tile schedulers of the four conv_wres launches (8 each), 32..47 = IEF barriers
// (one per block of 16 players: 256 / 16)
constexpr int kIefCounterLines = 16;
constexpr int kCounterWords = (32 + kIefCounterLines) * 32;

struct ConvDesc { const char* name; const char* bn; int cout, cin, k, stride; };
const ConvDesc kConvs[SPSP_NUM_CONVS] = {
    {"image_encoder.conv1", "image_encoder.bn1", 64, 18, 7, 2},
    {"image_encoder.layer1.0.conv1", "image_encoder.layer1.0.bn1", 64, 64, 3, 1},
    {"image_encoder.layer1.0.conv2", "image_encoder.layer1.0.bn2", 64, 64, 3, 1},
    {"image_encoder.layer1.1.conv1", "image_encoder.layer1.1.bn1", 64, 64, 3, 1},
    {"image_encoder.layer1.1.conv2", "image_encoder.layer1.1.bn2", 64, 64, 3, 1},
    {"image_encoder.layer2.0.conv1", "image_encoder.layer2.0.bn1", 128, 64, 3, 2},
    {"image_encoder.layer2.0.conv2", "image_encoder.layer2.0.bn2", 128, 128, 3, 1},
    {"image_encoder.layer2.0.downsample.0", "image_encoder.layer2.0.downsample.1", 128, 64, 1, 2},
    {"image_encoder.layer2.1.conv1", "image_encoder.layer2.1.bn1", 128, 128, 3, 1},
    {"image_encoder.layer2.1.conv2", "image_encoder.layer2.1.bn2", 128, 128, 3, 1},
    {"image_encoder.layer3.0.conv1", "image_encoder.layer3.0.bn1", 256, 128, 3, 2},
    {"image_encoder.layer3.0.conv2", "image_encoder.layer3.0.bn2", 256, 256, 3, 1},
    {"image_encoder.layer3.0.downsample.0", "image_encoder.layer3.0.downsample.1", 256, 128, 1, 2},
    {"image_encoder.layer3.1.conv1", "image_encoder.layer3.1.bn1", 256, 256, 3, 1},
    {"image_encoder.layer3.1.conv2", "image_encoder.layer3.1.bn2", 256, 256, 3, 1},
    {"image_encoder.layer4.0.conv1", "image_encoder.layer4.0.bn1", 512, 256, 3, 2},
    {"image_encoder.layer4.0.conv2", "image_encoder.layer4.0.bn2", 512, 512, 3, 1},
    {"image_encoder.layer4.0.downsample.0", "image_encoder.layer4.0.downsample.1", 512, 256, 1, 2},
    {"image_encoder.layer4.1.conv1", "image_encoder.layer4.1.bn1", 512, 512, 3, 1},
    {"image_encoder.layer4.1.conv2", "image_encoder.layer4.1.bn2", 512, 512, 3, 1},
};

// ---- layout kernels -------------------------------------------------------------------------------------
// NCHW fp32 [B,18,256,256] -> native [B,262,264,18] bf16 / fp16 (interior + zero border).
template <bool F16>
__global__ void proxy_pack_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst) {
    const int xp = blockIdx.x * blockDim.x + threadIdx.x, yp = blockIdx.y, b = blockIdx.z;
    if (xp >= kWWp) return;
    const int x = xp - kPad, y = yp - kPad;
    const bool inside = x >= 0 && x < kWH && y >= 0 && y < kWH;
    uint32_t* d = reinterpret_cast<uint32_t*>(dst + ((size_t)b * kWHp * kWWp + (size_t)yp * kWWp + xp) * kCin);
    const float* s = src + (size_t)b * kCin * kWH * kWH + (size_t)y * kWH + x;
#pragma unroll
    for (int c = 0; c < kCin; c += 2) {
        float v0 = 0.f, v1 = 0.f;
        if (inside) { v0 = s[(size_t)c * kWH * kWH]; v1 = s[(size_t)(c + 1) * kWH * kWH]; }
        d[c / 2] = Elem<F16>::pack2(v0, v1);
    }
}

// padded NHWC bf16 -> NCHW fp32 (parity tap)
template <bool F16>
__global__ void read_stage_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, int B, int H, int W, int C, int pad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * C * H * W;
    if (idx >= total) return;
    const int x = idx % W;
    size_t t = idx / W;
    const int y = t % H; t /= H;
    const int c = t % C;
    const int b = (int)(t / C);
    const uint16_t v = in[(((size_t)b * (H + 2 * pad) + y + pad) * (W + 2 * pad) + x + pad) * C + c];
    out[idx] = Elem<F16>::one(v);
}

// ---- IEF (PR/models/ief_module.py:48-64) in ONE launch -------------------------------------------------------------
// params_0 = init;  params_{i+1} = params_i + fc3(relu(fc2(relu(fc1([feat | params_i])))))  for i < 3:  nine dependent
// layers of 512 / 512 / 157 neurons on 32 players.  As nine launches (round 1) the layers cost ~10 us each in isolation
// and ~21 us plus a ~6 us gap inside the 4-stream pipeline, where every launch queues behind other streams' chip-filling
// convolutions: ~250 us of a stream's 1.4 ms chain (tools/chain_experiment.py: removing them is worth 9 % of throughput).
//
// Algebra first (exact in real arithmetic; the fp32 reassociation is ~1e-6, the parity tolerance 2e-5).  With
// W1 = [W1f | W1p] split by input columns and S_i = h2_0 + ... + h2_{i-1}:
//     params_i = init + i*b3 + W3 S_i
//     h1_i     = relu(W1f feat + c_i + W13 S_i),   c_i = b1 + W1p (init + i*b3),   W13 = W1p W3   (host, fp64 -> fp32)
//     h2_i     = relu(b2 + W2 h1_i)
//     result   = init + 3*b3 + W3 S_3
// so the 157-wide params vector never has to exist between layers: fc3 of one iteration and fc1 of the next collapse
// into ONE 512x512 product, the feature product W1f feat is done once, and h1_0 needs no product at all.  Seven K=512
// products and six grid-wide hand-offs instead of nine products of K = 669 / 512 / 512 and ten hand-offs.
//
// Mapping: 64 workgroups per block of 32 players stay resident for the whole IEF; workgroup g owns neurons 8g..8g+7 of
// every 512-wide layer (and of fc3 while 8g < 157), whose four 16 KB weight slices it loads into LDS once.  A thread is
// (player pair, k-slice s of 16): it holds its two players' x[k], k = 4*(16u+s)..+3, in registers (coalesced float4
// loads straight from L2 -- the activations are 64 KB and written by other workgroups), reads each neuron's weights as
// broadcast ds_read_b128 (16 lanes x 16 B contiguous: conflict-free), and the 16 k-slices of a DPP row are summed with
// four row operations.  Lane s of a row then owns output (neuron s&7, player s>>3) for the whole kernel: W1f feat and
// the running sum S live in its registers.
// Hand-off between layers (round 3): NO fences.  The hidden vectors that cross workgroups are stored with agent-scope (sc1,
// write-through) stores and read with sc1 loads to registers, which bypass the reader's L1 and see other XCDs' stores -- the
// release (L2 write-back) and acquire (L1 invalidate) of the guide's Guideline 16 barrier, ~3.4 us of the ~5 us a hand-off cost
// in round 2, are gone.  The form is the guide's (MI355X_MICROARCH.md, "Valid forms", third row of its table): every storing
// wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane adds to the monotonic counter (agent-scope atomic,
// zeroed by a memset node before the launch); the consumer's lane 0 polls it with sc1 loads (s_sleep between polls), workgroup
// barrier between the poll and EVERY load of the handed-off bytes; each 128-byte line is written whole by one store
// instruction of one wave -- for that the vectors live in a BLOCKED layout [neuron block g][player][8 neurons]: workgroup g
// writes 1 KB contiguous, thread t float t -- and every load of them is a global_load_dwordx4 sc1.  Spins are bounded; a
// time-out leaves NaNs in the outputs, raises the handle's status word and the next call returns SPSP_EHIP.
constexpr int kIefG = 64;          // workgroups per block of 32 players
constexpr int kIefLds = 4 * 8 * 512 * 4;   // four [8][512] fp32 weight slices

struct IefArgs {
    const float *feats, *w1f, *w13, *w2, *w3;   // [512][512] x3, [157][512]
    const float *cvec, *b2, *p0;                // c_i [iters][512], b2 [512], init + iters*b3 [157]
    float *h1, *s;                              // [B][512] each: written and re-read across workgroups (never const/restrict)
    float* hid;                                 // the polling form: [block of 16 players][phase 2 iters][neuron block 64][player 16][8], zero = not written yet
    float *cam, *pose, *betas;
    unsigned int* counters;                     // one monotonic arrival counter per block of 32 players
    unsigned int* err;                          // host-visible status word of the handle: a hand-off that timed out sets bit 0
    unsigned int spin_limit;                    // polls of one hand-off before it gives up
    int B, iters;
};

__device__ __forceinline__ bool ief_grid_barrier(unsigned int* counter, unsigned int target, unsigned int spin_limit) {
    __shared__ int ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every wave: its (write-through) stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        int good = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {      // global_load_dword sc1
            __builtin_amdgcn_s_sleep(2);
            if (++spins > spin_limit) { good = 0; break; }        // 2^22: ~ seconds -- a workgroup never became resident
        }
        ok = good;
    }
    __syncthreads();                                              // between the poll and every load of the handed-off bytes
    return ok != 0;
}



// 2 x 8 float4 of the two players' k-slices, every load sc1 (L1-bypassing, sees other XCDs' write-through stores); the wait is
// inside the statement: hipcc considers asm outputs written when the statement ends
template <int PPB>
__device__ __forceinline__ void ief_load_sc1(const float* x0, const float* x1, int s, bool blocked, float4 (&xa)[8], float4 (&xb)[8]) {
    const float *pa[8], *pb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = 4 * (16 * u + s);
        const int off = blocked ? ((k >> 3) * PPB) * 8 + (k & 7) : k;       // blocked: x0 / x1 point at the player's slot of block 0
        pa[u] = x0 + off;
        pb[u] = x1 + off;
    }
    asm volatile(
        "global_load_dwordx4 %0, %16, off sc1\n\tglobal_load_dwordx4 %1, %17, off sc1\n\tglobal_load_dwordx4 %2, %18, off sc1\n\t"
        "global_load_dwordx4 %3, %19, off sc1\n\tglobal_load_dwordx4 %4, %20, off sc1\n\tglobal_load_dwordx4 %5, %21, off sc1\n\t"
        "global_load_dwordx4 %6, %22, off sc1\n\tglobal_load_dwordx4 %7, %23, off sc1\n\tglobal_load_dwordx4 %8, %24, off sc1\n\t"
        "global_load_dwordx4 %9, %25, off sc1\n\tglobal_load_dwordx4 %10, %26, off sc1\n\tglobal_load_dwordx4 %11, %27, off sc1\n\t"
        "global_load_dwordx4 %12, %28, off sc1\n\tglobal_load_dwordx4 %13, %29, off sc1\n\tglobal_load_dwordx4 %14, %30, off sc1\n\t"
        "global_load_dwordx4 %15, %31, off sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(xa[0]), "=&v"(xa[1]), "=&v"(xa[2]), "=&v"(xa[3]), "=&v"(xa[4]), "=&v"(xa[5]), "=&v"(xa[6]), "=&v"(xa[7]),
          "=&v"(xb[0]), "=&v"(xb[1]), "=&v"(xb[2]), "=&v"(xb[3]), "=&v"(xb[4]), "=&v"(xb[5]), "=&v"(xb[6]), "=&v"(xb[7])
        : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pa[4]), "v"(pa[5]), "v"(pa[6]), "v"(pa[7]),
          "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]), "v"(pb[4]), "v"(pb[5]), "v"(pb[6]), "v"(pb[7])
        : "memory");
}

// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ float ief_row_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// One 8-neuron x 2-player product over K = 512: x rows from global memory (blocked: a handed-off vector, sc1 loads from the
// blocked layout; else the pooled features of the previous launch, plain rows), the weight slice from LDS.  Returns, in lane s
// of each row, the dot product for (neuron s & 7, player s >> 3).
template <int PPB>
__device__ __forceinline__ float ief_product(const float* wl, const float* x0, const float* x1, int s, bool blocked) {
    float4 xa[8], xb[8];
    ief_load_sc1<PPB>(x0, x1, s, blocked, xa, xb);
    float mine = 0.f;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 w = *reinterpret_cast<const float4*>(wl + n * 512 + 4 * (16 * u + s));
            a0 = fmaf(w.x, xa[u].x, a0); a0 = fmaf(w.y, xa[u].y, a0); a0 = fmaf(w.z, xa[u].z, a0); a0 = fmaf(w.w, xa[u].w, a0);
            a1 = fmaf(w.x, xb[u].x, a1); a1 = fmaf(w.y, xb[u].y, a1); a1 = fmaf(w.z, xb[u].z, a1); a1 = fmaf(w.w, xb[u].w, a1);
        }
        a0 = ief_row_sum(a0);
        a1 = ief_row_sum(a1);
        mine = s == n ? a0 : (s == n + 8 ? a1 : mine);
    }
    return mine;
}

// PPB = players per block (a block = the 64 workgroups that hand their hidden vectors to each other): 32 (rounds 2-4: 256 threads)
// or, round 5, 16 (128 threads): two INDEPENDENT halves per 32 players, each with its own counter -- a workgroup then pulls 32 KB
// instead of 64 KB through six cross-XCD hand-offs, and the halves' hand-offs interleave.
template <int PPB>
__global__ __launch_bounds__(8 * PPB) void ief_fused_kernel(const IefArgs a) {
    extern __shared__ __align__(16) float ief_w[];           // [4][8][512]: W1f, W13, W2, W3 rows 8g..8g+7
    constexpr int NT = 8 * PPB;                              // threads: (player pair, k-slice of 16)
    const int t = threadIdx.x, g = blockIdx.x, pb = blockIdx.y;
    const int s = t & 15, pair = t >> 4;                     // k-slice, player pair of this block
    const int bA = pb * PPB + 2 * pair, bB = bA + 1;
    const int n = 8 * g + (s & 7);                           // the neuron this lane owns ...
    const int b = (s >> 3) ? bB : bA;                        // ... for this player
    const bool live = b < a.B;
    const int rA = bA < a.B ? bA : a.B - 1, rB = bB < a.B ? bB : a.B - 1;   // idle lanes recompute the last player (no stores)
    unsigned int* const counter = a.counters + pb * 32;      // one 128-byte line per counter
    unsigned int phase = 0;

    {   // weight slices -> LDS (rows 8g.. of each matrix are 16 KB contiguous; fc3 has 157 rows)
        const float* src[4] = {a.w1f, a.w13, a.w2, a.w3};
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 1024 / NT; ++q) {
                const int e = 4 * (t + NT * q);              // float index in the [8][512] slice
                const int row = 8 * g + e / 512;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < 3 || row < kNPar) v = *reinterpret_cast<const float4*>(src[m] + (size_t)row * 512 + e % 512);
                *reinterpret_cast<float4*>(ief_w + m * 4096 + e) = v;
            }
    }
    __syncthreads();

    // hidden vectors of this block of PPB players: blocked layout [neuron block g][player][8], this thread's element (its player,
    // its neuron) = float g * NT + t of the block; players beyond B are computed (from the last real one) and stored too, so that
    // every 128-byte line is written whole by one store instruction
    float* const h1b = a.h1 + (size_t)pb * PPB * kHid;
    float* const sb = a.s + (size_t)pb * PPB * kHid;
    const int mine = g * NT + t;
    const int pA = 2 * pair, pB = 2 * pair + 1;                    // the pair's slots in the block
    const float f1 = ief_product<PPB>(ief_w, a.feats + (size_t)rA * kFeat, a.feats + (size_t)rB * kFeat, s, false);
    __hip_atomic_store(h1b + mine, fmaxf(f1 + a.cvec[n], 0.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok = ief_grid_barrier(counter, kIefG * ++phase, a.spin_limit);
    float ssum = 0.f;
    const float b2 = a.b2[n];
    for (int it = 0; it < a.iters; ++it) {
        if (it > 0) {
            const float v = ief_product<PPB>(ief_w + 4096, sb + pA * 8, sb + pB * 8, s, true);
            __hip_atomic_store(h1b + mine, fmaxf(f1 + a.cvec[it * kHid + n] + v, 0.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok &= ief_grid_barrier(counter, kIefG * ++phase, a.spin_limit);
        }
        const float v = ief_product<PPB>(ief_w + 2 * 4096, h1b + pA * 8, h1b + pB * 8, s, true);
        ssum += fmaxf(b2 + v, 0.f);
        __hip_atomic_store(sb + mine, ssum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok &= ief_grid_barrier(counter, kIefG * ++phase, a.spin_limit);
    }
    // A hand-off that gave up: the outputs become NaN (nothing downstream can mistake them for a result) AND the handle's
    // host-visible status word is set, so that the next call on the handle returns SPSP_EHIP (spsp_regressor_status).
    if (!ok && t == 0) __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (8 * g < kNPar) {
        float v = a.p0[n < kNPar ? n : 0] + ief_product<PPB>(ief_w + 3 * 4096, sb + pA * 8, sb + pB * 8, s, true);
        if (!ok) v = __builtin_nanf("");
        if (live && n < kNPar) {
            if (n < 3) a.cam[b * 3 + n] = v;
            else if (n < 147) a.pose[(size_t)b * 144 + (n - 3)] = v;
            else a.betas[b * 10 + (n - 147)] = v;
        }
    }
}

// ---- round 5: hand-offs WITHOUT a counter -- the data is its own flag ---------------------------------------------------------------
// A hand-off of ief_fused_kernel costs two trips across the fabric: the arrival counter (drain, barrier, atomic add, poll), then the
// sc1 loads of the handed-off vector.  Every hidden value is >= +0 (a ReLU output or a sum of them), so it is STORED NEGATED -- its
// sign bit always set, -0.0 included -- into a buffer that starts as zeros, one buffer per phase, and a consumer simply loads the
// 2 x 8 float4 it needs (sc1) until none of the 64 words is zero: one trip.  The negation is undone by the FMA's source modifier,
// so the sums are ief_fused_kernel's bit for bit.  No workgroup barrier anywhere: a wave leaves its poll loop when all its lanes
// have their data (the DPP row sums need the whole row).  The buffers are zeroed by the forward's first kernel (the stem, every
// workgroup a share: 393 KB at B = 32) -- or by a memset when the IEF is called without a backbone call in front.
constexpr int kIefPhaseFloats = kIefG * 16 * 8;     // one phase of one block of 16 players: 32 KB

// 2 x U float4 of the two players' k-slices (U = 128 / KS; KS lanes share a product), sc1
template <int KS>
__device__ __forceinline__ void ief_load_sc1_ks(const float* x0, const float* x1, int s, float4 (&xa)[128 / KS], float4 (&xb)[128 / KS]) {
    if constexpr (KS == 16) ief_load_sc1<16>(x0, x1, s, true, xa, xb);
    else {
        const float *pa[4], *pb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * (32 * u + s);
            const int off = ((k >> 3) * 16) * 8 + (k & 7);
            pa[u] = x0 + off;
            pb[u] = x1 + off;
        }
        asm volatile(
            "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
            "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
            "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
            : "=&v"(xa[0]), "=&v"(xa[1]), "=&v"(xa[2]), "=&v"(xa[3]), "=&v"(xb[0]), "=&v"(xb[1]), "=&v"(xb[2]), "=&v"(xb[3])
            : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3])
            : "memory");
    }
}

template <int KS>
__device__ __forceinline__ bool ief_load_poll(const float* x0, const float* x1, int s, float4 (&xa)[128 / KS], float4 (&xb)[128 / KS], unsigned int spin_limit) {
    unsigned int spins = 0;
    for (;;) {
        ief_load_sc1_ks<KS>(x0, x1, s, xa, xb);
        unsigned int m = 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < 128 / KS; ++u) {
            m = min(m, min(__builtin_bit_cast(unsigned int, xa[u].x), __builtin_bit_cast(unsigned int, xa[u].y)));
            m = min(m, min(__builtin_bit_cast(unsigned int, xa[u].z), __builtin_bit_cast(unsigned int, xa[u].w)));
            m = min(m, min(__builtin_bit_cast(unsigned int, xb[u].x), __builtin_bit_cast(unsigned int, xb[u].y)));
            m = min(m, min(__builtin_bit_cast(unsigned int, xb[u].z), __builtin_bit_cast(unsigned int, xb[u].w)));
        }
        if (__builtin_amdgcn_ballot_w64(m == 0u) == 0ull) return true;
        if (++spins > spin_limit) return false;           // (wave-uniform: the ballot is)
        __builtin_amdgcn_s_sleep(2);
    }
}

// the sum over a PAIR of DPP rows (32 lanes), in every lane: v_permlane16_swap of the value with a COPY of itself (two registers:
// swapping a register with itself exchanges its own rows) leaves rows 0 / 2 in one result and rows 1 / 3 in the other
__device__ __forceinline__ float ief_row_pair_sum(float v) {
    v = ief_row_sum(v);
    float c = v;
    // (as asm: with both operands the same value hipcc's builtin form added the first result to itself; the wait states between a
    // VALU write and the swap's read are the asm's own)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(c));
    return v + c;
}

// a product over K = 512 on a polled (negated) vector; KS lanes share it (k-slices), lanes s < 16 of the group end with
// (neuron s & 7, player s >> 3); the stored sign is undone by the FMA's source modifier.  KS = 16: ief_product's FMAs in its order.
template <int KS>
__device__ __forceinline__ float ief_product_poll(const float* wl, const float* x0, const float* x1, int s, unsigned int spin_limit, bool& ok) {
    constexpr int U = 128 / KS;
    float4 xa[U], xb[U];
    ok &= ief_load_poll<KS>(x0, x1, s, xa, xb, spin_limit);
    float mine = 0.f;
    const int sl = s & 15;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 w = *reinterpret_cast<const float4*>(wl + n * 512 + 4 * (KS * u + s));
            a0 = fmaf(w.x, -xa[u].x, a0); a0 = fmaf(w.y, -xa[u].y, a0); a0 = fmaf(w.z, -xa[u].z, a0); a0 = fmaf(w.w, -xa[u].w, a0);
            a1 = fmaf(w.x, -xb[u].x, a1); a1 = fmaf(w.y, -xb[u].y, a1); a1 = fmaf(w.z, -xb[u].z, a1); a1 = fmaf(w.w, -xb[u].w, a1);
        }
        a0 = KS == 32 ? ief_row_pair_sum(a0) : ief_row_sum(a0);
        a1 = KS == 32 ? ief_row_pair_sum(a1) : ief_row_sum(a1);
        mine = sl == n ? a0 : (sl == n + 8 ? a1 : mine);
    }
    return mine;
}

// the first product: the pooled features of the previous launch (plain rows, plain loads), KS lanes per player pair
template <int KS>
__device__ __forceinline__ float ief_product_feats(const float* wl, const float* x0, const float* x1, int s) {
    if constexpr (KS == 16) return ief_product<16>(wl, x0, x1, s, false);
    else {
        float mine = 0.f;
        float4 xa[4], xb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xa[u] = *reinterpret_cast<const float4*>(x0 + 4 * (32 * u + s));
            xb[u] = *reinterpret_cast<const float4*>(x1 + 4 * (32 * u + s));
        }
        const int sl = s & 15;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 w = *reinterpret_cast<const float4*>(wl + n * 512 + 4 * (32 * u + s));
                a0 = fmaf(w.x, xa[u].x, a0); a0 = fmaf(w.y, xa[u].y, a0); a0 = fmaf(w.z, xa[u].z, a0); a0 = fmaf(w.w, xa[u].w, a0);
                a1 = fmaf(w.x, xb[u].x, a1); a1 = fmaf(w.y, xb[u].y, a1); a1 = fmaf(w.z, xb[u].z, a1); a1 = fmaf(w.w, xb[u].w, a1);
            }
            a0 = ief_row_pair_sum(a0);
            a1 = ief_row_pair_sum(a1);
            mine = sl == n ? a0 : (sl == n + 8 ? a1 : mine);
        }
        return mine;
    }
}

template <int KS>
__global__ __launch_bounds__(8 * KS) void ief_poll_kernel(const IefArgs a) {
    extern __shared__ __align__(16) float ief_w[];           // [4][8][512]: W1f, W13, W2, W3 rows 8g..8g+7
    constexpr int PPB = 16, NT = 8 * KS;                     // eight player pairs x KS lanes
    const int t = threadIdx.x, g = blockIdx.x, pb = blockIdx.y;
    const int s = t % KS, pair = t / KS;
    const bool owner = s < 16;                               // the lanes that own an output: (neuron s & 7, player s >> 3) of the pair
    const int bA = pb * PPB + 2 * pair, bB = bA + 1;
    const int n = 8 * g + (s & 7);
    const int b = ((s >> 3) & 1) ? bB : bA;
    const bool live = owner && b < a.B;
    const int rA = bA < a.B ? bA : a.B - 1, rB = bB < a.B ? bB : a.B - 1;
    {
        const float* src[4] = {a.w1f, a.w13, a.w2, a.w3};
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 1024 / NT; ++q) {
                const int e = 4 * (t + NT * q);
                const int row = 8 * g + e / 512;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < 3 || row < kNPar) v = *reinterpret_cast<const float4*>(src[m] + (size_t)row * 512 + e % 512);
                *reinterpret_cast<float4*>(ief_w + m * 4096 + e) = v;
            }
    }
    __syncthreads();
    float* const hid = a.hid + (size_t)pb * (2 * a.iters) * kIefPhaseFloats;      // phase 2 it: h1 of iteration it; 2 it + 1: S after it
    const int mine = g * 128 + 16 * pair + (s & 15);                               // [g][player][8]: a 128-byte line per wave-store
    const int pA = 2 * pair, pB = 2 * pair + 1;
    bool ok = true;
    const float f1 = ief_product_feats<KS>(ief_w, a.feats + (size_t)rA * kFeat, a.feats + (size_t)rB * kFeat, s);
    if (owner) __hip_atomic_store(hid + mine, -fmaxf(f1 + a.cvec[n], 0.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float ssum = 0.f;
    const float b2 = a.b2[n];
    for (int it = 0; it < a.iters; ++it) {
        float* const h1p = hid + (size_t)(2 * it) * kIefPhaseFloats;
        float* const sp = h1p + kIefPhaseFloats;
        if (it > 0) {
            const float* const sprev = h1p - kIefPhaseFloats;
            const float v = ief_product_poll<KS>(ief_w + 4096, sprev + pA * 8, sprev + pB * 8, s, a.spin_limit, ok);
            if (owner) __hip_atomic_store(h1p + mine, -fmaxf(f1 + a.cvec[it * kHid + n] + v, 0.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float v = ief_product_poll<KS>(ief_w + 2 * 4096, h1p + pA * 8, h1p + pB * 8, s, a.spin_limit, ok);
        ssum += fmaxf(b2 + v, 0.f);
        if (owner) __hip_atomic_store(sp + mine, -ssum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (8 * g < kNPar) {
        const float* const slast = hid + (size_t)(2 * a.iters - 1) * kIefPhaseFloats;
        float v = a.p0[n < kNPar ? n : 0] + ief_product_poll<KS>(ief_w + 3 * 4096, slast + pA * 8, slast + pB * 8, s, a.spin_limit, ok);
        if (!ok) v = __builtin_nanf("");
        if (live && n < kNPar) {
            if (n < 3) a.cam[b * 3 + n] = v;
            else if (n < 147) a.pose[(size_t)b * 144 + (n - 3)] = v;
            else a.betas[b * 10 + (n - 147)] = v;
        }
    }
    // a poll that gave up: NaN outputs (above, where this workgroup has any) and the handle's host-visible status word
    if (!ok && (t & 63) == 0) __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// IEF with no iterations (never used by the reference): outputs = the initial estimate
__global__ void ief_init_only_kernel(const float* __restrict__ init, int B, float* __restrict__ cam, float* __restrict__ pose, float* __restrict__ betas) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * kNPar) return;
    const int b = i / kNPar, jj = i % kNPar;
    const float v = init[jj];
    if (jj < 3) cam[b * 3 + jj] = v;
    else if (jj < 147) pose[b * 144 + (jj - 3)] = v;
    else betas[b * 10 + (jj - 147)] = v;
}

}  // namespace

struct spsp_regressor {
    spsp::DevicePool pool;
    int max_batch;
    int ief_iters;
    int f16;               // 0: bf16 storage + bf16 MFMA (precision 0), 1: fp16 storage + f16 MFMA (precision 1)
    uint16_t* w[SPSP_NUM_CONVS];
    float* bias[SPSP_NUM_CONVS];
    uint16_t* wrs[3] = {nullptr, nullptr, nullptr};   // layer2's three 3x3 / 1 convs (6, 8, 9) as register-order fragments (conv_rs.hip)
    int rs_wgs = 0;        // layer2's 3x3 / 1 convs on conv_rs.hip with this many workgroups (plan "throughput": 64); 0: conv_p4.hip
    uint16_t* wq2 = nullptr; // layer2.0's 3x3 / 2 + 1x1 / 2 pair as register-order MFMA fragments (conv_s2q.hip)
    bool use_s2q = true;   // layer2.0 on conv_s2q.hip (all 128 output channels per workgroup); off: conv_s2rs.hip (SPSP_DEBUG_PLAN_S2Q)
    uint16_t* wfb[2];      // layer1's two BasicBlocks: both convolutions' weights as register-order MFMA fragments (conv_fb.hip)
    int stem32 = 1;        // the stem on v_mfma_f32_32x32x16: 1 = pooling inside the K loop (conv1_pool32r_kernel), 2 = behind it (conv1_pool32_kernel); 0: the 16x16x32 form (SPSP_DEBUG_PLAN_STEM32)
    bool use_fb = true;    // layer1: one launch per BasicBlock (conv_fb.hip); off: two conv_wres launches per block (SPSP_DEBUG_PLAN_FB)
    float *w1f, *w13, *w2, *w3;   // IEF weights in the fused kernel's form (see ief_fused_kernel)
    float *cvec, *b2, *p0;
    float* init_params;
    // feature maps (bf16, zero border)
    uint16_t* x0;          // [B][262][262][18]
    uint16_t* p1;          // [B][66][66][64] max-pool output
    uint16_t* l[4][3];     // layer L: [B][S+2][S+2][C], S = 64,32,16,8; C = 64,128,256,512
    float *feats, *h1, *h2;
    const uint16_t* stage_ptr[6];
    int last_B;
    unsigned int* counters;  // tile-scheduler words of the persistent kernels (zeroed at the start of every forward)
    int next_counter;
    RingTile ring[4];      // per residual stage: tile of the ring kernel (stride-2 / 1x1 convs; every conv of a ragged batch)
    bool patch_db[4];      // per residual stage: two patch buffers in the patch kernel
    int patch_tile[4];     // per residual stage: output pixels per patch-kernel tile (256: 8 waves, 128: 4 waves)
    int patch_ring[4];     // per residual stage: depth of the patch kernel's weight ring, in groups of three slabs (2 or 3)
    int patch_nch[4];      // per residual stage: output channels per patch-kernel workgroup (64; 32 = twice the workgroups, layer4)
    bool patch_prod[4];    // per residual stage: four producer waves issue the patch kernel's LDS-DMA (128-pixel tiles only)
    unsigned int* err_host = nullptr;   // pinned, mapped, coherent host word: device-side failures of this handle's kernels (bit 0: IEF hand-off time-out)
    unsigned int* err_dev = nullptr;    // the same word as the kernels see it
    int ief_capacity = 0;               // IEF workgroups the device can hold at once (occupancy x CUs)
    int debug_ief_stall = 0;
    bool use_s2 = true;                 // stride-2 stages on conv_s2.hip (experiment builds: SPSP_NO_S2 = the ring kernel of round 2)
    // bit L: the 3x3 / 1 convs of residual stage L on conv_p4.hip (256-pixel tiles, no K split, producer waves).  layer2 always
    // (256 workgroups at B = 32: 13.1 -> 11.7 us per launch).  layer3 = one image per workgroup, 128 workgroups at B = 32: 12.0 ->
    // 14.9 us with one batch in flight, but 40 % less CU time -- the four-slot pipeline gains 1.6-1.9 % (tools/ab_pipeline.py):
    // plan "throughput" only
    int p4_layers = 2;
    // Plan "throughput": with four batches in flight on four streams a kernel that fills the chip for a short time competes with
    // three others for every CU, and pays its prologue (layer1: 72 KB of weights + the first patch per workgroup, a third of its
    // run time at two tiles per workgroup) once per workgroup.  Persistent kernels are therefore launched NARROW there: layer1 on a
    // quarter of the CUs (8 tiles per resident weight set), the stem in bands of 32 conv rows (128 workgroups at B = 32).  Four-slot
    // pipeline, same box: 120.8 k -> 127.6 k players/s with layer1 on 1/4 (1/8: 124.3 k, 1/16: 106.9 k), 128.6 k with the stem bands
    // as well (bands of 128 rows = 32 workgroups: 110.9 k) -- tools/ab_pipeline.py, profiles/r04_ab_narrow.txt.  One batch in flight
    // pays for it (layer1 61 -> ~240 us): plan "latency" keeps the chip-filling grids.  Round 4, later: layer2.0 (conv_s2rs: 80 KB of
    // weights in registers per workgroup) with four tiles per workgroup instead of two: +1.6 % (eight: the same).
    int s2rs_cu_div = 1;                // conv_s2rs: tiles per resident weight set as if the chip had num_cus / this CUs (2 -> 4 -> 8 tiles)
    int p4_cu_div = 1;                  // conv_p4's workgroups: at most num_cus / this (persistent over the tiles beyond)
    int stem_band = 0;                  // conv rows per stem workgroup when coarser than the grid-filling default (32, 64)
    int wres_cu_div = 1;                // layer1's persistent workgroups: num_cus / this (2: twice the tiles per workgroup, half the chip)
    bool use_s2rs = true;               // layer2.0 on conv_s2rs.hip (weights stationary in registers); off: conv_s2.hip like the other two
    // The fused IEF's arrival counters are monotonic and must start at zero.  A backbone call zeroes them (the stem kernel's first
    // workgroup does, with the tile-scheduler words); an IEF call uses them up.  spsp_regressor_ief on a handle whose counters are
    // not fresh (IEF run twice on the same features, IEF before any backbone) zeroes its eight lines itself with a memset node.
    bool ief_counters_fresh = false;
    float* ief_hid = nullptr;           // the polling IEF's per-phase hand-off buffers (zeroed by every forward's stem kernel)
    size_t ief_hid_block = 0;           // floats per block of 16 players: 2 iters x 32 KB
    int ief_poll = 2;                   // 2: ief_poll_kernel<32> (the data is its own flag; 32 lanes per player pair), 1: <16> (ief_fused_kernel's sums bit for bit), 0: ief_fused_kernel (arrival counters; SPSP_DEBUG_IEF_POLL)
    int ief_block = 16;                 // players per IEF block (64 workgroups that hand off to each other): 32, or 16 = two independent halves
    ~spsp_regressor() { if (err_host) (void)hipHostFree(err_host); }
};

namespace {

const int kLayerS[4] = {64, 32, 16, 8};
const int kLayerC[4] = {64, 128, 256, 512};

// conv `ci` reading map `in` (spatial Si, channels Ci, border bi) writing map `out` (spatial So, border 1)
// (ci2 >= 0: the 1x1 stride-2 downsample conv `ci2` of the stage rides in the same launch, writing out2)
int run_conv(spsp_ctx* ctx, spsp_regressor* r, int ci, const uint16_t* in, int Si, int bi, uint16_t* out, const uint16_t* res, int relu,
             int B, float* pool_out, hipStream_t st, int L, int ci2 = -1, uint16_t* out2 = nullptr) {
    const ConvDesc& d = kConvs[ci];
    ConvArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = in; a.w = r->w[ci]; a.bias = r->bias[ci]; a.res = res; a.out = out; a.pool_out = pool_out;
    a.f16 = r->f16;
    const int Wi = Si + 2 * bi;
    a.in_pix = d.cin; a.in_row = Wi * d.cin; a.in_img = Wi * Wi * d.cin;
    const int conv_pad = d.k / 2;
    a.in_off = (bi - conv_pad) * (a.in_row + a.in_pix);
    const int So = Si / d.stride;
    a.out_pix = d.cout; a.out_row = (So + 2) * d.cout; a.out_img = (So + 2) * (So + 2) * d.cout;
    a.out_off = a.out_row + a.out_pix;
    a.Ho = So; a.Wo = So; a.M = B * So * So; a.N = d.cout;
    a.stride = d.stride;
    a.kw_steps = d.k; a.chunks = d.cin / 64; a.nsteps = d.k * d.k * a.chunks;
    a.relu = relu;
    a.splitk = 1;
    a.counter = r->counters + 32 * 8 * (r->next_counter < 3 ? r->next_counter : 3);   // 8 scheduler words (one per XCD class), a 128-byte line each
    if (conv_wres_supported(a)) {                      // layer1: 3x3 / 1, 64 -> 64, weights resident in LDS
        ++r->next_counter;
        ProfScope ps(ctx, kKConvWres, st, true);
        return launch_conv_wres(a, r->wres_cu_div >= 32 ? r->wres_cu_div : ctx->num_cus / r->wres_cu_div, st);      // (>= 32: an explicit workgroup count, experiments)
    }
    if (L == 1 && r->rs_wgs >= 2 && (ci == 6 || ci == 8 || ci == 9) && conv_patch_supported(a) && a.Wo == 32 && a.N == 128 && !pool_out) {
        ProfScope ps(ctx, kKConvRs, st, true);          // plan "throughput": a channel block's weights in registers, persistent workgroups
        return launch_conv_rs(a.in, r->wrs[ci == 6 ? 0 : ci - 7], a.bias, a.res, a.out, B, r->f16, a.relu, r->rs_wgs, st);
    }
    if (conv_patch_supported(a)) {                     // 3x3 / 1 of layer2-4 at whole multiples of 128 pixels
        if (((r->p4_layers >> L) & 1) && conv_p4_supported(a)) {       // layer2; layer3 in plan "throughput": no K split, producer waves
            ProfScope ps(ctx, kKConvP4, st, true);
            return launch_conv_p4(a, ctx->num_cus / r->p4_cu_div, st);
        }
        ProfScope ps(ctx, a.Wo == 8 ? kKPatchW8 : kKPatchW16, st, true);
        // (8 x 8 maps: a batch of 2 mod 4 crops -- the 22-player frame -- is whole 128-pixel tiles, two images each, but no whole 256-pixel ones)
        const int tile = a.M % 256 == 0 ? r->patch_tile[L] : 128;
        return launch_conv_patch(a, r->patch_db[L], tile, r->patch_ring[L], r->patch_nch[L],
                                 r->patch_prod[L] && tile == 128 && r->patch_ring[L] == 3 && r->patch_db[L], st);
    }
    // stride-2 3x3, 1x1 downsample, and every conv of a batch that does not fill whole patch tiles: LDS-DMA ring kernel
    if (ci2 >= 0) { a.w2 = r->w[ci2]; a.bias2 = r->bias[ci2]; a.out2 = out2; a.n2 = kConvs[ci2].cout; }
    if (r->use_s2 && r->use_s2q && r->wq2 && conv_s2rs_supported(a) && a.N == 128 && a.n2 == 128) {      // layer2.0, all channels per workgroup
        ProfScope ps(ctx, kKConvS2q, st, true);
        return launch_conv_s2q(a.in, r->wq2, a.bias, a.bias2, a.out, a.out2, B, r->f16, a.relu, ctx->num_cus / r->s2rs_cu_div, st);
    }
    if (r->use_s2 && r->use_s2rs && conv_s2rs_supported(a)) {   // layer2.0: the same planes, weights stationary in registers
        ProfScope ps(ctx, kKConvS2, st, true);
        return launch_conv_s2rs(a, ctx->num_cus / r->s2rs_cu_div, st);
    }
    if (r->use_s2 && conv_s2_supported(a)) {           // 3x3 / 2 + its 1x1 / 2 downsample in whole 128-pixel tiles: parity-plane patch kernel
        ProfScope ps(ctx, kKConvS2, st, true);
        return launch_conv_s2(a, st);
    }
    const RingTile rt = r->ring[L];
    const int bm = (rt == kRing64x128s3 || rt == kRing64x64s4) ? 64 : 128;
    const int bn = (rt == kRing128x64s3 || rt == kRing128x64s4 || rt == kRing64x64s4) ? 64 : 128;
    SPSP_REQUIRE(a.M % bm == 0 && a.N % bn == 0, "regressor: conv %s at batch %d does not tile (%d x %d)", d.name, B, bm, bn);
    ProfScope ps(ctx, kKConvRing, st, true);
    return launch_conv_ring(a, rt, st);
}

}  // namespace

extern "C" {

const char* spsp_conv_name(int i) { return (i >= 0 && i < SPSP_NUM_CONVS) ? kConvs[i].name : nullptr; }
const char* spsp_conv_bn_name(int i) { return (i >= 0 && i < SPSP_NUM_CONVS) ? kConvs[i].bn : nullptr; }
size_t spsp_proxy_native_bytes(void) { return kNativeElems * 2; }

int spsp_regressor_load(spsp_ctx* ctx, const spsp_regressor_weights* w, int max_batch, int precision,
                        spsp_regressor** out) {
    SPSP_REQUIRE(ctx && w && out && max_batch > 0, "spsp_regressor_load: bad arguments");
    SPSP_REQUIRE(precision == 0 || precision == 1, "spsp_regressor_load: precision %d (0 = bf16 MFMA, 1 = fp16 MFMA)", precision);
    SPSP_REQUIRE(w->ief_iters >= 0 && w->ief_iters <= 16, "spsp_regressor_load: ief_iters=%d", w->ief_iters);
    // the fused IEF holds eight blocks of 32 players per launch (one barrier counter line each)
    SPSP_REQUIRE(max_batch <= SPSP_REGRESSOR_MAX_BATCH, "spsp_regressor_load: max_batch=%d exceeds %d players per call", max_batch, SPSP_REGRESSOR_MAX_BATCH);
    for (int i = 0; i < SPSP_NUM_CONVS; ++i)
        SPSP_REQUIRE(w->conv[i].weight && w->conv[i].bn_weight && w->conv[i].bn_bias && w->conv[i].bn_mean && w->conv[i].bn_var,
                     "spsp_regressor_load: NULL array for %s", kConvs[i].name);
    for (int i = 0; i < 3; ++i) SPSP_REQUIRE(w->fc_weight[i] && w->fc_bias[i], "spsp_regressor_load: NULL fc%d", i + 1);
    SPSP_REQUIRE(w->init_params, "spsp_regressor_load: NULL init_params");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    spsp_regressor* r = new spsp_regressor;
    r->max_batch = max_batch;
    r->ief_iters = w->ief_iters;
    r->f16 = precision;
    r->last_B = 0;
    int rc = SPSP_OK;
#define TRY(x) do { rc = (x); if (rc != SPSP_OK) { delete r; return rc; } } while (0)
    for (int ci = 0; ci < SPSP_NUM_CONVS; ++ci) {
        const ConvDesc& d = kConvs[ci];
        const spsp_conv_bn& c = w->conv[ci];
        const int ktot = ci == 0 ? 7 * 128 : d.k * d.k * d.cin;
        std::vector<uint16_t> wt((size_t)d.cout * ktot, 0);
        std::vector<float> bias(d.cout);
        for (int n = 0; n < d.cout; ++n) {
            SPSP_REQUIRE(c.bn_var[n] + 1e-5 > 0.0, "spsp_regressor_load: %s running_var[%d] <= -eps", d.bn, n);
            const double s = (double)c.bn_weight[n] / std::sqrt((double)c.bn_var[n] + 1e-5);
            bias[n] = (float)((double)c.bn_bias[n] - (double)c.bn_mean[n] * s);
            for (int cin = 0; cin < d.cin; ++cin)
                for (int kr = 0; kr < d.k; ++kr)
                    for (int kc = 0; kc < d.k; ++kc) {
                        const float wv = (float)((double)c.weight[(((size_t)n * d.cin + cin) * d.k + kr) * d.k + kc] * s);
                        const size_t kidx = ci == 0 ? (size_t)kr * 128 + kc * d.cin + cin
                                                    : ((size_t)kr * d.k + kc) * d.cin + cin;
                        wt[(size_t)n * ktot + kidx] = precision ? f32_to_f16_host(wv) : f32_to_bf16_host(wv);
                    }
        }
        TRY(r->pool.upload(&r->w[ci], wt.data(), wt.size()));
        TRY(r->pool.upload(&r->bias[ci], bias.data(), bias.size()));
        if (ci == 6 || ci == 8 || ci == 9) {           // layer2's 3x3 / 1 convs for conv_rs.hip
            std::vector<uint16_t> frag;
            pack_rs_fragments(wt.data(), frag);
            TRY(r->pool.upload(&r->wrs[ci == 6 ? 0 : ci - 7], frag.data(), frag.size()));
        }
        if (ci == 5 || ci == 7) {                      // layer2.0: the 3x3 / 2 conv (5) and its downsample (7) for conv_s2q.hip
            static thread_local std::vector<uint16_t> main3x3;
            if (ci == 5) main3x3 = wt;
            else {
                std::vector<uint16_t> frag;
                pack_s2q_fragments(main3x3.data(), wt.data(), frag);
                TRY(r->pool.upload(&r->wq2, frag.data(), frag.size()));
            }
        }
        if (ci >= 1 && ci <= 4) {                      // layer1: convs 1, 2 = block 0; 3, 4 = block 1
            static thread_local std::vector<uint16_t> first;
            if (ci & 1) first = wt;
            else {
                std::vector<uint16_t> frag;
                pack_block_fragments(first.data(), wt.data(), frag);
                TRY(r->pool.upload(&r->wfb[(ci - 1) / 2], frag.data(), frag.size()));
            }
        }
    }
    {   // fc1 [512][669] split by input columns (ief_module.py:55 concatenates [feat | params]); products in fp64
        const int it_n = w->ief_iters > 0 ? w->ief_iters : 1;
        std::vector<float> w1f((size_t)kHid * kFeat), w13((size_t)kHid * kHid), cvec((size_t)it_n * kHid), p0(kNPar);
        std::vector<double> acc(kHid);
        const float *W1 = w->fc_weight[0], *W3 = w->fc_weight[2], *b1 = w->fc_bias[0], *b3 = w->fc_bias[2];
        for (int n = 0; n < kHid; ++n) {
            const float* row = W1 + (size_t)n * (kFeat + kNPar);
            std::memcpy(&w1f[(size_t)n * kFeat], row, kFeat * sizeof(float));
            std::fill(acc.begin(), acc.end(), 0.0);
            for (int p = 0; p < kNPar; ++p) {
                const double wp = row[kFeat + p];
                const float* w3row = W3 + (size_t)p * kHid;
                for (int k = 0; k < kHid; ++k) acc[k] += wp * (double)w3row[k];
            }
            for (int k = 0; k < kHid; ++k) w13[(size_t)n * kHid + k] = (float)acc[k];
            for (int it = 0; it < it_n; ++it) {
                double c = b1[n];
                for (int p = 0; p < kNPar; ++p) c += (double)row[kFeat + p] * ((double)w->init_params[p] + (double)it * (double)b3[p]);
                cvec[(size_t)it * kHid + n] = (float)c;
            }
        }
        for (int p = 0; p < kNPar; ++p) p0[p] = (float)((double)w->init_params[p] + (double)w->ief_iters * (double)b3[p]);
        TRY(r->pool.upload(&r->w1f, w1f.data(), w1f.size()));
        TRY(r->pool.upload(&r->w13, w13.data(), w13.size()));
        TRY(r->pool.upload(&r->w2, w->fc_weight[1], (size_t)kHid * kHid));
        TRY(r->pool.upload(&r->w3, W3, (size_t)kNPar * kHid));
        TRY(r->pool.upload(&r->cvec, cvec.data(), cvec.size()));
        TRY(r->pool.upload(&r->b2, w->fc_bias[1], (size_t)kHid));
        TRY(r->pool.upload(&r->p0, p0.data(), p0.size()));
    }
    TRY(r->pool.upload(&r->init_params, w->init_params, (size_t)kNPar));
    const size_t B = max_batch;
    TRY(r->pool.alloc(&r->x0, B * kNativeElems, true));
    TRY(r->pool.alloc(&r->p1, B * 66 * 66 * 64, true));
    // (layer3's and layer4's maps and the feature rows hold one crop more: an odd batch runs layer4 on B + 1 crops -- below)
    for (int L = 0; L < 4; ++L)
        for (int j = 0; j < 3; ++j)
            TRY(r->pool.alloc(&r->l[L][j], (B + (L >= 2 ? 1 : 0)) * (kLayerS[L] + 2) * (kLayerS[L] + 2) * kLayerC[L], true));
    TRY(r->pool.alloc(&r->counters, kCounterWords, true));
    {   // status word the kernels can raise and the host can read without a synchronisation
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { delete r; set_error("spsp_regressor_load: hipHostMalloc(status word) failed"); return SPSP_ENOMEM; }
        r->err_host = static_cast<unsigned int*>(hp);
        std::memset(hp, 0, 64);
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { delete r; set_error("spsp_regressor_load: hipHostGetDevicePointer failed"); return SPSP_EHIP; }
        r->err_dev = static_cast<unsigned int*>(dp);
    }
    {   // The fused IEF spins on grid-wide hand-offs under an ordinary launch, so the 64 workgroups of a block of 32 players
        // must be able to be resident TOGETHER.  Every other kernel of the path runs to completion without waiting on
        // anything, and a launch's workgroups are dispatched in order, so at most ONE unit per concurrent IEF launch is
        // ever partly resident; progress is guaranteed while (concurrent forwards) x 64 fits into what the device holds:
        // occupancy x CUs (2 per CU at 64 KB of LDS: 512 on MI355X = 8 concurrent forwards).  Checked here, documented in
        // spsp.h; a hand-off that still starves times out and is reported (spsp_regressor_status).
        TRY(ensure_dynamic_lds<&ief_fused_kernel<32>>(kIefLds));
        TRY(ensure_dynamic_lds<&ief_fused_kernel<16>>(kIefLds));
        // (a unit = the 64 workgroups of one block, of 32 or of 16 players: 64 KB of LDS each either way)
        int per_cu = 0, per_cu16 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ief_fused_kernel<32>, 256, kIefLds) != hipSuccess) per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu16, ief_fused_kernel<16>, 128, kIefLds) != hipSuccess) per_cu16 = 0;
        if (per_cu16 < per_cu) per_cu = per_cu16;
        TRY(ensure_dynamic_lds<&ief_poll_kernel<16>>(kIefLds));
        TRY(ensure_dynamic_lds<&ief_poll_kernel<32>>(kIefLds));
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu16, ief_poll_kernel<16>, 128, kIefLds) != hipSuccess) per_cu16 = 0;
        if (per_cu16 < per_cu) per_cu = per_cu16;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu16, ief_poll_kernel<32>, 256, kIefLds) != hipSuccess) per_cu16 = 0;
        if (per_cu16 < per_cu) per_cu = per_cu16;
        r->ief_capacity = per_cu * ctx->num_cus;
        if (w->ief_iters > 0 && r->ief_capacity < kIefG * SPSP_REGRESSOR_MAX_CONCURRENT) {
            const int cap = r->ief_capacity;
            delete r;
            set_error("spsp_regressor_load: the device holds %d IEF workgroups at once, %d concurrent forwards need %d", cap, SPSP_REGRESSOR_MAX_CONCURRENT, kIefG * SPSP_REGRESSOR_MAX_CONCURRENT);
            return SPSP_EINVAL;
        }
    }
    {   // plan of the residual stages (ONE plan; measured at 32 crops)
        const RingTile def_t[4] = {kRing128x64s3, kRing64x128s3, kRing64x128s3, kRing64x64s4};
        // two patch buffers where the layer has fewer tiles than the chip has CUs (layer3, layer4); layer2 fills every
        // CU exactly once and keeps ONE (89 KB), so that a fit workgroup (38 KB) already resident on a CU does not push
        // its tile into a second round
        const bool def_db[4] = {false, false, true, true};
        // 128-pixel tiles where 256-pixel ones cover half (layer3: 128) or a quarter (layer4: 64) of the 256 CUs
        const int def_tile[4] = {256, 256, 128, 128};
        const int def_ring[4] = {3, 3, 3, 3};
        // layer4: 64-channel blocks (128 workgroups) for throughput -- with several batches in flight what counts is the CU time a
        // launch consumes, and 256 workgroups of 32 channels stage the input patch twice as often (A/B in one process,
        // tools/ab_pipeline.py: 116.0 k vs 113.4 k players/s); a single batch in flight finishes sooner on 256 workgroups
        // (18.9 -> 13.2 us per launch, tools/ab_plan.py): spsp_regressor_set_plan(SPSP_PLAN_LATENCY)
        // (a fresh handle is in plan "latency": spsp_regressor_set_plan)
        const int def_nch[4] = {64, 64, 64, 32};
        // layer3 / layer4 (128-pixel tiles, four accumulating waves): four DMA-only producer waves per workgroup (conv_patch.hip,
        // PROD) -- layer4's launch 18.2 -> 15.5 us (64-channel blocks), 13.0 -> 12.6 (32), layer3's 12.7 -> 12.1, bit-identical
        // results (tools/ab_plan.py, run g3aa); in the four-slot pipeline the gain is within the noise (+0.4 %, tools/ab_pipeline.py)
        const bool def_prod[4] = {false, false, true, true};
        for (int L = 0; L < 4; ++L) { r->ring[L] = def_t[L]; r->patch_db[L] = def_db[L]; r->patch_tile[L] = def_tile[L]; r->patch_ring[L] = def_ring[L]; r->patch_nch[L] = def_nch[L]; r->patch_prod[L] = def_prod[L]; }
#if defined(SPSP_TUNING) || defined(SPSP_CONV_STAMPS)   // experiment builds only (TUNING=1, or the stamped twin library)
        if (getenv("SPSP_NO_PATCH_PRODUCERS")) for (int L = 0; L < 4; ++L) r->patch_prod[L] = false;
        if (const char* e = getenv("SPSP_PATCH_TILE")) (void)sscanf(e, "%d,%d,%d,%d", &r->patch_tile[0], &r->patch_tile[1], &r->patch_tile[2], &r->patch_tile[3]);
        if (const char* e = getenv("SPSP_PATCH_RING")) (void)sscanf(e, "%d,%d,%d,%d", &r->patch_ring[0], &r->patch_ring[1], &r->patch_ring[2], &r->patch_ring[3]);
        if (const char* e = getenv("SPSP_RING_TILE")) {      // RingTile enum values: 0 128x128s3, 1 128x128s2, 2 128x64s3, 3 128x64s4, 4 64x128s3, 5 64x64s4
            int d[4] = {2, 4, 4, 5};
            (void)sscanf(e, "%d,%d,%d,%d", &d[0], &d[1], &d[2], &d[3]);
            for (int L = 0; L < 4; ++L) if (d[L] >= 0 && d[L] < (int)kRingNone) r->ring[L] = (RingTile)d[L];
        }
        if (getenv("SPSP_NO_S2")) r->use_s2 = false;
        if (const char* e = getenv("SPSP_P4")) r->p4_layers = atoi(e) & 6;
        if (const char* e = getenv("SPSP_PATCH_DB")) {
            int d[4] = {0, 0, 1, 1};
            (void)sscanf(e, "%d,%d,%d,%d", &d[0], &d[1], &d[2], &d[3]);
            for (int L = 0; L < 4; ++L) r->patch_db[L] = d[L] != 0;
        }
#endif
    }
    TRY(r->pool.alloc(&r->feats, (B + 1) * kFeat));
    TRY(r->pool.alloc(&r->h1, (B + 31) / 32 * 32 * kHid));      // blocked hand-off layout: whole blocks of 32 players
    TRY(r->pool.alloc(&r->h2, (B + 31) / 32 * 32 * kHid));
    r->ief_hid_block = (size_t)2 * (w->ief_iters > 0 ? w->ief_iters : 1) * kIefPhaseFloats;
    TRY(r->pool.alloc(&r->ief_hid, (size_t)((B + 15) / 16) * r->ief_hid_block, true));
#undef TRY
    *out = r;
    return SPSP_OK;
}

void spsp_regressor_free(spsp_regressor* r) { delete r; }

int spsp_proxy_pack(spsp_ctx* ctx, const float* proxy_nchw, int B, int precision, void* native_out, void* stream) {
    SPSP_REQUIRE(ctx && proxy_nchw && native_out && B >= 0, "spsp_proxy_pack: bad arguments");
    SPSP_REQUIRE(precision == 0 || precision == 1, "spsp_proxy_pack: precision %d (0 = bf16, 1 = fp16)", precision);
    if (B == 0) return SPSP_OK;
    dim3 grid((kWWp + 63) / 64, kWHp, B);
    ProfScope ps(ctx, kKProxyPack, as_stream(stream));
    if (precision) proxy_pack_kernel<true><<<grid, 64, 0, as_stream(stream)>>>(proxy_nchw, static_cast<uint16_t*>(native_out));
    else proxy_pack_kernel<false><<<grid, 64, 0, as_stream(stream)>>>(proxy_nchw, static_cast<uint16_t*>(native_out));
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

// The backbone alone: proxy crops -> pooled features, rows [feat_row, feat_row + B) of the handle's feature buffer.
int spsp_regressor_backbone(spsp_ctx* ctx, spsp_regressor* r, const void* proxy, int proxy_format, int B, int feat_row, void* stream) {
    SPSP_REQUIRE(ctx && r, "spsp_regressor_backbone: NULL handle");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));      // a process may hold contexts on several GPUs
    SPSP_TRY(spsp_regressor_status(r));             // a device-side failure of an earlier forward surfaces on the next call
    if (B == 0) return SPSP_OK;
    SPSP_REQUIRE(proxy, "spsp_regressor_backbone: NULL argument");
    SPSP_REQUIRE(B >= 0 && feat_row >= 0 && feat_row + B <= r->max_batch, "spsp_regressor_backbone: rows %d..%d exceed max_batch=%d", feat_row, feat_row + B, r->max_batch);
    SPSP_REQUIRE(proxy_format == SPSP_PROXY_NCHW_F32 || proxy_format == SPSP_PROXY_NATIVE,
                 "spsp_regressor_backbone: unknown proxy_format %d", proxy_format);
    hipStream_t st = as_stream(stream);
    int rc;
#define TRY(x) do { rc = (x); if (rc != SPSP_OK) return rc; } while (0)
    const uint16_t* x0 = static_cast<const uint16_t*>(proxy);
    if (proxy_format == SPSP_PROXY_NCHW_F32) {
        TRY(spsp_proxy_pack(ctx, static_cast<const float*>(proxy), B, r->f16, r->x0, stream));
        x0 = r->x0;
    }
    // the tile-scheduler counters of this forward's persistent launches and the IEF's arrival counters are zeroed by the stem
    // kernel's first workgroup (a memset node here cost ~5 us of every forward: 4 us + a launch boundary)
    r->next_counter = 0;
    // stem: conv1 + BN + ReLU + max-pool in ONE kernel (conv_stem.hip); the 128 x 128 x 64 map is never materialised
    {
        ProfScope ps(ctx, kKStem, st, true);
        // ... and the polling IEF's hand-off buffers of the blocks of 16 players this batch's rows touch, every workgroup a share
        const int blk0 = feat_row / 16, blk1 = (feat_row + B + 15) / 16;
        TRY(launch_conv_stem(x0, r->w[0], r->bias[0], r->p1, B, r->f16, ctx->num_cus, r->counters, kCounterWords, st, r->stem_band, r->stem32,
                             r->ief_hid + (size_t)blk0 * r->ief_hid_block, (int)((size_t)(blk1 - blk0) * r->ief_hid_block / 4)));
    }
    r->stage_ptr[0] = nullptr;
    r->stage_ptr[1] = r->p1;
    // residual stages
    const uint16_t* x = r->p1;
    int xs = 64;           // spatial size of x
    int ci = 1;
    for (int L = 0; L < 4; ++L) {
        const int S = kLayerS[L];
        uint16_t *ba = r->l[L][0], *bb = r->l[L][1], *bc = r->l[L][2];
        if (L == 0 && r->use_fb) {
            // one launch per BasicBlock: out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x), the intermediate map stays in LDS
            const int wgs = r->wres_cu_div >= 32 ? r->wres_cu_div : ctx->num_cus / r->wres_cu_div;
            { ProfScope ps(ctx, kKConvFb, st, true);
              TRY(launch_conv_fb(x, r->wfb[0], r->bias[ci], r->bias[ci + 1], bb, B, r->f16, wgs, st)); }
            { ProfScope ps(ctx, kKConvFb, st, true);
              TRY(launch_conv_fb(bb, r->wfb[1], r->bias[ci + 2], r->bias[ci + 3], bc, B, r->f16, wgs, st)); }
            ci += 4;
            x = bc;
        } else if (L == 0) {
            TRY(run_conv(ctx, r, ci, x, xs, 1, ba, nullptr, 1, B, nullptr, st, L));         // t   = relu(bn1(conv1(x)))
            TRY(run_conv(ctx, r, ci + 1, ba, S, 1, bb, x, 1, B, nullptr, st, L));           // out = relu(bn2(conv2(t)) + x)
            ci += 2;
            TRY(run_conv(ctx, r, ci, bb, S, 1, ba, nullptr, 1, B, nullptr, st, L));
            TRY(run_conv(ctx, r, ci + 1, ba, S, 1, bc, bb, 1, B, nullptr, st, L));
            ci += 2;
            x = bc;
        } else {
            // layer4's tiles are PAIRS of 8 x 8 images (conv_s2 / conv_patch): an odd batch would fall back to the implicit-GEMM
            // ring kernel for all four launches (99 instead of 52 us at 21 crops).  It runs on B + 1 crops instead: the extra crop
            // is whatever the buffers hold behind the batch (zero-filled at load, finite ever after), every crop's results depend
            // on that crop alone, and its map / feature row -- room for them is allocated -- are never read.
            const int Bl = (L == 3 && (B & 1)) ? B + 1 : B;
            // t = relu(bn1(conv1(x))) (3x3 stride 2) and identity = downsample(x) (1x1 stride 2, same input pixels) in ONE launch
            TRY(run_conv(ctx, r, ci, x, xs, 1, ba, nullptr, 1, Bl, nullptr, st, L, ci + 2, bb));
            TRY(run_conv(ctx, r, ci + 1, ba, S, 1, bc, bb, 1, Bl, nullptr, st, L));         // out -> bc
            ci += 3;
            TRY(run_conv(ctx, r, ci, bc, S, 1, ba, nullptr, 1, Bl, nullptr, st, L));        // t
            float* pool = (L == 3) ? r->feats + (size_t)feat_row * kFeat : nullptr;
            TRY(run_conv(ctx, r, ci + 1, ba, S, 1, bb, bc, 1, Bl, pool, st, L));            // out -> bb (+ avg-pool)
            ci += 2;
            x = bb;
        }
        xs = S;
        r->stage_ptr[2 + L] = x;
    }
    r->last_B = B;
    r->ief_counters_fresh = true;
    SPSP_HIP_CHECK(hipGetLastError());
#undef TRY
    return SPSP_OK;
}

// The IEF alone (ief_module.py:48-64) on rows [0, B) of the handle's feature buffer:
// params = init; 3x { params += fc3(relu(fc2(relu(fc1([feat | params]))))) } -- one launch
int spsp_regressor_ief(spsp_ctx* ctx, spsp_regressor* r, int B, float* cam, float* pose6d, float* betas, float* feats, void* stream) {
    SPSP_REQUIRE(ctx && r, "spsp_regressor_ief: NULL handle");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    if (B == 0) return SPSP_OK;
    SPSP_REQUIRE(cam && pose6d && betas, "spsp_regressor_ief: NULL argument");
    SPSP_REQUIRE(B >= 0 && B <= r->max_batch, "spsp_regressor_ief: B=%d exceeds max_batch=%d", B, r->max_batch);
    hipStream_t st = as_stream(stream);
    if (r->ief_iters > 0) {
        const int ppb = r->ief_block;               // players per block of 64 workgroups: 16 (two independent halves per 32 players) or 32
        const int npb = (B + ppb - 1) / ppb;
        IefArgs ia;
        ia.feats = r->feats; ia.w1f = r->w1f; ia.w13 = r->w13; ia.w2 = r->w2; ia.w3 = r->w3;
        ia.cvec = r->cvec; ia.b2 = r->b2; ia.p0 = r->p0;
        ia.h1 = r->h1; ia.s = r->h2; ia.hid = r->ief_hid; ia.cam = cam; ia.pose = pose6d; ia.betas = betas;
        ia.counters = r->counters + 32 * 32;     // lines 32..39 of the per-forward counter block (zeroed by the backbone call)
        // ... or here, when no backbone call has run since the last IEF on this handle: with used-up counters every hand-off
        // of the launch would pass at once and the kernel would read hidden vectors that are not written yet
        if (!r->ief_counters_fresh) {
            SPSP_HIP_CHECK(hipMemsetAsync(ia.counters, 0, kIefCounterLines * 32 * sizeof(unsigned int), st));
            SPSP_HIP_CHECK(hipMemsetAsync(r->ief_hid, 0, (size_t)((B + 15) / 16) * r->ief_hid_block * sizeof(float), st));
        }
        r->ief_counters_fresh = false;
        ia.B = B; ia.iters = r->ief_iters;
        ia.err = r->err_dev;
        ia.spin_limit = 1u << 22;
        int ief_wgs = kIefG;
        if (r->debug_ief_stall) {       // test hook: one workgroup short of what the hand-offs wait for, and a short fuse
            r->debug_ief_stall = 0;
            ief_wgs = kIefG - 1;
            ia.spin_limit = 1u << 9;
        }
        SPSP_TRY(ensure_dynamic_lds<&ief_fused_kernel<32>>(kIefLds));
        SPSP_TRY(ensure_dynamic_lds<&ief_fused_kernel<16>>(kIefLds));
        SPSP_TRY(ensure_dynamic_lds<&ief_poll_kernel<16>>(kIefLds));
        SPSP_TRY(ensure_dynamic_lds<&ief_poll_kernel<32>>(kIefLds));
        ProfScope ps(ctx, kKIefFc, st, true);
        if (r->ief_poll == 2) launch_kernel(&ief_poll_kernel<32>, dim3(ief_wgs, (B + 15) / 16), dim3(256), kIefLds, st, ia);
        else if (r->ief_poll) launch_kernel(&ief_poll_kernel<16>, dim3(ief_wgs, (B + 15) / 16), dim3(128), kIefLds, st, ia);
        else if (ppb == 16) launch_kernel(&ief_fused_kernel<16>, dim3(ief_wgs, npb), dim3(128), kIefLds, st, ia);
        else launch_kernel(&ief_fused_kernel<32>, dim3(ief_wgs, npb), dim3(256), kIefLds, st, ia);
    } else {
        ProfScope ps(ctx, kKIefMisc, st);
        ief_init_only_kernel<<<(B * kNPar + 255) / 256, 256, 0, st>>>(r->init_params, B, cam, pose6d, betas);
    }
    SPSP_HIP_CHECK(hipGetLastError());
    if (feats) SPSP_HIP_CHECK(hipMemcpyAsync(feats, r->feats, (size_t)B * kFeat * sizeof(float), hipMemcpyDeviceToDevice, st));
    return SPSP_OK;
}

int spsp_regressor_forward(spsp_ctx* ctx, spsp_regressor* r, const void* proxy, int proxy_format, int B, float* cam,
                           float* pose6d, float* betas, float* feats, void* stream) {
    SPSP_REQUIRE(ctx && r, "spsp_regressor_forward: NULL handle");
    if (B == 0) { SPSP_HIP_CHECK(hipSetDevice(ctx->device)); SPSP_TRY(spsp_regressor_status(r)); return SPSP_OK; }
    SPSP_REQUIRE(proxy && cam && pose6d && betas, "spsp_regressor_forward: NULL argument");
    SPSP_REQUIRE(B >= 0 && B <= r->max_batch, "spsp_regressor_forward: B=%d exceeds max_batch=%d", B, r->max_batch);
    SPSP_TRY(spsp_regressor_backbone(ctx, r, proxy, proxy_format, B, 0, stream));
    return spsp_regressor_ief(ctx, r, B, cam, pose6d, betas, feats, stream);
}

int spsp_regressor_status(spsp_regressor* r) {
    SPSP_REQUIRE(r, "spsp_regressor_status: NULL handle");
    if (!r->err_host) return SPSP_OK;
    const unsigned int e = __atomic_exchange_n(r->err_host, 0u, __ATOMIC_ACQ_REL);
    if (e == 0) return SPSP_OK;
    set_error("spsp_regressor: a grid-wide hand-off of the fused IEF kernel timed out in an earlier forward on this handle (status %u): "
              "its cam / pose6d / betas are NaN.  More concurrent forwards than the device can hold IEF units for (%d workgroups)?",
              e, r->ief_capacity);
    return SPSP_EHIP;
}

int spsp_regressor_set_plan(spsp_regressor* r, int plan) {
    SPSP_REQUIRE(r, "spsp_regressor_set_plan: NULL handle");
    SPSP_REQUIRE(plan == SPSP_PLAN_THROUGHPUT || plan == SPSP_PLAN_LATENCY, "spsp_regressor_set_plan: unknown plan %d", plan);
    r->patch_nch[3] = plan == SPSP_PLAN_LATENCY ? 32 : 64;
    r->p4_layers = plan == SPSP_PLAN_LATENCY ? 2 : 6;
    r->wres_cu_div = plan == SPSP_PLAN_LATENCY ? 1 : 4;
    r->stem_band = plan == SPSP_PLAN_LATENCY ? 0 : 32;      // (bands of 32 and of 64 rows tie or 32 wins by 1-1.5 %: profiles/r04_ab_narrow.txt)
    r->s2rs_cu_div = plan == SPSP_PLAN_LATENCY ? 1 : 2;      // layer2.0: four tiles per resident weight set (128 workgroups at B = 32)
    r->rs_wgs = plan == SPSP_PLAN_LATENCY ? 0 : 64;          // layer2's 3x3 / 1 convs on conv_rs.hip: eight tiles per resident weight set
    return SPSP_OK;
}

int spsp_regressor_debug(spsp_regressor* r, int what, int value) {
    SPSP_REQUIRE(r, "spsp_regressor_debug: NULL handle");
    switch (what) {
        case SPSP_DEBUG_IEF_STALL: r->debug_ief_stall = value != 0; return SPSP_OK;
        case SPSP_DEBUG_IEF_BLOCK:
            SPSP_REQUIRE(value == 16 || value == 32, "spsp_regressor_debug: IEF blocks of %d players (16 or 32)", value);
            r->ief_block = value;
            return SPSP_OK;
        case SPSP_DEBUG_IEF_POLL: if (value < 0 || value > 2) return SPSP_EINVAL; r->ief_poll = value; return SPSP_OK;
        case SPSP_DEBUG_PLAN_RS:
            SPSP_REQUIRE(value == 0 || (value >= 2 && value <= 512 && value % 2 == 0), "spsp_regressor_debug: conv_rs on %d workgroups (0: off; even)", value);
            r->rs_wgs = value;
            return SPSP_OK;
        case SPSP_DEBUG_PLAN_FB: r->use_fb = value != 0; return SPSP_OK;
        case SPSP_DEBUG_PLAN_STEM32: if (value < 0 || value > 2) return SPSP_EINVAL; r->stem32 = value; return SPSP_OK;
        case SPSP_DEBUG_PLAN_S2Q: r->use_s2q = value != 0; return SPSP_OK;
        case SPSP_DEBUG_PLAN_S2: r->use_s2 = value != 0; return SPSP_OK;
        case SPSP_DEBUG_PLAN_S2RS: r->use_s2rs = value != 0; return SPSP_OK;
        case SPSP_DEBUG_PLAN_P4: r->p4_layers = value & 6; return SPSP_OK;      // layer2 (bit 1), layer3 (bit 2)
        case SPSP_DEBUG_PLAN_S2RS_CUS:
            SPSP_REQUIRE(value == 1 || value == 2 || value == 4 || value == 8, "spsp_regressor_debug: conv_s2rs on 1 / %d of the CUs", value);
            r->s2rs_cu_div = value;
            return SPSP_OK;
        case SPSP_DEBUG_PLAN_P4_CUS:
            SPSP_REQUIRE(value == 1 || value == 2 || value == 4 || value == 8, "spsp_regressor_debug: conv_p4 on 1 / %d of the CUs", value);
            r->p4_cu_div = value;
            return SPSP_OK;
        case SPSP_DEBUG_PLAN_WRES_CUS:
            SPSP_REQUIRE(value == 1 || value == 2 || value == 4 || value == 8 || value == 16 || (value >= 32 && value <= 256 && value % 8 == 0),
                         "spsp_regressor_debug: layer1 on 1 / %d of the CUs (or 32 .. 256 workgroups, a multiple of 8)", value);
            r->wres_cu_div = value;
            return SPSP_OK;
        case SPSP_DEBUG_PLAN_STEM_BAND:
            // (bands stop growing where fewer than a quarter of the CUs would get a workgroup -- launch_conv_stem: 64 rows at B = 32)
            SPSP_REQUIRE(value == 0 || value == 16 || value == 32 || value == 64, "spsp_regressor_debug: stem bands of %d rows", value);
            r->stem_band = value;
            return SPSP_OK;
        case SPSP_DEBUG_PLAN_LAYER4_CHANNELS:
            SPSP_REQUIRE(value == 32 || value == 64, "spsp_regressor_debug: layer4 blocks of %d channels", value);
            r->patch_nch[3] = value;
            return SPSP_OK;
        case SPSP_DEBUG_PLAN_PATCH: {       // value = layer * 10000 + tile_pixels * 10 + ring_groups * 2 + two_patch_buffers
            const int L = value / 10000, tile = (value % 10000) / 10, ring = (value % 10) / 2, db = value & 1;
            SPSP_REQUIRE(L >= 1 && L <= 3 && (tile == 128 || tile == 256) && (ring == 2 || ring == 3), "spsp_regressor_debug: patch plan %d", value);
            r->patch_tile[L] = tile; r->patch_ring[L] = ring; r->patch_db[L] = db != 0;
            return SPSP_OK;
        }
        case SPSP_DEBUG_PLAN_PATCH_PRODUCER: {      // value = layer * 10 + (0 | 1)
            const int L = value / 10;
            SPSP_REQUIRE(L >= 1 && L <= 3 && value % 10 <= 1, "spsp_regressor_debug: producer plan %d", value);
            r->patch_prod[L] = value % 10 != 0;
            return SPSP_OK;
        }
        default: break;
    }
    spsp::set_error("spsp_regressor_debug: unknown hook %d", what);
    return SPSP_EINVAL;
}

int spsp_regressor_read_stage(spsp_ctx* ctx, spsp_regressor* r, int stage, int B, float* out_nchw, void* stream) {
    SPSP_REQUIRE(ctx && r && out_nchw, "spsp_regressor_read_stage: NULL argument");
    SPSP_REQUIRE(stage >= 0 && stage < 6, "spsp_regressor_read_stage: stage %d out of range", stage);
    SPSP_REQUIRE(stage != 0, "spsp_regressor_read_stage: stage 0 (conv1 + ReLU) is never materialised -- the stem kernel pools in registers; read stage 1");
    SPSP_REQUIRE(B > 0 && B <= r->last_B, "spsp_regressor_read_stage: B=%d but the last forward ran %d crops", B, r->last_B);
    const int S = stage == 0 ? 128 : kLayerS[stage < 2 ? 0 : stage - 2];
    const int C = stage < 2 ? 64 : kLayerC[stage - 2];
    const size_t total = (size_t)B * C * S * S;
    if (r->f16) read_stage_kernel<true><<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(r->stage_ptr[stage], out_nchw, B, S, S, C, 1);
    else read_stage_kernel<false><<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(r->stage_ptr[stage], out_nchw, B, S, S, C, 1);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // extern "C"
