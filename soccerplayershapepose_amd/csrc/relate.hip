// PoseRelationModule (PR/models/regressor_relate.py:9-139) for the players of one frame, fp32 VALU.
//   relate_project_kernel : K / Q / V projections of the 216-d pose feature for all 27 relation units
//                           ([N,216] x [216,648]; weights stored transposed so lanes read consecutive columns)
//   relate_attend_kernel  : one workgroup per player m: 8-d sin/cos embedding of the pairwise box geometry
//                           (__PositionalEmbedding, :34-74), geometry gate relu(WG e + b), logits
//                           log(clamp(gate, 1e-6)) + K_m . Q_n / sqrt(8), soft-max over n, unit output
//                           (sum_n softmax_mn) * V_m  -- the reference multiplies the attention row of m with m's OWN
//                           value (:133-138), kept as is -- concatenated over the units, + the input pose.
// ~0.2 MFLOP per player: latency-bound, two launches; no MFMA (contraction depth 216 on 22 rows).
#include <cmath>
#include <vector>

#include "common.hpp"

namespace {
using namespace spsp;

constexpr int kNRel = 27, kDk = 8, kDa = 216, kProj = kNRel * 3 * kDk;   // 648 projection outputs per player

__global__ __launch_bounds__(256) void relate_project_kernel(const float* __restrict__ fa, const float* __restrict__ wt,
                                                             const float* __restrict__ bias, float* __restrict__ kqv) {
    __shared__ float sF[kDa];
    const int m = blockIdx.x;
    for (int i = threadIdx.x; i < kDa; i += 256) sF[i] = fa[(size_t)m * kDa + i];
    __syncthreads();
    for (int o = threadIdx.x; o < kProj; o += 256) {
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < kDa; ++k) acc = fmaf(sF[k], wt[(size_t)k * kProj + o], acc);
        kqv[(size_t)m * kProj + o] = acc + bias[o];
    }
}

__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int kMaxN = 512;          // players per frame the embedding cache is sized for

__global__ __launch_bounds__(256) void relate_attend_kernel(const float* __restrict__ fa, const float* __restrict__ boxes,
                                                            const float* __restrict__ kqv, const float* __restrict__ wg,
                                                            const float* __restrict__ bg, int N, float* __restrict__ out) {
    __shared__ float sE[kMaxN][kDk];
    __shared__ float sS[kNRel];
    const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float bxm0 = boxes[m * 4], bym0 = boxes[m * 4 + 1], bxm1 = boxes[m * 4 + 2], bym1 = boxes[m * 4 + 3];
    const float cxm = (bxm0 + bxm1) * 0.5f, cym = (bym0 + bym1) * 0.5f, wm = (bxm1 - bxm0) + 1.f, hm = (bym1 - bym0) + 1.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float x0 = boxes[n * 4], y0 = boxes[n * 4 + 1], x1 = boxes[n * 4 + 2], y1 = boxes[n * 4 + 3];
        const float cxn = (x0 + x1) * 0.5f, cyn = (y0 + y1) * 0.5f, wn = (x1 - x0) + 1.f, hn = (y1 - y0) + 1.f;
        const float d[4] = {logf(fmaxf(fabsf((cxm - cxn) / wm), 1e-3f)), logf(fmaxf(fabsf((cym - cyn) / hm), 1e-3f)),
                            logf(wm / wn), logf(hm / hn)};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = 100.f * d[i];          // dim_g / 8 = 1 frequency: 1 / 1000^0 = 1
            sE[n][i] = sinf(a);
            sE[n][4 + i] = cosf(a);
        }
    }
    __syncthreads();
    for (int r = wave; r < kNRel; r += 4) {
        float g[kDk], km[kDk];
#pragma unroll
        for (int i = 0; i < kDk; ++i) { g[i] = wg[r * kDk + i]; km[i] = kqv[(size_t)m * kProj + (r * 3 + 0) * kDk + i]; }
        const float b = bg[r];
        // pass 1: logits (kept in registers for up to 8 players per lane), running max
        float lg[kMaxN / 64];
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < kMaxN / 64; ++t) {
            const int n = lane + 64 * t;
            lg[t] = -INFINITY;
            if (n < N) {
                float gate = b, dot = 0.f;
#pragma unroll
                for (int i = 0; i < kDk; ++i) {
                    gate = fmaf(g[i], sE[n][i], gate);
                    dot = fmaf(km[i], kqv[(size_t)n * kProj + (r * 3 + 1) * kDk + i], dot);
                }
                lg[t] = logf(fmaxf(fmaxf(gate, 0.f), 1e-6f)) + dot / 2.8284271247461903f;
                mx = fmaxf(mx, lg[t]);
            }
        }
        mx = wave_max(mx);
        float se = 0.f;
#pragma unroll
        for (int t = 0; t < kMaxN / 64; ++t) {
            lg[t] = (lane + 64 * t < N) ? expf(lg[t] - mx) : 0.f;
            se += lg[t];
        }
        se = wave_sum(se);
        float sw = 0.f;
#pragma unroll
        for (int t = 0; t < kMaxN / 64; ++t) sw += lg[t] / se;
        sw = wave_sum(sw);                      // sum_n softmax_mn: 1 up to rounding, exactly what the reference sums
        if (lane == 0) sS[r] = sw;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < kDa; c += 256) {
        const int r = c / kDk, d = c - r * kDk;
        out[(size_t)m * kDa + c] = sS[r] * kqv[(size_t)m * kProj + (r * 3 + 2) * kDk + d] + fa[(size_t)m * kDa + c];
    }
}

}  // namespace

struct spsp_relate {
    spsp::DevicePool pool;
    float *wt, *bias, *wg, *bg, *kqv;
    int max_players;
};

extern "C" {

int spsp_relate_load(spsp_ctx* ctx, const spsp_relate_weights* w, int max_players, spsp_relate** out) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && w && out, "spsp_relate_load: NULL argument");
    SPSP_REQUIRE(max_players > 0 && max_players <= kMaxN, "spsp_relate_load: max_players=%d (1..%d)", max_players, kMaxN);
    for (int r = 0; r < kNRel; ++r)
        SPSP_REQUIRE(w->unit[r].wg_weight && w->unit[r].wg_bias && w->unit[r].wk_weight && w->unit[r].wk_bias && w->unit[r].wq_weight &&
                         w->unit[r].wq_bias && w->unit[r].wv_weight && w->unit[r].wv_bias,
                     "spsp_relate_load: NULL array in relation unit %d", r);
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<float> wt((size_t)kDa * kProj), bias(kProj), wg(kNRel * kDk), bg(kNRel);
    for (int r = 0; r < kNRel; ++r) {
        const spsp_relate_unit& u = w->unit[r];
        const float* pw[3] = {u.wk_weight, u.wq_weight, u.wv_weight};
        const float* pb[3] = {u.wk_bias, u.wq_bias, u.wv_bias};
        for (int p = 0; p < 3; ++p)
            for (int d = 0; d < kDk; ++d) {
                const int o = (r * 3 + p) * kDk + d;
                bias[o] = pb[p][d];
                for (int k = 0; k < kDa; ++k) wt[(size_t)k * kProj + o] = pw[p][d * kDa + k];     // nn.Linear weight [out][in]
            }
        for (int i = 0; i < kDk; ++i) wg[r * kDk + i] = u.wg_weight[i];
        bg[r] = u.wg_bias[0];
    }
    spsp_relate* h = new spsp_relate;
    h->max_players = max_players;
    int rc;
#define TRY(x) do { rc = (x); if (rc != SPSP_OK) { delete h; return rc; } } while (0)
    TRY(h->pool.upload(&h->wt, wt.data(), wt.size()));
    TRY(h->pool.upload(&h->bias, bias.data(), bias.size()));
    TRY(h->pool.upload(&h->wg, wg.data(), wg.size()));
    TRY(h->pool.upload(&h->bg, bg.data(), bg.size()));
    TRY(h->pool.alloc(&h->kqv, (size_t)max_players * kProj));
#undef TRY
    *out = h;
    return SPSP_OK;
}

void spsp_relate_free(spsp_relate* h) { delete h; }

int spsp_relate_forward(spsp_ctx* ctx, spsp_relate* h, const float* pose_rotmats, const float* boxes, int N, float* out,
                        void* stream) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && h, "spsp_relate_forward: NULL handle");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));      // a process may hold contexts on several GPUs
    if (N == 0) return SPSP_OK;
    SPSP_REQUIRE(pose_rotmats && boxes && out, "spsp_relate_forward: NULL argument");
    SPSP_REQUIRE(N > 0 && N <= h->max_players, "spsp_relate_forward: N=%d exceeds max_players=%d", N, h->max_players);
    hipStream_t st = as_stream(stream);
    ProfScope ps(ctx, kKRelate, st);
    relate_project_kernel<<<N, 256, 0, st>>>(pose_rotmats, h->wt, h->bias, h->kqv);
    SPSP_HIP_CHECK(hipGetLastError());
    relate_attend_kernel<<<N, 256, 0, st>>>(pose_rotmats, boxes, h->kqv, h->wg, h->bg, N, out);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // extern "C"
