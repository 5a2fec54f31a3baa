// Reverse mode of the SMPL forward's VERTICES (smplx lbs, PR/models/smpl_official.py:27-41): what the silhouette term of the
// fit needs -- dL/d(vertices) [B,6890,3] -> dL/dR [B,24,3,3], dL/dbetas [B,10].
//
// With x_v = sum_k w_vk (Rw_k (vp_v - J_k) + tw_k), vp = template + S beta + P f (f = vec(R_1..23 - I)), J = Jbar + JS beta and
// h_v = dL/dx_v, everything the per-vertex part contributes collapses into a few small "seeds" (derivation: fit_core.hpp, which
// has the same structure for its 5 face vertices):
//     S0_k = sum_v w_vk h_v                 (3)      = the direct adjoint of tw_k
//     S1_k = sum_v w_vk h_v vp_v^T          (3 x 3)  -> B_k = S1_k - S0_k J_k^T, the direct adjoint of Rw_k
//     dvp_v = sum_k w_vk Rw_k^T h_v         (3 per vertex)
//     gcoef = basis . dvp                   (217)    = [S^T dvp (betas through the shape blend) | P dvp (pose feature)]
// after which the reverse sweep over the 24-joint tree, the joint regressor and the parameter gradients are exactly the phases
// of the reprojection fit (fit_core.hpp) with these seeds injected -- the VJP kernel below is the fit's backward half with no
// joints2D loss in front of it.
//
//   smpl_seed_kernel     (vertex chunk of 256, mesh): dvp per vertex; chunk partials of [S1 | S0] with a thread per (joint, entry)
//   smpl_seed_sum_kernel fixed-order sum of the 27 chunk partials
//   smpl_gcoef_kernel    (basis row, 8 meshes): the row is read once for eight meshes
//   smpl_vjp_kernel      one 256-lane workgroup per mesh: forward chain, inject, reverse sweep, parameter gradients
#include "smpl.hpp"
#include "fit_core.hpp"
#include "fit_dense.hpp"

namespace {
using namespace spsp;

constexpr int kChunk = 256, kNChunk = (kV + kChunk - 1) / kChunk;     // 27

__global__ __launch_bounds__(320) void smpl_seed_kernel(const float* __restrict__ gverts, const float* __restrict__ vposed,
                                                        const float* __restrict__ A, const float* __restrict__ wT,
                                                        float* __restrict__ dvp, float* __restrict__ seedp) {
    __shared__ float sA[288], sh[kChunk][3], svp[kChunk][3];
    const int chunk = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    if (t < 288) sA[t] = A[(size_t)b * 288 + t];
    __syncthreads();
    const int v = chunk * kChunk + t;
    if (t < kChunk) {
        float h0 = 0.f, h1 = 0.f, h2 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f;
        if (v < kV) {
            const float* g = gverts + ((size_t)b * kV + v) * 3;
            const float* q = vposed + ((size_t)b * kV + v) * 3;
            h0 = g[0]; h1 = g[1]; h2 = g[2]; p0 = q[0]; p1 = q[1]; p2 = q[2];
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            for (int k = 0; k < kNJ; ++k) {
                const float w = wT[(size_t)k * kV + v];
                const float* a = sA + k * 12;                       // A_k rows = Rw_k rows: (Rw^T h)[l] = sum_i Rw[i][l] h[i]
                d0 = fmaf(w, fmaf(a[0], h0, fmaf(a[4], h1, a[8] * h2)), d0);
                d1 = fmaf(w, fmaf(a[1], h0, fmaf(a[5], h1, a[9] * h2)), d1);
                d2 = fmaf(w, fmaf(a[2], h0, fmaf(a[6], h1, a[10] * h2)), d2);
            }
            float* o = dvp + ((size_t)b * kV + v) * 3;
            o[0] = d0; o[1] = d1; o[2] = d2;
        }
        sh[t][0] = h0; sh[t][1] = h1; sh[t][2] = h2;
        svp[t][0] = p0; svp[t][1] = p1; svp[t][2] = p2;
    }
    __syncthreads();
    if (t < 288) {                                                  // entry (k, i, c): S1_k[i][c] (c < 3) or S0_k[i] (c = 3)
        const int k = t / 12, e = t % 12, i = e >> 2, c = e & 3;
        const int n = min(kChunk, kV - chunk * kChunk);
        const float* w = wT + (size_t)k * kV + chunk * kChunk;
        float acc = 0.f;
        for (int u = 0; u < n; ++u) acc = fmaf(w[u] * sh[u][i], c < 3 ? svp[u][c] : 1.f, acc);
        seedp[((size_t)b * kNChunk + chunk) * 288 + t] = acc;
    }
}

__global__ __launch_bounds__(320) void smpl_seed_sum_kernel(const float* __restrict__ seedp, float* __restrict__ seeds) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (t >= 288) return;
    float acc = 0.f;
    for (int c = 0; c < kNChunk; ++c) acc += seedp[((size_t)b * kNChunk + c) * 288 + t];
    seeds[(size_t)b * 288 + t] = acc;
}

__global__ __launch_bounds__(256) void smpl_gcoef_kernel(const float* __restrict__ basis, const float* __restrict__ dvp, int B,
                                                         float* __restrict__ gcoef) {
    __shared__ float red[8][256];
    const int j = blockIdx.x, b0 = blockIdx.y * 8, t = threadIdx.x;
    const int nb = min(8, B - b0);
    float acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = 0.f;
    const float* row = basis + (size_t)j * kV3;
    for (int e = t; e < kV3; e += 256) {
        const float r = row[e];
#pragma unroll
        for (int p = 0; p < 8; ++p)
            if (p < nb) acc[p] = fmaf(r, dvp[(size_t)(b0 + p) * kV3 + e], acc[p]);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) red[p][t] = acc[p];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int p = 0; p < 8; ++p) red[p][t] += red[p][t + s];
        }
        __syncthreads();
    }
    if (t < nb) gcoef[(size_t)(b0 + t) * kNCoef + j] = red[t][0];
}

__global__ __launch_bounds__(spsp::kFitLanes) void smpl_vjp_kernel(const float* __restrict__ fitc, const int* __restrict__ parents,
                                                                   const int* __restrict__ level_joints, const int* __restrict__ level_start,
                                                                   const int* __restrict__ child_start, const int* __restrict__ child_list,
                                                                   int num_levels, const float* __restrict__ R, const float* __restrict__ betas,
                                                                   const float* __restrict__ seeds, const float* __restrict__ gcoef,
                                                                   float* __restrict__ grad_R, float* __restrict__ grad_betas) {
    __shared__ FitShared s;
    __shared__ float zero_cam[4], zero_j2d[34];
    FitLane r;
    const int lane = threadIdx.x, b = blockIdx.x, L = num_levels;
    if (lane < 4) zero_cam[lane] = lane == 0 ? 1.f : 0.f;
    if (lane < 34) zero_j2d[lane] = 0.f;
    __syncthreads();
    fit_init(s, r, lane, fitc, parents, level_joints, level_start, child_start, child_list, R + (size_t)b * 216, betas + (size_t)b * 10,
             zero_cam, zero_j2d, 0u);
    __syncthreads();
    fit_p1(s, r, lane); __syncthreads();
    fit_p2(s, lane); __syncthreads();
    for (int l = 1; l < L; ++l) { fit_chain_level(s, r, r.lvl[l]); __syncthreads(); }
    // no joints2D loss: its seeds (ga, h) stay zero, so P7 leaves zero adjoints for the dense seeds to add to
    fit_p7(s, r, lane); __syncthreads();
    if (lane < 15) { s.dvp[lane] = 0.f; s.adj[lane] = 0.f; }
    fit_dense_inject(s, lane, seeds + (size_t)b * 288);
    __syncthreads();
    for (int l = L - 2; l >= 0; --l) { fit_reverse_level(s, r, r.lvl[l]); __syncthreads(); }
    fit_p10(s, r, lane); __syncthreads();
    fit_p11(s, r, lane); __syncthreads();
    if (lane < 216) grad_R[(size_t)b * 216 + lane] = s.grad[kPrmR + lane] + (lane >= 9 ? gcoef[(size_t)b * kNCoef + kNB + (lane - 9)] : 0.f);
    if (lane < 10) grad_betas[(size_t)b * 10 + lane] = fit_beta_grad(s, lane) + gcoef[(size_t)b * kNCoef + lane];
}

}  // namespace

namespace spsp {

int smpl_vjp_workspace(spsp_smpl* s) {
    if (s->ws_vposed) return SPSP_OK;
    const size_t mb = s->max_batch;
    SPSP_TRY(s->pool.alloc(&s->ws_vposed, mb * kV3));
    SPSP_TRY(s->pool.alloc(&s->ws_dvp, mb * kV3));
    SPSP_TRY(s->pool.alloc(&s->ws_seedp, mb * kNChunk * 288));
    SPSP_TRY(s->pool.alloc(&s->ws_seeds, mb * 288));
    SPSP_TRY(s->pool.alloc(&s->ws_gcoef, mb * kNCoef));
    SPSP_TRY(s->pool.alloc(&s->ws_gverts, mb * kV3));
    SPSP_TRY(s->pool.alloc(&s->ws_adam, mb * 2 * 232, true));
    SPSP_TRY(s->pool.alloc(&s->ws_misc, mb * 16, true));
    return SPSP_OK;
}

int smpl_vjp_seeds(spsp_ctx* ctx, spsp_smpl* s, const float* grad_verts, int B, void* stream) {
    hipStream_t st = as_stream(stream);
    smpl_seed_kernel<<<dim3(kNChunk, B), 320, 0, st>>>(grad_verts, s->ws_vposed, s->ws_A, s->wT, s->ws_dvp, s->ws_seedp);
    smpl_seed_sum_kernel<<<B, 320, 0, st>>>(s->ws_seedp, s->ws_seeds);
    smpl_gcoef_kernel<<<dim3(kNCoef, (B + 7) / 8), 256, 0, st>>>(s->basis, s->ws_dvp, B, s->ws_gcoef);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp

extern "C" {

int spsp_smpl_vertices_vjp(spsp_ctx* ctx, spsp_smpl* s, const float* R, const float* betas, int B, const float* grad_verts, float* grad_R,
                           float* grad_betas, void* stream) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && s, "spsp_smpl_vertices_vjp: NULL handle");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    SPSP_REQUIRE(B >= 0 && B <= s->max_batch, "spsp_smpl_vertices_vjp: B=%d exceeds max_batch=%d", B, s->max_batch);
    if (B == 0) return SPSP_OK;
    SPSP_REQUIRE(R && betas && grad_verts && grad_R && grad_betas, "spsp_smpl_vertices_vjp: NULL argument");
    SPSP_TRY(smpl_vjp_workspace(s));
    SPSP_TRY(smpl_forward_impl(ctx, s, R, betas, B, s->ws_verts, nullptr, s->ws_vposed, stream));
    SPSP_TRY(smpl_vjp_seeds(ctx, s, grad_verts, B, stream));
    smpl_vjp_kernel<<<B, kFitLanes, 0, as_stream(stream)>>>(s->fitc, s->parents, s->level_joints, s->d_level_start, s->child_start, s->child_list,
                                                           s->num_levels, R, betas, s->ws_seeds, s->ws_gcoef, grad_R, grad_betas);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // extern "C"
