// The fit with the silhouette IN the loss: `single_view_optimization_test` (PR/player_recon.py:1429-1557), SURVEY 8f-4.
//
//   parameters  {global_orient, body_pose idx {0-5, 8-20}, pred_cam_wp, betas} + the criterion's two log-variances (:1453)
//   loss        mse(joints2D) exp(-lv_j) + lv_j  -  S exp(-lv_s) + lv_s,   S = sum_px [p ln(t + 1e-6) + (1 - p) ln(1 - t + 1e-6)]
//               (PR/losses/multi_task_loss.py:97-113, 132-144; init weights 1.0 / 1e6, PR/player_recon.py:851)
//   optimiser   torch.optim.Adam(lr 1e-3), 50 epochs; the camera translation handed to the renderer is recomputed from cam_wp
//               after every step (:1539) -- so the silhouette term reaches the camera from the SECOND epoch on (the first epoch's
//               translation was computed under no_grad, :1414-1427)
//
// One epoch = a short chain of launches, every tensor staying on the device (the host loop only advances Adam's bias terms):
//   SMPL forward (vertices + posed rest vertices)  ->  translation of cam_wp  ->  ownership raster, loss sum, neural_renderer's
//   backward pass to dL/dvertices and dL/dt (silhouette_grad.hip)  ->  per-joint seeds and basis . dvp (smpl_vjp.hip)  ->
//   fit_sil_step_kernel: the reprojection fit's forward and backward phases (fit_core.hpp) with the dense seeds injected
//   (fit_dense.hpp), the log-variance gradients, and the Adam step of all 231 parameters (moments in the handle's workspace).
#include <cmath>

#include "smpl.hpp"
#include "fit_core.hpp"
#include "fit_dense.hpp"
#include "silhouette.hpp"

namespace {
using namespace spsp;

// misc block per player (16 floats): 0..2 translation | 3..5 dL/dtranslation | 6 S | 7 exp(-lv_s)
__global__ void fit_sil_prepare_kernel(const float* __restrict__ cam, const float* __restrict__ logvars, int B, float focal, float res,
                                       float* __restrict__ misc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float* m = misc + (size_t)b * 16;
    m[0] = cam[b * 3 + 1];
    m[1] = cam[b * 3 + 2];
    m[2] = 2.f * focal / (res * cam[b * 3] + 1e-9f);              // cam_utils.py:28-34
    m[7] = expf(-logvars[b * 2 + 1]);
}

struct SilStep {
    float* R; float* betas; float* cam; float* logvars;           // in / out
    const float* joints2d;
    const float* seeds; const float* gcoef; const float* misc;
    float* adam;                                                  // [B][2][232]: m, v of parameter p (229 = lv_j, 230 = lv_s)
    float* loss_hist; float* l2e_hist; float* sil_hist; float* param_hist; float* grad_hist;
    int it, iters, cam_from_silhouette;
    float focal, res;
    float w1, b2, w2, eps, step_size, bc2_sqrt;
};

__global__ __launch_bounds__(spsp::kFitLanes) void fit_sil_step_kernel(FitParams P, const float* __restrict__ fitc,
                                                                       const int* __restrict__ parents, const int* __restrict__ level_joints,
                                                                       const int* __restrict__ level_start, const int* __restrict__ child_start,
                                                                       const int* __restrict__ child_list, const SilStep a) {
    __shared__ FitShared s;
    FitLane r;
    const int lane = threadIdx.x, b = blockIdx.x, L = P.num_levels;
    float* Rb = a.R + (size_t)b * 216;
    float* bb = a.betas + (size_t)b * 10;
    float* cb = a.cam + (size_t)b * 3;
    const float lv_j = a.logvars[b * 2], lv_s = a.logvars[b * 2 + 1];
    P.log_var = lv_j;
    P.weight = expf(-lv_j);
    fit_init(s, r, lane, fitc, parents, level_joints, level_start, child_start, child_list, Rb, bb, cb, a.joints2d + (size_t)b * 34, P.param_mask);
    __syncthreads();
    fit_p1(s, r, lane); __syncthreads();
    fit_p2(s, lane); __syncthreads();
    for (int l = 1; l < L; ++l) { fit_chain_level(s, r, r.lvl[l]); __syncthreads(); }
    fit_p4(s, r, lane); __syncthreads();
    fit_p5(s, lane); __syncthreads();
    fit_p6(s, r, lane, P); __syncthreads();
    fit_p7(s, r, lane);
    fit_dense_inject(s, lane, a.seeds + (size_t)b * 288);
    __syncthreads();
    fit_p8(s, lane, P, 0, nullptr, nullptr); __syncthreads();
    for (int l = L - 2; l >= 0; --l) { fit_reverse_level(s, r, r.lvl[l]); __syncthreads(); }
    fit_p10(s, r, lane); __syncthreads();
    fit_p11(s, r, lane); __syncthreads();
    const float* misc = a.misc + (size_t)b * 16;
    const float S = misc[6], w_s = misc[7];
    const float mse = s.red[0] / 34.0f;
    if (lane == 255) {
        if (a.loss_hist) a.loss_hist[(size_t)b * a.iters + a.it] = (mse * P.weight + lv_j) + (-S * w_s + lv_s);
        if (a.l2e_hist) a.l2e_hist[(size_t)b * a.iters + a.it] = s.red[1] / 17.0f;
        if (a.sil_hist) a.sil_hist[(size_t)b * a.iters + a.it] = -S;
    }
    if (lane < kNumPrm + 2) {
        const int p = lane;
        float cur, g;
        if (p < kNumPrm) {
            cur = s.prm[p];
            if (p < kPrmR) {
                g = s.red[2 + p];
                if (a.cam_from_silhouette) {                        // t = [tx, ty, 2 f / (res s + 1e-9)]
                    const float den = a.res * cur + 1e-9f;
                    if (p == 0) g += misc[5] * (-2.f * a.focal * a.res / (den * den));
                    else g += misc[2 + p];
                }
            } else if (p < kPrmBeta) {
                g = s.grad[p] + (p >= kPrmR + 9 ? a.gcoef[(size_t)b * kNCoef + kNB + (p - kPrmR - 9)] : 0.f);
            } else {
                g = fit_beta_grad(s, p - kPrmBeta) + a.gcoef[(size_t)b * kNCoef + (p - kPrmBeta)];
            }
        } else if (p == kNumPrm) {                                  // d/d lv_j [mse exp(-lv_j) + lv_j]
            cur = lv_j;
            g = -mse * P.weight + 1.f;
        } else {                                                    // d/d lv_s [-S exp(-lv_s) + lv_s]
            cur = lv_s;
            g = S * w_s + 1.f;
        }
        if (a.param_hist) a.param_hist[((size_t)b * a.iters + a.it) * (kNumPrm + 2) + p] = cur;
        if (a.grad_hist) a.grad_hist[((size_t)b * a.iters + a.it) * (kNumPrm + 2) + p] = g;
        const bool trainable = p >= kNumPrm || r.trainable;
        if (trainable) {
            float* mv = a.adam + (size_t)b * 2 * 232;
            float m = mv[p], v = mv[232 + p];
            m = m + a.w1 * (g - m);
            v = v * a.b2 + (a.w2 * g) * g;
            mv[p] = m; mv[232 + p] = v;
            const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
            const float nv = cur + ((-a.step_size) * m) / denom;
            if (p < kPrmR) cb[p] = nv;
            else if (p < kPrmBeta) Rb[p - kPrmR] = nv;
            else if (p < kNumPrm) bb[p - kPrmBeta] = nv;
            else a.logvars[b * 2 + (p - kNumPrm)] = nv;
        }
    }
}

}  // namespace

extern "C" {

void spsp_fit_silhouette_config_default(spsp_fit_silhouette_config* c) {
    if (!c) return;
    c->iters = 50;                      // PR/player_recon.py:1457
    c->lr = 1e-3;                       // S/global_var.py:71
    c->beta1 = 0.9; c->beta2 = 0.999; c->eps = 1e-8;
    c->joints2d_weight = 1.0;           // PR/player_recon.py:851 init_loss_weights
    c->silhouette_weight = 1e6;
    c->param_mask = SPSP_FIT_MASK_POSE_SHAPE;
    c->img_wh = 512.f; c->loss_wh = 256.f;
    c->focal = 5000.f;
}

int spsp_fit_silhouette(spsp_ctx* ctx, spsp_smpl* s, spsp_rasterizer* rast, const spsp_fit_silhouette_config* cfg, float* R, float* betas,
                        float* cam, float* log_vars, const float* joints2d, const uint8_t* target, int target_shared, int B,
                        float* loss_hist, float* l2e_hist, float* sil_hist, float* param_hist, float* grad_hist, void* stream) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && s && rast && cfg, "spsp_fit_silhouette: NULL handle");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    SPSP_REQUIRE(B >= 0 && B <= s->max_batch && B <= rast->max_batch, "spsp_fit_silhouette: B=%d exceeds a handle's max_batch", B);
    if (B == 0) return SPSP_OK;
    SPSP_REQUIRE(R && betas && cam && log_vars && joints2d && target, "spsp_fit_silhouette: NULL argument");
    SPSP_REQUIRE(cfg->iters >= 0 && cfg->iters <= 100000, "spsp_fit_silhouette: iters=%d out of range", cfg->iters);
    SPSP_REQUIRE(cfg->loss_wh > 0.f && cfg->img_wh > 0.f && (int)cfg->img_wh == rast->wh, "spsp_fit_silhouette: img_wh must equal the rasteriser's image size");
    SPSP_TRY(smpl_vjp_workspace(s));
    hipStream_t st = as_stream(stream);
    FitParams P;
    P.iters = cfg->iters;
    P.lr = cfg->lr; P.beta1 = cfg->beta1; P.beta2 = cfg->beta2; P.eps = cfg->eps;
    P.param_mask = cfg->param_mask;
    P.img_wh = cfg->img_wh; P.loss_wh = cfg->loss_wh;
    P.log_var = 0.f; P.weight = 1.f;                 // per player, from log_vars, inside the kernel
    P.select_mode = SPSP_FIT_SELECT_LAST;
    P.num_levels = s->num_levels;
    SPSP_HIP_CHECK(hipMemsetAsync(s->ws_adam, 0, (size_t)B * 2 * 232 * sizeof(float), st));
    double b1t = 1.0, b2t = 1.0;
    for (int it = 0; it < cfg->iters; ++it) {
        b1t *= cfg->beta1;
        b2t *= cfg->beta2;
        SPSP_TRY(smpl_forward_impl(ctx, s, R, betas, B, s->ws_verts, nullptr, s->ws_vposed, stream));
        fit_sil_prepare_kernel<<<(B + 63) / 64, 64, 0, st>>>(cam, log_vars, B, cfg->focal, cfg->img_wh, s->ws_misc);
        // misc is [B][16]: the silhouette call reads translation / weight and writes dL/dt / S with a stride of 16 floats
        SPSP_TRY(silhouette_loss_grad_impl(ctx, rast, s->ws_verts, s->ws_misc, 16, cfg->focal, B, target, target_shared, s->ws_misc + 7, 16,
                                           s->ws_misc + 6, 16, s->ws_gverts, s->ws_misc + 3, 16, nullptr, stream));
        SPSP_TRY(smpl_vjp_seeds(ctx, s, s->ws_gverts, B, stream));
        SilStep a;
        a.R = R; a.betas = betas; a.cam = cam; a.logvars = log_vars; a.joints2d = joints2d;
        a.seeds = s->ws_seeds; a.gcoef = s->ws_gcoef; a.misc = s->ws_misc; a.adam = s->ws_adam;
        a.loss_hist = loss_hist; a.l2e_hist = l2e_hist; a.sil_hist = sil_hist; a.param_hist = param_hist; a.grad_hist = grad_hist;
        a.it = it; a.iters = cfg->iters; a.cam_from_silhouette = it > 0;
        a.focal = cfg->focal; a.res = cfg->img_wh;
        a.w1 = (float)(1.0 - cfg->beta1); a.b2 = (float)cfg->beta2; a.w2 = (float)(1.0 - cfg->beta2); a.eps = (float)cfg->eps;
        a.step_size = (float)(cfg->lr / (1.0 - b1t));
        a.bc2_sqrt = (float)std::sqrt(1.0 - b2t);
        fit_sil_step_kernel<<<B, kFitLanes, 0, st>>>(P, s->fitc, s->parents, s->level_joints, s->d_level_start, s->child_start, s->child_list, a);
        SPSP_HIP_CHECK(hipGetLastError());
    }
    return SPSP_OK;
}

}  // extern "C"
