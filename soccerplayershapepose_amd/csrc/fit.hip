// GPU driver of the reprojection fit: one 64-lane wavefront per player, all iterations in one launch,
// state in LDS (26 KB), Adam moments in registers.  The arithmetic lives in fit_core.hpp (shared with the
// host emulation used by the CPU tests).  Latency-bound by design (~27 single-wave barriers / iteration);
// players are independent, so a batch fills one CU per player.
#include <cmath>

#include "smpl.hpp"
#include "fit_core.hpp"

namespace {
using namespace spsp;

__global__ __launch_bounds__(64) void fit_kernel(FitParams P, const float* __restrict__ fitc,
                                                 const int* __restrict__ parents, const int* __restrict__ level_joints,
                                                 const int* __restrict__ level_start, const int* __restrict__ child_start,
                                                 const int* __restrict__ child_list, float* __restrict__ R,
                                                 float* __restrict__ betas, float* __restrict__ cam,
                                                 const float* __restrict__ joints2d, float* __restrict__ loss_hist,
                                                 float* __restrict__ l2e_hist) {
    __shared__ FitShared s;
    FitLane r;
    const int lane = threadIdx.x, b = blockIdx.x;
    float* Rb = R + (size_t)b * 216;
    float* bb = betas + (size_t)b * 10;
    float* cb = cam + (size_t)b * 3;
    fit_init(s, r, lane, fitc, parents, level_joints, level_start, child_start, child_list, Rb, bb, cb,
             joints2d + (size_t)b * 34);
    __syncthreads();
    float* lh = loss_hist ? loss_hist + (size_t)b * P.iters : nullptr;
    float* eh = l2e_hist ? l2e_hist + (size_t)b * P.iters : nullptr;
    double b1t = 1.0, b2t = 1.0;
    const int L = P.num_levels;
    for (int it = 0; it < P.iters; ++it) {
        b1t *= P.beta1;
        b2t *= P.beta2;
        const float step_size = (float)(P.lr / (1.0 - b1t));
        const float bc2_sqrt = (float)sqrt(1.0 - b2t);
        fit_p1(s, lane);                       __syncthreads();
        fit_p2(s, lane);                       __syncthreads();
        for (int l = 1; l < L; ++l) { fit_chain_level(s, lane, l); __syncthreads(); }
        fit_p4(s, lane);                       __syncthreads();
        fit_p5(s, lane);                       __syncthreads();
        fit_p6(s, lane, P);                    __syncthreads();
        fit_p7(s, lane);                       __syncthreads();
        fit_p8(s, lane, P, it, lh, eh);        __syncthreads();
        for (int l = L - 2; l >= 0; --l) { fit_reverse_level(s, lane, l); __syncthreads(); }
        fit_p10(s, lane);                      __syncthreads();
        fit_p11(s, lane);                      __syncthreads();
        fit_p12(s, r, lane, P, step_size, bc2_sqrt);
        __syncthreads();
    }
    const float* src = (P.select_mode == SPSP_FIT_SELECT_BEST_L2E) ? s.best : s.prm;
    for (int p = lane; p < kNumPrm; p += 64) {
        const float v = src[p];
        if (p < kPrmR) cb[p] = v;
        else if (p < kPrmBeta) Rb[p - kPrmR] = v;
        else bb[p - kPrmBeta] = v;
    }
}

}  // namespace

extern "C" {

void spsp_fit_config_default(spsp_fit_config* c, int variant) {
    if (!c) return;
    c->iters = 50;
    c->beta1 = 0.9;
    c->beta2 = 0.999;
    c->eps = 1e-8;
    c->img_wh = 512.f;
    c->loss_wh = 256.f;
    c->loss_weight = 1.0;
    if (variant == 0) {        // optimize_camera, PR/PyTorch3DTest.py:3074-3080
        c->lr = 0.01;
        c->param_mask = SPSP_FIT_MASK_CAMERA;
        c->select_mode = SPSP_FIT_SELECT_LAST;
    } else {                   // single_view_optimization, PR/player_recon.py:1175-1200, S/global_var.py:71
        c->lr = 1e-3;
        c->param_mask = SPSP_FIT_MASK_POSE_SHAPE;
        c->select_mode = SPSP_FIT_SELECT_LAST;
    }
}

int spsp_fit(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_config* cfg, float* R, float* betas, float* cam,
             const float* joints2d, int B, float* loss_hist, float* l2e_hist, void* stream) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && s && cfg, "spsp_fit: NULL handle");
    SPSP_REQUIRE(B >= 0, "spsp_fit: negative batch");
    if (B == 0) return SPSP_OK;
    SPSP_REQUIRE(R && betas && cam && joints2d, "spsp_fit: NULL argument");
    SPSP_REQUIRE(cfg->iters >= 0 && cfg->iters <= 100000, "spsp_fit: iters=%d out of range", cfg->iters);
    SPSP_REQUIRE(cfg->loss_wh > 0.f && cfg->img_wh > 0.f, "spsp_fit: img_wh / loss_wh must be positive");
    SPSP_REQUIRE(cfg->select_mode == SPSP_FIT_SELECT_LAST || cfg->select_mode == SPSP_FIT_SELECT_BEST_L2E,
                 "spsp_fit: unknown select_mode %d", cfg->select_mode);
    if (B == 0) return SPSP_OK;
    FitParams P;
    P.iters = cfg->iters;
    P.lr = cfg->lr; P.beta1 = cfg->beta1; P.beta2 = cfg->beta2; P.eps = cfg->eps;
    P.param_mask = cfg->param_mask;
    P.img_wh = cfg->img_wh;
    P.loss_wh = cfg->loss_wh;
    // log_var = -np.log(w0 + 1e-6) stored as fp32; the optimiser never touches it in the joints-only fits
    // (multi_task_loss.py:37-38, player_recon.py:1197)
    P.log_var = (float)(-std::log(cfg->loss_weight + 1e-6));
    P.weight = std::exp(-P.log_var);
    P.select_mode = cfg->select_mode;
    P.num_levels = s->num_levels;
    ProfScope ps(ctx, kKFit, as_stream(stream));
    fit_kernel<<<B, 64, 0, as_stream(stream)>>>(P, s->fitc, s->parents, s->level_joints, s->d_level_start,
                                                s->child_start, s->child_list, R, betas, cam, joints2d, loss_hist,
                                                l2e_hist);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // extern "C"
