// HOST EMULATION of the fit kernel -- test infrastructure only (never loaded by the product).
// Runs the phases of soccerplayershapepose_amd/csrc/fit_core.hpp -- the very code the gfx950 kernel
// executes -- with `for lane in 0..63` in place of the wavefront and a phase boundary in place of the
// barrier, so the CPU test-suite can check the kernel's index math and its hand-derived gradient against
// the autograd oracle without a GPU.  Built by tests/test_fit_emulation.py with g++.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../soccerplayershapepose_amd/csrc/fit_core.hpp"
#include "../../soccerplayershapepose_amd/csrc/smpl_host.hpp"

using namespace spsp;

#define ALL_LANES(stmt) for (int lane = 0; lane < kFitLanes; ++lane) { stmt; }

extern "C" int fit_emulate(const spsp_smpl_arrays* model, int iters, double lr,
                           double beta1, double beta2, double eps, unsigned param_mask, float img_wh, float loss_wh,
                           double loss_weight, int select_mode, float* R, float* betas, float* cam,
                           const float* joints2d, int B, float* loss_hist, float* l2e_hist, float* grad_first) {
    SmplHostImage h;
    if (!smpl_prepare(*model, h, false).empty()) return -1;
    const float* fitc = h.fitc.data();
    const int *parents = h.parents, *level_joints = h.level_joints.data(), *level_start = h.level_start;
    const int *child_start = h.child_start, *child_list = h.child_list;
    const int num_levels = h.num_levels;
    FitParams P;
    P.iters = iters; P.lr = lr; P.beta1 = beta1; P.beta2 = beta2; P.eps = eps;
    P.param_mask = param_mask; P.img_wh = img_wh; P.loss_wh = loss_wh;
    P.log_var = (float)(-std::log(loss_weight + 1e-6));
    P.weight = std::exp(-P.log_var);
    P.select_mode = select_mode; P.num_levels = num_levels;
    std::vector<FitShared> sv(2);
    FitShared& s = sv[0];
    // the v2 schedule of the product kernel (fit_core.hpp "round 4", fit.hip fit_kernel2) runs beside the phases it
    // re-arranges, on a state of its own; after every iteration all 229 parameters must agree BIT FOR BIT
    FitShared& s2 = sv[1];
    std::vector<Fit2Shared> xv(1);
    Fit2Shared& x2 = xv[0];
    std::vector<float> lh2(iters > 0 ? iters : 1), eh2(iters > 0 ? iters : 1);
    for (int b = 0; b < B; ++b) {
        FitLane r[kFitLanes];
        std::vector<FitLane> r2v(kFitLanes);
        FitLane* r2 = r2v.data();
        const bool v2 = h.walk_ok;
        if (v2) {
            float* Rb2 = R + (size_t)b * 216; float* bb2 = betas + (size_t)b * 10; float* cb2 = cam + (size_t)b * 3;
            ALL_LANES(fit_init(s2, r2[lane], lane, fitc, parents, level_joints, level_start, child_start, child_list, Rb2, bb2,
                               cb2, joints2d + (size_t)b * 34, param_mask, h.walk));
            ALL_LANES(fit2_init0(x2, lane));
            ALL_LANES(fit2_init(s2, x2, r2[lane], lane, fitc, param_mask));
        }
        float* Rb = R + (size_t)b * 216; float* bb = betas + (size_t)b * 10; float* cb = cam + (size_t)b * 3;
        ALL_LANES(fit_init(s, r[lane], lane, fitc, parents, level_joints, level_start, child_start, child_list, Rb, bb,
                           cb, joints2d + (size_t)b * 34, param_mask, h.walk_ok ? h.walk : nullptr));
        float* lh = loss_hist ? loss_hist + (size_t)b * iters : nullptr;
        float* eh = l2e_hist ? l2e_hist + (size_t)b * iters : nullptr;
        double b1t = 1.0, b2t = 1.0;
        const int L = num_levels;
        for (int it = 0; it < iters; ++it) {
            b1t *= beta1; b2t *= beta2;
            const float step_size = (float)(lr / (1.0 - b1t));
            const float bc2_sqrt = (float)std::sqrt(1.0 - b2t);
            ALL_LANES(fit_p1(s, r[lane], lane));
            ALL_LANES(fit_p2(s, lane));
            // the product kernel walks the tree (fit_chain_walk / fit_reverse_walk_*); the level sweeps it replaced run beside it
            // here on a copy of the state, and the two must agree BIT FOR BIT (same operations in the same order)
            FitShared lv = s;
            for (int l = 1; l < L; ++l) { ALL_LANES(fit_chain_level(lv, r[lane], r[lane].lvl[l])); }
            if (h.walk_ok) {
                fit_chain_walk_host(s, h.walk);
                if (std::memcmp(s.W, lv.W, sizeof(float) * 24 * 12) != 0) return -2;
            } else s = lv;
            ALL_LANES(fit_p4(s, r[lane], lane));
            ALL_LANES(fit_p6(s, r[lane], lane, P, true));      // (P5 folded in, as in fit.hip)
            ALL_LANES(fit_p7(s, r[lane], lane));
            ALL_LANES(fit_p8(s, lane, P, it, lh, eh));
            if (h.walk_ok) {
                lv = s;
                for (int l = L - 2; l >= 0; --l) { ALL_LANES(fit_reverse_level(lv, r[lane], r[lane].lvl[l])); }
                fit_reverse_walk_host(s, h.walk, h.walk_nb);
                for (int i = 0; i < 24 * 12; ++i)
                    if (!(s.Afin[i] == lv.Abar[i])) return -3;          // (== : +0 and -0 agree)
                ALL_LANES(fit_p10(s, r[lane], lane, true));
            } else {
                for (int l = L - 2; l >= 0; --l) { ALL_LANES(fit_reverse_level(s, r[lane], r[lane].lvl[l])); }
                ALL_LANES(fit_p10(s, r[lane], lane));
            }
            ALL_LANES(fit_p11(s, r[lane], lane));
            if (it == 0 && grad_first) {      // gradient of the first iteration, record order [cam|R|betas]
                float* g = grad_first + (size_t)b * kNumPrm;
                for (int p = 0; p < kNumPrm; ++p)
                    g[p] = p < kPrmR ? s.red[2 + p]
                         : p < kPrmBeta ? s.grad[p]
                         : [&] { float t = 0.f; for (int q = 0; q < 16; ++q) t += s.part[(p - kPrmBeta) * 16 + q]; return t; }();
            }
            ALL_LANES(fit_p12(s, r[lane], lane, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, step_size, bc2_sqrt));
            if (v2) {
                const float w1 = (float)(1.0 - beta1), b2f = (float)beta2, w2 = (float)(1.0 - beta2), epsf = (float)eps;
                ALL_LANES(fit2_p1(s2, x2, r2[lane], lane));
                fit_chain_walk_host(s2, h.walk);                             // wave 0 ...
                ALL_LANES(fit2_vp(s2, lane));                                // ... beside wave 1
                ALL_LANES(fit_p4(s2, r2[lane], lane));
                {   // P6 on two lanes per joint: the x lane's links of the three fma chains go to the y lane (DPP on the GPU)
                    Fit2P6 q[34];
                    float d[34], dp[34], gu[34], xt[34];
                    for (int lane = 0; lane < 34; ++lane) q[lane] = fit2_p6a(s2, lane, P, d[lane], dp[lane], gu[lane], xt[lane]);
                    for (int lane = 1; lane < 34; lane += 2) fit2_p6b(s2, lane, q[lane - 1], d[lane], dp[lane], gu[lane], xt[lane]);
                }
                ALL_LANES(fit_p7(s2, r2[lane], lane));
                fit_reverse_walk_host(s2, h.walk, h.walk_nb);                // wave 0 ...
                for (int lane = 64; lane < 128; ++lane) fit_p8(s2, lane - 64, P, it, lh2.data(), eh2.data());   // ... beside wave 1
                ALL_LANES(fit2_dj(s2, r2[lane], lane));
                float g_rot[kFitLanes];
                for (int lane = 192; lane < 256; ++lane) fit2_p11a(s2, x2, lane);                               // wave 3, in front of its gradients and steps
                ALL_LANES(g_rot[lane] = fit2_dr(s2, r2[lane], lane));
                ALL_LANES(fit2_step(s2, x2, r2[lane], lane, g_rot[lane], w1, b2f, w2, epsf, step_size, bc2_sqrt, 1.0f / bc2_sqrt, nullptr));
                if (std::memcmp(s.prm, s2.prm, sizeof(float) * kNumPrm) != 0) return -4;
                if (std::memcmp(s.best, s2.best, sizeof(float) * kNumPrm) != 0) return -5;
                if (lh && std::memcmp(&lh[it], &lh2[it], sizeof(float)) != 0) return -6;
                if (eh && std::memcmp(&eh[it], &eh2[it], sizeof(float)) != 0) return -7;
            }
        }
        const float* src = (select_mode == SPSP_FIT_SELECT_BEST_L2E) ? s.best : s.prm;
        for (int p = 0; p < kNumPrm; ++p) {
            if (p < kPrmR) cb[p] = src[p];
            else if (p < kPrmBeta) Rb[p - kPrmR] = src[p];
            else bb[p - kPrmBeta] = src[p];
        }
    }
    return 0;
}
