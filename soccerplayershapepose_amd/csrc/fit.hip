// GPU driver of the reprojection fit: one 256-thread workgroup (4 wavefronts) per player, all iterations in one launch, state in
// LDS (19 KB), Adam moments and the model constants a lane multiplies with in registers (one parameter per lane).  The
// arithmetic lives in fit_core.hpp (shared with the host emulation used by the CPU tests).  Latency-bound by design: the two
// tree sweeps as register-resident walks inside wavefront 0.  Players are independent, so a batch holds one CU per player.
// Round 4: fit_kernel2 -- the same operations on a shorter critical path (eight workgroup barriers per iteration instead of
// ten, a rotation entry's Adam step in the lane that computed its gradient, waves 1 and 3 working beside wave 0's walks;
// fit_core.hpp "round 4"): 7.6 k instead of 8.4 k cycles per iteration, 186 -> 164 us for 50 iterations, every iterate
// bit-identical.  fit_kernel<WALK> keeps the ten-barrier schedule (models with another tree; the A/B switch of the tests).  (Four players per workgroup -- 8 CUs per batch -- was built and measured in
// round 3: the fit got 1.5 x longer and the four-slot pipeline slower, because a slot's stream waits for its fit)
#include <cmath>
#include <cstdlib>

#include "smpl.hpp"
#include "fit_core.hpp"
#include "geometry_device.hpp"

namespace {
using namespace spsp;

// Between two phases that run back to back inside ONE wavefront (no workgroup barrier): the hardware issues a wave's
// LDS instructions in order, so the earlier phase's ds_write is visible to the later phase's ds_read of any lane.  What
// must not happen is the COMPILER moving one lane's read above another phase's write: a wavefront-scope release /
// acquire pair around a wave barrier pins the order (zero instructions are emitted for it).
__device__ __forceinline__ void wave_phase_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// WALK: the two tree sweeps as barrier-free walks (fit_core.hpp; every SMPL tree) instead of level by level inside wavefront 0.
template <bool WALK>
__global__ __launch_bounds__(spsp::kFitLanes) void fit_kernel(FitParams P, const uint32_t* __restrict__ walk, const float* __restrict__ fitc,
                                                 const int* __restrict__ parents, const int* __restrict__ level_joints,
                                                 const int* __restrict__ level_start, const int* __restrict__ child_start,
                                                 const int* __restrict__ child_list, float* __restrict__ R,
                                                 float* __restrict__ betas, float* __restrict__ cam,
                                                 const float* __restrict__ joints2d, float* __restrict__ loss_hist,
                                                 float* __restrict__ l2e_hist, float* __restrict__ param_hist,
                                                 const spsp::FitEnds E, unsigned long long* __restrict__ dbg) {
    __shared__ FitShared s;
    __shared__ float sR0[216];
    FitLane r;
    const int lane = threadIdx.x, b = blockIdx.x;
    const bool wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
    float* Rb = R + (size_t)b * 216;
    float* bb = betas + (size_t)b * 10;
    float* cb = cam + (size_t)b * 3;
    const float* Rsrc = Rb;
    if (E.pose6d) {      // straight from the regressor: the 24 rotation matrices are built here (spsp_rot6d_to_rotmat's statements)
        if (lane < 24) rot6d_one(E.pose6d + ((size_t)b * 24 + lane) * 6, sR0 + lane * 9);
        __syncthreads();
        Rsrc = sR0;
    }
    fit_init(s, r, lane, fitc, parents, level_joints, level_start, child_start, child_list, Rsrc, bb, cb,
             joints2d + (size_t)b * 34, P.param_mask, WALK ? walk : nullptr);
    __syncthreads();
    float* lh = loss_hist ? loss_hist + (size_t)b * P.iters : nullptr;
    float* eh = l2e_hist ? l2e_hist + (size_t)b * P.iters : nullptr;
    double b1t = 1.0, b2t = 1.0;
    const int L = P.num_levels;
    const float w1 = (float)(1.0 - P.beta1), b2f = (float)P.beta2, w2 = (float)(1.0 - P.beta2), epsf = (float)P.eps;
    for (int it = 0; it < P.iters; ++it) {
        b1t *= P.beta1;
        b2t *= P.beta2;
        const float step_size = (float)(P.lr / (1.0 - b1t));
        const float bc2_sqrt = (float)sqrt(1.0 - b2t);
        // Phases that only involve lanes 0..63 run back to back inside wavefront 0 with NO workgroup barrier: a
        // wave issues its LDS instructions in order, so a ds_write is visible to the following ds_read of any lane
        // of the same wave.  That folds the 8 + 8 level-synchronous chain / reverse-sweep steps (<= 60 active lanes
        // each) into two barrier-free runs: 9 barriers per iteration instead of 27.
#ifdef SPSP_FIT_STAMPS      // diagnostic build only: per-phase s_memtime stamps of iteration 2, player 0
#define STAMP(i) if (dbg && it == 2 && b == 0 && lane == 0) dbg[i] = __builtin_amdgcn_s_memtime()
#else
#define STAMP(i)
#endif
        STAMP(0);
        fit_p1(s, r, lane);                       __syncthreads();
        STAMP(1);
        fit_p2(s, lane);                          __syncthreads();
        if constexpr (WALK) { if (wave0) fit_chain_walk(s, r, lane); }
        else if (wave0) {
            uint32_t ql[kMaxLevels];
#pragma unroll
            for (int l = 1; l < kMaxLevels; ++l) ql[l] = r.lvl[l];
#pragma unroll
            for (int l = 1; l < kMaxLevels; ++l)       // static l: the role words stay in registers
                if (l < L) { fit_chain_level(s, r, ql[l]); wave_phase_fence(); }
        }
        STAMP(2);
        __syncthreads();
        STAMP(3);
        fit_p4(s, r, lane);                       __syncthreads();
        STAMP(4);
        if (wave0) fit_p6(s, r, lane, P, true);      // (P5 folded in)
        STAMP(5);
        __syncthreads();
        fit_p7(s, r, lane);                       __syncthreads();
        STAMP(6);
        if constexpr (WALK) {
            if (wave0) {
                fit_p8(s, lane, P, it, lh, eh);
                fit_reverse_walk_a(s, r, lane);
                wave_phase_fence();
                fit_reverse_walk_b(s, r, lane);
            }
        } else if (wave0) {
            uint32_t ql[kMaxLevels];
#pragma unroll
            for (int l = 0; l < kMaxLevels - 1; ++l) ql[l] = r.lvl[l];
            fit_p8(s, lane, P, it, lh, eh);
            wave_phase_fence();
#pragma unroll
            for (int l = kMaxLevels - 2; l >= 0; --l)
                if (l <= L - 2) { fit_reverse_level(s, r, ql[l]); wave_phase_fence(); }
        }
        STAMP(7);
        __syncthreads();
        fit_p10(s, r, lane, WALK);                   __syncthreads();
        STAMP(8);
        fit_p11(s, r, lane);                      __syncthreads();
        STAMP(9);
        // the iterate this iteration evaluated (parameters BEFORE the step): what the reference snapshots when it saves
        if (param_hist && lane < kNumPrm) param_hist[((size_t)b * P.iters + it) * kNumPrm + lane] = s.prm[lane];
        fit_p12(s, r, lane, w1, b2f, w2, epsf, step_size, bc2_sqrt);
        __syncthreads();
        STAMP(10);
    }
    const float* src = (P.select_mode == SPSP_FIT_SELECT_BEST_L2E) ? s.best : s.prm;
    if (lane < kNumPrm) {
        const int p = lane;
        const float v = src[p];
        if (p < kPrmR) cb[p] = v;
        else if (p < kPrmBeta) Rb[p - kPrmR] = v;
        else bb[p - kPrmBeta] = v;
        // the all-gather payload [cam | R | betas] is this very array (spsp_pack_record)
        if (E.record) E.record[(size_t)b * kNumPrm + p] = v;
    }
    if (E.translation && lane < 3)      // spsp_wp_to_translation's statement on the returned camera
        E.translation[(size_t)b * 3 + lane] = wp_translation(src[0], src[1], src[2], E.focal, E.resolution, lane);
}

// The v2 schedule of the same fit (fit_core.hpp, "round 4"): eight workgroup barriers per iteration, the Adam step of a rotation
// entry in the lane that computed its gradient, wave 1 / wave 3 working beside the tree walks of wave 0.  SMPL's tree only (the
// walks are compiled for it); every iterate bit-identical to fit_kernel<true>.
__global__ __launch_bounds__(spsp::kFitLanes) void fit_kernel2(FitParams P, const uint32_t* __restrict__ walk, const float* __restrict__ fitc,
                                                  const int* __restrict__ parents, const int* __restrict__ level_joints,
                                                  const int* __restrict__ level_start, const int* __restrict__ child_start,
                                                  const int* __restrict__ child_list, float* __restrict__ R,
                                                  float* __restrict__ betas, float* __restrict__ cam,
                                                  const float* __restrict__ joints2d, float* __restrict__ loss_hist,
                                                  float* __restrict__ l2e_hist, float* __restrict__ param_hist,
                                                  const spsp::FitEnds E, unsigned long long* __restrict__ dbg, const spsp::FitSchedule sched) {
    __shared__ FitShared s;
    __shared__ Fit2Shared x;
    __shared__ float sR0[216];
    FitLane r;
    const int lane = threadIdx.x, b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* Rb = R + (size_t)b * 216;
    float* bb = betas + (size_t)b * 10;
    float* cb = cam + (size_t)b * 3;
    const float* Rsrc = Rb;
    if (E.pose6d) {      // straight from the regressor: the 24 rotation matrices are built here (spsp_rot6d_to_rotmat's statements)
        if (lane < 24) rot6d_one(E.pose6d + ((size_t)b * 24 + lane) * 6, sR0 + lane * 9);
        __syncthreads();
        Rsrc = sR0;
    }
    fit_init(s, r, lane, fitc, parents, level_joints, level_start, child_start, child_list, Rsrc, bb, cb,
             joints2d + (size_t)b * 34, P.param_mask, walk);
    fit2_init0(x, lane);
    __syncthreads();
    fit2_init(s, x, r, lane, fitc, P.param_mask);
    __syncthreads();
    float* lh = loss_hist ? loss_hist + (size_t)b * P.iters : nullptr;
    float* eh = l2e_hist ? l2e_hist + (size_t)b * P.iters : nullptr;
    double b1t = 1.0, b2t = 1.0;
    const float w1 = (float)(1.0 - P.beta1), b2f = (float)P.beta2, w2 = (float)(1.0 - P.beta2), epsf = (float)P.eps;
    for (int it = 0; it < P.iters; ++it) {
        // Adam's bias corrections: from the launch's table (host, double); beyond it, in the loop as before round 5
        float step_size, inv_bc2;
        if (it < kFitSchedN) { step_size = sched.step_size[it]; inv_bc2 = sched.inv_bc2[it]; }
        else {
            if (it == kFitSchedN) { b1t = sched.b1t_end; b2t = sched.b2t_end; }
            b1t *= P.beta1;
            b2t *= P.beta2;
            step_size = (float)(P.lr / (1.0 - b1t));
            inv_bc2 = (float)(1.0 / sqrt(1.0 - b2t));
        }
        float* const ph = param_hist ? param_hist + ((size_t)b * P.iters + it) * kNumPrm : nullptr;
        STAMP(0);
        fit2_p1(s, x, r, lane);                                      __syncthreads();
        STAMP(1);
        if (wave == 0) fit_chain_walk(s, r, lane);
        else if (wave == 1) fit2_vp(s, lane);
        STAMP(2);
        __syncthreads();
        STAMP(3);
        fit_p4(s, r, lane);                                          __syncthreads();
        STAMP(4);
        if (wave == 0) fit2_p6(s, lane, P);
        STAMP(5);
        __syncthreads();
        fit_p7(s, r, lane);                                          __syncthreads();
        STAMP(6);
        if (wave == 0) {
            fit_reverse_walk_a(s, r, lane);
            wave_phase_fence();
            fit_reverse_walk_b(s, r, lane);
        } else if (wave == 1) fit_p8(s, lane - 64, P, it, lh, eh);
        STAMP(7);
        __syncthreads();
        fit2_dj(s, r, lane);
        __syncthreads();
        STAMP(8);
        if (wave == 3) { fit2_p11a(s, x, lane); wave_phase_fence(); }
        const float g_rot = fit2_dr(s, r, lane);
        fit2_step(s, x, r, lane, g_rot, w1, b2f, w2, epsf, step_size, 0.f, inv_bc2, ph);
        STAMP(9);
        __syncthreads();
        STAMP(10);
    }
    const float* src = (P.select_mode == SPSP_FIT_SELECT_BEST_L2E) ? s.best : s.prm;
    if (lane < kNumPrm) {
        const int p = lane;
        const float v = src[p];
        if (p < kPrmR) cb[p] = v;
        else if (p < kPrmBeta) Rb[p - kPrmR] = v;
        else bb[p - kPrmBeta] = v;
        if (E.record) E.record[(size_t)b * kNumPrm + p] = v;
    }
    if (E.translation && lane < 3)
        E.translation[(size_t)b * 3 + lane] = wp_translation(src[0], src[1], src[2], E.focal, E.resolution, lane);
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

int spsp_smpl_debug(spsp_smpl* s, int what, int value) {
    SPSP_REQUIRE(s, "spsp_smpl_debug: NULL handle");
    if (what == SPSP_DEBUG_FIT_SCHEDULE) {
        SPSP_REQUIRE(value >= 0 && value <= 2, "spsp_smpl_debug: fit schedule %d (0 level sweeps, 1 ten-barrier walks, 2 default)", value);
        s->fit_schedule = value;
        return SPSP_OK;
    }
    if (what == SPSP_DEBUG_SKIN_PLAYERS) {
        SPSP_REQUIRE(value == 8 || value == 16, "spsp_smpl_debug: %d players per skin workgroup (8 or 16)", value);
        s->skin_players = value;
        return SPSP_OK;
    }
    spsp::set_error("spsp_smpl_debug: unknown hook %d", what);
    return SPSP_EINVAL;
}

int spsp_fit(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_config* cfg, float* R, float* betas, float* cam,
             const float* joints2d, int B, float* loss_hist, float* l2e_hist, void* stream) {
    return spsp_fit_trajectory(ctx, s, cfg, R, betas, cam, joints2d, B, loss_hist, l2e_hist, nullptr, stream);
}

int spsp_fit_trajectory(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_config* cfg, float* R, float* betas, float* cam,
                        const float* joints2d, int B, float* loss_hist, float* l2e_hist, float* param_hist, void* stream) {
    return spsp::fit_launch(ctx, s, cfg, R, betas, cam, joints2d, B, loss_hist, l2e_hist, param_hist, spsp::FitEnds{}, stream);
}

}  // extern "C"

namespace spsp {

int fit_launch(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_config* cfg, float* R, float* betas, float* cam, const float* joints2d,
               int B, float* loss_hist, float* l2e_hist, float* param_hist, const FitEnds& ends, void* stream) {
    SPSP_REQUIRE(ctx && s && cfg, "spsp_fit: NULL handle");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));      // a process may hold contexts on several GPUs
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
    P.walk_nb = s->walk_nb;
    ProfScope ps(ctx, kKFit, as_stream(stream), true);
    unsigned long long* dbg = nullptr;
#ifdef SPSP_FIT_STAMPS
    if (getenv("SPSP_FIT_DEBUG")) (void)hipMalloc(&dbg, 16 * sizeof(unsigned long long));
#endif
#define SPSP_FIT_LAUNCH(WALK)                                                                                                        \
    launch_kernel(&fit_kernel<WALK>, dim3(B), dim3(kFitLanes), 0, as_stream(stream), P, s->walk, s->fitc, s->parents, s->level_joints,  \
                  s->d_level_start, s->child_start, s->child_list, R, betas, cam, joints2d, loss_hist, l2e_hist, param_hist, ends, dbg)
    // (the level sweeps remain for models whose tree is not SMPL's; the older schedules are the A/B switch of tests and tools:
    //  spsp_smpl_debug(s, SPSP_DEBUG_FIT_SCHEDULE, 1) = the ten-barrier schedule of rounds 1-3 with the walks, 0 = with the level sweeps)
    if (s->walk_ok && s->fit_schedule == 2) {
        FitSchedule sched;
        double b1t = 1.0, b2t = 1.0;
        for (int t = 0; t < kFitSchedN; ++t) {
            b1t *= cfg->beta1;
            b2t *= cfg->beta2;
            sched.step_size[t] = (float)(cfg->lr / (1.0 - b1t));
            sched.inv_bc2[t] = (float)(1.0 / std::sqrt(1.0 - b2t));
        }
        sched.b1t_end = b1t; sched.b2t_end = b2t;
        launch_kernel(&fit_kernel2, dim3(B), dim3(kFitLanes), 0, as_stream(stream), P, s->walk, s->fitc, s->parents, s->level_joints,
                      s->d_level_start, s->child_start, s->child_list, R, betas, cam, joints2d, loss_hist, l2e_hist, param_hist, ends, dbg, sched);
    }
    else if (s->walk_ok && s->fit_schedule == 1) SPSP_FIT_LAUNCH(true); else SPSP_FIT_LAUNCH(false);
#undef SPSP_FIT_LAUNCH
    if (dbg) {
        unsigned long long h[16];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost);
        const char* names[10] = {"p1+bar", "w0:p2+chain", "bar", "p4+bar", "w0:p5+p6", "bar+p7+bar", "w0:p8+reverse", "bar+p10+bar", "p11+bar", "p12+bar"};
        for (int i = 0; i < 10; ++i) fprintf(stderr, "fit stamp %-14s %6llu cycles\n", names[i], h[i + 1] - h[i]);
        fprintf(stderr, "fit iteration total %llu cycles\n", h[10] - h[0]);
        (void)hipFree(dbg);
    }
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp
