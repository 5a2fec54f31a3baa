// Multi-view joint fit: phases added to the per-player fit of fit_core.hpp.
//
// Replaces the optimisation body of `multi_view_optimization` (PR/player_recon.py:1691-1979): V <= 4 views of ONE
// player share the body pose and betas and have their own camera [s,tx,ty] and global orientation.  Three rounds of
//   block 1  torch.optim.Adam([cam_wp_mult, global_orient_mult])   -- the stacked per-view tensors (:1734-1742)
//   block 2  torch.optim.Adam([body_poses_without_hands_feet, betas_mean])  (:1862-1871)
// with `iters` epochs each; an epoch is one TRAIN step per view (joints2D loss of that view only, backward, optimiser
// step) followed by a VALIDATION pass over the views whose mean joints2D_l2es decides the best iterate (:1818-1838).
// Semantics that must be kept and are easy to miss:
//   * the optimiser of block 1 owns the STACKED tensors, so every step moves EVERY view's camera and orientation: the
//     view being trained with its gradient, the others with a zero gradient through their Adam moments; the step
//     counter advances once per view;
//   * the best-iterate record is global over blocks and rounds (`best_epoch_val_metrics` is initialised once, :1717),
//     and at the end of a block the block's parameters go back to their values at the block's best epoch
//     (`cam_wp_mult = best_cam_wp_mult`, :1844; `betas_mean = best_betas_mean`, :1956) -- to the values at the block's
//     start if no epoch of the block reached the record;
//   * in the reference the posed tensor handed to SMPL in block 2 is concatenated ONCE per round, before
//     `requires_grad` is switched on (:1724-1731), so no gradient ever reaches the body pose and block 2 moves betas
//     only.  mask_shared = BETAS reproduces that; BETAS | body-19 is what the loop evidently intends (the posed tensor
//     rebuilt from the leaf every step, as the single-view loop does at :1202-1206).
// Not reproduced: the reference shuffles the view order of every pass with `random.shuffle` (the order here is
// 0..V-1), and its record also gates on a rasterised-silhouette metric (row 8f-4).
//
// Like fit_core.hpp every phase is a pure function of (lane, LDS state), shared with the host emulation under tests/emul.
#pragma once
#include "fit_core.hpp"

namespace spsp {

constexpr int kMaxViews = 4;
constexpr int kViewPrm = 12;                  // [cam(3) | G(9)] per view

struct alignas(16) MvShared {
    float vprm[kMaxViews * kViewPrm];         // current per-view parameters
    float vtgt[kMaxViews * 34];               // per-view detected joints (pixels)
    float blk_views[kMaxViews * kViewPrm];    // block 1: values at the block's best epoch (block start until one is found)
    float blk_shared[232];                    // block 2: the same for the shared [. | R | betas] record
    float best_views[kMaxViews * kViewPrm];   // global best record
    float best_shared[232];
    float best_val;                           // lowest mean joints2D_l2es so far (inf at start)
    float val_acc;                            // sum over views of the current validation pass
    int improved;
};

// view v's camera / orientation / targets into the single-view state the phases of fit_core.hpp work on
SPSP_HD void mv_load_view(FitShared& s, const MvShared& m, int lane, int v) {
    if (lane < kViewPrm) s.prm[lane] = m.vprm[v * kViewPrm + lane];
    if (lane >= 64 && lane < 98) s.tgt_px[lane - 64] = m.vtgt[v * 34 + (lane - 64)];
}

// block start: fresh optimiser (moments and trainable set per lane), block-best := current
SPSP_HD void mv_block_start(FitShared& s, MvShared& m, FitLane& r, int lane, int block, int V, uint32_t mask_shared) {
    r.m = r.v = 0.f;
    if (block == 0) {
        r.trainable = lane < V * kViewPrm;                    // lane = view * 12 + entry
        if (lane < V * kViewPrm) m.blk_views[lane] = m.vprm[lane];
    } else {
        r.trainable = lane >= kViewPrm && lane < kNumPrm ? (int)fit_trainable(mask_shared, lane) : 0;
        if (lane < 232) m.blk_shared[lane] = s.prm[lane];
    }
    if (lane == 0) s.take_best = 0;
}

// block 1 optimiser step after the backward pass of view v: torch.optim.Adam over the stacked [V,...] tensors --
// every view's entries step, only view v's carry a gradient (cam: red[2..4], G: grad[kPrmR..kPrmR+8])
SPSP_HD void mv_adam_views(FitShared& s, MvShared& m, FitLane& r, int lane, int v, int V, float w1, float b2, float w2,
                           float eps, float step_size, float bc2_sqrt) {
    if (lane >= V * kViewPrm) return;
    const int u = lane / kViewPrm, p = lane - u * kViewPrm;
    const float g = u == v ? (p < kPrmR ? s.red[2 + p] : s.grad[p]) : 0.f;
    r.m = r.m + w1 * (g - r.m);
    r.v = r.v * b2 + (w2 * g) * g;
    const float denom = sqrtf(r.v) / bc2_sqrt + eps;
    m.vprm[lane] = m.vprm[lane] + ((-step_size) * r.m) / denom;
}

// validation: joints2D_l2es of the view just evaluated (red[1] = sum over the 17 joints of the pixel distance)
SPSP_HD void mv_val_add(const FitShared& s, MvShared& m, int lane, int first) {
    if (lane == 0) m.val_acc = (first ? 0.f : m.val_acc) + s.red[1] / 17.0f;
}

// end of an epoch's validation pass: mean over views against the record ('>' keeps ties -> the later epoch wins)
SPSP_HD void mv_epoch_decide(MvShared& m, int lane, int V, float* val_hist, int idx) {
    if (lane != 0) return;
    const float val = m.val_acc / (float)V;
    if (val_hist) val_hist[idx] = val;
    m.improved = !(val > m.best_val);
    if (m.improved) m.best_val = val;
}

SPSP_HD void mv_epoch_snapshot(const FitShared& s, MvShared& m, int lane, int block, int V) {
    if (!m.improved) return;
    if (lane < V * kViewPrm) { m.best_views[lane] = m.vprm[lane]; if (block == 0) m.blk_views[lane] = m.vprm[lane]; }
    if (lane < 232) { m.best_shared[lane] = s.prm[lane]; if (block == 1) m.blk_shared[lane] = s.prm[lane]; }
}

// block end: the block's parameters return to their values at the block's best epoch
SPSP_HD void mv_block_end(FitShared& s, MvShared& m, int lane, int block, int V) {
    if (block == 0) { if (lane < V * kViewPrm) m.vprm[lane] = m.blk_views[lane]; }
    else if (lane >= kViewPrm && lane < 232) s.prm[lane] = m.blk_shared[lane];
}

}  // namespace spsp
