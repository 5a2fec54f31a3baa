// The per-player reprojection fit as a sequence of lane-parallel PHASES over one LDS-resident state.
//
// Replaces the reference's per-player Adam loops (optimize_camera PR/PyTorch3DTest.py:3055-3124;
// single_view_optimization PR/player_recon.py:1172-1294; census in SURVEY Appendix E): instead of a full
// 6890-vertex SMPL.forward + autograd per iteration it evaluates only what the joints2D loss touches
// (24 chain joints + 5 face vertices), with a hand-derived reverse sweep over the
// kinematic tree and torch-default Adam, ALL iterations inside one launch, one wavefront per player.
//
// A phase is a pure function of (lane, shared state): it reads LDS written by earlier phases and writes
// disjoint LDS slots; a barrier separates consecutive phases.  No cross-lane register traffic is used
// (reductions go through LDS partials), so the very same code runs
//   * on the GPU   (fit.hip: kFitLanes = 256 lanes = 4 wavefronts per player, barrier = s_barrier), and
//   * on the host  (tests/emul: `for lane in 0..255` per phase) -- a CPU check of the kernel's index math.
// 256 lanes make every phase a SINGLE pass with <= ~25 FMAs per lane (one parameter per lane in the Adam
// phase); the first version used one wavefront and spent its time in 2-4 pass loops (2470 VALU + 750 LDS
// instructions per iteration, measured) -- the fit is latency-bound, so lanes are cheaper than passes.
//
// Reverse sweep (Appendix D).  With W_k = (Rw_k, tw_k) the world transform of joint k, p = parent(k):
//   direct adjoints  a_k = dL/d tw_k = g_k + sum_v w_vk h_v,   B_k = dL/d Rw_k = sum_v w_vk h_v (vp_v - J_k)^T
//   leaves -> root   abar_p += abar_c,   Bbar_p += abar_c (J_c - J_p)^T + Bbar_c R_c^T
//   parameters       dL/dR_k = Rw_p^T Bbar_k (k>0), dL/dG = Bbar_0,  dL/dR_k[e] += sum_c P5[(k-1)*9+e][c] dvp[c]
//                    dL/dJ_k = -sum_v w_vk Rw_k^T h_v + Rw_p^T abar_k (k>0; abar_0 for the root) - Rw_k^T (abar_k - a_k)
//                    dL/dvp_v = sum_k w_vk Rw_k^T h_v,   dL/dbeta = S5^T dvp + JS^T dJ
#pragma once
#include <math.h>
#include <stdint.h>

#include "smpl_consts.hpp"

#if defined(__HIPCC__)
#define SPSP_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define SPSP_HD inline
#endif

namespace spsp {

constexpr int kFitLanes = 256;                // lanes (threads) cooperating on one player
constexpr int kNumPrm = SPSP_RECORD_FLOATS;   // 229: [cam(3) | R(216) | betas(10)]
constexpr int kPrmR = 3, kPrmBeta = 219;

// fit joint i reads output joint kFitSrc[i] (PR/config.py:38); < 24: chain joint, >= 24: face vertex (idx-24)
#define SPSP_FIT_SRC {24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8}

struct FitParams {           // kernel arguments (plain data)
    int iters;
    double lr, beta1, beta2, eps;
    uint32_t param_mask;
    float img_wh, loss_wh;
    float log_var, weight;   // lv = -ln(loss_weight + 1e-6) as fp32, w = exp(-lv)
    int select_mode;
    int num_levels;
};

struct FitShared {
    float fc[kFcTotal];
    float prm[232];
    float best[232];
    float grad[232];
    float J[72];
    float W[288];
    float vp[16];
    float xf[16];
    float part[256];
    float part2[256];
    float g[34];
    float ga[72];            // dL/d(tw_k) seeds of the chain joints that are fit targets (others stay 0; z stays 0)
    float h[16];             // dL/d(x_v) of the 5 face vertices (z stays 0)
    int fit_src[17];         // fit joint i reads output joint fit_src[i] (PR/config.py:38)
    float rpart[17 * 5];     // per-fit-joint partials: loss, l2e, gs, gtx, gty
    float red[8];            // reduced: loss, l2e, gs, gtx, gty
    float adir[72], abar[72], Bbar[216];
    float dvp[16], dJ[72];
    float adj[96];           // [dvp(15) | dJ_total(72) | 0(9)]: what d(beta) contracts with fc[kFcC]
    float tgt_px[34];
    float best_l2e;
    int take_best;
    int parents[24], level_joints[24], level_start[26], child_start[25], child_list[24];
};

constexpr int kMaxLevels = 12;   // tree depth limit (SMPL: 9 levels); checked at spsp_smpl_load

// Per-lane registers that live across iterations: Adam moments of parameter `lane` and this lane's ROLE in the
// level-synchronous phases (which joint, its parent, its children), looked up ONCE.  Reading the tree tables from
// LDS inside the loop cost 4-5 dependent LDS round trips per level (measured: ~900 cycles per phase).
struct FitLane {
    float m, v;
    signed char ck[kMaxLevels], cp[kMaxLevels];        // forward chain, level l: joint / parent handled (or -1)
    signed char rp[kMaxLevels], rc[kMaxLevels][3];     // reverse sweep, level l: joint and its (<= 3) children (or -1)
    signed char par216, par72;                         // parent of joint lane/9 (P10 rotation item) / lane/3 (dJ item)
    signed char src17;                                 // P6: source joint of fit joint `lane`
};

SPSP_HD bool fit_trainable(uint32_t mask, int p) {
    if (p < kPrmR) return (mask >> 24) & 1u;
    if (p < kPrmBeta) return (mask >> ((p - kPrmR) / 9)) & 1u;
    return (mask >> 25) & 1u;
}

// ---- one-time set-up -----------------------------------------------------------------------------------
SPSP_HD void fit_init(FitShared& s, FitLane& r, int lane, const float* fitc, const int* parents,
                      const int* level_joints, const int* level_start, const int* child_start,
                      const int* child_list, const float* R, const float* betas, const float* cam,
                      const float* joints2d) {
    for (int i = lane; i < kFcTotal; i += kFitLanes) s.fc[i] = fitc[i];
    if (lane < 232) {
        const int p = lane;
        float v = 0.f;
        if (p < kPrmR) v = cam[p];
        else if (p < kPrmBeta) v = R[p - kPrmR];
        else if (p < kNumPrm) v = betas[p - kPrmBeta];
        s.prm[p] = v;
        s.best[p] = v;
        s.grad[p] = 0.f;
    }
    if (lane < 34) s.tgt_px[lane] = joints2d[lane];
    if (lane < 24) {
        s.parents[lane] = parents[lane];
        s.level_joints[lane] = level_joints[lane];
        s.child_list[lane] = child_list[lane];
    }
    if (lane < 26) s.level_start[lane] = level_start[lane];
    if (lane < 25) s.child_start[lane] = child_start[lane];
    if (lane == 0) { s.best_l2e = INFINITY; s.take_best = 0; }
    {
        const int src_tab[17] = SPSP_FIT_SRC;
        if (lane < 17) s.fit_src[lane] = src_tab[lane];
        if (lane < 72) s.ga[lane] = 0.f;
        if (lane < 16) s.h[lane] = 0.f;
        if (lane < 96) s.adj[lane] = 0.f;
    }
    r.m = r.v = 0.f;
    // roles (the tree tables are kernel arguments in global memory here; never touched again)
    const int slot = lane / 12;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int l = 0; l < kMaxLevels; ++l) {
        r.ck[l] = r.cp[l] = r.rp[l] = -1;
        r.rc[l][0] = r.rc[l][1] = r.rc[l][2] = -1;
        const int idx = level_start[l < 25 ? l : 25] + slot;
        if (l < 25 && slot < 20 && idx < level_start[l + 1]) {
            const int k = level_joints[idx];
            r.ck[l] = (signed char)k;
            r.cp[l] = (signed char)parents[k];
            r.rp[l] = (signed char)k;
            const int c0 = child_start[k], nc = child_start[k + 1] - c0;
            if (nc > 0) r.rc[l][0] = (signed char)child_list[c0];
            if (nc > 1) r.rc[l][1] = (signed char)child_list[c0 + 1];
            if (nc > 2) r.rc[l][2] = (signed char)child_list[c0 + 2];
        }
    }
    r.par216 = (signed char)(lane < 216 ? parents[lane / 9] : -1);
    r.par72 = (signed char)(lane < 72 ? parents[lane / 3] : -1);
    {
        const int src_tab2[17] = SPSP_FIT_SRC;
        r.src17 = (signed char)(lane < 17 ? src_tab2[lane] : 0);
    }
}

// ---- forward ---------------------------------------------------------------------------------------------
// P1: J = Jbar + JS.beta (lanes 0..71); partial sums of the 5 posed face vertices over 16 chunks of 13 pose
//     features (+ template + shape term in chunk 0).  lane = q*16 + c, c = v*3+i < 15.
//     vp = vt5 + S5.beta + sum_j P5[j] (R[9+j] - delta_j): the delta part is folded into fc[kFcVt5] at load.
SPSP_HD void fit_p1(FitShared& s, int lane) {
    const float* beta = s.prm + kPrmBeta;
    if (lane < 72) {
        float acc = s.fc[kFcJbar + lane];
        for (int i = 0; i < 10; ++i) acc = fmaf(s.fc[kFcJS + i * kJSPitch + lane], beta[i], acc);
        s.J[lane] = acc;
    }
    const int q = lane >> 4, c = lane & 15;
    float acc = 0.f;
    if (c < 15) {
        if (q == 15) {      // the shortest chunk (12 features) also carries the template + shape term
            acc = s.fc[kFcVt5 + c];
            for (int i = 0; i < 10; ++i) acc = fmaf(s.fc[kFcS5 + i * kF5Pitch + c], beta[i], acc);
        }
        // 16 x 13 = 208 rows: row 207 of P5 is a zero pad (prm[kPrmR + 216] = beta[0] is a finite dummy factor)
        const float* P5 = s.fc + kFcP5 + (q * 13) * kF5Pitch + c;
        const float* Rv = s.prm + kPrmR + 9 + q * 13;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int t = 0; t < 13; ++t) acc = fmaf(P5[t * kF5Pitch], Rv[t], acc);
    }
    s.part[lane] = acc;
}

// P2: finish vp (lanes 0..14: fixed-order sum of the 16 partials); root transform W_0 = [G | J_0] (lanes 16..27).
SPSP_HD void fit_p2(FitShared& s, int lane) {
    if (lane < 15) {
        float acc = 0.f;
        for (int q = 0; q < 16; ++q) acc += s.part[q * 16 + lane];
        s.vp[lane] = acc;
    }
    if (lane >= 16 && lane < 28) {
        const int e = lane - 16, i = e >> 2, c = e & 3;
        s.W[e] = (c < 3) ? s.prm[kPrmR + i * 3 + c] : s.J[i];
    }
}

// P3[l]: kinematic chain, level l >= 1: W_k = W_p * [R_k | J_k - J_p]; lane = (joint slot of the level, element).
SPSP_HD void fit_chain_level(FitShared& s, const FitLane& r, int lane, int l) {
    const int k = r.ck[l];
    if (k < 0) return;
    const int p = r.cp[l];
    const int e = lane % 12, i = e >> 2, c = e & 3;
    const float* Rm = s.prm + kPrmR;
    const float* Wp = s.W + p * 12 + i * 4;
    float acc;
    if (c < 3) {
        acc = Wp[0] * Rm[k * 9 + c];
        acc = fmaf(Wp[1], Rm[k * 9 + 3 + c], acc);
        acc = fmaf(Wp[2], Rm[k * 9 + 6 + c], acc);
    } else {
        acc = Wp[0] * (s.J[k * 3] - s.J[p * 3]);
        acc = fmaf(Wp[1], s.J[k * 3 + 1] - s.J[p * 3 + 1], acc);
        acc = fmaf(Wp[2], s.J[k * 3 + 2] - s.J[p * 3 + 2], acc);
        acc += Wp[3];
    }
    s.W[k * 12 + e] = acc;
}

// P4: face vertices x_v = sum_k w_vk (Rw_k (vp_v - J_k) + tw_k); lane = q*16 + (v*3+i), q < 12 = pair of joints.
SPSP_HD void fit_p4(FitShared& s, int lane) {
    const int q = lane >> 4, o = lane & 15;
    if (q >= 12) return;
    float acc = 0.f;
    if (o < 15) {
        const int v = o / 3, i = o % 3;
        for (int k = q * 2; k < q * 2 + 2; ++k) {
            const float* w = s.W + k * 12 + i * 4;
            float t = fmaf(w[0], s.vp[v * 3] - s.J[k * 3],
                           fmaf(w[1], s.vp[v * 3 + 1] - s.J[k * 3 + 1],
                                fmaf(w[2], s.vp[v * 3 + 2] - s.J[k * 3 + 2], w[3])));
            acc = fmaf(s.fc[kFcW5 + v * 24 + k], t, acc);
        }
    }
    s.part[lane] = acc;
}

SPSP_HD void fit_p5(FitShared& s, int lane) {
    if (lane < 15) {
        float acc = 0.f;
        for (int q = 0; q < 12; ++q) acc += s.part[q * 16 + lane];
        s.xf[lane] = acc;
    }
}

// P6: project the 17 fit joints, residuals, per-joint loss / l2e / camera-gradient partials, dL/dX.
//     pixels = (s (X + t) + 1) * img_wh/2  (cam_utils.py:5-26, joints2d_utils.py:5-10);
//     n = 2 px / loss_wh - 1 on both sides, L = mean(d^2) * w + lv  (multi_task_loss.py:97-113).
SPSP_HD void fit_p6(FitShared& s, const FitLane& r, int lane, const FitParams& P) {
    if (lane >= 17) return;
    const int src = r.src17;
    float X[2];
    if (src < 24) { X[0] = s.W[src * 12 + 3]; X[1] = s.W[src * 12 + 7]; }
    else { X[0] = s.xf[(src - 24) * 3]; X[1] = s.xf[(src - 24) * 3 + 1]; }
    const float sc = s.prm[0];
    const float half = P.img_wh / 2.0f;
    const float dn_du = (2.0f * half) / P.loss_wh;           // d n / d u
    const float coef = (2.0f * P.weight) / 34.0f;            // d L / d n = coef * (n - n*)
    float loss = 0.f, l2 = 0.f, gs = 0.f, gt[2];
    for (int c = 0; c < 2; ++c) {
        const float t = s.prm[1 + c];
        const float u = sc * (X[c] + t);
        const float px = (u + 1.f) * half;
        const float y = s.tgt_px[lane * 2 + c];
        const float n = (2.0f * px) / P.loss_wh - 1.0f;
        const float ns = (2.0f * y) / P.loss_wh - 1.0f;
        const float d = n - ns;
        loss = fmaf(d, d, loss);
        const float dp = px - y;
        l2 = fmaf(dp, dp, l2);
        const float gu = coef * d * dn_du;                   // dL/du
        s.g[lane * 2 + c] = gu * sc;                         // dL/dX_c
        if (src < 24) s.ga[src * 3 + c] = gu * sc;           // seeds of the reverse sweep
        else s.h[(src - 24) * 3 + c] = gu * sc;
        gs = fmaf(gu, X[c] + t, gs);
        gt[c] = gu * sc;
    }
    float* rp = s.rpart + lane * 5;
    rp[0] = loss; rp[1] = sqrtf(l2); rp[2] = gs; rp[3] = gt[0]; rp[4] = gt[1];
}

// ---- backward ------------------------------------------------------------------------------------------
// P7: reductions of P6 (lanes 0..4); lanes 64..135: direct adjoints a_k, B_k and direct dJ for item (k, i);
//     lanes 136..255: dvp partials over 3 joints each.
SPSP_HD void fit_p7(FitShared& s, int lane) {
    if (lane < 5) {
        float acc = 0.f;
        for (int i = 0; i < 17; ++i) acc += s.rpart[i * 5 + lane];
        s.red[lane] = acc;
    }
    if (lane >= 64 && lane < 136) {
        const int it = lane - 64, k = it / 3, i = it % 3;
        float a = s.ga[it];
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, dj = 0.f;
        for (int v = 0; v < 5; ++v) {
            const float w = s.fc[kFcW5 + v * 24 + k];
            const float wh = w * s.h[v * 3 + i];
            a += wh;
            b0 = fmaf(wh, s.vp[v * 3] - s.J[k * 3], b0);
            b1 = fmaf(wh, s.vp[v * 3 + 1] - s.J[k * 3 + 1], b1);
            b2 = fmaf(wh, s.vp[v * 3 + 2] - s.J[k * 3 + 2], b2);
            // direct dJ[k][l = i] = -sum_v w_vk sum_i' Rw_k[i'][l] h_v[i']   (h_v[2] == 0)
            dj = fmaf(w, fmaf(s.W[k * 12 + i], s.h[v * 3], s.W[k * 12 + 4 + i] * s.h[v * 3 + 1]), dj);
        }
        s.adir[it] = a;
        s.abar[it] = a;
        s.Bbar[k * 9 + i * 3] = b0; s.Bbar[k * 9 + i * 3 + 1] = b1; s.Bbar[k * 9 + i * 3 + 2] = b2;
        s.dJ[it] = -dj;
    }
    if (lane >= 136) {   // dvp[v][l] partial over joints 3q..3q+2: sum_k w_vk sum_i Rw_k[i][l] h_v[i]
        const int idx = lane - 136, q = idx / 15, o = idx % 15;
        const int v = o / 3, l = o % 3;
        const float h0 = s.h[v * 3], h1 = s.h[v * 3 + 1];
        float acc = 0.f;
        for (int k = q * 3; k < q * 3 + 3; ++k)
            acc = fmaf(s.fc[kFcW5 + v * 24 + k], fmaf(s.W[k * 12 + l], h0, s.W[k * 12 + 4 + l] * h1), acc);
        s.part2[q * 16 + o] = acc;
    }
}

// P8: finish dvp; lane 63 turns the reductions into loss / l2e, history and the best-iterate decision.
SPSP_HD void fit_p8(FitShared& s, int lane, const FitParams& P, int it, float* loss_hist, float* l2e_hist) {
    if (lane < 15) {
        float acc = 0.f;
        for (int q = 0; q < 8; ++q) acc += s.part2[q * 16 + lane];
        s.dvp[lane] = acc;
        s.adj[lane] = acc;
    }
    if (lane == 63) {
        const float loss = (s.red[0] / 34.0f) * P.weight + P.log_var;
        const float l2e = s.red[1] / 17.0f;
        if (loss_hist) loss_hist[it] = loss;
        if (l2e_hist) l2e_hist[it] = l2e;
        const int take = !(l2e > s.best_l2e);                // ties -> later iterate (tracker uses '>')
        s.take_best = take;
        if (take) s.best_l2e = l2e;
    }
}

// P9[l]: reverse sweep -- every joint p of level l gathers from its children (level l+1, already final).
SPSP_HD void fit_reverse_level(FitShared& s, const FitLane& r, int lane, int l) {
    const int p = r.rp[l];
    if (p < 0) return;
    const int e = lane % 12;
    const float* Rm = s.prm + kPrmR;
    if (e < 3) {
        float acc = s.abar[p * 3 + e];
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int n = 0; n < 3; ++n) {
            const int ch = r.rc[l][n];
            if (ch >= 0) acc += s.abar[ch * 3 + e];
        }
        s.abar[p * 3 + e] = acc;
    } else {
        const int i = (e - 3) / 3, c = (e - 3) % 3;
        float acc = s.Bbar[p * 9 + i * 3 + c];
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int n = 0; n < 3; ++n) {
            const int ch = r.rc[l][n];
            if (ch < 0) continue;
            acc = fmaf(s.abar[ch * 3 + i], s.J[ch * 3 + c] - s.J[p * 3 + c], acc);
            const float* Bc = s.Bbar + ch * 9 + i * 3;
            const float* Rc = Rm + ch * 9 + c * 3;     // row c of R_child:  (Bbar_c R_c^T)[i][c] = sum_m Bbar_c[i][m] R_c[c][m]
            acc = fmaf(Bc[0], Rc[0], fmaf(Bc[1], Rc[1], fmaf(Bc[2], Rc[2], acc)));
        }
        s.Bbar[p * 9 + i * 3 + c] = acc;
    }
}

// P10: rotation gradients incl. the pose-blend-shape path through the face vertices (lanes 0..215) and the
//      total dJ (lanes 0..71, after their rotation item).
SPSP_HD void fit_p10(FitShared& s, const FitLane& r, int lane) {
    if (lane < 216) {                                        // item (k, l, c): dR_k[l][c]
        const int k = lane / 9, e = lane % 9, l = e / 3, c = e % 3;
        float acc;
        if (k == 0) acc = s.Bbar[e];
        else {
            const float* Wp = s.W + r.par216 * 12;
            acc = Wp[l] * s.Bbar[k * 9 + c];
            acc = fmaf(Wp[4 + l], s.Bbar[k * 9 + 3 + c], acc);
            acc = fmaf(Wp[8 + l], s.Bbar[k * 9 + 6 + c], acc);
            const float* P5 = s.fc + kFcP5 + ((k - 1) * 9 + e) * kF5Pitch;
            for (int o = 0; o < 15; ++o) acc = fmaf(P5[o], s.dvp[o], acc);
        }
        s.grad[kPrmR + lane] = acc;
    }
    if (lane < 72) {                                         // dJ_total[k][l]
        const int k = lane / 3, l = lane % 3;
        float acc = s.dJ[lane];
        if (k == 0) acc += s.abar[l];
        else {
            const float* Wp = s.W + r.par72 * 12;
            acc = fmaf(Wp[l], s.abar[k * 3], fmaf(Wp[4 + l], s.abar[k * 3 + 1], fmaf(Wp[8 + l], s.abar[k * 3 + 2], acc)));
        }
        const float* Wk = s.W + k * 12;
        const float c0 = s.abar[k * 3] - s.adir[k * 3], c1 = s.abar[k * 3 + 1] - s.adir[k * 3 + 1],
                    c2 = s.abar[k * 3 + 2] - s.adir[k * 3 + 2];
        acc -= fmaf(Wk[l], c0, fmaf(Wk[4 + l], c1, Wk[8 + l] * c2));
        s.adj[15 + lane] = acc;
    }
}

// P11: d beta partials over 16 chunks of 6 of the adjoint vector adj = [dvp(15) | dJ_total(72) | 0(9)];
//      lane = q*16 + i, i < 10.
SPSP_HD void fit_p11(FitShared& s, int lane) {
    const int q = lane >> 4, i = lane & 15;
    float acc = 0.f;
    if (i < 10) {
        const float* C = s.fc + kFcC + i * kCPitch + q * 6;
        const float* A = s.adj + q * 6;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int t = 0; t < 6; ++t) acc = fmaf(C[t], A[t], acc);
    }
    s.part[lane] = acc;
}

// P12: best-iterate snapshot (pre-step parameters) and the torch.optim.Adam step (single-tensor path), one
//   parameter per lane:
//   m = lerp(m, g, 1-b1); v = v*b2 + (1-b2)*g*g; denom = sqrt(v)/sqrt(1-b2^t) + eps; p += (-lr/(1-b1^t) * m) / denom
SPSP_HD void fit_p12(FitShared& s, FitLane& r, int lane, const FitParams& P, float step_size, float bc2_sqrt) {
    const int p = lane;
    if (p >= kNumPrm) return;
    const float w1 = (float)(1.0 - P.beta1), b2 = (float)P.beta2, w2 = (float)(1.0 - P.beta2), eps = (float)P.eps;
    const float cur = s.prm[p];
    if (s.take_best) s.best[p] = cur;
    if (!fit_trainable(P.param_mask, p)) return;
    float g;
    if (p < kPrmR) g = s.red[2 + p];
    else if (p < kPrmBeta) g = s.grad[p];
    else {
        const int i = p - kPrmBeta;
        g = 0.f;
        for (int q = 0; q < 16; ++q) g += s.part[q * 16 + i];
    }
    r.m = r.m + w1 * (g - r.m);
    r.v = r.v * b2 + (w2 * g) * g;
    const float denom = sqrtf(r.v) / bc2_sqrt + eps;
    s.prm[p] = cur + ((-step_size) * r.m) / denom;
}

}  // namespace spsp
