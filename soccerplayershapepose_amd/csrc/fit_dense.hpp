// Phases added to the reprojection fit (fit_core.hpp) when a loss on ALL vertices -- the silhouette term, SURVEY 8f-4 --
// contributes: its per-vertex gradient arrives condensed into seeds (smpl_vjp.hip) that are added to the direct adjoints the
// joints2D loss leaves behind, before the reverse sweep over the kinematic tree.
#pragma once
#include "fit_core.hpp"

namespace spsp {

// seeds [24][12]: row (k, i) = (S1_k[i][0], S1_k[i][1], S1_k[i][2], S0_k[i]),  S0_k = sum_v w_vk h_v,  S1_k = sum_v w_vk h_v vp_v^T.
// Lane 64 + (k * 3 + i) owns item (k, i) in P7 and here (no barrier needed between the two):
//   a_k[i]   += S0_k[i]                                  direct adjoint of tw_k
//   B_k[i][c] += S1_k[i][c] - S0_k[i] J_k[c]             direct adjoint of Rw_k  (x_v = sum_k w_vk (Rw_k (vp_v - J_k) + tw_k))
//   dJ_k[i]  -= sum_i' Rw_k[i'][i] S0_k[i']              direct part of dL/dJ_k
SPSP_HD void fit_dense_inject(FitShared& s, int lane, const float* seeds) {
    if (lane < 64 || lane >= 136) return;
    const int it = lane - 64, k = it / 3, i = it % 3;
    const float* sd = seeds + k * 12 + i * 4;
    const float a = sd[3];
    float* row = s.Abar + k * 12 + i * 4;
    row[0] += sd[0] - a * s.J[k * 3];
    row[1] += sd[1] - a * s.J[k * 3 + 1];
    row[2] += sd[2] - a * s.J[k * 3 + 2];
    row[3] += a;
    s.adir[it] += a;
    const float* Wk = s.W + k * 12;
    s.dJ[it] -= fmaf(Wk[i], seeds[k * 12 + 3], fmaf(Wk[4 + i], seeds[k * 12 + 7], Wk[8 + i] * seeds[k * 12 + 11]));
}

// d(beta_i) of the sparse path after P11 (the sum P12 takes before its Adam step), i < 10
SPSP_HD float fit_beta_grad(const FitShared& s, int i) {
    const float* pp = s.part + i * 16;
    const F4 a = ld4(pp), b = ld4(pp + 4), c = ld4(pp + 8), d = ld4(pp + 12);
    return (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
}

}  // namespace spsp
