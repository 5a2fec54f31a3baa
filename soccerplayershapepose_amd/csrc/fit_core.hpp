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
#include <string.h>
#include <stddef.h>
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
constexpr int kMaxLevels = 9;                 // tree depth limit = the 9 levels of the SMPL tree; checked at spsp_smpl_load (smpl_host.hpp)

// fit joint i reads output joint kFitSrc[i] (PR/config.py:38); < 24: chain joint, >= 24: face vertex (idx-24)
#define SPSP_FIT_SRC {24, 26, 25, 28, 27, 16, 17, 18, 19, 20, 21, 1, 2, 4, 5, 7, 8}

#if defined(__HIPCC__)
#define SPSP_UNROLL _Pragma("unroll")
#else
#define SPSP_UNROLL
#endif
// keeps the optimiser from hoisting what is derived from a loop-invariant register out of the iteration loop (GPU only)
#if defined(__HIP_DEVICE_COMPILE__)
#define SPSP_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define SPSP_OPAQUE(x) (void)(x)
#endif

// four consecutive floats from a 16-byte aligned address: ONE ds_read_b128 on the GPU (a lone wave issues LDS
// instructions slowly, so the fit counts LDS instructions, not bytes)
struct F4 { float x, y, z, w; };
SPSP_HD F4 ld4(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float4 v = *reinterpret_cast<const float4*>(p);
    return F4{v.x, v.y, v.z, v.w};
#else
    return F4{p[0], p[1], p[2], p[3]};
#endif
}
SPSP_HD void st4(float* p, float a, float b, float c, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
#else
    p[0] = a; p[1] = b; p[2] = c; p[3] = d;
#endif
}
SPSP_HD float dot4(const F4& a, const F4& b, float acc) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc)))); }

// Round 5: the v2 schedule stops being bit-identical to the older ones ON THE GPU (the parity bar is the oracle: 5e-5 per golden
// iterate, 1e-4 live).  Its Adam step and the l2e metric take the hardware's 1-ulp square root and reciprocal (v_sqrt_f32,
// v_rcp_f32) in place of the IEEE sequences (~30 dependent instructions of a lone wave on the critical path of every
// iteration), and the two bias corrections come from a per-launch table (FitSchedule) instead of ~32 fp64 instructions --
// a division and a square root in double -- per iteration.  The host twins keep the IEEE forms, so tests/emul still runs the
// schedules side by side bit for bit.
#if defined(__HIP_DEVICE_COMPILE__)
SPSP_HD float fit_sqrt_1ulp(float x) { return __builtin_amdgcn_sqrtf(x); }
SPSP_HD float fit_rcp_1ulp(float x) { return __builtin_amdgcn_rcpf(x); }
#else
SPSP_HD float fit_sqrt_1ulp(float x) { return sqrtf(x); }
SPSP_HD float fit_rcp_1ulp(float x) { return 1.0f / x; }
#endif

// torch.optim.Adam's two bias-correction factors of iterations 1 .. kFitSchedN, computed on the host in double as the kernels'
// loops do (running products of beta1 / beta2) and passed by value: step_size = lr / (1 - beta1^t), inv_bc2 = 1 / sqrt(1 - beta2^t).
// Longer fits compute the factors in the loop from iteration kFitSchedN on.
constexpr int kFitSchedN = 112;
struct FitSchedule { float step_size[kFitSchedN]; float inv_bc2[kFitSchedN]; double b1t_end, b2t_end; };   // (..._end: beta^kFitSchedN, where the loop takes over)

struct FitParams {           // kernel arguments (plain data)
    int iters;
    double lr, beta1, beta2, eps;
    uint32_t param_mask;
    float img_wh, loss_wh;
    float log_var, weight;   // lv = -ln(loss_weight + 1e-6) as fp32, w = exp(-lv)
    int select_mode;
    int num_levels;
    int walk_nb;             // joints with a branching subtree (SmplHostImage::walk_nb)
};

// LDS state of one player.  Float offsets of the arrays the level sweeps address by precomputed lane offsets:
struct alignas(16) FitShared {
    alignas(16) float W[25 * 12];      // world transforms, 3x4 row-major [R | t]; slot 24 unused
    alignas(16) float Abar[25 * 12];   // reverse-sweep adjoints, 3x4 row-major [Bbar | abar]; slot 24 = zeros (absent child)
    alignas(16) float Mt[25 * 16];     // (slot 24 = identity: padding of the walks)  forward operand: Mt[k][c] = column c of [[R_k, Jrel_k], [0, 1]]  (4 floats each)
    alignas(16) float Tt[25 * 16];     // reverse operand: Tt[k][c] = (row c of R_k, Jrel_k[c]) for c < 3, (0,0,0,1) for c = 3
    alignas(16) float part[256];
    alignas(16) float part2[256];
    alignas(16) float dvp[16];
    alignas(16) float unit[4];         // (0, 0, 0, 1)
    float prm[232];          // [cam | R(24x9) | betas]; prm[229..231] unused
    float J[72];             // rest joints
    float Jrel[72];          // J_k - J_parent(k) (root: J_0)
    float best[232];
    float grad[232];
    alignas(16) float vp[16];
    float xf[16];
    float g[34];
    float ga[72];            // dL/d(tw_k) seeds of the chain joints that are fit targets (others stay 0; z stays 0)
    alignas(16) float h[16]; // dL/d(x_v) of the 5 face vertices (z stays 0)
    float rpart[17 * 5];     // per-fit-joint partials: loss, l2e, gs, gtx, gty
    float red[8];            // reduced: loss, l2e, gs, gtx, gty
    float adir[72];          // direct part of abar
    float dJ[72];
    alignas(16) float adj[96];   // [dvp(15) | dJ_total(72) | 0(9)]: what d(beta) contracts with fc[kFcC]
    float tgt_px[34];
    float best_l2e;
    int take_best;
    alignas(16) float Afin[26 * 12];   // reverse WALK: the total adjoints (Abar keeps the direct ones); slot 24 = zeros, 25 = spare
    uint32_t childw[25];               // children of joint k, 5 bits each, 24 = none (SmplHostImage::walk[288 + k])
};

// Per-lane registers that live across iterations: Adam moments of parameter `lane`, this lane's ROLE in the
// level-synchronous sweeps (float offsets into FitShared, looked up once), and the model constants this lane
// multiplies with in every iteration (loop-invariant: kept in registers instead of re-read from LDS).
struct FitLane {
    float m, v;
    int trainable;
    // Role in the level-synchronous sweeps, ONE word per level (lane = slot*12 + e handles element e of the slot-th joint of the
    // level): bits 0-4 joint k, 5-9 its parent, 10-14 / 15-19 / 20-24 its children (24 = absent: the all-zero slot), bit 25 =
    // takes part in the forward chain (k has a parent), bit 26 = takes part in the reverse sweep.  The LDS offsets are rebuilt
    // from it with a handful of integer operations that do not depend on data (round 3: the 120 precomputed 16-bit offsets of
    // rounds 1-2 cost 120 registers per lane -- 269 VGPRs; now 164-200, so a patch-convolution wave fits on the SIMD beside it).
    uint32_t lvl[kMaxLevels];
    uint32_t wlk[4];             // lanes 0..71: this lane's programs of the barrier-free tree walks (SmplHostImage::walk)
    uint32_t eic;                // e | (4 i) << 8 | (4 c) << 16 with e = lane % 12 = 4 i + c
    uint32_t role2;              // bits 0-9 / 10-19: P10 parent transform offsets (floats) of rotation entry / dJ entry `lane`;
                                 // bits 20-24: P6 source joint of fit joint `lane`
    // loop-invariant model constants
    float cJ[11];                // P1 lanes 0..143: bias + 10 shape coefficients of J / Jrel element; lanes 240..254 (chunk 15 of the
                                 //    face-vertex sums): template + 10 shape coefficients of coordinate lane & 15
    float cP[13];                // P1: posedirs coefficients of the face-vertex coordinate this lane accumulates
    float cQ[15];                // P10: posedirs row of this lane's rotation entry
    float cC[6];                 // P11: d(beta) coefficients
    float w2[2];                 // LBS weights of the face vertices used in P4
    float w5[5];                 // ... in P7: lanes 64..135 five vertices of one joint, lanes 136..255 w5[0..2] = three joints of one vertex
};

constexpr uint32_t kLvlChain = 1u << 25, kLvlReverse = 1u << 26;

#define SPSP_OFF(member) ((int)(offsetof(FitShared, member) / sizeof(float)))

SPSP_HD bool fit_trainable(uint32_t mask, int p) {
    if (p < kPrmR) return (mask >> 24) & 1u;
    if (p < kPrmBeta) return (mask >> ((p - kPrmR) / 9)) & 1u;
    return (mask >> 25) & 1u;
}

SPSP_HD float* fit_f(FitShared& s) { return reinterpret_cast<float*>(&s); }

// ---- one-time set-up -----------------------------------------------------------------------------------
SPSP_HD void fit_init(FitShared& s, FitLane& r, int lane, const float* fitc, const int* parents,
                      const int* level_joints, const int* level_start, const int* child_start,
                      const int* child_list, const float* R, const float* betas, const float* cam,
                      const float* joints2d, uint32_t param_mask, const uint32_t* walk = nullptr) {
    SPSP_UNROLL
    for (int j = 0; j < 4; ++j) r.wlk[j] = walk ? walk[(lane >> 2) * 4 + j] : 0xffffffffu;
    if (walk && lane < 25) s.childw[lane] = walk[256 + lane];
    for (int i = lane; i < 26 * 12; i += kFitLanes) s.Afin[i] = 0.f;
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
    if (lane < 4) s.unit[lane] = lane == 3 ? 1.f : 0.f;
    if (lane < 34) s.tgt_px[lane] = joints2d[lane];
    if (lane == 0) { s.best_l2e = INFINITY; s.take_best = 0; }
    if (lane < 72) s.ga[lane] = 0.f;
    if (lane < 16) s.h[lane] = 0.f;
    if (lane < 96) s.adj[lane] = 0.f;
    for (int i = lane; i < 25 * 12; i += kFitLanes) { s.Abar[i] = 0.f; s.W[i] = 0.f; }
    for (int i = lane; i < 25 * 16; i += kFitLanes) {       // static parts: row 3 of Mt columns, Tt[.][3] = (0,0,0,1)
        const int c = (i >> 2) & 3, m = i & 3;
        s.Tt[i] = ((c == 3 || i >= 24 * 16) && m == c) ? 1.f : 0.f;       // (slot 24 of both: the identity)
        s.Mt[i] = ((c == 3 || i >= 24 * 16) && m == c) ? 1.f : 0.f;
    }
    r.m = r.v = 0.f;
    r.trainable = lane < kNumPrm ? (int)fit_trainable(param_mask, lane) : 0;

    // ---- roles of this lane in the level sweeps (lane = slot*12 + e, e = i*4 + c) ----
    const int slot = lane / 12, e = lane % 12, i = e >> 2, c = e & 3;
    r.eic = (uint32_t)e | ((uint32_t)(i * 4) << 8) | ((uint32_t)(c * 4) << 16);
    SPSP_UNROLL
    for (int l = 0; l < kMaxLevels; ++l) {
        uint32_t q = 0;
        const int idx = level_start[l < 25 ? l : 25] + slot;
        if (l < 25 && slot < 5 && idx < level_start[l + 1]) {
            const int k = level_joints[idx];
            const int p = parents[k];
            q = (uint32_t)k | kLvlReverse;
            if (p >= 0) q |= ((uint32_t)p << 5) | kLvlChain;
            const int c0 = child_start[k], nc = child_start[k + 1] - c0;
            SPSP_UNROLL
            for (int n = 0; n < 3; ++n) q |= (uint32_t)(n < nc ? child_list[c0 + n] : 24) << (10 + 5 * n);
        }
        r.lvl[l] = q;
    }
    {
        const int src_tab[17] = SPSP_FIT_SRC;
        const uint32_t wp216 = lane < 216 && lane >= 9 ? SPSP_OFF(W) + parents[lane / 9] * 12 : SPSP_OFF(W);
        const uint32_t wp72 = lane < 72 && lane >= 3 ? SPSP_OFF(W) + parents[lane / 3] * 12 : SPSP_OFF(W);
        r.role2 = wp216 | (wp72 << 10) | ((uint32_t)(lane < 17 ? src_tab[lane] : 0) << 20);
    }

    // ---- loop-invariant constants (read from the global copy of the fit constants) ----
    SPSP_UNROLL
    for (int t = 0; t < 11; ++t) r.cJ[t] = 0.f;
    if (lane < 144) {
        const int base = lane < 72 ? kFcJbar : kFcJrel, o = lane < 72 ? lane : lane - 72;
        const int js = lane < 72 ? kFcJS : kFcJrel + 72;
        r.cJ[0] = fitc[base + o];
        SPSP_UNROLL
        for (int t = 0; t < 10; ++t) r.cJ[1 + t] = fitc[js + t * kJSPitch + o];
    }
    {
        const int q = lane >> 4, cc = lane & 15;
        SPSP_UNROLL
        for (int t = 0; t < 13; ++t) r.cP[t] = cc < 15 ? fitc[kFcP5 + (q * 13 + t) * kF5Pitch + cc] : 0.f;
        if (q == 15 && cc < 15) {       // (lanes 240..254: not among the J lanes 0..143)
            r.cJ[0] = fitc[kFcVt5 + cc];
            SPSP_UNROLL
            for (int t = 0; t < 10; ++t) r.cJ[1 + t] = fitc[kFcS5 + t * kF5Pitch + cc];
        }
        SPSP_UNROLL
        for (int t = 0; t < 6; ++t) r.cC[t] = cc < 10 ? fitc[kFcC + cc * kCPitch + q * 6 + t] : 0.f;
        // P4: lane = q*16 + (v*3+i), joints 2q, 2q+1
        const int v4 = (cc < 15 ? cc : 0) / 3;
        SPSP_UNROLL
        for (int t = 0; t < 2; ++t) r.w2[t] = (q < 12 && cc < 15) ? fitc[kFcW5 + v4 * 24 + q * 2 + t] : 0.f;
    }
    SPSP_UNROLL
    for (int t = 0; t < 15; ++t) r.cQ[t] = (lane >= 9 && lane < 216) ? fitc[kFcP5 + (lane - 9) * kF5Pitch + t] : 0.f;
    SPSP_UNROLL
    for (int v = 0; v < 5; ++v) r.w5[v] = (lane >= 64 && lane < 136) ? fitc[kFcW5 + v * 24 + (lane - 64) / 3] : 0.f;
    if (lane >= 136) {
        const int idx = lane - 136, q = idx / 15, o = idx % 15, v = o / 3;
        SPSP_UNROLL
        for (int t = 0; t < 3; ++t) r.w5[t] = fitc[kFcW5 + v * 24 + q * 3 + t];
    }
}

// ---- forward ---------------------------------------------------------------------------------------------
// P1: J (lanes 0..71) and J_k - J_parent (lanes 72..143) as affine functions of beta; partial sums of the 5 posed
//     face vertices over 16 chunks of 13 pose features (lane = q*16 + c, c = v*3+i < 15; chunk 15 also carries
//     template + shape term).  vp = vt5 + S5.beta + sum_j P5[j] (R[9+j] - delta_j): the delta part is folded into
//     the template constant at load; row 207 of P5 is a zero pad (its factor prm[kPrmR+216] = beta[0] is finite).
SPSP_HD void fit_p1(FitShared& s, const FitLane& r, int lane) {
    const float* beta = s.prm + kPrmBeta;
    float b[10];
    SPSP_UNROLL
    for (int i = 0; i < 10; ++i) b[i] = beta[i];
    if (lane < 144) {
        float acc = r.cJ[0];
        SPSP_UNROLL
        for (int i = 0; i < 10; ++i) acc = fmaf(r.cJ[1 + i], b[i], acc);
        if (lane < 72) s.J[lane] = acc; else s.Jrel[lane - 72] = acc;
    }
    const int q = lane >> 4;
    float acc = 0.f;
    if (lane >= 240) {
        acc = r.cJ[0];
        SPSP_UNROLL
        for (int i = 0; i < 10; ++i) acc = fmaf(r.cJ[1 + i], b[i], acc);
    }
    const float* Rv = s.prm + kPrmR + 9 + q * 13;
    SPSP_UNROLL
    for (int t = 0; t < 13; ++t) acc = fmaf(r.cP[t], Rv[t], acc);
    s.part[(lane & 15) * 16 + q] = acc;                      // [coordinate][chunk]: the finishing sum reads 4 x b128
}

// P2 [all lanes]: finish vp (lanes 0..14); root transform W_0 = [G | J_0] (lanes 16..27); the per-joint operands of
//     the two level sweeps, 16 bytes each so that a sweep step is two ds_read_b128 per term:
//     lanes 64..159:  Mt[k][c] = (R_k[0][c], R_k[1][c], R_k[2][c], 0) for c < 3, (Jrel_k, 1) for c = 3
//     lanes 160..231: Tt[k][c] = (R_k[c][0], R_k[c][1], R_k[c][2], Jrel_k[c]),  c < 3
SPSP_HD void fit_p2(FitShared& s, int lane) {
    if (lane < 15) {
        const float* p = s.part + lane * 16;
        const F4 a = ld4(p), b = ld4(p + 4), c = ld4(p + 8), d = ld4(p + 12);
        s.vp[lane] = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
    }
    if (lane >= 16 && lane < 28) {
        const int e = lane - 16, i = e >> 2, c = e & 3;
        s.W[e] = (c < 3) ? s.prm[kPrmR + i * 3 + c] : s.J[i];
    }
    if (lane >= 64 && lane < 160) {
        const int it = lane - 64, k = it >> 2, c = it & 3;
        const float* R = s.prm + kPrmR + k * 9;
        if (c < 3) st4(s.Mt + it * 4, R[c], R[3 + c], R[6 + c], 0.f);
        else st4(s.Mt + it * 4, s.Jrel[k * 3], s.Jrel[k * 3 + 1], s.Jrel[k * 3 + 2], 1.f);
    }
    if (lane >= 160 && lane < 232) {
        const int it = lane - 160, k = it / 3, c = it % 3;
        const float* R = s.prm + kPrmR + k * 9 + c * 3;
        st4(s.Tt + k * 16 + c * 4, R[0], R[1], R[2], s.Jrel[k * 3 + c]);
    }
}

// P3[l]: kinematic chain, level l >= 1, in homogeneous form: W_k = W_p * [[R_k, Jrel_k], [0, 1]];
//        element (i,c) = <row i of W_p (4), column c of the operand (4)>: two ds_read_b128 + 4 FMA.
SPSP_HD void fit_chain_level(FitShared& s, const FitLane& r, uint32_t q) {      // q = the lane's role word of the level (FitLane::lvl)
    SPSP_OPAQUE(q);
    if (!(q & kLvlChain)) return;
    const int k = q & 31, p = (q >> 5) & 31, e = r.eic & 255, i4 = (r.eic >> 8) & 255, c4 = (r.eic >> 16) & 255;
    s.W[k * 12 + e] = dot4(ld4(s.W + p * 12 + i4), ld4(s.Mt + k * 16 + c4), 0.f);
}

// P3 as a WALK (round 3).  A quad of lanes carries ONE ROW of a world transform (lane & 3 = column) from the root down the path
// to a leaf: W_k[i][c] = <row i of W_parent, column c of [[R_k, Jrel_k], [0, 1]]> with the row's four entries fetched from the
// quad's lanes by DPP quad_perm -- no LDS hand-off between the tree levels, whose eight dependent round trips (~240 cycles
// each) were the chain's cost; the operands (Mt, P2) are all there when the walk starts.  Every joint on the way is stored (the
// rows shared by several leaves more than once, with the same value).  Same operations in the same order as the level sweep:
// bit-identical (tests/emul runs both).  Paths are padded with joint 24 = the identity (Mt[24]; W[24] takes the stores).
#if defined(__HIPCC__)
template <int M>
__device__ __forceinline__ float fit_quad_bcast(float v) {      // the value lane (lane & ~3) + M of this quad holds
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), M * 0x55, 0xf, 0xf, false));
}
#define SPSP_QDOT(v, T, init) fmaf(fit_quad_bcast<3>(v), (T).w, fmaf(fit_quad_bcast<2>(v), (T).z, fmaf(fit_quad_bcast<1>(v), (T).y, fmaf(fit_quad_bcast<0>(v), (T).x, init))))
__device__ __forceinline__ void fit_chain_walk(FitShared& s, const FitLane& r, int lane) {
    uint32_t lo = r.wlk[0], hi = r.wlk[1];
    SPSP_OPAQUE(lo);
    SPSP_OPAQUE(hi);
    if ((lo & 31u) == 31u) return;                           // idle quad
    const int i = (lane >> 2) % 3, c = lane & 3;
    float v = s.W[i * 4 + c];                                // the root's row (P2)
    int a[8];
    F4 m[8];                                                 // all eight operands first: nothing below waits for LDS again
    SPSP_UNROLL
    for (int j = 0; j < 8; ++j) {
        a[j] = (int)((j < 6 ? lo >> (5 * j) : hi >> (5 * (j - 6))) & 31u);
        m[j] = ld4(s.Mt + a[j] * 16 + c * 4);
    }
    SPSP_UNROLL
    for (int j = 0; j < 8; ++j) {
        v = SPSP_QDOT(v, m[j], 0.f);
        s.W[a[j] * 12 + i * 4 + c] = v;
    }
}

// P9 as two WALKS.  Abar (P7) keeps the DIRECT adjoints, the totals go to Afin.
//   a: quad 3 f + i starts with row i of leaf f's adjoint and climbs while the subtree is a path:
//      row <- direct row of the parent + row x [[R_c^T, 0], [Jrel_c^T, 1]]  (the single-child case of the level sweep).
//   b: quads 0..2 climb from the deepest joint with a BRANCHING subtree to the root -- in the SMPL tree 9, 6, 3, 0 with the
//      children (12, 13, 14), (9), (6), (1, 2, 3): the structure is compiled in (kWalkB / kWalkBChildren; a model with another
//      tree keeps the level sweeps, smpl_host.hpp), so every operand that does not come out of the walk itself is fetched up
//      front.  At every joint the children are taken in the level sweep's order, the one just left from registers, the others
//      (finished in a; same wavefront, LDS is in order) from Afin.  Same operations in the same order as the level sweep.
__device__ __forceinline__ void fit_reverse_walk_a(FitShared& s, const FitLane& r, int lane) {
    uint32_t p = r.wlk[2];
    SPSP_OPAQUE(p);
    if ((p & 31u) == 31u) return;
    const int i = (lane >> 2) % 3, c = lane & 3;
    int d = (int)(p & 31u);
    float v = s.Abar[d * 12 + i * 4 + c];
    int e[6];
    float dir[6];
    F4 T[6];
    SPSP_UNROLL
    for (int j = 1; j < 6; ++j) {
        e[j] = (int)((p >> (5 * j)) & 31u);                  // 24 = padding: zero direct row, identity operand, store to the spare slot
        dir[j] = s.Abar[e[j] * 12 + i * 4 + c];
        T[j] = ld4(s.Tt + (e[j] == 24 ? 24 : d) * 16 + c * 4);
        d = e[j] == 24 ? d : e[j];
    }
    s.Afin[(int)(p & 31u) * 12 + i * 4 + c] = v;
    SPSP_UNROLL
    for (int j = 1; j < 6; ++j) {
        v = SPSP_QDOT(v, T[j], dir[j]);
        s.Afin[(e[j] == 24 ? 25 : e[j]) * 12 + i * 4 + c] = v;
    }
}

__device__ __forceinline__ void fit_reverse_walk_b(FitShared& s, const FitLane& r, int lane) {
    uint32_t p = r.wlk[3];
    SPSP_OPAQUE(p);
    if ((p & 31u) == 31u) return;
    const int i = (lane >> 2) % 3, c = lane & 3;
    float dir[kWalkNB];
    F4 T[kWalkNB][3], fin[kWalkNB][3];
    SPSP_UNROLL
    for (int j = 0; j < kWalkNB; ++j) {
        dir[j] = s.Abar[kWalkB[j] * 12 + i * 4 + c];
        SPSP_UNROLL
        for (int n = 0; n < 3; ++n) {
            const int ch = kWalkBChildren[j][n];
            if (ch == 24) continue;
            T[j][n] = ld4(s.Tt + ch * 16 + c * 4);
            if (j == 0 || ch != kWalkB[j - 1]) fin[j][n] = ld4(s.Afin + ch * 12 + i * 4);
        }
    }
    float prev = 0.f;
    SPSP_UNROLL
    for (int j = 0; j < kWalkNB; ++j) {
        float acc = dir[j];
        SPSP_UNROLL
        for (int n = 0; n < 3; ++n) {
            const int ch = kWalkBChildren[j][n];
            if (ch == 24) continue;                          // (the level sweep multiplies a zero row here)
            if (j > 0 && ch == kWalkB[j - 1]) acc = SPSP_QDOT(prev, T[j][n], acc);
            else acc = dot4(fin[j][n], T[j][n], acc);
        }
        s.Afin[kWalkB[j] * 12 + i * 4 + c] = acc;
        prev = acc;
    }
}
#undef SPSP_QDOT
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
// host twins (tests/emul): the same walks, a quad at a time
inline void fit_chain_walk_host(FitShared& s, const uint32_t* walk) {
    for (int q = 0; q < 64; ++q) {
        const uint32_t lo = walk[q * 4], hi = walk[q * 4 + 1];
        if ((lo & 31u) == 31u) continue;
        const int i = q % 3;
        F4 row = ld4(s.W + i * 4);
        for (int j = 0; j < 8; ++j) {
            const int a = (int)((j < 6 ? lo >> (5 * j) : hi >> (5 * (j - 6))) & 31u);
            const float* M = s.Mt + a * 16;
            row = F4{dot4(row, ld4(M), 0.f), dot4(row, ld4(M + 4), 0.f), dot4(row, ld4(M + 8), 0.f), dot4(row, ld4(M + 12), 0.f)};
            st4(s.W + a * 12 + i * 4, row.x, row.y, row.z, row.w);
        }
    }
}
inline void fit_reverse_walk_host(FitShared& s, const uint32_t* walk, int nb) {
    for (int q = 0; q < 64; ++q) {
        const uint32_t p = walk[q * 4 + 2];
        if ((p & 31u) == 31u) continue;
        const int i = q % 3;
        int d = (int)(p & 31u);
        F4 v = ld4(s.Abar + d * 12 + i * 4);
        st4(s.Afin + d * 12 + i * 4, v.x, v.y, v.z, v.w);
        for (int j = 1; j < 6; ++j) {
            const int e = (int)((p >> (5 * j)) & 31u);
            const F4 dir = ld4(s.Abar + e * 12 + i * 4);
            const float* T = s.Tt + (e == 24 ? 24 : d) * 16;
            v = F4{dot4(v, ld4(T), dir.x), dot4(v, ld4(T + 4), dir.y), dot4(v, ld4(T + 8), dir.z), dot4(v, ld4(T + 12), dir.w)};
            st4(s.Afin + (e == 24 ? 25 : e) * 12 + i * 4, v.x, v.y, v.z, v.w);
            d = e == 24 ? d : e;
        }
    }
    for (int q = 0; q < 3; ++q) {
        const uint32_t p = walk[q * 4 + 3];
        if ((p & 31u) == 31u) continue;
        const int i = q % 3;
        F4 prev = F4{0.f, 0.f, 0.f, 0.f};
        int pb = 31;
        for (int j = 0; j < nb; ++j) {
            const int b = (int)((p >> (5 * j)) & 31u);
            F4 acc = ld4(s.Abar + b * 12 + i * 4);
            const uint32_t cw = s.childw[b];
            for (int n = 0; n < 3; ++n) {
                const int ch = (int)((cw >> (5 * n)) & 31u);
                if (ch == 24) continue;
                const F4 row = ch == pb ? prev : ld4(s.Afin + ch * 12 + i * 4);
                const float* T = s.Tt + ch * 16;
                acc = F4{dot4(row, ld4(T), acc.x), dot4(row, ld4(T + 4), acc.y), dot4(row, ld4(T + 8), acc.z), dot4(row, ld4(T + 12), acc.w)};
            }
            st4(s.Afin + b * 12 + i * 4, acc.x, acc.y, acc.z, acc.w);
            prev = acc;
            pb = b;
        }
    }
}
#endif

// P4: face vertices x_v = sum_k w_vk (Rw_k (vp_v - J_k) + tw_k); lane = q*16 + (v*3+i), q < 12 = pair of joints.
SPSP_HD void fit_p4(FitShared& s, const FitLane& r, int lane) {
    const int q = lane >> 4, o = lane & 15;
    if (q >= 12) return;
    float acc = 0.f;
    if (o < 15) {
        const int v = o / 3, i = o % 3;
        const float v0 = s.vp[v * 3], v1 = s.vp[v * 3 + 1], v2 = s.vp[v * 3 + 2];
        SPSP_UNROLL
        for (int t = 0; t < 2; ++t) {
            const int k = q * 2 + t;
            const float* w = s.W + k * 12 + i * 4;
            const float tt = fmaf(w[0], v0 - s.J[k * 3], fmaf(w[1], v1 - s.J[k * 3 + 1], fmaf(w[2], v2 - s.J[k * 3 + 2], w[3])));
            acc = fmaf(r.w2[t], tt, acc);
        }
    }
    s.part[o * 16 + q] = acc;
}

SPSP_HD void fit_p5(FitShared& s, int lane) {
    if (lane < 15) {
        const float* p = s.part + lane * 16;
        const F4 a = ld4(p), b = ld4(p + 4), c = ld4(p + 8);
        s.xf[lane] = ((a.x + a.y) + (a.z + a.w)) + (((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)));
    }
}

// P6: project the 17 fit joints, residuals, per-joint loss / l2e / camera-gradient partials, dL/dX.
//     pixels = (s (X + t) + 1) * img_wh/2  (cam_utils.py:5-26, joints2d_utils.py:5-10);
//     n = 2 px / loss_wh - 1 on both sides, L = mean(d^2) * w + lv  (multi_task_loss.py:97-113).
// from_partials: the face-vertex coordinates are summed here from P4's partials (P5's sum, same order) -- one LDS hand-off less.
SPSP_HD void fit_p6(FitShared& s, const FitLane& r, int lane, const FitParams& P, bool from_partials = false) {
    if (lane >= 17) return;
    uint32_t role2 = r.role2;
    SPSP_OPAQUE(role2);
    const int src = (role2 >> 20) & 31;
    float X[2];
    if (src < 24) { X[0] = s.W[src * 12 + 3]; X[1] = s.W[src * 12 + 7]; }
    else if (from_partials) {
        SPSP_UNROLL
        for (int c = 0; c < 2; ++c) {
            const float* p = s.part + ((src - 24) * 3 + c) * 16;
            const F4 a = ld4(p), b = ld4(p + 4), cc = ld4(p + 8);
            X[c] = ((a.x + a.y) + (a.z + a.w)) + (((b.x + b.y) + (b.z + b.w)) + ((cc.x + cc.y) + (cc.z + cc.w)));
        }
    } else { X[0] = s.xf[(src - 24) * 3]; X[1] = s.xf[(src - 24) * 3 + 1]; }
    const float sc = s.prm[0];
    // x / 2^k = x * 2^-k exactly: one multiplication in place of the division sequence (loss_wh = 256 in every reference fit)
    uint32_t lw_bits;
    memcpy(&lw_bits, &P.loss_wh, 4);
    const bool lw_pow2 = (lw_bits & 0x807fffffu) == 0 && lw_bits >= 0x20000000u && lw_bits <= 0x5f000000u;
    const float lw_inv = 1.0f / P.loss_wh;
    const float half = P.img_wh / 2.0f;
    const float dn_du = (2.0f * half) / P.loss_wh;           // d n / d u
    const float coef = (2.0f * P.weight) / 34.0f;            // d L / d n = coef * (n - n*)
    float loss = 0.f, l2 = 0.f, gs = 0.f, gt[2];
    SPSP_UNROLL
    for (int c = 0; c < 2; ++c) {
        const float t = s.prm[1 + c];
        const float u = sc * (X[c] + t);
        const float px = (u + 1.f) * half;
        const float y = s.tgt_px[lane * 2 + c];
        float n, ns;
        if (lw_pow2) { n = (2.0f * px) * lw_inv - 1.0f; ns = (2.0f * y) * lw_inv - 1.0f; }
        else { n = (2.0f * px) / P.loss_wh - 1.0f; ns = (2.0f * y) / P.loss_wh - 1.0f; }
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
// P7: reductions of P6 (lanes 0..4); lanes 64..135: direct adjoints of item (k, i): row i of [B_k | a_k] and the
//     direct dJ[k][i]; lanes 136..255: dvp partials over 3 joints each.
SPSP_HD void fit_p7(FitShared& s, const FitLane& r, int lane) {
    if (lane < 5) {
        float acc = 0.f;
        SPSP_UNROLL
        for (int i = 0; i < 17; ++i) acc += s.rpart[i * 5 + lane];
        s.red[lane] = acc;
    }
    if (lane >= 64 && lane < 136) {
        const int it = lane - 64, k = it / 3, i = it % 3;
        float a = s.ga[it];
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, dj = 0.f;
        const float j0 = s.J[k * 3], j1 = s.J[k * 3 + 1], j2 = s.J[k * 3 + 2];
        const float wki = s.W[k * 12 + i], wki4 = s.W[k * 12 + 4 + i];
        SPSP_UNROLL
        for (int v = 0; v < 5; ++v) {
            const float w = r.w5[v];
            const float wh = w * s.h[v * 3 + i];
            a += wh;
            b0 = fmaf(wh, s.vp[v * 3] - j0, b0);
            b1 = fmaf(wh, s.vp[v * 3 + 1] - j1, b1);
            b2 = fmaf(wh, s.vp[v * 3 + 2] - j2, b2);
            // direct dJ[k][l = i] = -sum_v w_vk sum_i' Rw_k[i'][l] h_v[i']   (h_v[2] == 0)
            dj = fmaf(w, fmaf(wki, s.h[v * 3], wki4 * s.h[v * 3 + 1]), dj);
        }
        s.adir[it] = a;
        st4(s.Abar + k * 12 + i * 4, b0, b1, b2, a);
        s.dJ[it] = -dj;
    }
    if (lane >= 136) {   // dvp[v][l] partial over joints 3q..3q+2: sum_k w_vk sum_i Rw_k[i][l] h_v[i]
        const int idx = lane - 136, q = idx / 15, o = idx % 15;
        const int v = o / 3, l = o % 3;
        const float h0 = s.h[v * 3], h1 = s.h[v * 3 + 1];
        float acc = 0.f;
        SPSP_UNROLL
        for (int t = 0; t < 3; ++t) {
            const int k = q * 3 + t;
            acc = fmaf(r.w5[t], fmaf(s.W[k * 12 + l], h0, s.W[k * 12 + 4 + l] * h1), acc);
        }
        s.part2[o * 16 + q] = acc;
    }
}

// P8: finish dvp; lane 63 turns the reductions into loss / l2e, history and the best-iterate decision.
SPSP_HD void fit_p8(FitShared& s, int lane, const FitParams& P, int it, float* loss_hist, float* l2e_hist) {
    if (lane < 15) {
        const F4 a = ld4(s.part2 + lane * 16), b = ld4(s.part2 + lane * 16 + 4);
        const float acc = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
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

// P9[l]: reverse sweep in homogeneous form.  With Abar_k = [Bbar_k | abar_k] (3x4):
//   Abar_p += sum over children ch of  Abar_ch * [[R_ch^T, 0], [Jrel_ch^T, 1]]
//   element (i,c) += <row i of Abar_ch (4), Tt[ch][c] (4)>: two ds_read_b128 + 4 FMA per child.
//   Absent children point at the all-zero slot 24 of Abar.
SPSP_HD void fit_reverse_level(FitShared& s, const FitLane& r, uint32_t q) {
    SPSP_OPAQUE(q);
    if (!(q & kLvlReverse)) return;
    const int k = q & 31, e = r.eic & 255, i4 = (r.eic >> 8) & 255, c4 = (r.eic >> 16) & 255;
    float acc = s.Abar[k * 12 + e];
    SPSP_UNROLL
    for (int n = 0; n < 3; ++n) {
        const int ch = (q >> (10 + 5 * n)) & 31;
        acc = dot4(ld4(s.Abar + ch * 12 + i4), ld4(s.Tt + ch * 16 + c4), acc);
    }
    s.Abar[k * 12 + e] = acc;
}

// P10: rotation gradients incl. the pose-blend-shape path through the face vertices (lanes 0..215) and the
//      total dJ (lanes 0..71, after their rotation item).
SPSP_HD void fit_p10(FitShared& s, const FitLane& r, int lane, bool walked = false) {
    const float* const AB = walked ? s.Afin : s.Abar;        // where the reverse pass left the total adjoints
    float* f = fit_f(s);
    uint32_t role2 = r.role2;
    SPSP_OPAQUE(role2);
    if (lane < 216) {                                        // item (k, l, c): dR_k[l][c]
        const int k = lane / 9, e = lane % 9, l = e / 3, c = e % 3;
        float acc;
        if (k == 0) acc = AB[l * 4 + c];
        else {
            const float* Wp = f + (role2 & 1023);
            const float* A = AB + k * 12 + c;
            acc = Wp[l] * A[0];
            acc = fmaf(Wp[4 + l], A[4], acc);
            acc = fmaf(Wp[8 + l], A[8], acc);
            const F4 d0 = ld4(s.dvp), d1 = ld4(s.dvp + 4), d2 = ld4(s.dvp + 8), d3 = ld4(s.dvp + 12);
            const float dv[16] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w, d3.x, d3.y, d3.z, d3.w};
            SPSP_UNROLL
            for (int o = 0; o < 15; ++o) acc = fmaf(r.cQ[o], dv[o], acc);
        }
        s.grad[kPrmR + lane] = acc;
    }
    if (lane < 72) {                                         // dJ_total[k][l]
        const int k = lane / 3, l = lane % 3;
        const float* A = AB + k * 12 + 3;                // abar_k[i] = A[4 i]
        float acc = s.dJ[lane];
        if (k == 0) acc += A[4 * l];
        else {
            const float* Wp = f + ((role2 >> 10) & 1023);
            acc = fmaf(Wp[l], A[0], fmaf(Wp[4 + l], A[4], fmaf(Wp[8 + l], A[8], acc)));
        }
        const float* Wk = s.W + k * 12;
        const float c0 = A[0] - s.adir[k * 3], c1 = A[4] - s.adir[k * 3 + 1], c2 = A[8] - s.adir[k * 3 + 2];
        acc -= fmaf(Wk[l], c0, fmaf(Wk[4 + l], c1, Wk[8 + l] * c2));
        s.adj[15 + lane] = acc;
    }
}

// P11: d beta partials over 16 chunks of 6 of the adjoint vector adj = [dvp(15) | dJ_total(72) | 0(9)];
//      lane = q*16 + i, i < 10.
SPSP_HD void fit_p11(FitShared& s, const FitLane& r, int lane) {
    const int q = lane >> 4;
    const float* A = s.adj + q * 6;
    float acc = 0.f;
    SPSP_UNROLL
    for (int t = 0; t < 6; ++t) acc = fmaf(r.cC[t], A[t], acc);
    s.part[(lane & 15) * 16 + q] = acc;
}

// P12: best-iterate snapshot (pre-step parameters) and the torch.optim.Adam step (single-tensor path), one
//   parameter per lane:
//   m = lerp(m, g, 1-b1); v = v*b2 + (1-b2)*g*g; denom = sqrt(v)/sqrt(1-b2^t) + eps; p += (-lr/(1-b1^t) * m) / denom
SPSP_HD void fit_p12(FitShared& s, FitLane& r, int lane, float w1, float b2, float w2, float eps, float step_size,
                     float bc2_sqrt) {
    const int p = lane;
    if (p >= kNumPrm) return;
    const float cur = s.prm[p];
    if (s.take_best) s.best[p] = cur;
    if (!r.trainable) return;
    float g;
    if (p < kPrmR) g = s.red[2 + p];
    else if (p < kPrmBeta) g = s.grad[p];
    else {
        const float* pp = s.part + (p - kPrmBeta) * 16;
        const F4 a = ld4(pp), b = ld4(pp + 4), c = ld4(pp + 8), d = ld4(pp + 12);
        g = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
    }
    r.m = r.m + w1 * (g - r.m);
    r.v = r.v * b2 + (w2 * g) * g;
    const float denom = sqrtf(r.v) / bc2_sqrt + eps;
    s.prm[p] = cur + ((-step_size) * r.m) / denom;
}

// ==== round 4: the SAME arithmetic on a shorter critical path (schedule "v2"; fit.hip, fit_kernel2) ==========================
// The phases above run P12 | P1 | P2 | chain | P4 | P6 | P7 | P8 + reverse | P10 | P11 | with a workgroup barrier at every bar: ten
// LDS hand-offs per Adam iteration, 8.4 k cycles (stamps of round 3).  Nothing below changes a formula or the order of the
// operations that produce a value -- every iterate stays BIT-IDENTICAL to the phases above (tests/emul runs both side by side,
// the GPU test compares all iterates) -- only WHO computes it WHEN:
//   * a parameter belongs to the lane that computes its gradient (rotation entry e: lane e, camera: lanes 216..218, betas: lanes
//     219..228), so the Adam step of the 216 rotation entries follows their gradient in the same lane, in registers: no
//     gradient array hand-off, no barrier between P10 and P12;
//   * the Adam step PUBLISHES the new value in every form the next iteration reads (prm, the chain operand Mt, the reverse
//     operand Tt), P1 publishes J_0 and J_k - J_parent into W_0 / Mt / Tt: P2 shrinks to the 15 face-vertex sums, which wave 1
//     finishes while wave 0 walks the chain;
//   * P8 (sums of the dvp partials, loss history, best-iterate decision) runs on wave 1 while wave 0 walks the tree back;
//   * d(beta) (P11) runs inside wave 3 (4 chunks per lane instead of 1, the 16 partial sums and their final tree unchanged)
//     in front of that wave's Adam steps, while the other waves step theirs: one workgroup barrier and one Adam latency less.
// Eight barriers per iteration.
struct alignas(16) Fit2Shared {
    alignas(16) float cC[10 * 96];     // d(beta) coefficients, row i = beta_i over adj[0..95] (fitc[kFcC], pitch 96: ds_read_b128)
    alignas(16) float beta4[12];       // the betas again, 16-byte aligned: P1 reads them with three ds_read_b128 instead of ten b32
    alignas(16) float pf[16 * 16];     // the pose feature entries R[1:] in P1's chunks of 13, one chunk per 16 floats (four b128)
};

SPSP_HD int fit2_owned_param(int lane) {
    return lane < 216 ? lane + kPrmR : (lane < 219 ? lane - 216 : (lane < kNumPrm ? lane : -1));
}

// the forms of rotation entry p (kPrmR <= p < kPrmBeta) the sweeps read.  NOT W_0: the rotation gradients of this very phase
// still read the root transform; fit2_p1 copies G there behind the barrier.
SPSP_HD void fit2_publish_rotation(FitShared& s, Fit2Shared& x, int p, float v) {
    const int e = p - kPrmR, k = e / 9, rc = e - 9 * k, r = rc / 3, c = rc - 3 * r;
    s.Mt[k * 16 + c * 4 + r] = v;          // Mt[k][c] = (R[0][c], R[1][c], R[2][c], 0)
    s.Tt[k * 16 + r * 4 + c] = v;          // Tt[k][r] = (R[r][0], R[r][1], R[r][2], Jrel[r])
    if (e >= 9) { const int j = e - 9, q = j / 13; x.pf[q * 16 + (j - 13 * q)] = v; }
}

// before fit2_init (a barrier between them): the pads of pf (chunk 15's 13th feature is P5's zero row)
SPSP_HD void fit2_init0(Fit2Shared& x, int lane) { x.pf[lane] = 0.f; }
// after fit_init (and a barrier): ownership, the d(beta) coefficients in LDS, the operands of the initial rotations
SPSP_HD void fit2_init(FitShared& s, Fit2Shared& x, FitLane& r, int lane, const float* fitc, uint32_t param_mask) {
    const int p = fit2_owned_param(lane);
    r.trainable = p >= 0 ? (int)fit_trainable(param_mask, p) : 0;
    for (int i = lane; i < 10 * 96; i += kFitLanes) x.cC[i] = fitc[kFcC + (i / 96) * kCPitch + (i % 96)];
    if (lane < 12) x.beta4[lane] = lane < 10 ? s.prm[kPrmBeta + lane] : 0.f;
    if (lane < 216) fit2_publish_rotation(s, x, lane + kPrmR, s.prm[lane + kPrmR]);
}

// P1 of v2: fit_p1 + the operands that depend on J (P2's share of them) + the root rotation
SPSP_HD void fit2_p1(FitShared& s, const Fit2Shared& x, const FitLane& r, int lane) {
    const F4 b0 = ld4(x.beta4), b1 = ld4(x.beta4 + 4), b2 = ld4(x.beta4 + 8);
    const float b[10] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y};
    if (lane < 144) {
        float acc = r.cJ[0];
        SPSP_UNROLL
        for (int i = 0; i < 10; ++i) acc = fmaf(r.cJ[1 + i], b[i], acc);
        if (lane < 72) {
            s.J[lane] = acc;
            if (lane < 3) s.W[lane * 4 + 3] = acc;                // W_0 = [G | J_0]
        } else {
            const int o = lane - 72, k = o / 3, i = o - 3 * k;
            s.Jrel[o] = acc;
            s.Mt[k * 16 + 12 + i] = acc;                          // Mt[k][3] = (Jrel_k, 1)
            s.Tt[k * 16 + i * 4 + 3] = acc;                       // Tt[k][i] = (row i of R_k, Jrel_k[i])
        }
    } else if (lane < 153) {                                      // W_0 = [G | J_0]: the rotation the last step left in prm
        const int e = lane - 144;
        s.W[(e / 3) * 4 + e % 3] = s.prm[kPrmR + e];
    }
    const int q = lane >> 4;
    float acc = 0.f;
    if (lane >= 240) {
        acc = r.cJ[0];
        SPSP_UNROLL
        for (int i = 0; i < 10; ++i) acc = fmaf(r.cJ[1 + i], b[i], acc);
    }
    const F4 r0 = ld4(x.pf + q * 16), r1 = ld4(x.pf + q * 16 + 4), r2 = ld4(x.pf + q * 16 + 8), r3 = ld4(x.pf + q * 16 + 12);
    const float Rv[13] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x};
    SPSP_UNROLL
    for (int t = 0; t < 13; ++t) acc = fmaf(r.cP[t], Rv[t], acc);
    s.part[(lane & 15) * 16 + q] = acc;
}

// what is left of P2: the 15 face-vertex coordinates from P1's partial sums (lanes 64..78: wave 1, beside wave 0's chain walk)
SPSP_HD void fit2_vp(FitShared& s, int lane) {
    const int l = lane - 64;
    if (l < 0 || l >= 15) return;
    const float* p = s.part + l * 16;
    const F4 a = ld4(p), b = ld4(p + 4), c = ld4(p + 8), d = ld4(p + 12);
    s.vp[l] = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
}

// torch.optim.Adam step of the lane's own parameter p from gradient g (fit_p12's statements), published where it is read
SPSP_HD void fit2_adam(FitShared& s, Fit2Shared& x, FitLane& r, int p, float g, float w1, float b2, float w2, float eps, float step_size,
                       float bc2_sqrt, float inv_bc2, float* param_hist_row) {
    const float cur = s.prm[p];
    if (param_hist_row) param_hist_row[p] = cur;                  // the iterate this iteration evaluated
    if (s.take_best) s.best[p] = cur;
    if (!r.trainable) return;
    r.m = r.m + w1 * (g - r.m);
    r.v = r.v * b2 + (w2 * g) * g;
#if defined(__HIP_DEVICE_COMPILE__)
    (void)bc2_sqrt;
    const float denom = fmaf(fit_sqrt_1ulp(r.v), inv_bc2, eps);  // (a denormal v reads as 0: denom = eps, as sqrt(v) / bc2 + eps rounds to)
    const float nv = fmaf((-step_size) * r.m, fit_rcp_1ulp(denom), cur);
#else
    (void)inv_bc2;
    const float denom = sqrtf(r.v) / bc2_sqrt + eps;
    const float nv = cur + ((-step_size) * r.m) / denom;
#endif
    s.prm[p] = nv;
    if (p >= kPrmBeta) x.beta4[p - kPrmBeta] = nv;
    else if (p >= kPrmR) fit2_publish_rotation(s, x, p, nv);
}

// P6 of v2: TWO lanes per fit joint (lane = 2 joint + coordinate; 34 lanes) -- the 17-lane form above spends ~1000 cycles of
// wave 0 on two coordinates one after the other.  The sums that run over both coordinates (loss, l2, d loss / d scale) are the
// SAME fma chains: the x lane's link goes to the y lane (its neighbour: DPP row_shr 1 on the GPU, the caller's array on the host),
// which finishes the chain and writes the joint's partials.  fit2_p6a: everything up to this lane's links; fit2_p6b: the rest.
struct Fit2P6 { float loss, l2, gs; };
SPSP_HD int fit2_src(int joint) {                                // SPSP_FIT_SRC, five bits per fit joint
    constexpr int t[17] = SPSP_FIT_SRC;
    constexpr uint64_t lo = (uint64_t)t[0] | (uint64_t)t[1] << 5 | (uint64_t)t[2] << 10 | (uint64_t)t[3] << 15 | (uint64_t)t[4] << 20 |
                            (uint64_t)t[5] << 25 | (uint64_t)t[6] << 30 | (uint64_t)t[7] << 35 | (uint64_t)t[8] << 40 |
                            (uint64_t)t[9] << 45 | (uint64_t)t[10] << 50 | (uint64_t)t[11] << 55;
    constexpr uint64_t hi = (uint64_t)t[12] | (uint64_t)t[13] << 5 | (uint64_t)t[14] << 10 | (uint64_t)t[15] << 15 | (uint64_t)t[16] << 20;
    return (int)((joint < 12 ? lo >> (5 * joint) : hi >> (5 * (joint - 12))) & 31u);
}
// lane < 34.  Returns this lane's links of the three chains (started at 0); d, dp, gu, Xt are handed to fit2_p6b.
SPSP_HD Fit2P6 fit2_p6a(FitShared& s, int lane, const FitParams& P, float& d_out, float& dp_out, float& gu_out, float& xt_out) {
    const int joint = lane >> 1, c = lane & 1;
    const int src = fit2_src(joint);
    float X;
    if (src < 24) X = s.W[src * 12 + 3 + 4 * c];
    else {
        const float* p = s.part + ((src - 24) * 3 + c) * 16;
        const F4 a = ld4(p), b = ld4(p + 4), cc = ld4(p + 8);
        X = ((a.x + a.y) + (a.z + a.w)) + (((b.x + b.y) + (b.z + b.w)) + ((cc.x + cc.y) + (cc.z + cc.w)));
    }
    const float sc = s.prm[0];
    uint32_t lw_bits;
    memcpy(&lw_bits, &P.loss_wh, 4);
    const bool lw_pow2 = (lw_bits & 0x807fffffu) == 0 && lw_bits >= 0x20000000u && lw_bits <= 0x5f000000u;
    const float lw_inv = 1.0f / P.loss_wh;
    const float half = P.img_wh / 2.0f;
    const float dn_du = (2.0f * half) / P.loss_wh;
    const float coef = (2.0f * P.weight) / 34.0f;
    const float t = s.prm[1 + c];
    const float u = sc * (X + t);
    const float px = (u + 1.f) * half;
    const float y = s.tgt_px[joint * 2 + c];
    float n, ns;
#if defined(__HIP_DEVICE_COMPILE__)
    // round 5: x * (1 / loss_wh) for every loss_wh (exact for the powers of two the reference uses; else within an ulp of the
    // division) -- hipcc if-converted the two forms and ran the two IEEE divisions in every iteration
    (void)lw_pow2;
    n = (2.0f * px) * lw_inv - 1.0f; ns = (2.0f * y) * lw_inv - 1.0f;
#else
    if (lw_pow2) { n = (2.0f * px) * lw_inv - 1.0f; ns = (2.0f * y) * lw_inv - 1.0f; }
    else { n = (2.0f * px) / P.loss_wh - 1.0f; ns = (2.0f * y) / P.loss_wh - 1.0f; }
#endif
    const float d = n - ns;
    const float dp = px - y;
    const float gu = coef * d * dn_du;                       // dL/du
    const float gx = gu * sc;                                // dL/dX_c
    s.g[joint * 2 + c] = gx;
    if (src < 24) s.ga[src * 3 + c] = gx;                    // seeds of the reverse sweep
    else s.h[(src - 24) * 3 + c] = gx;
    s.rpart[joint * 5 + 3 + c] = gx;                         // gt[c]
    d_out = d; dp_out = dp; gu_out = gu; xt_out = X + t;
    return Fit2P6{fmaf(d, d, 0.f), fmaf(dp, dp, 0.f), fmaf(gu, X + t, 0.f)};
}
// the y lane (lane & 1) finishes the joint's chains on the x lane's links
SPSP_HD void fit2_p6b(FitShared& s, int lane, const Fit2P6& x_link, float d, float dp, float gu, float xt) {
    if (!(lane & 1)) return;
    float* rp = s.rpart + (lane >> 1) * 5;
    rp[0] = fmaf(d, d, x_link.loss);
    rp[1] = fit_sqrt_1ulp(fmaf(dp, dp, x_link.l2));
    rp[2] = fmaf(gu, xt, x_link.gs);
}
#if defined(__HIPCC__)
__device__ __forceinline__ void fit2_p6(FitShared& s, int lane, const FitParams& P) {     // wave 0
    float d = 0.f, dp = 0.f, gu = 0.f, xt = 0.f;
    Fit2P6 q{0.f, 0.f, 0.f};
    if (lane < 34) q = fit2_p6a(s, lane, P, d, dp, gu, xt);
    Fit2P6 prev;                                             // lane l <- lane l - 1 (DPP row_shr 1: pairs never straddle a row of 16)
    prev.loss = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q.loss), 0x111, 0xf, 0xf, false));
    prev.l2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q.l2), 0x111, 0xf, 0xf, false));
    prev.gs = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q.gs), 0x111, 0xf, 0xf, false));
    if (lane < 34) fit2_p6b(s, lane, prev, d, dp, gu, xt);
}
#endif

// P10 of v2 in two pieces.  fit2_dj: the total dJ -- all that wave 3's d(beta) sums wait for.  fit2_dr: the rotation gradient,
// behind the barrier, which STAYS IN THE LANE's register: its Adam step (fit2_step)
// follows at once, beside wave 3's d(beta) sums instead of in front of them (the first form of v2 stepped in front: stamps
// showed the ~800 cycles of an Adam step -- a square root and two IEEE divisions on a lone wave -- twice on the critical path,
// once for the rotations and once for the betas).
SPSP_HD void fit2_dj_item(FitShared& s, int item, int wp_off) {      // dJ_total[k][l], item = 3 k + l  (fit_p10's statements)
    const float* const AB = s.Afin;
    float* f = fit_f(s);
    const int k = item / 3, l = item % 3;
    const float* A = AB + k * 12 + 3;
    float acc = s.dJ[item];
    if (k == 0) acc += A[4 * l];
    else {
        const float* Wp = f + wp_off;
        acc = fmaf(Wp[l], A[0], fmaf(Wp[4 + l], A[4], fmaf(Wp[8 + l], A[8], acc)));
    }
    const float* Wk = s.W + k * 12;
    const float c0 = A[0] - s.adir[k * 3], c1 = A[4] - s.adir[k * 3 + 1], c2 = A[8] - s.adir[k * 3 + 2];
    acc -= fmaf(Wk[l], c0, fmaf(Wk[4 + l], c1, Wk[8 + l] * c2));
    s.adj[15 + item] = acc;
}
// ... as a short phase of its own on lanes 0..71 (item = lane).  Measured in round 4: run by wave 0 right behind its reverse walk
// instead (two passes, no barrier pair) the iteration got 200 cycles LONGER -- the walk's wave is the critical one.
SPSP_HD void fit2_dj(FitShared& s, const FitLane& r, int lane) {
    if (lane >= 72) return;
    uint32_t role2 = r.role2;
    SPSP_OPAQUE(role2);
    fit2_dj_item(s, lane, (int)((role2 >> 10) & 1023));
}
SPSP_HD float fit2_dr(FitShared& s, const FitLane& r, int lane) {
    const float* const AB = s.Afin;
    float* f = fit_f(s);
    uint32_t role2 = r.role2;
    SPSP_OPAQUE(role2);
    float acc = 0.f;
    if (lane < 216) {                                        // item (k, l, c): dR_k[l][c]  (fit_p10's statements)
        const int k = lane / 9, e = lane % 9, l = e / 3, c = e % 3;
        if (k == 0) acc = AB[l * 4 + c];
        else {
            const float* Wp = f + (role2 & 1023);
            const float* A = AB + k * 12 + c;
            acc = Wp[l] * A[0];
            acc = fmaf(Wp[4 + l], A[4], acc);
            acc = fmaf(Wp[8 + l], A[8], acc);
            const F4 d0 = ld4(s.dvp), d1 = ld4(s.dvp + 4), d2 = ld4(s.dvp + 8), d3 = ld4(s.dvp + 12);
            const float dv[16] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w, d2.x, d2.y, d2.z, d2.w, d3.x, d3.y, d3.z, d3.w};
            SPSP_UNROLL
            for (int o = 0; o < 15; ++o) acc = fmaf(r.cQ[o], dv[o], acc);
        }
        s.grad[kPrmR + lane] = acc;                              // (kept for the host emulation's gradient read-out)
    }
    return acc;
}

// P11, wave 3 only (lanes 192..255): lane 192 + 16 g + i: the partial sums of beta_i over chunks 4 g .. 4 g + 3 of the adjoint
// vector (fit_p11's statements per chunk); the betas' lanes (219..228, the same wave) sum them in fit2_step.
SPSP_HD void fit2_p11a(FitShared& s, const Fit2Shared& x, int lane) {
    const int l = lane - 192;
    if (l < 0) return;
    const int i = l & 15, g = l >> 4;
    if (i >= 10) return;
    const float* C = x.cC + i * 96 + g * 24;
    const float* A = s.adj + g * 24;
    F4 c[6], a[6];
    SPSP_UNROLL
    for (int u = 0; u < 6; ++u) { c[u] = ld4(C + 4 * u); a[u] = ld4(A + 4 * u); }
    const float cf[24] = {c[0].x, c[0].y, c[0].z, c[0].w, c[1].x, c[1].y, c[1].z, c[1].w, c[2].x, c[2].y, c[2].z, c[2].w,
                          c[3].x, c[3].y, c[3].z, c[3].w, c[4].x, c[4].y, c[4].z, c[4].w, c[5].x, c[5].y, c[5].z, c[5].w};
    const float af[24] = {a[0].x, a[0].y, a[0].z, a[0].w, a[1].x, a[1].y, a[1].z, a[1].w, a[2].x, a[2].y, a[2].z, a[2].w,
                          a[3].x, a[3].y, a[3].z, a[3].w, a[4].x, a[4].y, a[4].z, a[4].w, a[5].x, a[5].y, a[5].z, a[5].w};
    float o[4];
    SPSP_UNROLL
    for (int u = 0; u < 4; ++u) {
        float acc = 0.f;
        SPSP_UNROLL
        for (int t = 0; t < 6; ++t) acc = fmaf(cf[u * 6 + t], af[u * 6 + t], acc);
        o[u] = acc;
    }
    st4(s.part + i * 16 + g * 4, o[0], o[1], o[2], o[3]);
}
// The Adam step of every parameter, in its owner's lane: rotation entries from the register gradient of fit2_p10, the camera from
// P7's reductions, the betas (wave 3, behind fit2_p11a and a wave-local hand-off) from the 16 partial sums in fit_p12's order.
SPSP_HD void fit2_step(FitShared& s, Fit2Shared& x, FitLane& r, int lane, float g_rot, float w1, float b2, float w2, float eps,
                       float step_size, float bc2_sqrt, float inv_bc2, float* param_hist_row) {
    const int p = fit2_owned_param(lane);
    if (p < 0) return;
    float g = g_rot;
    if (p < kPrmR) g = s.red[2 + p];
    else if (p >= kPrmBeta) {
        const float* pp = s.part + (p - kPrmBeta) * 16;
        const F4 a = ld4(pp), b = ld4(pp + 4), c = ld4(pp + 8), d = ld4(pp + 12);
        g = (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
    }
    fit2_adam(s, x, r, p, g, w1, b2, w2, eps, step_size, bc2_sqrt, inv_bc2, param_hist_row);
}


}  // namespace spsp
