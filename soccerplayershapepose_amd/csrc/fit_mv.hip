// GPU driver of the multi-view joint fit (phases and semantics: fit_mv_core.hpp): one 256-thread
// workgroup per player, every epoch of every block of every round inside ONE launch, state in LDS, Adam moments in
// registers.  Latency-bound like the single-view fit (a barrier after every phase: the sequence is short and the
// players are independent -- a frame's players fill one CU each).
#include <cmath>

#include "smpl.hpp"
#include "fit_mv_core.hpp"

namespace {
using namespace spsp;

struct MvArgs {
    FitParams P;
    int V, rounds;
    uint32_t mask_shared;
    const float* fitc;
    const int *parents, *level_joints, *level_start, *child_start, *child_list;
    float* body;            // [B][23][9]   in: initial shared body pose (mean over the views), out: best
    float* betas;           // [B][10]      in / out
    float* cam;             // [B][V][3]    in / out
    float* G;               // [B][V][9]    in / out
    const float* joints2d;  // [B][V][17][2]
    float* val_hist;        // [B][rounds * 2 * iters] or NULL
    const int* view_order;  // [rounds * 2 * iters][V] view visited at each train step of each epoch, or NULL (0 .. V-1)
    // single-block mode (only_block >= 0): ONE block of `iters` epochs from the given state, no record keeping -- every epoch's
    // parameters are written out instead (the reference's two-metric record rule is applied outside, spsp_fit_multiview_block)
    int only_block;
    float* snap_views;      // [B][iters][V][12]
    float* snap_shared;     // [B][iters][217]  body pose (207) | betas (10)
};

__global__ __launch_bounds__(spsp::kFitLanes) void fit_multiview_kernel(const MvArgs a) {
    __shared__ FitShared s;
    __shared__ MvShared m;
    __shared__ float Rinit[216];
    FitLane r;
    const int lane = threadIdx.x, b = blockIdx.x, V = a.V, L = a.P.num_levels;
    const FitParams& P = a.P;
    // single-view state: joint 0 = view 0's orientation, joints 1..23 = the shared body pose
    if (lane < 216) Rinit[lane] = lane < 9 ? a.G[((size_t)b * V) * 9 + lane] : a.body[(size_t)b * 207 + (lane - 9)];
    if (lane < V * kViewPrm) {
        const int u = lane / kViewPrm, p = lane - u * kViewPrm;
        m.vprm[lane] = p < 3 ? a.cam[((size_t)b * V + u) * 3 + p] : a.G[((size_t)b * V + u) * 9 + (p - 3)];
    }
    for (int i = lane; i < V * 34; i += kFitLanes) m.vtgt[i] = a.joints2d[(size_t)b * V * 34 + i];
    if (lane == 0) { m.best_val = INFINITY; m.val_acc = 0.f; m.improved = 0; }
    __syncthreads();
    fit_init(s, r, lane, a.fitc, a.parents, a.level_joints, a.level_start, a.child_start, a.child_list, Rinit,
             a.betas + (size_t)b * 10, a.cam + (size_t)b * V * 3, a.joints2d + (size_t)b * V * 34, 0u);
    __syncthreads();
    // the record before any epoch: the initial parameters (best_model_wts / best_model_view at :1711-1716)
    if (lane < V * kViewPrm) m.best_views[lane] = m.vprm[lane];
    if (lane < 232) m.best_shared[lane] = s.prm[lane];
    __syncthreads();
    float* const vh = a.val_hist ? a.val_hist + (size_t)b * a.rounds * 2 * P.iters : nullptr;
    const float w1 = (float)(1.0 - P.beta1), b2f = (float)P.beta2, w2 = (float)(1.0 - P.beta2), epsf = (float)P.eps;

    // forward of the current view in s (P1..P7: joints, projection, residuals, loss / l2e reductions)
#define MV_FORWARD()                                                                         \
    fit_p1(s, r, lane); __syncthreads();                                                     \
    fit_p2(s, lane); __syncthreads();                                                        \
    for (int l = 1; l < L; ++l) { fit_chain_level(s, r, r.lvl[l]); __syncthreads(); }               \
    fit_p4(s, r, lane); __syncthreads();                                                     \
    fit_p5(s, lane); __syncthreads();                                                        \
    fit_p6(s, r, lane, P); __syncthreads();                                                  \
    fit_p7(s, r, lane); __syncthreads();
#define MV_BACKWARD()                                                                        \
    fit_p8(s, lane, P, 0, nullptr, nullptr); __syncthreads();                                \
    for (int l = L - 2; l >= 0; --l) { fit_reverse_level(s, r, r.lvl[l]); __syncthreads(); }        \
    fit_p10(s, r, lane); __syncthreads();                                                    \
    fit_p11(s, r, lane); __syncthreads();

    for (int round = 0; round < a.rounds; ++round)
        for (int block = a.only_block >= 0 ? a.only_block : 0; block < (a.only_block >= 0 ? a.only_block + 1 : 2); ++block) {
            mv_block_start(s, m, r, lane, block, V, a.mask_shared);
            __syncthreads();
            double b1t = 1.0, b2t = 1.0;
            for (int ep = 0; ep < P.iters; ++ep) {
                const int* ord = a.view_order ? a.view_order + (size_t)((round * 2 + (a.only_block >= 0 ? 0 : block)) * P.iters + ep) * V : nullptr;
                for (int vi = 0; vi < V; ++vi) {              // ---- train: one optimiser step per view
                    const int v = ord ? ord[vi] : vi;
                    b1t *= P.beta1;
                    b2t *= P.beta2;
                    const float step_size = (float)(P.lr / (1.0 - b1t));
                    const float bc2_sqrt = (float)sqrt(1.0 - b2t);
                    mv_load_view(s, m, lane, v);
                    __syncthreads();
                    MV_FORWARD()
                    MV_BACKWARD()
                    if (block == 0) mv_adam_views(s, m, r, lane, v, V, w1, b2f, w2, epsf, step_size, bc2_sqrt);
                    else fit_p12(s, r, lane, w1, b2f, w2, epsf, step_size, bc2_sqrt);
                    __syncthreads();
                }
                for (int v = 0; v < V; ++v) {                 // ---- validation: joints2D_l2es of every view
                    mv_load_view(s, m, lane, v);
                    __syncthreads();
                    MV_FORWARD()
                    mv_val_add(s, m, lane, v == 0);
                    __syncthreads();
                }
                if (a.only_block >= 0) {                      // no record here: the epoch's validation metric and parameters go out
                    if (lane == 0 && vh) vh[ep] = m.val_acc / (float)V;
                    if (lane < V * kViewPrm) a.snap_views[((size_t)b * P.iters + ep) * V * kViewPrm + lane] = m.vprm[lane];
                    if (lane < 217) a.snap_shared[((size_t)b * P.iters + ep) * 217 + lane] = s.prm[kPrmR + 9 + lane];
                    __syncthreads();
                    continue;
                }
                mv_epoch_decide(m, lane, V, vh, (round * 2 + block) * P.iters + ep);
                __syncthreads();
                mv_epoch_snapshot(s, m, lane, block, V);
                __syncthreads();
            }
            if (a.only_block < 0) mv_block_end(s, m, lane, block, V);
            __syncthreads();
        }
    if (a.only_block >= 0) {                                  // the LAST iterate of the block
        if (lane < 207) a.body[(size_t)b * 207 + lane] = s.prm[kPrmR + 9 + lane];
        if (lane < 10) a.betas[(size_t)b * 10 + lane] = s.prm[kPrmBeta + lane];
        if (lane < V * kViewPrm) {
            const int u = lane / kViewPrm, p = lane - u * kViewPrm;
            if (p < 3) a.cam[((size_t)b * V + u) * 3 + p] = m.vprm[lane];
            else a.G[((size_t)b * V + u) * 9 + (p - 3)] = m.vprm[lane];
        }
        return;
    }
#undef MV_BACKWARD
#undef MV_FORWARD
    // what the reference saves (:1976-1979): the record's shared pose / betas and every view's camera / orientation
    if (lane < 207) a.body[(size_t)b * 207 + lane] = m.best_shared[kPrmR + 9 + lane];
    if (lane < 10) a.betas[(size_t)b * 10 + lane] = m.best_shared[kPrmBeta + lane];
    if (lane < V * kViewPrm) {
        const int u = lane / kViewPrm, p = lane - u * kViewPrm;
        if (p < 3) a.cam[((size_t)b * V + u) * 3 + p] = m.best_views[lane];
        else a.G[((size_t)b * V + u) * 9 + (p - 3)] = m.best_views[lane];
    }
}

}  // namespace

extern "C" {

void spsp_fit_multiview_config_default(spsp_fit_multiview_config* c) {
    if (!c) return;
    c->iters = 50;                 // S/global_var.py:79 player_recon_multi_view_iteration
    c->rounds = 3;                 // PR/player_recon.py:1720
    c->lr = 1e-3;                  // S/global_var.py:71
    c->beta1 = 0.9; c->beta2 = 0.999; c->eps = 1e-8;
    c->loss_weight = 1.0;
    c->img_wh = 512.f; c->loss_wh = 256.f;
    c->optimise_body_pose = 0;     // the reference's block 2 moves betas only (fit_mv_core.hpp)
}

static int fit_multiview_launch(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_multiview_config* cfg, int B, int V, float* body_pose, float* betas,
                                float* cam, float* global_orient, const float* joints2d, const int32_t* view_order, float* val_hist,
                                int only_block, float* snap_views, float* snap_shared, void* stream) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && s && cfg, "spsp_fit_multiview: NULL handle");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    SPSP_REQUIRE(B >= 0, "spsp_fit_multiview: negative batch");
    SPSP_REQUIRE(V >= 1 && V <= kMaxViews, "spsp_fit_multiview: %d views (1..%d)", V, kMaxViews);
    if (B == 0) return SPSP_OK;
    SPSP_REQUIRE(body_pose && betas && cam && global_orient && joints2d, "spsp_fit_multiview: NULL argument");
    SPSP_REQUIRE(cfg->iters >= 0 && cfg->iters <= 100000 && cfg->rounds >= 0 && cfg->rounds <= 1000, "spsp_fit_multiview: iters / rounds out of range");
    SPSP_REQUIRE(cfg->loss_wh > 0.f && cfg->img_wh > 0.f, "spsp_fit_multiview: img_wh / loss_wh must be positive");
    MvArgs a;
    a.P.iters = cfg->iters;
    a.P.lr = cfg->lr; a.P.beta1 = cfg->beta1; a.P.beta2 = cfg->beta2; a.P.eps = cfg->eps;
    a.P.param_mask = 0;
    a.P.img_wh = cfg->img_wh; a.P.loss_wh = cfg->loss_wh;
    a.P.log_var = (float)(-std::log(cfg->loss_weight + 1e-6));
    a.P.weight = std::exp(-a.P.log_var);
    a.P.select_mode = SPSP_FIT_SELECT_LAST;
    a.P.num_levels = s->num_levels;
    a.V = V; a.rounds = only_block >= 0 ? 1 : cfg->rounds;
    a.mask_shared = SPSP_FIT_BETAS | (cfg->optimise_body_pose ? 0x003FFE7Eu : 0u);
    a.fitc = s->fitc; a.parents = s->parents; a.level_joints = s->level_joints; a.level_start = s->d_level_start;
    a.child_start = s->child_start; a.child_list = s->child_list;
    a.body = body_pose; a.betas = betas; a.cam = cam; a.G = global_orient; a.joints2d = joints2d; a.val_hist = val_hist;
    a.view_order = view_order; a.only_block = only_block; a.snap_views = snap_views; a.snap_shared = snap_shared;
    ProfScope ps(ctx, kKFit, as_stream(stream));
    fit_multiview_kernel<<<B, kFitLanes, 0, as_stream(stream)>>>(a);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

int spsp_fit_multiview(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_multiview_config* cfg, int B, int V, float* body_pose,
                       float* betas, float* cam, float* global_orient, const float* joints2d, float* val_hist, void* stream) {
    return fit_multiview_launch(ctx, s, cfg, B, V, body_pose, betas, cam, global_orient, joints2d, nullptr, val_hist, -1, nullptr, nullptr, stream);
}

int spsp_fit_multiview_ordered(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_multiview_config* cfg, int B, int V, float* body_pose,
                               float* betas, float* cam, float* global_orient, const float* joints2d, const int32_t* view_order,
                               float* val_hist, void* stream) {
    return fit_multiview_launch(ctx, s, cfg, B, V, body_pose, betas, cam, global_orient, joints2d, view_order, val_hist, -1, nullptr, nullptr, stream);
}

int spsp_fit_multiview_block(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_multiview_config* cfg, int block, int B, int V, float* body_pose,
                             float* betas, float* cam, float* global_orient, const float* joints2d, const int32_t* view_order,
                             float* val_l2e, float* snap_views, float* snap_shared, void* stream) {
    SPSP_REQUIRE(block == 0 || block == 1, "spsp_fit_multiview_block: block %d (0: cameras + orientations, 1: shared pose / betas)", block);
    SPSP_REQUIRE(B == 0 || (snap_views && snap_shared), "spsp_fit_multiview_block: NULL snapshot buffer");
    return fit_multiview_launch(ctx, s, cfg, B, V, body_pose, betas, cam, global_orient, joints2d, view_order, val_l2e, block, snap_views, snap_shared, stream);
}

}  // extern "C"
