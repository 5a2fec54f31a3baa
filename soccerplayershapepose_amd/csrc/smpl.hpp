// Device-side SMPL model image shared by the forward kernels (smpl.hip) and the fit kernel (fit.hip).
#pragma once
#include "common.hpp"
#include "smpl_consts.hpp"



struct spsp_smpl {
    spsp::DevicePool pool;
    int max_batch;
    // forward
    float* basis;        // [217][20670]: rows 0..9 shapedirs^T, rows 10..216 posedirs
    float* v_template;   // [20670]
    float* wT;           // [24][6890] lbs weights, joint-major
    float* jconst;       // [72 + 720] Jbar[24][3] then JS[10][72]
    int* parents;        // [24]
    int* level_joints;   // [24] joints sorted by depth (root first)
    int level_start[26]; // host copy: level l occupies level_joints[level_start[l] .. level_start[l+1])
    int num_levels;
    int* d_level_start;  // device copy [26]
    int* child_start;    // [25] children of joint k: child_list[child_start[k] .. child_start[k+1])
    int* child_list;     // [24] (23 used)
    uint32_t* walk;      // [64 * 4 + 25] programs of the fit's tree walks (SmplHostImage::walk)
    bool walk_ok;
    int walk_nb;
    int skin_players = 8;   // players per workgroup of the skin kernel (8; 16: experiment, spsp_smpl_debug)
    int fit_schedule = 2;   // 2: fit_kernel2 (default) | 1: the ten-barrier schedule with the walks | 0: level sweeps (spsp_smpl_debug)
    int* pick_ids;       // [21]
    int* csr_ptr;        // [46] extra-joint regressors (9 + 19 + 17 rows) in CSR
    int* csr_col;
    float* csr_val;
    // fit
    float* fitc;         // [kFcTotal]
    int face_ids[spsp::kNFace];
    // workspace
    float* ws_A;         // [max_batch][24][12]
    float* ws_verts;     // [max_batch][6890][3] (only used when the caller passes verts == NULL)
    int host_parents[24];
    // differentiable path (smpl_vjp.hip, fit_sil.hip), allocated at first use
    float* ws_vposed = nullptr;  // [max_batch][6890][3]
    float* ws_dvp = nullptr;     // [max_batch][6890][3]   dL/d(posed rest vertex)
    float* ws_seedp = nullptr;   // [max_batch][27][288]   chunk partials of the per-joint seeds
    float* ws_seeds = nullptr;   // [max_batch][24][12]    rows [S1_k[i][0..2] | S0_k[i]]
    float* ws_gcoef = nullptr;   // [max_batch][217]       basis . dvp
    float* ws_gverts = nullptr;  // [max_batch][6890][3]
    float* ws_adam = nullptr;    // [max_batch][2][232]    Adam moments of the silhouette fit
    float* ws_misc = nullptr;    // [max_batch][16]        translation(3) | grad_translation(3) | sil_sum | weight | ...
};

namespace spsp {
// What spsp_pipeline lets the fit kernel do at its two ends instead of launching three more kernels around it (each ~4 us plus
// a launch boundary of the serial tail a pipeline slot waits for):
//   pose6d      [B,144] the regressor's 6-D rotations: R is built from them in the kernel (spsp_rot6d_to_rotmat) -- the R
//               argument is then output only;
//   translation [B,3]   spsp_wp_to_translation of the returned camera;      record [B,229] spsp_pack_record of the returned state.
struct FitEnds {
    const float* pose6d = nullptr;
    float* translation = nullptr;
    float focal = 0.f, resolution = 0.f;
    float* record = nullptr;
};
int fit_launch(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_config* cfg, float* R, float* betas, float* cam, const float* joints2d,
               int B, float* loss_hist, float* l2e_hist, float* param_hist, const FitEnds& ends, void* stream);
int smpl_forward_impl(spsp_ctx* ctx, spsp_smpl* s, const float* R, const float* betas, int B, float* verts, float* joints,
                      float* vposed, void* stream);
int smpl_vjp_workspace(spsp_smpl* s);
// grad_verts [B,6890,3] -> seeds + gcoef workspaces of the handle (needs a forward with vposed for the same R / betas before it)
int smpl_vjp_seeds(spsp_ctx* ctx, spsp_smpl* s, const float* grad_verts, int B, void* stream);
}
