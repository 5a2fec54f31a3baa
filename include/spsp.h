/* spsp.h -- C ABI of libspsp.so: the MI355X (gfx950) hot path of SoccerPlayerShapePose.
 *
 *   proxy crop [18,256,256] -> ResNet-18 + IEF -> (cam, pose6d, betas) -> rot6d -> R[24,3,3]
 *   -> per-player 2D-reprojection fit (analytic gradient + Adam, all iterations in one launch)
 *   -> SMPL forward (blend shapes + 24-joint LBS) -> vertices [6890,3], joints [90,3]
 *
 * The reference has no FFI: its boundary is a set of Python callables.  Each entry
 * point below names the reference callable it replaces; INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller (PyTorch) unless it is marked
 *     "host"; the library never allocates or frees caller-visible memory.
 *   - all fp32 tensors are dense row-major with the shapes given; R is [B,24,3,3] row-major with
 *     joint 0 = global orientation, joints 1..23 = body pose (smplx full_pose order).
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream) and performs no host synchronisation.
 *   - return value: SPSP_OK or a negative status; spsp_last_error() gives a thread-local message.
 *   - opaque handles are immutable after load; a handle's internal workspace serves ONE call in
 *     flight at a time (enqueue calls that share a handle on one stream, or load one handle per stream).
 */
#ifndef SPSP_H
#define SPSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPSP_OK       0
#define SPSP_EINVAL  (-1)
#define SPSP_ENOMEM  (-2)
#define SPSP_EHIP    (-3)
#define SPSP_ENCCL   (-4)

#define SPSP_NUM_VERTS      6890
#define SPSP_NUM_JOINTS     24
#define SPSP_NUM_BETAS      10
#define SPSP_NUM_POSE_BASIS 207
#define SPSP_NUM_OUT_JOINTS 90   /* 24 chain + 21 picked vertices + 9 extra + 19 cocoplus + 17 h36m */
#define SPSP_NUM_FIT_JOINTS 17   /* PR/config.py:38 SMPL_TO_KPRCNN_MAP */
#define SPSP_RECORD_FLOATS  229  /* [cam(3) | R(24*9) | betas(10)] -- the all-gather record, SURVEY 8e */
#define SPSP_NUM_CONVS      20   /* convs of resnet18(in_channels=18), order of spsp_conv_name() */
#define SPSP_PROXY_C        18
#define SPSP_PROXY_WH       256
#define SPSP_PROXY_PAD      3    /* zero border of the backbone-native input layout */
#define SPSP_PROXY_NATIVE_H  262  /* 256 + 2*3 rows */
#define SPSP_PROXY_NATIVE_W  264  /* 3 + 256 + 5 columns: row pitch 264*18*2 B = 9504 B is 16-byte aligned */

typedef struct spsp_ctx       spsp_ctx;
typedef struct spsp_regressor spsp_regressor;
typedef struct spsp_smpl      spsp_smpl;

/* ---- context -------------------------------------------------------------------------------- */
/* Binds the calling thread to HIP device `device` and creates the context. */
int         spsp_init(int device, spsp_ctx** out);
void        spsp_destroy(spsp_ctx* ctx);
const char* spsp_last_error(void);
const char* spsp_version(void);

/* ---- per-kernel timing (bench.py roofline leg): while enabled, every launch site records a HIP-event
 *      pair on the launch stream; spsp_profile_read synchronises them, sums milliseconds and launch counts
 *      per kernel id (a scope that issues several launches of one kernel counts as one record per launch
 *      site) and clears the records.  An event pair is a barrier on its in-order stream (no back-to-back
 *      dispatch across it), so spsp_profile_select(kernel_id) restricts recording to one kernel id
 *      (-1 = all) for use inside a throughput-timed region. */
int         spsp_profile_enable(spsp_ctx* ctx, int on);
int         spsp_profile_select(spsp_ctx* ctx, int kernel_id);
int         spsp_profile_num_kernels(void);
const char* spsp_profile_kernel_name(int id);
/* Algorithmic FLOPs one launch of kernel `id` does per crop, when every launch of it does the same work (the MFMA
 * convolutions; 0 for everything else), and whether launch plan `plan` (SPSP_PLAN_THROUGHPUT | SPSP_PLAN_LATENCY, below) launches
 * the kernel for a batch of 32. */
double      spsp_profile_kernel_flop_per_crop(int id);
int         spsp_profile_kernel_in_plan(int id, int plan);
int         spsp_profile_read(spsp_ctx* ctx, int n, double* total_ms, int* launches);
/* the same, and per kernel the workgroups of its single-launch scopes summed (0 for the others): launches narrower than the chip
 * (plan "throughput") are priced against the CUs they occupy */
int         spsp_profile_read_grids(spsp_ctx* ctx, int n, double* total_ms, int* launches, double* total_workgroups);
/* Instead of the sums: one (kernel id, stream, start, stop) record per bracketed launch site, times in ms relative to the
 * first record; clears the records.  For timeline studies of the multi-stream pipeline (tools/timeline.py). */
int         spsp_profile_timeline(spsp_ctx* ctx, int max_records, int* ids, unsigned long long* streams, double* start_ms,
                                  double* stop_ms, int* count);

/* ---- relation module: replaces PoseRelationModule (PR/models/regressor_relate.py:9-139; used between rot6d and
 *      the fit at PR/PyTorch3DTest.py:3758-3759 when use_relate) -- the players of ONE frame attend to each other.
 *      Host arrays in nn.Linear layout: W*.weight [out][in], W*.bias [out]. */
#define SPSP_NUM_RELATIONS 27
typedef struct {
    const float* wg_weight;   /* [1][8]   */
    const float* wg_bias;     /* [1]      */
    const float* wk_weight;   /* [8][216] */
    const float* wk_bias;     /* [8]      */
    const float* wq_weight;   /* [8][216] */
    const float* wq_bias;     /* [8]      */
    const float* wv_weight;   /* [8][216] */
    const float* wv_bias;     /* [8]      */
} spsp_relate_unit;
typedef struct { spsp_relate_unit unit[SPSP_NUM_RELATIONS]; } spsp_relate_weights;
typedef struct spsp_relate spsp_relate;
int  spsp_relate_load(spsp_ctx* ctx, const spsp_relate_weights* w, int max_players, spsp_relate** out);
void spsp_relate_free(spsp_relate* h);
/* pose_rotmats [N,24,3,3] and boxes [N,4] = (x_min, y_min, x_max, y_max) on the device -> out [N,24,3,3]
 * (may not alias pose_rotmats).  N = all players of the frame (the operator couples them: shard BY FRAME). */
int  spsp_relate_forward(spsp_ctx* ctx, spsp_relate* h, const float* pose_rotmats, const float* boxes, int N,
                         float* out, void* stream);

/* ---- regressor: replaces SingleInputRegressor (PR/models/regressor.py:13-48) ------------------ */
typedef struct {
    const float* weight;        /* host [cout,cin,k,k] fp32 (torch Conv2d layout), no bias (resnet.py:30-36) */
    const float* bn_weight;     /* host [cout] */
    const float* bn_bias;       /* host [cout] */
    const float* bn_mean;       /* host [cout] running_mean */
    const float* bn_var;        /* host [cout] running_var  */
} spsp_conv_bn;

typedef struct {
    spsp_conv_bn conv[SPSP_NUM_CONVS]; /* eval-mode BN is folded into the conv at load (eps 1e-5) */
    const float* fc_weight[3];         /* host fc1 [512,669], fc2 [512,512], fc3 [157,512] (ief_module.py:16-18) */
    const float* fc_bias[3];           /* host [512], [512], [157] */
    const float* init_params;          /* host [157] = [0.9,0,0 | mean pose6d(144) | mean shape(10)] (ief_module.py:33-46) */
    int          ief_iters;            /* 3 (regressor.py:31-34) */
} spsp_regressor_weights;

/* State-dict prefix of conv i (e.g. "image_encoder.layer2.0.downsample.0"); NULL when out of range. */
const char* spsp_conv_name(int i);
/* State-dict prefix of the BatchNorm that follows conv i. */
const char* spsp_conv_bn_name(int i);

/* precision: 0 = bf16 weights/activations on v_mfma_f32_16x16x32_bf16, fp32 accumulate (the default);
 *            1 = fp16 weights/activations on v_mfma_f32_16x16x32_f16, fp32 accumulate: same layouts, bytes and
 *                speed, 3 more mantissa bits (activations are O(1..10), well inside fp16 range) -- the mode in
 *                which the END-TO-END vertices stay within 1e-3 m of the all-fp32 reference path. */
#define SPSP_PRECISION_BF16 0
#define SPSP_PRECISION_F16  1
/* max_batch <= SPSP_REGRESSOR_MAX_BATCH players per forward (the fused IEF kernel holds eight blocks of 32; larger sets:
 * several calls).  At most SPSP_REGRESSOR_MAX_CONCURRENT forwards (of any handles) may be in flight on a device at once:
 * the fused IEF's workgroups wait for each other, and that many launches can always make progress (checked at load
 * against the device's occupancy).  Beyond it a hand-off may starve; it then times out, the outputs are NaN and the
 * next call on the handle -- or spsp_regressor_status -- returns SPSP_EHIP. */
#define SPSP_REGRESSOR_MAX_BATCH      256
#define SPSP_REGRESSOR_MAX_CONCURRENT 8
int  spsp_regressor_load(spsp_ctx* ctx, const spsp_regressor_weights* w, int max_batch, int precision,
                         spsp_regressor** out);
void spsp_regressor_free(spsp_regressor* r);

#define SPSP_PROXY_NCHW_F32   0  /* [B,18,256,256] fp32 -- what the reference feeds (predict_3D.py:126) */
#define SPSP_PROXY_NATIVE     1  /* [B,262,264,18] NHWC in the handle's element type (bf16 / fp16), pixel (y,x) at [y+3][x+3], everything else ZERO (caller keeps it so) */

/* Bytes of one crop in the native layout. */
size_t spsp_proxy_native_bytes(void);
/* NCHW fp32 -> native layout in the element type of `precision` (writes interior AND border). */
int spsp_proxy_pack(spsp_ctx* ctx, const float* proxy_nchw, int B, int precision, void* native_out, void* stream);

/* Proxy-representation builder (replaces create_proxy_representation, PR/predict/predict_3D.py:67-81, and
 * convert_2Djoints_to_gaussian_heatmaps, PR/utils/label_conversions.py:58-87, for a batch): device silhouettes
 * [B,512,512] (uint8 or fp32, values 0/1) + device joints2D [B,17,joints_stride] (x, y[, conf]; 512-px space)
 * -> native layout [B,262,264,18] in the element type of `precision`, border included.  Bit-identical to
 * spsp_proxy_pack applied to the reference routine's [18,256,256] output. */
#define SPSP_SIL_U8  0
#define SPSP_SIL_F32 1
int spsp_proxy_build(spsp_ctx* ctx, const void* silhouettes, int sil_format, const float* joints2d, int joints_stride,
                     int B, int precision, void* native_out, void* stream);

/* Front half of the proxy builder: the detected boxes of ONE frame -> 512 x 512 player crops (replaces crop_image,
 * S/global_utils.py:257-290, + cv2.resize(..., player_crop_size, INTER_LINEAR), S/PlayerDetection/player_detection.py:
 * 181-184, for all boxes in one launch).  frame: device [H,W,3] uint8 (BGR, as cv2.imread returns it); boxes: device
 * [N,4] int32 (x0, y0, x1, y1 -- the detector's boxes after astype(int32), player_detection.py:30); border:
 * player_crop_border (40) or player_crop_border_broad (15); crops: device [N,out_wh,out_wh,3] uint8.  The box grows by
 * `border`, is clamped to the frame, padded to a square with BGR (53, 88, 83) and resized with OpenCV's 8-bit bilinear
 * arithmetic (restated in csrc/crop.hip; OpenCV itself is an un-vendored dependency of the reference). */
int spsp_crop_players(spsp_ctx* ctx, const uint8_t* frame, int H, int W, const int32_t* boxes, int N, int border, int out_wh,
                      uint8_t* crops, void* stream);

/* forward: proxy -> cam [B,3], pose6d [B,144], betas [B,10]; feats [B,512] optional (NULL to skip). */
int spsp_regressor_forward(spsp_ctx* ctx, spsp_regressor* r, const void* proxy, int proxy_format, int B,
                           float* cam, float* pose6d, float* betas, float* feats, void* stream);
/* The two halves of spsp_regressor_forward on their own (it is exactly backbone(..., 0) + ief):
 *   backbone: proxy crops -> pooled features, rows [feat_row, feat_row + B) of the handle's feature buffer (max_batch rows);
 *   ief:      the iterative-error-feedback MLP on rows [0, B) of that buffer -> cam / pose6d / betas (feats optional).
 * Several backbone calls with different feat_row, then ONE ief over all their rows: how spsp_pipeline_group batches the
 * latency-bound tail of the path (IEF -> fit -> SMPL) over consecutive batches while the convolutions keep their batch.
 * The fused IEF kernel's grid-wide hand-offs rely on state of the handle that must start as zeros: the per-phase hand-off buffers a
 * consumer polls until its words are non-zero (round 5; 2 x iterations x 32 KB per 16 players) and, in the arrival-counter forms, eight
 * monotonic counters.  A backbone call leaves both zeroed for the rows it covers (its stem kernel does it, every workgroup a share), an
 * ief call uses them up.  spsp_regressor_ief tracks that on the handle (host side) and, when no backbone call has run since the last
 * ief -- ief twice on the same features, ief on a fresh handle -- zeroes them itself with memsets on `stream` first; both orders give
 * the same results.  (Under stream capture the choice is made when
 * the call is captured: capture backbone + ief together, or an ief alone that is preceded by a backbone call at capture time
 * only if every replay is.) */
int spsp_regressor_backbone(spsp_ctx* ctx, spsp_regressor* r, const void* proxy, int proxy_format, int B, int feat_row, void* stream);
int spsp_regressor_ief(spsp_ctx* ctx, spsp_regressor* r, int B, float* cam, float* pose6d, float* betas, float* feats, void* stream);
/* Launch plan of the handle's forwards (same results up to the fp32 summation order of layer2 and layer3; takes effect at the next forward,
 * not inside a captured graph).
 * SPSP_PLAN_LATENCY (a fresh handle's plan): for ONE batch in flight -- every kernel's grid fills the chip.
 * SPSP_PLAN_THROUGHPUT: for several batches in flight on different streams (four pipeline slots) -- the kernels with the least CU
 * time per batch: the persistent kernels run NARROW (layer1 on a quarter of the CUs with eight tiles per resident weight set,
 * the stem in bands of 32 rows, layer2.0 four tiles per workgroup), layer3 one image per workgroup without the K split (conv_p4.hip, half the chip), layer4 in
 * 64-channel blocks (half the chip).  One batch alone takes ~0.2 ms longer this way; four in flight finish 6-8 % more players/s. */
#define SPSP_PLAN_THROUGHPUT 0
#define SPSP_PLAN_LATENCY    1
int spsp_regressor_set_plan(spsp_regressor* r, int plan);
/* SPSP_OK, or SPSP_EHIP when a kernel of an earlier forward on this handle reported a device-side failure (the fused
 * IEF's bounded hand-off wait).  Reads a host-visible status word: no synchronisation, callable at any time -- e.g. after a
 * HIP-graph replay, which makes no library call that could report it.  Reading clears the status. */
int spsp_regressor_status(spsp_regressor* r);
/* Test hooks.  SPSP_DEBUG_IEF_STALL (value != 0): the NEXT forward launches its IEF one workgroup short of what its
 * hand-offs wait for, with a short spin bound -- exercises the time-out report without hanging anything. */
#define SPSP_DEBUG_IEF_STALL 1
/* Plan switches for A/B measurements inside one process (tools/ab_plan.py); results are parity-checked in every setting:
 * SPSP_DEBUG_PLAN_S2 (0: the stride-2 stages on the round-2 implicit-GEMM ring kernel, 1: parity-plane patch kernel, default),
 * SPSP_DEBUG_PLAN_LAYER4_CHANNELS (64 or 32 output channels per workgroup of layer4's 3x3 convolutions; default: the handle's plan),
 * SPSP_DEBUG_PLAN_PATCH: value = layer (1..3 = layer2..4) * 10000 + tile pixels (128 | 256) * 10 + ring groups (2 | 3) * 2 + two patch buffers (0 | 1). */
#define SPSP_DEBUG_PLAN_S2 2
#define SPSP_DEBUG_PLAN_LAYER4_CHANNELS 3
#define SPSP_DEBUG_PLAN_PATCH 4
#define SPSP_DEBUG_PLAN_PATCH_PRODUCER 5   /* value = layer (1..3) * 10 + (0 | 1): four DMA-only producer waves in that stage's patch kernel */
#define SPSP_DEBUG_PLAN_S2RS_CUS 11        /* conv_s2rs: tiles per workgroup as for a chip of 1 / value of the CUs (1, 2, 4, 8: 2, 4, 8 tiles at B = 32) */
#define SPSP_DEBUG_PLAN_P4_CUS 10          /* conv_p4's workgroups: at most 1 / value of the CUs (1, 2, 4, 8), persistent over the tiles beyond */
#define SPSP_DEBUG_PLAN_STEM_BAND 9        /* conv rows per workgroup of the stem kernel (0: grid-filling default, 16, 32, 64; capped where fewer than a quarter of the CUs would get a workgroup) */
#define SPSP_DEBUG_PLAN_WRES_CUS 8         /* layer1's persistent workgroups on 1 / value of the CUs (1, 2, 4, 8, 16): more tiles per resident weight set */
#define SPSP_DEBUG_PLAN_P4 7               /* bit L set: the 3x3 stride-1 convs of layer L+1 on conv_p4.hip (256-pixel tiles, no K split, producer waves); default: bit 1 = layer2 */
#define SPSP_DEBUG_PLAN_RS 16             /* layer2's 3x3 stride-1 convs on conv_rs.hip (a channel block's weights in registers, persistent) with this many workgroups; 0: conv_p4.hip.  Plan "throughput": 64, plan "latency": 0 */
#define SPSP_DEBUG_IEF_POLL 17            /* the IEF whose hand-offs poll the data itself (ief_poll_kernel): 2 (default) with 32 lanes per player pair, 1 with 16 (the counter forms' sums bit for bit); 0: the arrival-counter forms (SPSP_DEBUG_IEF_BLOCK) */
#define SPSP_DEBUG_PLAN_STEM32 15         /* the stem kernel: 1 (default) on v_mfma_f32_32x32x16 with the pooling inside the K loop, 2: the same MFMAs with the pooling behind the loop, 0: on 16x16x32 (rounds 1-4) */
#define SPSP_DEBUG_PLAN_S2Q 14            /* 1 (default): layer2.0's stride-2 pair on conv_s2q.hip (all 128 output channels per workgroup), 0: conv_s2rs.hip (64-channel blocks) */
#define SPSP_DEBUG_PLAN_FB 13             /* 1 (default): layer1 as one launch per BasicBlock (conv_fb.hip), 0: two launches per block (conv_wres.hip) */
#define SPSP_DEBUG_IEF_BLOCK 12            /* players per block of the fused IEF (the 64 workgroups that hand their hidden vectors to each other): 32, or 16 = two independent halves per 32 players */
#define SPSP_DEBUG_PLAN_S2RS 6             /* 0: layer2.0's stride-2 pair on the streaming kernel of the other stages, 1 (default): weights stationary in registers */
int spsp_regressor_debug(spsp_regressor* r, int what, int value);
/* Test hook of an SMPL handle.  SPSP_DEBUG_FIT_SCHEDULE: which form of the reprojection-fit kernel spsp_fit / spsp_pipeline launch on
 * this handle -- 2 (default) the current schedule, 1 the ten-barrier schedule of rounds 1-3 (tree walks, every IEEE division and
 * square root), 0 the level-by-level sweeps
 * (what a model with a tree other than SMPL's always gets).  Same algorithm; A/B switch of tests and tools, never read from the
 * environment. */
#define SPSP_DEBUG_FIT_SCHEDULE 1
/* SPSP_DEBUG_SKIN_PLAYERS: players per workgroup of the SMPL skin kernel, 8 (default) or 16 (the blend basis read half as often; slower). */
#define SPSP_DEBUG_SKIN_PLAYERS 2
int spsp_smpl_debug(spsp_smpl* s, int what, int value);
/* Debug/parity tap: copies the (unpadded) NHWC bf16 output of stage `stage`
 * (1 maxpool, 2..5 layer1..layer4; stage 0 -- conv1+relu -- no longer exists as a map since conv1 and the max-pool are one
 * kernel: SPSP_EINVAL) of the LAST forward as fp32 NCHW into `out`. */
int spsp_regressor_read_stage(spsp_ctx* ctx, spsp_regressor* r, int stage, int B, float* out_nchw, void* stream);

/* ---- small geometry ops ---------------------------------------------------------------------- */
/* rot6d_to_rotmat (PR/utils/rigid_transform_utils.py:27-41): pose6d [n,6] -> R [n,3,3]. */
int spsp_rot6d_to_rotmat(spsp_ctx* ctx, const float* pose6d, int n, float* R, void* stream);
/* orthographic_project_torch + undo_keypoint_normalisation (PR/utils/cam_utils.py:5-26,
 * PR/utils/joints2d_utils.py:5-10): pts [B,n,3], cam [B,3] -> out [B,n,2].
 * img_wh > 0: pixels (uv+1)*img_wh/2;  img_wh == 0: normalised uv. */
int spsp_project(spsp_ctx* ctx, const float* pts, const float* cam, int B, int n, float img_wh,
                 float* out, void* stream);
/* convert_weak_perspective_to_camera_translation_torch (PR/utils/cam_utils.py:28-34). */
int spsp_wp_to_translation(spsp_ctx* ctx, const float* cam, int B, float focal, float resolution,
                           float* translation, void* stream);
/* [cam | R | betas] -> record [B,229] and back (the RCCL all-gather payload). */
int spsp_pack_record(spsp_ctx* ctx, const float* cam, const float* R, const float* betas, int B,
                     float* record, void* stream);
int spsp_unpack_record(spsp_ctx* ctx, const float* record, int B, float* cam, float* R, float* betas,
                       void* stream);

/* ---- SMPL: replaces SMPL.forward (PR/models/smpl_official.py:27-41 -> smplx.lbs.lbs) ---------- */
typedef struct {
    const float*   v_template;          /* host [6890,3] */
    const float*   shapedirs;           /* host [6890,3,10] */
    const float*   posedirs;            /* host [207,20670] (smplx layout) */
    const float*   J_regressor;         /* host [24,6890] dense */
    const float*   lbs_weights;         /* host [6890,24] */
    const int32_t* parents;             /* host [24], parents[0] = -1 */
    const int32_t* vertex_joint_ids;    /* host [21] VertexJointSelector ids */
    const float*   J_regressor_extra;   /* host [9,6890]  (smpl_official.py:17) */
    const float*   J_regressor_cocoplus;/* host [19,6890] (smpl_official.py:18) */
    const float*   J_regressor_h36m;    /* host [17,6890] (smpl_official.py:19) */
} spsp_smpl_arrays;

int  spsp_smpl_load(spsp_ctx* ctx, const spsp_smpl_arrays* a, int max_batch, spsp_smpl** out);
void spsp_smpl_free(spsp_smpl* s);

/* R [B,24,9], betas [B,10] -> verts [B,6890,3] (NULL to skip), joints [B,90,3] (NULL to skip). */
int spsp_smpl_forward(spsp_ctx* ctx, spsp_smpl* s, const float* R, const float* betas, int B,
                      float* verts, float* joints, void* stream);

/* ---- fit: replaces optimize_camera (PR/PyTorch3DTest.py:3055-3124) and the fit body of
 *      single_view_optimization (PR/player_recon.py:1172-1294) -------------------------------- */
#define SPSP_FIT_GLOBAL_ORIENT  (1u << 0)           /* joint 0 */
#define SPSP_FIT_BODY_JOINT(k)  (1u << (k))          /* SMPL joint k = 1..23 (body_pose index k-1) */
#define SPSP_FIT_CAM            (1u << 24)
#define SPSP_FIT_BETAS          (1u << 25)
/* variant A: {cam, global_orient};  variant B: + body_pose idx {0-5, 8-20} = joints {1-6, 9-21} + betas */
#define SPSP_FIT_MASK_CAMERA    (SPSP_FIT_GLOBAL_ORIENT | SPSP_FIT_CAM)
#define SPSP_FIT_MASK_POSE_SHAPE (SPSP_FIT_GLOBAL_ORIENT | SPSP_FIT_CAM | SPSP_FIT_BETAS | 0x003FFE7Eu)

#define SPSP_FIT_SELECT_LAST     0  /* what optimize_camera leaves in place */
#define SPSP_FIT_SELECT_BEST_L2E 1  /* lowest joints2D_l2es iterate (ties -> later iterate) */

typedef struct {
    int      iters;        /* 50 */
    double   lr;           /* 0.01 (A) / 1e-3 (B); doubles because torch derives the step size in Python floats */
    double   beta1, beta2, eps;  /* torch.optim.Adam defaults 0.9, 0.999, 1e-8 */
    double   loss_weight;  /* 1.0 -> log_var = -ln(w + 1e-6) (multi_task_loss.py:37-38) */
    uint32_t param_mask;   /* SPSP_FIT_* bits */
    float    img_wh;       /* 512: pixel space of joints2D / projections (S/global_var.py:32) */
    float    loss_wh;      /* 256: divisor used by the loss normalisation (multi_task_loss.py:106-107) */
    int      select_mode;  /* SPSP_FIT_SELECT_* */
} spsp_fit_config;

void spsp_fit_config_default(spsp_fit_config* c, int variant /* 0 = A camera, 1 = B pose+shape */);

/* In place on R [B,24,9], betas [B,10], cam [B,3]; joints2d [B,17,2] pixels (img_wh space).
 * loss_hist / l2e_hist: optional [B,iters] (value BEFORE the step of each iteration), NULL to skip. */
int spsp_fit(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_config* cfg, float* R, float* betas, float* cam,
             const float* joints2d, int B, float* loss_hist, float* l2e_hist, void* stream);

/* The same fit, also recording the trajectory: param_hist [B,iters,229] = the record [cam | R | betas] each iteration
 * EVALUATED (the parameters before its step -- what the reference snapshots when it saves an iterate), NULL to skip. */
int spsp_fit_trajectory(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_config* cfg, float* R, float* betas, float* cam,
                        const float* joints2d, int B, float* loss_hist, float* l2e_hist, float* param_hist, void* stream);

/* ---- silhouette term: the rendered silhouette of the posed mesh, the tracker's "silhouette_iou" metric and the
 *      reference's exact best-iterate rule.  Replaces `nmr(vertices, faces, t=translation, mode='silhouettes')`
 *      (neural_renderer, camera_mode='projection', K from FOCAL_LENGTH, R = I, image 512, anti-aliasing on; PR/player_recon.py:
 *      1075-1080, 1226-1229), TrainingLossesAndMetricsTracker's silhouette_iou (PR/metrics/train_loss_and_metrics_tracker.py:
 *      224-229) and determine_save_model_weights_this_epoch over both tracked metrics (:291-298, PR/player_recon.py:1259-1271).
 *      The differentiable half (the silhouette LOSS of single_view_optimization_test) follows below: spsp_silhouette_loss_grad. */
typedef struct spsp_rasterizer spsp_rasterizer;
/* faces: host [F,3] int32 vertex ids (smpl.faces, 13776 rows); img_wh: 512; workspace for max_batch meshes per call. */
int  spsp_rasterizer_load(spsp_ctx* ctx, const int32_t* faces, int F, int max_batch, int img_wh, spsp_rasterizer** out);
void spsp_rasterizer_free(spsp_rasterizer* h);
/* verts [B,6890,3], translation [B,3] (camera translation [tx,ty,tz]), focal (5000) -> image [B,img_wh,img_wh] fp32 in
 * {0,.25,.5,.75,1} (NULL to skip) and / or metric [B] = -sum_px [p ln(t+1e-6) + (1-p) ln(1-t+1e-6)] against the detected
 * silhouette target (uint8 0/1; [B,img_wh,img_wh], or one [img_wh,img_wh] for all meshes when target_shared != 0). */
int  spsp_render_silhouettes(spsp_ctx* ctx, spsp_rasterizer* h, const float* verts, const float* translation, float focal, int B,
                             const uint8_t* target, int target_shared, float* image, float* metric, void* stream);
/* The reference's rule over a recorded trajectory: iterate it becomes the best iff l2e_hist[b][it] <= its value at the previous
 * best AND (sil_hist given) sil_hist[b][it] <= likewise; the best iterate's record is scattered into R / betas / cam;
 * best_iter [B] optional. */
int  spsp_fit_select(spsp_ctx* ctx, const float* l2e_hist, const float* sil_hist, int iters, int B, const float* param_hist, float* R,
                     float* betas, float* cam, int32_t* best_iter, void* stream);

/* ---- the differentiable half of the silhouette term: the loss of single_view_optimization_test
 *      (PR/player_recon.py:1429-1557; PR/losses/multi_task_loss.py:132-144) and neural_renderer's approximate backward pass,
 *      restated (un-vendored dependency: parity unpinned; csrc/silhouette_grad.hip, oracle/silhouette_grad.py).
 *      verts [B,6890,3], translation [B,3], target uint8 0/1 ([B,wh,wh], or one [wh,wh] when target_shared), weight [B] =
 *      exp(-silhouette_log_var) (required with grad_verts).  Out (each optional): sil_sum [B] = sum_px [p ln(t + 1e-6) +
 *      (1 - p) ln(1 - t + 1e-6)]; grad_verts [B,6890,3] and grad_translation [B,3] of L = -sil_sum * weight; image [B,wh,wh]
 *      (equal to spsp_render_silhouettes': the ownership raster and the coverage raster agree). */
int  spsp_silhouette_loss_grad(spsp_ctx* ctx, spsp_rasterizer* h, const float* verts, const float* translation, float focal, int B,
                               const uint8_t* target, int target_shared, const float* weight, float* sil_sum, float* grad_verts,
                               float* grad_translation, float* image, void* stream);
/* Reverse mode of spsp_smpl_forward's vertices: grad_verts [B,6890,3] -> grad_R [B,24,9], grad_betas [B,10] at (R, betas). */
int  spsp_smpl_vertices_vjp(spsp_ctx* ctx, spsp_smpl* s, const float* R, const float* betas, int B, const float* grad_verts,
                            float* grad_R, float* grad_betas, void* stream);
/* The fit with the silhouette IN the loss (single_view_optimization_test): parameters {global_orient, body_pose idx {0-5, 8-20},
 * cam, betas, joints2D_log_var, silhouette_log_var}, torch.optim.Adam, `iters` epochs, the LAST iterate left in place.
 * log_vars [B,2] in/out (the caller initialises them to -ln(weight + 1e-6), multi_task_loss.py:37-46); histories [B,iters] optional:
 * loss, joints2D_l2es, the silhouette metric (-sil_sum); param_hist [B,iters,231] = [cam | R | betas | log_vars] each epoch
 * evaluated, grad_hist [B,iters,231] the gradient of the loss there (entries of frozen parameters are not used).  Every epoch is a
 * chain of launches on `stream`; no host synchronisation. */
typedef struct {
    int      iters;                              /* 50 (PR/player_recon.py:1457) */
    double   lr, beta1, beta2, eps;              /* 1e-3 (S/global_var.py:71), torch.optim.Adam defaults */
    double   joints2d_weight, silhouette_weight; /* 1.0, 1e6 (PR/player_recon.py:851) */
    uint32_t param_mask;                         /* SPSP_FIT_MASK_POSE_SHAPE */
    float    img_wh, loss_wh, focal;             /* 512, 256, 5000 */
} spsp_fit_silhouette_config;
void spsp_fit_silhouette_config_default(spsp_fit_silhouette_config* c);
int  spsp_fit_silhouette(spsp_ctx* ctx, spsp_smpl* s, spsp_rasterizer* rast, const spsp_fit_silhouette_config* cfg, float* R, float* betas,
                         float* cam, float* log_vars, const float* joints2d, const uint8_t* target, int target_shared, int B,
                         float* loss_hist, float* l2e_hist, float* sil_hist, float* param_hist, float* grad_hist, void* stream);

/* ---- multi-view joint fit: replaces the optimisation body of multi_view_optimization (PR/player_recon.py:1691-1979):
 *      V <= 4 views of one player share body pose and betas, each view has its camera and global orientation.
 *      `rounds` x { block 1: Adam over the stacked per-view cameras + orientations; block 2: Adam over the shared
 *      parameters }, `iters` epochs per block, an epoch = one train step per view + a validation pass; the iterate with
 *      the lowest mean joints2D_l2es is kept (per block and as the global record that is returned).  optimise_body_pose
 *      = 0 reproduces the reference exactly (its block 2 reaches betas only: the posed tensor is concatenated before
 *      requires_grad is set, :1724-1731); 1 also optimises body_pose idx {0-5, 8-20} as the loop intends.  This entry visits the
 *      views in order 0..V-1 and keeps the record by joints2D_l2es alone; spsp_fit_multiview_ordered takes the reference's
 *      shuffled orders, spsp_fit_multiview_block serves its two-metric record rule. */
typedef struct {
    int    iters;               /* 50 epochs per block (S/global_var.py:79) */
    int    rounds;              /* 3 (PR/player_recon.py:1720) */
    double lr;                  /* 1e-3 (S/global_var.py:71) */
    double beta1, beta2, eps;   /* torch.optim.Adam defaults */
    double loss_weight;         /* 1.0 */
    float  img_wh, loss_wh;     /* 512, 256 */
    int    optimise_body_pose;  /* 0 = as the reference behaves, 1 = as it intends */
} spsp_fit_multiview_config;
void spsp_fit_multiview_config_default(spsp_fit_multiview_config* c);
/* In place: body_pose [B,23,9] (in: the mean of the views' single-view results, :1691-1694), betas [B,10] (in: mean),
 * cam [B,V,3], global_orient [B,V,9] (in: per view), joints2d [B,V,17,2] pixels; out: the best record.
 * val_hist: optional [B, rounds*2*iters] mean joints2D_l2es of every epoch's validation pass. */
int spsp_fit_multiview(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_multiview_config* cfg, int B, int V, float* body_pose,
                       float* betas, float* cam, float* global_orient, const float* joints2d, float* val_hist, void* stream);

/* The same with the views visited in a given order: view_order [rounds * 2 * iters][V] int32 (device), row = the train pass of
 * (round, block, epoch) -- the reference shuffles the views before every pass (`random.shuffle`, PR/player_recon.py:1748, 1876);
 * NULL = 0 .. V-1. */
int spsp_fit_multiview_ordered(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_multiview_config* cfg, int B, int V, float* body_pose,
                               float* betas, float* cam, float* global_orient, const float* joints2d, const int32_t* view_order,
                               float* val_hist, void* stream);
/* ONE block (0: Adam over the stacked cameras + orientations, 1: Adam over the shared pose / betas; a fresh optimiser) of
 * cfg->iters epochs from the given state, WITHOUT record keeping: in place the block's LAST iterate; val_l2e [B,iters] the mean
 * joints2D_l2es of every epoch's validation pass; snap_views [B,iters,V,12] ([cam | orientation] per view) and snap_shared
 * [B,iters,217] ([body pose (207) | betas]) the parameters every epoch's validation pass saw.  For the reference's record rule over
 * BOTH tracked metrics (determine_save_model_weights_this_epoch with joints2D_l2es and the rendered-silhouette metric,
 * PR/player_recon.py:1820, 1937): the caller renders the snapshots, decides, and starts the next block from the block's best epoch
 * (Python: player_recon.multi_view_optimization(select_mode='reference')).  view_order: [iters][V] or NULL. */
int spsp_fit_multiview_block(spsp_ctx* ctx, spsp_smpl* s, const spsp_fit_multiview_config* cfg, int block, int B, int V, float* body_pose,
                             float* betas, float* cam, float* global_orient, const float* joints2d, const int32_t* view_order,
                             float* val_l2e, float* snap_views, float* snap_shared, void* stream);

/* ---- exchange step without torch: ONE RCCL all-gather of the packed records over xGMI, for hosts that
 *      drive this library directly.  Rank 0 calls spsp_comm_unique_id and ships the 128 bytes to the other ranks
 *      (any channel); every rank calls spsp_comm_init_rank on its own device; spsp_allgather_records gathers
 *      n_per_rank records of every rank into all [nranks * n_per_rank, 229] in rank order, asynchronously on `stream`
 *      (ragged frames: pad the last shards, as soccerplayershapepose_amd.sharding does).  RCCL is resolved at run
 *      time; inside a PyTorch process the Python mirror uses torch.distributed for the same collective. */
typedef struct spsp_comm spsp_comm;
int  spsp_comm_unique_id(void* id128 /* host, 128 bytes */);
int  spsp_comm_init_rank(spsp_ctx* ctx, const void* id128, int nranks, int rank, spsp_comm** out);
void spsp_comm_destroy(spsp_comm* comm);
int  spsp_allgather_records(spsp_ctx* ctx, spsp_comm* comm, const float* local /* [n_per_rank,229] */, int n_per_rank,
                            float* all /* [nranks*n_per_rank,229] */, void* stream);

/* ---- the whole path in one call: the core of evaluate_model_2d (PR/PyTorch3DTest.py:3738-3776) ----------------
 * proxy -> regressor -> rot6d -> [relation module, if `relate`] -> [fit, if `fit`] -> SMPL forward -> translation
 * [-> packed record].  Issues the stage launches in order on `stream`; every buffer is a caller-owned device
 * buffer.  cam / pose6d / betas / rotmats are required; vertices, joints, translation, record may be NULL (skipped);
 * relate_tmp [B,216] is required with `relate`.  With a fit, cam / rotmats / betas hold the FITTED values. */
typedef struct {
    float* cam;          /* [B,3]        */
    float* pose6d;       /* [B,144]      regressor output (not touched by the fit) */
    float* betas;        /* [B,10]       */
    float* rotmats;      /* [B,24,3,3]   */
    float* vertices;     /* [B,6890,3]   or NULL */
    float* joints;       /* [B,90,3]     or NULL */
    float* translation;  /* [B,3]        or NULL */
    float* record;       /* [B,229]      or NULL: [cam | R | betas], the all-gather payload */
    float* relate_tmp;   /* [B,216]      scratch for the relation module, or NULL */
} spsp_pipeline_out;
int spsp_pipeline(spsp_ctx* ctx, spsp_regressor* regressor, spsp_smpl* smpl, spsp_relate* relate, const void* proxy,
                  int proxy_format, const float* boxes, const float* joints2d, const spsp_fit_config* fit, int B,
                  float focal, float resolution, const spsp_pipeline_out* out, void* stream);
/* The same path for `ngroups` consecutive batches of B crops each with ONE tail: the backbone runs batch by batch (the launches
 * of spsp_pipeline at batch B, proxies[g] -> feature rows g*B ..), then IEF -> fit -> SMPL forward run ONCE over all
 * ngroups*B players.  Those stages are latency-bound chains of small launches (one workgroup per player in the fit) whose
 * duration does not depend on the number of players, so a stream spends half (a third, ...) as long in them per batch; every
 * player's results are bit-identical to spsp_pipeline's.  joints2d and every buffer of `out` hold ngroups*B players (group g at
 * rows g*B ..); both handles need max_batch >= ngroups*B (<= 256).  No relation module in this form. */
int spsp_pipeline_group(spsp_ctx* ctx, spsp_regressor* regressor, spsp_smpl* smpl, const void* const* proxies, int ngroups,
                        int proxy_format, const float* joints2d, const spsp_fit_config* fit, int B, float focal,
                        float resolution, const spsp_pipeline_out* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPSP_H */
