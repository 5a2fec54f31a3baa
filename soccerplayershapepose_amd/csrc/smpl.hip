// SMPL forward for gfx950: blend shapes + joint regression + kinematic chain + linear-blend skinning
// + the 90-joint set of the reference wrapper.  Replaces SMPL.forward (PR/models/smpl_official.py:27-41
// -> smplx.lbs.lbs [EXT]).  VALU fp32 on purpose: the contraction is [B<=32..256] x 217 -- far too thin
// for MFMA; the kernel is bound by streaming the 17.9 MB blend basis (HBM / Infinity Cache).
//
//   skin kernel    one thread per vertex, PB players per workgroup kept in registers so every basis row is
//                  read ONCE per PB players; coef = [betas | R[1:]-I] and the transforms A in LDS (broadcast reads);
//                  joint-major LBS weights for coalesced loads.  Eight more waves per workgroup compute the players'
//                  kinematic chains meanwhile (J = Jbar + JS.betas, level-parallel chain in LDS, rel. transforms
//                  A[24][3x4], posed joints 0..23) -- a kernel of its own until round 4.
//   joints kernel  21 picked vertices + 45 regressed joints (CSR of the dense regressor rows) from the
//                  FINAL vertices, as the reference does (smpl_official.py:30-34).
#include <algorithm>
#include <cmath>

#include "smpl.hpp"
#include "smpl_device.hpp"
#include "smpl_host.hpp"

namespace {
using namespace spsp;

constexpr int kSkinThreads = 256;     // the four accumulating / skinning waves
constexpr int kChainWaves = 4;        // + four waves that turn (R, betas) of the workgroup's players into the skinning transforms, two
                                      // players each, one after the other (a chain is ~2.5 us of dependent LDS round trips: two fit
                                      // under the blend phase, four -- two chain waves -- do not: 27.9 instead of 22.9 us).  One wave
                                      // PER PLAYER (12 waves, 62 KB of LDS: two workgroups per CU) is as fast alone and cost the four-slot
                                      // pipeline 1 %: a workgroup that size finds room beside the other slots' convolutions less often
constexpr int kSkinBlock = kSkinThreads + 64 * kChainWaves;
// players per skin workgroup: 8.  16 (the blend basis read twice instead of four times per 32 players: `spsp_smpl_debug(
// SPSP_DEBUG_SKIN_PLAYERS, 16)`) measured slower in round 3 (27.7 vs 19.0 us: 236 VGPRs and 82 KB of LDS leave one workgroup of four
// waves per CU) and again in round 5 with the chain waves in the kernel (bytes and time)
constexpr int kPBDefault = 8;

// Per chain wave: its players' rotation matrices and betas (fetched before the workgroup's first barrier) and one player's joints.
template <int CP>
struct ChainLds {
    float R[CP][216], beta[CP][16], J[72];
};

// One wavefront, one player: J = Jbar + JS . betas, the level-parallel kinematic chain, A_k = [W_k.R | W_k.t - W_k.R J_k]
// (smplx batch_rigid_transform); the posed joints 0..23 are W_k.t.  Until round 4 this was a kernel of its own (one wavefront
// per player, 10.6 us + a launch boundary in front of the skin kernel -- part of the serial tail every pipeline slot waits
// for); now four extra waves of every skin workgroup run it for the workgroup's players WHILE the other four stream the blend
// basis.  Same statements in the same order as before: bit-identical transforms.  The world transforms are built IN the
// player's slot of sA (same 3 x 4 layout) and only their translation column is rewritten at the end.  Everything read from
// global memory (R, betas, the joint constants, the tree) is fetched BEFORE the workgroup's first barrier: a dependent global
// round trip behind the blend loop's loads takes microseconds (first version: the skin kernel 19.5 -> 29 us).
__device__ __forceinline__ void chain_one_player(const float* Rm, const float* beta, float* J, const float* sJc, const TreeLds& tree,
                                                 float* A_lds, float* A_out, float* joints_out, int lane) {
    joints_from_betas(sJc, sJc + 72, beta, J, lane);
    lds_phase_sync<true>();
    chain_forward<true>(Rm, J, A_lds, tree, lane);
    for (int o = lane; o < 72; o += 64) {
        const int k = o / 3, i = o % 3;
        float* w = A_lds + k * 12 + i * 4;
        const float w3 = w[3];
        const float t = w3 - (w[0] * J[k * 3] + w[1] * J[k * 3 + 1] + w[2] * J[k * 3 + 2]);
        w[3] = t;
        if (A_out) { float* g = A_out + k * 12 + i * 4; g[0] = w[0]; g[1] = w[1]; g[2] = w[2]; g[3] = t; }
        if (joints_out) joints_out[k * 3 + i] = w3;
    }
    lds_phase_sync<true>();                                // (J is reused by the wave's next player)
}

// Workgroup = 4 waves over the SAME 64 vertices and kPB players: wave w accumulates rows [55w, 55w+55) of the
// 217-row blend basis (split-K: 4x more waves and 4x shorter dependent-load chains at no extra basis traffic),
// partial sums meet in LDS, then wave w skins players (kPB / 4) w .. + kPB / 4 - 1.
// Waves 4..11 (round 4): the kinematic chain of one of the workgroup's players each (chain_one_player), into sA -- and, from the
// workgroups of the first vertex tile, into the handle's A workspace (the VJP reads it) and joints 0..23.
template <int kPB>
__global__ __launch_bounds__(kSkinBlock) void smpl_skin_kernel(const float* __restrict__ basis,
                                                               const float* __restrict__ v_template,
                                                               const float* __restrict__ wT,
                                                               const float* __restrict__ R, const float* __restrict__ betas,
                                                               const float* __restrict__ jconst,
                                                               const int* __restrict__ parents, const int* __restrict__ level_joints,
                                                               const int* __restrict__ level_start, int num_levels,
                                                               float* __restrict__ A_ws, float* __restrict__ joints_out, int B,
                                                               float* __restrict__ verts, float* __restrict__ vposed) {
    constexpr int kChainPlayers = kPB / kChainWaves;
    static_assert(kPB % kChainWaves == 0, "players split evenly over the chain waves");
    typedef ChainLds<kChainPlayers> ChainLdsT;
    __shared__ float sCoef[kPB][kNCoef + 3];
    __shared__ __attribute__((aligned(16))) float sA[kPB][288];
    __shared__ float sPart[4][kPB * 3][64];                // [k-slice][player*3 + c][lane]
    __shared__ ChainLdsT sChain[kChainWaves];
    __shared__ TreeLds sTree;
    __shared__ float sJc[792];
    const int lane = threadIdx.x & 63, ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int vtile = (int)blockIdx.x, grp = (int)blockIdx.y;
    const int b0 = grp * kPB;
    const int nb = min(kPB, B - b0);
    if (ks >= 4) {
        ChainLdsT& c = sChain[ks - 4];
        if (ks == 4) load_tree(sTree, parents, level_joints, level_start, num_levels, lane);
        for (int i = threadIdx.x - kSkinThreads; i < 792; i += 64 * kChainWaves) sJc[i] = jconst[i];
#pragma unroll
        for (int q = 0; q < kChainPlayers; ++q) {
            const int p = (ks - 4) * kChainPlayers + q;
            const size_t b = b0 + (p < nb ? p : 0);
            for (int i = lane; i < 216; i += 64) c.R[q][i] = R[b * 216 + i];
            if (lane < 10) c.beta[q][lane] = betas[b * 10 + lane];
        }
        __syncthreads();                                   // (the staging barrier of the other waves)
        const bool first = vtile == 0;
        for (int q = 0; q < kChainPlayers; ++q) {
            const int p = (ks - 4) * kChainPlayers + q;
            if (p < nb) {
                const size_t b = b0 + p;
                chain_one_player(c.R[q], c.beta[q], c.J, sJc, sTree, &sA[p][0], first ? A_ws + b * 288 : nullptr,
                                 (first && joints_out) ? joints_out + b * SPSP_NUM_OUT_JOINTS * 3 : nullptr, lane);
            } else {
                for (int i = lane; i < 288; i += 64) sA[p][i] = 0.f;
            }
        }
        __syncthreads();                                   // (the barrier in front of the skinning: sA is complete)
        return;
    }
    // blend coefficients: betas, then the pose feature (R[1:] - I) flattened row-major (smplx lbs, pose2rot=False)
    for (int i = threadIdx.x; i < kPB * kNCoef; i += kSkinThreads) {
        const int p = i / kNCoef, j = i % kNCoef;
        float v = 0.f;
        if (p < nb) {
            if (j < 10) v = betas[(size_t)(b0 + p) * 10 + j];
            else {
                const int jj = j - 10, e = jj % 9;
                v = R[(size_t)(b0 + p) * 216 + 9 + jj] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
            }
        }
        sCoef[p][j] = v;
    }
    __syncthreads();
    const int v = min(vtile * 64 + lane, kV - 1);
    float px[kPB], py[kPB], pz[kPB];
#pragma unroll
    for (int p = 0; p < kPB; ++p) px[p] = py[p] = pz[p] = 0.f;
    constexpr int kRows = (kNCoef + 3) / 4;                 // 55 rows per slice (last slice: 52)
    const int j0 = ks * kRows, j1 = min(j0 + kRows, kNCoef);
    // (measured in round 4, all within 1 us of each other: coordinate PLANES of the basis -- 256 contiguous bytes per load instead
    //  of every third float of 768 --, the player groups of a vertex tile on one XCD, 11 or 18 rows per unrolled batch; the
    //  joint loop below fully unrolled spilled: 64 us)
    const float* bp = basis + 3 * v;
#pragma unroll 5
    for (int j = j0; j < j1; ++j) {
        const float d0 = bp[(size_t)j * kV3], d1 = bp[(size_t)j * kV3 + 1], d2 = bp[(size_t)j * kV3 + 2];
#pragma unroll
        for (int p = 0; p < kPB; ++p) {
            const float c = sCoef[p][j];
            px[p] = fmaf(c, d0, px[p]);
            py[p] = fmaf(c, d1, py[p]);
            pz[p] = fmaf(c, d2, pz[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < kPB; ++p) {
        sPart[ks][p * 3 + 0][lane] = px[p];
        sPart[ks][p * 3 + 1][lane] = py[p];
        sPart[ks][p * 3 + 2][lane] = pz[p];
    }
    __syncthreads();
    // skinning: this wave takes players NQ ks .. NQ ks + NQ - 1.  T = sum_k w[v][k] A_k (3x4), vert = T [v_posed; 1]
    constexpr int NQ = kPB / 4;
    const bool vok = vtile * 64 + lane < kV;
    float qx[NQ], qy[NQ], qz[NQ], ox[NQ], oy[NQ], oz[NQ];
    const float t0 = v_template[3 * v], t1 = v_template[3 * v + 1], t2 = v_template[3 * v + 2];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int p = ks * NQ + q;
        qx[q] = t0 + ((sPart[0][p * 3][lane] + sPart[1][p * 3][lane]) + (sPart[2][p * 3][lane] + sPart[3][p * 3][lane]));
        qy[q] = t1 + ((sPart[0][p * 3 + 1][lane] + sPart[1][p * 3 + 1][lane]) + (sPart[2][p * 3 + 1][lane] + sPart[3][p * 3 + 1][lane]));
        qz[q] = t2 + ((sPart[0][p * 3 + 2][lane] + sPart[1][p * 3 + 2][lane]) + (sPart[2][p * 3 + 2][lane] + sPart[3][p * 3 + 2][lane]));
        ox[q] = oy[q] = oz[q] = 0.f;
        // (the differentiable path keeps the posed rest vertex: d vertex / d A_k needs it, smpl_vjp.hip)
        if (vposed && p < nb && vok) {
            float* o = vposed + ((size_t)(b0 + p) * kV + v) * 3;
            o[0] = qx[q]; o[1] = qy[q]; o[2] = qz[q];
        }
    }
#pragma unroll 4
    for (int k = 0; k < kNJ; ++k) {
        const float w = wT[(size_t)k * kV + v];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4* a = reinterpret_cast<const float4*>(&sA[ks * NQ + q][k * 12]);
            const float4 r0 = a[0], r1 = a[1], r2 = a[2];
            const float rx = fmaf(r0.x, qx[q], fmaf(r0.y, qy[q], fmaf(r0.z, qz[q], r0.w)));
            const float ry = fmaf(r1.x, qx[q], fmaf(r1.y, qy[q], fmaf(r1.z, qz[q], r1.w)));
            const float rz = fmaf(r2.x, qx[q], fmaf(r2.y, qy[q], fmaf(r2.z, qz[q], r2.w)));
            ox[q] = fmaf(w, rx, ox[q]);
            oy[q] = fmaf(w, ry, oy[q]);
            oz[q] = fmaf(w, rz, oz[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int p = ks * NQ + q;
        if (p < nb && vok) {
            float* o = verts + ((size_t)(b0 + p) * kV + v) * 3;
            o[0] = ox[q]; o[1] = oy[q]; o[2] = oz[q];
        }
    }
}

// joints 24..44: picked vertices; joints 45..89: sparse regressor rows applied to the final vertices.
__global__ __launch_bounds__(256) void smpl_extra_joints_kernel(const float* __restrict__ verts,
                                                                const int* __restrict__ pick_ids,
                                                                const int* __restrict__ csr_ptr,
                                                                const int* __restrict__ csr_col,
                                                                const float* __restrict__ csr_val,
                                                                float* __restrict__ joints) {
    // grid (players, 12): ONE regressor row per wavefront (round 3; one workgroup per player walked its 45 rows four at a time,
    // 20 us of dependent gathers for 32 players -- part of the serial tail every pipeline slot waits for)
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float* vb = verts + (size_t)b * kV3;
    float* jb = joints + (size_t)b * SPSP_NUM_OUT_JOINTS * 3;
    if (blockIdx.y == 0 && threadIdx.x < 63) {
        int j = threadIdx.x / 3, c = threadIdx.x % 3;
        jb[(24 + j) * 3 + c] = vb[pick_ids[j] * 3 + c];
    }
    for (int r = blockIdx.y * 4 + wave; r < 45; r += 48) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int e = csr_ptr[r] + lane; e < csr_ptr[r + 1]; e += 64) {
            const float w = csr_val[e];
            const float* p = vb + csr_col[e] * 3;
            sx = fmaf(w, p[0], sx);
            sy = fmaf(w, p[1], sy);
            sz = fmaf(w, p[2], sz);
        }
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
        if (lane == 0) {
            jb[(45 + r) * 3 + 0] = sx;
            jb[(45 + r) * 3 + 1] = sy;
            jb[(45 + r) * 3 + 2] = sz;
        }
    }
}

}  // namespace

extern "C" {

int spsp_smpl_load(spsp_ctx* ctx, const spsp_smpl_arrays* a, int max_batch, spsp_smpl** out) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && a && out && max_batch > 0, "spsp_smpl_load: bad arguments");
    SmplHostImage h;
    const std::string err = smpl_prepare(*a, h);
    SPSP_REQUIRE(err.empty(), "spsp_smpl_load: %s", err.c_str());
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    spsp_smpl* s = new spsp_smpl;
    s->max_batch = max_batch;
    s->num_levels = h.num_levels;
    std::memcpy(s->level_start, h.level_start, sizeof h.level_start);
    std::memcpy(s->host_parents, h.parents, sizeof h.parents);
    std::memcpy(s->face_ids, h.face_ids, sizeof h.face_ids);
    int rc = SPSP_OK;
#define TRY(x) do { rc = (x); if (rc != SPSP_OK) { delete s; return rc; } } while (0)
    TRY(s->pool.upload(&s->basis, h.basis.data(), h.basis.size()));
    TRY(s->pool.upload(&s->v_template, a->v_template, (size_t)kV3));
    TRY(s->pool.upload(&s->wT, h.wT.data(), h.wT.size()));
    TRY(s->pool.upload(&s->jconst, h.jconst.data(), h.jconst.size()));
    TRY(s->pool.upload(&s->parents, h.parents, (size_t)kNJ));
    TRY(s->pool.upload(&s->level_joints, h.level_joints.data(), h.level_joints.size()));
    TRY(s->pool.upload(&s->d_level_start, h.level_start, (size_t)26));
    TRY(s->pool.upload(&s->child_start, h.child_start, (size_t)25));
    TRY(s->pool.upload(&s->child_list, h.child_list, (size_t)24));
    TRY(s->pool.upload(&s->walk, h.walk, (size_t)(64 * 4 + 25)));
    s->walk_ok = h.walk_ok;
    s->walk_nb = h.walk_nb;
    TRY(s->pool.upload(&s->pick_ids, a->vertex_joint_ids, (size_t)21));
    TRY(s->pool.upload(&s->csr_ptr, h.csr_ptr.data(), h.csr_ptr.size()));
    TRY(s->pool.upload(&s->csr_col, h.csr_col.data(), h.csr_col.size()));
    TRY(s->pool.upload(&s->csr_val, h.csr_val.data(), h.csr_val.size()));
    TRY(s->pool.upload(&s->fitc, h.fitc.data(), h.fitc.size()));
    TRY(s->pool.alloc(&s->ws_A, (size_t)max_batch * 288));
    TRY(s->pool.alloc(&s->ws_verts, (size_t)max_batch * kV3));
#undef TRY
    *out = s;
    return SPSP_OK;
}

void spsp_smpl_free(spsp_smpl* s) { delete s; }

int spsp_smpl_forward(spsp_ctx* ctx, spsp_smpl* s, const float* R, const float* betas, int B, float* verts,
                      float* joints, void* stream) {
    return spsp::smpl_forward_impl(ctx, s, R, betas, B, verts, joints, nullptr, stream);
}

}  // extern "C"

namespace spsp {
// vposed: optional [B,6890,3] -- the posed REST vertices (template + blend shapes) the skinning multiplies, kept for the VJP
int smpl_forward_impl(spsp_ctx* ctx, spsp_smpl* s, const float* R, const float* betas, int B, float* verts, float* joints,
                      float* vposed, void* stream) {
    SPSP_REQUIRE(ctx && s, "spsp_smpl_forward: NULL handle");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));      // a process may hold contexts on several GPUs
    if (B == 0) return SPSP_OK;          // empty batch: nothing to do (pointers of empty tensors may be NULL)
    SPSP_REQUIRE(R && betas, "spsp_smpl_forward: NULL argument");
    SPSP_REQUIRE(B >= 0 && B <= s->max_batch, "spsp_smpl_forward: B=%d exceeds max_batch=%d", B, s->max_batch);
    if (B == 0 || (!verts && !joints)) return SPSP_OK;
    hipStream_t st = as_stream(stream);
    float* v = verts ? verts : s->ws_verts;
    const int pb = s->skin_players == 16 ? 16 : kPBDefault;
    dim3 grid((kV + 63) / 64, (B + pb - 1) / pb);
    { ProfScope ps(ctx, kKSmplSkin, st, true);
      if (pb == 16)
          launch_kernel(&smpl_skin_kernel<16>, grid, dim3(kSkinBlock), 0, st, s->basis, s->v_template, s->wT, R, betas, s->jconst, s->parents,
                        s->level_joints, s->d_level_start, s->num_levels, s->ws_A, joints, B, v, vposed);
      else
          launch_kernel(&smpl_skin_kernel<kPBDefault>, grid, dim3(kSkinBlock), 0, st, s->basis, s->v_template, s->wT, R, betas, s->jconst, s->parents,
                        s->level_joints, s->d_level_start, s->num_levels, s->ws_A, joints, B, v, vposed); }
    SPSP_HIP_CHECK(hipGetLastError());
    if (joints) {
        ProfScope ps(ctx, kKSmplJoints, st, true);
        launch_kernel(&smpl_extra_joints_kernel, dim3(B, 12), dim3(256), 0, st, v, s->pick_ids, s->csr_ptr, s->csr_col, s->csr_val, joints);
        SPSP_HIP_CHECK(hipGetLastError());
    }
    return SPSP_OK;
}

}  // namespace spsp

