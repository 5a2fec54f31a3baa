// Wave-cooperative device helpers shared by the SMPL chain kernel and the fit kernel.
// All arrays live in LDS; a workgroup is exactly ONE 64-lane wavefront, so __syncthreads() is a
// single-wave barrier that orders LDS writes against the other lanes' reads.
#pragma once
#include <hip/hip_runtime.h>

namespace spsp {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct TreeLds {
    int parents[24];
    int level_joints[24];
    int level_start[26];
    int num_levels;
};

__device__ __forceinline__ void load_tree(TreeLds& t, const int* parents, const int* level_joints,
                                          const int* level_start, int num_levels, int lane) {
    if (lane < 24) {
        t.parents[lane] = parents[lane];
        t.level_joints[lane] = level_joints[lane];
    }
    if (lane < 26) t.level_start[lane] = level_start[lane];
    if (lane == 0) t.num_levels = num_levels;
}

// J[k][c] = Jbar[k][c] + sum_i JS[i][k*3+c] * beta[i]      (72 outputs; Jbar/JS precomputed at load
// as J_regressor @ v_template and J_regressor @ shapedirs -- smplx: J = J_regressor @ v_shaped)
__device__ __forceinline__ void joints_from_betas(const float* Jbar, const float* JS, const float* beta,
                                                  float* J, int lane) {
    for (int o = lane; o < 72; o += 64) {
        float acc = Jbar[o];
#pragma unroll
        for (int i = 0; i < 10; ++i) acc = fmaf(JS[i * 72 + o], beta[i], acc);
        J[o] = acc;
    }
}

// Orders the LDS writes of one phase against the reads of the next.  WAVE = false: the workgroup is ONE wavefront and
// __syncthreads() is that wave's barrier; WAVE = true: the caller is one wavefront of a larger workgroup working on LDS of its
// own -- a wave issues its LDS instructions in order, so only the COMPILER has to be kept from moving a read above another
// lane's write: wavefront-scope release / acquire around a wave barrier (no instruction is emitted for it; fit.hip).
template <bool WAVE>
__device__ __forceinline__ void lds_phase_sync() {
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// Kinematic chain (smplx batch_rigid_transform): W_0 = [R_0 | J_0], W_k = W_parent * [R_k | J_k - J_parent].
// Rm [24][9] row-major, J [24][3], W [24][12] row-major 3x4.  Level-parallel: 5 joints x 12 elements per pass.
template <bool WAVE = false>
__device__ __forceinline__ void chain_forward(const float* Rm, const float* J, float* W, const TreeLds& t, int lane) {
    if (lane < 12) {
        int i = lane >> 2, c = lane & 3;
        W[lane] = (c < 3) ? Rm[i * 3 + c] : J[i];
    }
    lds_phase_sync<WAVE>();
    const int slot = lane / 12, e = lane % 12;
    const int i = e >> 2, c = e & 3;
    for (int l = 1; l < t.num_levels; ++l) {
        const int beg = t.level_start[l], end = t.level_start[l + 1];
        if (slot < 5) {
            for (int idx = beg + slot; idx < end; idx += 5) {
                const int k = t.level_joints[idx];
                const int p = t.parents[k];
                const float* Wp = W + p * 12;
                float acc;
                if (c < 3) {
                    acc = Wp[i * 4 + 0] * Rm[k * 9 + 0 + c];
                    acc = fmaf(Wp[i * 4 + 1], Rm[k * 9 + 3 + c], acc);
                    acc = fmaf(Wp[i * 4 + 2], Rm[k * 9 + 6 + c], acc);
                } else {
                    acc = Wp[i * 4 + 0] * (J[k * 3 + 0] - J[p * 3 + 0]);
                    acc = fmaf(Wp[i * 4 + 1], J[k * 3 + 1] - J[p * 3 + 1], acc);
                    acc = fmaf(Wp[i * 4 + 2], J[k * 3 + 2] - J[p * 3 + 2], acc);
                    acc += Wp[i * 4 + 3];
                }
                W[k * 12 + e] = acc;
            }
        }
        lds_phase_sync<WAVE>();
    }
}

}  // namespace spsp
