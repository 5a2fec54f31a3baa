// Small per-player geometry kernels: rot6d -> rotmat, weak-perspective projection, camera
// conversion, record pack/unpack.  All are trivially HBM/latency bound (SURVEY 8a a5,a7,a12).
#include "common.hpp"
#include "geometry_device.hpp"

namespace {

// rot6d_to_rotmat, PR/utils/rigid_transform_utils.py:27-41 (the formula: geometry_device.hpp)
__global__ void rot6d_kernel(const float* __restrict__ x, int n, float* __restrict__ R) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    spsp::rot6d_one(x + (size_t)i * 6, R + (size_t)i * 9);
}

// orthographic_project_torch (+ undo_keypoint_normalisation), PR/utils/cam_utils.py:5-26.
__global__ void project_kernel(const float* __restrict__ pts, const float* __restrict__ cam, int n,
                               float img_wh, float* __restrict__ out) {
    int b = blockIdx.y;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    const float* p = pts + ((size_t)b * n + i) * 3;
    float u = s * (p[0] + tx), v = s * (p[1] + ty);
    if (img_wh > 0.f) {
        float h = img_wh / 2.0f;
        u = (u + 1.f) * h;
        v = (v + 1.f) * h;
    }
    float* o = out + ((size_t)b * n + i) * 2;
    o[0] = u;
    o[1] = v;
}

// convert_weak_perspective_to_camera_translation_torch, PR/utils/cam_utils.py:28-34.
__global__ void wp2t_kernel(const float* __restrict__ cam, int B, float focal, float res, float* __restrict__ t) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[b * 3 + i] = spsp::wp_translation(s, tx, ty, focal, res, i);
}

__global__ void pack_record_kernel(const float* __restrict__ cam, const float* __restrict__ R,
                                   const float* __restrict__ betas, int B, float* __restrict__ rec) {
    int b = blockIdx.x;
    for (int i = threadIdx.x; i < SPSP_RECORD_FLOATS; i += blockDim.x) {
        float v = i < 3 ? cam[b * 3 + i] : (i < 219 ? R[(size_t)b * 216 + (i - 3)] : betas[b * 10 + (i - 219)]);
        rec[(size_t)b * SPSP_RECORD_FLOATS + i] = v;
    }
}

__global__ void unpack_record_kernel(const float* __restrict__ rec, int B, float* __restrict__ cam,
                                     float* __restrict__ R, float* __restrict__ betas) {
    int b = blockIdx.x;
    for (int i = threadIdx.x; i < SPSP_RECORD_FLOATS; i += blockDim.x) {
        float v = rec[(size_t)b * SPSP_RECORD_FLOATS + i];
        if (i < 3) cam[b * 3 + i] = v;
        else if (i < 219) R[(size_t)b * 216 + (i - 3)] = v;
        else betas[b * 10 + (i - 219)] = v;
    }
}

}  // namespace

extern "C" {

int spsp_rot6d_to_rotmat(spsp_ctx* ctx, const float* pose6d, int n, float* R, void* stream) {
    SPSP_REQUIRE(ctx && pose6d && R && n >= 0, "spsp_rot6d_to_rotmat: bad arguments");
    if (n == 0) return SPSP_OK;
    spsp::ProfScope ps(ctx, spsp::kKRot6d, spsp::as_stream(stream), true);
    spsp::launch_kernel(&rot6d_kernel, dim3((n + 255) / 256), dim3(256), 0, spsp::as_stream(stream), pose6d, n, R);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

int spsp_project(spsp_ctx* ctx, const float* pts, const float* cam, int B, int n, float img_wh, float* out,
                 void* stream) {
    SPSP_REQUIRE(ctx && pts && cam && out && B >= 0 && n >= 0, "spsp_project: bad arguments");
    if (B == 0 || n == 0) return SPSP_OK;
    dim3 grid((n + 255) / 256, B);
    project_kernel<<<grid, 256, 0, spsp::as_stream(stream)>>>(pts, cam, n, img_wh, out);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

int spsp_wp_to_translation(spsp_ctx* ctx, const float* cam, int B, float focal, float resolution,
                           float* translation, void* stream) {
    SPSP_REQUIRE(ctx && cam && translation && B >= 0, "spsp_wp_to_translation: bad arguments");
    if (B == 0) return SPSP_OK;
    wp2t_kernel<<<(B + 63) / 64, 64, 0, spsp::as_stream(stream)>>>(cam, B, focal, resolution, translation);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

int spsp_pack_record(spsp_ctx* ctx, const float* cam, const float* R, const float* betas, int B, float* record,
                     void* stream) {
    SPSP_REQUIRE(ctx && cam && R && betas && record && B >= 0, "spsp_pack_record: bad arguments");
    if (B == 0) return SPSP_OK;
    pack_record_kernel<<<B, 256, 0, spsp::as_stream(stream)>>>(cam, R, betas, B, record);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

int spsp_unpack_record(spsp_ctx* ctx, const float* record, int B, float* cam, float* R, float* betas,
                       void* stream) {
    SPSP_REQUIRE(ctx && cam && R && betas && record && B >= 0, "spsp_unpack_record: bad arguments");
    if (B == 0) return SPSP_OK;
    unpack_record_kernel<<<B, 256, 0, spsp::as_stream(stream)>>>(record, B, cam, R, betas);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // extern "C"
