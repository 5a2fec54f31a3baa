// Per-player geometry statements shared by the stand-alone kernels (geometry.hip) and the fit kernel, which runs them at its
// two ends when it is called from spsp_pipeline (fit.hip): ONE body per formula, so both paths give the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace spsp {

// rot6d_to_rotmat, PR/utils/rigid_transform_utils.py:27-41.  x.view(-1,3,2): a1 = x[0,2,4], a2 = x[1,3,5]; F.normalize divides
// by max(||v||, 1e-12); columns of R are (b1, b2, b3).  p: 6 floats, o: 9 floats (row-major 3x3).
__device__ __forceinline__ void rot6d_one(const float* __restrict__ p, float* __restrict__ o) {
    float a1x = p[0], a1y = p[2], a1z = p[4];
    float a2x = p[1], a2y = p[3], a2z = p[5];
    float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    float d = b1x * a2x + b1y * a2y + b1z * a2z;
    float ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
    float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
    float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    float b3x = b1y * b2z - b1z * b2y;
    float b3y = b1z * b2x - b1x * b2z;
    float b3z = b1x * b2y - b1y * b2x;
    o[0] = b1x; o[1] = b2x; o[2] = b3x;
    o[3] = b1y; o[4] = b2y; o[5] = b3y;
    o[6] = b1z; o[7] = b2z; o[8] = b3z;
}

// convert_weak_perspective_to_camera_translation_torch, PR/utils/cam_utils.py:28-34: component i of [t_x, t_y, 2 f / (res s + 1e-9)]
__device__ __forceinline__ float wp_translation(float s, float tx, float ty, float focal, float res, int i) {
    return i == 0 ? tx : (i == 1 ? ty : 2.f * focal / (res * s + 1e-9f));
}

}  // namespace spsp
