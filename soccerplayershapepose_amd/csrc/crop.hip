// Front half of the proxy builder: detected boxes of ONE frame -> the 512 x 512 player crops the
// silhouette / key-point networks take.  Replaces, for all boxes of a frame in one launch,
//   crop_image(image, box, border)                     S/global_utils.py:257-290   (box +- border, clamped to the frame;
//                                                      padded to a square with BGR (53, 88, 83): cv2.copyMakeBorder)
//   cv2.resize(crop, player_crop_size, INTER_LINEAR)   S/PlayerDetection/player_detection.py:181-184, size (512, 512)
// and never materialises the intermediate crop or the padded square: an output pixel computes its four source taps in
// the virtual padded square and reads the frame (or the fill colour) directly.  HBM-bound: 786 KB written per crop.
//
// The resize is OpenCV's 8-bit INTER_LINEAR, restated from its published source (cv::resize, modules/imgproc/src/
// resize.cpp; OpenCV is an un-vendored, un-pinned dependency of the reference -- `opencv-python` in
// PR/requirements.txt -- so this piece is "parity unpinned", like smplx):
//   scale = 1 / (dsize / ssize) in double;  f = (float)((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s
//   s < 0 -> (s, f) = (0, 0);  s >= ssize - 1 -> (s, f) = (ssize - 1, 0)
//   coefficients as 11-bit fixed point: a1 = cvRound(f * 2048), a0 = cvRound((1 - f) * 2048)   (round half to even)
//   rows first:  H = S[s] a0 + S[s + 1] a1        (exact integers, <= 255 * 2048)
//   columns:     dst = ((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2     (the 16-bit SIMD form)
//   an exact 2 x 2 shrink (ssize == 2 * dsize) is routed to INTER_AREA: (p00 + p01 + p10 + p11 + 2) >> 2.
#include <cmath>

#include "common.hpp"

namespace {
using namespace spsp;

struct CropGeom { int left, top, w, h, pad_left, pad_top, size; };

__device__ __forceinline__ CropGeom crop_geom(const int* box, int border, int H, int W) {
    CropGeom g;
    int left = box[0] - border, top = box[1] - border, right = box[2] + border, bottom = box[3] + border;
    left = left < 0 ? 0 : left;
    top = top < 0 ? 0 : top;
    right = right > W ? W : right;
    bottom = bottom > H ? H : bottom;
    g.left = left; g.top = top;
    g.w = right - left > 0 ? right - left : 0;            // numpy slicing: an inverted range is empty
    g.h = bottom - top > 0 ? bottom - top : 0;
    g.size = g.h > g.w ? g.h : g.w;
    g.pad_left = g.pad_top = 0;
    if (g.size == g.h) { const int d = g.size - g.w; g.pad_left = d / 2 + (d & 1); }
    else { const int d = g.size - g.h; g.pad_top = d / 2 + (d & 1); }
    return g;
}

__device__ __forceinline__ void tap(int d, int ssize, double scale, int& s, int& a0, int& a1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
    a0 = (int)rintf((1.f - f) * 2048.f);                   // cvRound: round half to even
    a1 = (int)rintf(f * 2048.f);
}

// pixel (y, x) of the virtual padded square of crop g
__device__ __forceinline__ uchar3 padded_pixel(const uint8_t* __restrict__ frame, int W, const CropGeom& g, int y, int x) {
    const int cy = y - g.pad_top, cx = x - g.pad_left;
    if ((unsigned)cy >= (unsigned)g.h || (unsigned)cx >= (unsigned)g.w) return make_uchar3(53, 88, 83);
    const uint8_t* p = frame + ((size_t)(g.top + cy) * W + (g.left + cx)) * 3;
    return make_uchar3(p[0], p[1], p[2]);
}

__global__ __launch_bounds__(256) void crop_resize_kernel(const uint8_t* __restrict__ frame, int H, int W, const int* __restrict__ boxes,
                                                          int border, int out_wh, uint8_t* __restrict__ out) {
    const int n = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= out_wh * out_wh) return;
    const int dy = idx / out_wh, dx = idx - dy * out_wh;
    const CropGeom g = crop_geom(boxes + 4 * n, border, H, W);
    uint8_t* o = out + ((size_t)n * out_wh * out_wh + idx) * 3;
    if (g.size <= 0) { o[0] = 53; o[1] = 88; o[2] = 83; return; }          // empty crop (cv2 would raise): the fill colour
    if (g.size == 2 * out_wh) {                                            // exact 2 x 2 shrink -> INTER_AREA
        const uchar3 p00 = padded_pixel(frame, W, g, 2 * dy, 2 * dx), p01 = padded_pixel(frame, W, g, 2 * dy, 2 * dx + 1);
        const uchar3 p10 = padded_pixel(frame, W, g, 2 * dy + 1, 2 * dx), p11 = padded_pixel(frame, W, g, 2 * dy + 1, 2 * dx + 1);
        o[0] = (uint8_t)((p00.x + p01.x + p10.x + p11.x + 2) >> 2);
        o[1] = (uint8_t)((p00.y + p01.y + p10.y + p11.y + 2) >> 2);
        o[2] = (uint8_t)((p00.z + p01.z + p10.z + p11.z + 2) >> 2);
        return;
    }
    const double scale = 1.0 / ((double)out_wh / (double)g.size);
    int sx, ax0, ax1, sy, by0, by1;
    tap(dx, g.size, scale, sx, ax0, ax1);
    tap(dy, g.size, scale, sy, by0, by1);
    const int sx1 = sx + 1 < g.size ? sx + 1 : sx, sy1 = sy + 1 < g.size ? sy + 1 : sy;   // weight 0 there when clamped
    const uchar3 p00 = padded_pixel(frame, W, g, sy, sx), p01 = padded_pixel(frame, W, g, sy, sx1);
    const uchar3 p10 = padded_pixel(frame, W, g, sy1, sx), p11 = padded_pixel(frame, W, g, sy1, sx1);
    auto mix = [&](int v00, int v01, int v10, int v11) -> uint8_t {
        const int h0 = v00 * ax0 + v01 * ax1, h1 = v10 * ax0 + v11 * ax1;
        return (uint8_t)((((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2);
    };
    o[0] = mix(p00.x, p01.x, p10.x, p11.x);
    o[1] = mix(p00.y, p01.y, p10.y, p11.y);
    o[2] = mix(p00.z, p01.z, p10.z, p11.z);
}

}  // namespace

extern "C" {

int spsp_crop_players(spsp_ctx* ctx, const uint8_t* frame, int H, int W, const int32_t* boxes, int N, int border, int out_wh,
                      uint8_t* crops, void* stream) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && N >= 0, "spsp_crop_players: bad arguments");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    if (N == 0) return SPSP_OK;
    SPSP_REQUIRE(frame && boxes && crops, "spsp_crop_players: NULL argument");
    SPSP_REQUIRE(H > 0 && W > 0 && H <= 16384 && W <= 16384, "spsp_crop_players: frame %d x %d", H, W);
    SPSP_REQUIRE(out_wh > 0 && out_wh <= 4096 && border >= 0, "spsp_crop_players: out_wh=%d border=%d", out_wh, border);
    dim3 grid((out_wh * out_wh + 255) / 256, N);
    ProfScope ps(ctx, kKProxyBuild, as_stream(stream));
    crop_resize_kernel<<<grid, 256, 0, as_stream(stream)>>>(frame, H, W, boxes, border, out_wh, crops);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // extern "C"
