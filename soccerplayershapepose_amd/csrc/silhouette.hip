// Silhouette term of the fit drivers: the 512 x 512 silhouette of the posed 6890-vertex / 13 776-face mesh
// as the reference renders it every fit iteration, and the metric its best-iterate rule gates on.
//
//   render   `nmr(vertices, faces, t=translation, mode='silhouettes')` with `nr.Renderer(camera_mode='projection', K, R=I,
//            image_size=512, orig_size=512)` (PR/player_recon.py:1075-1080, 1226-1229).  neural_renderer is an un-vendored,
//            un-pinned dependency (PR/requirements.txt); its forward pass is restated here from the published
//            implementation (parity unpinned, like smplx and OpenCV):
//              projection   p = v + t;  x' = x / (z + 1e-9), y' likewise;  (u, v) = (f x' + cx, f y' + cy);  v = 512 - v;
//                           u_n = 2 (u - 256) / 512, v_n = 2 (v - 256) / 512                       (y up, in [-1, 1])
//              rasterise    anti-aliasing on: a 1024 x 1024 grid, sample (xp, yp) = ((2 xi + 1 - 1024) / 1024, ...), output row
//                           r <-> yi = 1023 - r; a sample is covered by face (p0, p1, p2) unless one of
//                           (yp - y0)(x1 - x0) < (xp - x0)(y1 - y0), (yp - y1)(x2 - x1) < (xp - x1)(y2 - y1),
//                           (yp - y2)(x0 - x2) < (xp - x2)(y0 - y2) holds; fill_back adds every face reversed, i.e. a sample
//                           is covered when all three hold with >= or all three with <=; near / far = 0.1 / 100 on the
//                           perspective-correct depth of the sample; silhouette = 2 x 2 average of the coverage.
//   metric   "silhouette_iou" of TrainingLossesAndMetricsTracker (PR/metrics/train_loss_and_metrics_tracker.py:224-229):
//            -sum_px [ p ln(t + 1e-6) + (1 - p) ln(1 - t + 1e-6) ],  p = rendered, t = detected silhouette.
//   rule     an iterate is the new best iff joints2D_l2es AND this metric are both <= their values at the previous
//            best (determine_save_model_weights_this_epoch, :291-298; use at PR/player_recon.py:1259-1271).
// Not built: neural_renderer's approximate BACKWARD pass (only `single_view_optimization_test`, :1429, puts the
// silhouette into the loss; the production fits track it as a metric only).
//
// Kernels: one workgroup per mesh projects the vertices into LDS and walks the faces (bounding box of samples, the
// inequality test above, atomicOr into a 1024 x 1024 bit mask, 128 KB per mesh -- the union needs no depth order);
// a second pass turns 2 x 2 bits into the silhouette value and accumulates the metric (fixed-order block sums: the
// result is bitwise reproducible).  Atomics / HBM-bound by nature, ~1 M sample tests per mesh.
#include <cmath>

#include "common.hpp"

namespace {
using namespace spsp;

constexpr int kRV = SPSP_NUM_VERTS;

__global__ __launch_bounds__(1024) void raster_kernel(const float* __restrict__ verts, const float* __restrict__ trans,
                                                      const int* __restrict__ faces, int F, float focal, int wh,
                                                      unsigned int* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) float s_raster[];      // 3 x 6890 floats = 82 680 B
    float* const sx = s_raster; float* const sy = s_raster + kRV; float* const sz = s_raster + 2 * kRV;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int is = 2 * wh;
    const float tx = trans[b * 3], ty = trans[b * 3 + 1], tz = trans[b * 3 + 2];
    const float half = 0.5f * (float)wh, fwh = (float)wh;
    for (int v = tid; v < kRV; v += 1024) {
        const float* p = verts + ((size_t)b * kRV + v) * 3;
        const float x = p[0] + tx, y = p[1] + ty, z = p[2] + tz;
        const float xd = x / (z + 1e-9f), yd = y / (z + 1e-9f);
        const float u = fmaf(focal, xd, half);
        float vv = fmaf(focal, yd, half);
        vv = fwh - vv;
        sx[v] = 2.0f * (u - half) / fwh;
        sy[v] = 2.0f * (vv - half) / fwh;
        sz[v] = z;
    }
    __syncthreads();
    unsigned int* const m = mask + (size_t)b * ((size_t)is * is / 32);
    const float fis = (float)is;
    for (int f = tid; f < F; f += 1024) {
        const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        const float x0 = sx[i0], y0 = sy[i0], x1 = sx[i1], y1 = sy[i1], x2 = sx[i2], y2 = sy[i2];
        const float z0 = sz[i0], z1 = sz[i1], z2 = sz[i2];
        const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
        const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
        if (!(xmax >= -1.f && xmin <= 1.f && ymax >= -1.f && ymin <= 1.f)) continue;       // also drops NaNs
        // sample xi has xp = (2 xi + 1 - is) / is: candidates with xp in [xmin, xmax], one sample of slack each side
        int c0 = (int)floorf((xmin * fis + fis - 1.f) * 0.5f) - 1, c1 = (int)ceilf((xmax * fis + fis - 1.f) * 0.5f) + 1;
        int q0 = (int)floorf((ymin * fis + fis - 1.f) * 0.5f) - 1, q1 = (int)ceilf((ymax * fis + fis - 1.f) * 0.5f) + 1;
        c0 = max(c0, 0); q0 = max(q0, 0); c1 = min(c1, is - 1); q1 = min(q1, is - 1);
        const float e0x = x1 - x0, e0y = y1 - y0, e1x = x2 - x1, e1y = y2 - y1, e2x = x0 - x2, e2y = y0 - y2;
        // barycentric weights for the depth test: w0 = area(p, p1, p2) / area, ... (signs cancel in the ratio)
        const float area = e0x * (y2 - y0) - e0y * (x2 - x0);
        for (int yi = q0; yi <= q1; ++yi) {
            const float yp = (2.f * (float)yi + 1.f - fis) / fis;
            for (int xi = c0; xi <= c1; ++xi) {
                const float xp = (2.f * (float)xi + 1.f - fis) / fis;
                const float l0 = (yp - y0) * e0x, r0 = (xp - x0) * e0y;
                const float l1 = (yp - y1) * e1x, r1 = (xp - x1) * e1y;
                const float l2 = (yp - y2) * e2x, r2 = (xp - x2) * e2y;
                const bool front = !(l0 < r0) && !(l1 < r1) && !(l2 < r2);
                const bool back = !(l0 > r0) && !(l1 > r1) && !(l2 > r2);      // the reversed copy added by fill_back
                if (!(front || back)) continue;
                if (area != 0.f) {
                    float w0 = ((xp - x1) * (y2 - y1) - (yp - y1) * (x2 - x1)) / area;
                    float w1 = ((xp - x2) * (y0 - y2) - (yp - y2) * (x0 - x2)) / area;
                    float w2 = 1.f - w0 - w1;
                    w0 = fminf(fmaxf(w0, 0.f), 1.f); w1 = fminf(fmaxf(w1, 0.f), 1.f); w2 = fminf(fmaxf(w2, 0.f), 1.f);
                    const float ws = w0 + w1 + w2;
                    const float zp = ws / (w0 / z0 + w1 / z1 + w2 / z2);
                    if (!(zp > 0.1f && zp < 100.f)) continue;
                }
                const int r = is - 1 - yi;                                     // output row
                atomicOr(&m[((size_t)r * is + xi) >> 5], 1u << (xi & 31));
            }
        }
    }
}

// 2 x 2 coverage bits -> silhouette value in {0, .25, .5, .75, 1}; optional image; metric partial per workgroup
__global__ __launch_bounds__(256) void silhouette_kernel(const unsigned int* __restrict__ mask, const uint8_t* __restrict__ target,
                                                         size_t target_stride, int wh, float* __restrict__ image,
                                                         float* __restrict__ partial) {
    __shared__ float red[256];
    const int b = blockIdx.y, is = 2 * wh;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float term = 0.f;
    if (idx < wh * wh) {
        const int y = idx / wh, x = idx - y * wh;
        const unsigned int* m = mask + (size_t)b * ((size_t)is * is / 32);
        const int bit = (2 * x) & 31;
        const unsigned int w0 = m[((size_t)(2 * y) * is + 2 * x) >> 5] >> bit, w1 = m[((size_t)(2 * y + 1) * is + 2 * x) >> 5] >> bit;
        const float p = 0.25f * (float)((w0 & 1u) + ((w0 >> 1) & 1u) + (w1 & 1u) + ((w1 >> 1) & 1u));
        if (image) image[(size_t)b * wh * wh + idx] = p;
        if (target) {
            const float t = (float)target[(size_t)b * target_stride + idx];
            term = -(p * logf(t + 1e-6f) + (1.f - p) * logf(1.f - t + 1e-6f));
        }
    }
    red[threadIdx.x] = term;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && partial) partial[(size_t)b * gridDim.x + blockIdx.x] = red[0];
}

__global__ void metric_sum_kernel(const float* __restrict__ partial, int nblk, int B, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += partial[(size_t)b * nblk + i];
    out[b] = s;
}

// the reference's best-iterate rule over a recorded trajectory (one thread per player; iterations are sequential)
__global__ void select_kernel(const float* __restrict__ l2e, const float* __restrict__ sil, int iters, int B,
                              const float* __restrict__ param_hist, float* __restrict__ R, float* __restrict__ betas,
                              float* __restrict__ cam, int* __restrict__ best_iter) {
    const int b = blockIdx.x;
    __shared__ int s_best;
    if (threadIdx.x == 0) {
        float bl = INFINITY, bs = INFINITY;
        int best = 0;
        for (int it = 0; it < iters; ++it) {
            const float a = l2e[(size_t)b * iters + it], c = sil ? sil[(size_t)b * iters + it] : 0.f;
            if (!(a > bl) && !(sil && c > bs)) { bl = a; bs = c; best = it; }
        }
        s_best = best;
        if (best_iter) best_iter[b] = best;
    }
    __syncthreads();
    const float* src = param_hist + ((size_t)b * iters + s_best) * SPSP_RECORD_FLOATS;
    for (int p = threadIdx.x; p < SPSP_RECORD_FLOATS; p += blockDim.x) {
        const float v = src[p];
        if (p < 3) cam[b * 3 + p] = v;
        else if (p < 219) R[(size_t)b * 216 + (p - 3)] = v;
        else betas[b * 10 + (p - 219)] = v;
    }
}

}  // namespace

struct spsp_rasterizer {
    spsp::DevicePool pool;
    int* faces;
    int F, max_batch, wh;
    unsigned int* mask;      // [max_batch][(2 wh)^2 / 32]
    float* partial;          // [max_batch][ceil(wh^2 / 256)]
};

extern "C" {

int spsp_rasterizer_load(spsp_ctx* ctx, const int32_t* faces, int F, int max_batch, int img_wh, spsp_rasterizer** out) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && faces && out && F > 0 && max_batch > 0, "spsp_rasterizer_load: bad arguments");
    SPSP_REQUIRE(img_wh >= 16 && img_wh <= 2048 && img_wh % 16 == 0, "spsp_rasterizer_load: img_wh=%d (a multiple of 16 up to 2048)", img_wh);
    for (int i = 0; i < 3 * F; ++i) SPSP_REQUIRE(faces[i] >= 0 && faces[i] < SPSP_NUM_VERTS, "spsp_rasterizer_load: face index %d out of range", faces[i]);
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    spsp_rasterizer* h = new spsp_rasterizer;
    h->F = F; h->max_batch = max_batch; h->wh = img_wh;
    int rc = h->pool.upload(&h->faces, faces, (size_t)3 * F);
    if (rc == SPSP_OK) rc = h->pool.alloc(&h->mask, (size_t)max_batch * (4 * (size_t)img_wh * img_wh / 32));
    if (rc == SPSP_OK) rc = h->pool.alloc(&h->partial, (size_t)max_batch * ((img_wh * img_wh + 255) / 256));
    if (rc != SPSP_OK) { delete h; return rc; }
    *out = h;
    return SPSP_OK;
}

void spsp_rasterizer_free(spsp_rasterizer* h) { delete h; }

int spsp_render_silhouettes(spsp_ctx* ctx, spsp_rasterizer* h, const float* verts, const float* translation, float focal, int B,
                            const uint8_t* target, int target_shared, float* image, float* metric, void* stream) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && h, "spsp_render_silhouettes: NULL handle");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    SPSP_REQUIRE(B >= 0 && B <= h->max_batch, "spsp_render_silhouettes: B=%d exceeds max_batch=%d", B, h->max_batch);
    if (B == 0) return SPSP_OK;
    SPSP_REQUIRE(verts && translation && (image || metric), "spsp_render_silhouettes: NULL argument");
    SPSP_REQUIRE(!metric || target, "spsp_render_silhouettes: the metric needs a target silhouette");
    hipStream_t st = as_stream(stream);
    const int wh = h->wh, nblk = (wh * wh + 255) / 256;
    SPSP_HIP_CHECK(hipMemsetAsync(h->mask, 0, (size_t)B * (4 * (size_t)wh * wh / 32) * sizeof(unsigned int), st));
    ProfScope ps(ctx, kKProxyBuild, st);
    SPSP_TRY(ensure_dynamic_lds<&raster_kernel>(3 * kRV * (int)sizeof(float)));
    raster_kernel<<<B, 1024, 3 * kRV * sizeof(float), st>>>(verts, translation, h->faces, h->F, focal, wh, h->mask);
    silhouette_kernel<<<dim3(nblk, B), 256, 0, st>>>(h->mask, target, target_shared ? 0 : (size_t)wh * wh, wh, image,
                                                     metric ? h->partial : nullptr);
    if (metric) metric_sum_kernel<<<(B + 63) / 64, 64, 0, st>>>(h->partial, nblk, B, metric);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

int spsp_fit_select(spsp_ctx* ctx, const float* l2e_hist, const float* sil_hist, int iters, int B, const float* param_hist, float* R,
                    float* betas, float* cam, int32_t* best_iter, void* stream) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && l2e_hist && param_hist && R && betas && cam && iters > 0 && B >= 0, "spsp_fit_select: bad arguments");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    if (B == 0) return SPSP_OK;
    select_kernel<<<B, 64, 0, as_stream(stream)>>>(l2e_hist, sil_hist, iters, B, param_hist, R, betas, cam, best_iter);
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // extern "C"
