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
// The differentiable half -- neural_renderer's approximate BACKWARD pass and the loss of `single_view_optimization_test`
// (:1429) -- is silhouette_grad.hip.
//
// Kernels: a workgroup = (mesh, strip of 128 sample rows): it projects the vertices into LDS, walks ALL faces (bounding box
// of samples clipped to its strip, the inequality test above) and ORs the covered samples into its strip's bit mask IN LDS
// (16 KB; the union needs no depth order), then writes the strip out with plain stores -- 8 strips x B workgroups, no
// global atomics and no memset.  (First version: one workgroup per mesh and atomicOr straight into the 128 KB global
// mask: 7.2 ms for 32 meshes, bound by ~1 M global atomics per mesh on neighbouring words.)  A second pass turns 2 x 2
// bits into the silhouette value and accumulates the metric (fixed-order block sums: bitwise reproducible).
#include <cmath>

#include "common.hpp"
#include "silhouette.hpp"

namespace {
using namespace spsp;

constexpr int kRV = SPSP_NUM_VERTS;

constexpr int kStrips = 8;                  // sample-row strips per mesh (1024 / 8 = 128 rows = 16 KB of mask bits)

__global__ __launch_bounds__(1024) void raster_kernel(const float* __restrict__ verts, const float* __restrict__ trans,
                                                      const int* __restrict__ faces, int F, float focal, int wh,
                                                      unsigned int* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) float s_raster[];      // 3 x 6890 floats = 82 680 B, then the strip's mask
    float* const sx = s_raster; float* const sy = s_raster + kRV; float* const sz = s_raster + 2 * kRV;
    unsigned int* const lmask = reinterpret_cast<unsigned int*>(s_raster + 3 * kRV);
    const int b = blockIdx.x, strip = blockIdx.y, tid = threadIdx.x;
    const int is = 2 * wh;
    const int rows = is / kStrips, r_lo = strip * rows;                    // output rows [r_lo, r_lo + rows) of this strip
    const int words = rows * is / 32;
    for (int i = tid; i < words; i += 1024) lmask[i] = 0u;
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
    const float fis = (float)is;
    // output row r <-> sample row yi = is - 1 - r: this strip holds yi in [yi_lo, yi_hi]
    const int yi_hi = is - 1 - r_lo, yi_lo = is - r_lo - rows;
    const int lane = tid & 63;
    // the coverage test of one sample against one face (the reference's arithmetic, see the header); sets the bit
    auto test_sample = [&](int xi, int yi, float x0, float y0, float x1, float y1, float x2, float y2, float z0, float z1, float z2) {
#pragma clang fp contract(off)      // the restated renderer rounds every product: no fused multiply-add in the coverage / depth tests
        const float yp = (2.f * (float)yi + 1.f - fis) / fis, xp = (2.f * (float)xi + 1.f - fis) / fis;
        const float e0x = x1 - x0, e0y = y1 - y0, e1x = x2 - x1, e1y = y2 - y1, e2x = x0 - x2, e2y = y0 - y2;
        const float l0 = (yp - y0) * e0x, r0 = (xp - x0) * e0y;
        const float l1 = (yp - y1) * e1x, r1 = (xp - x1) * e1y;
        const float l2 = (yp - y2) * e2x, r2 = (xp - x2) * e2y;
        const bool front = !(l0 < r0) && !(l1 < r1) && !(l2 < r2);
        const bool back = !(l0 > r0) && !(l1 > r1) && !(l2 > r2);              // the reversed copy added by fill_back
        if (!(front || back)) return;
        // barycentric weights for the depth test: w0 = area(p, p1, p2) / area, ... (signs cancel in the ratio)
        const float area = e0x * (y2 - y0) - e0y * (x2 - x0);
        if (area != 0.f) {
            float w0 = ((xp - x1) * (y2 - y1) - (yp - y1) * (x2 - x1)) / area;
            float w1 = ((xp - x2) * (y0 - y2) - (yp - y2) * (x0 - x2)) / area;
            float w2 = 1.f - w0 - w1;
            w0 = fminf(fmaxf(w0, 0.f), 1.f); w1 = fminf(fmaxf(w1, 0.f), 1.f); w2 = fminf(fmaxf(w2, 0.f), 1.f);
            const float ws = w0 + w1 + w2;
            const float zp = ws / (w0 / z0 + w1 / z1 + w2 / z2);
            if (!(zp > 0.1f && zp < 100.f)) return;
        }
        const int r = is - 1 - yi;                                             // output row
        atomicOr(&lmask[((r - r_lo) * is + xi) >> 5], 1u << (xi & 31));
    };
    // A wave takes 64 faces at a time, one per lane: the lane fetches the face and clips its box of candidate samples to
    // the strip.  Small boxes (<= 64 samples) are tested by their own lane; every larger one is then taken up by the WHOLE
    // wave in turn -- its data broadcast with readlane, the box walked in 8 x 8 tiles, one sample per lane.  (A thread per
    // face alone left a wave waiting for its largest box: 2.5-3 M candidate samples per posed mesh, boxes from 40 to
    // 1900 samples -- 5 ms per batch of 32 meshes however many workgroups shared it.)
    constexpr int kSmall = 64;
    const int wave = tid >> 6;
    for (int fbase = wave * 64; fbase < F; fbase += 16 * 64) {
        const int f = fbase + lane;
        float x0 = 0.f, y0 = 0.f, x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f, z0 = 1.f, z1 = 1.f, z2 = 1.f;
        int c0 = 0, c1 = -1, q0 = 0, q1 = -1;
        if (f < F) {
            const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
            x0 = sx[i0]; y0 = sy[i0]; x1 = sx[i1]; y1 = sy[i1]; x2 = sx[i2]; y2 = sy[i2];
            z0 = sz[i0]; z1 = sz[i1]; z2 = sz[i2];
            const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
            const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
            if (xmax >= -1.f && xmin <= 1.f && ymax >= -1.f && ymin <= 1.f) {              // (false for NaNs)
                // sample xi has xp = (2 xi + 1 - is) / is: candidates with xp in [xmin, xmax], one sample of slack each side
                c0 = max((int)floorf((xmin * fis + fis - 1.f) * 0.5f) - 1, 0);
                c1 = min((int)ceilf((xmax * fis + fis - 1.f) * 0.5f) + 1, is - 1);
                q0 = max((int)floorf((ymin * fis + fis - 1.f) * 0.5f) - 1, yi_lo);
                q1 = min((int)ceilf((ymax * fis + fis - 1.f) * 0.5f) + 1, yi_hi);
            }
        }
        const bool any = c0 <= c1 && q0 <= q1;
        const bool small = any && (c1 - c0 + 1) * (q1 - q0 + 1) <= kSmall;
        if (small)
            for (int yi = q0; yi <= q1; ++yi)
                for (int xi = c0; xi <= c1; ++xi) test_sample(xi, yi, x0, y0, x1, y1, x2, y2, z0, z1, z2);
        unsigned long long big = __builtin_amdgcn_ballot_w64(any && !small);
        while (big) {
            const int src = __builtin_ctzll(big);
            big &= big - 1;
            auto bc = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src)); };
            const float bx0 = bc(x0), by0 = bc(y0), bx1 = bc(x1), by1 = bc(y1), bx2 = bc(x2), by2 = bc(y2);
            const float bz0 = bc(z0), bz1 = bc(z1), bz2 = bc(z2);
            const int bc0 = __builtin_amdgcn_readlane(c0, src), bc1 = __builtin_amdgcn_readlane(c1, src);
            const int bq0 = __builtin_amdgcn_readlane(q0, src), bq1 = __builtin_amdgcn_readlane(q1, src);
            for (int ty = bq0; ty <= bq1; ty += 8)
                for (int tx = bc0; tx <= bc1; tx += 8) {
                    const int yi = ty + (lane >> 3), xi = tx + (lane & 7);
                    if (yi <= bq1 && xi <= bc1) test_sample(xi, yi, bx0, by0, bx1, by1, bx2, by2, bz0, bz1, bz2);
                }
        }
    }
    __syncthreads();
    unsigned int* const m = mask + (size_t)b * ((size_t)is * is / 32) + (size_t)r_lo * is / 32;
    for (int i = tid; i < words; i += 1024) m[i] = lmask[i];
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

extern "C" {

int spsp_rasterizer_load(spsp_ctx* ctx, const int32_t* faces, int F, int max_batch, int img_wh, spsp_rasterizer** out) {
    using namespace spsp;
    SPSP_REQUIRE(ctx && faces && out && F > 0 && max_batch > 0, "spsp_rasterizer_load: bad arguments");
    SPSP_REQUIRE(img_wh >= 16 && img_wh <= 2048 && img_wh % 16 == 0, "spsp_rasterizer_load: img_wh=%d (a multiple of 16 up to 2048)", img_wh);
    for (int i = 0; i < 3 * F; ++i) SPSP_REQUIRE(faces[i] >= 0 && faces[i] < SPSP_NUM_VERTS, "spsp_rasterizer_load: face index %d out of range", faces[i]);
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    spsp_rasterizer* h = new spsp_rasterizer;
    h->F = F; h->max_batch = max_batch; h->wh = img_wh;
    h->host_faces.assign(faces, faces + (size_t)3 * F);
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
    ProfScope ps(ctx, kKProxyBuild, st);
    const int raster_lds = 3 * kRV * (int)sizeof(float) + (2 * wh / kStrips) * (2 * wh / 32) * (int)sizeof(unsigned int);   // vertices + the strip's mask
    SPSP_REQUIRE((2 * wh) % (32 * kStrips) == 0 && raster_lds <= 160 * 1024, "spsp_render_silhouettes: image size %d", wh);
    SPSP_TRY(ensure_dynamic_lds<&raster_kernel>(raster_lds));
    raster_kernel<<<dim3(B, kStrips), 1024, raster_lds, st>>>(verts, translation, h->faces, h->F, focal, wh, h->mask);
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
