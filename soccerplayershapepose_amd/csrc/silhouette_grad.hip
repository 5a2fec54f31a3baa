// The DIFFERENTIABLE half of the silhouette term: the loss of `single_view_optimization_test`
// (PR/player_recon.py:1429-1557; PR/losses/multi_task_loss.py:132-144) on the rendered silhouette, and neural_renderer's
// approximate backward pass down to the mesh vertices and the camera translation.
//
// neural_renderer is an un-vendored, un-pinned dependency of the reference (PR/requirements.txt): PARITY UNPINNED, restated from
// the published implementation (Kato et al., CVPR 2018; the PyTorch port's rasterize_cuda_kernel.cu) -- oracle/silhouette_grad.py
// is the CPU twin of this file and documents the algorithm.  In short:
//   forward   every face also with reversed vertex order (fill_back); the copy that is not back-facing takes part; per sample of
//             the 2x super-sampled grid the covering copy with the smallest perspective-correct depth in (0.1, 100) OWNS the
//             sample (face_index_map); alpha = 1 on owned samples; vertical flip; 2 x 2 average.
//   loss      S = sum_px [p ln(t + 1e-6) + (1 - p) ln(1 - t + 1e-6)],  L = -S w + lv,  w = exp(-lv)   (p rendered, t detected)
//   backward  dL/dp = -(ln(t + 1e-6) - ln(1 - t + 1e-6)) w, a quarter of it on each of the pixel's four samples; then per front
//             copy, edge and axis the scan of `backward_pixel_map`: pixels whose alpha differs from the in- / out-pixel's in the
//             direction the loss asks for pull the edge's two vertices by -diff / dist (dist: how far the vertex has to move for
//             the edge to reach the pixel, eps-padded); x / y only.  Then the vertex gather, the projection
//             (x_n = 2 f x / (wh z), y_n = -2 f y / (wh z), v = vertex + t) and the sum over vertices for dL/dt.
//
// Kernels
//   sil_project_kernel   vertices + t -> (x_n, y_n, z) per vertex                                   [B x 6890]
//   sil_owner_kernel     workgroup = (mesh, 64 x 64-sample tile): every thread walks faces, the tile's z-buffer lives in LDS as
//                        packed (depth bits << 32 | copy index) under ds_min_u64 -- the smallest depth wins, ties go to the
//                        lower copy index, exactly the sequential rule of the reference loop                [B x 256 tiles]
//   sil_value_kernel     2 x 2 samples -> p, the terms of S (fixed-order block sums), optional image
//   sil_backward_kernel  one wave per (mesh, face): the three edges x two axes scans, d1 over the lanes -> d/d(x_n, y_n) of its three vertices
//   sil_vertex_kernel    per vertex: fixed-order sum over its incident face corners (CSR built at load), projection backward,
//                        block partial sums of dL/dt
// Same fp32 operations in the same order as the CPU twin wherever a comparison decides something (coverage, ownership,
// scan limits): #pragma clang fp contract(off).
#include <cmath>
#include <vector>

#include "common.hpp"
#include "silhouette.hpp"

namespace {
using namespace spsp;

constexpr int kRV = SPSP_NUM_VERTS;
constexpr int kTile = 64;                    // samples per tile edge
constexpr float kNear = 0.1f, kFar = 100.f, kEps = 1e-4f;

__global__ __launch_bounds__(256) void sil_project_kernel(const float* __restrict__ verts, const float* __restrict__ trans, int tstride, float focal,
                                                          int wh, float4* __restrict__ proj) {
#pragma clang fp contract(off)
    const int v = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (v >= kRV) return;
    const float* p = verts + ((size_t)b * kRV + v) * 3;
    const float x = p[0] + trans[b * tstride], y = p[1] + trans[b * tstride + 1], z = p[2] + trans[b * tstride + 2];
    const float half = 0.5f * (float)wh, fwh = (float)wh;
    const float xd = x / (z + 1e-9f), yd = y / (z + 1e-9f);
    const float u = __builtin_fmaf(focal, xd, half);
    float vv = __builtin_fmaf(focal, yd, half);
    vv = fwh - vv;
    proj[(size_t)b * kRV + v] = make_float4(2.0f * (u - half) / fwh, 2.0f * (vv - half) / fwh, z, 0.f);
}

// The front-facing copy of face f: vertex order (i0, i1, i2) or reversed, its index in the doubled list, its coordinates.
struct FrontFace {
    float x[3], y[3], z[3];
    int fn;            // f or f + F
    bool rev;
};
__device__ __forceinline__ FrontFace front_copy(const float4* __restrict__ pj, const int* __restrict__ faces, int f, int F) {
#pragma clang fp contract(off)
    FrontFace o;
    const float4 a = pj[faces[3 * f]], b = pj[faces[3 * f + 1]], c = pj[faces[3 * f + 2]];
    const bool back = (c.y - a.y) * (b.x - a.x) < (b.y - a.y) * (c.x - a.x);
    o.rev = back;
    o.fn = back ? f + F : f;
    const float4 p0 = back ? c : a, p2 = back ? a : c;
    o.x[0] = p0.x; o.y[0] = p0.y; o.z[0] = p0.z;
    o.x[1] = b.x; o.y[1] = b.y; o.z[1] = b.z;
    o.x[2] = p2.x; o.y[2] = p2.y; o.z[2] = p2.z;
    return o;
}

__global__ __launch_bounds__(256) void sil_owner_kernel(const float4* __restrict__ proj, const int* __restrict__ faces, int F, int is,
                                                        int* __restrict__ owner) {
#pragma clang fp contract(off)
    __shared__ unsigned long long zbuf[kTile * kTile];
    const int b = blockIdx.x, tile = blockIdx.y, tid = threadIdx.x;
    const int tpr = is / kTile;
    const int ty0 = (tile / tpr) * kTile, tx0 = (tile % tpr) * kTile;          // sample rows are yi (y up)
    for (int i = tid; i < kTile * kTile; i += 256) zbuf[i] = ~0ull;
    __syncthreads();
    const float4* pj = proj + (size_t)b * kRV;
    const float fis = (float)is;
    for (int f = tid; f < F; f += 256) {
        const FrontFace t = front_copy(pj, faces, f, F);
        const float x0 = t.x[0], x1 = t.x[1], x2 = t.x[2], y0 = t.y[0], y1 = t.y[1], y2 = t.y[2];
        const float xmin = fminf(x0, fminf(x1, x2)), xmax = fmaxf(x0, fmaxf(x1, x2));
        const float ymin = fminf(y0, fminf(y1, y2)), ymax = fmaxf(y0, fmaxf(y1, y2));
        if (!(xmax >= -1.f && xmin <= 1.f && ymax >= -1.f && ymin <= 1.f)) continue;
        int c0 = max((int)floorf((xmin * fis + fis - 1.f) * 0.5f) - 1, 0), c1 = min((int)ceilf((xmax * fis + fis - 1.f) * 0.5f) + 1, is - 1);
        int q0 = max((int)floorf((ymin * fis + fis - 1.f) * 0.5f) - 1, 0), q1 = min((int)ceilf((ymax * fis + fis - 1.f) * 0.5f) + 1, is - 1);
        c0 = max(c0, tx0); c1 = min(c1, tx0 + kTile - 1); q0 = max(q0, ty0); q1 = min(q1, ty0 + kTile - 1);
        if (c1 < c0 || q1 < q0) continue;
        // pixel coordinates and the inverse of the face's [x y 1] matrix (barycentric weights of a sample)
        const float p0x = 0.5f * (x0 * fis + fis - 1.f), p1x = 0.5f * (x1 * fis + fis - 1.f), p2x = 0.5f * (x2 * fis + fis - 1.f);
        const float p0y = 0.5f * (y0 * fis + fis - 1.f), p1y = 0.5f * (y1 * fis + fis - 1.f), p2y = 0.5f * (y2 * fis + fis - 1.f);
        const float den = p2x * (p0y - p1y) + p0x * (p1y - p2y) + p1x * (p2y - p0y);
        if (den == 0.f) continue;
        const float a0 = (p1y - p2y) / den, b0 = (p2x - p1x) / den, g0 = (p1x * p2y - p2x * p1y) / den;
        const float a1 = (p2y - p0y) / den, b1 = (p0x - p2x) / den, g1 = (p2x * p0y - p0x * p2y) / den;
        const float a2 = (p0y - p1y) / den, b2 = (p1x - p0x) / den, g2 = (p0x * p1y - p1x * p0y) / den;
        const float e0x = x1 - x0, e0y = y1 - y0, e1x = x2 - x1, e1y = y2 - y1, e2x = x0 - x2, e2y = y0 - y2;
        for (int yi = q0; yi <= q1; ++yi) {
            const float fy = (float)yi, yp = (2.f * fy + 1.f - fis) / fis;
            for (int xi = c0; xi <= c1; ++xi) {
                const float fx = (float)xi, xp = (2.f * fx + 1.f - fis) / fis;
                if (((yp - y0) * e0x < (xp - x0) * e0y) || ((yp - y1) * e1x < (xp - x1) * e1y) || ((yp - y2) * e2x < (xp - x2) * e2y)) continue;
                float w0 = fminf(fmaxf(a0 * fx + b0 * fy + g0, 0.f), 1.f);
                float w1 = fminf(fmaxf(a1 * fx + b1 * fy + g1, 0.f), 1.f);
                float w2 = fminf(fmaxf(a2 * fx + b2 * fy + g2, 0.f), 1.f);
                const float ws = w0 + w1 + w2;
                w0 = w0 / ws; w1 = w1 / ws; w2 = w2 / ws;
                const float zp = 1.f / (w0 / t.z[0] + w1 / t.z[1] + w2 / t.z[2]);
                if (!(zp > kNear && zp < kFar)) continue;
                const unsigned long long key = ((unsigned long long)__float_as_uint(zp) << 32) | (unsigned int)t.fn;
                atomicMin(&zbuf[(yi - ty0) * kTile + (xi - tx0)], key);
            }
        }
    }
    __syncthreads();
    int* ob = owner + (size_t)b * is * is;
    for (int i = tid; i < kTile * kTile; i += 256) {
        const unsigned long long k = zbuf[i];
        ob[(size_t)(ty0 + i / kTile) * is + tx0 + (i % kTile)] = (k == ~0ull) ? -1 : (int)(unsigned int)(k & 0xFFFFFFFFull);
    }
}

// 2 x 2 owned samples -> p; optional image; the terms of S as fixed-order block partial sums
__global__ __launch_bounds__(256) void sil_value_kernel(const int* __restrict__ owner, const uint8_t* __restrict__ target, size_t target_stride,
                                                        int wh, float* __restrict__ image, float* __restrict__ partial) {
    __shared__ float red[256];
    const int b = blockIdx.y, is = 2 * wh;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float term = 0.f;
    if (idx < wh * wh) {
        const int r = idx / wh, c = idx - r * wh;                  // output row r <-> sample rows yi = is - 1 - 2r, is - 2 - 2r
        const int* ob = owner + (size_t)b * is * is;
        const int ya = is - 1 - 2 * r, yb = ya - 1;
        const int n = (ob[(size_t)ya * is + 2 * c] >= 0) + (ob[(size_t)ya * is + 2 * c + 1] >= 0) + (ob[(size_t)yb * is + 2 * c] >= 0) +
                      (ob[(size_t)yb * is + 2 * c + 1] >= 0);
        const float p = 0.25f * (float)n;
        if (image) image[(size_t)b * wh * wh + idx] = p;
        if (target) {
            const float t = (float)target[(size_t)b * target_stride + idx];
            term = p * logf(t + 1e-6f) + (1.f - p) * logf(1.f - t + 1e-6f);
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

__global__ void sil_sum_kernel(const float* __restrict__ partial, int nblk, int B, float* __restrict__ out, int ostride) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += partial[(size_t)b * nblk + i];
    out[(size_t)b * ostride] = s;
}

// backward_pixel_map: one WAVE per (mesh, face); gface [B][F][3 corners in the face's ORIGINAL vertex order][2].
// The walk over edges, axes and columns d0 is wave-uniform (every lane computes it); the scans along d1 -- up to the image
// border for the out-part: a thread per face spent its time in 1024-step loops, 50 ms per 32 meshes -- are spread over the 64
// lanes, each lane keeps its own partial sums, one butterfly reduction per face at the end.
__global__ __launch_bounds__(256) void sil_backward_kernel(const float4* __restrict__ proj, const int* __restrict__ faces, int F, int is,
                                                           const int* __restrict__ owner, const uint8_t* __restrict__ target,
                                                           size_t target_stride, const float* __restrict__ weight, int wstride, float ga_t0, float ga_t1,
                                                           float* __restrict__ gface) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (f >= F) return;
    const FrontFace t = front_copy(proj + (size_t)b * kRV, faces, f, F);
    const int* ob = owner + (size_t)b * is * is;
    const uint8_t* tg = target + (size_t)b * target_stride;
    const int wh = is / 2;
    const float w = weight[(size_t)b * wstride];
    const float fis = (float)is, two_over_is = 2.f / fis;
    float g[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};            // this lane's share, [corner of the front copy][x, y]
    float px[3], py[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { px[k] = 0.5f * (t.x[k] * fis + fis - 1.f); py[k] = 0.5f * (t.y[k] * fis + fis - 1.f); }
    // owner and dL/dalpha of sample (column d0, row d1) in scan coordinates of `axis`
    auto at = [&](int axis, int d0, int d1, int& own, float& ga) {
        const int xi = axis == 0 ? d0 : d1, yi = axis == 0 ? d1 : d0;
        own = ob[(size_t)yi * is + xi];
        const int r = (is - 1 - yi) >> 1, c = xi >> 1;
        ga = (tg[(size_t)r * wh + c] ? ga_t1 : ga_t0) * w;
    };
    for (int edge = 0; edge < 3; ++edge) {
        const int i0 = edge, i1 = (edge + 1) % 3, i2 = (edge + 2) % 3;
        for (int axis = 0; axis < 2; ++axis) {
            const float a0 = axis == 0 ? px[i0] : py[i0], b0 = axis == 0 ? py[i0] : px[i0];
            const float a1 = axis == 0 ? px[i1] : py[i1], b1 = axis == 0 ? py[i1] : px[i1];
            const float a2 = axis == 0 ? px[i2] : py[i2], b2 = axis == 0 ? py[i2] : px[i2];
            const int direction = axis == 0 ? (a0 < a1 ? -1 : 1) : (a0 < a1 ? 1 : -1);
            const int d0_from = max((int)ceilf(fminf(a0, a1)), 0);
            const int d0_to = (int)fminf(fmaxf(a0, a1), (float)(is - 1));
            float acc0 = 0.f, acc1 = 0.f;
            for (int d0 = d0_from; d0 <= d0_to; ++d0) {
                const float fd0 = (float)d0;
                const float cross = (b1 - b0) / (a1 - a0) * (fd0 - a0) + b0;
                if (!isfinite(cross)) continue;
                const int d1_in = direction > 0 ? (int)floorf(cross) : (int)ceilf(cross);
                const int d1_out = d1_in + direction;
                if (d1_in < 0 || d1_in >= is || d1_out < 0 || d1_out >= is) continue;
                int own_in, own_out;
                float ga_in, ga_out;
                at(axis, d0, d1_in, own_in, ga_in);
                at(axis, d0, d1_out, own_out, ga_out);
                const float alpha_in = own_in >= 0 ? 1.f : 0.f, alpha_out = own_out >= 0 ? 1.f : 0.f;
                const bool m0 = a1 != fd0, m1 = a0 != fd0;
                const float r0 = (a1 - a0) / (a1 - fd0), r1 = (a1 - a0) / (fd0 - a0);
                auto pull = [&](int d1, float diff) {
                    if (!(diff > 0.f)) return;
                    const float off = (float)d1 - cross;
                    if (m0) { float dist = r0 * off * two_over_is; dist = dist > 0.f ? dist + kEps : dist - kEps; acc0 += diff / dist; }
                    if (m1) { float dist = r1 * off * two_over_is; dist = dist > 0.f ? dist + kEps : dist - kEps; acc1 += diff / dist; }
                };
                if (own_in == t.fn) {                             // out-part: from the out-pixel to the image border
                    const int lim = direction > 0 ? is - 1 : 0;
                    const int lo = max(min(d1_out, lim), 0), hi = min(max(d1_out, lim), is - 1);
                    for (int d1 = lo + lane; d1 <= hi; d1 += 64) {
                        int own; float ga;
                        at(axis, d0, d1, own, ga);
                        pull(d1, ((own >= 0 ? 1.f : 0.f) - alpha_in) * ga);
                    }
                }
                {                                                 // in-part: the copy's own pixels up to its far boundary
                    float cross2;
                    if ((fd0 - a0) * (fd0 - a2) < 0.f) cross2 = (b2 - b0) / (a2 - a0) * (fd0 - a0) + b0;
                    else cross2 = (b1 - b2) / (a1 - a2) * (fd0 - a2) + b2;
                    if (isfinite(cross2)) {
                        const int lim = direction > 0 ? (int)ceilf(cross2) : (int)floorf(cross2);
                        const int lo = max(min(d1_in, lim), 0), hi = min(max(d1_in, lim), is - 1);
                        for (int d1 = lo + lane; d1 <= hi; d1 += 64) {
                            int own; float ga;
                            at(axis, d0, d1, own, ga);
                            if (own == t.fn) pull(d1, (1.f - alpha_out) * ga);
                        }
                    }
                }
            }
            g[i0][1 - axis] -= acc0;
            g[i1][1 - axis] -= acc1;
        }
    }
    // six sums over the lanes (fixed butterfly order: reproducible from run to run)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float v = g[k][c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            g[k][c] = v;
        }
    if (lane == 0) {
        float* o = gface + ((size_t)b * F + f) * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int orig = t.rev ? 2 - k : k;                      // corner of the ORIGINAL vertex order
            o[orig * 2] = g[k][0]; o[orig * 2 + 1] = g[k][1];
        }
    }
}

// per vertex: gather over incident corners, projection backward; block partials of dL/dt
__global__ __launch_bounds__(256) void sil_vertex_kernel(const float* __restrict__ verts, const float* __restrict__ trans, int tstride, float focal, int wh,
                                                         const int* __restrict__ adj_ptr, const int* __restrict__ adj_idx,
                                                         const float* __restrict__ gface, int F, float* __restrict__ grad_verts,
                                                         float* __restrict__ tpart) {
    __shared__ float red[3][256];
    const int v = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (v < kRV) {
        float gxn = 0.f, gyn = 0.f;
        const float* gf = gface + (size_t)b * F * 6;
        for (int e = adj_ptr[v]; e < adj_ptr[v + 1]; ++e) { gxn += gf[adj_idx[e] * 2]; gyn += gf[adj_idx[e] * 2 + 1]; }
        const float* p = verts + ((size_t)b * kRV + v) * 3;
        const float x = p[0] + trans[b * tstride], y = p[1] + trans[b * tstride + 1], z = p[2] + trans[b * tstride + 2] + 1e-9f;
        const float k = 2.f * focal / ((float)wh * z);
        gx = gxn * k;
        gy = -gyn * k;
        gz = (-gxn * x + gyn * y) * k / z;
        float* o = grad_verts + ((size_t)b * kRV + v) * 3;
        o[0] = gx; o[1] = gy; o[2] = gz;
    }
    red[0][threadIdx.x] = gx; red[1][threadIdx.x] = gy; red[2][threadIdx.x] = gz;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; red[2][threadIdx.x] += red[2][threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x < 3) tpart[((size_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void sil_tsum_kernel(const float* __restrict__ tpart, int nblk, int B, float* __restrict__ out, int ostride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 3) return;
    const int b = i / 3, c = i % 3;
    float s = 0.f;
    for (int k = 0; k < nblk; ++k) s += tpart[((size_t)b * nblk + k) * 3 + c];
    out[(size_t)b * ostride + c] = s;
}

}  // namespace

namespace spsp {

// workspaces of the differentiable path, allocated at first use (a renderer that only draws never pays for them)
int rasterizer_grad_workspace(spsp_rasterizer* h) {
    if (h->proj) return SPSP_OK;
    const size_t is = 2 * (size_t)h->wh, mb = h->max_batch;
    SPSP_TRY(h->pool.alloc(&h->proj, mb * kRV * 4));
    SPSP_TRY(h->pool.alloc(&h->owner, mb * is * is));
    SPSP_TRY(h->pool.alloc(&h->gface, mb * (size_t)h->F * 6));
    SPSP_TRY(h->pool.alloc(&h->tpart, mb * ((kRV + 255) / 256) * 3));
    SPSP_TRY(h->pool.alloc(&h->vpart, mb * ((size_t)h->wh * h->wh + 255) / 256));
    // vertex -> incident (face, corner) pairs, CSR, corners in ascending face order (the fixed summation order)
    std::vector<int> ptr(kRV + 1, 0), idx((size_t)h->F * 3);
    for (int i = 0; i < 3 * h->F; ++i) ++ptr[h->host_faces[i] + 1];
    for (int v = 0; v < kRV; ++v) ptr[v + 1] += ptr[v];
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int i = 0; i < 3 * h->F; ++i) idx[fill[h->host_faces[i]]++] = i;       // i = face * 3 + corner
    SPSP_TRY(h->pool.upload(&h->adj_ptr, ptr.data(), ptr.size()));
    SPSP_TRY(h->pool.upload(&h->adj_idx, idx.data(), idx.size()));
    return SPSP_OK;
}

}  // namespace spsp

namespace spsp {

// translation [B][tstride], weight [B][wstride], sil_sum [B][sstride], grad_translation [B][gstride]: the silhouette fit keeps
// these per-player scalars in one 16-float block (fit_sil.hip)
int silhouette_loss_grad_impl(spsp_ctx* ctx, spsp_rasterizer* h, const float* verts, const float* translation, int tstride, float focal, int B,
                              const uint8_t* target, int target_shared, const float* weight, int wstride, float* sil_sum, int sstride,
                              float* grad_verts, float* grad_translation, int gstride, float* image, void* stream) {
    SPSP_REQUIRE(ctx && h, "spsp_silhouette_loss_grad: NULL handle");
    SPSP_HIP_CHECK(hipSetDevice(ctx->device));
    SPSP_REQUIRE(B >= 0 && B <= h->max_batch, "spsp_silhouette_loss_grad: B=%d exceeds max_batch=%d", B, h->max_batch);
    if (B == 0) return SPSP_OK;
    SPSP_REQUIRE(verts && translation && target, "spsp_silhouette_loss_grad: NULL argument");
    SPSP_REQUIRE(!grad_verts || weight, "spsp_silhouette_loss_grad: the gradient needs weight [B] = exp(-silhouette_log_var)");
    SPSP_REQUIRE((2 * h->wh) % kTile == 0, "spsp_silhouette_loss_grad: image size %d", h->wh);
    SPSP_TRY(rasterizer_grad_workspace(h));
    hipStream_t st = as_stream(stream);
    const int wh = h->wh, is = 2 * wh, nblk = (wh * wh + 255) / 256, nvb = (kRV + 255) / 256;
    const size_t tgs = target_shared ? 0 : (size_t)wh * wh;
    float4* proj = reinterpret_cast<float4*>(h->proj);
    sil_project_kernel<<<dim3(nvb, B), 256, 0, st>>>(verts, translation, tstride, focal, wh, proj);
    sil_owner_kernel<<<dim3(B, (is / kTile) * (is / kTile)), 256, 0, st>>>(proj, h->faces, h->F, is, h->owner);
    if (sil_sum || image) {
        sil_value_kernel<<<dim3(nblk, B), 256, 0, st>>>(h->owner, target, tgs, wh, image, sil_sum ? h->vpart : nullptr);
        if (sil_sum) sil_sum_kernel<<<(B + 63) / 64, 64, 0, st>>>(h->vpart, nblk, B, sil_sum, sstride);
    }
    if (grad_verts) {
        // dL/dp of a pixel with t = 0 / t = 1, a quarter of it per sample (before the weight): -(ln(t + 1e-6) - ln(1 - t + 1e-6)) / 4
        const float A0 = std::log(0.f + 1e-6f), B0 = std::log(1.f - 0.f + 1e-6f), A1 = std::log(1.f + 1e-6f), B1 = std::log(1.f - 1.f + 1e-6f);
        const float ga_t0 = -(A0 - B0) * 0.25f, ga_t1 = -(A1 - B1) * 0.25f;
        sil_backward_kernel<<<dim3((h->F + 3) / 4, B), 256, 0, st>>>(proj, h->faces, h->F, is, h->owner, target, tgs, weight, wstride, ga_t0, ga_t1, h->gface);
        sil_vertex_kernel<<<dim3(nvb, B), 256, 0, st>>>(verts, translation, tstride, focal, wh, h->adj_ptr, h->adj_idx, h->gface, h->F, grad_verts, h->tpart);
        if (grad_translation) sil_tsum_kernel<<<(B * 3 + 63) / 64, 64, 0, st>>>(h->tpart, nvb, B, grad_translation, gstride);
    }
    SPSP_HIP_CHECK(hipGetLastError());
    return SPSP_OK;
}

}  // namespace spsp

extern "C" {

int spsp_silhouette_loss_grad(spsp_ctx* ctx, spsp_rasterizer* h, const float* verts, const float* translation, float focal, int B,
                              const uint8_t* target, int target_shared, const float* weight, float* sil_sum, float* grad_verts,
                              float* grad_translation, float* image, void* stream) {
    return spsp::silhouette_loss_grad_impl(ctx, h, verts, translation, 3, focal, B, target, target_shared, weight, 1, sil_sum, 1, grad_verts,
                                           grad_translation, 3, image, stream);
}

}  // extern "C"
