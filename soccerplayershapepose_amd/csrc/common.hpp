// Shared host-side plumbing for libspsp.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/spsp.h"

namespace spsp {

void set_error(const char* fmt, ...);

#define SPSP_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            spsp::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return SPSP_EHIP;                                                             \
        }                                                                                 \
    } while (0)

#define SPSP_REQUIRE(cond, ...)                                                           \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            spsp::set_error(__VA_ARGS__);                                                 \
            return SPSP_EINVAL;                                                           \
        }                                                                                 \
    } while (0)

// Device allocations owned by a handle; freed together.
struct DevicePool {
    std::vector<void*> ptrs;
    ~DevicePool() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T>
    int alloc(T** out, size_t count, bool zero = false) {
        void* p = nullptr;
        SPSP_HIP_CHECK(hipMalloc(&p, count * sizeof(T) + 256));
        if (zero) SPSP_HIP_CHECK(hipMemset(p, 0, count * sizeof(T) + 256));
        ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return SPSP_OK;
    }
    template <typename T>
    int upload(T** out, const T* host, size_t count) {
        int rc = alloc(out, count);
        if (rc != SPSP_OK) return rc;
        SPSP_HIP_CHECK(hipMemcpy(*out, host, count * sizeof(T), hipMemcpyHostToDevice));
        return SPSP_OK;
    }
};

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

// Kernels that need more than 64 KB of dynamic LDS must be told so once PER DEVICE (the attribute belongs to the
// device's copy of the function): one flag per (kernel, device) -- a process may hold contexts on several GPUs.
template <auto Kernel>
inline int ensure_dynamic_lds(int bytes) {
    static bool done[64] = {};
    int dev = 0;
    SPSP_HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && done[dev]) return SPSP_OK;
    SPSP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev >= 0 && dev < 64) done[dev] = true;
    return SPSP_OK;
}
#define SPSP_TRY(expr) do { const int rc_ = (expr); if (rc_ != SPSP_OK) return rc_; } while (0)

// float -> bf16 bits, round-to-nearest-even (host side; finite inputs).
inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)((u >> 16) | ((u & 0xFFFF) ? 0x40 : 0));
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
// float -> fp16 bits, round-to-nearest-even (host side)
inline uint16_t f32_to_f16_host(float f) {
    const _Float16 h = (_Float16)f;
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}
inline float bf16_to_f32_host(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

}  // namespace spsp

// ---- optional per-kernel timing (HIP events on the launch stream), for bench.py's roofline leg ------------
namespace spsp {
enum KernelId {
    kKStem = 0,        // conv1_pool_kernel (conv1 + BN + ReLU + max-pool; weights in registers, rolling input window)
    kKConvWres,        // conv3x3_wres_kernel (layer1: 3x3 / 1, 64 -> 64, weights resident in LDS, persistent)
    kKPatchW16,        // conv3x3_patch_kernel<false, ...> (3x3 / 1 of layer2, layer3: 16 x 16-pixel tiles)
    kKPatchW8,         // conv3x3_patch_kernel<true, ...>  (3x3 / 1 of layer4: four 8 x 8 images per tile)
    kKConvRing,        // conv_ring_kernel<...> (stride-2 3x3 and 1x1 convs; every conv of a ragged batch)
    kKIefFc, kKIefMisc, kKProxyPack, kKProxyBuild, kKRot6d, kKFit, kKSmplChain, kKSmplSkin, kKSmplJoints, kKRelate,
    kKConvS2,          // conv3x3s2_kernel (the three 3x3 / 2 convs + their fused 1x1 / 2 downsamples, parity-plane patch)
    kKConvP4,          // conv3x3_p4_kernel (3x3 / 1 of layer2 and, plan "throughput", layer3: 256-pixel tiles, no K split, producer waves)
    kKConvS2q,         // conv3x3s2q_kernel (layer2.0's stride-2 pair, all output channels per workgroup, weights in registers)
    kKConvRs,          // conv3x3rs_kernel (plan "throughput": layer2's 3x3 / 1 convs, a channel block's weights in registers, persistent)
    kKConvFb,          // conv_basic_block_kernel (layer1: a whole BasicBlock per launch, weights in registers, intermediate tile in LDS)
    kKNumKernels
};
struct ProfRec { int id; hipEvent_t start, stop; hipStream_t stream; unsigned workgroups; };
}  // namespace spsp

struct spsp_ctx {
    int device;
    int num_cus;
    bool prof_on = false;
    int prof_only = -1;                 // >= 0: record only this kernel id (spsp_profile_select)
    std::vector<spsp::ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_free;
};

namespace spsp {
// Profiling events of the scope the current thread is inside (nullptr: none / profiling off).
struct ProfLaunch { hipEvent_t start, stop; bool single, used; unsigned workgroups; };
inline thread_local ProfLaunch* tl_prof = nullptr;

// Times the launches issued during its lifetime when profiling is enabled.
//   single = true : the scope holds exactly ONE launch, issued through spsp::launch_kernel() below, which hands the two events to
//                   hipExtLaunchKernelGGL -- they then carry the kernel's own begin / end timestamps (what rocprofv3's
//                   kernel trace reports), not the stream's idle time and event packets around it (~2-4 us per launch,
//                   a quarter of a 10 us convolution);
//   single = false: legacy bracket, an event recorded before and after whatever the scope launches.
struct ProfScope {
    ProfLaunch pl{nullptr, nullptr, false, false, 0u};
    hipStream_t st;
    spsp_ctx* ctx_ = nullptr;
    size_t rec_ = 0;
    bool active = false;
    ProfScope(spsp_ctx* c, int id, hipStream_t s, bool single = false) : st(s) {
        if (!c || !c->prof_on || (c->prof_only >= 0 && c->prof_only != id)) return;
        hipEvent_t ev[2];
        for (auto& e : ev) {
            if (!c->prof_free.empty()) { e = c->prof_free.back(); c->prof_free.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) return;
        }
        pl = {ev[0], ev[1], single, false, 0u};
        active = true;
        if (single) tl_prof = &pl;
        else (void)hipEventRecord(ev[0], st);
        ctx_ = c; rec_ = c->prof_recs.size();
        c->prof_recs.push_back({id, ev[0], ev[1], st, 0u});
    }
    ~ProfScope() {
        if (!active) return;
        if (rec_ < ctx_->prof_recs.size()) ctx_->prof_recs[rec_].workgroups = pl.workgroups;
        if (pl.single) {
            tl_prof = nullptr;
            if (!pl.used) { (void)hipEventRecord(pl.start, st); (void)hipEventRecord(pl.stop, st); }   // nothing was launched
        } else {
            (void)hipEventRecord(pl.stop, st);
        }
    }
};

// kernel<<<grid, block, lds, st>>>(args...) -- through the profiling events of the enclosing single-launch ProfScope, if any
template <class... KArgs, class... Args>
inline void launch_kernel(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args&&... args) {
    ProfLaunch* const p = tl_prof;
    if (p && p->single && !p->used) {
        p->used = true;
        p->workgroups = grid.x * grid.y * grid.z;
        hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, p->start, p->stop, 0, static_cast<KArgs>(args)...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, static_cast<KArgs>(args)...);
    }
}
}  // namespace spsp
