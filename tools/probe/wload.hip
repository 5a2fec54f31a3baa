// Probe (not part of the product): how fast can every workgroup of a launch pull the SAME weight fragments into its registers?
// The register-stationary convolution kernels (conv_fb / conv_rs / conv_s2q / the stem) measure ~10 B per cycle and CU in their
// prologues.  Each wave issues NF 16-byte-per-lane loads (1 KiB per instruction, coalesced) back to back and waits once.
//   mode 0: every workgroup reads the same bytes            mode 1: a private copy per workgroup
//   mode 2: the same bytes, the order rotated by workgroup  mode 3: the same bytes with sc1 (bypass the per-CU cache)
//   mode 4: the same bytes, nt (non-temporal / streaming)   mode 5: the same bytes through LDS-DMA (144 KiB) + nothing else
// build: hipcc -O3 --offload-arch=gfx950 wload.hip -o wload     usage: ./wload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int NF = 56;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const u32x4* __restrict__ w, size_t wg_stride, unsigned long long* __restrict__ out, unsigned int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* base = w + (MODE == 1 ? (size_t)blockIdx.x * wg_stride : 0) + (size_t)wave * NF * 64 + lane;
    const int rot = MODE == 2 ? (int)((blockIdx.x * 13u) % NF) : 0;
    u32x4 v[NF];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        int g = f + rot; g -= g >= NF ? NF : 0;
        const u32x4* p = base + (size_t)g * 64;
        if constexpr (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[f]) : "v"(p) : "memory");
        else if constexpr (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[f]) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[f]) : "v"(p) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int f = 0; f < NF; ++f) { asm volatile("" : "+v"(v[f])); acc ^= v[f]; }
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

__global__ __launch_bounds__(256, 1) void probe_dma(const unsigned char* __restrict__ w, unsigned long long* __restrict__ out, unsigned int* sink) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int p = wave; p < 144; p += 4)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(w + (size_t)p * 1024 + lane * 16),
                                         (void __attribute__((address_space(3)))*)(smem + p * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (reinterpret_cast<unsigned int*>(smem)[threadIdx.x] == 0x12345u) sink[0] = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const size_t per_wg = (size_t)4 * NF * 64;                  // u32x4 elements per workgroup
    const int max_wgs = 256;
    u32x4* w; unsigned long long* out; unsigned int* sink;
    CK(hipMalloc(&w, per_wg * max_wgs * sizeof(u32x4)));
    CK(hipMemset(w, 1, per_wg * max_wgs * sizeof(u32x4)));
    CK(hipMalloc(&out, max_wgs * sizeof(unsigned long long)));
    CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute((const void*)probe_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    const char* names[6] = {"same bytes", "private copy per workgroup", "same bytes, rotated order", "same bytes, sc1", "same bytes, nt", "same bytes, LDS-DMA 144 KiB"};
    unsigned char* big; const size_t big_bytes = (size_t)1 << 30;
    CK(hipMalloc(&big, big_bytes));
    for (int cold : {0, 1})
    for (int wgs : {64, 256}) {
        for (int mode = 0; mode < 6; ++mode) {
            if (cold && (mode == 2 || mode == 4)) continue;
            std::vector<double> med;
            for (int rep = 0; rep < 5; ++rep) {
                if (cold) CK(hipMemsetAsync(big, rep, big_bytes));      // 1 GiB through L2 and the Infinity Cache: the weights come from HBM again
                switch (mode) {
                    case 0: probe<0><<<wgs, 256>>>(w, per_wg, out, sink); break;
                    case 1: probe<1><<<wgs, 256>>>(w, per_wg, out, sink); break;
                    case 2: probe<2><<<wgs, 256>>>(w, per_wg, out, sink); break;
                    case 3: probe<3><<<wgs, 256>>>(w, per_wg, out, sink); break;
                    case 4: probe<4><<<wgs, 256>>>(w, per_wg, out, sink); break;
                    case 5: probe_dma<<<wgs, 256, 144 * 1024>>>(reinterpret_cast<const unsigned char*>(w), out, sink); break;
                }
                CK(hipDeviceSynchronize());
                std::vector<unsigned long long> h(wgs);
                CK(hipMemcpy(h.data(), out, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                med.push_back((double)h[wgs / 2]);
            }
            std::sort(med.begin(), med.end());
            const double bytes = mode == 5 ? 144.0 * 1024 : (double)per_wg * 16;
            printf("%s %3d workgroups, %-28s: median workgroup %7.0f cycles (100 MHz x ~21) for %6.0f KiB = %5.1f B / cycle / CU\n", cold ? "cold" : "warm", wgs, names[mode], med[2], bytes / 1024,
                   bytes / med[2]);
        }
    }
    return 0;
}
