// Storage / MFMA element type of the backbone: bf16 (precision 0, the default) or fp16 (precision 1).
// Both are 2-byte elements with fp32 accumulation on v_mfma_f32_16x16x32_{bf16,f16} at the same rate, so every
// layout, LDS image and DMA in the conv kernels is identical; only the MFMA opcode and the fp32 <-> element
// conversions of the epilogues differ.  fp16 carries 3 more mantissa bits (activations here are O(1..10), far
// inside its range): ~10x smaller deviation from the fp32 reference, which is what brings the END-TO-END
// vertices within 1e-3 m of the all-fp32 oracle (tests/test_gpu_parity.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spsp {

template <bool F16>
struct Elem;

template <>
struct Elem<false> {
    typedef __bf16 x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 x2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    template <typename V>
    static __device__ __forceinline__ f4 mfma(V a, V b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x8, a), __builtin_bit_cast(x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        f2 v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, x2));
    }
    static __device__ __forceinline__ float lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
    static __device__ __forceinline__ float hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
    static __device__ __forceinline__ float one(uint16_t u) { return __builtin_bit_cast(float, (uint32_t)u << 16); }
};

template <>
struct Elem<true> {
    typedef _Float16 x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 x2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    template <typename V>
    static __device__ __forceinline__ f4 mfma(V a, V b, f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x8, a), __builtin_bit_cast(x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        f2 v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, x2));      // round-to-nearest-even
    }
    static __device__ __forceinline__ float lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xFFFFu)); }
    static __device__ __forceinline__ float hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
    static __device__ __forceinline__ float one(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
};

}  // namespace spsp
