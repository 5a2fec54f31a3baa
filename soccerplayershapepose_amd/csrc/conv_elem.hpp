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
    // c += A . B with the A operand read STRAIGHT from the accumulation register file.  For register-stationary weights beyond the 256
    // architectural VGPRs: hipcc homes such long-lived fragments in AGPRs and copies them back in front of every MFMA (four
    // v_accvgpr_read per fragment use: 578 of them per 320 MFMAs in conv_s2rs, round 4) although the matrix core takes AGPR operands.
    // Inline asm: nothing inside is padded by hipcc -- callers keep VALU writes of c two states away and put >= 18 wait states between
    // the last MFMA and the first VALU read of c (cdna_hip_programming.md 5.7).
    template <typename V>
    static __device__ __forceinline__ void mfma_agpr(f4& c, const V& a, const V& b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
    }
    template <typename V>
    static __device__ __forceinline__ void mfma_vgpr(f4& c, const V& a, const V& b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    // the FIRST product of an accumulation chain: c = A . B + c0 with c0 in registers of its own (a bias) or the constant 0.  An
    // accumulator initialised by VALU moves is a hazard in front of an asm MFMA: hipcc rematerialises the moves right in front of
    // their first use and pads nothing (round 5, conv_s2rs: sums off in the fourth digit).
    template <typename V>
    static __device__ __forceinline__ void mfma_agpr_from(f4& c, const V& a, const V& b, const f4& c0) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(c) : "a"(a), "v"(b), "v"(c0));
    }
    template <typename V>
    static __device__ __forceinline__ void mfma_agpr_zero(f4& c, const V& a, const V& b) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "a"(a), "v"(b));
    }
    // the 32 x 32 x 16 shape (one MFMA per 32 cycles where the 16 x 16 x 32 shape issues two: half the instructions per FLOP for a
    // loop that is bound by its issue slots -- conv_stem.hip): 16 accumulator registers per lane, A / B fragments of 8 elements
    typedef float f16v __attribute__((ext_vector_type(16)));
    template <typename V>
    static __device__ __forceinline__ void mfma32_agpr(f16v& c, const V& a, const V& b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
    }
    template <typename V>
    static __device__ __forceinline__ void mfma32_agpr_from(f16v& c, const V& a, const V& b, const f16v& c0) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(c) : "a"(a), "v"(b), "v"(c0));
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
    template <typename V>
    static __device__ __forceinline__ void mfma_agpr(f4& c, const V& a, const V& b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
    }
    template <typename V>
    static __device__ __forceinline__ void mfma_vgpr(f4& c, const V& a, const V& b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    // the FIRST product of an accumulation chain: c = A . B + c0 with c0 in registers of its own (a bias) or the constant 0.  An
    // accumulator initialised by VALU moves is a hazard in front of an asm MFMA: hipcc rematerialises the moves right in front of
    // their first use and pads nothing (round 5, conv_s2rs: sums off in the fourth digit).
    template <typename V>
    static __device__ __forceinline__ void mfma_agpr_from(f4& c, const V& a, const V& b, const f4& c0) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(c) : "a"(a), "v"(b), "v"(c0));
    }
    template <typename V>
    static __device__ __forceinline__ void mfma_agpr_zero(f4& c, const V& a, const V& b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(c) : "a"(a), "v"(b));
    }
    // the 32 x 32 x 16 shape (one MFMA per 32 cycles where the 16 x 16 x 32 shape issues two: half the instructions per FLOP for a
    // loop that is bound by its issue slots -- conv_stem.hip): 16 accumulator registers per lane, A / B fragments of 8 elements
    typedef float f16v __attribute__((ext_vector_type(16)));
    template <typename V>
    static __device__ __forceinline__ void mfma32_agpr(f16v& c, const V& a, const V& b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
    }
    template <typename V>
    static __device__ __forceinline__ void mfma32_agpr_from(f16v& c, const V& a, const V& b, const f16v& c0) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(c) : "a"(a), "v"(b), "v"(c0));
    }
    static __device__ __forceinline__ uint32_t pack2(float a, float b) {
        f2 v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, x2));      // round-to-nearest-even
    }
    static __device__ __forceinline__ float lo(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xFFFFu)); }
    static __device__ __forceinline__ float hi(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
    static __device__ __forceinline__ float one(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
};

// Between the last asm MFMA of a chain and the first VALU / LDS / store read of its result: the wait states hipcc would have padded
// (18 cover every XDL op), tied to the accumulators by operands so that no reader is scheduled above them.
typedef float f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_drain8(f4_t& a, f4_t& b, f4_t& c, f4_t& d, f4_t& e, f4_t& f, f4_t& g, f4_t& h) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
}

}  // namespace spsp
