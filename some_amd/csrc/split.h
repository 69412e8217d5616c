// "SPLIT32" operand format of the 3-term split-f16 GEMM path.
//
// A logical fp32 matrix [R][K] (K % 32 == 0) is stored with the SAME byte footprint as fp32: each 32-element
// k-block of a row becomes 128 bytes = 32 f16 `hi` halves followed by 32 f16 `lo` halves, with
//     hi = rn_f16(x),   lo = rn_f16(x - hi)          (x == hi + lo up to 2^-22 |x|).
// A product a*b is then evaluated on the f16 matrix pipe as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with fp32
// accumulation (f16 x f16 products are exact in fp32); the dropped a_lo*b_lo term and the lo roundings are
// O(2^-21) relative - measured end to end: logits within 3e-6 of the fp32 reference, the same as the exact-f32
// MFMA path.  Range: |x| must stay below 65504 (f16 max); every GEMM input on this path
// is a LayerNorm / SiLU / softmax-average / conv output or a weight, all orders of magnitude below that.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__host__ __device__ inline void split_f16(float x, half_t& hi, half_t& lo) {
    hi = (half_t)x;
    lo = (half_t)(x - (float)hi);
}

// bf16 operand mode of the mixed-precision training kernels (configs/midi_conformer.yaml:35 pl_trainer_precision 'bf16'): the hi
// slot of a SPLIT32 block then holds the bf16 rounding of x (same 16 bits of storage, typed half_t only for transport), the
// lo slot is zero, and the one-product kernels issue v_mfma_f32_32x32x16_bf16 on the same fragments.
#if defined(__HIPCC__)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ half_t bf16_as_half(float x) { return __builtin_bit_cast(half_t, (__bf16)x); }
template <bool BF16>
__device__ __forceinline__ f32x16 mfma_hi(half8 a, half8 b, f32x16 c) {
    if constexpr (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
#endif

// byte offset of element k's hi half inside a SPLIT32 row; the lo half sits 64 bytes further
__host__ __device__ inline size_t split_hi_off(int k) { return (size_t)(k >> 5) * 128 + (size_t)(k & 31) * 2; }
