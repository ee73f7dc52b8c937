// fp32 products on the bf16 matrix pipe: the three-way split of an fp32 value into bf16 pieces and the operand types of
// v_mfma_f32_16x16x32_bf16 (shared by glx_conv2d.hip and glx_deconv2d.hip; the arithmetic is described in glx_conv2d.hip).
#pragma once
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short i16x4;

// x = a + b + c up to 2^-24 |x|: a = bf16(x), b = bf16(x - a), c = bf16(x - a - b); the subtractions are exact
__device__ __forceinline__ void cv_split(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  float r = x - (float)a;
  b = (__bf16)r;
  r = r - (float)b;
  c = (__bf16)r;
}

// the two 4-element halves of a transposing LDS read (ds_read_b64_tr_b16) as one MFMA operand
__device__ __forceinline__ bf16x8 wg_join(i16x4 lo, i16x4 hi) {
  typedef __attribute__((ext_vector_type(8))) short i16x8;
  i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// the six piece products (i + j <= 4) of a 16x16x32 tile, smallest first; w = row operand, x = column operand
#define BF3_MFMA6(ACC, W, X)                                                    \
  {                                                                             \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[2], (X)[0], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (X)[2], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[1], (X)[1], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[1], (X)[0], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (X)[1], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (X)[0], ACC, 0, 0, 0); \
  }
