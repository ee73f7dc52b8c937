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

// Two fp16 pieces of a value that was scaled into fp16's range (|xs| < 2^16): xs = a + b up to 2^-22 |xs| while b is a normal
// fp16 number (|xs| >= 2^-3), less below (b's subnormal step is 2^-24).  With a.a', a.b', b.a' the product carries ~21 bits.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
__device__ __forceinline__ void cv_split2(float xs, _Float16& a, _Float16& b) {
  a = (_Float16)xs;
  b = (_Float16)(xs - (float)a);
}
// the exponent e that puts m = max |x| of a block of values into [2^14, 2^15): x * 2^e is converted; 127 = "no value yet"
__device__ __forceinline__ int cv_block_exponent(float m) {
  if (!(m > 0.f)) return 127;
  const int e = 14 - (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xFF) + 127;     // 14 - floor(log2 m) for normal m
  return e > 110 ? 110 : (e < -110 ? -110 : e);
}

// the two 4-element halves of a transposing LDS read (ds_read_b64_tr_b16) as one MFMA operand
__device__ __forceinline__ bf16x8 wg_join(i16x4 lo, i16x4 hi) {
  typedef __attribute__((ext_vector_type(8))) short i16x8;
  i16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// the six piece products (i + j <= 4) of a 16x16x32 tile, smallest first; w = row operand, x = column operand
// the three piece products of the fp16 form, smallest first
#define F2_MFMA3(ACC, W, X)                                                     \
  {                                                                             \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16((W)[1], (X)[0], ACC, 0, 0, 0);  \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16((W)[0], (X)[1], ACC, 0, 0, 0);  \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16((W)[0], (X)[0], ACC, 0, 0, 0);  \
  }

#define BF3_MFMA6(ACC, W, X)                                                    \
  {                                                                             \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[2], (X)[0], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (X)[2], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[1], (X)[1], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[1], (X)[0], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (X)[1], ACC, 0, 0, 0); \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16((W)[0], (X)[0], ACC, 0, 0, 0); \
  }
