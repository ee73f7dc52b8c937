// What the matrix pipe sustains on this part for v_mfma_f32_16x16x32_f16 issued back to back (no memory, no LDS):
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
// waves per SIMD 1, 2, 4; independent accumulators per wave 4, 8.  Prints TFLOP/s and the fraction of 2.5 PFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f4 acc[NACC];
  for (int n = 0; n < NACC; ++n) acc[n] = f4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  if (s == 12345.678f) out[0] = s;
}
// the layer-3 loop's register pattern: 32 weight fragments stay (two planes x 4 tiles x 4 k-steps), a step takes two point fragments
// (here: rotated registers, LDS = 0, or ds_read_b128 one step ahead, LDS = 1) and issues 12 MFMAs on 4 accumulators
// RANDOM = 1: every operand bit pattern is pseudo-random (sign, mantissa, exponent 8 .. 23): the switching activity of real data
__device__ inline unsigned rnd(unsigned& st) { st = st * 1664525u + 1013904223u; return st; }
__device__ inline unsigned rnd_h2(unsigned& st) {         // two random finite fp16 values
  const unsigned r = rnd(st) >> 3, e = rnd(st) >> 7;
  return (r & 0x83FF83FFu) | ((8u + (e & 15u)) << 10) | ((8u + ((e >> 8) & 15u)) << 26);
}
template <int LDS, int RANDOM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_step(float* out, int iters) {
  __shared__ uint4 sy[32 * 64];
  unsigned st = threadIdx.x * 977u + blockIdx.x * 131u + 7u;
  for (int i = threadIdx.x; i < 32 * 64; i += blockDim.x)
    sy[i] = RANDOM ? uint4{rnd_h2(st), rnd_h2(st), rnd_h2(st), rnd_h2(st)} : uint4{(unsigned)i, 0x3c003c00u, 0x38003800u, (unsigned)i * 7u};
  __syncthreads();
  const int lane = threadIdx.x & 63;
  h8 Wa[4][4], Wb[4][4];
  for (int a = 0; a < 4; ++a)
    for (int s = 0; s < 4; ++s) {
      for (int i = 0; i < 8; ++i) { Wa[a][s][i] = (_Float16)(0.01f * (a + s + i + lane)); Wb[a][s][i] = (_Float16)(0.001f * (a * s + i)); }
      if (RANDOM) {
        Wa[a][s] = __builtin_bit_cast(h8, uint4{rnd_h2(st), rnd_h2(st), rnd_h2(st), rnd_h2(st)});
        Wb[a][s] = __builtin_bit_cast(h8, uint4{rnd_h2(st), rnd_h2(st), rnd_h2(st), rnd_h2(st)});
      }
    }
  f4 acc[4];
  for (int n = 0; n < 4; ++n) acc[n] = f4{0.f, 0.f, 0.f, 0.f};
  uint4 y0 = sy[lane], y1 = sy[64 + lane];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const h8 Ya = __builtin_bit_cast(h8, y0), Yb = __builtin_bit_cast(h8, y1);
      if (LDS) {
        y0 = sy[((st + 1) & 15) * 128 + lane];
        y1 = sy[((st + 1) & 15) * 128 + 64 + lane];
      } else {
        y0.x += 1; y1.y ^= y0.x;
      }
      __builtin_amdgcn_sched_barrier(0);
      const int s = st & 3;
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya, Wb[a][s], acc[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Yb, Wa[a][s], acc[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya, Wa[a][s], acc[a], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sum = 0.f;
  for (int n = 0; n < 4; ++n) sum += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  if (sum == 12345.678f) out[0] = sum;
}
template <int LDS, int RANDOM>
static void run_step(int waves_per_simd, float* d) {
  const int iters = 20000, blocks = 256, threads = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_step<LDS, RANDOM>), dim3(blocks), dim3(threads), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_step<LDS, RANDOM>), dim3(blocks), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * (threads / 64) * iters * 16.0 * 12 * 16 * 16 * 32 * 2;
  printf("layer-3 step pattern, %s operands, point fragments from %s, waves/SIMD %d: %.3f ms  %.1f TFLOP/s  %.3f of 2500\n",
         RANDOM ? "random" : "regular", LDS ? "LDS" : "registers", waves_per_simd, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0);
}
// the other two matrix instructions of the hot path on random operand bits: bf16 (weight gradients, bf16 x 3) and fp32 (rows kernels)
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
template <int KIND>      // 0: v_mfma_f32_16x16x32_bf16, 1: v_mfma_f32_16x16x4_f32
__global__ __launch_bounds__(512) void k_other(float* out, int iters, int random) {
  unsigned st = threadIdx.x * 977u + blockIdx.x * 131u + 7u;
  uint4 A[8], Bv[8];
  for (int i = 0; i < 8; ++i) {
    unsigned w[8];
    for (int c = 0; c < 8; ++c) {
      const unsigned r = rnd(st), e = rnd(st) >> 9;
      if (KIND == 0) w[c] = random ? ((r >> 4) & 0x807F807Fu) | ((120u + (e & 15u)) << 7) | ((120u + ((e >> 6) & 15u)) << 23) : 0x3F803F80u;
      else w[c] = random ? ((r >> 3) & 0x807FFFFFu) | ((120u + (e & 15u)) << 23) : 0x3F800000u;
    }
    A[i] = uint4{w[0], w[1], w[2], w[3]};
    Bv[i] = uint4{w[4], w[5], w[6], w[7]};
  }
  f4 acc[4];
  for (int n = 0; n < 4; ++n) acc[n] = f4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (KIND == 0)
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, A[i]), __builtin_bit_cast(b8, Bv[(i + n) & 7]), acc[n], 0, 0, 0);
        else
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, A[i].x), __builtin_bit_cast(float, Bv[(i + n) & 7].y), acc[n], 0, 0, 0);
      }
  }
  float sum = 0.f;
  for (int n = 0; n < 4; ++n) sum += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
  if (sum == 12345.678f) out[0] = sum;
}
template <int KIND>
static void run_other(int waves_per_simd, int random, float* d) {
  const int iters = 20000, blocks = 256, threads = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_other<KIND>, dim3(blocks), dim3(threads), 0, 0, d, 10, random);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_other<KIND>, dim3(blocks), dim3(threads), 0, 0, d, iters, random);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double k = KIND == 0 ? 32 : 4, nominal = KIND == 0 ? 2500.0 : 157.3;
  const double flop = (double)blocks * (threads / 64) * iters * 32.0 * 16 * 16 * k * 2;
  printf("%s, %s operands, waves/SIMD %d: %.3f ms  %.1f TFLOP/s  %.3f of %.1f\n", KIND == 0 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x4_f32",
         random ? "random" : "regular", waves_per_simd, ms, flop / ms * 1e-9, flop / ms * 1e-9 / nominal, nominal);
}
template <int NACC>
static void run(int waves_per_simd, float* d) {
  const int iters = 20000, blocks = 256, threads = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * (threads / 64) * iters * 4.0 * NACC * 16 * 16 * 32 * 2;
  printf("waves/SIMD %d  accumulators %d  %.3f ms  %.1f TFLOP/s  %.3f of 2500\n", waves_per_simd, NACC, ms, flop / ms * 1e-9,
         flop / ms * 1e-9 / 2500.0);
}
int main() {
  float* d;
  hipMalloc(&d, 4);
  for (int rep = 0; rep < 2; ++rep) {
    run<4>(1, d); run<4>(2, d); run<4>(4, d);
    run<8>(1, d); run<8>(2, d);
    run_step<0, 0>(1, d); run_step<0, 0>(2, d); run_step<1, 0>(1, d); run_step<1, 0>(2, d);
    run_step<1, 1>(1, d); run_step<1, 1>(2, d);
    run_other<0>(2, 0, d); run_other<0>(2, 1, d); run_other<1>(2, 0, d); run_other<1>(2, 1, d);
  }
  return 0;
}
