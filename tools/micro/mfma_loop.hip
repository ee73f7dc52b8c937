// Microbenchmark (GPU box): cycles per step of the sparse-conv multiply loop.
//   mode 0: 4 independent fp32 16x16x4 MFMAs per step, operands in registers
//   mode 1: + one ds_read_b128 of the W fragment per step, prefetch distance 1 (what the kernel does)
//   mode 2: same, prefetch distance 3
//   mode 3: two row tiles per W read (8 MFMAs per ds_read_b128), distance 1
//   mode 4: ONE dependent accumulator chain (16 MFMAs), W fragment read as 4 ds_read_b128 up front
//   mode 5: two independent chains
// hipcc -O3 --offload-arch=gfx950 mfma_loop.hip -o mfma_loop && ./mfma_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* __restrict__ out,
                                         long long* __restrict__ cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float s[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = w[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a0 = w[lane], a1 = w[lane + 64];
  const float* bp = s + q * 1024 + r * 4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a0, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, a0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, a1, acc[3], 0, 0, 0);
      }
    } else if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        f32x4 bv = *reinterpret_cast<const f32x4*>(bp + t * 64);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[0], a0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[1], a0, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[2], a0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[3], a0, acc[3], 0, 0, 0);
        if constexpr (MODE == 3) {
          acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[0], a1, acc[4], 0, 0, 0);
          acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[1], a1, acc[5], 0, 0, 0);
          acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[2], a1, acc[6], 0, 0, 0);
          acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[3], a1, acc[7], 0, 0, 0);
        }
      }
    } else if constexpr (MODE == 4 || MODE == 5) {
      const float* cp = s + (q * 16 + r) * 20;
      f32x4 wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wv[i] = *reinterpret_cast<const f32x4*>(cp + 4 * i);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][e], a0, acc[0], 0, 0, 0);
          if constexpr (MODE == 5) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][e], a1, acc[1], 0, 0, 0);
        }
      }
    } else {
      f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 64),
            b2 = *reinterpret_cast<const f32x4*>(bp + 128);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        f32x4 b3 = *reinterpret_cast<const f32x4*>(bp + ((t + 3) & 15) * 64);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[0], a0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[1], a0, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[2], a0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[3], a0, acc[3], 0, 0, 0);
        b0 = b1; b1 = b2; b2 = b3;
      }
    }
  }
  long long t1 = clock64();
  f32x4 sum = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum[0] + sum[1] + sum[2] + sum[3];
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int threads, int mfma_per_step) {
  float *w, *out; long long* cyc;
  int waves = blocks * threads / 64;
  hipMalloc(&w, 4096 * 4); hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, waves * 8);
  std::vector<float> hw(4096, 0.5f);
  hipMemcpy(w, hw.data(), 4096 * 4, hipMemcpyHostToDevice);
  const int iters = 200;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, threads, 16384>>>(w, out, cyc, iters);
  hipEventRecord(e0);
  k<MODE><<<blocks, threads, 16384>>>(w, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> hc(waves);
  hipMemcpy(hc.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto c : hc) mean += c; mean /= waves;
  double steps = (double)iters * 16;
  printf("%-34s blocks %5d x %3d thr: %7.1f ticks/step (%5.1f per MFMA), kernel %.1f us -> %.1f ns/step, %.1f TFLOP/s\n",
         name, blocks, threads, mean / steps, mean / steps / mfma_per_step, ms * 1e3, ms * 1e6 / steps,
         (double)waves * steps * mfma_per_step * 2048 / (ms * 1e-3) / 1e12);
  hipFree(w); hipFree(out); hipFree(cyc);
}

int main() {
  for (int occ = 1; occ <= 3; ++occ) {
    printf("-- %d wave(s) per SIMD\n", occ);
    run<0>("mfma only", 256 * occ, 256, 4);
    run<1>("ds_read_b128 dist 1", 256 * occ, 256, 4);
    run<2>("ds_read_b128 dist 3", 256 * occ, 256, 4);
    run<3>("2 row tiles per read", 256 * occ, 256, 8);
    run<4>("1 dependent chain x16", 256 * occ, 256, 1);
    run<5>("2 chains x16", 256 * occ, 256, 2);
  }
  return 0;
}
