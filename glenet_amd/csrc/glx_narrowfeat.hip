// The CVAE decoder's small point extractor in TRAINING mode (cvae_uncertainty/point_net.py:31-49 SimPointNetfeat(x = 0.5): Conv1d(C, 8, 1) +
// BatchNorm1d + ReLU, Conv1d(8, 8, 1) + BatchNorm1d + ReLU, Conv1d(8, 8, 1) + BatchNorm1d, max over the points; trained by
// model.py:200-243) without any intermediate tensor.  The layers are 8 wide and the batch is 2.1 M point rows at configs[3]: as layers
// (row kernels + BatchNorm passes) every stage moves 134 MB several times and the extractor costs 1.2 ms of the step for ~300 flops per
// row.  Here every pass reads the POINTS (16 bytes per row) and recomputes what it needs of the layers in front:
//   forward   S0: moments of x            -> BatchNorm 1's batch statistics (mean / variance of W1 x from the mean / covariance of x)
//             S1: moments of h1           -> BatchNorm 2's        S2: moments of h2 -> BatchNorm 3's
//             S3: y = bn3(conv3(h2)), max over the points of an object, the point it occurs at (lowest on ties), xhat there
//   backward  G:  dbeta3, dgamma3 from the output gradient and xhat at the extremes
//             S4: dz3 (the max's sparse gradient through BatchNorm 3: sparse + a row-wise affine part) -> dW3, dbeta2, dgamma2
//             S5: ... through layer 2 -> dW2, dbeta1, dgamma1         S6: ... through layer 1 -> dW1
// A pass leaves per-block partial sums (fp32 over a block's rows, a block walks over whole objects), k_narrow_reduce adds the blocks in a
// fixed order in double.  The convolutions' biases only move the batch means (they go into the running means; their gradient is exactly
// zero).  Coefficients per layer l (8 each): A = gamma invstd, Cc = beta - A mean(u), m = mean(u), is = invstd, for u = W h (no bias).
#include <hip/hip_runtime.h>
#include <cstdint>

#include "glx_common.h"
#include "../../include/glenet_hip.h"

#define NF_W 8                  // layer width
#define NF_THREADS 256
#define NF_BLOCKS 512
#define NF_MOM (NF_W + NF_W * (NF_W + 1) / 2)      // 44: sums + upper triangle of the products
#define NF_NV 80                // the most a pass accumulates (64 weight-gradient entries + 2 x 8 sums)

struct NarrowArgs {
  const float* x;               // (B, C, P)
  int B, C, P;
  const float *w1, *w2, *w3;    // (8, C), (8, 8), (8, 8)
  const float *b1, *b2, *b3;    // conv biases (forward: running means only) or NULL
  const float *gamma[3], *beta[3];
  float *rmean[3], *rvar[3];    // running statistics (forward, block 0) or NULL
  float eps, momentum;
  float* coef;                  // (3, 4, 8): A, Cc, m, is per layer
  const double* totals;         // the previous pass's sums (what it holds depends on the stage)
  float* partial;               // (blocks, NF_NV) this pass's sums
  // S3
  float* out; int* arg; float* xh_ext;
  // backward
  const float* gout;            // (B, 8)
  const int* argc; const float* xhc;
  const double* tot_g;          // dbeta3, dgamma3 (16)
  const double* tot_4;          // S4's totals: dW3 (64), dbeta2 (8), dgamma2 (8)
  const double* tot_5;          // S5's: dW2, dbeta1, dgamma1
};

__device__ __forceinline__ float nf_row_sum(float v) {
#define NF_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
  NF_DPP_ADD(0xB1);
  NF_DPP_ADD(0x4E);
  NF_DPP_ADD(0x141);
  NF_DPP_ADD(0x140);
#undef NF_DPP_ADD
  return v;
}
__device__ __forceinline__ float nf_wave_sum(float v) {     // all 64 lanes, fixed order
  v = nf_row_sum(v);
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = __builtin_bit_cast(float, (unsigned)s16[0]) + __builtin_bit_cast(float, (unsigned)s16[1]);
  const unsigned w = __builtin_bit_cast(unsigned, v);
  const auto s32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return __builtin_bit_cast(float, (unsigned)s32[0]) + __builtin_bit_cast(float, (unsigned)s32[1]);
}

// Layer l's coefficients from the moments of its input (K channels: sums, upper triangle of the products; N rows) and its K-column
// weight: thread c < 8.  mean(u_c) = w_c . mean(h), var(u_c) = w_c^T Cov(h) w_c.
__device__ void nf_layer_coef(const double* T, int K, double N, const float* w /* LDS (8, 8) */, const float* gamma, const float* beta,
                              float eps, float* A, float* Cc, float* m, float* is, int c, double* mean_out, double* var_out) {
  double mean[NF_W];
  for (int k = 0; k < K; ++k) mean[k] = T[k] / N;
  double mu = 0.0, var = 0.0;
  for (int k = 0; k < K; ++k) mu += (double)w[c * NF_W + k] * mean[k];
  for (int k = 0; k < K; ++k)
    for (int l = k; l < K; ++l) {
      const double cov = T[NF_W + k * NF_W - k * (k - 1) / 2 + (l - k)] / N - mean[k] * mean[l];      // (the layout is always 8 wide)
      const double ww = (double)w[c * NF_W + k] * (double)w[c * NF_W + l];
      var += (k == l ? 1.0 : 2.0) * ww * cov;
    }
  if (var < 0.0) var = 0.0;
  const float inv = (float)(1.0 / sqrt(var + (double)eps));
  const float a_ = gamma[c] * inv;
  A[c] = a_;
  m[c] = (float)mu;
  is[c] = inv;
  Cc[c] = beta[c] - a_ * (float)mu;
  *mean_out = mu;
  *var_out = var;
}

// Layer l + 1's coefficients from the moments of its input (one block of 8 threads, once per layer and step), the running statistics
__global__ void k_narrow_coef(NarrowArgs a, int l) {
  __shared__ float s_w[NF_W * NF_W];
  const int tid = threadIdx.x;
  const int K = l == 0 ? a.C : NF_W;
  const float* w = l == 0 ? a.w1 : l == 1 ? a.w2 : a.w3;
  for (int e = tid; e < NF_W * NF_W; e += blockDim.x) s_w[e] = (e & 7) < K ? w[(e >> 3) * K + (e & 7)] : 0.f;
  __syncthreads();
  if (tid >= NF_W) return;
  const double N = (double)a.B * (double)a.P;
  float A, Cc, m, is;
  double mu, var;
  float As[NF_W], Cs[NF_W], ms[NF_W], iss[NF_W];
  nf_layer_coef(a.totals, K, N, s_w, a.gamma[l], a.beta[l], a.eps, As, Cs, ms, iss, tid, &mu, &var);
  A = As[tid]; Cc = Cs[tid]; m = ms[tid]; is = iss[tid];
  a.coef[(l * 4 + 0) * NF_W + tid] = A;
  a.coef[(l * 4 + 1) * NF_W + tid] = Cc;
  a.coef[(l * 4 + 2) * NF_W + tid] = m;
  a.coef[(l * 4 + 3) * NF_W + tid] = is;
  if (a.rmean[l]) {
    const float* bp = l == 0 ? a.b1 : l == 1 ? a.b2 : a.b3;
    const float bias = bp ? bp[tid] : 0.f;
    const double unb = N > 1.0 ? var * N / (N - 1.0) : var;
    a.rmean[l][tid] = (1.f - a.momentum) * a.rmean[l][tid] + a.momentum * ((float)mu + bias);
    a.rvar[l][tid] = (1.f - a.momentum) * a.rvar[l][tid] + a.momentum * (float)unb;
  }
}

// STAGE 0 .. 2: moments of x / h1 / h2;  4 .. 6: the backward passes.  A block walks over objects blockIdx.x, + gridDim.x, ...
template <int STAGE>
__global__ __launch_bounds__(NF_THREADS) void k_narrow_pass(NarrowArgs a) {
  __shared__ float s_w1[NF_W * NF_W], s_w2[NF_W * NF_W], s_w3[NF_W * NF_W];
  __shared__ float s_A[3][NF_W], s_C[3][NF_W], s_m[3][NF_W], s_is[3][NF_W];
  __shared__ float s_kb[3][NF_W], s_kg[3][NF_W];          // backward: dbeta / N, dgamma / N of layers 3, 2, 1 (index layer - 1)
  __shared__ float s_g[NF_W];
  __shared__ int s_arg[NF_W];
  __shared__ float s_red[NF_THREADS / 64][NF_NV];
  const int tid = threadIdx.x;
  const double N = (double)a.B * (double)a.P;
  if (tid < NF_W * NF_W) {
    const int c = tid >> 3, k = tid & 7;
    s_w1[tid] = k < a.C ? a.w1[c * a.C + k] : 0.f;
    s_w2[tid] = a.w2[tid];
    s_w3[tid] = a.w3[tid];
  }
  // coefficients of the layers whose statistics are known: from `coef` (earlier passes wrote them)
  constexpr int KNOWN = STAGE <= 2 ? STAGE : 3;          // layers 1 .. KNOWN have coefficients in a.coef (k_narrow_coef wrote them)
  if (tid < NF_W) {
    for (int l = 0; l < KNOWN; ++l) {
      s_A[l][tid] = a.coef[(l * 4 + 0) * NF_W + tid];
      s_C[l][tid] = a.coef[(l * 4 + 1) * NF_W + tid];
      s_m[l][tid] = a.coef[(l * 4 + 2) * NF_W + tid];
      s_is[l][tid] = a.coef[(l * 4 + 3) * NF_W + tid];
    }
  }
  if constexpr (STAGE >= 4) {
    if (tid < NF_W) {
      s_kb[2][tid] = (float)(a.tot_g[tid] / N);
      s_kg[2][tid] = (float)(a.tot_g[NF_W + tid] / N);
      if (STAGE >= 5) { s_kb[1][tid] = (float)(a.tot_4[64 + tid] / N); s_kg[1][tid] = (float)(a.tot_4[72 + tid] / N); }
      if (STAGE >= 6) { s_kb[0][tid] = (float)(a.tot_5[64 + tid] / N); s_kg[0][tid] = (float)(a.tot_5[72 + tid] / N); }
    }
  }
  __syncthreads();

  constexpr int NV = STAGE <= 2 ? NF_MOM : (STAGE == 6 ? 64 : 80);
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.f;

  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    if constexpr (STAGE >= 4) {
      __syncthreads();
      if (tid < NF_W) { s_g[tid] = a.gout[(long long)b * NF_W + tid]; s_arg[tid] = a.argc[(long long)b * NF_W + tid]; }
      __syncthreads();
    }
    const float* xo = a.x + (long long)b * a.C * a.P;
    for (int p = tid; p < a.P; p += NF_THREADS) {
      float x[NF_W];
#pragma unroll
      for (int k = 0; k < NF_W; ++k) x[k] = k < a.C ? xo[(long long)k * a.P + p] : 0.f;
      if constexpr (STAGE == 0) {
#pragma unroll
        for (int k = 0; k < NF_W; ++k) acc[k] += x[k];
        int n = NF_W;
#pragma unroll
        for (int k = 0; k < NF_W; ++k)
#pragma unroll
          for (int l = k; l < NF_W; ++l) acc[n++] += x[k] * x[l];
        continue;
      }
      float u1[NF_W], h1[NF_W];
#pragma unroll
      for (int c = 0; c < NF_W; ++c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NF_W; ++k) s = fmaf(s_w1[c * NF_W + k], x[k], s);
        u1[c] = s;
        h1[c] = fmaxf(fmaf(s_A[0][c], s, s_C[0][c]), 0.f);
      }
      if constexpr (STAGE == 1) {
#pragma unroll
        for (int k = 0; k < NF_W; ++k) acc[k] += h1[k];
        int n = NF_W;
#pragma unroll
        for (int k = 0; k < NF_W; ++k)
#pragma unroll
          for (int l = k; l < NF_W; ++l) acc[n++] += h1[k] * h1[l];
        continue;
      }
      float u2[NF_W], h2[NF_W];
#pragma unroll
      for (int c = 0; c < NF_W; ++c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NF_W; ++k) s = fmaf(s_w2[c * NF_W + k], h1[k], s);
        u2[c] = s;
        h2[c] = fmaxf(fmaf(s_A[1][c], s, s_C[1][c]), 0.f);
      }
      if constexpr (STAGE == 2) {
#pragma unroll
        for (int k = 0; k < NF_W; ++k) acc[k] += h2[k];
        int n = NF_W;
#pragma unroll
        for (int k = 0; k < NF_W; ++k)
#pragma unroll
          for (int l = k; l < NF_W; ++l) acc[n++] += h2[k] * h2[l];
        continue;
      }
      if constexpr (STAGE >= 4) {
        // the max's gradient through BatchNorm 3: du3 = A3 (dy - dbeta3 / N - xhat3 dgamma3 / N), dy = g at the extreme's point
        float du3[NF_W];
#pragma unroll
        for (int c = 0; c < NF_W; ++c) {
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < NF_W; ++k) s = fmaf(s_w3[c * NF_W + k], h2[k], s);
          const float xh = (s - s_m[2][c]) * s_is[2][c];
          const float dy = s_arg[c] == p ? s_g[c] : 0.f;
          du3[c] = s_A[2][c] * (dy - s_kb[2][c] - xh * s_kg[2][c]);
        }
        float dq2[NF_W], xh2[NF_W];
#pragma unroll
        for (int k = 0; k < NF_W; ++k) {
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < NF_W; ++c) s = fmaf(s_w3[c * NF_W + k], du3[c], s);
          dq2[k] = h2[k] > 0.f ? s : 0.f;
          xh2[k] = (u2[k] - s_m[1][k]) * s_is[1][k];
        }
        if constexpr (STAGE == 4) {
#pragma unroll
          for (int c = 0; c < NF_W; ++c)
#pragma unroll
            for (int k = 0; k < NF_W; ++k) acc[c * NF_W + k] = fmaf(du3[c], h2[k], acc[c * NF_W + k]);
#pragma unroll
          for (int k = 0; k < NF_W; ++k) {
            acc[64 + k] += dq2[k];
            acc[72 + k] = fmaf(dq2[k], xh2[k], acc[72 + k]);
          }
          continue;
        }
        float du2[NF_W];
#pragma unroll
        for (int k = 0; k < NF_W; ++k) du2[k] = s_A[1][k] * (dq2[k] - s_kb[1][k] - xh2[k] * s_kg[1][k]);
        float dq1[NF_W], xh1[NF_W];
#pragma unroll
        for (int j = 0; j < NF_W; ++j) {
          float s = 0.f;
#pragma unroll
          for (int k = 0; k < NF_W; ++k) s = fmaf(s_w2[k * NF_W + j], du2[k], s);
          dq1[j] = h1[j] > 0.f ? s : 0.f;
          xh1[j] = (u1[j] - s_m[0][j]) * s_is[0][j];
        }
        if constexpr (STAGE == 5) {
#pragma unroll
          for (int k = 0; k < NF_W; ++k)
#pragma unroll
            for (int j = 0; j < NF_W; ++j) acc[k * NF_W + j] = fmaf(du2[k], h1[j], acc[k * NF_W + j]);
#pragma unroll
          for (int j = 0; j < NF_W; ++j) {
            acc[64 + j] += dq1[j];
            acc[72 + j] = fmaf(dq1[j], xh1[j], acc[72 + j]);
          }
          continue;
        }
        if constexpr (STAGE == 6) {
#pragma unroll
          for (int j = 0; j < NF_W; ++j) {
            const float du1 = s_A[0][j] * (dq1[j] - s_kb[0][j] - xh1[j] * s_kg[0][j]);
#pragma unroll
            for (int i = 0; i < NF_W; ++i) acc[j * NF_W + i] = fmaf(du1, x[i], acc[j * NF_W + i]);
          }
        }
      }
    }
  }
  // ---- the block's sums
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float v = nf_wave_sum(acc[i]);
    if (lane == 0) s_red[wave][i] = v;
  }
  __syncthreads();
  if (tid < NV) a.partial[(long long)blockIdx.x * NF_NV + tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
}

// totals[i] = sum over the blocks of partial[block][i], in double, fixed order; block i, 256 threads
__global__ __launch_bounds__(256) void k_narrow_reduce(const float* __restrict__ partial, int nblocks, double* __restrict__ totals,
                                                       float* __restrict__ totals_f) {
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += (double)partial[(long long)b * NF_NV + blockIdx.x];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    totals[blockIdx.x] = red[0];
    if (totals_f) totals_f[blockIdx.x] = (float)red[0];
  }
}

// S3: a WAVE per object (no barrier, nothing recomputed): y = A3 u3 + C3 over the object's points, per lane the running maximum per
// channel with the point it occurs at and xhat there; the 64 lanes meet by DPP / register-half swaps: the maximum, then the lowest point
// among the lanes that hold it, and that lane writes.
__device__ __forceinline__ float nf_wave_max(float v) {
#define NF_DPP_MAX(ctrl) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true)))
  NF_DPP_MAX(0xB1);
  NF_DPP_MAX(0x4E);
  NF_DPP_MAX(0x141);
  NF_DPP_MAX(0x140);
#undef NF_DPP_MAX
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = fmaxf(__builtin_bit_cast(float, (unsigned)s16[0]), __builtin_bit_cast(float, (unsigned)s16[1]));
  const unsigned w = __builtin_bit_cast(unsigned, v);
  const auto s32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)s32[0]), __builtin_bit_cast(float, (unsigned)s32[1]));
}
__device__ __forceinline__ int nf_wave_min(int v) {
#define NF_DPP_MIN(ctrl) { const int o_ = __builtin_amdgcn_update_dpp(0x7fffffff, v, ctrl, 0xF, 0xF, false); v = o_ < v ? o_ : v; }
  NF_DPP_MIN(0xB1);
  NF_DPP_MIN(0x4E);
  NF_DPP_MIN(0x141);
  NF_DPP_MIN(0x140);
#undef NF_DPP_MIN
  const auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  v = (int)s16[0] < (int)s16[1] ? (int)s16[0] : (int)s16[1];
  const auto s32 = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return (int)s32[0] < (int)s32[1] ? (int)s32[0] : (int)s32[1];
}
__global__ __launch_bounds__(NF_THREADS) void k_narrow_max(NarrowArgs a) {
  __shared__ float s_w1[NF_W * NF_W], s_w2[NF_W * NF_W], s_w3[NF_W * NF_W];
  __shared__ float s_A[3][NF_W], s_C[3][NF_W], s_m[3][NF_W], s_is[3][NF_W];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < NF_W * NF_W) {
    const int c = tid >> 3, k = tid & 7;
    s_w1[tid] = k < a.C ? a.w1[c * a.C + k] : 0.f;
    s_w2[tid] = a.w2[tid];
    s_w3[tid] = a.w3[tid];
  }
  if (tid < NF_W)
    for (int l = 0; l < 3; ++l) {
      s_A[l][tid] = a.coef[(l * 4 + 0) * NF_W + tid];
      s_C[l][tid] = a.coef[(l * 4 + 1) * NF_W + tid];
      s_m[l][tid] = a.coef[(l * 4 + 2) * NF_W + tid];
      s_is[l][tid] = a.coef[(l * 4 + 3) * NF_W + tid];
    }
  __syncthreads();
  for (long long b = (long long)blockIdx.x * (NF_THREADS / 64) + wave; b < a.B; b += (long long)gridDim.x * (NF_THREADS / 64)) {
    const float* xo = a.x + b * a.C * a.P;
    float best[NF_W], bx[NF_W];
    int bi[NF_W];
#pragma unroll
    for (int c = 0; c < NF_W; ++c) { best[c] = -3.402823466e38f; bi[c] = 0x7fffffff; bx[c] = 0.f; }
    for (int p = lane; p < a.P; p += 64) {
      float x[NF_W], h1[NF_W], h2[NF_W];
#pragma unroll
      for (int k = 0; k < NF_W; ++k) x[k] = k < a.C ? xo[(long long)k * a.P + p] : 0.f;
#pragma unroll
      for (int c = 0; c < NF_W; ++c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NF_W; ++k) s = fmaf(s_w1[c * NF_W + k], x[k], s);
        h1[c] = fmaxf(fmaf(s_A[0][c], s, s_C[0][c]), 0.f);
      }
#pragma unroll
      for (int c = 0; c < NF_W; ++c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NF_W; ++k) s = fmaf(s_w2[c * NF_W + k], h1[k], s);
        h2[c] = fmaxf(fmaf(s_A[1][c], s, s_C[1][c]), 0.f);
      }
#pragma unroll
      for (int c = 0; c < NF_W; ++c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NF_W; ++k) s = fmaf(s_w3[c * NF_W + k], h2[k], s);
        const float y = fmaf(s_A[2][c], s, s_C[2][c]);
        if (y > best[c]) { best[c] = y; bi[c] = p; bx[c] = (s - s_m[2][c]) * s_is[2][c]; }    // (a lane's points ascend: ties keep the lower)
      }
    }
#pragma unroll
    for (int c = 0; c < NF_W; ++c) {
      const float vmax = nf_wave_max(best[c]);
      const int imin = nf_wave_min(best[c] == vmax ? bi[c] : 0x7fffffff);
      if (best[c] == vmax && bi[c] == imin) {              // exactly one lane: a point belongs to one lane
        a.out[b * NF_W + c] = vmax;
        a.arg[b * NF_W + c] = imin;
        a.xh_ext[b * NF_W + c] = bx[c];
      }
    }
  }
}

// G: per block partial sums of g (8) and g xhat_ext (8) over its objects
__global__ __launch_bounds__(NF_THREADS) void k_narrow_gsum(const float* __restrict__ g, const float* __restrict__ xh, int B,
                                                            float* __restrict__ partial) {
  __shared__ float s_red[NF_THREADS / 64][2 * NF_W];
  const int tid = threadIdx.x, c = tid & 7, lane = tid & 63, wave = tid >> 6;
  float s0 = 0.f, s1 = 0.f;
  for (long long b = (long long)blockIdx.x * (NF_THREADS / NF_W) + (tid >> 3); b < B; b += (long long)gridDim.x * (NF_THREADS / NF_W)) {
    const float gv = g[b * NF_W + c];
    s0 += gv;
    s1 = fmaf(gv, xh[b * NF_W + c], s1);
  }
  // lanes with the same c: 8 apart inside a wave
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
  if (lane < NF_W) { s_red[wave][lane] = s0; s_red[wave][NF_W + lane] = s1; }
  __syncthreads();
  if (tid < 2 * NF_W) {
    float t = 0.f;
    for (int w = 0; w < NF_THREADS / 64; ++w) t += s_red[w][tid];
    partial[(long long)blockIdx.x * NF_NV + tid] = t;
  }
}

// grads (4, 64): dW1 is written by the last reduce; dW2, dW3 and the six BatchNorm vectors come out of the earlier passes' totals
__global__ void k_narrow_emit(const double* __restrict__ t, float* __restrict__ g) {
  const int i = threadIdx.x;
  g[64 + i] = (float)t[2 * NF_NV + i];                    // dW2 (S5)
  g[128 + i] = (float)t[NF_NV + i];                       // dW3 (S4)
  if (i < NF_W) {
    g[192 + i] = (float)t[2 * NF_NV + 72 + i];            // dgamma1
    g[200 + i] = (float)t[2 * NF_NV + 64 + i];            // dbeta1
    g[208 + i] = (float)t[NF_NV + 72 + i];                // dgamma2
    g[216 + i] = (float)t[NF_NV + 64 + i];                // dbeta2
    g[224 + i] = (float)t[NF_W + i];                      // dgamma3
    g[232 + i] = (float)t[i];                             // dbeta3
  }
}

// workspace: partial (NF_BLOCKS x NF_NV floats) + totals for S0, S1, S2 (forward) or G, S4, S5, S6 (backward): 4 x NF_NV doubles
extern "C" size_t glx_narrowfeat_workspace_bytes(void) {
  return glx_align((size_t)NF_BLOCKS * NF_NV * sizeof(float)) + glx_align((size_t)4 * NF_NV * sizeof(double));
}

static int nf_blocks(int B) { return B < NF_BLOCKS ? B : NF_BLOCKS; }

extern "C" int glx_narrowfeat_train_forward(const float* points, int B, int C, int P, const float* w1, const float* b1, const float* gamma1,
                                            const float* beta1, float* rmean1, float* rvar1, const float* w2, const float* b2,
                                            const float* gamma2, const float* beta2, float* rmean2, float* rvar2, const float* w3,
                                            const float* b3, const float* gamma3, const float* beta3, float* rmean3, float* rvar3,
                                            float eps, float momentum, float* out, int32_t* arg, float* xh_ext, float* coef,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(points && w1 && w2 && w3 && gamma1 && beta1 && gamma2 && beta2 && gamma3 && beta3 && out && arg && xh_ext && coef && workspace,
              "glx_narrowfeat_train_forward: null pointer");
  GLX_REQUIRE(C >= 1 && C <= NF_W && P >= 1, "glx_narrowfeat_train_forward: 1 <= C <= 8 point features, P >= 1 (got %d, %d)", C, P);
  GLX_REQUIRE(workspace_bytes >= glx_narrowfeat_workspace_bytes(), "glx_narrowfeat_train_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  double* totals = (double*)((char*)workspace + glx_align((size_t)NF_BLOCKS * NF_NV * sizeof(float)));
  NarrowArgs a = {};
  a.x = points; a.B = B; a.C = C; a.P = P;
  a.w1 = w1; a.w2 = w2; a.w3 = w3; a.b1 = b1; a.b2 = b2; a.b3 = b3;
  a.gamma[0] = gamma1; a.gamma[1] = gamma2; a.gamma[2] = gamma3;
  a.beta[0] = beta1; a.beta[1] = beta2; a.beta[2] = beta3;
  a.rmean[0] = rmean1; a.rmean[1] = rmean2; a.rmean[2] = rmean3;
  a.rvar[0] = rvar1; a.rvar[1] = rvar2; a.rvar[2] = rvar3;
  a.eps = eps; a.momentum = momentum; a.coef = coef; a.partial = partial;
  a.out = out; a.arg = (int*)arg; a.xh_ext = xh_ext;
  const int nb = nf_blocks(B);
  a.totals = totals;
  hipLaunchKernelGGL(k_narrow_pass<0>, dim3(nb), dim3(NF_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_narrow_reduce, dim3(NF_MOM), dim3(256), 0, st, partial, nb, totals, (float*)nullptr);
  hipLaunchKernelGGL(k_narrow_coef, dim3(1), dim3(64), 0, st, a, 0);
  hipLaunchKernelGGL(k_narrow_pass<1>, dim3(nb), dim3(NF_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_narrow_reduce, dim3(NF_MOM), dim3(256), 0, st, partial, nb, totals, (float*)nullptr);
  hipLaunchKernelGGL(k_narrow_coef, dim3(1), dim3(64), 0, st, a, 1);
  hipLaunchKernelGGL(k_narrow_pass<2>, dim3(nb), dim3(NF_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_narrow_reduce, dim3(NF_MOM), dim3(256), 0, st, partial, nb, totals, (float*)nullptr);
  hipLaunchKernelGGL(k_narrow_coef, dim3(1), dim3(64), 0, st, a, 2);
  hipLaunchKernelGGL(k_narrow_max, dim3((B + 3) / 4 < 1024 ? (B + 3) / 4 : 1024), dim3(NF_THREADS), 0, st, a);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// grads: (4, 64) floats = dW1 (8 x 8, the first C columns count), dW2, dW3, then dgamma1, dbeta1, dgamma2, dbeta2, dgamma3, dbeta3 (8 each:
// 48 of the last 64).
extern "C" int glx_narrowfeat_train_backward(const float* points, int B, int C, int P, const float* w1, const float* w2, const float* w3,
                                             const float* coef, const float* gout, const int32_t* arg, const float* xh_ext, float* grads,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(points && w1 && w2 && w3 && coef && gout && arg && xh_ext && grads && workspace, "glx_narrowfeat_train_backward: null pointer");
  GLX_REQUIRE(C >= 1 && C <= NF_W && P >= 1, "glx_narrowfeat_train_backward: 1 <= C <= 8 point features, P >= 1 (got %d, %d)", C, P);
  GLX_REQUIRE(workspace_bytes >= glx_narrowfeat_workspace_bytes(), "glx_narrowfeat_train_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  double* totals = (double*)((char*)workspace + glx_align((size_t)NF_BLOCKS * NF_NV * sizeof(float)));
  NarrowArgs a = {};
  a.x = points; a.B = B; a.C = C; a.P = P;
  a.w1 = w1; a.w2 = w2; a.w3 = w3;
  a.coef = const_cast<float*>(coef); a.partial = partial;
  a.gout = gout; a.argc = (const int*)arg; a.xhc = xh_ext;
  a.tot_g = totals; a.tot_4 = totals + NF_NV; a.tot_5 = totals + 2 * NF_NV;
  const int nb = nf_blocks(B);
  const int gb = B / (NF_THREADS / NF_W) + 1 < 64 ? B / (NF_THREADS / NF_W) + 1 : 64;
  hipLaunchKernelGGL(k_narrow_gsum, dim3(gb), dim3(NF_THREADS), 0, st, gout, xh_ext, B, partial);
  // dbeta3 -> grads[3][40 ..], dgamma3 -> grads[3][32 ..]: the reduce writes (dbeta3, dgamma3) in its own order, emitted below
  hipLaunchKernelGGL(k_narrow_reduce, dim3(2 * NF_W), dim3(256), 0, st, partial, gb, totals, (float*)nullptr);
  hipLaunchKernelGGL(k_narrow_pass<4>, dim3(nb), dim3(NF_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_narrow_reduce, dim3(NF_NV), dim3(256), 0, st, partial, nb, totals + NF_NV, (float*)nullptr);
  hipLaunchKernelGGL(k_narrow_pass<5>, dim3(nb), dim3(NF_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_narrow_reduce, dim3(NF_NV), dim3(256), 0, st, partial, nb, totals + 2 * NF_NV, (float*)nullptr);
  hipLaunchKernelGGL(k_narrow_pass<6>, dim3(nb), dim3(NF_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_narrow_reduce, dim3(64), dim3(256), 0, st, partial, nb, totals + 3 * NF_NV, grads);
  hipLaunchKernelGGL(k_narrow_emit, dim3(1), dim3(64), 0, st, (const double*)totals, grads);       // the other gradients, from the doubles
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
