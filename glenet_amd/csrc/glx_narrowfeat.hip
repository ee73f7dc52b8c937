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

// ================================================================================================ the wide extractors' first layer
// Conv1d(C, 64, 1) + BatchNorm1d + ReLU of PointNetfeat (cvae_uncertainty/point_net.py:10-16) in training mode, in the same manner: the
// batch statistics of W1 x from the mean and covariance of x (k_narrow_pass<0>'s 44 sums), ONE pass that reads the points and writes
// h1 = relu(A (W1 x) + Cc) as (B P, 64) rows (the row kernels in front wrote the raw product, read it back for the statistics' transform and
// wrote h1: 0.38 ms per extractor at configs[3], 0.11 here), and for the backward ONE pass over the gradient of h1 that takes, per
// channel, sum dq, sum dq xhat and sum dq (x) x (dq = the gradient where the ReLU passes; mask and xhat recomputed from the point): the
// weight gradient follows from those and the moments of x,
//   dW[c][i] = A_c (sum dq_c x_i - (dbeta_c / N) sum x_i - (dgamma_c / N) is_c (W_c . M[:, i] - m_c sum x_i)),   M = sum x x^T.
#define L1_W 64
#define L1_NV (L1_W * (2 + NF_W))        // per channel: dbeta, dgamma, 8 products with x
#define L1_BLOCKS 2048                   // of the backward pass (a streaming read: eight blocks per CU)

__global__ void k_l1_coef(const double* __restrict__ T, int B, int C, int P, const float* __restrict__ w, const float* __restrict__ bias,
                          const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
                          float* __restrict__ rvar, float eps, float momentum, float* __restrict__ coef) {
  const int c = threadIdx.x;          // 64 threads
  const double N = (double)B * (double)P;
  double mean[NF_W];
  for (int k = 0; k < C; ++k) mean[k] = T[k] / N;
  double mu = 0.0, var = 0.0;
  for (int k = 0; k < C; ++k) mu += (double)w[c * C + k] * mean[k];
  for (int k = 0; k < C; ++k)
    for (int l = k; l < C; ++l) {
      const double cov = T[NF_W + k * NF_W - k * (k - 1) / 2 + (l - k)] / N - mean[k] * mean[l];
      var += (k == l ? 1.0 : 2.0) * (double)w[c * C + k] * (double)w[c * C + l] * cov;
    }
  if (var < 0.0) var = 0.0;
  const float inv = (float)(1.0 / sqrt(var + (double)eps));
  const float A = gamma[c] * inv;
  coef[c] = A;
  coef[L1_W + c] = beta[c] - A * (float)mu;
  coef[2 * L1_W + c] = (float)mu;
  coef[3 * L1_W + c] = inv;
  if (rmean) {
    const double unb = N > 1.0 ? var * N / (N - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * ((float)mu + (bias ? bias[c] : 0.f));
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

// 16 lanes per row, four channels per lane: a wave instruction writes (reads) four whole rows of h1 (of its gradient)
template <bool BWD, int CW>      // CW: 4 or 8 point features at most (the loops' trip counts)
__global__ __launch_bounds__(NF_THREADS) void k_l1_rows(const float* __restrict__ x, int B, int C, int P, const float* __restrict__ w,
                                                        const float* __restrict__ coef, float* __restrict__ h1,
                                                        const float* __restrict__ dh1, float* __restrict__ partial) {
  const int tid = threadIdx.x, cq = tid & 15, slot = tid >> 4;
  float wr[4][CW], A[4], Cc[4], m[4], is[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * cq + e;
#pragma unroll
    for (int k = 0; k < CW; ++k) wr[e][k] = k < C ? w[c * C + k] : 0.f;
    A[e] = coef[c]; Cc[e] = coef[L1_W + c]; m[e] = coef[2 * L1_W + c]; is[e] = coef[3 * L1_W + c];
  }
  float sb[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f}, sx[4][CW];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < CW; ++k) sx[e][k] = 0.f;
  for (long long b = blockIdx.x; b < B; b += gridDim.x)          // a block walks over whole objects (no division per row)
  for (int p = slot; p < P; p += 16) {
    const long long r = b * P + p;
    const float* xo = x + b * C * P + p;
    float xv[CW];
#pragma unroll
    for (int k = 0; k < CW; ++k) xv[k] = k < C ? xo[(long long)k * P] : 0.f;
    float u[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < CW; ++k) s = fmaf(wr[e][k], xv[k], s);
      u[e] = s;
    }
    if constexpr (!BWD) {
      float4 o;
      o.x = fmaxf(fmaf(A[0], u[0], Cc[0]), 0.f);
      o.y = fmaxf(fmaf(A[1], u[1], Cc[1]), 0.f);
      o.z = fmaxf(fmaf(A[2], u[2], Cc[2]), 0.f);
      o.w = fmaxf(fmaf(A[3], u[3], Cc[3]), 0.f);
      *reinterpret_cast<float4*>(h1 + r * L1_W + 4 * cq) = o;
    } else {
      const float4 g4 = *reinterpret_cast<const float4*>(dh1 + r * L1_W + 4 * cq);
      const float g[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dq = fmaf(A[e], u[e], Cc[e]) > 0.f ? g[e] : 0.f;
        sb[e] += dq;
        sg[e] = fmaf(dq, (u[e] - m[e]) * is[e], sg[e]);
#pragma unroll
        for (int k = 0; k < CW; ++k) sx[e][k] = fmaf(dq, xv[k], sx[e][k]);
      }
    }
  }
  if constexpr (BWD) {
    // the 16 row slots of the block: lanes 16 apart inside a wave (register-half swaps), then the four waves through LDS
    __shared__ float s_red[NF_THREADS / 64][16][4 * (2 + NF_W)];
    const int lane = tid & 63, wave = tid >> 6;
    auto quad = [&](float v) {
      const unsigned u_ = __builtin_bit_cast(unsigned, v);
      const auto s16 = __builtin_amdgcn_permlane16_swap(u_, u_, false, false);
      v = __builtin_bit_cast(float, (unsigned)s16[0]) + __builtin_bit_cast(float, (unsigned)s16[1]);
      const unsigned w_ = __builtin_bit_cast(unsigned, v);
      const auto s32 = __builtin_amdgcn_permlane32_swap(w_, w_, false, false);
      return __builtin_bit_cast(float, (unsigned)s32[0]) + __builtin_bit_cast(float, (unsigned)s32[1]);
    };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float vb = quad(sb[e]), vg = quad(sg[e]);
      if (lane < 16) { s_red[wave][cq][e * (2 + NF_W) + 0] = vb; s_red[wave][cq][e * (2 + NF_W) + 1] = vg; }
#pragma unroll
      for (int k = 0; k < NF_W; ++k) {
        const float vx = k < CW ? quad(sx[e][k < CW ? k : 0]) : 0.f;
        if (lane < 16) s_red[wave][cq][e * (2 + NF_W) + 2 + k] = vx;
      }
    }
    __syncthreads();
    for (int i = tid; i < L1_NV; i += NF_THREADS) {          // i = channel * 10 + j, channel = 4 cq + e
      const int c = i / (2 + NF_W), jj = i - c * (2 + NF_W);
      const int q_ = c >> 2, e = c & 3;
      float t = 0.f;
      for (int w_ = 0; w_ < NF_THREADS / 64; ++w_) t += s_red[w_][q_][e * (2 + NF_W) + jj];
      partial[(long long)blockIdx.x * L1_NV + i] = t;
    }
  }
}

__global__ __launch_bounds__(256) void k_l1_reduce(const float* __restrict__ partial, int nblocks, double* __restrict__ totals) {
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += (double)partial[(long long)b * L1_NV + blockIdx.x];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = red[0];
}

// grads: dW (64, C) | dgamma (64) | dbeta (64)
__global__ void k_l1_emit(const double* __restrict__ S, const double* __restrict__ T, int B, int C, int P, const float* __restrict__ w,
                          const float* __restrict__ coef, float* __restrict__ grads) {
  const int c = threadIdx.x;          // 64 threads
  const double N = (double)B * (double)P;
  const double db = S[c * (2 + NF_W)], dg = S[c * (2 + NF_W) + 1];
  const double A = (double)coef[c], m = (double)coef[2 * L1_W + c], is = (double)coef[3 * L1_W + c];
  for (int i = 0; i < C; ++i) {
    double wm = 0.0;                  // W_c . M[:, i], M = sum x x^T (upper triangle stored)
    for (int k = 0; k < C; ++k) {
      const int lo = k < i ? k : i, hi = k < i ? i : k;
      wm += (double)w[c * C + k] * T[NF_W + lo * NF_W - lo * (lo - 1) / 2 + (hi - lo)];
    }
    const double sxh = is * (wm - m * T[i]);      // sum_r xhat[r, c] x[r, i]
    grads[c * C + i] = (float)(A * (S[c * (2 + NF_W) + 2 + i] - (db / N) * T[i] - (dg / N) * sxh));
  }
  grads[L1_W * C + c] = (float)dg;
  grads[L1_W * C + L1_W + c] = (float)db;
}

extern "C" size_t glx_point_layer1_workspace_bytes(void) {
  return glx_align((size_t)L1_BLOCKS * L1_NV * sizeof(float)) + glx_align((size_t)L1_NV * sizeof(double));
}

// moments (44 doubles, out): the sums of x and of its products -- backward takes them again; coef (4, 64, out): A | Cc | mean | invstd.
extern "C" int glx_point_layer1_train_forward(const float* points, int B, int C, int P, const float* w, const float* bias, const float* gamma,
                                              const float* beta, float* rmean, float* rvar, float eps, float momentum, float* h1,
                                              float* coef, double* moments, void* workspace, size_t workspace_bytes, void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(points && w && gamma && beta && h1 && coef && moments && workspace, "glx_point_layer1_train_forward: null pointer");
  GLX_REQUIRE(C >= 1 && C <= NF_W && P >= 1, "glx_point_layer1_train_forward: 1 <= C <= 8 point features, P >= 1 (got %d, %d)", C, P);
  GLX_REQUIRE(workspace_bytes >= glx_point_layer1_workspace_bytes(), "glx_point_layer1_train_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  NarrowArgs a = {};
  a.x = points; a.B = B; a.C = C; a.P = P; a.w1 = w; a.w2 = w; a.w3 = w; a.partial = partial;      // (stage 0 reads the points only)
  const int nb = nf_blocks(B);
  hipLaunchKernelGGL(k_narrow_pass<0>, dim3(nb), dim3(NF_THREADS), 0, st, a);
  hipLaunchKernelGGL(k_narrow_reduce, dim3(NF_MOM), dim3(256), 0, st, partial, nb, moments, (float*)nullptr);
  hipLaunchKernelGGL(k_l1_coef, dim3(1), dim3(L1_W), 0, st, (const double*)moments, B, C, P, w, bias, gamma, beta, rmean, rvar, eps, momentum,
                     coef);
  const int rb = B < 4096 ? B : 4096;
  if (C <= 4)
    hipLaunchKernelGGL((k_l1_rows<false, 4>), dim3(rb), dim3(NF_THREADS), 0, st, points, B, C, P, w, (const float*)coef, h1,
                       (const float*)nullptr, (float*)nullptr);
  else
    hipLaunchKernelGGL((k_l1_rows<false, 8>), dim3(rb), dim3(NF_THREADS), 0, st, points, B, C, P, w, (const float*)coef, h1,
                       (const float*)nullptr, (float*)nullptr);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_point_layer1_train_backward(const float* points, int B, int C, int P, const float* w, const float* coef,
                                               const double* moments, const float* dh1, float* grads, void* workspace,
                                               size_t workspace_bytes, void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(points && w && coef && moments && dh1 && grads && workspace, "glx_point_layer1_train_backward: null pointer");
  GLX_REQUIRE(C >= 1 && C <= NF_W && P >= 1, "glx_point_layer1_train_backward: 1 <= C <= 8 point features, P >= 1 (got %d, %d)", C, P);
  GLX_REQUIRE(workspace_bytes >= glx_point_layer1_workspace_bytes(), "glx_point_layer1_train_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* partial = (float*)workspace;
  double* totals = (double*)((char*)workspace + glx_align((size_t)L1_BLOCKS * L1_NV * sizeof(float)));
  const int rb = B < L1_BLOCKS ? B : L1_BLOCKS;
  if (C <= 4)
    hipLaunchKernelGGL((k_l1_rows<true, 4>), dim3(rb), dim3(NF_THREADS), 0, st, points, B, C, P, w, coef, (float*)nullptr, dh1, partial);
  else
    hipLaunchKernelGGL((k_l1_rows<true, 8>), dim3(rb), dim3(NF_THREADS), 0, st, points, B, C, P, w, coef, (float*)nullptr, dh1, partial);
  hipLaunchKernelGGL(k_l1_reduce, dim3(L1_NV), dim3(256), 0, st, (const float*)partial, rb, totals);
  hipLaunchKernelGGL(k_l1_emit, dim3(1), dim3(L1_W), 0, st, (const double*)totals, moments, B, C, P, w, coef, grads);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
