// Training-mode BatchNorm1d (+ ReLU) over the rows of a sparse tensor's feature matrix (N, C):
// the nn.BatchNorm1d / nn.ReLU pairs that follow every sparse conv in the backbone
// (pcdet/models/backbones_3d/spconv_backbone.py:21-25, eps 1e-3, momentum 0.01).  In eval mode
// they are folded into the conv epilogue; in training mode PyTorch spends more device time on
// them (per layer: statistics 41 us + transform + ReLU + three backward kernels) than on the
// convolutions.  Here: forward = column statistics (fp64 partial sums, fixed reduction order,
// so bitwise reproducible) + a one-block finalize + one fused normalise/affine/ReLU pass; backward
// = column sums of dz and dz*xhat + finalize + one fused dx pass.  HBM-bound (x is read twice).
#include "glx_common.h"
#include "glx_bn_state.h"

typedef float bf32x4 __attribute__((ext_vector_type(4)));

// 512 threads: a statistics block keeps 8 (forward) or 2 x 4 (backward) 16-byte loads per thread in flight, 64 KB per CU
// with one block per CU -- with 256 threads (32 KB per CU) the 72 MB layers read at 3.6-4 TB/s; measured on the training
// step 11.99 / 11.89 ms at 256 against 11.80 at 512 (-DBN_THREADS=... through GLX_HIPCC_EXTRA to repeat it)
#ifndef BN_THREADS
#define BN_THREADS 512
#endif
#define BN_SLABS 256   // row slabs = blocks of the statistics kernels (one per CU) on the fixed-order path
// with a state buffer the slab count is free, but more blocks did not pay: measured on the training step (caps of the
// statistics / transform grids 256/512: 12.97 ms, 512/1024: 13.05, 1024/2048: 13.17, 2048/4096: 13.26 -- more blocks queue
// on the accumulator atomics); env GLX_BN_STATS_BLOCKS / GLX_BN_APPLY_BLOCKS override the caps
static int bn_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static int bn_stats_slabs(int N, int C) {
  static const int cap = bn_env("GLX_BN_STATS_BLOCKS", 256);
  const long long want = ((long long)N * C * 4 + 32767) / 32768;
  const int lo = glx_divup(N, 8) > BN_SLABS ? BN_SLABS : glx_divup(N, 8);
  return want <= lo ? lo : (want > cap ? cap : (int)want);
}
static int bn_apply_blocks(int N, int C) {
  static const int cap = bn_env("GLX_BN_APPLY_BLOCKS", 512);
  const int want = glx_divup((long long)N * C / 4, BN_THREADS * 4);
  return want > cap ? cap : (want < 1 ? 1 : want);
}
#define BN_FIN_THREADS 1024   // the one-block finalize kernels: 1024 / C slab groups per channel

// partial[slab][2][C] (fp64): column sums of (a, a*b) over the slab's rows.
//   forward : a = x,  b = x                         -> sum x, sum x^2
//   backward: a = dz, b = xhat = (x - mean)*invstd  -> sum dz, sum dz*xhat   (dz = dy * [y > 0])
template <bool BWD>
__device__ __forceinline__ void bn_slab_sums(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, int relu, int N, int C, long long dy_pitch, double (&a0)[4],
    double (&a1)[4]) {
  const int c4n = C >> 2;                       // float4 columns
  const int col = threadIdx.x % c4n, rlane = threadIdx.x / c4n;
  const int rstep = BN_THREADS / c4n;           // rows covered per pass by the block
  const int rows_per_slab = (N + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per_slab, r1 = min(N, r0 + rows_per_slab);
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  bf32x4 mu = bf32x4{0, 0, 0, 0}, is = mu;
  float sc[4] = {0, 0, 0, 0}, sh[4] = {0, 0, 0, 0};
  const bool remask = BWD && relu && !y;        // ReLU mask from x (y not read)
  if (BWD) {
    mu = *reinterpret_cast<const bf32x4*>(mean + 4 * col);
    is = *reinterpret_cast<const bf32x4*>(invstd + 4 * col);
    if (remask) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float gm = gamma ? gamma[4 * col + i] : 1.f, bt = beta ? beta[4 * col + i] : 0.f;
        sc[i] = bn_scale(is[i], gm);
        sh[i] = bn_shift(bt, mu[i], is[i], gm);
      }
    }
  }
  if (rlane < rstep) {
    // U rows per trip with all their loads issued first: a slab is ~190 rows over 16 row lanes, and one dependent
    // 16-byte load per trip left the kernel latency-bound (14 us for 12 MB).  The forward pass has one stream (x)
    // where backward has two: 8 rows keep as many loads in flight.
    constexpr int U = BWD ? 4 : 8;
    for (int r = r0 + rlane; r < r1; r += U * rstep) {
      bf32x4 xv[U], g[U], yv[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int rr = r + j * rstep;
        const long long o = (long long)(rr < r1 ? rr : r) * C + 4 * col;
        xv[j] = *reinterpret_cast<const bf32x4*>(x + o);
        if (BWD) {
          g[j] = *reinterpret_cast<const bf32x4*>(dy + (long long)(rr < r1 ? rr : r) * dy_pitch + 4 * col);
          if (relu && !remask) yv[j] = *reinterpret_cast<const bf32x4*>(y + o);
        }
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if (r + j * rstep >= r1) break;
        if (BWD) {
          if (remask) {
#pragma unroll
            for (int i = 0; i < 4; ++i) g[j][i] = bn_affine(xv[j][i], sc[i], sh[i]) > 0.f ? g[j][i] : 0.f;
          } else if (relu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) g[j][i] = yv[j][i] > 0.f ? g[j][i] : 0.f;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s0[i] += (double)g[j][i];
            s1[i] += (double)g[j][i] * (double)((xv[j][i] - mu[i]) * is[i]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s0[i] += (double)xv[j][i];
            s1[i] += (double)xv[j][i] * (double)xv[j][i];
          }
        }
      }
    }
  }
  // fixed-order reduction over the row lanes of the block
  __shared__ double red[2][BN_THREADS][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { red[0][threadIdx.x][i] = s0[i]; red[1][threadIdx.x][i] = s1[i]; }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) a0[i] = a1[i] = 0;
  if (threadIdx.x < c4n) {       // these threads return the slab's sums of float4 column threadIdx.x
    for (int l = 0; l < rstep; ++l) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0[i] += red[0][l * c4n + threadIdx.x][i];
        a1[i] += red[1][l * c4n + threadIdx.x][i];
      }
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(BN_THREADS) void k_bn_partial(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, int relu, int N, int C, long long dy_pitch,
    const int* __restrict__ n_live, double* __restrict__ partial) {
  if (n_live) N = min(N, *n_live);
  double a0[4], a1[4];
  bn_slab_sums<BWD>(x, dy, y, mean, invstd, gamma, beta, relu, N, C, dy_pitch, a0, a1);
  if (threadIdx.x < (C >> 2)) {
    double* p = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) { p[4 * threadIdx.x + i] = a0[i]; p[C + 4 * threadIdx.x + i] = a1[i]; }
  }
}

// column sums of the slab partials, all 256 threads busy: thread (c, g) adds slabs g, g+G, ...
// in a fixed order, then the G groups are combined in LDS.
__device__ __forceinline__ void bn_column_sums(const double* __restrict__ partial, int slabs, int C,
                                               int c, int g, int G, double (*s_red)[2],
                                               double& s, double& ss) {
  s = 0; ss = 0;
  for (int b = g; b < slabs; b += G) { s += partial[(size_t)b * 2 * C + c]; ss += partial[(size_t)b * 2 * C + C + c]; }
  s_red[threadIdx.x][0] = s;
  s_red[threadIdx.x][1] = ss;
  __syncthreads();
  if (g == 0) {
    for (int k = 1; k < G; ++k) { s += s_red[k * C + c][0]; ss += s_red[k * C + c][1]; }
  }
  __syncthreads();
}

// one block: statistics -> scale / shift (coef[0..C), coef[C..2C)), saved mean / invstd, running stats
__global__ __launch_bounds__(BN_FIN_THREADS) void k_bn_finalize_fwd(
    const double* __restrict__ partial, int slabs, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, int N, int C,
    const int* __restrict__ n_live, float* __restrict__ coef, float* __restrict__ save_mean,
    float* __restrict__ save_invstd, float* __restrict__ running_mean,
    float* __restrict__ running_var) {
  __shared__ double s_red[BN_FIN_THREADS][2];
  int n = N;
  if (n_live) n = min(N, *n_live);
  const int CB = C < BN_FIN_THREADS ? C : BN_FIN_THREADS;      // channels per pass
  const int G = BN_FIN_THREADS / CB;
  for (int c0 = 0; c0 < C; c0 += CB) {
    const int c = c0 + threadIdx.x % CB, g = threadIdx.x / CB;
    double s, ss;
    bn_column_sums(partial, slabs, C, c, g, G, s_red, s, ss);
    if (g == 0) {
      const double cnt = n > 0 ? (double)n : 1.0;
      const double m = s / cnt;
      double var = ss / cnt - m * m;
      if (var < 0) var = 0;
      const float is = (float)(1.0 / sqrt(var + (double)eps));
      const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
      coef[c] = bn_scale(is, gm);
      coef[C + c] = bn_shift(bt, (float)m, is, gm);
      save_mean[c] = (float)m;
      save_invstd[c] = is;
      if (running_mean) {   // nn.BatchNorm semantics: unbiased variance in the running estimate
        const double unb = n > 1 ? var * cnt / (cnt - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    }
  }
}

// The transform kernels: a thread keeps ONE float4 column (its four coefficients live in registers) and strides over
// the rows -- no index division, no LDS lookups per element; memory order is that of the flat loop (a block covers
// BN_THREADS / (C/4) consecutive rows per trip).
__global__ __launch_bounds__(BN_THREADS) void k_bn_forward_apply(
    const float* __restrict__ x, const float* __restrict__ coef, int relu, int N, int C,
    const int* __restrict__ n_live, float* __restrict__ y, long long y_pitch, const float* __restrict__ res) {
  // res != NULL (N x C, dense): y = relu?(x scale + shift + res) -- the identity branch of a residual block joins here
  int n = N;
  if (n_live) n = min(N, *n_live);
  const int c4n = C >> 2;
  const int col = threadIdx.x % c4n, rl = threadIdx.x / c4n, rpb = BN_THREADS / c4n;
  const bf32x4 sc = *reinterpret_cast<const bf32x4*>(coef + 4 * col);
  const bf32x4 sh = *reinterpret_cast<const bf32x4*>(coef + C + 4 * col);
  for (int r = blockIdx.x * rpb + rl; r < n; r += gridDim.x * rpb) {
    bf32x4 v = *reinterpret_cast<const bf32x4*>(x + (long long)r * C + 4 * col);
    bf32x4 rv = bf32x4{0.f, 0.f, 0.f, 0.f};
    if (res) rv = *reinterpret_cast<const bf32x4*>(res + (long long)r * C + 4 * col);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = bn_affine(v[i], sc[i], sh[i]) + rv[i];
      v[i] = relu ? fmaxf(t, 0.f) : t;
    }
    *reinterpret_cast<bf32x4*>(y + (long long)r * y_pitch + 4 * col) = v;
  }
  // rows past the live count of a capacity-sized matrix: zeros, so that nothing downstream (a GEMM's weight
  // gradient multiplies them by zero gradients) ever meets the NaN an uninitialised row may hold
  for (int r = n + blockIdx.x * rpb + rl; r < N; r += gridDim.x * rpb)
    *reinterpret_cast<bf32x4*>(y + (long long)r * y_pitch + 4 * col) = bf32x4{0.f, 0.f, 0.f, 0.f};
}

// one block: dgamma / dbeta and the coefficients of dx = a * (dz - b - xhat * cc)
__global__ __launch_bounds__(BN_FIN_THREADS) void k_bn_finalize_bwd(
    const double* __restrict__ partial, int slabs, const float* __restrict__ gamma,
    const float* __restrict__ invstd, int N, int C, const int* __restrict__ n_live,
    float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double s_red[BN_FIN_THREADS][2];
  int n = N;
  if (n_live) n = min(N, *n_live);
  const int CB = C < BN_FIN_THREADS ? C : BN_FIN_THREADS;
  const int G = BN_FIN_THREADS / CB;
  for (int c0 = 0; c0 < C; c0 += CB) {
    const int c = c0 + threadIdx.x % CB, g = threadIdx.x / CB;
    double s, sx;
    bn_column_sums(partial, slabs, C, c, g, G, s_red, s, sx);
    if (g == 0) {
      const double cnt = n > 0 ? (double)n : 1.0;
      coef[c] = (gamma ? gamma[c] : 1.f) * invstd[c];
      coef[C + c] = (float)(s / cnt);
      coef[2 * C + c] = (float)(sx / cnt);
      if (dgamma) dgamma[c] = (float)sx;
      if (dbeta) dbeta[c] = (float)s;
    }
  }
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_backward_apply(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ coef, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu, int N, int C,
    long long dy_pitch, const int* __restrict__ n_live, float* __restrict__ dx) {
  int n = N;
  if (n_live) n = min(N, *n_live);
  const bool remask = relu && !y;
  const int c4n = C >> 2;
  const int col = threadIdx.x % c4n, rl = threadIdx.x / c4n, rpb = BN_THREADS / c4n;
  const bf32x4 a = *reinterpret_cast<const bf32x4*>(coef + 4 * col);
  const bf32x4 b = *reinterpret_cast<const bf32x4*>(coef + C + 4 * col);
  const bf32x4 cc = *reinterpret_cast<const bf32x4*>(coef + 2 * C + 4 * col);
  const bf32x4 mu = *reinterpret_cast<const bf32x4*>(mean + 4 * col);
  const bf32x4 is = *reinterpret_cast<const bf32x4*>(invstd + 4 * col);
  float sc[4], sh[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float gm = gamma ? gamma[4 * col + i] : 1.f, bt = beta ? beta[4 * col + i] : 0.f;
    sc[i] = bn_scale(is[i], gm);
    sh[i] = bn_shift(bt, mu[i], is[i], gm);
  }
  for (int r = blockIdx.x * rpb + rl; r < n; r += gridDim.x * rpb) {
    const long long o = (long long)r * C + 4 * col;
    const bf32x4 xv = *reinterpret_cast<const bf32x4*>(x + o);
    bf32x4 g = *reinterpret_cast<const bf32x4*>(dy + (long long)r * dy_pitch + 4 * col);
    if (remask) {
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = bn_affine(xv[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
    } else if (relu) {
      const bf32x4 yv = *reinterpret_cast<const bf32x4*>(y + o);
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = yv[i] > 0.f ? g[i] : 0.f;
    }
    bf32x4 ov;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (xv[i] - mu[i]) * is[i];
      ov[i] = a[i] * (g[i] - b[i] - xh * cc[i]);
    }
    *reinterpret_cast<bf32x4*>(dx + o) = ov;
  }
  for (int r = n + blockIdx.x * rpb + rl; r < N; r += gridDim.x * rpb)      // dead rows: zero gradient
    *reinterpret_cast<bf32x4*>(dx + (long long)r * C + 4 * col) = bf32x4{0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------ statistics + finalize in one launch
// The one-block finalize kernels above are launch latency (8.8 us each, 64 of them in a GLENet-VR training step,
// for a few hundred flops).  With a persistent accumulator the statistics kernel finishes the job itself: every
// block adds its slab sums to one of BN_SETS accumulator sets with fp64 atomics (16 sets: at most 16 blocks
// queue on an address), takes a ticket, and the block that draws the last ticket sums the sets -- exchanging them
// for zero, so the state is clean for the next call -- and does the finalize.
// Ordering without fences (a release fence writes the XCD's whole dirty L2 back -- the conv output that was just
// produced -- and made an earlier version of this slower than the separate kernel): the atomics execute at the
// device-coherent level and RETURN their old value; a block consumes the returns (s_waitcnt) before its ticket
// atomic is issued, so whoever sees the last ticket finds every contribution in place.  The sums of a set arrive in
// a run-dependent order: the fp64 totals can differ by an ulp of 2^-52 between runs, their float roundings
// practically never -- callers that need the fixed-order guarantee pass state = NULL (three launches).
// (BnState, BnFinalize, bn_contribute, bn_finalize_sets: glx_bn_state.h, shared with the sparse-conv epilogue)

template <bool BWD>
__global__ __launch_bounds__(BN_THREADS) void k_bn_stats(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ invstd, int relu, int N, int C,
    long long dy_pitch, const int* __restrict__ n_live, BnState* __restrict__ st, BnFinalize f) {
  __shared__ int s_last;
  if (n_live) N = min(N, *n_live);
  double a0[4], a1[4];
  bn_slab_sums<BWD>(x, dy, y, mean, invstd, f.gamma, f.beta, relu, N, C, dy_pitch, a0, a1);
  if (!bn_contribute(st, C, a0, a1, gridDim.x, &s_last)) return;
  __shared__ double s_fin[BN_THREADS][2];
  bn_finalize_sets<BWD, BN_THREADS>(st, f, C, N, s_fin);
}

extern "C" size_t glx_bn_state_bytes(void) { return glx_align(sizeof(BnState)); }

// ------------------------------------------------------------------ short matrices: one launch
// The RoI head's FC towers normalise (512, 256) matrices: the three-launch scheme above (and torch's five
// forward / three backward kernels) is pure launch latency there.  A block owns 16 channels over ALL rows:
// column statistics (fp64, fixed order: 16 row lanes per wave by shuffles, then the 4 waves in LDS), the
// finalize and the transform in one kernel; x is read a second time from L2 for the transform.
#define BN_SMALL_N 4096
#define BN_SMALL_CQ 4    // float4 columns = 16 channels per block

// sum of v over the threads of the block that share `q` (= threadIdx.x % 4); every thread gets the result
__device__ __forceinline__ void bn_small_reduce(double (&s0)[4], double (&s1)[4], double (*s_part)[BN_SMALL_CQ][8]) {
#pragma unroll
  for (int off = BN_SMALL_CQ; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s0[i] += __shfl_xor(s0[i], off, 64);
      s1[i] += __shfl_xor(s1[i], off, 64);
    }
  }
  const int q = threadIdx.x % BN_SMALL_CQ, wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < BN_SMALL_CQ) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { s_part[wave][q][i] = s0[i]; s_part[wave][q][4 + i] = s1[i]; }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s0[i] = s_part[0][q][i];
    s1[i] = s_part[0][q][4 + i];
    for (int w = 1; w < BN_THREADS / 64; ++w) { s0[i] += s_part[w][q][i]; s1[i] += s_part[w][q][4 + i]; }
  }
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_small_forward(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, float momentum, int relu, int N, int C, const int* __restrict__ n_live,
    float* __restrict__ y, long long y_pitch, float* __restrict__ save_mean, float* __restrict__ save_invstd,
    float* __restrict__ running_mean, float* __restrict__ running_var) {
  __shared__ double s_part[BN_THREADS / 64][BN_SMALL_CQ][8];
  int n = N;
  if (n_live) n = min(N, *n_live);
  const int q = threadIdx.x % BN_SMALL_CQ, rl = threadIdx.x / BN_SMALL_CQ;
  const int c0 = (blockIdx.x * BN_SMALL_CQ + q) * 4;
  const bool cok = c0 < C;
  constexpr int RL = BN_THREADS / BN_SMALL_CQ;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (cok) {
    for (int r = rl; r < n; r += RL) {
      const bf32x4 v = *reinterpret_cast<const bf32x4*>(x + (long long)r * C + c0);
#pragma unroll
      for (int i = 0; i < 4; ++i) { s0[i] += (double)v[i]; s1[i] += (double)v[i] * (double)v[i]; }
    }
  }
  bn_small_reduce(s0, s1, s_part);
  if (!cok) return;
  float sc[4], sh[4];
  const double cnt = n > 0 ? (double)n : 1.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double m = s0[i] / cnt;
    double var = s1[i] / cnt - m * m;
    if (var < 0) var = 0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float gm = gamma ? gamma[c0 + i] : 1.f, bt = beta ? beta[c0 + i] : 0.f;
    sc[i] = bn_scale(is, gm);
    sh[i] = bn_shift(bt, (float)m, is, gm);
    if (rl == 0) {
      save_mean[c0 + i] = (float)m;
      save_invstd[c0 + i] = is;
      if (running_mean) {
        const double unb = n > 1 ? var * cnt / (cnt - 1.0) : var;
        running_mean[c0 + i] = (1.f - momentum) * running_mean[c0 + i] + momentum * (float)m;
        running_var[c0 + i] = (1.f - momentum) * running_var[c0 + i] + momentum * (float)unb;
      }
    }
  }
  for (int r = rl; r < n; r += RL) {
    bf32x4 v = *reinterpret_cast<const bf32x4*>(x + (long long)r * C + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = bn_affine(v[i], sc[i], sh[i]);
      v[i] = relu ? fmaxf(t, 0.f) : t;
    }
    *reinterpret_cast<bf32x4*>(y + (long long)r * y_pitch + c0) = v;
  }
  for (int r = n + rl; r < N; r += RL) *reinterpret_cast<bf32x4*>(y + (long long)r * y_pitch + c0) = bf32x4{0.f, 0.f, 0.f, 0.f};
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_small_backward(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, int relu, int N, int C, long long dy_pitch,
    const int* __restrict__ n_live, float* __restrict__ dx, float* __restrict__ dgamma,
    float* __restrict__ dbeta) {
  __shared__ double s_part[BN_THREADS / 64][BN_SMALL_CQ][8];
  int n = N;
  if (n_live) n = min(N, *n_live);
  const int q = threadIdx.x % BN_SMALL_CQ, rl = threadIdx.x / BN_SMALL_CQ;
  const int c0 = (blockIdx.x * BN_SMALL_CQ + q) * 4;
  const bool cok = c0 < C;
  constexpr int RL = BN_THREADS / BN_SMALL_CQ;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  bf32x4 mu = bf32x4{0, 0, 0, 0}, is = mu;
  float sc[4] = {0, 0, 0, 0}, sh[4] = {0, 0, 0, 0};
  const bool remask = relu && !y;
  if (cok) {
    mu = *reinterpret_cast<const bf32x4*>(mean + c0);
    is = *reinterpret_cast<const bf32x4*>(invstd + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float gm = gamma ? gamma[c0 + i] : 1.f, bt = beta ? beta[c0 + i] : 0.f;
      sc[i] = bn_scale(is[i], gm);
      sh[i] = bn_shift(bt, mu[i], is[i], gm);
    }
    for (int r = rl; r < n; r += RL) {
      const long long o = (long long)r * C + c0;
      const bf32x4 xv = *reinterpret_cast<const bf32x4*>(x + o);
      bf32x4 g = *reinterpret_cast<const bf32x4*>(dy + (long long)r * dy_pitch + c0);
      if (remask) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = bn_affine(xv[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
      } else if (relu) {
        const bf32x4 yv = *reinterpret_cast<const bf32x4*>(y + o);
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = yv[i] > 0.f ? g[i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s0[i] += (double)g[i];
        s1[i] += (double)g[i] * (double)((xv[i] - mu[i]) * is[i]);
      }
    }
  }
  bn_small_reduce(s0, s1, s_part);
  if (!cok) return;
  const double cnt = n > 0 ? (double)n : 1.0;
  float a[4], b[4], cc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = (gamma ? gamma[c0 + i] : 1.f) * is[i];
    b[i] = (float)(s0[i] / cnt);
    cc[i] = (float)(s1[i] / cnt);
    if (rl == 0) {
      if (dgamma) dgamma[c0 + i] = (float)s1[i];
      if (dbeta) dbeta[c0 + i] = (float)s0[i];
    }
  }
  for (int r = rl; r < n; r += RL) {
    const long long o = (long long)r * C + c0;
    const bf32x4 xv = *reinterpret_cast<const bf32x4*>(x + o);
    bf32x4 g = *reinterpret_cast<const bf32x4*>(dy + (long long)r * dy_pitch + c0);
    if (remask) {
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = bn_affine(xv[i], sc[i], sh[i]) > 0.f ? g[i] : 0.f;
    } else if (relu) {
      const bf32x4 yv = *reinterpret_cast<const bf32x4*>(y + o);
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = yv[i] > 0.f ? g[i] : 0.f;
    }
    bf32x4 ov;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (xv[i] - mu[i]) * is[i];
      ov[i] = a[i] * (g[i] - b[i] - xh * cc[i]);
    }
    *reinterpret_cast<bf32x4*>(dx + o) = ov;
  }
  for (int r = n + rl; r < N; r += RL) *reinterpret_cast<bf32x4*>(dx + (long long)r * C + c0) = bf32x4{0.f, 0.f, 0.f, 0.f};
}

// Any channel count on a short matrix (GLENet's reg_std branch normalises (512, 7)): one block per channel, scalar
// loads; statistics in fp64 with a fixed-order tree.  Same arithmetic as the kernels above.
__device__ __forceinline__ void bn_col_reduce(double& s0, double& s1, double (*red)[2]) {
  red[threadIdx.x][0] = s0;
  red[threadIdx.x][1] = s1;
  __syncthreads();
  for (int off = BN_THREADS / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      red[threadIdx.x][0] += red[threadIdx.x + off][0];
      red[threadIdx.x][1] += red[threadIdx.x + off][1];
    }
    __syncthreads();
  }
  s0 = red[0][0];
  s1 = red[0][1];
  __syncthreads();
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_column_forward(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float momentum, int relu, int N, int C, const int* __restrict__ n_live, float* __restrict__ y,
    long long y_pitch, float* __restrict__ save_mean, float* __restrict__ save_invstd,
    float* __restrict__ running_mean, float* __restrict__ running_var) {
  __shared__ double red[BN_THREADS][2];
  int n = N;
  if (n_live) n = min(N, *n_live);
  const int c = blockIdx.x;
  double s0 = 0, s1 = 0;
  for (int r = threadIdx.x; r < n; r += BN_THREADS) {
    const double v = (double)x[(long long)r * C + c];
    s0 += v;
    s1 += v * v;
  }
  bn_col_reduce(s0, s1, red);
  const double cnt = n > 0 ? (double)n : 1.0;
  const double m = s0 / cnt;
  double var = s1 / cnt - m * m;
  if (var < 0) var = 0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  const float sc = bn_scale(is, gm), sh = bn_shift(bt, (float)m, is, gm);
  if (threadIdx.x == 0) {
    save_mean[c] = (float)m;
    save_invstd[c] = is;
    if (running_mean) {
      const double unb = n > 1 ? var * cnt / (cnt - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
  for (int r = threadIdx.x; r < N; r += BN_THREADS) {
    float t = 0.f;
    if (r < n) {
      t = bn_affine(x[(long long)r * C + c], sc, sh);
      if (relu) t = fmaxf(t, 0.f);
    }
    y[(long long)r * y_pitch + c] = t;
  }
}

__global__ __launch_bounds__(BN_THREADS) void k_bn_column_backward(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, int relu, int N, int C, long long dy_pitch,
    const int* __restrict__ n_live, float* __restrict__ dx, float* __restrict__ dgamma,
    float* __restrict__ dbeta) {
  __shared__ double red[BN_THREADS][2];
  int n = N;
  if (n_live) n = min(N, *n_live);
  const int c = blockIdx.x;
  const float mu = mean[c], is = invstd[c];
  const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  const float sc = bn_scale(is, gm), sh = bn_shift(bt, mu, is, gm);
  const bool remask = relu && !y;
  double s0 = 0, s1 = 0;
  for (int r = threadIdx.x; r < n; r += BN_THREADS) {
    const float xv = x[(long long)r * C + c];
    float g = dy[(long long)r * dy_pitch + c];
    if (remask ? !(bn_affine(xv, sc, sh) > 0.f) : (relu && !(y[(long long)r * C + c] > 0.f))) g = 0.f;
    s0 += (double)g;
    s1 += (double)g * (double)((xv - mu) * is);
  }
  bn_col_reduce(s0, s1, red);
  const double cnt = n > 0 ? (double)n : 1.0;
  const float a = gm * is, b = (float)(s0 / cnt), cc = (float)(s1 / cnt);
  if (threadIdx.x == 0) {
    if (dgamma) dgamma[c] = (float)s1;
    if (dbeta) dbeta[c] = (float)s0;
  }
  for (int r = threadIdx.x; r < N; r += BN_THREADS) {
    float o = 0.f;
    if (r < n) {
      const float xv = x[(long long)r * C + c];
      float g = dy[(long long)r * dy_pitch + c];
      if (remask ? !(bn_affine(xv, sc, sh) > 0.f) : (relu && !(y[(long long)r * C + c] > 0.f))) g = 0.f;
      o = a * (g - b - (xv - mu) * is * cc);
    }
    dx[(long long)r * C + c] = o;
  }
}

static bool bn_channels_ok(int C) { return C >= 4 && C <= BN_MAXC && (C & 3) == 0 && BN_THREADS % (C >> 2) == 0; }

// workspace: slab partials (fp64) + 3*C coefficient floats
static size_t bn_coef_offset(int C) { return glx_align((size_t)BN_SLABS * 2 * C * sizeof(double)); }
extern "C" size_t glx_bn_workspace_bytes(int C) { return bn_coef_offset(C) + glx_align((size_t)3 * C * sizeof(float)) + 256; }

extern "C" int glx_bn_relu_train_forward(const float* x, int N, int C, const float* gamma,
                                         const float* beta, float eps, float momentum, int relu,
                                         float* running_mean, float* running_var, float* y,
                                         float* save_mean, float* save_invstd,
                                         const int32_t* n_live, void* workspace,
                                         size_t workspace_bytes, void* state, int y_stride,
                                         void* stream) {
  GLX_REQUIRE(y && save_mean && save_invstd && (N == 0 || x), "glx_bn_relu_train_forward: null pointer");
  if (!bn_channels_ok(C)) {                      // any width on a short matrix: one block per channel
    GLX_REQUIRE(C >= 1 && C <= BN_MAXC && N <= BN_SMALL_N,
                "glx_bn_relu_train_forward: C=%d needs a multiple of 4 dividing 1024 (<= 512), or at most %d rows", C,
                BN_SMALL_N);
    GLX_REQUIRE(y_stride == 0 || y_stride >= C, "glx_bn_relu_train_forward: y_stride %d", y_stride);
    if (N <= 0) return GLX_OK;
    hipLaunchKernelGGL(k_bn_column_forward, dim3(C), dim3(BN_THREADS), 0, (hipStream_t)stream, x, gamma, beta, eps,
                       momentum, relu, N, C, n_live, y, (long long)(y_stride ? y_stride : C), save_mean,
                       save_invstd, running_mean, running_var);
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  GLX_REQUIRE(y_stride == 0 || (y_stride >= C && (y_stride & 3) == 0), "glx_bn_relu_train_forward: y_stride %d", y_stride);
  const long long y_pitch = y_stride ? y_stride : C;
  if (!workspace || workspace_bytes < glx_bn_workspace_bytes(C) - 256) {
    glx_set_error("glx_bn_relu_train_forward: workspace %zu < %zu bytes", workspace_bytes,
                  glx_bn_workspace_bytes(C) - 256);
    return GLX_EWORKSPACE;
  }
  if (N <= 0) return GLX_OK;
  hipStream_t st = (hipStream_t)stream;
  if (N <= BN_SMALL_N) {
    hipLaunchKernelGGL(k_bn_small_forward, dim3(glx_divup(C, 4 * BN_SMALL_CQ)), dim3(BN_THREADS), 0, st, x,
                       gamma, beta, eps, momentum, relu, N, C, n_live, y, y_pitch, save_mean, save_invstd,
                       running_mean, running_var);
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  const int slabs = state ? bn_stats_slabs(N, C) : (glx_divup(N, 8) > BN_SLABS ? BN_SLABS : glx_divup(N, 8));
  const int blocks = bn_apply_blocks(N, C);
  float* coef = (float*)((char*)workspace + bn_coef_offset(C));
  if (state) {
    BnFinalize f{gamma, beta, eps, momentum, coef, save_mean, save_invstd, running_mean, running_var,
                 nullptr, nullptr, nullptr};
    hipLaunchKernelGGL((k_bn_stats<false>), dim3(slabs), dim3(BN_THREADS), 0, st, x, nullptr, nullptr,
                       nullptr, nullptr, 0, N, C, (long long)C, n_live, (BnState*)state, f);
  } else {
    hipLaunchKernelGGL((k_bn_partial<false>), dim3(slabs), dim3(BN_THREADS), 0, st, x, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, 0, N, C, (long long)C, n_live, (double*)workspace);
    hipLaunchKernelGGL(k_bn_finalize_fwd, dim3(1), dim3(BN_FIN_THREADS), 0, st, (const double*)workspace,
                       slabs, gamma, beta, eps, momentum, N, C, n_live, coef, save_mean, save_invstd,
                       running_mean, running_var);
  }
  hipLaunchKernelGGL(k_bn_forward_apply, dim3(blocks < 1 ? 1 : blocks), dim3(BN_THREADS), 0, st, x,
                     (const float*)coef, relu, N, C, n_live, y, y_pitch, (const float*)nullptr);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// The transform alone, for statistics that were taken elsewhere (the sparse conv's epilogue,
// glx_sconv_opts.bn): y = relu?(x * coef[c] + coef[C + c]) on the live rows, zeros on the rest.
extern "C" int glx_bn_apply_forward(const float* x, const float* coef, int relu, int N, int C, const int32_t* n_live,
                                    float* y, int y_stride, void* stream) {
  GLX_REQUIRE(coef && y && (N == 0 || x), "glx_bn_apply_forward: null pointer");
  GLX_REQUIRE(bn_channels_ok(C), "glx_bn_apply_forward: C=%d needs a multiple of 4 dividing 1024 (<= 512)", C);
  GLX_REQUIRE(y_stride == 0 || (y_stride >= C && (y_stride & 3) == 0), "glx_bn_apply_forward: y_stride %d", y_stride);
  if (N <= 0) return GLX_OK;
  const int blocks = bn_apply_blocks(N, C);
  hipLaunchKernelGGL(k_bn_forward_apply, dim3(blocks < 1 ? 1 : blocks), dim3(BN_THREADS), 0, (hipStream_t)stream, x,
                     coef, relu, N, C, n_live, y, (long long)(y_stride ? y_stride : C), (const float*)nullptr);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// y = relu?(x * scale + shift + res): the transform with the identity branch of a residual block added (SparseBasicBlock,
// pcdet/models/backbones_3d/spconv_backbone.py:30-64: out = relu(bn2(conv2(.)) + identity)), one launch instead of three.
extern "C" int glx_bn_apply_add_forward(const float* x, const float* coef, const float* res, int relu, int N, int C,
                                        const int32_t* n_live, float* y, void* stream) {
  GLX_REQUIRE(coef && y && res && (N == 0 || x), "glx_bn_apply_add_forward: null pointer");
  GLX_REQUIRE(bn_channels_ok(C), "glx_bn_apply_add_forward: C=%d needs a multiple of 4 dividing 1024 (<= 512)", C);
  if (N <= 0) return GLX_OK;
  const int blocks = bn_apply_blocks(N, C);
  hipLaunchKernelGGL(k_bn_forward_apply, dim3(blocks < 1 ? 1 : blocks), dim3(BN_THREADS), 0, (hipStream_t)stream, x,
                     coef, relu, N, C, n_live, y, (long long)C, res);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_bn_relu_backward(const float* x, const float* dy, const float* y, int N, int C,
                                    const float* gamma, const float* beta, const float* save_mean,
                                    const float* save_invstd, int relu, float* dx, float* dgamma,
                                    float* dbeta, const int32_t* n_live, void* workspace,
                                    size_t workspace_bytes, void* state, int dy_stride, void* stream) {
  GLX_REQUIRE(save_mean && save_invstd && (N == 0 || (x && dy && dx)), "glx_bn_relu_backward: null pointer");
  if (!bn_channels_ok(C)) {
    GLX_REQUIRE(C >= 1 && C <= BN_MAXC && N <= BN_SMALL_N, "glx_bn_relu_backward: C=%d not supported for %d rows", C, N);
    GLX_REQUIRE(dy_stride == 0 || dy_stride >= C, "glx_bn_relu_backward: dy_stride %d", dy_stride);
    if (N <= 0) return GLX_OK;
    hipLaunchKernelGGL(k_bn_column_backward, dim3(C), dim3(BN_THREADS), 0, (hipStream_t)stream, x, dy, y, gamma, beta,
                       save_mean, save_invstd, relu, N, C, (long long)(dy_stride ? dy_stride : C), n_live, dx,
                       dgamma, dbeta);
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  GLX_REQUIRE(dy_stride == 0 || (dy_stride >= C && (dy_stride & 3) == 0), "glx_bn_relu_backward: dy_stride %d", dy_stride);
  const long long dy_pitch = dy_stride ? dy_stride : C;
  if (!workspace || workspace_bytes < glx_bn_workspace_bytes(C) - 256) {
    glx_set_error("glx_bn_relu_backward: workspace %zu < %zu bytes", workspace_bytes,
                  glx_bn_workspace_bytes(C) - 256);
    return GLX_EWORKSPACE;
  }
  if (N <= 0) return GLX_OK;
  hipStream_t st = (hipStream_t)stream;
  if (N <= BN_SMALL_N) {
    hipLaunchKernelGGL(k_bn_small_backward, dim3(glx_divup(C, 4 * BN_SMALL_CQ)), dim3(BN_THREADS), 0, st, x,
                       dy, y, gamma, beta, save_mean, save_invstd, relu, N, C, dy_pitch, n_live, dx, dgamma, dbeta);
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  const int slabs = state ? bn_stats_slabs(N, C) : (glx_divup(N, 8) > BN_SLABS ? BN_SLABS : glx_divup(N, 8));
  const int blocks = bn_apply_blocks(N, C);
  float* coef = (float*)((char*)workspace + bn_coef_offset(C));
  if (state) {
    BnFinalize f{gamma, beta, 0.f, 0.f, coef, nullptr, nullptr, nullptr, nullptr, save_invstd, dgamma, dbeta};
    hipLaunchKernelGGL((k_bn_stats<true>), dim3(slabs), dim3(BN_THREADS), 0, st, x, dy, y, save_mean,
                       save_invstd, relu, N, C, dy_pitch, n_live, (BnState*)state, f);
  } else {
    hipLaunchKernelGGL((k_bn_partial<true>), dim3(slabs), dim3(BN_THREADS), 0, st, x, dy, y, save_mean,
                       save_invstd, gamma, beta, relu, N, C, dy_pitch, n_live, (double*)workspace);
    hipLaunchKernelGGL(k_bn_finalize_bwd, dim3(1), dim3(BN_FIN_THREADS), 0, st, (const double*)workspace,
                       slabs, gamma, save_invstd, N, C, n_live, coef, dgamma, dbeta);
  }
  hipLaunchKernelGGL(k_bn_backward_apply, dim3(blocks < 1 ? 1 : blocks), dim3(BN_THREADS), 0, st, x,
                     dy, y, (const float*)coef, save_mean, save_invstd, gamma, beta, relu, N, C, dy_pitch, n_live, dx);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// The statistics half of glx_bn_relu_backward alone (one launch): dgamma, dbeta and coef3 = (a, b, c) of
// dx = a (dy [y > 0] - b - c xhat) for a consumer that applies the transform on load (glx_rows_linear_bn_backward).  coef3: 3 C
// floats of the caller's; state: glx_bn_state_bytes(), required.
extern "C" int glx_bn_backward_sums(const float* x, const float* dy, int N, int C, const float* gamma, const float* beta,
                                    const float* save_mean, const float* save_invstd, int relu, float* dgamma, float* dbeta,
                                    const int32_t* n_live, float* coef3, void* state, void* stream) {
  GLX_REQUIRE(save_mean && save_invstd && coef3 && state && (N == 0 || (x && dy)), "glx_bn_backward_sums: null pointer");
  GLX_REQUIRE(bn_channels_ok(C), "glx_bn_backward_sums: C=%d needs a multiple of 4 dividing 1024 (<= 512)", C);
  const int slabs = bn_stats_slabs(N > 0 ? N : 1, C);
  BnFinalize f{gamma, beta, 0.f, 0.f, coef3, nullptr, nullptr, nullptr, nullptr, save_invstd, dgamma, dbeta};
  hipLaunchKernelGGL((k_bn_stats<true>), dim3(slabs), dim3(BN_THREADS), 0, (hipStream_t)stream, x, dy, (const float*)nullptr,
                     save_mean, save_invstd, relu, N > 0 ? N : 0, C, (long long)C, (const int*)n_live, (BnState*)state, f);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_bn_backward_apply(const float* x, const float* dz, const float* coef, const float* mean, const float* invstd,
                                     int N, int C, const int32_t* n_live, float* dx, void* stream) {
  GLX_REQUIRE(coef && mean && invstd && (N == 0 || (x && dz && dx)), "glx_bn_backward_apply: null pointer");
  GLX_REQUIRE(bn_channels_ok(C), "glx_bn_backward_apply: C=%d needs a multiple of 4 dividing 1024 (<= 512)", C);
  if (N <= 0) return GLX_OK;
  const int blocks = bn_apply_blocks(N, C);
  hipLaunchKernelGGL(k_bn_backward_apply, dim3(blocks < 1 ? 1 : blocks), dim3(BN_THREADS), 0, (hipStream_t)stream, x, dz,
                     (const float*)nullptr, coef, mean, invstd, (const float*)nullptr, (const float*)nullptr, 0, N, C,
                     (long long)C, (const int*)n_live, dx);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ================================================================================================ channel-major form
// Training-mode BatchNorm of a tensor in the reference's STACKED convention: batch dimension 1, channels next, the rows
// along the rest -- (1, C, M) and (1, C, M, nsample), what pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:70-130
// feeds its BatchNorm1d / BatchNorm2d layers.  M is the number of active voxels / grid points of the batch, a new value every
// training step, and the vendor library prepares (on first sight: builds) a BatchNorm kernel per problem size: 0.45 s per new
// voxel count measured in a fresh process.  A channel's values are contiguous here (x is (C, L) row-major), so the reductions
// are plain coalesced sums: per (channel, chunk) fp64 partials, every block of the transform re-adds its channel's <= 64
// partials (fixed order: bitwise reproducible), two launches per direction, nothing depends on L but the grid.
#define BNCM_THREADS 256
#define BNCM_CHUNKS 64                   // partials per channel at most

__device__ __forceinline__ double bncm_block_sum(double v, double* s_red) {      // all threads get the block's sum
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// part[(c * chunks + k) * 2 ..]: BWD = false: sum x, sum x^2 of chunk k of channel c; BWD = true: sum dy, sum dy * xhat
template <bool BWD>
__global__ __launch_bounds__(BNCM_THREADS) void k_bncm_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                               long long L, int chunks, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, double* __restrict__ part) {
  __shared__ double s_red[4];
  const int c = blockIdx.y, k = blockIdx.x;
  const long long per = (L + chunks - 1) / chunks, lo = (long long)k * per, hi = min(L, lo + per);
  const float* xr = x + (long long)c * L;
  const float* gr = BWD ? dy + (long long)c * L : nullptr;
  const float mu = BWD ? mean[c] : 0.f, is = BWD ? invstd[c] : 0.f;
  double a = 0.0, b = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += BNCM_THREADS) {
    const float v = xr[i];
    if (BWD) {
      const float g = gr[i];
      a += (double)g;
      b += (double)(g * ((v - mu) * is));
    } else {
      a += (double)v;
      b += (double)v * (double)v;
    }
  }
  a = bncm_block_sum(a, s_red);
  b = bncm_block_sum(b, s_red);
  if (threadIdx.x == 0) {
    part[((long long)c * chunks + k) * 2] = a;
    part[((long long)c * chunks + k) * 2 + 1] = b;
  }
}

__global__ __launch_bounds__(BNCM_THREADS) void k_bncm_forward(const float* __restrict__ x, long long L, int chunks,
                                                               const double* __restrict__ part, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float momentum,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                               float* __restrict__ y) {
  __shared__ float s_coef[2];
  const int c = blockIdx.y;
  if (threadIdx.x == 0) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < chunks; ++k) { s += part[((long long)c * chunks + k) * 2]; q += part[((long long)c * chunks + k) * 2 + 1]; }
    const double mean = s / (double)L;
    double var = q / (double)L - mean * mean;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    s_coef[0] = g * is;
    s_coef[1] = bt - (float)mean * g * is;
    if (blockIdx.x == 0) {
      save_mean[c] = (float)mean;
      save_invstd[c] = is;
      if (running_mean) {        // nn.BatchNorm: running_var takes the UNBIASED batch variance
        const double unb = L > 1 ? var * (double)L / (double)(L - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    }
  }
  __syncthreads();
  const float sc = s_coef[0], sh = s_coef[1];
  const float* xr = x + (long long)c * L;
  float* yr = y + (long long)c * L;
  const long long stride = (long long)gridDim.x * BNCM_THREADS;
  for (long long i = (long long)blockIdx.x * BNCM_THREADS + threadIdx.x; i < L; i += stride) yr[i] = __fmaf_rn(xr[i], sc, sh);
}

__global__ __launch_bounds__(BNCM_THREADS) void k_bncm_backward(const float* __restrict__ x, const float* __restrict__ dy,
                                                                long long L, int chunks, const double* __restrict__ part,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float* __restrict__ dx,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float s_coef[4];
  const int c = blockIdx.y;
  if (threadIdx.x == 0) {
    double s = 0.0, q = 0.0;     // sum dy, sum dy * xhat
    for (int k = 0; k < chunks; ++k) { s += part[((long long)c * chunks + k) * 2]; q += part[((long long)c * chunks + k) * 2 + 1]; }
    const float g = gamma ? gamma[c] : 1.f;
    s_coef[0] = g * invstd[c];                       // a
    s_coef[1] = (float)(s / (double)L);              // b = dbeta / n
    s_coef[2] = (float)(q / (double)L);              // c = dgamma / n
    if (blockIdx.x == 0) {
      if (dgamma) dgamma[c] = (float)q;
      if (dbeta) dbeta[c] = (float)s;
    }
  }
  __syncthreads();
  const float a = s_coef[0], b = s_coef[1], cc = s_coef[2], mu = mean[c], is = invstd[c];
  const float* xr = x + (long long)c * L;
  const float* gr = dy + (long long)c * L;
  float* dr = dx + (long long)c * L;
  const long long stride = (long long)gridDim.x * BNCM_THREADS;
  for (long long i = (long long)blockIdx.x * BNCM_THREADS + threadIdx.x; i < L; i += stride)
    dr[i] = a * (gr[i] - b - cc * ((xr[i] - mu) * is));
}

static int bncm_chunks(long long L) {
  long long k = (L + 8191) / 8192;
  return (int)(k < 1 ? 1 : k > BNCM_CHUNKS ? BNCM_CHUNKS : k);
}
static int bncm_apply_blocks(long long L, int C) {      // ~2048 blocks over all channels, >= 1 per channel
  long long per = (L + 4 * BNCM_THREADS - 1) / (4 * BNCM_THREADS), cap = 2048 / (C > 0 ? C : 1);
  if (cap < 1) cap = 1;
  return (int)(per < 1 ? 1 : per > cap ? cap : per);
}

extern "C" size_t glx_bn_cm_workspace_bytes(int C) { return glx_align((size_t)(C > 0 ? C : 1) * BNCM_CHUNKS * 2 * sizeof(double)); }

extern "C" int glx_bn_cm_train_forward(const float* x, int C, long long L, const float* gamma, const float* beta, float eps,
                                       float momentum, float* running_mean, float* running_var, float* y, float* save_mean,
                                       float* save_invstd, void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(C > 0 && L > 0, "glx_bn_cm_train_forward: empty tensor (C = %d, L = %lld)", C, L);
  GLX_REQUIRE(x && y && save_mean && save_invstd && workspace, "glx_bn_cm_train_forward: null pointer");
  GLX_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "glx_bn_cm_train_forward: running statistics: both or none");
  GLX_REQUIRE(workspace_bytes >= glx_bn_cm_workspace_bytes(C), "glx_bn_cm_train_forward: workspace too small");
  GLX_REQUIRE(C <= 65535, "glx_bn_cm_train_forward: C = %d", C);
  hipStream_t st = (hipStream_t)stream;
  const int chunks = bncm_chunks(L);
  hipLaunchKernelGGL((k_bncm_partial<false>), dim3(chunks, C), dim3(BNCM_THREADS), 0, st, x, (const float*)nullptr, L, chunks,
                     (const float*)nullptr, (const float*)nullptr, (double*)workspace);
  hipLaunchKernelGGL(k_bncm_forward, dim3(bncm_apply_blocks(L, C), C), dim3(BNCM_THREADS), 0, st, x, L, chunks,
                     (const double*)workspace, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, y);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_bn_cm_backward(const float* x, const float* dy, int C, long long L, const float* gamma, const float* save_mean,
                                  const float* save_invstd, float* dx, float* dgamma, float* dbeta, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(C > 0 && L > 0, "glx_bn_cm_backward: empty tensor (C = %d, L = %lld)", C, L);
  GLX_REQUIRE(x && dy && dx && save_mean && save_invstd && workspace, "glx_bn_cm_backward: null pointer");
  GLX_REQUIRE(workspace_bytes >= glx_bn_cm_workspace_bytes(C), "glx_bn_cm_backward: workspace too small");
  GLX_REQUIRE(C <= 65535, "glx_bn_cm_backward: C = %d", C);
  hipStream_t st = (hipStream_t)stream;
  const int chunks = bncm_chunks(L);
  hipLaunchKernelGGL((k_bncm_partial<true>), dim3(chunks, C), dim3(BNCM_THREADS), 0, st, x, dy, L, chunks, save_mean, save_invstd,
                     (double*)workspace);
  hipLaunchKernelGGL(k_bncm_backward, dim3(bncm_apply_blocks(L, C), C), dim3(BNCM_THREADS), 0, st, x, dy, L, chunks,
                     (const double*)workspace, gamma, save_mean, save_invstd, dx, dgamma, dbeta);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
