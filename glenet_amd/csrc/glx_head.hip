// The anchor head's 1x1 convolutions (pcdet/models/dense_heads/anchor_head_single.py:19-37: conv_cls, conv_box,
// conv_dir_cls on the (B, 256, H, W) BEV map) as ONE pass per direction over the channels-last map.  The map is 144 MB at
// the KITTI size and the three heads have 2 + 14 + 4 output channels: the work is reading (forward, weight gradient) or
// writing (input gradient) the map once.  Through the vendor's convolution the training step paid a concatenation of the
// three filters, a zero fill, the convolution, a bias pass, three `.contiguous()` copies forward and the same in reverse
// plus two fills backward (0.3 ms, 20 launches); here: one launch forward writing the three prediction tensors, one for
// the input gradient, two for the filter / bias gradients.
// All three kernels are exact-fp32 v_mfma_f32_16x16x4_f32 GEMMs with M = pixels; the concatenated filters (<= 32 rows,
// zero padded) sit in LDS.
#include "glx_common.h"
#include "glx_bn_state.h"
#include <stdlib.h>

typedef float hf32x4 __attribute__((ext_vector_type(4)));

#define HD_THREADS 256
#define HD_MAXO 32                       // output channels of the three heads together, at most
#define HD_PIX 64                        // pixels per block and pass (4 waves x 16)

struct HeadOut {
  float* p[3];                           // cls / box / dir predictions (M, n[k]) row-major (dir may have n = 0)
  int n[3];
};
struct HeadGrad {
  const float* p[3];
  int n[3];
};

// concatenated filters into LDS, row o = output channel (zero rows above the total), row stride C + 4
__device__ __forceinline__ void hd_load_w(const float* const* W, const int* n, int C, float* s_w) {
  const int ld = C + 4;
  for (int e = threadIdx.x; e < HD_MAXO * (C / 4); e += HD_THREADS) {
    const int o = e / (C / 4), c4 = e - o * (C / 4);
    hf32x4 v = hf32x4{0.f, 0.f, 0.f, 0.f};
    int base = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (o >= base && o < base + n[k]) v = *reinterpret_cast<const hf32x4*>(W[k] + (long long)(o - base) * C + 4 * c4);
      base += n[k];
    }
    *reinterpret_cast<hf32x4*>(s_w + o * ld + 4 * c4) = v;
  }
}

// The map the head reads: one (M, C) matrix, or two column parts x[0] (M, c0) | x[1] (M, C - c0) (c0 a multiple of 16) that
// are transformed on load, x' = max(x * scale + shift, 0) with coef[p] = scale | shift of part p -- the deblocks' raw
// transposed-convolution outputs with their training-mode BatchNorm + ReLU applied here instead of being written as the
// concatenated map first (base_bev_backbone.py:100-104).
struct HeadIn {
  const float* x[2];
  const float* coef[2];
  int c0;
  int quad;      // k_head_wgrad, C = 256: a lane's four column tiles as one 16-byte load (GLX_HEAD_WGRAD_QUAD=0: off)
};

struct HeadW {
  const float* w[3];
  const float* b[3];
  int n[3];
};

// out[m, o] = sum_c x[m, c] W[o, c] + b[o].  A = the pixel tile (lane (i, kq): pixel i, channels 16 t + 4 kq + e), B = W^T.
__global__ __launch_bounds__(HD_THREADS) void k_head_fwd(HeadIn in, long long M, int C, HeadW hw, HeadOut out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w = smem;                                    // HD_MAXO x (C + 4)
  const int ld = C + 4;
  float* s_coef = s_w + HD_MAXO * ld;                   // scale[C] | shift[C] in the concatenated channel order
  hd_load_w(hw.w, hw.n, C, s_w);
  const bool pre = in.coef[0] != nullptr;
  const int c0 = in.c0, c1 = C - in.c0;
  if (pre) {
    for (int c = threadIdx.x; c < C; c += HD_THREADS) {
      const bool a = c < c0;
      s_coef[c] = a ? in.coef[0][c] : in.coef[1][c - c0];
      s_coef[C + c] = a ? in.coef[0][c0 + c] : in.coef[1][c1 + c - c0];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int total = hw.n[0] + hw.n[1] + hw.n[2];
  for (long long m0 = (long long)blockIdx.x * HD_PIX; m0 < M; m0 += (long long)gridDim.x * HD_PIX) {
    const long long m = m0 + wave * 16 + i;
    const long long mc = m < M ? m : M - 1;
    const float* row0 = in.x[0] + mc * c0 + 4 * kq;
    const float* row1 = in.x[1] ? in.x[1] + mc * c1 + 4 * kq - c0 : row0;      // indexed with the concatenated channel
    hf32x4 acc0 = hf32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    for (int t0 = 0; t0 < C / 16; t0 += 8) {           // 8 channel groups of 16 at a time: 8 row loads in flight
      hf32x4 xa[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int ch = 16 * (t0 + u);
        xa[u] = t0 + u < C / 16 ? *reinterpret_cast<const hf32x4*>((ch < c0 ? row0 : row1) + ch) : hf32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (pre) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (t0 + u < C / 16) {
            const hf32x4 sc = *reinterpret_cast<const hf32x4*>(s_coef + 16 * (t0 + u) + 4 * kq);
            const hf32x4 sh = *reinterpret_cast<const hf32x4*>(s_coef + C + 16 * (t0 + u) + 4 * kq);
#pragma unroll
            for (int e = 0; e < 4; ++e) xa[u][e] = fmaxf(__fmaf_rn(xa[u][e], sc[e], sh[e]), 0.f);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (t0 + u < C / 16) {
          const hf32x4 w0 = *reinterpret_cast<const hf32x4*>(s_w + i * ld + 16 * (t0 + u) + 4 * kq);
          const hf32x4 w1 = *reinterpret_cast<const hf32x4*>(s_w + (16 + i) * ld + 16 * (t0 + u) + 4 * kq);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][e], w0[e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][e], w1[e], acc1, 0, 0, 0);
          }
        }
      }
    }
    // D: rows = pixels 4 kq + e of the wave's 16, columns = output channels i (acc0) and 16 + i (acc1)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int o = 16 * half + i;
      if (o >= total) continue;
      int k = 0, oo = o;
      if (oo >= hw.n[0]) { oo -= hw.n[0]; k = 1; if (oo >= hw.n[1]) { oo -= hw.n[1]; k = 2; } }
      const float bias = hw.b[k] ? hw.b[k][oo] : 0.f;
      float* dst = out.p[k];
      const int nk = out.n[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long long mm = m0 + wave * 16 + 4 * kq + e;
        if (mm < M) dst[mm * nk + oo] = (half ? acc1[e] : acc0[e]) + bias;
      }
    }
  }
}

// gx[m, c] = sum_o g[m, o] W[o, c]:  A = the gradient tile (pixels x 32 output channels), B = W (32 x C), a wave owns a
// quarter of the C columns of the block's 64 pixels... no: a wave owns 16 pixels and walks all column tiles.
__global__ __launch_bounds__(HD_THREADS) void k_head_dgrad(HeadGrad g, long long M, int C, HeadW hw, float* __restrict__ gx) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w = smem;                                    // HD_MAXO x (C + 4)
  const int ld = C + 4;
  hd_load_w(hw.w, hw.n, C, s_w);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  for (long long m0 = (long long)blockIdx.x * HD_PIX; m0 < M; m0 += (long long)gridDim.x * HD_PIX) {
    const long long m = m0 + wave * 16 + i;
    // A operand: lane (i, kq) holds g[pixel i][o = 4 s + kq] for the 8 steps s
    float ga[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int o = 4 * s + kq;
      float v = 0.f;
      if (m < M) {
        if (o < g.n[0]) v = g.p[0][m * g.n[0] + o];
        else if (o < g.n[0] + g.n[1]) v = g.p[1][m * g.n[1] + (o - g.n[0])];
        else if (o < g.n[0] + g.n[1] + g.n[2]) v = g.p[2][m * g.n[2] + (o - g.n[0] - g.n[1])];
      }
      ga[s] = v;
    }
    for (int t = 0; t < C / 16; ++t) {
      hf32x4 acc = hf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float wv = s_w[(4 * s + kq) * ld + 16 * t + i];      // B[kq][j = i]: W[o = 4 s + kq][c = 16 t + i]
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], wv, acc, 0, 0, 0);
      }
      // D rows = pixels 4 kq + e, column = channel 16 t + i
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long long mm = m0 + wave * 16 + 4 * kq + e;
        if (mm < M) gx[mm * C + 16 * t + i] = acc[e];
      }
    }
  }
}

// The input gradient for a head that read its map through two BatchNorm + ReLU transforms (HeadIn with coefficients): the
// gradient of the transformed map is masked with the ReLU (re-derived from the raw value, like every BatchNorm backward here),
// written per part as dz, and its two BatchNorm-backward sums per channel -- sum dz, sum dz * xhat -- are taken on the way
// (block sums -> the stream's BnState accumulators -> the last block writes gamma * invstd | mean dz | mean dz xhat for
// glx_bn_backward_apply, and dgamma / dbeta): the two statistics passes over the 144 MB map (31 + 43 us) are gone.  C = 256.
struct HeadBnBwd {
  const float* y[2];       // raw parts (M, c_p)
  const float* coef[2];    // forward scale | shift
  const float* mean[2];
  const float* invstd[2];
  const float* gamma[2];
  float* dz[2];            // (M, c_p) masked gradients
  float* coef3[2];         // 3 * c_p
  float* dgamma[2];
  float* dbeta[2];
  BnState* state;
  int c0;
};

__global__ __launch_bounds__(HD_THREADS) void k_head_dgrad_bn(HeadGrad g, long long M, HeadW hw, HeadBnBwd bb) {
  constexpr int C = 256, NT = C / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w = smem;                                    // HD_MAXO x (C + 4)
  const int ld = C + 4;
  float* s_par = s_w + HD_MAXO * ld;                    // scale | shift | mean | invstd, C each (concatenated channel order)
  float* s_sum = s_par + 4 * C;                         // 4 waves x C x 2
  __shared__ int s_last;
  hd_load_w(hw.w, hw.n, C, s_w);
  const int c0 = bb.c0, c1 = C - bb.c0;
  for (int c = threadIdx.x; c < C; c += HD_THREADS) {
    const int p = c < c0 ? 0 : 1, cc = c - (p ? c0 : 0), cp = p ? c1 : c0;
    s_par[c] = bb.coef[p][cc];
    s_par[C + c] = bb.coef[p][cp + cc];
    s_par[2 * C + c] = bb.mean[p][cc];
    s_par[3 * C + c] = bb.invstd[p][cc];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  for (int e = threadIdx.x; e < 4 * C * 2; e += HD_THREADS) s_sum[e] = 0.f;     // per wave and channel: owned by one lane
  __syncthreads();
  for (long long m0 = (long long)blockIdx.x * HD_PIX; m0 < M; m0 += (long long)gridDim.x * HD_PIX) {
    const long long m = m0 + wave * 16 + i;
    float ga[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int o = 4 * s + kq;
      float v = 0.f;
      if (m < M) {
        if (o < g.n[0]) v = g.p[0][m * g.n[0] + o];
        else if (o < g.n[0] + g.n[1]) v = g.p[1][m * g.n[1] + (o - g.n[0])];
        else if (o < g.n[0] + g.n[1] + g.n[2]) v = g.p[2][m * g.n[2] + (o - g.n[0] - g.n[1])];
      }
      ga[s] = v;
    }
#pragma unroll 2
    for (int t = 0; t < NT; ++t) {
      // operands swapped against k_head_dgrad: D rows = channels 16 t + 4 kq + e, column = pixel i -- a lane holds FOUR
      // CONSECUTIVE CHANNELS of one pixel, so the raw map is read and dz written with 16-byte accesses
      const int c = 16 * t + 4 * kq;
      const int p = 16 * t < c0 ? 0 : 1, cc = c - (p ? c0 : 0), cp = p ? c1 : c0;
      const long long mc = m < M ? m : M - 1;
      const hf32x4 yv = *reinterpret_cast<const hf32x4*>(bb.y[p] + mc * cp + cc);
      hf32x4 acc = hf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float wv = s_w[(4 * s + kq) * ld + 16 * t + i];      // A[m = channel 16 t + i][k = o = 4 s + kq]
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, ga[s], acc, 0, 0, 0);
      }
      const hf32x4 sc = *reinterpret_cast<const hf32x4*>(s_par + c), sh = *reinterpret_cast<const hf32x4*>(s_par + C + c);
      const hf32x4 mu = *reinterpret_cast<const hf32x4*>(s_par + 2 * C + c), is = *reinterpret_cast<const hf32x4*>(s_par + 3 * C + c);
      hf32x4 gv, gx2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        gv[e] = (m < M && bn_affine(yv[e], sc[e], sh[e]) > 0.f) ? acc[e] : 0.f;
        gx2[e] = gv[e] * ((yv[e] - mu[e]) * is[e]);
      }
      if (m < M) *reinterpret_cast<hf32x4*>(bb.dz[p] + m * cp + cc) = gv;
      // sums over the wave's 16 pixels (the lanes i of a kq group)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int x = 1; x < 16; x <<= 1) {
          gv[e] += __shfl_xor(gv[e], x, 64);
          gx2[e] += __shfl_xor(gx2[e], x, 64);
        }
      }
      if (i == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s_sum[(wave * C + c + e) * 2] += gv[e];
          s_sum[(wave * C + c + e) * 2 + 1] += gx2[e];
        }
      }
    }
  }
  __syncthreads();
  double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
  if (threadIdx.x < C / 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a0[j] += (double)s_sum[(w * C + 4 * threadIdx.x + j) * 2];
        a1[j] += (double)s_sum[(w * C + 4 * threadIdx.x + j) * 2 + 1];
      }
  }
  if (!bn_contribute(bb.state, C, a0, a1, gridDim.x, &s_last)) return;
  // the last block: the sets' totals (exchanged for zero), the coefficients of the transform, dgamma / dbeta
  const double cnt = (double)M;
  for (int c = threadIdx.x; c < C; c += HD_THREADS) {
    double s = 0, ss = 0;
    for (int k = 0; k < BN_SETS; ++k) { s += bn_take(&bb.state->acc[k][c]); ss += bn_take(&bb.state->acc[k][BN_MAXC + c]); }
    const int p = c < c0 ? 0 : 1, cc = c - (p ? c0 : 0), cp = p ? c1 : c0;
    bb.coef3[p][cc] = (bb.gamma[p] ? bb.gamma[p][cc] : 1.f) * bb.invstd[p][cc];
    bb.coef3[p][cp + cc] = (float)(s / cnt);
    bb.coef3[p][2 * cp + cc] = (float)(ss / cnt);
    bb.dgamma[p][cc] = (float)ss;
    bb.dbeta[p][cc] = (float)s;
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&bb.state->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Second form of k_head_dgrad_bn (c0 a multiple of 64): a WAVE owns 64 channels (four column tiles) of the block's 64
// pixels instead of 16 pixels x all channels.  What that buys: the filter operands of its tiles live in registers (no LDS
// read per MFMA), the BatchNorm-backward sums stay in per-lane registers over all passes and are reduced across lanes
// ONCE at the end (the first form spent 32 cross-lane steps per 16-channel group and pass, as many issue cycles as its
// MFMAs), and eight 16-byte loads of the raw map per lane are in flight before the first product is needed (the first
// form had two).  The gradient rows of the 64 pixels (80 bytes each) are read by every wave (L2 hits).
__global__ __launch_bounds__(HD_THREADS) void k_head_dgrad_bn2(HeadGrad g, long long M, HeadW hw, HeadBnBwd bb) {
  constexpr int C = 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w = smem;                                    // HD_MAXO x (C + 4)
  const int ld = C + 4;
  float* s_par = s_w + HD_MAXO * ld;                    // scale | shift | mean | invstd, C each (concatenated channel order)
  float* s_sum = s_par + 4 * C;                         // C x 2
  float* s_g = s_sum + 2 * C;                           // HD_PIX x 33: the pass's gradient rows (zero beyond the heads' total)
  __shared__ int s_last;
  hd_load_w(hw.w, hw.n, C, s_w);
  const int c0 = bb.c0, c1 = C - bb.c0;
  for (int c = threadIdx.x; c < C; c += HD_THREADS) {
    const int p = c < c0 ? 0 : 1, cc = c - (p ? c0 : 0), cp = p ? c1 : c0;
    s_par[c] = bb.coef[p][cc];
    s_par[C + c] = bb.coef[p][cp + cc];
    s_par[2 * C + c] = bb.mean[p][cc];
    s_par[3 * C + c] = bb.invstd[p][cc];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int cw = 64 * wave;                             // the wave's first channel (concatenated order)
  const int p = cw < c0 ? 0 : 1, cp = p ? c1 : c0;      // its part, wave-uniform (c0 % 64 == 0)
  const int cpart = cw - (p ? c0 : 0);                  // first channel inside the part
  const float* __restrict__ yp = bb.y[p];
  float* __restrict__ dzp = bb.dz[p];
  float sg[4][4], sx[4][4];                             // per lane: sum dz, sum dz * xhat of channel cw + 16 tl + 4 kq + e
#pragma unroll
  for (int tl = 0; tl < 4; ++tl)
#pragma unroll
    for (int e = 0; e < 4; ++e) { sg[tl][e] = 0.f; sx[tl][e] = 0.f; }
  const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
  for (long long m0 = (long long)blockIdx.x * HD_PIX; m0 < M; m0 += (long long)gridDim.x * HD_PIX) {
    {   // the pass's gradient rows into LDS: thread = (pixel, eight output channels)
      const int px = threadIdx.x >> 2, o0 = (threadIdx.x & 3) * 8;
      const long long m = m0 + px;
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int o = o0 + u;
        v[u] = 0.f;
        if (m < M) {
          if (o < n0) v[u] = g.p[0][m * n0 + o];
          else if (o < n0 + n1) v[u] = g.p[1][m * n1 + (o - n0)];
          else if (o < n0 + n1 + n2) v[u] = g.p[2][m * n2 + (o - n0 - n1)];
        }
      }
      __syncthreads();          // the previous pass has read its rows
#pragma unroll
      for (int u = 0; u < 8; ++u) s_g[px * 33 + o0 + u] = v[u];
      __syncthreads();
    }
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {          // two pixel groups of 16 at a time: 8 loads of the raw map in flight per lane
      hf32x4 yv[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const long long m = m0 + 16 * (2 * h + q) + i;
        const long long mc = m < M ? m : M - 1;
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) yv[q][tl] = *reinterpret_cast<const hf32x4*>(yp + mc * cp + cpart + 16 * tl + 4 * kq);
      }
      float ga[2][8];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int s = 0; s < 8; ++s) ga[q][s] = s_g[(16 * (2 * h + q) + i) * 33 + 4 * s + kq];
      int lo = 0;
      asm volatile("" : "+v"(lo));         // an opaque 0: the LDS reads below stay inside the loop (hoisted, the per-channel
                                           // coefficients and filter columns of all four tiles were 96 live registers)
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) {
        const int c = cw + 16 * tl + 4 * kq + lo;
        const hf32x4 sc = *reinterpret_cast<const hf32x4*>(s_par + c), sh = *reinterpret_cast<const hf32x4*>(s_par + C + c);
        const hf32x4 mu = *reinterpret_cast<const hf32x4*>(s_par + 2 * C + c), is = *reinterpret_cast<const hf32x4*>(s_par + 3 * C + c);
        float wv[8];                                      // A[m = channel cw + 16 tl + i][k = o = 4 s + kq]
#pragma unroll
        for (int s = 0; s < 8; ++s) wv[s] = s_w[(4 * s + kq) * ld + cw + 16 * tl + i + lo];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const long long m = m0 + 16 * (2 * h + q) + i;
          hf32x4 acc = hf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s], ga[q][s], acc, 0, 0, 0);
          hf32x4 gv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = yv[q][tl][e];
            gv[e] = (m < M && bn_affine(y, sc[e], sh[e]) > 0.f) ? acc[e] : 0.f;
            sg[tl][e] += gv[e];
            sx[tl][e] += gv[e] * ((y - mu[e]) * is[e]);
          }
          if (m < M) *reinterpret_cast<hf32x4*>(dzp + m * cp + cpart + 16 * tl + 4 * kq) = gv;
        }
      }
    }
  }
  // sums over the lanes i of a kq group, once
#pragma unroll
  for (int tl = 0; tl < 4; ++tl)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int x = 1; x < 16; x <<= 1) {
        sg[tl][e] += __shfl_xor(sg[tl][e], x, 64);
        sx[tl][e] += __shfl_xor(sx[tl][e], x, 64);
      }
      if (i == 0) {
        s_sum[(cw + 16 * tl + 4 * kq + e) * 2] = sg[tl][e];
        s_sum[(cw + 16 * tl + 4 * kq + e) * 2 + 1] = sx[tl][e];
      }
    }
  __syncthreads();
  double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
  if (threadIdx.x < C / 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0[j] = (double)s_sum[(4 * threadIdx.x + j) * 2];
      a1[j] = (double)s_sum[(4 * threadIdx.x + j) * 2 + 1];
    }
  }
  if (!bn_contribute(bb.state, C, a0, a1, gridDim.x, &s_last)) return;
  const double cnt = (double)M;
  for (int c = threadIdx.x; c < C; c += HD_THREADS) {
    double s = 0, ss = 0;
    for (int k = 0; k < BN_SETS; ++k) { s += bn_take(&bb.state->acc[k][c]); ss += bn_take(&bb.state->acc[k][BN_MAXC + c]); }
    const int pp = c < c0 ? 0 : 1, cc = c - (pp ? c0 : 0), cq = pp ? c1 : c0;
    bb.coef3[pp][cc] = (bb.gamma[pp] ? bb.gamma[pp][cc] : 1.f) * bb.invstd[pp][cc];
    bb.coef3[pp][cq + cc] = (float)(s / cnt);
    bb.coef3[pp][2 * cq + cc] = (float)(ss / cnt);
    bb.dgamma[pp][cc] = (float)ss;
    bb.dbeta[pp][cc] = (float)s;
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&bb.state->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// gW[o, c] = sum_m g[m, o] x[m, c], gb[o] = sum_m g[m, o]: contraction over pixels.  A = g^T (lane (i, kq): output channel
// i (+16), pixel 4 s + kq), B = x (lane (j, kq): pixel 4 s + kq, channel 16 t + j).  A wave owns C / 64 column tiles... a block
// of 4 waves covers C = 256 with 4 column tiles per wave; partial sums per block go to the workspace, k_head_wreduce adds
// them in block order (fixed summation order).
#define HD_WBLOCKS 512
__global__ __launch_bounds__(HD_THREADS) void k_head_wgrad(HeadGrad g, HeadIn in, long long M, int C, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int tpw = C / 64;                               // column tiles per wave (C = 256: 4)
  const bool pre = in.coef[0] != nullptr;
  const int c0 = in.c0, c1 = C - in.c0;
  float psc[8], psh[8];                                 // the input transform of this lane's channels 16 (wave tpw + u) + i
  // C = 256 (four column tiles per wave): tile u of lane i is channel 64 wave + 4 i + u, so that a lane's four tiles are ONE
  // 16-byte load of the pixel's row (16 lanes = 256 contiguous bytes); other widths: channel 16 (wave tpw + u) + i
  const bool quad = tpw == 4 && in.quad;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = quad ? 64 * wave + 4 * i + u : 16 * (wave * tpw + u) + i;
    psc[u] = 1.f; psh[u] = 0.f;
    if (pre && u < tpw) {
      psc[u] = c < c0 ? in.coef[0][c] : in.coef[1][c - c0];
      psh[u] = c < c0 ? in.coef[0][c0 + c] : in.coef[1][c1 + c - c0];
    }
  }
  hf32x4 acc[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[h][u] = hf32x4{0.f, 0.f, 0.f, 0.f};
  float gsum0 = 0.f, gsum1 = 0.f;
  const long long per = (M + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < M ? lo + per : M;
  auto gval = [&](long long m, int o) {
    if (m >= hi) return 0.f;
    if (o < g.n[0]) return g.p[0][m * g.n[0] + o];
    if (o < g.n[0] + g.n[1]) return g.p[1][m * g.n[1] + (o - g.n[0])];
    if (o < g.n[0] + g.n[1] + g.n[2]) return g.p[2][m * g.n[2] + (o - g.n[0] - g.n[1])];
    return 0.f;
  };
  for (long long m0 = lo; m0 < hi; m0 += 16) {          // 4 steps of 4 pixels
    float a0[4], a1[4], xb[4][8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const long long m = m0 + 4 * s + kq;
      a0[s] = gval(m, i);
      a1[s] = gval(m, 16 + i);
      const long long mc = m < hi ? m : hi - 1;
      if (quad) {
        const int ch = 64 * wave + 4 * i;
        const float* src = ch < c0 ? in.x[0] + mc * c0 + ch : in.x[1] + mc * c1 + ch - c0;
        const hf32x4 v4 = *reinterpret_cast<const hf32x4*>(src);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          float v = (u < 4 && m < hi) ? v4[u & 3] : 0.f;
          if (pre && u < 4 && m < hi) v = fmaxf(__fmaf_rn(v, psc[u], psh[u]), 0.f);
          xb[s][u] = v;
        }
      } else {
        const float* row0 = in.x[0] + mc * c0 + i;
        const float* row1 = in.x[1] ? in.x[1] + mc * c1 + i - c0 : row0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ch = 16 * (wave * tpw + u);
          float v = (u < tpw && m < hi) ? (ch < c0 ? row0 : row1)[ch] : 0.f;
          if (pre && u < tpw && m < hi) v = fmaxf(__fmaf_rn(v, psc[u], psh[u]), 0.f);
          xb[s][u] = v;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      gsum0 += a0[s];
      gsum1 += a1[s];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u < tpw) {
          acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], xb[s][u], acc[0][u], 0, 0, 0);
          acc[1][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], xb[s][u], acc[1][u], 0, 0, 0);
        }
      }
    }
  }
  // D rows = output channels 4 kq + e (+16), columns = channels 16 (wave tpw + u) + i
  float* dst = part + (long long)blockIdx.x * (HD_MAXO * (C + 1));
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (u < tpw)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          dst[(16 * h + 4 * kq + e) * (C + 1) + (quad ? 64 * wave + 4 * i + u : 16 * (wave * tpw + u) + i)] = acc[h][u][e];
  // bias gradient: lanes (i, kq) hold the sums over the pixels 4 s + kq of their kq
  gsum0 += __shfl_xor(gsum0, 16, 64); gsum0 += __shfl_xor(gsum0, 32, 64);
  gsum1 += __shfl_xor(gsum1, 16, 64); gsum1 += __shfl_xor(gsum1, 32, 64);
  if (wave == 0 && kq == 0) {
    dst[i * (C + 1) + C] = gsum0;
    dst[(16 + i) * (C + 1) + C] = gsum1;
  }
}

struct HeadWOut {
  float* w[3];
  float* b[3];
  int n[3];
};
// a wave per element: lane l adds the partials of blocks l, l + 64, ... (in that order), then a butterfly -- a fixed
// summation order, 64 loads in flight per element instead of one thread walking 512 slabs
__global__ void k_head_wreduce(const float* __restrict__ part, int nblocks, int C, HeadWOut out) {
  const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int total = out.n[0] + out.n[1] + out.n[2];
  if (e >= total * (C + 1)) return;
  const int o = e / (C + 1), c = e - o * (C + 1);
  float s = 0.f;
  for (int b = lane; b < nblocks; b += 64) s += part[(long long)b * (HD_MAXO * (C + 1)) + e];
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
  if (lane) return;
  int k = 0, oo = o;
  if (oo >= out.n[0]) { oo -= out.n[0]; k = 1; if (oo >= out.n[1]) { oo -= out.n[1]; k = 2; } }
  if (c < C) { if (out.w[k]) out.w[k][(long long)oo * C + c] = s; }
  else if (out.b[k]) out.b[k][oo] = s;
}

static int head_check(long long M, int C, const int32_t* n, const char* who) {
  GLX_REQUIRE(M >= 1 && C >= 64 && C % 64 == 0 && C <= 512, "%s: needs M >= 1 and C a multiple of 64 up to 512 (got %lld, %d)", who, M, C);
  GLX_REQUIRE(n[0] >= 1 && n[1] >= 0 && n[2] >= 0 && n[0] + n[1] + n[2] <= HD_MAXO, "%s: %d + %d + %d output channels (1..%d)",
              who, n[0], n[1], n[2], HD_MAXO);
  return GLX_OK;
}

static int head_lds_attr(const void* kern, size_t lds) {
  GLX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return GLX_OK;
}

static int head_in(const float* x0, const float* x1, int c0, const float* coef0, const float* coef1, int C, HeadIn* in,
                   const char* who) {
  GLX_REQUIRE(x0, "%s: null map", who);
  if (!x1) {
    GLX_REQUIRE(!coef0 && !coef1, "%s: the input transform needs the two-part form", who);
    *in = HeadIn{{x0, nullptr}, {nullptr, nullptr}, C, 0};
    return GLX_OK;
  }
  GLX_REQUIRE(c0 > 0 && c0 < C && c0 % 16 == 0, "%s: the first part has %d of %d channels (a multiple of 16 inside)", who, c0, C);
  GLX_REQUIRE((coef0 == nullptr) == (coef1 == nullptr), "%s: both parts or neither are transformed", who);
  static const int quad = 1;
  *in = HeadIn{{x0, x1}, {coef0, coef1}, c0, quad};
  return GLX_OK;
}

extern "C" int glx_head1x1_forward(const float* x, int64_t M, int C, const float* const* W, const float* const* bias,
                                   const int32_t* n, float* const* out, void* stream) {
  return glx_head1x1_forward_parts(x, nullptr, C, nullptr, nullptr, M, C, W, bias, n, out, stream);
}

extern "C" int glx_head1x1_forward_parts(const float* x0, const float* x1, int c0, const float* coef0, const float* coef1,
                                         int64_t M, int C, const float* const* W, const float* const* bias, const int32_t* n,
                                         float* const* out, void* stream) {
  GLX_REQUIRE(x0 && W && n && out, "glx_head1x1_forward: null pointer");
  int rc = head_check(M, C, n, "glx_head1x1_forward");
  if (rc != GLX_OK) return rc;
  HeadIn in;
  rc = head_in(x0, x1, c0, coef0, coef1, C, &in, "glx_head1x1_forward");
  if (rc != GLX_OK) return rc;
  HeadW hw; HeadOut ho;
  for (int k = 0; k < 3; ++k) {
    hw.w[k] = W[k]; hw.b[k] = bias ? bias[k] : nullptr; hw.n[k] = n[k];
    ho.p[k] = out[k]; ho.n[k] = n[k];
    GLX_REQUIRE(n[k] == 0 || (W[k] && out[k]), "glx_head1x1_forward: head %d has no weights / output", k);
  }
  const size_t lds = (size_t)HD_MAXO * (C + 4) * 4 + (size_t)2 * C * 4;
  rc = head_lds_attr((const void*)k_head_fwd, lds);
  if (rc != GLX_OK) return rc;
  long long blocks = (M + HD_PIX - 1) / HD_PIX;
  static const int fwd_blocks = 1024;    // the resident count (four 35 KB blocks per CU): 44.6 us; 512: 50.2, 768: 46.4, 1536: 51.5, 2048: 49.1
  if (blocks > fwd_blocks) blocks = fwd_blocks;
  hipLaunchKernelGGL(k_head_fwd, dim3((unsigned)blocks), dim3(HD_THREADS), lds, (hipStream_t)stream, in, (long long)M, C, hw, ho);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_head1x1_input_grad(const float* const* grad, int64_t M, int C, const float* const* W, const int32_t* n,
                                      float* gx, void* stream) {
  GLX_REQUIRE(grad && W && n && gx, "glx_head1x1_input_grad: null pointer");
  int rc = head_check(M, C, n, "glx_head1x1_input_grad");
  if (rc != GLX_OK) return rc;
  HeadW hw; HeadGrad hg;
  for (int k = 0; k < 3; ++k) {
    hw.w[k] = W[k]; hw.b[k] = nullptr; hw.n[k] = n[k];
    hg.p[k] = grad[k]; hg.n[k] = n[k];
    GLX_REQUIRE(n[k] == 0 || (W[k] && grad[k]), "glx_head1x1_input_grad: head %d has no weights / gradient", k);
  }
  const size_t lds = (size_t)HD_MAXO * (C + 4) * 4;
  rc = head_lds_attr((const void*)k_head_dgrad, lds);
  if (rc != GLX_OK) return rc;
  long long blocks = (M + HD_PIX - 1) / HD_PIX;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_head_dgrad, dim3((unsigned)blocks), dim3(HD_THREADS), lds, (hipStream_t)stream, hg, (long long)M, C, hw, gx);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_head1x1_input_grad_bn(const float* const* grad, int64_t M, int C, const float* const* W, const int32_t* n,
                                         const float* y0, const float* y1, int c0, const glx_bn_bwd_stats* bn0,
                                         const glx_bn_bwd_stats* bn1, float* dz0, float* dz1, void* stream) {
  return glx_head1x1_input_grad_bn_form(grad, M, C, W, n, y0, y1, c0, bn0, bn1, dz0, dz1, 0, stream);
}

extern "C" int glx_head1x1_input_grad_bn_form(const float* const* grad, int64_t M, int C, const float* const* W, const int32_t* n,
                                              const float* y0, const float* y1, int c0, const glx_bn_bwd_stats* bn0,
                                              const glx_bn_bwd_stats* bn1, float* dz0, float* dz1, int form, void* stream) {
  GLX_REQUIRE(grad && W && n && y0 && y1 && bn0 && bn1 && dz0 && dz1, "glx_head1x1_input_grad_bn: null pointer");
  GLX_REQUIRE(form == 0 || form == 1, "glx_head1x1_input_grad_bn_form: form %d (0: a wave owns 16 pixels, 1: 64 channels)", form);
  GLX_REQUIRE(C == 256 && c0 > 0 && c0 < C && c0 % 16 == 0, "glx_head1x1_input_grad_bn: C = %d (256), first part %d", C, c0);
  int rc = head_check(M, C, n, "glx_head1x1_input_grad_bn");
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(M < 2147483647LL / 4, "glx_head1x1_input_grad_bn: %lld pixels", (long long)M);
  const glx_bn_bwd_stats* b[2] = {bn0, bn1};
  HeadBnBwd bb;
  bb.y[0] = y0; bb.y[1] = y1; bb.dz[0] = dz0; bb.dz[1] = dz1; bb.c0 = c0;
  for (int p = 0; p < 2; ++p) {
    GLX_REQUIRE(b[p]->state && b[p]->coef_fwd && b[p]->mean && b[p]->invstd && b[p]->coef && b[p]->dgamma && b[p]->dbeta,
                "glx_head1x1_input_grad_bn: part %d: null pointer", p);
    GLX_REQUIRE(b[p]->state == bn0->state, "glx_head1x1_input_grad_bn: the parts share the stream's accumulator");
    bb.coef[p] = b[p]->coef_fwd; bb.mean[p] = b[p]->mean; bb.invstd[p] = b[p]->invstd; bb.gamma[p] = b[p]->gamma;
    bb.coef3[p] = b[p]->coef; bb.dgamma[p] = b[p]->dgamma; bb.dbeta[p] = b[p]->dbeta;
  }
  bb.state = (BnState*)bn0->state;
  HeadW hw; HeadGrad hg;
  for (int k = 0; k < 3; ++k) {
    hw.w[k] = W[k]; hw.b[k] = nullptr; hw.n[k] = n[k];
    hg.p[k] = grad[k]; hg.n[k] = n[k];
    GLX_REQUIRE(n[k] == 0 || (W[k] && grad[k]), "glx_head1x1_input_grad_bn: head %d has no weights / gradient", k);
  }
  long long blocks = (M + HD_PIX - 1) / HD_PIX;
  // form 1 measured (round 4): alone 74 us against form 0's 109; inside the recorded step the head's backward stage takes
  // 0.31 ms either way (the weight gradient reads the same 144 MB beside it on the weight-gradient stream, the RoI branch's
  // proposal kernels on a third) and the step is 6.085 against 6.074 ms over six alternating pairs: the caller's choice
  const int v2 = form;
  static const int v2_blocks = 512;   // two resident blocks per CU: 74 us; 768: 97, 1024: 87, 2048: 108 (per-block filter image + atomics)
  if (v2 && c0 % 64 == 0) {       // a wave owns 64 channels: weights and sums in registers, 16 loads in flight
    const size_t lds2 = (size_t)HD_MAXO * (C + 4) * 4 + (size_t)4 * C * 4 + (size_t)C * 2 * 4 + (size_t)HD_PIX * 33 * 4;
    rc = head_lds_attr((const void*)k_head_dgrad_bn2, lds2);
    if (rc != GLX_OK) return rc;
    if (blocks > v2_blocks) blocks = v2_blocks;
    hipLaunchKernelGGL(k_head_dgrad_bn2, dim3((unsigned)blocks), dim3(HD_THREADS), lds2, (hipStream_t)stream, hg, (long long)M, hw, bb);
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  const size_t lds = (size_t)HD_MAXO * (C + 4) * 4 + (size_t)4 * C * 4 + (size_t)4 * C * 2 * 4;
  rc = head_lds_attr((const void*)k_head_dgrad_bn, lds);
  if (rc != GLX_OK) return rc;
  static const int v1_blocks = 2048;
  if (blocks > v1_blocks) blocks = v1_blocks;
  hipLaunchKernelGGL(k_head_dgrad_bn, dim3((unsigned)blocks), dim3(HD_THREADS), lds, (hipStream_t)stream, hg, (long long)M, hw, bb);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" size_t glx_head1x1_wgrad_workspace_bytes(int C) { return (size_t)HD_WBLOCKS * HD_MAXO * (C + 1) * sizeof(float); }

extern "C" int glx_head1x1_weight_grad(const float* const* grad, const float* x, int64_t M, int C, const int32_t* n,
                                       float* const* gW, float* const* gb, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  return glx_head1x1_weight_grad_parts(grad, x, nullptr, C, nullptr, nullptr, M, C, n, gW, gb, workspace, workspace_bytes, stream);
}

extern "C" int glx_head1x1_weight_grad_parts(const float* const* grad, const float* x0, const float* x1, int c0,
                                             const float* coef0, const float* coef1, int64_t M, int C, const int32_t* n,
                                             float* const* gW, float* const* gb, void* workspace, size_t workspace_bytes,
                                             void* stream) {
  GLX_REQUIRE(grad && x0 && n && gW && workspace, "glx_head1x1_weight_grad: null pointer");
  int rc = head_check(M, C, n, "glx_head1x1_weight_grad");
  if (rc != GLX_OK) return rc;
  HeadIn in;
  rc = head_in(x0, x1, c0, coef0, coef1, C, &in, "glx_head1x1_weight_grad");
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(C <= 512 && C / 64 <= 8, "glx_head1x1_weight_grad: C = %d", C);
  GLX_REQUIRE(workspace_bytes >= glx_head1x1_wgrad_workspace_bytes(C), "glx_head1x1_weight_grad: workspace too small");
  HeadGrad hg; HeadWOut wo;
  for (int k = 0; k < 3; ++k) {
    hg.p[k] = grad[k]; hg.n[k] = n[k];
    wo.w[k] = gW[k]; wo.b[k] = gb ? gb[k] : nullptr; wo.n[k] = n[k];
    GLX_REQUIRE(n[k] == 0 || grad[k], "glx_head1x1_weight_grad: head %d has no gradient", k);
  }
  long long blocks = (M + 63) / 64;
  if (blocks > HD_WBLOCKS) blocks = HD_WBLOCKS;
  hipLaunchKernelGGL(k_head_wgrad, dim3((unsigned)blocks), dim3(HD_THREADS), 0, (hipStream_t)stream, hg, in, (long long)M, C,
                     (float*)workspace);
  const int total = n[0] + n[1] + n[2];
  hipLaunchKernelGGL(k_head_wreduce, dim3(glx_divup((long long)total * (C + 1), 4)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, (int)blocks, C, wo);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
