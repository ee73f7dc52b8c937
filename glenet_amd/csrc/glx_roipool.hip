// RoI-grid pooling, training path: the position branch of NeighborVoxelSAModuleMSG fused with the add + ReLU +
// max-pool over the neighbours (pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:88-108:
//   grouped_xyz - new_xyz -> mlps_pos = Conv2d(3, C, 1, bias=False) + BatchNorm2d(C)   [training-mode statistics]
//   relu(grouped_features + position_features) -> max over nsample).
// The reference materialises four (M, C, ns) tensors per scale for it (grouped features, positions, sum, ReLU;
// M = 4 x 128 x 216 grid points, ns = 16, C = 32: 226 MB each) and BatchNorm makes four more passes over the
// position tensor forward and backward; on this device that was 10.8 ms of a 24 ms training step.  None of
// those tensors is needed:
//   * the convolution is linear in rel = xyz[idx] - new_xyz, so the batch statistics of its output follow from
//     the first and second moments of rel (9 numbers): mean_c = w_c . E[rel], var_c = w_c^T Cov[rel] w_c.  One
//     reduction over the (m, s) pairs gives them (fp64, fixed order); BatchNorm then folds into the conv:
//     pos_c(rel) = (a_c w_c) . rel + b_c with a_c = gamma_c / sqrt(var_c + eps), b_c = beta_c - a_c mean_c;
//   * forward = one pass: per grid point, lanes = channels, the 16 neighbour rows are gathered (a neighbour's C
//     channels are one contiguous row), v = relu(feat + pos_c(rel)), running max and its slot;
//   * backward: the gradient lives at the winning slot only (one (m, c) entry each).  Into the features it is an
//     atomic add per (m, c); for BatchNorm and the conv weight
//         dbeta_c  = sum dz,  dgamma_c = sum dz * xhat,
//         dW_c     = a_c ( sum dz * rel  -  dbeta_c/n * S1  -  dgamma_c/n * (S2 w_c - mean_c S1) / sigma_c )
//     with S1 = sum rel, S2 = sum rel rel^T over all n = M * ns rows (the dense "minus the means" part of the
//     BatchNorm gradient only ever meets rel through those two sums) -- per-channel sums of 5 numbers over the
//     winners, one kernel.
// Empty balls (idx[m,0] < 0) count as ns rows of rel = 0 and feature 0, exactly as the reference's masked tensors.
#include "glx_common.h"
#include "glx_fill.h"
#include "glx_bn_state.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#define RP_SETS 16
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RP_THREADS 256
#define RP_MAX_BLOCKS 512
#define RP_MAXC 64

// `save` (6 * C floats, forward -> backward): mean[C], invstd[C], folded weights Wp[C][3], folded bias bp[C]

// ---- moments of rel over all (m, s): partial[block][9] = S1 (3), S2 (xx, xy, xz, yy, yz, zz)
__global__ __launch_bounds__(RP_THREADS) void k_rp_moments(const int* __restrict__ idx, const float* __restrict__ xyz,
                                                           const float* __restrict__ new_xyz, int M, int ns,
                                                           double* __restrict__ partial) {
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int m = blockIdx.x * RP_THREADS + threadIdx.x; m < M; m += gridDim.x * RP_THREADS) {
    const int* row = idx + (long long)m * ns;
    if (row[0] < 0) continue;
    const float qx = new_xyz[(long long)m * 3], qy = new_xyz[(long long)m * 3 + 1], qz = new_xyz[(long long)m * 3 + 2];
    for (int k = 0; k < ns; ++k) {
      const long long r = row[k];
      const float x = xyz[r * 3] - qx, y = xyz[r * 3 + 1] - qy, z = xyz[r * 3 + 2] - qz;
      s[0] += x; s[1] += y; s[2] += z;
      s[3] += (double)x * x; s[4] += (double)x * y; s[5] += (double)x * z;
      s[6] += (double)y * y; s[7] += (double)y * z; s[8] += (double)z * z;
    }
  }
  __shared__ double red[RP_THREADS];
  for (int q = 0; q < 9; ++q) {
    red[threadIdx.x] = s[q];
    __syncthreads();
    for (int o = RP_THREADS / 2; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[(long long)blockIdx.x * 9 + q] = red[0];
    __syncthreads();
  }
}

// Column sums of partial[nparts][Q] (Q <= RP_FIN_THREADS) by one block: thread (col, g) adds parts g, g + G, ...
// in a fixed order, the G groups are combined in LDS -> tot[col].  (A serial loop over 1024 partials per thread
// made the one-block finalize kernels the slowest launches of the stage: 263 us.)
#define RP_FIN_THREADS 1024
__device__ __forceinline__ void rp_column_sums(const double* __restrict__ partial, int nparts, int Q, double* s_grp,
                                               double* tot) {
  const int G = RP_FIN_THREADS / Q;
  const int col = threadIdx.x % Q, g = threadIdx.x / Q;
  double a = 0;
  if (g < G)
    for (int b = g; b < nparts; b += G) a += partial[(long long)b * Q + col];
  if (g < G) s_grp[g * Q + col] = a;
  __syncthreads();
  if (g == 0) {
    for (int k = 1; k < G; ++k) a += s_grp[k * Q + col];
    tot[col] = a;
  }
  __syncthreads();
}

// one block: statistics -> folded affine map, saved mean / invstd, running estimates
__global__ __launch_bounds__(RP_FIN_THREADS) void k_rp_finalize_fwd(
    const double* __restrict__ partial, int nparts, double n, const float* __restrict__ w,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, int training,
    float* __restrict__ running_mean, float* __restrict__ running_var, int C, float* __restrict__ save,
    double* __restrict__ moments) {
  __shared__ double s_grp[RP_FIN_THREADS];
  __shared__ double S[9];
  if (training) {
    rp_column_sums(partial, nparts, 9, s_grp, S);
  } else {
    if (threadIdx.x < 9) S[threadIdx.x] = 0;
    __syncthreads();
  }
  if (threadIdx.x < 9) moments[threadIdx.x] = S[threadIdx.x];
  const int c = threadIdx.x;
  if (c >= C) return;
  const double w0 = w[c * 3], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2];
  double mean, var;
  if (training) {
    const double mx = S[0] / n, my = S[1] / n, mz = S[2] / n;
    const double cxx = S[3] / n - mx * mx, cxy = S[4] / n - mx * my, cxz = S[5] / n - mx * mz;
    const double cyy = S[6] / n - my * my, cyz = S[7] / n - my * mz, czz = S[8] / n - mz * mz;
    mean = w0 * mx + w1 * my + w2 * mz;
    var = w0 * w0 * cxx + w1 * w1 * cyy + w2 * w2 * czz + 2 * (w0 * w1 * cxy + w0 * w2 * cxz + w1 * w2 * cyz);
    if (var < 0) var = 0;
    if (running_mean) {      // nn.BatchNorm: the running estimate takes the unbiased variance
      const double unb = n > 1 ? var * n / (n - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float a = (gamma ? gamma[c] : 1.f) * invstd;
  save[c] = (float)mean;
  save[C + c] = invstd;
  save[2 * C + c * 3] = a * w[c * 3];
  save[2 * C + c * 3 + 1] = a * w[c * 3 + 1];
  save[2 * C + c * 3 + 2] = a * w[c * 3 + 2];
  save[5 * C + c] = (beta ? beta[c] : 0.f) - (float)mean * a;
}

// forward: lanes = channels; a wave covers 64 / C grid points per pass
// Lanes = (grid point, float4 of channels): C / 4 lanes per point, 64 / (C / 4) points per wave -- with a lane per
// channel a wave had two points in flight and the kernel was one L2 latency per 2 x 16 rows (0.7 TB/s of gathers).
// OUT: the layer's output MLP rides along (voxel_pool_modules.py:105-108: mlps_out = Conv1d(C, C, 1, bias=False) +
// BatchNorm1d + ReLU on the pooled features): y_out[m, :] = W_out pooled[m, :] is formed from the point's pooled row while
// it is still in the lanes' registers (Q lanes per point exchange their float4s), and the BatchNorm's batch statistics of
// y_out are accumulated on the way (BnState: block sums -> accumulator set -> ticket -> the last block finalizes) -- the
// GEMM launch and the statistics pass over (M, C) disappear from the RoI branch's forward chain.
struct RpOut {
  const float* w_out;      // (C, C) row-major: y[co] = sum_c w_out[co][c] * pooled[c]
  float* y_out;            // (M, C)
  BnState* bn_state;
  BnFinalize bn;
};

template <int C, bool OUT = false>
__global__ __launch_bounds__(RP_THREADS) void k_rp_forward(const float* __restrict__ feats, const float* __restrict__ xyz,
                                                           const float* __restrict__ new_xyz, const int* __restrict__ idx,
                                                           int M, int ns, const float* __restrict__ save,
                                                           float* __restrict__ pooled, unsigned char* __restrict__ arg,
                                                           RpOut ro) {
  constexpr int Q = C / 4;                     // lanes per grid point
  constexpr int PPB = RP_THREADS / Q;          // grid points per block pass
  const int q = threadIdx.x % Q, sub = threadIdx.x / Q;
  __shared__ float s_wout[OUT ? C * (C + 4) : 1];      // row co at co * (C + 4): the 4 rows of a lane in different banks
  float osum[4] = {0.f, 0.f, 0.f, 0.f}, osq[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (OUT) {
    for (int e = threadIdx.x; e < C * C; e += RP_THREADS) s_wout[(e / C) * (C + 4) + (e % C)] = ro.w_out[e];
    __syncthreads();
  }
  float wx[4], wy[4], wz[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 * q + i;
    wx[i] = save[2 * C + c * 3]; wy[i] = save[2 * C + c * 3 + 1]; wz[i] = save[2 * C + c * 3 + 2]; b[i] = save[5 * C + c];
  }
  for (long long m = (long long)blockIdx.x * PPB + sub; m < M; m += (long long)gridDim.x * PPB) {
    const int* row = idx + m * ns;
    float best[4] = {-1.f, -1.f, -1.f, -1.f};
    int bi[4] = {0, 0, 0, 0};
    const int r0 = row[0];
    if (r0 < 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) best[i] = fmaxf(b[i], 0.f);      // every slot holds feature 0 and rel 0
    } else {
      const float qx = new_xyz[m * 3], qy = new_xyz[m * 3 + 1], qz = new_xyz[m * 3 + 2];
      // The query pads a ball with fewer than ns voxels by repeating its first hit (voxel_query_gpu.cu:75-86); the
      // repeats can never win the strict `>` below, so the scan stops at the first of them -- on LiDAR surfaces a
      // ball holds 4-6 voxels of 16, which is most of this kernel's gather traffic.
      bool more = true;
      for (int s0 = 0; s0 < ns && more; s0 += 4) {     // 4 row gathers in flight per lane
        f32x4 f[4];
        float x[4], y[4], z[4];
        int live = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = s0 + j < ns ? row[s0 + j] : r0;
          const bool ok = more && s0 + j < ns && (s0 + j == 0 || r != r0);
          more = ok;                             // uniform per grid point: its lanes read the same row
          if (ok) {
            f[j] = *reinterpret_cast<const f32x4*>(feats + (long long)r * C + 4 * q);
            x[j] = xyz[(long long)r * 3]; y[j] = xyz[(long long)r * 3 + 1]; z[j] = xyz[(long long)r * 3 + 2];
            live = j + 1;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j >= live) break;
          const float dx = x[j] - qx, dy = y[j] - qy, dz = z[j] - qz;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float p = dx * wx[i] + dy * wy[i] + dz * wz[i] + b[i];
            const float v = fmaxf(f[j][i] + p, 0.f);
            if (v > best[i]) { best[i] = v; bi[i] = s0 + j; }
          }
        }
      }
    }
    *reinterpret_cast<f32x4*>(pooled + m * C + 4 * q) = f32x4{best[0], best[1], best[2], best[3]};
    *reinterpret_cast<uchar4*>(arg + m * C + 4 * q) =
        make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
    if constexpr (OUT) {
      // the point's pooled row from its Q lanes (lane base + qq holds channels 4 qq ..), this lane's four outputs
      const int lane = threadIdx.x & 63, base = lane - q;
      float y[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qq = 0; qq < Q; ++qq) {
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = __shfl(best[i], base + qq, 64);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(s_wout + (4 * q + o) * (C + 4) + 4 * qq);
          y[o] = fmaf(wv[0], p[0], y[o]);
          y[o] = fmaf(wv[1], p[1], y[o]);
          y[o] = fmaf(wv[2], p[2], y[o]);
          y[o] = fmaf(wv[3], p[3], y[o]);
        }
      }
      *reinterpret_cast<f32x4*>(ro.y_out + m * C + 4 * q) = f32x4{y[0], y[1], y[2], y[3]};
#pragma unroll
      for (int o = 0; o < 4; ++o) { osum[o] += y[o]; osq[o] += y[o] * y[o]; }
    }
  }
  if constexpr (OUT) {
    // per-channel sums of the block: lanes with the same q, then the waves; fp64 from here on
    double d0[4], d1[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) { d0[o] = (double)osum[o]; d1[o] = (double)osq[o]; }
#pragma unroll
    for (int off = Q; off < 64; off <<= 1)
#pragma unroll
      for (int o = 0; o < 4; ++o) { d0[o] += __shfl_xor(d0[o], off, 64); d1[o] += __shfl_xor(d1[o], off, 64); }
    __shared__ double s_red[RP_THREADS / 64][2][C];
    __shared__ int s_last;
    __shared__ double s_fin[RP_THREADS][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < Q)
#pragma unroll
      for (int o = 0; o < 4; ++o) { s_red[wave][0][4 * lane + o] = d0[o]; s_red[wave][1][4 * lane + o] = d1[o]; }
    __syncthreads();
    double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
    if ((int)threadIdx.x < Q)
      for (int w = 0; w < RP_THREADS / 64; ++w)
#pragma unroll
        for (int o = 0; o < 4; ++o) { a0[o] += s_red[w][0][4 * threadIdx.x + o]; a1[o] += s_red[w][1][4 * threadIdx.x + o]; }
    if (!bn_contribute(ro.bn_state, C, a0, a1, gridDim.x, &s_last)) return;
    bn_finalize_sets<false, RP_THREADS>(ro.bn_state, ro.bn, C, M, s_fin);
  }
}

// backward: feature gradient by atomic adds at the winning slots + per-channel sums
// partial[block][c][5] = sum dz, sum dz*xhat, sum dz*rel (3)
// AGG: the feature gradient is summed per block in LDS first.  The winners of neighbouring grid points are the same few
// voxels (a RoI's 216 points crown some tens of rows, 512 RoIs sit on the same objects), and float atomics execute at the
// memory side where adds to one row serialise: without the atomics the launch takes 36 us on every scale, with one global atomic
// per (point, channel) 114 / 72 / 38 us (x_conv4 / x_conv3 / x_conv2: the coarser the scale the fewer the rows).  A block takes
// RP_AGG_POINTS CONSECUTIVE points, adds their gradients into a hash table of (row -> C sums) in LDS (ds_add_f32) and sends
// one global atomic per (occupied slot, channel) at the end, as whole 128-byte rows; a row that finds no slot within eight probes
// goes to memory directly.
#define RP_AGG_POINTS 256
#define RP_AGG_SLOTS 256
template <int C, bool AGG>
__global__ __launch_bounds__(RP_THREADS) void k_rp_backward(const float* __restrict__ dpooled,
                                                            const float* __restrict__ pooled,
                                                            const unsigned char* __restrict__ arg,
                                                            const int* __restrict__ idx, const float* __restrict__ xyz,
                                                            const float* __restrict__ new_xyz, int M, int ns,
                                                            const float* __restrict__ w, const float* __restrict__ save,
                                                            float* __restrict__ dfeats, double* __restrict__ partial) {
  // lanes = (grid point, float4 of channels) as in the forward kernel: four winners' rows in flight per lane
  constexpr int Q = C / 4, PPB = RP_THREADS / Q, PPW = 64 / Q;        // lanes per point, points per block / wave
  const int q = threadIdx.x % Q, sub = threadIdx.x / Q;
  float w0[4], w1[4], w2[4], mean[4], invstd[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 * q + i;
    w0[i] = w[c * 3]; w1[i] = w[c * 3 + 1]; w2[i] = w[c * 3 + 2];
    mean[i] = save[c]; invstd[i] = save[C + c];
  }
  double s[4][5];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 5; ++k) s[i][k] = 0;
  const int lane = threadIdx.x & 63;
  constexpr int NSLOT = AGG ? (C <= 32 ? RP_AGG_SLOTS : RP_AGG_SLOTS / 2) : 1;
  __shared__ int s_key[NSLOT];
  __shared__ float s_val[AGG ? NSLOT * C : 1];
  if (AGG) {
    for (int e = threadIdx.x; e < NSLOT; e += RP_THREADS) s_key[e] = -1;
    for (int e = threadIdx.x; e < NSLOT * C; e += RP_THREADS) s_val[e] = 0.f;
    __syncthreads();
  }
  // AGG: a block owns RP_AGG_POINTS consecutive points (passes of PPB); else passes strided over the grid
  const long long m_lo = AGG ? (long long)blockIdx.x * RP_AGG_POINTS : (long long)blockIdx.x * PPB;
  const long long m_hi = AGG ? (m_lo + RP_AGG_POINTS < M ? m_lo + RP_AGG_POINTS : (long long)M) : (long long)M;
  const long long m_step = AGG ? PPB : (long long)gridDim.x * PPB;
  for (long long m = m_lo + sub; m < m_hi; m += m_step) {
    // all PPW points of this wave's pass exist <=> the last one does (points of a wave are consecutive)
    const bool wave_full = m - (sub % PPW) + (PPW - 1) < M;
    const f32x4 pv = *reinterpret_cast<const f32x4*>(pooled + m * C + 4 * q);
    const f32x4 dv = *reinterpret_cast<const f32x4*>(dpooled + m * C + 4 * q);
    float g[4];
    bool any = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) { g[i] = pv[i] > 0.f ? dv[i] : 0.f; any |= g[i] != 0.f; }
    // no `continue` before the shuffles below: the lanes of a wave stay together
    const int* row = idx + m * ns;
    const bool filled = any && row[0] >= 0;
    float x[4] = {0.f, 0.f, 0.f, 0.f}, y[4] = {0.f, 0.f, 0.f, 0.f}, z[4] = {0.f, 0.f, 0.f, 0.f};
    int r[4] = {-1, -1, -1, -1};
    if (filled) {
      const uchar4 av = *reinterpret_cast<const uchar4*>(arg + m * C + 4 * q);
      const unsigned char a4[4] = {av.x, av.y, av.z, av.w};
      const float qx = new_xyz[m * 3], qy = new_xyz[m * 3 + 1], qz = new_xyz[m * 3 + 2];
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = g[i] != 0.f ? row[a4[i]] : -1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (r[i] < 0) continue;
        x[i] = xyz[(long long)r[i] * 3] - qx;
        y[i] = xyz[(long long)r[i] * 3 + 1] - qy;
        z[i] = xyz[(long long)r[i] * 3 + 2] - qz;
      }
    }
    // neighbouring grid points of a RoI mostly crown the same voxel: the points of a wave that share a winner add
    // their gradients first and send ONE atomic (the feature rows near a box are hot spots of the L2 atomic units)
    if (AGG) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (r[i] < 0) continue;
        unsigned sl = ((unsigned)r[i] * 2654435761u) >> 16 & (NSLOT - 1);
        bool put = false;
#pragma unroll 1
        for (int probe = 0; probe < 8 && !put; ++probe) {
          const int k = atomicCAS(&s_key[sl], -1, r[i]);
          if (k == -1 || k == r[i]) {
            atomicAdd(&s_val[sl * C + 4 * q + i], g[i]);
            put = true;
          } else {
            sl = (sl + 1) & (NSLOT - 1);
          }
        }
        if (!put) atomicAdd(dfeats + (long long)r[i] * C + 4 * q + i, g[i]);
      }
    } else if (wave_full) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float acc = r[i] >= 0 ? g[i] : 0.f;
        bool leader = r[i] >= 0;
#pragma unroll
        for (int off = Q; off < 64; off += Q) {
          const int src = (lane + off) & 63;
          const int ro = __shfl(r[i], src, 64);
          const float go = __shfl(g[i], src, 64);
          if (r[i] >= 0 && ro == r[i]) {
            acc += go;
            if (src < lane) leader = false;
          }
        }
        if (leader) atomicAdd(dfeats + (long long)r[i] * C + 4 * q + i, acc);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (r[i] >= 0) atomicAdd(dfeats + (long long)r[i] * C + 4 * q + i, g[i]);
    }
    if (!any) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (g[i] == 0.f) continue;
      const float xh = (x[i] * w0[i] + y[i] * w1[i] + z[i] * w2[i] - mean[i]) * invstd[i];
      s[i][0] += g[i];
      s[i][1] += (double)g[i] * xh;
      s[i][2] += (double)g[i] * x[i];
      s[i][3] += (double)g[i] * y[i];
      s[i][4] += (double)g[i] * z[i];
    }
  }
  if (AGG) {                 // the block's table -> memory: consecutive threads = the channels of a slot (128-byte rows)
    __syncthreads();
    for (int e = threadIdx.x; e < NSLOT * C; e += RP_THREADS) {
      const int row = s_key[e / C];
      const float v = s_val[e];
      if (row >= 0 && v != 0.f) atomicAdd(dfeats + (long long)row * C + (e % C), v);
    }
  }
  // sums over the points of the wave (lanes with the same q) by shuffles, over the 4 waves in LDS
#pragma unroll
  for (int off = Q; off < 64; off <<= 1)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 5; ++k) s[i][k] += __shfl_xor(s[i][k], off, 64);
  __shared__ double red[RP_THREADS / 64][16][4][5];        // [wave][q][i][quantity], Q <= 16
  const int wave = threadIdx.x >> 6;
  if (lane < Q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 5; ++k) red[wave][lane][i][k] = s[i][k];
  __syncthreads();
  for (int e = threadIdx.x; e < 5 * C; e += RP_THREADS) {       // e = k * C + c
    const int k = e / C, c = e - k * C;
    double a = 0;
    for (int wv = 0; wv < RP_THREADS / 64; ++wv) a += red[wv][c >> 2][c & 3][k];
    unsafeAtomicAdd(partial + ((long long)(blockIdx.x % RP_SETS) * 5 + k) * C + c, a);
  }
}

__global__ __launch_bounds__(RP_FIN_THREADS) void k_rp_finalize_bwd(
    const double* __restrict__ partial, int nparts, double n, const double* __restrict__ moments,
    const float* __restrict__ w, const float* __restrict__ gamma, const float* __restrict__ save, int training,
    int C, float* __restrict__ dW, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double s_grp[RP_FIN_THREADS];
  __shared__ double tot[RP_MAXC * 5];
  rp_column_sums(partial, nparts, 5 * C, s_grp, tot);       // tot[q * C + c]
  const int c = threadIdx.x;
  if (c >= C) return;
  double s[5];
  for (int q = 0; q < 5; ++q) s[q] = tot[q * C + c];
  const double mean = save[c], invstd = save[C + c];
  const double a = (gamma ? gamma[c] : 1.f) * invstd;
  if (dbeta) dbeta[c] = (float)s[0];
  if (dgamma) dgamma[c] = (float)s[1];
  double d[3] = {s[2], s[3], s[4]};
  if (training) {
    const double w0 = w[c * 3], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2];
    const double* S = moments;
    const double yx[3] = {S[3] * w0 + S[4] * w1 + S[5] * w2 - mean * S[0],      // sum y * rel - mean * sum rel
                          S[4] * w0 + S[6] * w1 + S[7] * w2 - mean * S[1],
                          S[5] * w0 + S[7] * w1 + S[8] * w2 - mean * S[2]};
    for (int k = 0; k < 3; ++k) d[k] -= s[0] / n * S[k] + s[1] / n * invstd * yx[k];
  }
  for (int k = 0; k < 3; ++k) dW[c * 3 + k] = (float)(a * d[k]);
}

static bool rp_channels_ok(int C) { return C == 16 || C == 32 || C == 64; }
static int rp_blocks(int M, int C) {
  const int ppb = RP_THREADS / C;
  long long b = ((long long)M + ppb * 4 - 1) / (ppb * 4);
  return (int)(b < 1 ? 1 : (b > RP_MAX_BLOCKS ? RP_MAX_BLOCKS : b));
}
static size_t rp_partial_bytes(int C) { return glx_align((size_t)RP_MAX_BLOCKS * (C * 5 > 9 ? C * 5 : 9) * sizeof(double)); }

extern "C" size_t glx_pos_pool_workspace_bytes(int C) { return rp_partial_bytes(C) + 256; }
extern "C" int glx_pos_pool_save_floats(int C) { return 6 * C; }

extern "C" int glx_pos_pool_forward_out(const float* feats, int N, int C, const float* xyz, const float* new_xyz,
                                        const int32_t* idx, int M, int nsample, const float* w_pos, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float momentum,
                                        float eps, int training, float* pooled, uint8_t* arg, float* save,
                                        double* moments, const float* w_out, float* y_out, const glx_bn_stats* bn_out,
                                        void* workspace, size_t workspace_bytes, void* stream);
extern "C" int glx_pos_pool_forward(const float* feats, int N, int C, const float* xyz, const float* new_xyz,
                                    const int32_t* idx, int M, int nsample, const float* w_pos, const float* gamma,
                                    const float* beta, float* running_mean, float* running_var, float momentum,
                                    float eps, int training, float* pooled, uint8_t* arg, float* save,
                                    double* moments, void* workspace, size_t workspace_bytes, void* stream) {
  return glx_pos_pool_forward_out(feats, N, C, xyz, new_xyz, idx, M, nsample, w_pos, gamma, beta, running_mean, running_var,
                                  momentum, eps, training, pooled, arg, save, moments, nullptr, nullptr, nullptr, workspace,
                                  workspace_bytes, stream);
}

// ... with the output MLP of the layer: y_out (M, C) = pooled @ w_out^T (w_out (C, C) row-major) and the training-mode
// BatchNorm statistics of y_out in the same launch (bn_out as glx_conv_opts.bn; all three NULL = glx_pos_pool_forward)
extern "C" int glx_pos_pool_forward_out(const float* feats, int N, int C, const float* xyz, const float* new_xyz,
                                        const int32_t* idx, int M, int nsample, const float* w_pos, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float momentum,
                                        float eps, int training, float* pooled, uint8_t* arg, float* save,
                                        double* moments, const float* w_out, float* y_out, const glx_bn_stats* bn_out,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(rp_channels_ok(C), "glx_pos_pool_forward: C=%d (16, 32 or 64)", C);
  GLX_REQUIRE((w_out == nullptr) == (y_out == nullptr) && (w_out == nullptr) == (bn_out == nullptr),
              "glx_pos_pool_forward_out: w_out, y_out and bn_out come together");
  GLX_REQUIRE(!bn_out || (C <= 32 && M > 0 && bn_out->state && bn_out->coef && bn_out->save_mean && bn_out->save_invstd),
              "glx_pos_pool_forward_out: the output MLP needs C <= 32, M > 0 and the statistics' buffers");
  RpOut ro{w_out, y_out, bn_out ? (BnState*)bn_out->state : nullptr, BnFinalize{}};
  if (bn_out)
    ro.bn = BnFinalize{bn_out->gamma, bn_out->beta, bn_out->eps, bn_out->momentum, bn_out->coef, bn_out->save_mean,
                       bn_out->save_invstd, bn_out->running_mean, bn_out->running_var, nullptr, nullptr, nullptr};
  GLX_REQUIRE(nsample > 0 && nsample <= 255, "glx_pos_pool_forward: nsample=%d", nsample);
  GLX_REQUIRE(w_pos && save && moments && (M == 0 || (feats && xyz && new_xyz && idx && pooled && arg)) &&
                  (training || (running_mean && running_var)),
              "glx_pos_pool_forward: null pointer");
  if (!workspace || workspace_bytes < glx_pos_pool_workspace_bytes(C) - 256) {
    glx_set_error("glx_pos_pool_forward: workspace %zu < %zu bytes", workspace_bytes, glx_pos_pool_workspace_bytes(C) - 256);
    return GLX_EWORKSPACE;
  }
  (void)N;
  hipStream_t st = (hipStream_t)stream;
  int nb = (M + RP_THREADS - 1) / RP_THREADS;
  nb = nb < 1 ? 1 : (nb > RP_MAX_BLOCKS ? RP_MAX_BLOCKS : nb);
  if (training && M > 0)
    hipLaunchKernelGGL(k_rp_moments, dim3(nb), dim3(RP_THREADS), 0, st, idx, xyz, new_xyz, M, nsample, (double*)workspace);
  hipLaunchKernelGGL(k_rp_finalize_fwd, dim3(1), dim3(RP_FIN_THREADS), 0, st, (const double*)workspace, M > 0 ? nb : 0,
                     (double)M * nsample, w_pos, gamma, beta, eps, momentum, training, running_mean, running_var, C, save,
                     moments);
  if (M > 0) {
    const int ppb_f = RP_THREADS / (C / 4);
    const int blocks = (int)(((long long)M + ppb_f - 1) / ppb_f > 4096 ? 4096 : ((long long)M + ppb_f - 1) / ppb_f);
    if (C == 16 && w_out)
      hipLaunchKernelGGL((k_rp_forward<16, true>), dim3(blocks), dim3(RP_THREADS), 0, st, feats, xyz, new_xyz, idx, M, nsample,
                         (const float*)save, pooled, arg, ro);
    else if (C == 32 && w_out)
      hipLaunchKernelGGL((k_rp_forward<32, true>), dim3(blocks), dim3(RP_THREADS), 0, st, feats, xyz, new_xyz, idx, M, nsample,
                         (const float*)save, pooled, arg, ro);
    else if (C == 16)
      hipLaunchKernelGGL((k_rp_forward<16>), dim3(blocks), dim3(RP_THREADS), 0, st, feats, xyz, new_xyz, idx, M, nsample,
                         (const float*)save, pooled, arg, ro);
    else if (C == 32)
      hipLaunchKernelGGL((k_rp_forward<32>), dim3(blocks), dim3(RP_THREADS), 0, st, feats, xyz, new_xyz, idx, M, nsample,
                         (const float*)save, pooled, arg, ro);
    else
      hipLaunchKernelGGL((k_rp_forward<64>), dim3(blocks), dim3(RP_THREADS), 0, st, feats, xyz, new_xyz, idx, M, nsample,
                         (const float*)save, pooled, arg, ro);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

static int g_rp_bwd_agg = 1;
// 1 (default): the feature gradient through the per-block LDS table; 0: one global atomic per (point, channel).  Returns the previous.
extern "C" int glx_pos_pool_set_backward_form(int aggregate) {
  const int old = g_rp_bwd_agg;
  g_rp_bwd_agg = aggregate ? 1 : 0;
  return old;
}

extern "C" int glx_pos_pool_backward(const float* dpooled, const float* pooled, const uint8_t* arg, const int32_t* idx,
                                     const float* xyz, const float* new_xyz, int M, int nsample, int C, int N,
                                     const float* w_pos, const float* gamma, const float* save, const double* moments,
                                     int training, float* dfeats, float* dW, float* dgamma, float* dbeta,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(rp_channels_ok(C), "glx_pos_pool_backward: C=%d (16, 32 or 64)", C);
  GLX_REQUIRE(w_pos && save && moments && dW && (N == 0 || dfeats) &&
                  (M == 0 || (dpooled && pooled && arg && idx && xyz && new_xyz)),
              "glx_pos_pool_backward: null pointer");
  if (!workspace || workspace_bytes < glx_pos_pool_workspace_bytes(C) - 256) {
    glx_set_error("glx_pos_pool_backward: workspace %zu < %zu bytes", workspace_bytes, glx_pos_pool_workspace_bytes(C) - 256);
    return GLX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  // the blocks add their five per-channel sums to one of RP_SETS accumulator sets with fp64 atomics (zeroed by the
  // launch that zeroes dfeats): the finalize reads 16 rows instead of one per block (512 x 160 doubles: 25 us)
  GlxFillJob zj[2] = {{dfeats, (size_t)N * C * sizeof(float), 0},
                      {workspace, (size_t)RP_SETS * 5 * C * sizeof(double), 0}};
  int rc = glx_fill_multi(zj, 2, st);
  if (rc != GLX_OK) return rc;
  const int ppb_b = RP_THREADS / (C / 4);
  // per launch the table pays where the winners are few (114 -> 57 us on x_conv4's 26 k rows, 72 -> 59 on x_conv3's 42 k, 38 -> 55
  // on x_conv2's 62 k; 4 frames x 128 RoIs x 216 points each) -- on the training step it pays on every scale: 5.46-5.48 ms with the
  // table everywhere against 5.52-5.55 with it on the coarsest scale only and 5.59-5.61 without (the serialised adds at the
  // memory side also hold up the BEV backward that runs beside them).  GLX_RP_BWD_AGG=0: one global atomic per (point, channel)
  const bool agg = g_rp_bwd_agg != 0;
  const long long want = agg ? ((long long)M + RP_AGG_POINTS - 1) / RP_AGG_POINTS : ((long long)M + ppb_b - 1) / ppb_b;
  // (the aggregating form needs every point covered by ITS block: no cap on the grid)
  const int blocks = M > 0 ? (int)(agg ? want : (want > 4096 ? 4096 : want)) : 0;
  if (M > 0) {
#define RP_BWD_LAUNCH(C_, A_)                                                                                              \
  hipLaunchKernelGGL((k_rp_backward<C_, A_>), dim3(blocks), dim3(RP_THREADS), 0, st, dpooled, pooled, arg, idx, xyz, new_xyz, M, \
                     nsample, w_pos, save, dfeats, (double*)workspace)
    if (C == 16) { if (agg) RP_BWD_LAUNCH(16, true); else RP_BWD_LAUNCH(16, false); }
    else if (C == 32) { if (agg) RP_BWD_LAUNCH(32, true); else RP_BWD_LAUNCH(32, false); }
    else { if (agg) RP_BWD_LAUNCH(64, true); else RP_BWD_LAUNCH(64, false); }
#undef RP_BWD_LAUNCH
  }
  hipLaunchKernelGGL(k_rp_finalize_bwd, dim3(1), dim3(RP_FIN_THREADS), 0, st, (const double*)workspace,
                     blocks > RP_SETS ? RP_SETS : blocks, (double)M * nsample,
                     moments, w_pos, gamma, save, training, C, dW, dgamma, dbeta);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
