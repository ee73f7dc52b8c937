// Fused PointNet feature extractor of the CVAE label-uncertainty generator (BASELINE config 4):
//   out[b, :] = max_p  W3 * relu(W2 * relu(W1 * x[b,:,p] + b1) + b2) + b3
// = PointNetfeat.forward in eval mode (cvae_uncertainty/point_net.py:10-28: three Conv1d(k=1) +
// BatchNorm1d, ReLU after the first two, max over the points) with the BatchNorms folded into
// (W, b) by the caller.  Widths 64 / 128 / 512 as in the reference (x = 1).
//
// The unfused PyTorch path writes and re-reads 5.9 GB of activations per pass at batch 4096 x 512
// points and runs at ~10 % of the fp32 matrix peak; here no activation leaves the CU:
//   * one block (4 waves) per object, 128 points per pass, every wave owns 2 tiles of 16 points;
//   * layer 1 (K = Cin <= 8) on the VALU, written directly in the register layout layer 2 wants;
//   * layers 2 and 3 on v_mfma_f32_16x16x4_f32 (exact fp32) with the point tile as the B operand:
//     the D registers of one layer ARE the B operands of the next (contraction index enumerated as
//     (tile, e) with the lane's quad as the k index), so activations never touch LDS either;
//   * W2 (32 KB, fragment order) sits in LDS for the whole block, W3 (256 KB) streams through two
//     8 KB LDS slabs, one 16-channel output tile at a time, prefetched through registers;
//   * the max over points is a 16-lane butterfly per output tile + a running max in LDS.
#include <float.h>

#include "glx_common.h"

typedef float pf32x4 __attribute__((ext_vector_type(4)));

#define PN_C1 64
#define PN_C2 128
#define PN_C3 512
#define PN_THREADS 256
#define PN_PTS 128          // points per pass (4 waves x 2 tiles x 16)
#define PN_MAXCIN 8
#define PN_SLAB (8 * 64 * 4)   // floats of one W3 slab: 8 input tiles x 64 lanes x 4

__global__ __launch_bounds__(PN_THREADS) void k_pointnet_feat(
    const float* __restrict__ pts, int CIN, int P, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W2p, const float* __restrict__ b2,
    const float* __restrict__ W3p, const float* __restrict__ b3, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w2 = smem;                          // 128 * 64   fragment order [t2][t1][lane][e]
  float* s_w3 = s_w2 + PN_C2 * PN_C1;          // 2 slabs
  float* s_w1 = s_w3 + 2 * PN_SLAB;            // 64 * 8
  float* s_b1 = s_w1 + PN_C1 * PN_MAXCIN;      // 64
  float* s_b2 = s_b1 + PN_C1;                  // 128
  float* s_max = s_b2 + PN_C2;                 // 4 waves * 512
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const long long obj = blockIdx.x;

  for (int e = tid; e < PN_C2 * PN_C1 / 4; e += PN_THREADS)
    reinterpret_cast<pf32x4*>(s_w2)[e] = reinterpret_cast<const pf32x4*>(W2p)[e];
  for (int e = tid; e < PN_C1 * PN_MAXCIN; e += PN_THREADS) {
    int c = e / PN_MAXCIN, ci = e - c * PN_MAXCIN;
    s_w1[e] = ci < CIN ? W1[c * CIN + ci] : 0.f;
  }
  if (tid < PN_C1) s_b1[tid] = b1[tid];
  if (tid < PN_C2) s_b2[tid] = b2[tid];
  for (int e = tid; e < 4 * PN_C3; e += PN_THREADS) s_max[e] = -FLT_MAX;
  // first W3 slab
  pf32x4 slab[2];
  slab[0] = reinterpret_cast<const pf32x4*>(W3p)[tid];
  slab[1] = reinterpret_cast<const pf32x4*>(W3p)[tid + PN_THREADS];
  reinterpret_cast<pf32x4*>(s_w3)[tid] = slab[0];
  reinterpret_cast<pf32x4*>(s_w3)[tid + PN_THREADS] = slab[1];
  __syncthreads();

  const float* xo = pts + obj * CIN * (long long)P;
  for (int p0 = 0; p0 < P; p0 += PN_PTS) {
    // ---- layer 1 on the VALU: h1[pt][t1*4+e] = channel 16*t1 + 4*q + e of point (tile pt, j)
    float h1[2][16];
    bool live[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int p = p0 + wave * 32 + pt * 16 + j;
      live[pt] = p < P;
      float x[PN_MAXCIN];
#pragma unroll
      for (int ci = 0; ci < PN_MAXCIN; ++ci) x[ci] = (ci < CIN && live[pt]) ? xo[(long long)ci * P + p] : 0.f;
#pragma unroll
      for (int t1 = 0; t1 < 4; ++t1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 16 * t1 + 4 * q + e;
          float a = s_b1[c];
#pragma unroll
          for (int ci = 0; ci < PN_MAXCIN; ++ci) a = fmaf(s_w1[c * PN_MAXCIN + ci], x[ci], a);
          h1[pt][t1 * 4 + e] = fmaxf(a, 0.f);
        }
      }
    }
    // ---- layer 2: 8 output tiles, K = 64 = 16 steps (t1, e); B operand = h1 registers
    float h2[2][32];
#pragma unroll
    for (int t2 = 0; t2 < 8; ++t2) {
      pf32x4 acc0 = pf32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int t1 = 0; t1 < 4; ++t1) {
        const pf32x4 a = *reinterpret_cast<const pf32x4*>(s_w2 + ((t2 * 4 + t1) * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], h1[0][t1 * 4 + e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], h1[1][t1 * 4 + e], acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float bb = s_b2[16 * t2 + 4 * q + e];
        h2[0][t2 * 4 + e] = fmaxf(acc0[e] + bb, 0.f);
        h2[1][t2 * 4 + e] = fmaxf(acc1[e] + bb, 0.f);
      }
    }
    // ---- layer 3: 32 output tiles, K = 128 = 32 steps (t2, e); W3 slabs double-buffered in LDS
    for (int t3 = 0; t3 < 32; ++t3) {
      const int nxt = (t3 + 1) & 31;     // slab 0 of the next pass follows slab 31
      const pf32x4* src = reinterpret_cast<const pf32x4*>(W3p + (size_t)nxt * PN_SLAB);
      slab[0] = src[tid];
      slab[1] = src[tid + PN_THREADS];
      const float* sw = s_w3 + (t3 & 1) * PN_SLAB;
      pf32x4 acc0 = pf32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int t2 = 0; t2 < 8; ++t2) {
        const pf32x4 a = *reinterpret_cast<const pf32x4*>(sw + (t2 * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], h2[0][t2 * 4 + e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], h2[1][t2 * 4 + e], acc1, 0, 0, 0);
        }
      }
      // max over the 32 points of this wave (padding points excluded), then into the running max
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = fmaxf(live[0] ? acc0[e] : -FLT_MAX, live[1] ? acc1[e] : -FLT_MAX);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        if (j == 0) {
          float* m = s_max + wave * PN_C3 + 16 * t3 + 4 * q + e;
          *m = fmaxf(*m, v);
        }
      }
      // publish the next slab into the other buffer (its last readers passed the previous barrier)
      float* dw = s_w3 + ((t3 + 1) & 1) * PN_SLAB;
      reinterpret_cast<pf32x4*>(dw)[tid] = slab[0];
      reinterpret_cast<pf32x4*>(dw)[tid + PN_THREADS] = slab[1];
      __syncthreads();
    }
  }
  // ---- max over the 4 waves, bias of layer 3 (max(x) + b == max(x + b))
  for (int c = tid; c < PN_C3; c += PN_THREADS) {
    float v = fmaxf(fmaxf(s_max[c], s_max[PN_C3 + c]), fmaxf(s_max[2 * PN_C3 + c], s_max[3 * PN_C3 + c]));
    out[obj * PN_C3 + c] = v + b3[c];
  }
}

extern "C" size_t glx_pointnet_feat_lds_bytes(void) {
  return (size_t)(PN_C2 * PN_C1 + 2 * PN_SLAB + PN_C1 * PN_MAXCIN + PN_C1 + PN_C2 + 4 * PN_C3) * 4;
}

extern "C" int glx_pointnet_feat(const float* points, int B, int Cin, int P, const float* W1,
                                 const float* b1, const float* W2p, const float* b2,
                                 const float* W3p, const float* b3, float* out, void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(points && W1 && b1 && W2p && b2 && W3p && b3 && out, "glx_pointnet_feat: null pointer");
  GLX_REQUIRE(Cin >= 1 && Cin <= PN_MAXCIN && P >= 1, "glx_pointnet_feat: Cin must be 1..8, P >= 1");
  const size_t lds = glx_pointnet_feat_lds_bytes();
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_pointnet_feat, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_pointnet_feat, dim3(B), dim3(PN_THREADS), lds, (hipStream_t)stream, points, Cin,
                     P, W1, b1, W2p, b2, W3p, b3, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// Small variant (all widths <= 16, e.g. the decoder's 4 -> 8 -> 8 -> 8 SimPointNetfeat,
// point_net.py:31-49): pure VALU, one block per object, a thread per point, the (folded) weights
// in LDS, max over points by wave butterflies + LDS.  Memory-bound on reading the points once.
#define PNS_MAXW 16
#define PNS_THREADS 256

__global__ __launch_bounds__(PNS_THREADS) void k_pointnet_feat_small(
    const float* __restrict__ pts, int CIN, int P, int C1, int C2, int C3,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2,
    const float* __restrict__ b2, const float* __restrict__ W3, const float* __restrict__ b3,
    float* __restrict__ out) {
  __shared__ float s_w1[PNS_MAXW * PN_MAXCIN], s_w2[PNS_MAXW * PNS_MAXW], s_w3[PNS_MAXW * PNS_MAXW];
  __shared__ float s_b[3 * PNS_MAXW];
  __shared__ float s_m[(PNS_THREADS / 64) * PNS_MAXW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < C1 * CIN; e += PNS_THREADS) s_w1[e] = W1[e];
  for (int e = tid; e < C2 * C1; e += PNS_THREADS) s_w2[e] = W2[e];
  for (int e = tid; e < C3 * C2; e += PNS_THREADS) s_w3[e] = W3[e];
  if (tid < C1) s_b[tid] = b1[tid];
  if (tid < C2) s_b[PNS_MAXW + tid] = b2[tid];
  if (tid < C3) s_b[2 * PNS_MAXW + tid] = b3[tid];
  __syncthreads();
  const float* xo = pts + (long long)blockIdx.x * CIN * P;
  float best[PNS_MAXW];
#pragma unroll
  for (int c = 0; c < PNS_MAXW; ++c) best[c] = -FLT_MAX;
  for (int p = tid; p < P; p += PNS_THREADS) {
    float x[PN_MAXCIN], h1[PNS_MAXW], h2[PNS_MAXW];
#pragma unroll
    for (int ci = 0; ci < PN_MAXCIN; ++ci) x[ci] = ci < CIN ? xo[(long long)ci * P + p] : 0.f;
#pragma unroll
    for (int c = 0; c < PNS_MAXW; ++c) {
      float a = 0.f;
      if (c < C1) {
        a = s_b[c];
#pragma unroll
        for (int ci = 0; ci < PN_MAXCIN; ++ci)
          if (ci < CIN) a = fmaf(s_w1[c * CIN + ci], x[ci], a);
      }
      h1[c] = fmaxf(a, 0.f);
    }
#pragma unroll
    for (int c = 0; c < PNS_MAXW; ++c) {
      float a = 0.f;
      if (c < C2) {
        a = s_b[PNS_MAXW + c];
#pragma unroll
        for (int k = 0; k < PNS_MAXW; ++k)
          if (k < C1) a = fmaf(s_w2[c * C1 + k], h1[k], a);
      }
      h2[c] = fmaxf(a, 0.f);
    }
#pragma unroll
    for (int c = 0; c < PNS_MAXW; ++c) {
      if (c < C3) {
        float a = s_b[2 * PNS_MAXW + c];
#pragma unroll
        for (int k = 0; k < PNS_MAXW; ++k)
          if (k < C2) a = fmaf(s_w3[c * C2 + k], h2[k], a);
        best[c] = fmaxf(best[c], a);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < PNS_MAXW; ++c) {
    float v = best[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if (lane == 0) s_m[wave * PNS_MAXW + c] = v;
  }
  __syncthreads();
  if (tid < C3) {
    float v = s_m[tid];
    for (int w = 1; w < PNS_THREADS / 64; ++w) v = fmaxf(v, s_m[w * PNS_MAXW + tid]);
    out[(long long)blockIdx.x * C3 + tid] = v;
  }
}

extern "C" int glx_pointnet_feat_small(const float* points, int B, int Cin, int P, int C1, int C2,
                                       int C3, const float* W1, const float* b1, const float* W2,
                                       const float* b2, const float* W3, const float* b3, float* out,
                                       void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(points && W1 && b1 && W2 && b2 && W3 && b3 && out, "glx_pointnet_feat_small: null pointer");
  GLX_REQUIRE(Cin >= 1 && Cin <= PN_MAXCIN && P >= 1 && C1 >= 1 && C2 >= 1 && C3 >= 1 &&
                  C1 <= PNS_MAXW && C2 <= PNS_MAXW && C3 <= PNS_MAXW,
              "glx_pointnet_feat_small: widths must be 1..16, Cin 1..8");
  hipLaunchKernelGGL(k_pointnet_feat_small, dim3(B), dim3(PNS_THREADS), 0, (hipStream_t)stream, points,
                     Cin, P, C1, C2, C3, W1, b1, W2, b2, W3, b3, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
