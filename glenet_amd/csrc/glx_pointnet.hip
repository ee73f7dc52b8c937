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
#include "glx_bf16x3.h"
#include "glx_bn_state.h"

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

// ------------------------------------------------------------------------------------------------ f16 x 2 form (round 6)
// The same extractor with layers 2 and 3 on v_mfma_f32_16x16x32_f16: every operand is split into two fp16 pieces of the value
// scaled by a power of two (the arithmetic of csrc/glx_conv2d.hip / glx_sconv.hip: three MFMAs per product tile -- b_w a_x,
// a_w b_x, a_w a_x --, >= 20.4 bits per product, fp32 sums), 96 matrix-pipe cycles per 16 x 16 x 64 block instead of 512.
// The register chaining of the fp32 kernel survives: a 16 x 16 x 32 MFMA wants, per lane (j, q), EIGHT consecutive
// contraction slots of its point j -- and the contraction index may be enumerated any way both operands agree on, so k-step s
// takes the lane's own D registers of input tiles 2 s and 2 s + 1 (channels 32 s + 16 h + 4 q + e, slot 4 h + e); the packed
// weights use the same enumeration (dense_path.PointFeat._packed_f16).  Scales: a weight row (output channel) by its own power of
// two (ew2 / ew3, host side), a point by the maximum of its activations over the channels (4 lanes of a quad column: two
// register-half swaps); both come out per product tile (ldexp).  Layer 1 (K <= 8) stays on the VALU in fp32.
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef int pi32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void pn_split2(float xs, _Float16& a, _Float16& b) {
  a = (_Float16)xs;
  b = (_Float16)(xs - (float)a);
}
// the exponent e that puts m = max |x| into [2^14, 2^15); 0 for m == 0
__device__ __forceinline__ int pn_exponent(float m) {
  if (!(m > 0.f)) return 0;
  const int e = 14 - (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xFF) + 127;
  return e > 110 ? 110 : (e < -110 ? -110 : e);
}
// max over the four lanes (j, q = 0..3) that hold one point: lanes 16 apart
__device__ __forceinline__ float pn_point_max(float m) {
  const unsigned mu = __builtin_bit_cast(unsigned, m);
  const auto s16 = __builtin_amdgcn_permlane16_swap(mu, mu, false, false);
  m = fmaxf(__builtin_bit_cast(float, (unsigned)s16[0]), __builtin_bit_cast(float, (unsigned)s16[1]));
  const unsigned mv = __builtin_bit_cast(unsigned, m);
  const auto s32 = __builtin_amdgcn_permlane32_swap(mv, mv, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)s32[0]), __builtin_bit_cast(float, (unsigned)s32[1]));
}

#define PNH_SLAB_U4 512          // uint4 per W3 slab: 4 k-steps x 2 planes x 64 lanes (8 KB)
#define PNH_RING 3               // slabs in LDS: one being read, two on their way

// W3's slabs go L2 -> LDS by DMA (no registers, no ds_write); the compiler does not see these operations, so the kernel counts
// them itself: every thread issues exactly two per slab, and nothing else of its own is in flight while the ring turns.
__device__ __forceinline__ unsigned pn_lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void pn_dma16(const void* base_uniform, unsigned off, unsigned lds_uniform) {
  // 64 lanes x 16 B from base + off (per lane) -> LDS at lds_uniform + 16 lane
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base_uniform), "s"(lds_uniform)
               : "memory");
}
__device__ __forceinline__ void pn_dma4(const void* base_uniform, unsigned off, unsigned lds_uniform) {
  // 64 lanes x 4 B from base + off (per lane) -> LDS at lds_uniform + 4 lane
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(off), "s"(base_uniform), "s"(lds_uniform)
               : "memory");
}
__device__ __forceinline__ float pn_row_sum(float v) {      // the sum over a row of 16 lanes, in every lane, always in this order
#define PN_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
  PN_DPP_ADD(0xB1);
  PN_DPP_ADD(0x4E);
  PN_DPP_ADD(0x141);
  PN_DPP_ADD(0x140);
#undef PN_DPP_ADD
  return v;
}
__device__ __forceinline__ float pn_row_max(float v) {      // the maximum over a row of 16 lanes, in every lane (four DPP steps)
#define PN_DPP_MAX(ctrl) \
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true)))
  PN_DPP_MAX(0xB1);      // quad_perm [1, 0, 3, 2]
  PN_DPP_MAX(0x4E);      // quad_perm [2, 3, 0, 1]
  PN_DPP_MAX(0x141);     // row_half_mirror
  PN_DPP_MAX(0x140);     // row_mirror
#undef PN_DPP_MAX
  return v;
}
template <int N>
__device__ __forceinline__ void pn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NARROW: the decoder's narrow extractor (4 -> 8 -> 8 -> 8, point_net.py:31-49) rides along on the same points: its folded
// weights (`nw`: W1 (8 x 8, input channels zero-padded), b1, W2, b2, W3, b3 = 216 floats) become MFMA operands (below):
// 2 x 12 fp32 MFMAs per pass beside the 864 fp16 ones, and a launch less.  (As VALU sums with the weights at uniform addresses the
// compiler hoisted 216 scalar loads out of the loop and spilled them to lanes: +290 us per launch.)
#define PNN_W 8
template <bool NARROW>
__global__ __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pointnet_feat_f16(
    const float* __restrict__ pts, int CIN, int P, const float* __restrict__ W1, const float* __restrict__ b1,
    const uint4* __restrict__ W2h, const int* __restrict__ ew2, const float* __restrict__ b2, const uint4* __restrict__ W3h,
    const int* __restrict__ ew3, const float* __restrict__ b3, float* __restrict__ out, const float* __restrict__ nw,
    float* __restrict__ nout) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* s_w2 = reinterpret_cast<uint4*>(smem);              // 8 t2 x 2 s x 2 planes x 64 lanes          (32 KB)
  uint4* s_w3 = s_w2 + 8 * 2 * 2 * 64;                       // PNH_RING slabs of PNH_SLAB_U4              (24 KB)
  float* s_w1 = reinterpret_cast<float*>(s_w3 + PNH_RING * PNH_SLAB_U4);   // 64 * 8
  float* s_b1 = s_w1 + PN_C1 * PN_MAXCIN;                    // 64
  float* s_b2 = s_b1 + PN_C1;                                // 128
  int* s_e2 = reinterpret_cast<int*>(s_b2 + PN_C2);          // 128
  int* s_e3 = s_e2 + PN_C2;                                  // 512: MINUS the rows' exponents
  float* s_max = reinterpret_cast<float*>(s_e3 + PN_C3);     // 4 waves * 512
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const long long obj = blockIdx.x;

  // the ring's first two slabs are on their way while the rest of the block's constants are staged
  const unsigned ring = pn_lds_addr(s_w3);
  const unsigned my = __builtin_amdgcn_readfirstlane(wave) * 1024u;     // this wave's 1 KB of each half slab
  const unsigned woff = tid * 16u;
#define PN_STAGE(slab, buf)                                                                                   \
  do {                                                                                                        \
    pn_dma16(W3h, woff + (unsigned)(slab) * (PNH_SLAB_U4 * 16u), ring + (buf) * (PNH_SLAB_U4 * 16u) + my);                      \
    pn_dma16(W3h, woff + (unsigned)(slab) * (PNH_SLAB_U4 * 16u) + 4096u, ring + (buf) * (PNH_SLAB_U4 * 16u) + 4096u + my);      \
  } while (0)
  PN_STAGE(0, 0);
  PN_STAGE(1, 1);
  for (int e = tid; e < 8 * 2 * 2 * 64; e += PN_THREADS) s_w2[e] = W2h[e];
  for (int e = tid; e < PN_C1 * PN_MAXCIN; e += PN_THREADS) {
    int c = e / PN_MAXCIN, ci = e - c * PN_MAXCIN;
    s_w1[e] = ci < CIN ? W1[c * CIN + ci] : 0.f;
  }
  if (tid < PN_C1) s_b1[tid] = b1[tid];
  if (tid < PN_C2) { s_b2[tid] = b2[tid]; s_e2[tid] = ew2[tid]; }
  for (int e = tid; e < PN_C3; e += PN_THREADS) s_e3[e] = -ew3[e];
  pn_wait_vm<0>();
  __syncthreads();

  // running maxima of layer 3, TRANSPOSED product (points x channels): lane (j, q) holds channel 16 t3 + j of points 4 q + e,
  // so the max over the points is a max over registers; the four q meet once, after the last pass
  float mx[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) mx[t] = -FLT_MAX;
  // NARROW: the three 8 x 8 layers as fp32 MFMAs (16 x 16 x 4, exact products) chained in registers like the wide layers: a
  // lane's D registers e = channels 4 q + e of its point are the B operands of the next layer's steps e (contraction index
  // 4 kq + e <-> step e, lane group kq), so the weights are per-lane A operands (12 registers, loaded once) and nothing moves
  // between lanes; channels 8 .. 15 are zero rows.
  float na[3][4];
  pf32x4 nmx = pf32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  if constexpr (NARROW) {
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int e = 0; e < 4; ++e) na[l][e] = (j < PNN_W && q < 2) ? nw[72 * l + j * 8 + 4 * q + e] : 0.f;
  }
  unsigned cur = 0;                                          // the ring buffer slab t3 of this pass is in

  const float* xo = pts + obj * CIN * (long long)P;
  for (int p0 = 0; p0 < P; p0 += PN_PTS) {
    // ---- layer 1 on the VALU (as k_pointnet_feat): h1[pt][t1*4+e] = channel 16 t1 + 4 q + e of point (tile pt, j); a point
    // past the end repeats the object's last one (a duplicate does not move a maximum: no masks further down)
    float h1[2][16];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int pp = p0 + wave * 32 + pt * 16 + j, p = pp < P ? pp : P - 1;
      float x[PN_MAXCIN];
#pragma unroll
      for (int ci = 0; ci < PN_MAXCIN; ++ci) x[ci] = ci < CIN ? xo[(long long)ci * P + p] : 0.f;
      if constexpr (NARROW) {
        pf32x4 d = pf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {          // layer 1: B = the point's inputs 4 q + e (q < 2)
          const float xb = q == 0 ? x[e] : (q == 1 ? x[4 + e] : 0.f);
          d = __builtin_amdgcn_mfma_f32_16x16x4f32(na[0][e], xb, d, 0, 0, 0);
        }
#pragma unroll
        for (int l = 1; l < 3; ++l) {
          const pf32x4 bias = (q < 2) ? *reinterpret_cast<const pf32x4*>(nw + 72 * (l - 1) + 64 + 4 * q) : pf32x4{0.f, 0.f, 0.f, 0.f};
          pf32x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = fmaxf(d[e] + bias[e], 0.f);
          d = pf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) d = __builtin_amdgcn_mfma_f32_16x16x4f32(na[l][e], h[e], d, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) nmx[e] = fmaxf(nmx[e], d[e]);      // (the last layer's bias: added at the end)
      }
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
    // ---- h1 -> two fp16 pieces at the point's own power of two; k-step s = input tiles 2 s, 2 s + 1
    pf16x8 Xa[2][2], Xb[2][2];
    int ex1[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      float m = 0.f;
#pragma unroll
      for (int v = 0; v < 16; ++v) m = fmaxf(m, h1[pt][v]);           // ReLU outputs: no sign
      ex1[pt] = pn_exponent(pn_point_max(m));
      const float sc = __builtin_bit_cast(float, (unsigned)(ex1[pt] + 127) << 23);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          _Float16 a, b;
          pn_split2(h1[pt][(2 * s + (jj >> 2)) * 4 + (jj & 3)] * sc, a, b);
          Xa[pt][s][jj] = a;
          Xb[pt][s][jj] = b;
        }
    }
    // ---- layer 2 (channels x points): 8 output tiles x 2 k-steps x 3 products per point tile
    float h2[2][32];
#pragma unroll
    for (int t2 = 0; t2 < 8; ++t2) {
      pf32x4 acc0 = pf32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const pf16x8 Wa = __builtin_bit_cast(pf16x8, s_w2[((t2 * 2 + s) * 2 + 0) * 64 + lane]);
        const pf16x8 Wb = __builtin_bit_cast(pf16x8, s_w2[((t2 * 2 + s) * 2 + 1) * 64 + lane]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wb, Xa[0][s], acc0, 0, 0, 0);     // smallest first
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wa, Xb[0][s], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wa, Xa[0][s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wb, Xa[1][s], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wa, Xb[1][s], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wa, Xa[1][s], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 16 * t2 + 4 * q + e;
        const float bb = s_b2[c];
        const int ew = s_e2[c];
        h2[0][t2 * 4 + e] = fmaxf(ldexpf(acc0[e], -(ew + ex1[0])) + bb, 0.f);
        h2[1][t2 * 4 + e] = fmaxf(ldexpf(acc1[e], -(ew + ex1[1])) + bb, 0.f);
      }
    }
    // ---- h2 -> pieces (4 k-steps)
    pf16x8 Ya[2][4], Yb[2][4];
    int nex[2][4];                                   // MINUS the exponents of points 4 q + e of the two tiles
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      float m = 0.f;
#pragma unroll
      for (int v = 0; v < 32; ++v) m = fmaxf(m, h2[pt][v]);
      const int ex2 = pn_exponent(pn_point_max(m));
      const float sc = __builtin_bit_cast(float, (unsigned)(ex2 + 127) << 23);
#pragma unroll
      for (int e = 0; e < 4; ++e) nex[pt][e] = -__shfl(ex2, 4 * q + e, 64);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          _Float16 a, b;
          pn_split2(h2[pt][(2 * s + (jj >> 2)) * 4 + (jj & 3)] * sc, a, b);
          Ya[pt][s][jj] = a;
          Yb[pt][s][jj] = b;
        }
    }
    // ---- layer 3 (points x channels): 32 output tiles, W3's slabs through the ring
#pragma unroll
    for (int t3 = 0; t3 < 32; ++t3) {
      const unsigned nb = cur >= 1 ? cur - 1 : PNH_RING - 1;          // (cur + 2) mod 3: read last in the step before
      PN_STAGE((t3 + 2) & 31, nb);                                    // slabs 0, 1 of the next pass follow 30, 31
      const uint4* sw = s_w3 + cur * PNH_SLAB_U4;
      pf32x4 acc0 = pf32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      uint4 wf[8];                         // the tile's eight fragments in one go: one LDS round trip per tile, not four
#pragma unroll
      for (int f = 0; f < 8; ++f) wf[f] = sw[f * 64 + lane];
      const int nw = s_e3[16 * t3 + j];
      asm volatile("" ::: "memory");       // (keeps the reads above the products)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const pf16x8 Wa = __builtin_bit_cast(pf16x8, wf[s * 2 + 0]);
        const pf16x8 Wb = __builtin_bit_cast(pf16x8, wf[s * 2 + 1]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya[0][s], Wb, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Yb[0][s], Wa, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya[0][s], Wa, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya[1][s], Wb, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Yb[1][s], Wa, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya[1][s], Wa, acc1, 0, 0, 0);
      }
      float v = mx[t3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v = fmaxf(v, ldexpf(acc0[e], nw + nex[0][e]));
        v = fmaxf(v, ldexpf(acc1[e], nw + nex[1][e]));
      }
      mx[t3] = v;
      cur = cur == PNH_RING - 1 ? 0 : cur + 1;
      pn_wait_vm<2>();                   // slab t3 + 1 has landed (this thread's part); t3 + 2 may still fly
      __syncthreads();
    }
  }
  pn_wait_vm<0>();
#undef PN_STAGE
#pragma unroll
  for (int t3 = 0; t3 < 32; ++t3) {
    const float v = pn_point_max(mx[t3]);
    if (q == 0) s_max[wave * PN_C3 + 16 * t3 + j] = v;
  }
  __syncthreads();
  for (int c = tid; c < PN_C3; c += PN_THREADS) {
    float v = fmaxf(fmaxf(s_max[c], s_max[PN_C3 + c]), fmaxf(s_max[2 * PN_C3 + c], s_max[3 * PN_C3 + c]));
    out[obj * PN_C3 + c] = v + b3[c];
  }
  if constexpr (NARROW) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = nmx[e];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));       // the 16 points of the lane group
      if (j == 0 && q < 2) s_max[wave * PNN_W + 4 * q + e] = v;
    }
    __syncthreads();
    if (tid < PNN_W)
      nout[obj * PNN_W + tid] = fmaxf(fmaxf(s_max[tid], s_max[PNN_W + tid]), fmaxf(s_max[2 * PNN_W + tid], s_max[3 * PNN_W + tid])) +
                                nw[144 + 64 + tid];
  }
}

// ------------------------------------------------------------------------------------------------ f16 x 2 form, W3 in registers
// The same extractor with the roles of layer 3's operands swapped in the memory hierarchy: a block of EIGHT waves stays on a CU and walks
// over objects, wave w keeps the fragments of W3's channel tiles 4 w .. 4 w + 3 in registers for the whole launch (128 registers, loaded
// once: no slab ring, no barrier per channel tile) and the POINTS go through LDS in half-passes of 64 -- four waves run layers 1 and 2
// for 16 points each and leave their two fp16 planes (operand order) + their exponents in one of three LDS buffers, then ALL waves
// multiply the four point tiles against their own channels.  A fragment read (8 x 1 KB per point tile) feeds 4 x 12 = 48 MFMAs instead
// of 24: layer 3's LDS traffic halves (the form above shares its time between the matrix pipe and LDS reads, profiles/r06_cvae.md).
// The two halves of the block alternate as preparers (half-pass t by waves 4 (t & 1) .. + 3), and there is no barrier in the loop: a
// wave runs  prepare, multiply, multiply  and meets the others through two counts per wave in LDS (see the loop).  The preparing wave
// has its SIMD's vector issue to itself, so its own latencies show: the points arrive by LDS-DMA one turn ahead, layer 1 and the narrow
// extractor's first layer take them as fp32 MFMA operands as they lie (k-step i, lane (j, q) = input channel 4 i + q of point j),
// every LDS round trip fetches a whole channel tile's operands, and the wave runs at raised priority.  A wave owns its channels
// outright: nothing to combine across waves at an object's end.  Measured (tools/sampler_time.py, 4096 objects x 512 points):
// 0.65 ms against 0.75 ms for the form above; layer 3 alone is 0.45 ms of it, which is what the matrix pipe sustains on operands
// with random bits (tools/experiments/mfma_peak.hip: 0.69 of the nominal rate, profiles/r06_mfma_ceiling.md).
#define PNW_THREADS 512
#define PNW_NM_RING 4
#define PNW_BUFS 3
template <bool NARROW>
__global__ __launch_bounds__(PNW_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pointnet_feat_f16w(
    const float* __restrict__ pts, int B, int CIN, int P, const float* __restrict__ W1, const float* __restrict__ b1,
    const uint4* __restrict__ W2h, const int* __restrict__ ew2, const float* __restrict__ b2, const uint4* __restrict__ W3h,
    const int* __restrict__ ew3, const float* __restrict__ b3, float* __restrict__ out, const float* __restrict__ nw,
    float* __restrict__ nout) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* s_w2 = reinterpret_cast<uint4*>(smem);              // 8 t2 x 2 s x 2 planes x 64 lanes                          (32 KB)
  uint4* s_y = s_w2 + 8 * 2 * 2 * 64;                        // [buffer 3][point tile 4][k-step 4][plane 2][lane 64]      (96 KB)
  pf32x4* s_w1f = reinterpret_cast<pf32x4*>(s_y + PNW_BUFS * 4 * 4 * 2 * 64);      // [k-step 2][lane 64]: layer 1's weights as fp32 MFMA fragments
  float* s_b1 = reinterpret_cast<float*>(s_w1f + 2 * 64);    // 64
  float* s_b2 = s_b1 + PN_C1;                                // 128
  int* s_e2 = reinterpret_cast<int*>(s_b2 + PN_C2);          // 128
  int* s_ex = s_e2 + PN_C2;                                  // [buffer 3][64]: MINUS the exponents of the half-pass's points
  int* s_pa = s_ex + PNW_BUFS * 64;                          // [wave 8]: half-passes this wave has prepared ...
  int* s_pb = s_pa + 8;                                      // [wave 8]: ... and half-passes it has multiplied
  int* s_e3 = s_pb + 8;                                      // 512: MINUS the exponents of layer 3's rows ...
  float* s_b3 = reinterpret_cast<float*>(s_e3 + PN_C3);      // 512: ... and its biases (an object's end must not wait for HBM)
  float* s_x = s_b3 + PN_C3;         // [wave 8][2][64]: the wave's next 16 points, channel (4 i + q) of point j
  pf32x4* s_na = reinterpret_cast<pf32x4*>(s_x + 8 * 2 * 64);     // NARROW: [layer 3][lane 64] weight fragments, [layer 2][lane 64] biases
  float* s_nm = reinterpret_cast<float*>(s_na + 5 * 64);     // NARROW: [object ring 4][8 waves][8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4, grp = wave >> 2, wt = wave & 3;

  for (int e = tid; e < 8 * 2 * 2 * 64; e += PNW_THREADS) s_w2[e] = W2h[e];
  if (tid < 2 * 64) {
    const int i = tid >> 6, ci = 4 * i + q;
    pf32x4 v;
#pragma unroll
    for (int t1 = 0; t1 < 4; ++t1) v[t1] = ci < CIN ? W1[(16 * t1 + j) * CIN + ci] : 0.f;
    s_w1f[tid] = v;
  }
  if (tid < PN_C1) s_b1[tid] = b1[tid];
  if (tid < PN_C2) { s_b2[tid] = b2[tid]; s_e2[tid] = ew2[tid]; }
  if (tid < 16) s_pa[tid] = 0;
  for (int e = tid; e < PN_C3; e += PNW_THREADS) { s_e3[e] = -ew3[e]; s_b3[e] = b3[e]; }
  // this wave's channels: tiles 4 wave + a, fragments [a][k-step][plane] straight from the packed image
  pf16x8 Wa[4][4], Wb[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      Wa[a][s] = __builtin_bit_cast(pf16x8, W3h[(((4 * wave + a) * 4 + s) * 2 + 0) * 64 + lane]);
      Wb[a][s] = __builtin_bit_cast(pf16x8, W3h[(((4 * wave + a) * 4 + s) * 2 + 1) * 64 + lane]);
    }
  float mx[4];                                               // lane (j, q): channel 16 (4 wave + a) + j, in the channel's scale
#pragma unroll
  for (int a = 0; a < 4; ++a) mx[a] = -FLT_MAX;
  pf32x4 nmx = pf32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  if constexpr (NARROW) {                                    // the narrow extractor's weights as MFMA fragments, zero outside 8 x 8
    if (tid < 3 * 64) {      // layer 0: [k-step i] = input channel 4 i + q (the staged points' order); layers 1, 2: [e] = channel 4 q + e (D's)
      const int l = tid >> 6;
      pf32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = l == 0 ? 4 * e + q : 4 * q + e;
        v[e] = (j < PNN_W && col < PNN_W) ? nw[72 * l + j * 8 + col] : 0.f;
      }
      s_na[tid] = v;
    }
    if (tid < 2 * 64) {
      const int l = tid >> 6;
      s_na[3 * 64 + tid] = (q < 2) ? *reinterpret_cast<const pf32x4*>(nw + 72 * l + 64 + 4 * q) : pf32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();

  const int H = (P + 63) >> 6;                               // half-passes per object
  const int G = __builtin_amdgcn_readfirstlane((int)gridDim.x);
  const int nobj = ((int)blockIdx.x < B) ? (B - 1 - (int)blockIdx.x) / G + 1 : 0;
  const int T = nobj * H;                                    // this block's half-passes, objects one after the other

  // the 16 points this wave prepares in half-pass t, on their way into its corner of LDS (a point past the end repeats the object's
  // last one, a channel past CIN the last channel: its weights are zero)
  const unsigned my_x = __builtin_amdgcn_readfirstlane(pn_lds_addr(s_x + wave * 128));
  auto stage = [&](int k, int hp) {
    const long long obj = (long long)blockIdx.x + (long long)k * G;
    const float* xo = pts + obj * CIN * (long long)P;
    const int pp = hp * 64 + wt * 16 + j, p = pp < P ? pp : P - 1;
    const int c0 = q < CIN ? q : CIN - 1, c1 = 4 + q < CIN ? 4 + q : CIN - 1;
    const unsigned long long xa = (unsigned long long)xo;    // (uniform, but the compiler has to be told)
    xo = reinterpret_cast<const float*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xa >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)xa));
    pn_dma4(xo, (unsigned)(c0 * P + p) * 4u, my_x);
    pn_dma4(xo, (unsigned)(c1 * P + p) * 4u, my_x + 256u);
  };
  // layers 1 and 2 for this wave's 16 points of half-pass t
  auto produce = [&](int t, int k, int hp) {
    const int buf = t % PNW_BUFS;
    __builtin_amdgcn_s_setprio(3);                           // the block waits for this wave: ahead of the SIMD's other wave
    pn_wait_vm<0>();
    // the staged points are MFMA operands as they lie: k-step i of a 16 x 16 x 4 product wants, in lane (j, q), input channel 4 i + q of
    // point j.  Layer 1 (and the narrow extractor's first layer) on the matrix pipe in fp32: one or two steps per channel tile
    // instead of eight FMAs per channel on a wave that has the SIMD's vector issue to itself
    const int KS = CIN > 4 ? 2 : 1;
    float xb[2];
    xb[0] = s_x[wave * 128 + lane];
    xb[1] = s_x[wave * 128 + 64 + lane];
    if constexpr (NARROW) {
      if (hp < 2) nmx = pf32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};      // this wave's first half-pass of the object
      const pf32x4 n0 = s_na[lane];                          // [0], [1]: the first layer's two k-steps
      pf32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(n0[0], xb[0], pf32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      if (KS > 1) d = __builtin_amdgcn_mfma_f32_16x16x4f32(n0[1], xb[1], d, 0, 0, 0);
#pragma unroll
      for (int l = 1; l < 3; ++l) {
        const pf32x4 bias = s_na[(2 + l) * 64 + lane];
        const pf32x4 na = s_na[l * 64 + lane];
        pf32x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = fmaxf(d[e] + bias[e], 0.f);
        pf32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(na[0], h[0], pf32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);     // two chains
        pf32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(na[1], h[1], pf32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(na[2], h[2], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(na[3], h[3], d1, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = d0[e] + d1[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) nmx[e] = fmaxf(nmx[e], d[e]);
      if (hp + 2 >= H) {                                     // ... and its last: the wave's share of the narrow maxima
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = pn_row_max(nmx[e]);
          if (j == 0 && q < 2) s_nm[((k & (PNW_NM_RING - 1)) * 8 + wave) * PNN_W + 4 * q + e] = v;
        }
      }
    }
    float h1[16];
    {
      const pf32x4 w0 = s_w1f[lane], w1 = s_w1f[64 + lane];  // k-step i, lane (j, q): W1[16 t1 + j][4 i + q] for t1 = 0 .. 3
#pragma unroll
      for (int t1 = 0; t1 < 4; ++t1) {
        pf32x4 acc = *reinterpret_cast<const pf32x4*>(s_b1 + 16 * t1 + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[t1], xb[0], acc, 0, 0, 0);
        if (KS > 1) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[t1], xb[1], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) h1[t1 * 4 + e] = fmaxf(acc[e], 0.f);
      }
    }
    float m = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) m = fmaxf(m, h1[v]);
    const int ex1 = pn_exponent(pn_point_max(m));
    const float sc1 = __builtin_bit_cast(float, (unsigned)(ex1 + 127) << 23);
    pf16x8 Xa[2], Xb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        _Float16 a, b;
        pn_split2(h1[(2 * s + (jj >> 2)) * 4 + (jj & 3)] * sc1, a, b);
        Xa[s][jj] = a;
        Xb[s][jj] = b;
      }
    // layer 2, one LDS round trip per channel tile: its four fragments, exponents and biases are in flight together
    float h2[32];
#pragma unroll
    for (int t2 = 0; t2 < 8; ++t2) {
      uint4 V[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) V[s][pl] = s_w2[((t2 * 2 + s) * 2 + pl) * 64 + lane];
      const pi32x4 e2v = *reinterpret_cast<const pi32x4*>(s_e2 + 16 * t2 + 4 * q);
      const pf32x4 b2v = *reinterpret_cast<const pf32x4*>(s_b2 + 16 * t2 + 4 * q);
      __builtin_amdgcn_sched_barrier(0);
      pf32x4 ac[2];                                          // the two k-steps as independent chains (a wave is alone here)
#pragma unroll
      for (int s = 0; s < 2; ++s) ac[s] = pf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) ac[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pf16x8, V[s][1]), Xa[s], ac[s], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 2; ++s) ac[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pf16x8, V[s][0]), Xb[s], ac[s], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 2; ++s) ac[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pf16x8, V[s][0]), Xa[s], ac[s], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) h2[t2 * 4 + e] = fmaxf(ldexpf(ac[0][e] + ac[1][e], -(e2v[e] + ex1)) + b2v[e], 0.f);
    }
    m = 0.f;
#pragma unroll
    for (int v = 0; v < 32; ++v) m = fmaxf(m, h2[v]);
    const int ex2 = pn_exponent(pn_point_max(m));
    const float sc2 = __builtin_bit_cast(float, (unsigned)(ex2 + 127) << 23);
    if (q == 0) s_ex[buf * 64 + 16 * wt + j] = -ex2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      pf16x8 ya, yb;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        _Float16 a, b;
        pn_split2(h2[(2 * s + (jj >> 2)) * 4 + (jj & 3)] * sc2, a, b);
        ya[jj] = a;
        yb[jj] = b;
      }
      s_y[(((buf * 4 + wt) * 4 + s) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, ya);
      s_y[(((buf * 4 + wt) * 4 + s) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, yb);
    }
    if (t + 2 < T) {                                         // this wave's next turn
      int k2 = k, h2p = hp + 2;
      while (h2p >= H) { h2p -= H; ++k2; }
      stage(k2, h2p);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the planes and exponents are in LDS before the count says so
    if (lane == 0) __hip_atomic_store(s_pa + wave, (t >> 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_s_setprio(0);
  };
  // No barrier in the loop: a wave runs  prepare, multiply, multiply  over and over and meets the others through two counts per wave in
  // LDS -- half-pass t may be multiplied once its four preparers have counted it, buffer t % 3 may be overwritten once all eight waves
  // have counted the multiplication of half-pass t - 3.  The two waves of a SIMD are one multiplication apart, so one prepares (vector
  // and LDS work, a long dependent chain) under the other's MFMAs instead of the block waiting for the slowest preparer.
  auto count_min = [&](const int* c, int n) {
    int m = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int i = 1; i < n; ++i) {
      const int v = __hip_atomic_load(c + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      m = v < m ? v : m;
    }
    return __builtin_amdgcn_readfirstlane(m);
  };

  if (grp < T) stage(grp >= H ? 1 : 0, grp >= H ? 0 : grp);  // group 0 starts with half-pass 0, group 1 with 1
  if (T > 0 && grp == 0) produce(0, 0, 0);
  int kc = 0, hc = 0;                                        // half-pass t = object kc of this block, its half-pass hc
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T && ((t + 1) & 1) == grp) {
      if (t + 1 >= PNW_BUFS) {
        while (count_min(s_pb, 8) < t + 2 - PNW_BUFS) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      produce(t + 1, hc + 1 == H ? kc + 1 : kc, hc + 1 == H ? 0 : hc + 1);
    }
    while (count_min(s_pa + 4 * (t & 1), 4) < (t >> 1) + 1) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- layer 3 (points x channels): the half-pass's four point tiles against this wave's four channel tiles, as 16 steps of 12
    // MFMAs; a step's two fragments are read one step ahead (an LDS round trip is shorter than 12 MFMAs, and this wave may be the
    // only one multiplying on its SIMD), a tile's maxima are taken under the next tile's first step
    const int buf = t % PNW_BUFS;
    {
      const uint4* yl = s_y + buf * (4 * 4 * 2 * 64) + lane;
      uint4 Yn0 = yl[0], Yn1 = yl[64];
      pf32x4 acc[2][4];
      pi32x4 nex[2];
      auto maxima = [&](int h) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float v = mx[a];
#pragma unroll
          for (int e = 0; e < 4; ++e) v = fmaxf(v, ldexpf(acc[h][a][e], nex[h][e]));
          mx[a] = v;
        }
      };
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        nex[pt & 1] = *reinterpret_cast<const pi32x4*>(s_ex + buf * 64 + 16 * pt + 4 * q);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const pf16x8 Ya = __builtin_bit_cast(pf16x8, Yn0), Yb = __builtin_bit_cast(pf16x8, Yn1);
          if (pt * 4 + s + 1 < 16) {
            Yn0 = yl[((pt * 4 + s + 1) * 2 + 0) * 64];
            Yn1 = yl[((pt * 4 + s + 1) * 2 + 1) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int a = 0; a < 4; ++a)
            acc[pt & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya, Wb[a][s], s == 0 ? pf32x4{0.f, 0.f, 0.f, 0.f} : acc[pt & 1][a], 0, 0, 0);
#pragma unroll
          for (int a = 0; a < 4; ++a) acc[pt & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Yb, Wa[a][s], acc[pt & 1][a], 0, 0, 0);
#pragma unroll
          for (int a = 0; a < 4; ++a) acc[pt & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya, Wa[a][s], acc[pt & 1][a], 0, 0, 0);
          if (s == 0 && pt > 0) maxima((pt - 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      maxima(1);
    }
    const int k = kc;
    const bool last = hc == H - 1;                           // the object's last half-pass: its features leave
    if (++hc == H) { hc = 0; ++kc; }
    if (last) {
      const long long obj = (long long)blockIdx.x + (long long)k * G;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float v = pn_point_max(mx[a]);
        const int c = 16 * (4 * wave + a) + j;
        if (q == 0) out[obj * PN_C3 + c] = ldexpf(v, s_e3[c]) + s_b3[c];
        mx[a] = -FLT_MAX;
      }
    }
    if constexpr (NARROW) {                                  // (every preparer of the object has counted its last half-pass by now)
      if (last && tid < PNN_W) {
        const long long obj = (long long)blockIdx.x + (long long)k * G;
        float v = -FLT_MAX;
#pragma unroll
        for (int w_ = 0; w_ < 8; ++w_)
          if (H >= 2 || ((k * H) & 1) == (w_ >> 2)) v = fmaxf(v, s_nm[((k & (PNW_NM_RING - 1)) * 8 + w_) * PNN_W + tid]);
        nout[obj * PNN_W + tid] = v + nw[144 + 64 + tid];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the reads of the buffer are behind this wave)
    if (lane == 0) __hip_atomic_store(s_pb + wave, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

static int glx_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}
static size_t pointnet_feat_f16w_lds_bytes() {
  return (size_t)(8 * 2 * 2 * 64 + PNW_BUFS * 4 * 4 * 2 * 64) * 16 +
         (size_t)(PN_C1 * PN_MAXCIN + PN_C1 + PN_C2 + PN_C2 + PNW_BUFS * 64 + 16 + 2 * PN_C3 + 8 * 2 * 64 + 5 * 64 * 4 + PNW_NM_RING * 8 * PNN_W) * 4;
}

extern "C" size_t glx_pointnet_feat_f16x2_lds_bytes(void) {
  return (size_t)(8 * 2 * 2 * 64 + PNH_RING * PNH_SLAB_U4) * 16 +
         (size_t)(PN_C1 * PN_MAXCIN + PN_C1 + PN_C2 + PN_C2 + PN_C3 + 4 * PN_C3) * 4;
}

// W2h / W3h: the folded (128, 64) / (512, 128) weights as two fp16 planes of w 2^ew[row] in the kernel's operand order
// ([output tile][k-step][plane][lane 16 q + m][slot 4 h + e] = W[16 tile + m][32 s + 16 h + 4 q + e]); ew2 / ew3: the rows'
// exponents (max |w| 2^ew in [2^14, 2^15), 0 for a zero row).
static int g_pointnet_w_stationary = 1;      // 1: k_pointnet_feat_f16w (W3 in registers, points through LDS); 0: k_pointnet_feat_f16
extern "C" int glx_pointnet_feat_set_form(int w_stationary) {      // returns the previous setting (measurements)
  const int old = g_pointnet_w_stationary;
  g_pointnet_w_stationary = w_stationary ? 1 : 0;
  return old;
}
extern "C" int glx_pointnet_feat_f16x2_pair(const float* points, int B, int Cin, int P, const float* W1, const float* b1,
                                            const void* W2h, const int32_t* ew2, const float* b2, const void* W3h,
                                            const int32_t* ew3, const float* b3, float* out, const float* narrow, float* narrow_out,
                                            void* stream);
extern "C" int glx_pointnet_feat_f16x2(const float* points, int B, int Cin, int P, const float* W1, const float* b1,
                                       const void* W2h, const int32_t* ew2, const float* b2, const void* W3h, const int32_t* ew3,
                                       const float* b3, float* out, void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(points && W1 && b1 && W2h && ew2 && b2 && W3h && ew3 && b3 && out, "glx_pointnet_feat_f16x2: null pointer");
  GLX_REQUIRE(Cin >= 1 && Cin <= PN_MAXCIN && P >= 1, "glx_pointnet_feat_f16x2: Cin must be 1..8, P >= 1");
  return glx_pointnet_feat_f16x2_pair(points, B, Cin, P, W1, b1, W2h, ew2, b2, W3h, ew3, b3, out, nullptr, nullptr, stream);
}

// ... and the narrow extractor (widths 8, 8, 8) of the same points in the same launch: narrow = 216 floats (W1 (8 x 8: input
// channels zero-padded to 8), b1 (8), W2 (8 x 8), b2, W3 (8 x 8), b3; eval-mode BatchNorm folded, no ReLU behind the last layer),
// narrow_out (B, 8).  narrow == NULL: glx_pointnet_feat_f16x2.
extern "C" int glx_pointnet_feat_f16x2_pair(const float* points, int B, int Cin, int P, const float* W1, const float* b1,
                                            const void* W2h, const int32_t* ew2, const float* b2, const void* W3h,
                                            const int32_t* ew3, const float* b3, float* out, const float* narrow, float* narrow_out,
                                            void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(points && W1 && b1 && W2h && ew2 && b2 && W3h && ew3 && b3 && out, "glx_pointnet_feat_f16x2: null pointer");
  GLX_REQUIRE(Cin >= 1 && Cin <= PN_MAXCIN && P >= 1, "glx_pointnet_feat_f16x2: Cin must be 1..8, P >= 1");
  GLX_REQUIRE(!narrow || narrow_out, "glx_pointnet_feat_f16x2_pair: narrow weights without an output");
  const size_t lds = glx_pointnet_feat_f16x2_lds_bytes();
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_pointnet_feat_f16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GLX_HIP(hipFuncSetAttribute((const void*)k_pointnet_feat_f16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  if (g_pointnet_w_stationary) {
    const size_t ldw = pointnet_feat_f16w_lds_bytes();
    static bool attr_w = false;
    if (!attr_w) {
      GLX_HIP(hipFuncSetAttribute((const void*)k_pointnet_feat_f16w<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldw));
      GLX_HIP(hipFuncSetAttribute((const void*)k_pointnet_feat_f16w<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldw));
      attr_w = true;
    }
    const int gw = B < glx_num_cus() ? B : glx_num_cus();          // one block per CU (LDS), each walks over its objects
    if (narrow)
      hipLaunchKernelGGL(k_pointnet_feat_f16w<true>, dim3(gw), dim3(PNW_THREADS), ldw, (hipStream_t)stream, points, B, Cin, P, W1, b1,
                         (const uint4*)W2h, ew2, b2, (const uint4*)W3h, ew3, b3, out, narrow, narrow_out);
    else
      hipLaunchKernelGGL(k_pointnet_feat_f16w<false>, dim3(gw), dim3(PNW_THREADS), ldw, (hipStream_t)stream, points, B, Cin, P, W1, b1,
                         (const uint4*)W2h, ew2, b2, (const uint4*)W3h, ew3, b3, out, (const float*)nullptr, (float*)nullptr);
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  if (narrow)
    hipLaunchKernelGGL(k_pointnet_feat_f16<true>, dim3(B), dim3(PN_THREADS), lds, (hipStream_t)stream, points, Cin, P, W1, b1,
                       (const uint4*)W2h, ew2, b2, (const uint4*)W3h, ew3, b3, out, narrow, narrow_out);
  else
    hipLaunchKernelGGL(k_pointnet_feat_f16<false>, dim3(B), dim3(PN_THREADS), lds, (hipStream_t)stream, points, Cin, P, W1, b1,
                       (const uint4*)W2h, ew2, b2, (const uint4*)W3h, ew3, b3, out, (const float*)nullptr, (float*)nullptr);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ the sampler's tail
// Everything of Generator.forward's eval branch (cvae_uncertainty/model.py:245-265) behind the two extractors, one launch:
//   (mu, logvar) = fc1 / fc2 of the prior encoder on the 512 features (model.py:33-50), z = eps exp(logvar / 2) + mu (:194-198),
//   the decoder's fc1 -> bn1 -> ReLU -> fc2 -> bn2 -> ReLU on cat(narrow features, z), its four heads (64 -> 64 -> 3 / 3 / 1 / bins,
//   model.py:82-142) and the heading taken out of its bin (:257-264).
// ~35 library / elementwise launches of ~5 us before.  A wave takes CT_G objects (every weight it fetches feeds CT_G sums), a lane
// is an output channel of the 64-wide layers; activations pass between layers through a wave-private LDS patch.  Weights: `w`, one
// float buffer (dense_path.CVAE._tail_pack): WL (16 x 512: fc1 | fc2 rows), bL (16), W1T (16 x 64: [input][output], BatchNorm
// folded), b1 (64), W2T (64 x 64), b2 (64), WhT (4 x 64 x 64: [head][input][output]), bh (4 x 64), Wo ((7 + bins) x 64).
#define CT_G 4
#define CT_WAVES 4
#define CT_KMAX 16
__global__ __launch_bounds__(64 * CT_WAVES) void k_cvae_tail(const float* __restrict__ f512, const float* __restrict__ f8,
                                                             const float* __restrict__ eps, const float* __restrict__ w, int B,
                                                             int bins, float dir_offset, float dir_limit, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float s_f[CT_WAVES][CT_G][512];
  __shared__ __attribute__((aligned(16))) float s_a[CT_WAVES][CT_G][64], s_b[CT_WAVES][CT_G][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int obj0 = (blockIdx.x * CT_WAVES + wave) * CT_G;
  if (obj0 >= B) return;
  const float* WL = w;
  const float* bL = WL + 16 * 512;
  const float* W1T = bL + 16;
  const float* b1 = W1T + 16 * 64;
  const float* W2T = b1 + 64;
  const float* b2 = W2T + 64 * 64;
  const float* WhT = b2 + 64;
  const float* bh = WhT + 4 * 64 * 64;
  const float* Wo = bh + 4 * 64;
  const int kout = 7 + bins;
#define CT_SYNC()                                            \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     \
  __builtin_amdgcn_wave_barrier();                           \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront")
  int ob[CT_G];
#pragma unroll
  for (int g = 0; g < CT_G; ++g) ob[g] = obj0 + g < B ? obj0 + g : B - 1;
#pragma unroll
  for (int g = 0; g < CT_G; ++g)
#pragma unroll
    for (int e = 0; e < 2; ++e)
      reinterpret_cast<pf32x4*>(s_f[wave][g])[lane + 64 * e] = reinterpret_cast<const pf32x4*>(f512 + (long long)ob[g] * 512)[lane + 64 * e];
  CT_SYNC();
  // ---- the latent Gaussian: lane (k, c) sums inputs 128 c .. of output k (mu: k < 8, logvar: k >= 8)
  const int k = lane & 15, c = lane >> 4;
  float lat[CT_G];
#pragma unroll
  for (int g = 0; g < CT_G; ++g) lat[g] = 0.f;
  {
    const pf32x4* wr = reinterpret_cast<const pf32x4*>(WL + k * 512 + c * 128);
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
      const pf32x4 wv = wr[i];
#pragma unroll
      for (int g = 0; g < CT_G; ++g) {
        const pf32x4 fv = *reinterpret_cast<const pf32x4*>(&s_f[wave][g][c * 128 + 4 * i]);
        lat[g] = fmaf(wv[0], fv[0], fmaf(wv[1], fv[1], fmaf(wv[2], fv[2], fmaf(wv[3], fv[3], lat[g]))));
      }
    }
  }
#pragma unroll
  for (int g = 0; g < CT_G; ++g) {
    lat[g] += __shfl_xor(lat[g], 16, 64);
    lat[g] += __shfl_xor(lat[g], 32, 64);
    lat[g] += bL[k];
    const float lv = __shfl(lat[g], (lane & 7) + 8, 64);
    if (lane < 8) {
      s_a[wave][g][lane] = f8[(long long)ob[g] * 8 + lane];
      s_a[wave][g][8 + lane] = eps[(long long)ob[g] * 8 + lane] * expf(0.5f * lv) + lat[g];
    }
  }
  CT_SYNC();
  // ---- decoder trunk: lane = output channel
  float a[CT_G];
#pragma unroll
  for (int g = 0; g < CT_G; ++g) a[g] = b1[lane];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float wv = W1T[i * 64 + lane];
#pragma unroll
    for (int g = 0; g < CT_G; ++g) a[g] = fmaf(wv, s_a[wave][g][i], a[g]);
  }
#pragma unroll
  for (int g = 0; g < CT_G; ++g) s_b[wave][g][lane] = fmaxf(a[g], 0.f);
  CT_SYNC();
#pragma unroll
  for (int g = 0; g < CT_G; ++g) a[g] = b2[lane];
#pragma unroll 8
  for (int i = 0; i < 64; ++i) {
    const float wv = W2T[i * 64 + lane];
#pragma unroll
    for (int g = 0; g < CT_G; ++g) a[g] = fmaf(wv, s_b[wave][g][i], a[g]);
  }
  CT_SYNC();                // (everyone has read s_a's 16 inputs)
#pragma unroll
  for (int g = 0; g < CT_G; ++g) s_a[wave][g][lane] = fmaxf(a[g], 0.f);
  CT_SYNC();
  // ---- the heads' hidden layers
  float hid[4][CT_G];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < CT_G; ++g) hid[t][g] = bh[t * 64 + lane];
#pragma unroll 4
  for (int i = 0; i < 64; ++i) {
    float hv[CT_G];
#pragma unroll
    for (int g = 0; g < CT_G; ++g) hv[g] = s_a[wave][g][i];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float wv = WhT[(t * 64 + i) * 64 + lane];
#pragma unroll
      for (int g = 0; g < CT_G; ++g) hid[t][g] = fmaf(wv, hv[g], hid[t][g]);
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < CT_G; ++g) hid[t][g] = fmaxf(hid[t][g], 0.f);
  // ---- their output layers (no bias): centre 0..2 <- head 0, size 3..5 <- head 1, heading residual 6 <- head 2, bins <- head 3
  float pred[CT_KMAX][CT_G];
#pragma unroll
  for (int o = 0; o < CT_KMAX; ++o) {
    if (o < kout) {
      const int t = o < 3 ? 0 : (o < 6 ? 1 : (o < 7 ? 2 : 3));
      const float wv = Wo[o * 64 + lane];
#pragma unroll
      for (int g = 0; g < CT_G; ++g) {
        float v = wv * (t == 0 ? hid[0][g] : (t == 1 ? hid[1][g] : (t == 2 ? hid[2][g] : hid[3][g])));
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
        pred[o][g] = v;
      }
    }
  }
  // ---- the heading out of its bin; lane g writes object g
  const float period = 6.283185307179586f / (float)bins, inv_period = 1.f / period;
#pragma unroll
  for (int g = 0; g < CT_G; ++g) {
    if (lane == g && obj0 + g < B) {
      int label = 0;
      float best = pred[7][g];
#pragma unroll
      for (int o = 8; o < CT_KMAX; ++o)
        if (o < kout && pred[o][g] > best) { best = pred[o][g]; label = o - 7; }
      const float val = pred[6][g] - dir_offset;
      const float rot = val - floorf(val * inv_period + dir_limit) * period;
      float* dst = out + (long long)(obj0 + g) * kout;
#pragma unroll
      for (int o = 0; o < CT_KMAX; ++o)
        if (o < kout) dst[o] = o == 6 ? rot + dir_offset + period * (float)label : pred[o][g];
    }
  }
#undef CT_SYNC
}

// f512 (B, 512), f8 (B, 8): the two extractors' features; eps (B, 8); w: see above; out (B, 7 + bins), 1 <= bins <= 9.
extern "C" int glx_cvae_sample_tail(const float* f512, const float* f8, const float* eps, const float* w, int B, int bins,
                                    float dir_offset, float dir_limit_offset, float* out, void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(f512 && f8 && eps && w && out, "glx_cvae_sample_tail: null pointer");
  GLX_REQUIRE(bins >= 1 && 7 + bins <= CT_KMAX, "glx_cvae_sample_tail: 1 <= bins <= 9");
  hipLaunchKernelGGL(k_cvae_tail, dim3(glx_divup(B, CT_WAVES * CT_G)), dim3(64 * CT_WAVES), 0, (hipStream_t)stream, f512, f8, eps, w, B,
                     bins, dir_offset, dir_limit_offset, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ the training losses, two launches
// Generator.get_training_loss's two data terms (cvae_uncertainty/model.py:296-345 reg_loss; :205-212 + torch.distributions' KL of two
// diagonal Gaussians with scale = exp(logvar) + 3e-22, model.py:49,77) with their gradients, instead of ~160 elementwise launches of
// ~4.5 us in the recorded step:
//   loss_loc = sum_b sum_k smoothL1_beta(cw_k (p'_bk - t'_bk)) / B * loc_weight   (k < 7; heading as sin(p) cos(t) vs cos(p) sin(t);
//              a NaN target contributes nothing),
//   loss_dir = 2 sum_b CE(logits_b, bin(t_b6)) * dir_weight    (the factor 2: see dense_path.cvae_reg_loss),
//   latent   = mean_b sum_j KL(N(mu1, s1) || N(mu2, s2))_bj * latent_weight.
// One block of 1024 threads walks the objects in a fixed order (B is a few thousand): bitwise reproducible sums.
struct CvaeLossArgs {
  const float* pred;      // (B, 7 + bins)
  const float* labels;    // (B, 7)
  const float* cw;        // 7 code weights
  const float* mu1; const float* lv1; const float* mu2; const float* lv2;     // (B, L): posterior, prior
  float* d_pred; float* d_mu1; float* d_lv1; float* d_mu2; float* d_lv2;      // gradients of the WEIGHTED terms
  float* out;             // loss_loc, loss_dir, latent
  int B, bins, L;
  float beta, loc_weight, dir_weight, latent_weight, dir_offset;
};

__global__ __launch_bounds__(1024) void k_cvae_losses(CvaeLossArgs a) {
  __shared__ double s_r[3][1024];
  const int tid = threadIdx.x, K = 7 + a.bins;
  double loc = 0, dir = 0, lat = 0;
  const float period = 6.283185307179586f / (float)a.bins;
  for (int b = tid; b < a.B; b += 1024) {
    const float* p = a.pred + (long long)b * K;
    const float* t = a.labels + (long long)b * 7;
    float* dp = a.d_pred + (long long)b * K;
    // ---- location: smooth L1 on the code-weighted differences
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      float pv = p[k], tv = t[k], dd = 1.f;
      if (k == 6) {
        const float sp = sinf(p[6]), cp = cosf(p[6]), st = sinf(t[6]), ct = cosf(t[6]);
        pv = sp * ct;
        tv = cp * st;
        dd = cp * ct + sp * st;                           // d (p' - t') / d p6
      }
      const bool nan_t = tv != tv;
      const float x = nan_t ? 0.f : (pv - tv) * a.cw[k];
      const float n = fabsf(x);
      loc += (double)(n < a.beta ? 0.5f * n * n / a.beta : n - 0.5f * a.beta);
      const float dx = n < a.beta ? x / a.beta : (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f));
      dp[k] = nan_t ? 0.f : dx * a.cw[k] * dd * (a.loc_weight / (float)a.B);
    }
    // ---- direction: cross-entropy against the bin of the label heading
    {
      const float v = t[6] - a.dir_offset;
      const float rot = v - floorf(v * (1.f / 6.283185307179586f)) * 6.283185307179586f;     // limit_period(v, 0, 2 pi)
      int bin = (int)floorf(rot * (1.f / period));
      bin = bin < 0 ? 0 : (bin > a.bins - 1 ? a.bins - 1 : bin);
      float mx = p[7];
      for (int k = 1; k < a.bins; ++k) mx = fmaxf(mx, p[7 + k]);
      float se = 0.f;
      for (int k = 0; k < a.bins; ++k) se += expf(p[7 + k] - mx);
      const float lse = mx + logf(se);
      dir += (double)(lse - p[7 + bin]);
      for (int k = 0; k < a.bins; ++k) dp[7 + k] = (expf(p[7 + k] - lse) - (k == bin ? 1.f : 0.f)) * (2.f * a.dir_weight);
    }
    // ---- KL(posterior || prior), scale = exp(logvar) + 3e-22
    for (int jj = 0; jj < a.L; ++jj) {
      const long long e = (long long)b * a.L + jj;
      const float e1 = expf(a.lv1[e]), e2 = expf(a.lv2[e]);
      const float s1 = e1 + 3e-22f, s2 = e2 + 3e-22f, dm = a.mu1[e] - a.mu2[e];
      const float vr = (s1 / s2) * (s1 / s2), t1 = (dm / s2) * (dm / s2);
      lat += (double)(0.5f * (vr + t1 - 1.f - logf(vr)));
      const float wgt = a.latent_weight / (float)a.B;
      a.d_mu1[e] = dm / (s2 * s2) * wgt;
      a.d_mu2[e] = -dm / (s2 * s2) * wgt;
      a.d_lv1[e] = (s1 / (s2 * s2) - 1.f / s1) * e1 * wgt;
      a.d_lv2[e] = (-(s1 * s1 + dm * dm) / (s2 * s2 * s2) + 1.f / s2) * e2 * wgt;
    }
  }
  s_r[0][tid] = loc; s_r[1][tid] = dir; s_r[2][tid] = lat;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) { s_r[0][tid] += s_r[0][tid + o]; s_r[1][tid] += s_r[1][tid + o]; s_r[2][tid] += s_r[2][tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    a.out[0] = (float)(s_r[0][0] / (double)a.B * (double)a.loc_weight);
    a.out[1] = (float)(2.0 * s_r[1][0] * (double)a.dir_weight);
    a.out[2] = (float)(s_r[2][0] / (double)a.B * (double)a.latent_weight);
  }
}

// pred (B, 7 + bins), labels (B, 7), code_weights (7); mu / logvar of the posterior (1) and the prior (2), (B, L) each.
// out (3): loss_loc, loss_dir, latent (each with its weight applied); d_*: the gradients of those weighted terms (d_pred holds both
// location and direction terms' columns).  1 <= bins <= 9.
extern "C" int glx_cvae_losses(const float* pred, const float* labels, const float* code_weights, int B, int bins, float beta,
                               float loc_weight, float dir_weight, float dir_offset, const float* mu1, const float* logvar1,
                               const float* mu2, const float* logvar2, int L, float latent_weight, float* out, float* d_pred,
                               float* d_mu1, float* d_logvar1, float* d_mu2, float* d_logvar2, void* stream) {
  GLX_REQUIRE(pred && labels && code_weights && mu1 && logvar1 && mu2 && logvar2 && out && d_pred && d_mu1 && d_logvar1 && d_mu2 && d_logvar2,
              "glx_cvae_losses: null pointer");
  GLX_REQUIRE(B > 0 && bins >= 1 && bins <= 9 && L >= 1, "glx_cvae_losses: B > 0, 1 <= bins <= 9, L >= 1");
  CvaeLossArgs a{pred, labels, code_weights, mu1, logvar1, mu2, logvar2, d_pred, d_mu1, d_logvar1, d_mu2, d_logvar2, out, B, bins, L,
                 beta, loc_weight, dir_weight, latent_weight, dir_offset};
  hipLaunchKernelGGL(k_cvae_losses, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
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

// ------------------------------------------------------------------------------------------------ training twin, layer 3
// Training-mode  out[b, :] = max_p BatchNorm3(W3 h2[b, p, :] + b3)  of PointNetfeat (point_net.py:22-28: conv3 + bn3, no
// ReLU, max over the points) WITHOUT the (B x P x 512) tensor: 4.3 GB at BASELINE configs[3]'s 4096 x 512 points, which the
// module-by-module path writes once and re-reads eight times per step.  BatchNorm3 is a per-channel affine map of the
// conv output y, so  max_p bn3(y) = scale * (scale >= 0 ? max_p y : min_p y) + shift:  one pass over h2 that keeps, per
// (object, channel), max / min of y with the points they occur at and the sums of y and y^2 (the batch statistics),
// is the whole forward.  Backward (glx_pointmax_scatter / glx_pointmax_wsum + 128 x 128 algebra on the host side,
// dense_path.PointMaxBN): with  dy[r, c] = a_c g^[r, c] - b_c - c_c y[r, c]  (g^ = the gradient at the extreme point of its
// (object, channel), zero elsewhere; a, b, c from the BatchNorm backward's two sums, which only involve the extreme
// values) and y = h2 W3^T,
//     dh2 = dy W3   = (rows of W3 scattered to the extreme points) - v - h2 (W3^T diag(c) W3),
//     dW3 = dy^T h2 = a * (g-weighted sums of the extreme points' h2 rows) - b (x) sum_r h2 - diag(c) W3 (h2^T h2),
// i.e. the 512-wide layer's backward costs two 128 x 128 products per row instead of two 128 x 512 ones.
// Kernel layout = layer 3 of k_pointnet_feat: one block (4 waves) per object, 128 points per pass, v_mfma_f32_16x16x4_f32
// with the point tile as the B operand (read from h2 in global memory: lane (j, q) holds channels 16 t2 + 4 q + e of point
// j), W3 streamed through two 8 KB LDS slabs in fragment order.
#define PM_STAT 6                      // running per-wave arrays in LDS: max, min, argmax, argmin, sum, sum of squares

__global__ __launch_bounds__(PN_THREADS) void k_pointmax_fwd(
    const float* __restrict__ h2, int P, const float* __restrict__ W3p, float* __restrict__ vmax, float* __restrict__ vmin,
    int* __restrict__ amax, int* __restrict__ amin, float* __restrict__ s1, float* __restrict__ s2) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w3 = smem;                                  // 2 slabs
  float* s_st = s_w3 + 2 * PN_SLAB;                    // PM_STAT x 4 waves x 512
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const long long obj = blockIdx.x;
  float* w_max = s_st + (0 * 4 + wave) * PN_C3;
  float* w_min = s_st + (1 * 4 + wave) * PN_C3;
  int* w_amax = reinterpret_cast<int*>(s_st + (2 * 4 + wave) * PN_C3);
  int* w_amin = reinterpret_cast<int*>(s_st + (3 * 4 + wave) * PN_C3);
  float* w_s1 = s_st + (4 * 4 + wave) * PN_C3;
  float* w_s2 = s_st + (5 * 4 + wave) * PN_C3;
  for (int e = tid; e < 4 * PN_C3; e += PN_THREADS) {
    s_st[e] = -FLT_MAX;
    s_st[4 * PN_C3 + e] = FLT_MAX;
    reinterpret_cast<int*>(s_st)[8 * PN_C3 + e] = 0;
    reinterpret_cast<int*>(s_st)[12 * PN_C3 + e] = 0;
    s_st[16 * PN_C3 + e] = 0.f;
    s_st[20 * PN_C3 + e] = 0.f;
  }
  pf32x4 slab[2];
  slab[0] = reinterpret_cast<const pf32x4*>(W3p)[tid];
  slab[1] = reinterpret_cast<const pf32x4*>(W3p)[tid + PN_THREADS];
  reinterpret_cast<pf32x4*>(s_w3)[tid] = slab[0];
  reinterpret_cast<pf32x4*>(s_w3)[tid + PN_THREADS] = slab[1];
  __syncthreads();

  const float* ho = h2 + obj * (long long)P * PN_C2;
  for (int p0 = 0; p0 < P; p0 += PN_PTS) {
    float hreg[2][32];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int p = p0 + wave * 32 + pt * 16 + j;
      const bool live = p < P;
      const float* row = ho + (long long)(live ? p : 0) * PN_C2 + 4 * q;
#pragma unroll
      for (int t2 = 0; t2 < 8; ++t2) {
        const pf32x4 v = *reinterpret_cast<const pf32x4*>(row + 16 * t2);
#pragma unroll
        for (int e = 0; e < 4; ++e) hreg[pt][t2 * 4 + e] = live ? v[e] : 0.f;
      }
    }
    for (int t3 = 0; t3 < 32; ++t3) {
      const int nxt = (t3 + 1) & 31;
      const pf32x4* src = reinterpret_cast<const pf32x4*>(W3p + (size_t)nxt * PN_SLAB);
      slab[0] = src[tid];
      slab[1] = src[tid + PN_THREADS];
      const float* sw = s_w3 + (t3 & 1) * PN_SLAB;
      // operands swapped against k_pointnet_feat (point tile = A, weights = B): D comes out TRANSPOSED, rows = the 16 points
      // of the tile (4 q + e), columns = channels 16 t3 + j -- the reduction over points is then mostly inside a lane
      // (8 values: 2 tiles x 4 rows) and two exchanges across the lane's quad group, instead of four 16-lane butterflies
      // per channel row (profiles/r04_cvae.md)
      pf32x4 acc0 = pf32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int t2 = 0; t2 < 8; ++t2) {
        const pf32x4 a = *reinterpret_cast<const pf32x4*>(sw + (t2 * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hreg[0][t2 * 4 + e], a[e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hreg[1][t2 * 4 + e], a[e], acc1, 0, 0, 0);
        }
      }
      {
        // this lane: channel 16 t3 + j, points pbase + 4 q + e (tile 0) and + 16 (tile 1), ascending in (tile, e)
        const int pb = p0 + wave * 32 + 4 * q;
        float mx = -FLT_MAX, mn = FLT_MAX, a1 = 0.f, a2 = 0.f;
        int ix = 0, in_ = 0;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int p = pb + 16 * pt + e;
            const float v = pt ? acc1[e] : acc0[e];
            if (p < P) {
              if (v > mx) { mx = v; ix = p; }
              if (v < mn) { mn = v; in_ = p; }
              a1 += v;
              a2 += v * v;
            }
          }
        }
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {            // the four lanes (q) of a channel
          const float omx = __shfl_xor(mx, o, 64), omn = __shfl_xor(mn, o, 64);
          const int oix = __shfl_xor(ix, o, 64), oin = __shfl_xor(in_, o, 64);
          if (omx > mx || (omx == mx && oix < ix)) { mx = omx; ix = oix; }
          if (omn < mn || (omn == mn && oin < in_)) { mn = omn; in_ = oin; }
          a1 += __shfl_xor(a1, o, 64);
          a2 += __shfl_xor(a2, o, 64);
        }
        if (q == 0) {
          const int c = 16 * t3 + j;
          if (mx > w_max[c]) { w_max[c] = mx; w_amax[c] = ix; }      // earlier passes hold lower point indices: ties keep them
          if (mn < w_min[c]) { w_min[c] = mn; w_amin[c] = in_; }
          w_s1[c] += a1;
          w_s2[c] += a2;
        }
      }
      float* dw = s_w3 + ((t3 + 1) & 1) * PN_SLAB;
      reinterpret_cast<pf32x4*>(dw)[tid] = slab[0];
      reinterpret_cast<pf32x4*>(dw)[tid + PN_THREADS] = slab[1];
      __syncthreads();
    }
  }
  // ---- the 4 waves' values (a wave holds points 32 w .. 32 w + 31 of every pass: ties go to the lower index)
  for (int c = tid; c < PN_C3; c += PN_THREADS) {
    float mx = -FLT_MAX, mn = FLT_MAX, a1 = 0.f, a2 = 0.f;
    int ix = 0, in_ = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float vmx = s_st[(0 * 4 + w) * PN_C3 + c], vmn = s_st[(1 * 4 + w) * PN_C3 + c];
      const int vix = reinterpret_cast<const int*>(s_st)[(2 * 4 + w) * PN_C3 + c];
      const int vin = reinterpret_cast<const int*>(s_st)[(3 * 4 + w) * PN_C3 + c];
      if (vmx > mx || (vmx == mx && vix < ix)) { mx = vmx; ix = vix; }
      if (vmn < mn || (vmn == mn && vin < in_)) { mn = vmn; in_ = vin; }
      a1 += s_st[(4 * 4 + w) * PN_C3 + c];
      a2 += s_st[(5 * 4 + w) * PN_C3 + c];
    }
    const long long o = obj * PN_C3 + c;
    vmax[o] = mx; vmin[o] = mn; amax[o] = ix; amin[o] = in_; s1[o] = a1; s2[o] = a2;
  }
}

extern "C" int glx_pointmax_forward(const float* h2, int B, int P, const float* W3p, float* vmax, float* vmin,
                                    int32_t* amax, int32_t* amin, float* s1, float* s2, void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(h2 && W3p && vmax && vmin && amax && amin && s1 && s2, "glx_pointmax_forward: null pointer");
  GLX_REQUIRE(P >= 1, "glx_pointmax_forward: P >= 1");
  const size_t lds = (size_t)(2 * PN_SLAB + PM_STAT * 4 * PN_C3) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_pointmax_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_pointmax_fwd, dim3(B), dim3(PN_THREADS), lds, (hipStream_t)stream, h2, P, W3p, vmax, vmin,
                     (int*)amax, (int*)amin, s1, s2);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// The same pass with f16 x 2 products (the arithmetic and the structure of k_pointnet_feat_f16's layer 3: a point's own power of
// two, the weight rows' powers of two from the host, three fp16 MFMAs per product tile, W3's slabs by DMA through a ring of
// three, the product transposed): a lane keeps the running maximum and the point it occurs at of ITS channel and its eight
// points per pass in registers for all 32 channel tiles; the four lanes of a channel and the four waves meet once, at the end.
// ONE extreme per channel: which of max / min the BatchNorm needs is the sign of its weight, known before the launch, so the
// caller hands in the rows of W3 with that sign and gets max_p (sign y) (dense_path.PointMaxBN) -- half the epilogue and half
// the running registers of a pass that keeps both.  The sums of y and y^2 are not taken here either: they are  W3 (sum_r h2)
// and  diag(W3 (h2^T h2) W3^T), two quantities the backward needs anyway.  Ties go to the lower point index.
__global__ __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pointmax_fwd_f16(
    const float* __restrict__ h2, int P, const uint4* __restrict__ W3h, const int* __restrict__ ew3, float* __restrict__ vext,
    int* __restrict__ aext, const float* __restrict__ pre) {
  // pre != NULL: h2 is the RAW output z of the layer in front and the rows are relu(z scale + shift) (pre = scale | shift, 128 each:
  // its BatchNorm's transform, applied on load -- the transformed matrix is never written)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* s_w3 = reinterpret_cast<uint4*>(smem);                        // PNH_RING slabs                       (24 KB)
  int* s_e3 = reinterpret_cast<int*>(s_w3 + PNH_RING * PNH_SLAB_U4);   // 512: MINUS the rows' exponents
  float* s_st = reinterpret_cast<float*>(s_e3 + PN_C3);                // 2 quantities x 4 waves x 512        (16 KB)
  float* s_pre = s_st + 2 * 4 * PN_C3;                                 // 2 x 128
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const long long obj = blockIdx.x;

  const unsigned ring = pn_lds_addr(s_w3);
  const unsigned my = __builtin_amdgcn_readfirstlane(wave) * 1024u;
  const unsigned woff = tid * 16u;
#define PM_STAGE(slab, buf)                                                                                              \
  do {                                                                                                                   \
    pn_dma16(W3h, woff + (unsigned)(slab) * (PNH_SLAB_U4 * 16u), ring + (buf) * (PNH_SLAB_U4 * 16u) + my);                 \
    pn_dma16(W3h, woff + (unsigned)(slab) * (PNH_SLAB_U4 * 16u) + 4096u, ring + (buf) * (PNH_SLAB_U4 * 16u) + 4096u + my); \
  } while (0)
  PM_STAGE(0, 0);
  PM_STAGE(1, 1);
  for (int e = tid; e < PN_C3; e += PN_THREADS) s_e3[e] = -ew3[e];
  if (pre && tid < 2 * PN_C2) s_pre[tid] = pre[tid];
  pn_wait_vm<0>();
  __syncthreads();

  // running maxima in the CHANNEL's scale (y 2^ew[c]: the row's power of two is the same for every point, it comes out once
  // at the end)
  float mx[32];
  int ix[32];
#pragma unroll
  for (int t = 0; t < 32; ++t) { mx[t] = -FLT_MAX; ix[t] = 0; }
  unsigned cur = 0;

  const float* ho = h2 + obj * (long long)P * PN_C2;
  for (int p0 = 0; p0 < P; p0 += PN_PTS) {
    // ---- the wave's 2 x 16 rows of h2 -> two fp16 pieces at the row's own power of two (a row past the end repeats the
    // object's last one, under that row's index: a duplicate moves neither an extreme value nor the point it is found at)
    pf16x8 Ya[2][4], Yb[2][4];
    int nex[2][4];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int pp = p0 + wave * 32 + pt * 16 + j, p = pp < P ? pp : P - 1;
      const float* row = ho + (long long)p * PN_C2 + 4 * q;
      pf32x4 v[8];
#pragma unroll
      for (int t2 = 0; t2 < 8; ++t2) v[t2] = *reinterpret_cast<const pf32x4*>(row + 16 * t2);
      if (pre) {
#pragma unroll
        for (int t2 = 0; t2 < 8; ++t2) {
          const pf32x4 sc = *reinterpret_cast<const pf32x4*>(s_pre + 16 * t2 + 4 * q);
          const pf32x4 sh = *reinterpret_cast<const pf32x4*>(s_pre + PN_C2 + 16 * t2 + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[t2][e] = fmaxf(__fmaf_rn(v[t2][e], sc[e], sh[e]), 0.f);
        }
      }
      float m = 0.f;
#pragma unroll
      for (int t2 = 0; t2 < 8; ++t2)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[t2][e]));
      const int ex2 = pn_exponent(pn_point_max(m));
      const float sc = __builtin_bit_cast(float, (unsigned)(ex2 + 127) << 23);
#pragma unroll
      for (int e = 0; e < 4; ++e) nex[pt][e] = -__shfl(ex2, 4 * q + e, 64);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          _Float16 a, b;
          pn_split2(v[2 * s + (jj >> 2)][jj & 3] * sc, a, b);
          Ya[pt][s][jj] = a;
          Yb[pt][s][jj] = b;
        }
    }
    const int pb = p0 + wave * 32 + 4 * q;       // this lane's points: pb + e (tile 0), pb + 16 + e (tile 1)
    int pidx[2][4];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int pp = pb + 16 * pt + e; pidx[pt][e] = pp < P ? pp : P - 1; }
#pragma unroll
    for (int t3 = 0; t3 < 32; ++t3) {
      const unsigned nb = cur >= 1 ? cur - 1 : PNH_RING - 1;
      PM_STAGE((t3 + 2) & 31, nb);
      const uint4* sw = s_w3 + cur * PNH_SLAB_U4;
      pf32x4 acc0 = pf32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
      uint4 wf[8];
#pragma unroll
      for (int f = 0; f < 8; ++f) wf[f] = sw[f * 64 + lane];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const pf16x8 Wa = __builtin_bit_cast(pf16x8, wf[s * 2 + 0]);
        const pf16x8 Wb = __builtin_bit_cast(pf16x8, wf[s * 2 + 1]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya[0][s], Wb, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Yb[0][s], Wa, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya[0][s], Wa, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya[1][s], Wb, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Yb[1][s], Wa, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya[1][s], Wa, acc1, 0, 0, 0);
      }
      float bmx = mx[t3];
      int bix = ix[t3];
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = ldexpf(pt ? acc1[e] : acc0[e], nex[pt][e]);
          if (v > bmx) { bmx = v; bix = pidx[pt][e]; }
        }
      mx[t3] = bmx; ix[t3] = bix;
      cur = cur == PNH_RING - 1 ? 0 : cur + 1;
      pn_wait_vm<2>();
      __syncthreads();
    }
  }
  pn_wait_vm<0>();
#undef PM_STAGE
  // ---- the four lanes of a channel (16 apart: ascending point blocks), then the four waves
#pragma unroll
  for (int t3 = 0; t3 < 32; ++t3) {
    float bmx = ldexpf(mx[t3], s_e3[16 * t3 + j]);
    int bix = ix[t3];
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      const float omx = __shfl_xor(bmx, o, 64);
      const int oix = __shfl_xor(bix, o, 64);
      if (omx > bmx || (omx == bmx && oix < bix)) { bmx = omx; bix = oix; }
    }
    if (q == 0) {
      const int c = 16 * t3 + j;
      s_st[wave * PN_C3 + c] = bmx;
      reinterpret_cast<int*>(s_st)[(4 + wave) * PN_C3 + c] = bix;
    }
  }
  __syncthreads();
  for (int c = tid; c < PN_C3; c += PN_THREADS) {
    float bmx = -FLT_MAX;
    int bix = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float vmx = s_st[w * PN_C3 + c];
      const int vix = reinterpret_cast<const int*>(s_st)[(4 + w) * PN_C3 + c];
      if (vmx > bmx || (vmx == bmx && vix < bix)) { bmx = vmx; bix = vix; }
    }
    const long long o = obj * PN_C3 + c;
    vext[o] = bmx;
    aext[o] = bix;
  }
}

// The same pass in the organisation of k_pointnet_feat_f16w (W3's fragments resident in the registers of eight waves that walk over
// objects, the rows through LDS in half-passes of 64, preparers and multipliers meeting through per-wave counts): preparing is light
// here -- a wave's 16 rows arrive by LDS-DMA one turn ahead (8 KB per wave: with the staging area there is room for two buffers of
// planes, not three), take the BatchNorm in front on the way, their power of two, and leave as two fp16 planes.  The multiplying side
// keeps the point index beside each running maximum.  Same results as k_pointmax_fwd_f16 bit for bit (the same products in the
// same order per accumulator; maxima and ties do not depend on who looks at them).
#define PMW_BUFS 2
__global__ __launch_bounds__(PNW_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pointmax_fwd_f16w(
    const float* __restrict__ h2, int B, int P, const uint4* __restrict__ W3h, const int* __restrict__ ew3, float* __restrict__ vext,
    int* __restrict__ aext, const float* __restrict__ pre) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* s_y = reinterpret_cast<uint4*>(smem);               // [buffer 2][point tile 4][k-step 4][plane 2][lane 64]      (64 KB)
  uint4* s_in = s_y + PMW_BUFS * 4 * 4 * 2 * 64;             // [wave 8][channel tile 8][lane 64]: a wave's next 16 rows    (64 KB)
  int* s_e3 = reinterpret_cast<int*>(s_in + 8 * 8 * 64);     // 512: MINUS the rows' exponents
  float* s_pre = reinterpret_cast<float*>(s_e3 + PN_C3);     // 2 x 128
  int* s_ex = reinterpret_cast<int*>(s_pre + 2 * PN_C2);     // [buffer 2][64]: MINUS the exponents of the half-pass's rows
  int* s_pa = s_ex + PMW_BUFS * 64;                          // [wave 8]: half-passes prepared, [wave 8]: half-passes multiplied
  int* s_pb = s_pa + 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4, grp = wave >> 2, wt = wave & 3;

  for (int e = tid; e < PN_C3; e += PNW_THREADS) s_e3[e] = -ew3[e];
  if (pre && tid < 2 * PN_C2) s_pre[tid] = pre[tid];
  if (tid < 16) s_pa[tid] = 0;
  pf16x8 Wa[4][4], Wb[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      Wa[a][s] = __builtin_bit_cast(pf16x8, W3h[(((4 * wave + a) * 4 + s) * 2 + 0) * 64 + lane]);
      Wb[a][s] = __builtin_bit_cast(pf16x8, W3h[(((4 * wave + a) * 4 + s) * 2 + 1) * 64 + lane]);
    }
  float mx[4];                                               // lane (j, q): channel 16 (4 wave + a) + j over the points 4 q + e of the tiles
  int ix[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) { mx[a] = -FLT_MAX; ix[a] = 0; }
  __syncthreads();

  const int H = (P + 63) >> 6;
  const int G = __builtin_amdgcn_readfirstlane((int)gridDim.x);
  const int nobj = ((int)blockIdx.x < B) ? (B - 1 - (int)blockIdx.x) / G + 1 : 0;
  const int T = nobj * H;
  const unsigned my_in = __builtin_amdgcn_readfirstlane(pn_lds_addr(s_in + wave * 8 * 64));

  auto stage = [&](int k, int hp) {      // the 16 rows this wave prepares in half-pass hp of object k, on their way into its corner of LDS
    const long long obj = (long long)blockIdx.x + (long long)k * G;
    const float* ho = h2 + obj * (long long)P * PN_C2;
    const unsigned long long xa = (unsigned long long)ho;
    ho = reinterpret_cast<const float*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xa >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((int)xa));
    const int pp = hp * 64 + wt * 16 + j, p = pp < P ? pp : P - 1;
    const unsigned off = ((unsigned)p * PN_C2 + 4u * q) * 4u;
#pragma unroll
    for (int t2 = 0; t2 < 8; ++t2) pn_dma16(ho, off + 64u * t2, my_in + 1024u * t2);
  };
  auto produce = [&](int t, int k, int hp) {
    const int buf = t % PMW_BUFS;
    __builtin_amdgcn_s_setprio(3);
    pn_wait_vm<0>();
    pf32x4 v[8];
#pragma unroll
    for (int t2 = 0; t2 < 8; ++t2) v[t2] = __builtin_bit_cast(pf32x4, s_in[(wave * 8 + t2) * 64 + lane]);
    if (pre) {
#pragma unroll
      for (int t2 = 0; t2 < 8; ++t2) {
        const pf32x4 sc = *reinterpret_cast<const pf32x4*>(s_pre + 16 * t2 + 4 * q);
        const pf32x4 sh = *reinterpret_cast<const pf32x4*>(s_pre + PN_C2 + 16 * t2 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[t2][e] = fmaxf(__fmaf_rn(v[t2][e], sc[e], sh[e]), 0.f);
      }
    }
    float m = 0.f;
#pragma unroll
    for (int t2 = 0; t2 < 8; ++t2)
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[t2][e]));
    const int ex2 = pn_exponent(pn_point_max(m));
    const float sc = __builtin_bit_cast(float, (unsigned)(ex2 + 127) << 23);
    if (q == 0) s_ex[buf * 64 + 16 * wt + j] = -ex2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      pf16x8 ya, yb;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        _Float16 a, b;
        pn_split2(v[2 * s + (jj >> 2)][jj & 3] * sc, a, b);
        ya[jj] = a;
        yb[jj] = b;
      }
      s_y[(((buf * 4 + wt) * 4 + s) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, ya);
      s_y[(((buf * 4 + wt) * 4 + s) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, yb);
    }
    if (t + 2 < T) {                                         // this wave's next turn (its corner has been read: the values are in registers)
      int k2 = k, h2p = hp + 2;
      while (h2p >= H) { h2p -= H; ++k2; }
      stage(k2, h2p);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(s_pa + wave, (t >> 1) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_s_setprio(0);
  };
  auto count_min = [&](const int* c, int n) {
    int m = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int i = 1; i < n; ++i) {
      const int v = __hip_atomic_load(c + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      m = v < m ? v : m;
    }
    return __builtin_amdgcn_readfirstlane(m);
  };

  if (grp < T) stage(grp >= H ? 1 : 0, grp >= H ? 0 : grp);
  if (T > 0 && grp == 0) produce(0, 0, 0);
  int kc = 0, hc = 0;
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T && ((t + 1) & 1) == grp) {
      if (t + 1 >= PMW_BUFS) {
        while (count_min(s_pb, 8) < t + 2 - PMW_BUFS) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      produce(t + 1, hc + 1 == H ? kc + 1 : kc, hc + 1 == H ? 0 : hc + 1);
    }
    while (count_min(s_pa + 4 * (t & 1), 4) < (t >> 1) + 1) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int buf = t % PMW_BUFS;
    {
      const uint4* yl = s_y + buf * (4 * 4 * 2 * 64) + lane;
      uint4 Yn0 = yl[0], Yn1 = yl[64];
      pf32x4 acc[2][4];
      pi32x4 nex[2];
      auto maxima = [&](int h, int pt) {                     // tile pt's products against the running maxima (ties: the lower point stays)
        int pidx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int pp = hc * 64 + pt * 16 + 4 * q + e; pidx[e] = pp < P ? pp : P - 1; }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          float bmx = mx[a];
          int bix = ix[a];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = ldexpf(acc[h][a][e], nex[h][e]);
            if (v > bmx) { bmx = v; bix = pidx[e]; }
          }
          mx[a] = bmx;
          ix[a] = bix;
        }
      };
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        nex[pt & 1] = *reinterpret_cast<const pi32x4*>(s_ex + buf * 64 + 16 * pt + 4 * q);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const pf16x8 Ya = __builtin_bit_cast(pf16x8, Yn0), Yb = __builtin_bit_cast(pf16x8, Yn1);
          if (pt * 4 + s + 1 < 16) {
            Yn0 = yl[((pt * 4 + s + 1) * 2 + 0) * 64];
            Yn1 = yl[((pt * 4 + s + 1) * 2 + 1) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int a = 0; a < 4; ++a)
            acc[pt & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya, Wb[a][s], s == 0 ? pf32x4{0.f, 0.f, 0.f, 0.f} : acc[pt & 1][a], 0, 0, 0);
#pragma unroll
          for (int a = 0; a < 4; ++a) acc[pt & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Yb, Wa[a][s], acc[pt & 1][a], 0, 0, 0);
#pragma unroll
          for (int a = 0; a < 4; ++a) acc[pt & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ya, Wa[a][s], acc[pt & 1][a], 0, 0, 0);
          if (s == 0 && pt > 0) maxima((pt - 1) & 1, pt - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      maxima(1, 3);
    }
    const int k = kc;
    const bool last = hc == H - 1;
    if (++hc == H) { hc = 0; ++kc; }
    if (last) {                                              // the object's extremes leave: the four lanes of a channel first
      const long long obj = (long long)blockIdx.x + (long long)k * G;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int c = 16 * (4 * wave + a) + j;
        float bmx = ldexpf(mx[a], s_e3[c]);
        int bix = ix[a];
        {
          const unsigned mu = __builtin_bit_cast(unsigned, bmx), iu = (unsigned)bix;
          const auto sm = __builtin_amdgcn_permlane16_swap(mu, mu, false, false);
          const auto si = __builtin_amdgcn_permlane16_swap(iu, iu, false, false);
          const float m0 = __builtin_bit_cast(float, (unsigned)sm[0]), m1 = __builtin_bit_cast(float, (unsigned)sm[1]);
          const int i0 = (int)si[0], i1 = (int)si[1];
          const bool first = m0 > m1 || (m0 == m1 && i0 < i1);
          bmx = first ? m0 : m1;
          bix = first ? i0 : i1;
        }
        {
          const unsigned mu = __builtin_bit_cast(unsigned, bmx), iu = (unsigned)bix;
          const auto sm = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
          const auto si = __builtin_amdgcn_permlane32_swap(iu, iu, false, false);
          const float m0 = __builtin_bit_cast(float, (unsigned)sm[0]), m1 = __builtin_bit_cast(float, (unsigned)sm[1]);
          const int i0 = (int)si[0], i1 = (int)si[1];
          const bool first = m0 > m1 || (m0 == m1 && i0 < i1);
          bmx = first ? m0 : m1;
          bix = first ? i0 : i1;
        }
        if (q == 0) {
          vext[obj * PN_C3 + c] = bmx;
          aext[obj * PN_C3 + c] = bix;
        }
        mx[a] = -FLT_MAX;
        ix[a] = 0;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(s_pb + wave, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

// vext[b, c] = max_p y[b, p, c], aext = the (lowest) point it occurs at, for y = h2 W^T with W the (512, 128) weight whose f16 x 2
// image W3h / ew3 is (as glx_pointnet_feat_f16x2 takes it).  A caller that needs the MINIMUM of a channel hands in that row negated.
extern "C" int glx_pointmax_forward_f16x2(const float* h2, int B, int P, const void* W3h, const int32_t* ew3, float* vext,
                                          int32_t* aext, const float* pre_coef, void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(h2 && W3h && ew3 && vext && aext, "glx_pointmax_forward_f16x2: null pointer");
  GLX_REQUIRE(P >= 1, "glx_pointmax_forward_f16x2: P >= 1");
  if (g_pointnet_w_stationary) {
    const size_t ldw = (size_t)(PMW_BUFS * 4 * 4 * 2 * 64 + 8 * 8 * 64) * 16 + (size_t)(PN_C3 + 2 * PN_C2 + PMW_BUFS * 64 + 16) * 4;
    static bool attr_w = false;
    if (!attr_w) {
      GLX_HIP(hipFuncSetAttribute((const void*)k_pointmax_fwd_f16w, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldw));
      attr_w = true;
    }
    const int gw = B < glx_num_cus() ? B : glx_num_cus();
    hipLaunchKernelGGL(k_pointmax_fwd_f16w, dim3(gw), dim3(PNW_THREADS), ldw, (hipStream_t)stream, h2, B, P, (const uint4*)W3h, ew3,
                       vext, (int*)aext, pre_coef);
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  const size_t lds = (size_t)PNH_RING * PNH_SLAB_U4 * 16 + (size_t)(PN_C3 + 2 * 4 * PN_C3 + 2 * PN_C2) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_pointmax_fwd_f16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_pointmax_fwd_f16, dim3(B), dim3(PN_THREADS), lds, (hipStream_t)stream, h2, P, (const uint4*)W3h, ew3, vext,
                     (int*)aext, pre_coef);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// y[r, :] = init + x[r, :] W^T  for a 128 x 128 W on a tall row matrix, f16 x 2 products: the dense part of the 128 -> 512 layer's
// input gradient (dh2 = ... - v - h2 (W3^T diag(c) W3), dense_path.PointMaxBN.backward), 69 GFLOP at configs[3] that the library ran
// at half the fp32 matrix peak behind a read-modify-write of the 1 GB result.  The whole weight image (two fp16 planes, operand
// order: 64 KB) stays in LDS; a wave takes 2 x 16 rows per trip (their own powers of two, as everywhere in this file), the product
// comes out rows-in-lanes (lane (j, q): channels 16 t + 4 q + e of row j), so the result goes out as 16-byte stores.  Memory-bound:
// 1 GB in, 1 GB out.
#define RM_BLOCKS 512
// SUMS (needs pre): y is a gradient with respect to relu(x scale + shift) and the layer behind x is a BatchNorm: the block's sums over its
// rows of  dz = [x scale + shift > 0] y  and of  dz x  (x RAW: what that BatchNorm's backward is made of) go to bsum[block][2][128] -- the
// pass over y and x that would take them afterwards (two reads of a GB each at configs[3]) is not needed.  The raw rows stay in the
// registers the transformed ones had (the transform is applied where a value is used: twice), and the channel-tile loop is unrolled so
// that they can be addressed (a load issued inside the loop would return behind the next trip's rows: loads retire in order).
template <bool SUMS>
__global__ __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_rows128_affine_f16(
    const float* __restrict__ x, long long rows, const uint4* __restrict__ Wh, const int* __restrict__ ew,
    const float* __restrict__ init, float* __restrict__ y, const float* __restrict__ pre, float* __restrict__ bsum) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* s_w = reinterpret_cast<uint4*>(smem);                         // 8 tiles x 4 k-steps x 2 planes x 64 lanes (64 KB)
  int* s_e = reinterpret_cast<int*>(s_w + 8 * 4 * 2 * 64);             // 128: MINUS the rows' exponents
  float* s_i = reinterpret_cast<float*>(s_e + PN_C2);                  // 128: init
  float* s_pre = s_i + PN_C2;                                          // pre != NULL: x is relu(x scale + shift) on load (2 x 128)
  float* s_sum = s_pre + 2 * PN_C2;                                    // SUMS: [wave 4][2][128]
  if (pre && threadIdx.x < 2 * PN_C2) s_pre[threadIdx.x] = pre[threadIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  for (int e = tid; e < 8 * 4 * 2 * 64; e += PN_THREADS) s_w[e] = Wh[e];
  if (tid < PN_C2) { s_e[tid] = -ew[tid]; s_i[tid] = init ? init[tid] : 0.f; }
  __syncthreads();
  constexpr int NPT = SUMS ? 1 : 2;        // 16-row tiles per trip (SUMS: one -- the raw rows have to stay in registers beside the pieces)
  const long long ntrips = (rows + 16 * NPT - 1) / (16 * NPT), stride = (long long)gridDim.x * 4;
  long long trip = (long long)blockIdx.x * 4 + wave;
  pf32x4 nv[NPT][8];                       // the next trip's rows, on their way while this one is multiplied
#define RM_LOAD(T)                                                                       \
  _Pragma("unroll") for (int pt = 0; pt < NPT; ++pt) {                                   \
    const long long r0_ = (T) * (16 * NPT) + pt * 16 + j, r_ = r0_ < rows ? r0_ : rows - 1; \
    const float* row_ = x + r_ * PN_C2 + 4 * q;                                          \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) nv[pt][t] = *reinterpret_cast<const pf32x4*>(row_ + 16 * t); \
  }
  if (trip < ntrips) { RM_LOAD(trip) }
  float bs1[4] = {0.f, 0.f, 0.f, 0.f}, bs2[4] = {0.f, 0.f, 0.f, 0.f};      // SUMS: channels 16 (j & 7) + 4 q + e, over this wave's rows
  for (; trip < ntrips; trip += stride) {
    pf16x8 Xa[NPT][4], Xb[NPT][4];
    int nex[NPT];
    long long rr[NPT];
    pf32x4 v[NPT][8];                      // SUMS: the RAW rows; otherwise the rows as the product takes them
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        v[pt][t] = nv[pt][t];
        if (!SUMS && pre) {
          const pf32x4 sc = *reinterpret_cast<const pf32x4*>(s_pre + 16 * t + 4 * q);
          const pf32x4 sh = *reinterpret_cast<const pf32x4*>(s_pre + PN_C2 + 16 * t + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[pt][t][e] = fmaxf(__fmaf_rn(v[pt][t][e], sc[e], sh[e]), 0.f);
        }
      }
    if (trip + stride < ntrips) { RM_LOAD(trip + stride) }
    // the value the product takes of the raw one (SUMS: transformed where it is used)
    auto taken = [&](int pt, int t, int e) -> float {
      if constexpr (SUMS) return fmaxf(__fmaf_rn(v[pt][t][e], s_pre[16 * t + 4 * q + e], s_pre[PN_C2 + 16 * t + 4 * q + e]), 0.f);
      else return v[pt][t][e];
    };
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
      rr[pt] = trip * (16 * NPT) + pt * 16 + j;
      float m = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(taken(pt, t, e)));
      const int ex = pn_exponent(pn_point_max(m));
      nex[pt] = -ex;
      const float sc = __builtin_bit_cast(float, (unsigned)(ex + 127) << 23);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          _Float16 a, b;
          pn_split2(taken(pt, 2 * s + (jj >> 2), jj & 3) * sc, a, b);
          Xa[pt][s][jj] = a;
          Xb[pt][s][jj] = b;
        }
    }
    const bool in0 = rr[0] < rows, in1 = NPT > 1 && rr[NPT - 1] < rows;
    auto tile = [&](int t, const pf32x4& z0) {
      pf32x4 acc0 = pf32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const pf16x8 Wa = __builtin_bit_cast(pf16x8, s_w[((t * 4 + s) * 2 + 0) * 64 + lane]);
        const pf16x8 Wb = __builtin_bit_cast(pf16x8, s_w[((t * 4 + s) * 2 + 1) * 64 + lane]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wb, Xa[0][s], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wa, Xb[0][s], acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wa, Xa[0][s], acc0, 0, 0, 0);
        if constexpr (NPT > 1) {
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wb, Xa[NPT - 1][s], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wa, Xb[NPT - 1][s], acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wa, Xa[NPT - 1][s], acc1, 0, 0, 0);
        }
      }
      const int c = 16 * t + 4 * q;
      pf32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o0[e] = s_i[c + e] + ldexpf(acc0[e], s_e[c + e] + nex[0]);
        o1[e] = s_i[c + e] + ldexpf(acc1[e], s_e[c + e] + nex[NPT - 1]);
      }
      if (in0) *reinterpret_cast<pf32x4*>(y + rr[0] * PN_C2 + c) = o0;
      if (in1) *reinterpret_cast<pf32x4*>(y + rr[NPT - 1] * PN_C2 + c) = o1;
      if constexpr (SUMS) {
        // the ReLU's mask re-derived as every reader of x forms it; the 16 rows of a lane group summed by DPP (fixed order, every
        // lane of the group then holds the sums), lane j keeps the sums of channel tile j & 7: vector instructions only
        const pf32x4 sc = *reinterpret_cast<const pf32x4*>(s_pre + c);
        const pf32x4 sh = *reinterpret_cast<const pf32x4*>(s_pre + PN_C2 + c);
        const bool mine = (j & 7) == t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d0 = (in0 && __fmaf_rn(z0[e], sc[e], sh[e]) > 0.f) ? o0[e] : 0.f;
          const float s1 = pn_row_sum(d0), s2 = pn_row_sum(d0 * z0[e]);
          bs1[e] += mine ? s1 : 0.f;
          bs2[e] += mine ? s2 : 0.f;
        }
      }
    };
    if constexpr (SUMS) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        tile(t, v[0][t]);
        __builtin_amdgcn_sched_barrier(0);   // (or the 64 fragment reads of the eight tiles are hoisted and the registers spill)
      }
    } else {
#pragma unroll 1
      for (int t = 0; t < 8; ++t) tile(t, v[0][0]);      // (rolled for the same reason; the row argument is not used)
    }
  }
  if constexpr (SUMS) {
    if (j < 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s_sum[(wave * 2 + 0) * PN_C2 + 16 * j + 4 * q + e] = bs1[e];
        s_sum[(wave * 2 + 1) * PN_C2 + 16 * j + 4 * q + e] = bs2[e];
      }
    }
    __syncthreads();
    if (tid < 2 * PN_C2)
      bsum[(long long)blockIdx.x * 2 * PN_C2 + tid] =
          ((s_sum[tid] + s_sum[2 * PN_C2 + tid]) + s_sum[4 * PN_C2 + tid]) + s_sum[6 * PN_C2 + tid];
  }
}

#undef RM_LOAD

// Wh / ew: the (128 out, 128 in) weight as two fp16 planes of w 2^ew[row] in operand order (dense_path.PointFeat._f16x2_image);
// init: 128 floats or NULL.  x, y: (rows, 128) fp32, y may not alias x.
static int rows128_affine_blocks(long long rows, bool sums) {
  const long long per_trip = sums ? 16 : 32, want = ((rows + per_trip - 1) / per_trip + 3) >> 2;
  return (int)(want < RM_BLOCKS ? want : RM_BLOCKS);
}
extern "C" int glx_rows128_affine_blocks(long long rows) { return rows > 0 ? rows128_affine_blocks(rows, true) : 0; }
extern "C" int glx_rows128_affine_f16x2_sums(const float* x, long long rows, const void* Wh, const int32_t* ew, const float* init,
                                             float* y, const float* pre_coef, float* bsum, void* stream);
extern "C" int glx_rows128_affine_f16x2(const float* x, long long rows, const void* Wh, const int32_t* ew, const float* init,
                                        float* y, const float* pre_coef, void* stream) {
  return glx_rows128_affine_f16x2_sums(x, rows, Wh, ew, init, y, pre_coef, nullptr, stream);
}
extern "C" int glx_rows128_affine_f16x2_sums(const float* x, long long rows, const void* Wh, const int32_t* ew, const float* init,
                                             float* y, const float* pre_coef, float* bsum, void* stream) {
  if (rows <= 0) return GLX_OK;
  GLX_REQUIRE(x && Wh && ew && y, "glx_rows128_affine_f16x2: null pointer");
  GLX_REQUIRE(!bsum || pre_coef, "glx_rows128_affine_f16x2_sums: the sums need the transform in front (pre_coef)");
  const size_t lds = (size_t)8 * 4 * 2 * 64 * 16 + (size_t)(4 + 8) * PN_C2 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_rows128_affine_f16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GLX_HIP(hipFuncSetAttribute((const void*)k_rows128_affine_f16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int blocks = rows128_affine_blocks(rows, bsum != nullptr);
  if (bsum)
    hipLaunchKernelGGL(k_rows128_affine_f16<true>, dim3(blocks), dim3(PN_THREADS), lds, (hipStream_t)stream, x, rows, (const uint4*)Wh,
                       ew, init, y, pre_coef, bsum);
  else
    hipLaunchKernelGGL(k_rows128_affine_f16<false>, dim3(blocks), dim3(PN_THREADS), lds, (hipStream_t)stream, x, rows, (const uint4*)Wh,
                       ew, init, y, pre_coef, bsum);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ the f16 x 2 weight image
// W (Cout, Cin) (any strides; optionally times a per-row factor and a constant) -> two fp16 planes of w 2^ew[row] in the operand order of
// the kernels above ([output tile][k-step][plane][lane 16 q + m][slot 4 h + e] = W[16 tile + m][32 s + 16 h + 4 q + e]) + the rows'
// exponents (max |w| 2^ew in [2^14, 2^15), 0 for a zero row).  One launch; as tensor statements the same image was ~25 launches of
// ~4.5 us, eight times per CVAE training step.
__global__ __launch_bounds__(128) void k_f16x2_pack(const float* __restrict__ w, int cout, int cin, long long s_row, long long s_col,
                                                    const float* __restrict__ row_scale, int sign_only, float scale,
                                                    _Float16* __restrict__ img, int* __restrict__ ew) {
  __shared__ float s_m[2];
  const int r = blockIdx.x, c = threadIdx.x;
  const float rs = row_scale ? (sign_only ? (row_scale[r] >= 0.f ? 1.f : -1.f) : row_scale[r]) : 1.f;
  const float f = scale * rs;
  const float v = c < cin ? w[r * s_row + c * s_col] * f : 0.f;
  float m = fabsf(v);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  const int e = pn_exponent(fmaxf(s_m[0], s_m[1]));
  if (c == 0) ew[r] = e;
  if (c < cin) {
    const int S = cin >> 5, tile = r >> 4, mm = r & 15;
    const int s_ = c >> 5, h = (c >> 4) & 1, q = (c >> 2) & 3, ee = c & 3;
    _Float16 a, b;
    pn_split2(ldexpf(v, e), a, b);
    const size_t base = ((size_t)(tile * S + s_) * 2 * 64 + 16 * q + mm) * 8 + 4 * h + ee;
    img[base] = a;
    img[base + (size_t)64 * 8] = b;
  }
}

// img: Cout x Cin x 2 halfs; ew: Cout int32.  Cout % 16 == 0, Cin % 32 == 0, Cin <= 128.  row_scale: Cout floats or NULL
// (row_scale_sign_only: only its sign is used: +1 for >= 0, -1 otherwise).
extern "C" int glx_f16x2_pack(const float* w, int cout, int cin, long long stride_row, long long stride_col, const float* row_scale,
                              int row_scale_sign_only, float scale, void* img, int32_t* ew, void* stream) {
  GLX_REQUIRE(w && img && ew, "glx_f16x2_pack: null pointer");
  GLX_REQUIRE(cout > 0 && cout % 16 == 0 && cin >= 32 && cin % 32 == 0 && cin <= 128, "glx_f16x2_pack: %d x %d (rows %% 16, 32 <= columns <= 128, %% 32)", cout, cin);
  hipLaunchKernelGGL(k_f16x2_pack, dim3(cout), dim3(128), 0, (hipStream_t)stream, w, cout, cin, stride_row, stride_col, row_scale,
                     row_scale_sign_only, scale, (_Float16*)img, (int*)ew);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ 64 -> 128 point layer, f16 x 2
// z (rows, 128) = x (rows, 64) W^T with the training-mode BatchNorm statistics of z in the epilogue: the CVAE's second point layer on
// 2.1 M rows (cvae_uncertainty/point_net.py:17,24).  The fp32-MFMA row kernel (csrc/glx_rows.hip, two column halves) is bound by its
// matrix instructions (0.53 ms: 34 GFLOP at 0.4 of the fp32 peak); with f16 x 2 products the pass is its memory traffic (0.5 GB in, 1 GB
// out).  Structure of k_rows128_affine_f16: the weight image (two fp16 planes in operand order, 32 KB) in LDS, a wave takes 2 x 16
// rows per trip (their own powers of two), rows prefetched a trip ahead; the sums of z and z^2 stay in fp32 per lane over the wave's
// trips (<= 64 values each at 2.1 M rows), meet in fp64 at the end and go through the shared accumulator / ticket / last-block
// finalize of glx_bn_state.h.
#define RF_BLOCKS 512
__global__ __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_rows_linear_64_128_f16(
    const float* __restrict__ x, int rows, const int* __restrict__ n_live, const uint4* __restrict__ Wh, const int* __restrict__ ew,
    float* __restrict__ z, BnState* __restrict__ st, BnFinalize f) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* s_w = reinterpret_cast<uint4*>(smem);                         // 8 tiles x 2 k-steps x 2 planes x 64 lanes (32 KB)
  int* s_e = reinterpret_cast<int*>(s_w + 8 * 2 * 2 * 64);             // 128: MINUS the rows' exponents
  double* s_red = reinterpret_cast<double*>(s_e + PN_C2);              // [4 waves][32 float4 columns][2 moments][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  int n = rows;
  if (n_live) n = min(rows, *n_live);
  for (int e = tid; e < 8 * 2 * 2 * 64; e += PN_THREADS) s_w[e] = Wh[e];
  if (tid < PN_C2) s_e[tid] = -ew[tid];
  __syncthreads();
  // the product TRANSPOSED (rows x channels): lane (j, q) holds channel 16 t + j of rows 4 q + e, so a lane's statistics are ONE
  // channel's per tile (16 registers instead of 64) and the reduction over rows is mostly inside the lane
  float s0[8], s1[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) s0[t] = s1[t] = 0.f;
  const int ntrips = (n + 31) >> 5, stride = gridDim.x * 4;
  int trip = blockIdx.x * 4 + wave;
  pf32x4 nv[2][4];
#define RF_LOAD(T)                                                                       \
  _Pragma("unroll") for (int pt = 0; pt < 2; ++pt) {                                     \
    const int r0_ = (T) * 32 + pt * 16 + j, r_ = r0_ < n ? r0_ : n - 1;                  \
    const float* row_ = x + (long long)r_ * PN_C1 + 4 * q;                               \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) nv[pt][t] = *reinterpret_cast<const pf32x4*>(row_ + 16 * t); \
  }
  if (trip < ntrips) { RF_LOAD(trip) }
  for (; trip < ntrips; trip += stride) {
    pf16x8 Xa[2][2], Xb[2][2];
    int nex[2][4];
    pf32x4 v[2][4];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
      for (int t = 0; t < 4; ++t) v[pt][t] = nv[pt][t];
    if (trip + stride < ntrips) { RF_LOAD(trip + stride) }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      float m = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[pt][t][e]));
      const int ex = pn_exponent(pn_point_max(m));
#pragma unroll
      for (int e = 0; e < 4; ++e) nex[pt][e] = -__shfl(ex, 4 * q + e, 64);
      const float sc = __builtin_bit_cast(float, (unsigned)(ex + 127) << 23);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          _Float16 a, b;
          pn_split2(v[pt][2 * s + (jj >> 2)][jj & 3] * sc, a, b);
          Xa[pt][s][jj] = a;
          Xb[pt][s][jj] = b;
        }
    }
    const int rb = trip * 32 + 4 * q;             // this lane's rows: rb + e (tile 0), rb + 16 + e (tile 1)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      pf32x4 acc0 = pf32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const pf16x8 Wa = __builtin_bit_cast(pf16x8, s_w[((t * 2 + s) * 2 + 0) * 64 + lane]);
        const pf16x8 Wb = __builtin_bit_cast(pf16x8, s_w[((t * 2 + s) * 2 + 1) * 64 + lane]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Xa[0][s], Wb, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Xb[0][s], Wa, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Xa[0][s], Wa, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Xa[1][s], Wb, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Xb[1][s], Wa, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Xa[1][s], Wa, acc1, 0, 0, 0);
      }
      const int c = 16 * t + j, nw = s_e[c];
      float a0_ = 0.f, a1_ = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r0_ = rb + e, r1_ = rb + 16 + e;
        const float o0 = r0_ < n ? ldexpf(acc0[e], nw + nex[0][e]) : 0.f;
        const float o1 = r1_ < n ? ldexpf(acc1[e], nw + nex[1][e]) : 0.f;
        if (r0_ < n) z[(long long)r0_ * PN_C2 + c] = o0;
        if (r1_ < n) z[(long long)r1_ * PN_C2 + c] = o1;
        a0_ += o0 + o1;
        a1_ = fmaf(o0, o0, fmaf(o1, o1, a1_));
      }
      s0[t] += a0_;
      s1[t] += a1_;
      __builtin_amdgcn_sched_barrier(0);          // (keeps the next tiles' fragment reads from being hoisted: registers)
    }
  }
#undef RF_LOAD
  if (!st) return;
  // ---- the four lanes of a channel (16 apart; fp64 from here), then the four waves: thread c4 < 32 ends up with float4 column c4
  __shared__ int s_last;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    double d0 = (double)s0[t], d1 = (double)s1[t];
    d0 += __shfl_xor(d0, 16, 64);
    d1 += __shfl_xor(d1, 16, 64);
    d0 += __shfl_xor(d0, 32, 64);
    d1 += __shfl_xor(d1, 32, 64);
    if (q == 0) {
      const int c = 16 * t + j;
      s_red[((wave * 32 + (c >> 2)) * 2 + 0) * 4 + (c & 3)] = d0;
      s_red[((wave * 32 + (c >> 2)) * 2 + 1) * 4 + (c & 3)] = d1;
    }
  }
  __syncthreads();
  double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
  if (tid < PN_C2 / 4) {
#pragma unroll
    for (int w_ = 0; w_ < 4; ++w_)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0[e] += s_red[((w_ * 32 + tid) * 2 + 0) * 4 + e];
        a1[e] += s_red[((w_ * 32 + tid) * 2 + 1) * 4 + e];
      }
  }
  if (!bn_contribute(st, PN_C2, a0, a1, gridDim.x, &s_last)) return;
  __shared__ double s_fin[PN_THREADS][2];
  bn_finalize_sets<false, PN_THREADS>(st, f, PN_C2, n, s_fin);
}

// x (rows, 64), Wh / ew: the (128, 64) weight as two fp16 planes of w 2^ew[row] in operand order (dense_path.PointFeat._f16x2_image);
// the other arguments as glx_rows_linear_bn_forward's (bn_state == NULL: the product alone).
extern "C" int glx_rows_linear_bn_forward_64_128_f16x2(const float* x, int rows, const void* Wh, const int32_t* ew, const int32_t* n_live,
                                                       float* z, const float* gamma, const float* beta, float eps, float momentum,
                                                       float* running_mean, float* running_var, float* coef, float* save_mean,
                                                       float* save_invstd, void* bn_state, void* stream) {
  GLX_REQUIRE(Wh && ew && (rows == 0 || (x && z)), "glx_rows_linear_bn_forward_64_128_f16x2: null pointer");
  GLX_REQUIRE(!bn_state || (coef && save_mean && save_invstd), "glx_rows_linear_bn_forward_64_128_f16x2: statistics without their outputs");
  if (rows <= 0) return GLX_OK;
  BnFinalize f{gamma, beta, eps, momentum, coef, save_mean, save_invstd, running_mean, running_var, nullptr, nullptr, nullptr, 0};
  const size_t lds = (size_t)8 * 2 * 2 * 64 * 16 + (size_t)PN_C2 * 4 + (size_t)4 * 32 * 2 * 4 * 8;
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_rows_linear_64_128_f16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int want = (((rows + 31) >> 5) + 3) >> 2;
  const int blocks = want < 1 ? 1 : (want > RF_BLOCKS ? RF_BLOCKS : want);
  hipLaunchKernelGGL(k_rows_linear_64_128_f16, dim3(blocks), dim3(PN_THREADS), lds, (hipStream_t)stream, x, rows, (const int*)n_live,
                     (const uint4*)Wh, ew, z, (BnState*)bn_state, f);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ 64 -> 128 point layer, backward
// Backward of z = x W^T (x (rows, 64), W (128, 64)) behind a training-mode BatchNorm (+ ReLU) in ONE pass over the rows, both products
// on the 16-bit matrix pipe:  dz = A [y > 0] dy + C z + D  is formed on load (the BatchNorm backward folded to three vectors, as
// csrc/glx_rows.hip's k_rows_linear_bwd forms it from seven),
//   dX = dz W      contraction over the 128 channels: f16 x 2 with the row's own power of two (k_rows128_affine_f16's arithmetic; the
//                  weight image W^T in LDS),
//   dW = dz^T x    contraction over the ROWS: bf16 x 3 (k_rows128_moments' arithmetic): dz and x go through LDS once and come back as
//                  operand fragments whose eight k values are eight rows.
// The fp32-MFMA kernel it replaces is bound by its matrix instructions (2 x 0.58 ms at 2.1 M rows as column halves, 1.21 ms as one
// pass).  A block of four waves takes 64 rows per step as two teams of 32; wave (t, h) owns rows 32 t + 16 h .. + 15 for dX and the
// output-channel tiles 4 h .. 4 h + 3 of its team's dW.  Per-block partial dW (the two teams added), summed by
// k_rows_bwd_64_128_reduce in block order.
#define RB_PITCH 132
#define RB_BLOCKS 512
struct RowsBwd128 {
  const float* coef_fwd;   // scale | shift of the forward transform (2 x 128): the ReLU mask is re-derived from z
  const float* coef3;      // a | b | c of dz = a (g - b - c xhat) (3 x 128); NULL: dy is the gradient of z itself
  const float* mean;
  const float* invstd;
  int relu;
};

__global__ __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_rows_bwd_64_128_f16(
    const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ dy, const uint4* __restrict__ Wth,
    const int* __restrict__ ewt, int rows, const int* __restrict__ n_live, RowsBwd128 bn, float* __restrict__ gx,
    float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint4* s_w = reinterpret_cast<uint4*>(smem);                          // W^T image: 4 tiles x 4 k-steps x 2 planes x 64 lanes (16 KB)
  float* s_dz = reinterpret_cast<float*>(s_w + 4 * 4 * 2 * 64);         // [team][32 rows][RB_PITCH]                       (33 KB)
  uint4* s_xf = reinterpret_cast<uint4*>(s_dz + 2 * 32 * RB_PITCH);     // [team][x tile 4][piece 3][lane]                  (24 KB)
  float* s_cf = reinterpret_cast<float*>(s_xf + 2 * 4 * 3 * 64);        // A | C | D | scale | shift, 128 each
  int* s_e = reinterpret_cast<int*>(s_cf + 5 * PN_C2);                  // 64: MINUS the exponents of W^T's rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4, team = wave >> 1, half = wave & 1;
  int n = rows;
  if (n_live) n = min(rows, *n_live);
  for (int e = tid; e < 4 * 4 * 2 * 64; e += PN_THREADS) s_w[e] = Wth[e];
  if (tid < PN_C1) s_e[tid] = -ewt[tid];
  const bool BN = bn.coef3 != nullptr;
  if (tid < PN_C2) {
    float a = 1.f, cC = 0.f, cD = 0.f, sc = 0.f, sh = 1.f;
    if (BN) {
      const float b = bn.coef3[PN_C2 + tid], cc = bn.coef3[2 * PN_C2 + tid], mu = bn.mean[tid], is = bn.invstd[tid];
      a = bn.coef3[tid];
      cC = -a * cc * is;
      cD = a * (cc * is * mu - b);
      sc = bn.coef_fwd[tid];
      sh = bn.coef_fwd[PN_C2 + tid];
    }
    s_cf[tid] = a; s_cf[PN_C2 + tid] = cC; s_cf[2 * PN_C2 + tid] = cD; s_cf[3 * PN_C2 + tid] = sc; s_cf[4 * PN_C2 + tid] = sh;
  }
  __syncthreads();
  const bool mask = BN && bn.relu;
  float* dzs = s_dz + team * 32 * RB_PITCH;
  uint4* xfs = s_xf + team * 4 * 3 * 64;
  pf32x4 accw[4][4];           // [output-channel tile 4 half + a][input-channel tile b]: dW[16 (4 half + a) + 4 q + e][16 b + j]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) accw[a][b] = pf32x4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = (n + 63) >> 6;
  // every global load of a step is issued in one go, one step AHEAD: behind the first barrier of the step before (the registers they
  // land in are dead there), so they fly during that step's weight-gradient phase and its second barrier
  pf32x4 dzv[8], zv[8];
  float xs[2][8];
#define RB_LOAD(S)                                                                                            \
  {                                                                                                           \
    const int tb_ = (S) * 64 + 32 * team, row_ = tb_ + 16 * half + j;                                         \
    const long long ro_ = (long long)(row_ < n ? row_ : n - 1) * PN_C2 + 4 * q;                               \
    _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) dzv[s_] = *reinterpret_cast<const pf32x4*>(dy + ro_ + 16 * s_); \
    if (BN) {                                                                                                 \
      _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) zv[s_] = *reinterpret_cast<const pf32x4*>(z + ro_ + 16 * s_); \
    }                                                                                                         \
  }
  if ((int)blockIdx.x < nsteps) RB_LOAD(blockIdx.x)
  for (int step = blockIdx.x; step < nsteps; step += gridDim.x) {
    const int tb0 = step * 64 + 32 * team;               // the team's first row
    const int row = tb0 + 16 * half + j;
    const bool live = row < n;
#pragma unroll
    for (int a = 0; a < 2; ++a)             // (x: 16 dwords per lane, mostly L2 hits -- the team's other wave reads the same rows)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int rr = tb0 + 8 * q + r;
        xs[a][r] = x[(long long)(rr < n ? rr : n - 1) * PN_C1 + 16 * (2 * half + a) + j];
      }
    // ---- dz of this wave's 16 rows: lane (j, q) = row j, channels 16 s + 4 q ..
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) {
      pf32x4 g = dzv[s_];
      if (BN) {
        const int c = 16 * s_ + 4 * q;
        const pf32x4 cA = *reinterpret_cast<const pf32x4*>(s_cf + c), cC = *reinterpret_cast<const pf32x4*>(s_cf + PN_C2 + c);
        const pf32x4 cD = *reinterpret_cast<const pf32x4*>(s_cf + 2 * PN_C2 + c);
        const pf32x4 sc = *reinterpret_cast<const pf32x4*>(s_cf + 3 * PN_C2 + c), sh = *reinterpret_cast<const pf32x4*>(s_cf + 4 * PN_C2 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float gg = g[e];
          if (mask) gg = bn_affine(zv[s_][e], sc[e], sh[e]) > 0.f ? gg : 0.f;
          g[e] = fmaf(cA[e], gg, fmaf(cC[e], zv[s_][e], cD[e]));
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = live ? g[e] : 0.f;
      dzv[s_] = g;
      *reinterpret_cast<pf32x4*>(dzs + (16 * half + j) * RB_PITCH + 16 * s_ + 4 * q) = g;
    }
    // ---- this wave's two tiles of x as operand fragments (lane (i, kg): rows 8 kg .. of channel 16 tile + i), for the team
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int tile = 2 * half + a;
      bf16x8 p0, p1, p2;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float v = tb0 + 8 * q + r < n ? xs[a][r] : 0.f;
        __bf16 b0, b1, b2;
        cv_split(v, b0, b1, b2);
        p0[r] = b0; p1[r] = b1; p2[r] = b2;
      }
      xfs[(tile * 3 + 0) * 64 + lane] = __builtin_bit_cast(uint4, p0);
      xfs[(tile * 3 + 1) * 64 + lane] = __builtin_bit_cast(uint4, p1);
      xfs[(tile * 3 + 2) * 64 + lane] = __builtin_bit_cast(uint4, p2);
    }
    // ---- dX of the wave's rows: f16 x 2, the row's own power of two
    if (gx) {
      float m = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(dzv[s_][e]));
      const int ex = pn_exponent(pn_point_max(m));
      const float scl = __builtin_bit_cast(float, (unsigned)(ex + 127) << 23);
      pf16x8 Xa[4], Xb[4];
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          _Float16 a_, b_;
          pn_split2(dzv[2 * s_ + (jj >> 2)][jj & 3] * scl, a_, b_);
          Xa[s_][jj] = a_;
          Xb[s_][jj] = b_;
        }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        pf32x4 acc = pf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
          const pf16x8 Wa = __builtin_bit_cast(pf16x8, s_w[((t * 4 + s_) * 2 + 0) * 64 + lane]);
          const pf16x8 Wb = __builtin_bit_cast(pf16x8, s_w[((t * 4 + s_) * 2 + 1) * 64 + lane]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wb, Xa[s_], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wa, Xb[s_], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wa, Xa[s_], acc, 0, 0, 0);
        }
        if (live) {
          const int c = 16 * t + 4 * q;
          pf32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = ldexpf(acc[e], s_e[c + e] - ex);
          *reinterpret_cast<pf32x4*>(gx + (long long)row * PN_C1 + c) = o;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                                      // the teams' dz patches and x fragments are complete
    if (step + (int)gridDim.x < nsteps) RB_LOAD(step + gridDim.x)
    // ---- dW: this wave's four output-channel tiles x the four input-channel tiles, k = the team's 32 rows
    if (part) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        bf16x8 fa[3];
        {
          bf16x8 p0, p1, p2;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            __bf16 b0, b1, b2;
            cv_split(dzs[(8 * q + r) * RB_PITCH + 16 * (4 * half + a) + j], b0, b1, b2);
            p0[r] = b0; p1[r] = b1; p2[r] = b2;
          }
          fa[0] = p0; fa[1] = p1; fa[2] = p2;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          bf16x8 fb[3];                                    // (re-read per output tile: the registers hold the next step's rows)
#pragma unroll
          for (int p = 0; p < 3; ++p) fb[p] = __builtin_bit_cast(bf16x8, xfs[(b * 3 + p) * 64 + lane]);
          pf32x4 t_ = accw[a][b];
          BF3_MFMA6(t_, fa, fb);
          accw[a][b] = t_;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                                      // before the next step overwrites the patches
  }
#undef RB_LOAD
  if (gx) {                               // rows past the live count: zero gradient
    const long long e0 = (long long)n * PN_C1, e1 = (long long)rows * PN_C1;
    for (long long e = e0 + ((long long)blockIdx.x * PN_THREADS + tid) * 4; e < e1; e += (long long)gridDim.x * PN_THREADS * 4)
      *reinterpret_cast<pf32x4*>(gx + e) = pf32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (!part) return;
  // ---- the block's partial: team 1 hands its sums to team 0 through LDS (same tiles), team 0 writes
  float* s_acc = s_dz;                                    // 2 waves x 16 tiles x 4 x 64 floats = 32 KB <= the patches' 33 KB
  __syncthreads();
  if (team == 1) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) s_acc[(((half * 4 + a) * 4 + b) * 4 + e) * 64 + lane] = accw[a][b][e];
  }
  __syncthreads();
  if (team == 0) {
    float* dst = part + (size_t)blockIdx.x * (PN_C2 * PN_C1);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k_ = (((half * 4 + a) * 4 + b) * 4 + e) * 64 + lane;
          dst[k_] = accw[a][b][e] + s_acc[k_];
        }
  }
}

// gw[co][ci] = the sum of the blocks' partials in block order (16 threads per element, the groups added in a fixed order);
// partial element (((half * 4 + a) * 4 + b) * 4 + e) * 64 + lane = dW[16 (4 half + a) + 4 (lane >> 4) + e][16 b + (lane & 15)]
__global__ __launch_bounds__(256) void k_rows_bwd_64_128_reduce(const float* __restrict__ part, int nblocks, float* __restrict__ gw) {
  __shared__ float s_g[16][16];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;                     // co * 64 + ci
  const int co = e >> 6, ci = e & 63;
  const int ta = co >> 4, half = ta >> 2, a = ta & 3, b = ci >> 4;
  const int idx = (((half * 4 + a) * 4 + b) * 4 + (co & 3)) * 64 + 16 * ((co & 15) >> 2) + (ci & 15);
  float sum = 0.f;
  for (int blk = g; blk < nblocks; blk += 16) sum += part[(size_t)blk * (PN_C2 * PN_C1) + idx];
  s_g[g][el] = sum;
  __syncthreads();
  if (g == 0) {
    float t_ = s_g[0][el];
#pragma unroll
    for (int k = 1; k < 16; ++k) t_ += s_g[k][el];
    gw[e] = t_;
  }
}

extern "C" size_t glx_rows_bwd_64_128_workspace_bytes(void) { return glx_align((size_t)RB_BLOCKS * PN_C2 * PN_C1 * sizeof(float)); }

// The backward of glx_rows_linear_bn_forward(_64_128_f16x2) for Cin = 64, Cout = 128, arguments as glx_rows_linear_bn_backward's,
// with the weight as Wth / ewt = the f16 x 2 image of W^T ((64, 128): dense_path.PointFeat._f16x2_image(w.t())).
extern "C" int glx_rows_linear_bn_backward_64_128_f16x2(const float* x, const float* z, const float* dy, int rows, const void* Wth,
                                                        const int32_t* ewt, const int32_t* n_live, const float* coef_fwd, int relu,
                                                        const float* coef3, const float* mean, const float* invstd, float* gx,
                                                        float* gw, void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(Wth && ewt && (rows == 0 || (x && dy)), "glx_rows_linear_bn_backward_64_128_f16x2: null pointer");
  GLX_REQUIRE(!coef3 || (z && coef_fwd && mean && invstd), "glx_rows_linear_bn_backward_64_128_f16x2: BatchNorm backward without z / coefficients");
  GLX_REQUIRE(!gw || (workspace && workspace_bytes >= glx_rows_bwd_64_128_workspace_bytes()),
              "glx_rows_linear_bn_backward_64_128_f16x2: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 0) {
    if (gw) GLX_HIP(hipMemsetAsync(gw, 0, (size_t)PN_C2 * PN_C1 * sizeof(float), st));
    return GLX_OK;
  }
  const size_t lds = (size_t)4 * 4 * 2 * 64 * 16 + (size_t)2 * 32 * RB_PITCH * 4 + (size_t)2 * 4 * 3 * 64 * 16 + (size_t)5 * PN_C2 * 4 + PN_C1 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_rows_bwd_64_128_f16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int nsteps = (rows + 63) >> 6;
  const int blocks = nsteps < RB_BLOCKS ? nsteps : RB_BLOCKS;
  RowsBwd128 bn{coef_fwd, coef3, mean, invstd, relu};
  hipLaunchKernelGGL(k_rows_bwd_64_128_f16, dim3(blocks), dim3(PN_THREADS), lds, st, x, z, dy, (const uint4*)Wth, ewt, rows,
                     (const int*)n_live, bn, gx, gw ? (float*)workspace : (float*)nullptr);
  if (gw)
    hipLaunchKernelGGL(k_rows_bwd_64_128_reduce, dim3(PN_C2 * PN_C1 / 16), dim3(256), 0, st, (const float*)workspace, blocks, gw);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ moments of a tall (rows, 128) matrix
// G = x^T x (128 x 128) and h = sum_r x[r, :] in ONE pass over x: the batch statistics of the 128 -> 512 layer's output (sum y = W3 h,
// sum y^2 = diag(W3 G W3^T)) and two terms of its weight gradient (dense_path.PointMaxBN).  The library's split-K product took 0.49 ms
// for G at 2.1 M rows and a reduction kernel another 0.15 for h; the pass is 1 GB of reads.  The contraction runs over the ROWS, so
// no power of two can be taken out per row: three bf16 pieces per value (fp32's own exponent range, products exact to 2^-22), six
// MFMAs per tile.  A block walks a contiguous range of rows, 32 at a time: the rows land in LDS as fp32, wave w turns channel tiles
// w and 7 - w into MFMA operand fragments (the A operand of a tile and its B operand are the same registers: lane (i, kg) holds rows
// 8 kg .. 8 kg + 7 of channel 16 t + i) and leaves them in LDS for the other waves; wave w then owns the tiles (w, >= w) and
// (7 - w, >= 7 - w) of the upper triangle, nine of the 36 each.  Per-block partial sums in accumulator order; k_rows128_moments_reduce
// adds them in block order in fp64 and mirrors the triangle.
#define MO_BLOCKS 768
#define MO_PITCH 132                       // floats per staged row (16-byte aligned rows; the column reads are two-way conflicted)
__global__ __launch_bounds__(256) void k_rows128_moments(const float* __restrict__ x, long long rows, float* __restrict__ part,
                                                         float* __restrict__ hpart, const float* __restrict__ pre) {
  __shared__ __attribute__((aligned(16))) float s_x[32 * MO_PITCH];
  __shared__ __attribute__((aligned(16))) uint4 s_f[8 * 3 * 64];          // [tile][piece][lane]: 8 bf16 each
  __shared__ float s_h[8][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kg = lane >> 4;
  const long long nsteps = (rows + 31) >> 5, per = (nsteps + gridDim.x - 1) / gridDim.x;
  const long long s0 = (long long)blockIdx.x * per, s1 = s0 + per < nsteps ? s0 + per : nsteps;
  pf32x4 acc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[a][t] = pf32x4{0.f, 0.f, 0.f, 0.f};
  pf32x4 hs = pf32x4{0.f, 0.f, 0.f, 0.f};
  const int c4 = tid & 31, r0 = tid >> 5;          // this thread's float4 column and its first row of a step (+ 8 u)
  // pre != NULL: the moments of relu(x scale + shift) (the transform of the BatchNorm in front, applied on load)
  const pf32x4 psc = pre ? *reinterpret_cast<const pf32x4*>(pre + 4 * c4) : pf32x4{1.f, 1.f, 1.f, 1.f};
  const pf32x4 psh = pre ? *reinterpret_cast<const pf32x4*>(pre + PN_C2 + 4 * c4) : pf32x4{0.f, 0.f, 0.f, 0.f};
  pf32x4 nv[4];
#define MO_LOAD(S)                                                                                           \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                            \
    const long long row_ = (S) * 32 + r0 + 8 * u;                                                            \
    nv[u] = row_ < rows ? *reinterpret_cast<const pf32x4*>(x + row_ * PN_C2 + 4 * c4) : pf32x4{0.f, 0.f, 0.f, 0.f}; \
  }
  if (s0 < s1) { MO_LOAD(s0) }
  const int mine[2] = {wave, 7 - wave};
  for (long long st = s0; st < s1; ++st) {
    pf32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = nv[u];
      if (pre) {
        const bool in_ = st * 32 + r0 + 8 * u < rows;     // (a row past the end stays zero)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = in_ ? fmaxf(__fmaf_rn(v[u][e], psc[e], psh[e]), 0.f) : 0.f;
      }
    }
    if (st + 1 < s1) { MO_LOAD(st + 1) }
    __syncthreads();                                     // the previous step's fragments have been read
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      *reinterpret_cast<pf32x4*>(s_x + (r0 + 8 * u) * MO_PITCH + 4 * c4) = v[u];
      hs += v[u];
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a) {                        // this wave's two channel tiles -> operand fragments
      const int t = mine[a];
      bf16x8 p0, p1, p2;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        __bf16 b0, b1, b2;
        cv_split(s_x[(8 * kg + r) * MO_PITCH + 16 * t + i], b0, b1, b2);
        p0[r] = b0; p1[r] = b1; p2[r] = b2;
      }
      s_f[(t * 3 + 0) * 64 + lane] = __builtin_bit_cast(uint4, p0);
      s_f[(t * 3 + 1) * 64 + lane] = __builtin_bit_cast(uint4, p1);
      s_f[(t * 3 + 2) * 64 + lane] = __builtin_bit_cast(uint4, p2);
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int ta = mine[a];
      bf16x8 fa[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) fa[q] = __builtin_bit_cast(bf16x8, s_f[(ta * 3 + q) * 64 + lane]);
#pragma unroll
      for (int tb = 0; tb < 8; ++tb) {
        if (tb >= ta) {                                  // (uniform in the wave)
          bf16x8 fb[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) fb[q] = __builtin_bit_cast(bf16x8, s_f[(tb * 3 + q) * 64 + lane]);
          pf32x4 t_ = acc[a][tb];
          BF3_MFMA6(t_, fa, fb);
          acc[a][tb] = t_;
        }
      }
    }
  }
#undef MO_LOAD
  // ---- the block's partials: element (((wave * 2 + a) * 8 + tb) * 4 + reg) * 64 + lane = G[16 ta + 4 kg + reg][16 tb + i]
  float* dst = part + (size_t)blockIdx.x * (PN_C2 * PN_C2);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int tb = 0; tb < 8; ++tb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) dst[(((wave * 2 + a) * 8 + tb) * 4 + reg) * 64 + lane] = acc[a][tb][reg];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) s_h[r0][4 * c4 + e] = hs[e];
  __syncthreads();
  if (tid < PN_C2) {
    float t_ = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t_ += s_h[g][tid];
    hpart[(size_t)blockIdx.x * PN_C2 + tid] = t_;
  }
}

__global__ __launch_bounds__(256) void k_rows128_moments_reduce(const float* __restrict__ part, const float* __restrict__ hpart,
                                                                int nblocks, double* __restrict__ G, float* __restrict__ h) {
  // 16 output elements per block, 16 threads per element: thread (el, g) adds the partials of blocks g, g + 16, ..; the groups meet
  // in LDS in a fixed order (one thread per element walking all the blocks was 256 dependent loads deep)
  __shared__ double s_g[16][16];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;                     // output element (ci, cj), or (past the matrix) column sum e - 128 * 128
  double sum = 0;
  if (e < PN_C2 * PN_C2) {
    const int ci = e >> 7, cj = e & 127;
    const int lo = ci <= cj ? ci : cj, hi = ci <= cj ? cj : ci;      // the upper triangle holds it as (lo, hi)
    const int ta = lo >> 4, tb = hi >> 4;
    const int wave = ta < 4 ? ta : 7 - ta, a = ta < 4 ? 0 : 1;
    const int rl = lo & 15, cl = hi & 15;
    // inside a diagonal tile (ta == tb) both triangles are computed: (lo, hi) is as good as (hi, lo)
    const int idx = (((wave * 2 + a) * 8 + tb) * 4 + (rl & 3)) * 64 + 16 * (rl >> 2) + cl;
    const float* src = part + idx;
    int b = g;
    for (; b + 48 < nblocks; b += 64) {
      const float v0 = src[(size_t)b * (PN_C2 * PN_C2)], v1 = src[(size_t)(b + 16) * (PN_C2 * PN_C2)],
                  v2 = src[(size_t)(b + 32) * (PN_C2 * PN_C2)], v3 = src[(size_t)(b + 48) * (PN_C2 * PN_C2)];
      sum = (((sum + (double)v0) + (double)v1) + (double)v2) + (double)v3;
    }
    for (; b < nblocks; b += 16) sum += (double)src[(size_t)b * (PN_C2 * PN_C2)];
  } else if (e < PN_C2 * PN_C2 + PN_C2) {
    for (int b = g; b < nblocks; b += 16) sum += (double)hpart[(size_t)b * PN_C2 + (e - PN_C2 * PN_C2)];
  }
  s_g[g][el] = sum;
  __syncthreads();
  if (g == 0) {
    double t_ = s_g[0][el];
#pragma unroll
    for (int k = 1; k < 16; ++k) t_ += s_g[k][el];
    if (e < PN_C2 * PN_C2) G[e] = t_;
    else if (e < PN_C2 * PN_C2 + PN_C2) h[e - PN_C2 * PN_C2] = (float)t_;
  }
}

extern "C" size_t glx_rows128_moments_workspace_bytes(void) {
  return glx_align((size_t)MO_BLOCKS * (PN_C2 * PN_C2 + PN_C2) * sizeof(float));
}

// G (128 x 128, fp64) = x^T x, h (128, fp32) = the column sums of x (rows, 128); bf16 x 3 products (exact to 2^-22), fp32 sums over a
// block's rows, fp64 over the blocks in a fixed order (bitwise reproducible).  workspace: glx_rows128_moments_workspace_bytes().
extern "C" int glx_rows128_moments(const float* x, long long rows, double* G, float* h, const float* pre_coef, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(x && G && h && workspace && rows >= 0, "glx_rows128_moments: null pointer");
  GLX_REQUIRE(workspace_bytes >= glx_rows128_moments_workspace_bytes(), "glx_rows128_moments: workspace too small");
  const long long nsteps = (rows + 31) >> 5;
  int blocks = (int)(nsteps < MO_BLOCKS ? nsteps : MO_BLOCKS);
  if (blocks < 1) blocks = 1;
  float* part = (float*)workspace;
  float* hpart = part + (size_t)MO_BLOCKS * PN_C2 * PN_C2;
  hipLaunchKernelGGL(k_rows128_moments, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, rows, part, hpart, pre_coef);
  hipLaunchKernelGGL(k_rows128_moments_reduce, dim3((PN_C2 * PN_C2 + PN_C2) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)part,
                     (const float*)hpart, blocks, G, h);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// dh2s[b * P + p, :] = sum over the channels c whose extreme point in object b is p of coef[b, c] * W3[c, :]  (zero rows
// for the other points; `init` (128 floats or NULL) is what every row starts from): the part of  dy W3  that comes from
// the max's gradient.  One block per object; a wave takes
// every fourth point and walks the channels in ascending order (ballots over the object's 512 extreme points in LDS):
// every row is written exactly once, no atomics, fixed summation order.
#define PMS_MASK_WORDS 256
__global__ __launch_bounds__(PN_THREADS) void k_pointmax_scatter(const int* __restrict__ arg, const float* __restrict__ coef,
                                                                 const float* __restrict__ W3, const float* __restrict__ init,
                                                                 int P, float* __restrict__ dh2, int accumulate,
                                                                 const float* __restrict__ chan_scale, const float* __restrict__ z,
                                                                 const float* __restrict__ pre, float* __restrict__ bsum) {
  // bsum != NULL (accumulate mode; z, pre given): dh2 is a gradient with respect to relu(z scale + shift); the object's sums of
  // [z scale + shift > 0] delta  and of  that times z  over the rows it moves (delta = what this launch adds) go to
  // bsum[object][2][128] -- the share of the BatchNorm-backward sums that k_rows128_affine_f16's own sums do not contain yet
  __shared__ int s_arg[PN_C3];
  __shared__ float s_coef[PN_C3];
  __shared__ float s_part[4][4][64];
  __shared__ unsigned s_mask[PMS_MASK_WORDS];           // accumulate: bit p = some channel points at row p (P <= 32 PMS_MASK_WORDS)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long obj = blockIdx.x;
  const bool sparse = accumulate && P <= 32 * PMS_MASK_WORDS;
  float sa1 = 0.f, sa2 = 0.f, sb1 = 0.f, sb2 = 0.f;
  float sc0 = 0.f, sh0 = 0.f, sc1 = 0.f, sh1 = 0.f;
  if (bsum) { sc0 = pre[lane]; sc1 = pre[64 + lane]; sh0 = pre[PN_C2 + lane]; sh1 = pre[PN_C2 + 64 + lane]; }
  if (sparse)
    for (int w = tid; w < PMS_MASK_WORDS; w += PN_THREADS) s_mask[w] = 0u;
  __syncthreads();
  for (int c = tid; c < PN_C3; c += PN_THREADS) {
    const int a = arg[obj * PN_C3 + c];
    const float cf = coef[obj * PN_C3 + c] * (chan_scale ? chan_scale[c] : 1.f);
    s_arg[c] = a;
    s_coef[c] = cf;
    if (sparse && cf != 0.f && a >= 0 && a < P) atomicOr(&s_mask[a >> 5], 1u << (a & 31));
  }
  __syncthreads();
  // the rows p = wave, wave + 4, ... in ascending order; accumulating, only those some channel points at are visited at all (the scan over
  // all P rows x 512 channels was most of the launch: ~150 of 512 rows move)
  unsigned word = sparse ? s_mask[0] & (0x11111111u << wave) : 0u;
  int wi = 0;
  for (int p = wave;; p += PN_THREADS / 64) {
    if (sparse) {
      while (!word) {
        if (++wi >= (P + 31) / 32) break;
        word = s_mask[wi] & (0x11111111u << wave);
      }
      if (!word) break;
      p = 32 * wi + __ffs((int)word) - 1;
      word &= word - 1;
    }
    if (p >= P) break;
    unsigned long long hits[PN_C3 / 64];
    bool touched = false;
#pragma unroll
    for (int k = 0; k < PN_C3 / 64; ++k) {
      hits[k] = __ballot(s_arg[64 * k + lane] == p && s_coef[64 * k + lane] != 0.f);
      touched |= hits[k] != 0;
    }
    float* row = dh2 + (obj * P + p) * PN_C2;
    float a0, a1;                          // channels k = lane and lane + 64 of the row
    float o0 = 0.f, o1 = 0.f, z0 = 0.f, z1 = 0.f;
    if (accumulate) {                      // the rows hold the dense part already: only a row some channel points at moves
      if (!touched) continue;
      a0 = o0 = row[lane];
      a1 = o1 = row[64 + lane];
      if (bsum) {                          // (in the same round trip as the row itself)
        const float* zr = z + (obj * P + p) * PN_C2;
        z0 = zr[lane];
        z1 = zr[64 + lane];
      }
    } else {
      a0 = init ? init[lane] : 0.f;
      a1 = init ? init[64 + lane] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < PN_C3 / 64; ++k) {
      unsigned long long hit = hits[k];
      while (hit) {
        const int c = 64 * k + __ffsll((long long)hit) - 1;
        hit &= hit - 1;
        const float w = s_coef[c];
        a0 = fmaf(w, W3[(long long)c * PN_C2 + lane], a0);
        a1 = fmaf(w, W3[(long long)c * PN_C2 + 64 + lane], a1);
      }
    }
    row[lane] = a0;
    row[64 + lane] = a1;
    if (bsum) {
      if (__fmaf_rn(z0, sc0, sh0) > 0.f) { const float d = a0 - o0; sa1 += d; sa2 += d * z0; }
      if (__fmaf_rn(z1, sc1, sh1) > 0.f) { const float d = a1 - o1; sb1 += d; sb2 += d * z1; }
    }
  }
  if (bsum) {
    s_part[wave][0][lane] = sa1; s_part[wave][1][lane] = sb1; s_part[wave][2][lane] = sa2; s_part[wave][3][lane] = sb2;
    __syncthreads();
    if (tid < 2 * PN_C2) {        // tid = moment * 128 + channel; channel = 64 half + lane
      const int k = tid >> 6, l = tid & 63;
      bsum[obj * 2 * PN_C2 + tid] = ((s_part[0][k][l] + s_part[1][k][l]) + s_part[2][k][l]) + s_part[3][k][l];
    }
  }
}

// The BatchNorm backward's vectors from sums that their producers took (k_rows128_affine_f16, k_pointmax_scatter): pa (na, 2, C) and
// pb (nb, 2, C) hold partial sums of dz (moment 0) and dz z (moment 1, z the BatchNorm's raw input); one block per channel adds them in a
// fixed order in double and finishes as glx_bn_backward_sums does:  sum dz xhat = invstd (sum dz z - mean sum dz),
// coef3 = (gamma invstd, sum dz / rows, sum dz xhat / rows), dgamma = sum dz xhat, dbeta = sum dz.
__global__ __launch_bounds__(256) void k_bn_bwd_from_partials(const float* __restrict__ pa, int na, const float* __restrict__ pb, int nb,
                                                              int C, double rows, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef3) {
  __shared__ double red[256];
  const int c = blockIdx.x, m = threadIdx.x >> 7, t = threadIdx.x & 127;
  double s = 0.0;
  for (int i = t; i < na; i += 128) s += (double)pa[((long long)i * 2 + m) * C + c];
  for (int i = t; i < nb; i += 128) s += (double)pb[((long long)i * 2 + m) * C + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (t < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double s1 = red[0], s2z = red[128];
    const double ss = (double)invstd[c] * (s2z - (double)mean[c] * s1);
    coef3[c] = (gamma ? gamma[c] : 1.f) * invstd[c];
    coef3[C + c] = (float)(s1 / rows);
    coef3[2 * C + c] = (float)(ss / rows);
    if (dgamma) dgamma[c] = (float)ss;
    if (dbeta) dbeta[c] = (float)s1;
  }
}
extern "C" int glx_bn_backward_from_partials(const float* pa, int na, const float* pb, int nb, int C, long long rows, const float* gamma,
                                             const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef3,
                                             void* stream) {
  GLX_REQUIRE(C >= 1 && rows >= 1 && mean && invstd && coef3 && (na == 0 || pa) && (nb == 0 || pb),
              "glx_bn_backward_from_partials: null pointer / empty problem");
  hipLaunchKernelGGL(k_bn_bwd_from_partials, dim3(C), dim3(256), 0, (hipStream_t)stream, pa, na, pb, nb, C, (double)rows, gamma, mean,
                     invstd, dgamma, dbeta, coef3);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

static int pointmax_scatter(const int32_t* arg, const float* coef, const float* W3, const float* init, int B, int P, float* dh2,
                            int accumulate, void* stream, const float* chan_scale = nullptr, const float* z = nullptr,
                            const float* pre = nullptr, float* bsum = nullptr) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(arg && coef && W3 && dh2 && P >= 1, "glx_pointmax_scatter: null pointer");
  GLX_REQUIRE(!bsum || (accumulate && z && pre), "glx_pointmax_scatter: the sums need the accumulating form, z and pre_coef");
  hipLaunchKernelGGL(k_pointmax_scatter, dim3(B), dim3(PN_THREADS), 0, (hipStream_t)stream, (const int*)arg, coef, W3, init, P, dh2,
                     accumulate, chan_scale, z, pre, bsum);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_pointmax_scatter(const int32_t* arg, const float* coef, const float* W3, const float* init, int B, int P,
                                    float* dh2, void* stream) {
  return pointmax_scatter(arg, coef, W3, init, B, P, dh2, 0, stream);
}

// ... added to what dh2 holds (the dense part from glx_rows128_affine_f16x2): rows no channel points at are left alone.
extern "C" int glx_pointmax_scatter_add(const int32_t* arg, const float* coef, const float* W3, int B, int P, float* dh2,
                                        void* stream) {
  return pointmax_scatter(arg, coef, W3, nullptr, B, P, dh2, 1, stream);
}

// ------------------------------------------------------------------------------------------------ the BatchNorm around the max, fused
// What dense_path.PointMaxBN did with ~40 tensor statements per direction (each a ~4.5 us launch inside the recorded step):
//   forward   k_pm_stats: per channel c  mean = W3[c] . h / R,  var = W3[c] G W3[c]^T / R - mean^2  in fp64 (G = h2^T h2, h = sum h2),
//             invstd, scale = gamma invstd, the running statistics (bias folded into the running mean);
//             k_pm_out: ext = sign(gamma) vext (the pass returned max_p (sign y)), out = (ext - mean) scale + beta;
//   backward  k_pm_bwd_sums: dbeta = sum_b g, dgamma = sum_b g xhat, bvec = scale dbeta / R, cvec = scale invstd dgamma / R;
//             k_pm_bwd_mats: v = (bvec - cvec mean) W3 (negated: the dense pass's init), M = W3^T diag(cvec) W3;
//             k_pm_bwd_dw: dW3 = scale T - bvec (x) h - diag(cvec) (W3 G - mean (x) h).
__global__ __launch_bounds__(PN_C2) void k_pm_stats(const float* __restrict__ W3, const double* __restrict__ G, const float* __restrict__ h,
                                                    long long R, const float* __restrict__ gamma, const float* __restrict__ bias, float eps,
                                                    float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale) {
  __shared__ double s_w[PN_C2], s_r[2][PN_C2];
  const int c = blockIdx.x, i = threadIdx.x;
  s_w[i] = (double)W3[(long long)c * PN_C2 + i];
  __syncthreads();
  double t = 0;
  for (int jj = 0; jj < PN_C2; ++jj) t += G[i * PN_C2 + jj] * s_w[jj];
  s_r[0][i] = s_w[i] * (double)h[i];
  s_r[1][i] = s_w[i] * t;
  __syncthreads();
  for (int o = PN_C2 / 2; o > 0; o >>= 1) {
    if (i < o) { s_r[0][i] += s_r[0][i + o]; s_r[1][i] += s_r[1][i + o]; }
    __syncthreads();
  }
  if (i == 0) {
    const double m = s_r[0][0] / (double)R;
    double var = s_r[1][0] / (double)R - m * m;
    if (var < 0) var = 0;
    const float mf = (float)m, is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = mf;
    invstd[c] = is;
    scale[c] = gamma[c] * is;
    if (running_mean) {
      const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (mf + (bias ? bias[c] : 0.f));
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

__global__ __launch_bounds__(256) void k_pm_out(float* __restrict__ vext, int B, const float* __restrict__ gamma,
                                                const float* __restrict__ mean, const float* __restrict__ scale,
                                                const float* __restrict__ beta, float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)B * PN_C3) return;
  const int c = (int)(e & (PN_C3 - 1));
  const float ext = gamma[c] >= 0.f ? vext[e] : -vext[e];
  vext[e] = ext;
  out[e] = (ext - mean[c]) * scale[c] + beta[c];
}

// a block takes 4 channels: thread (cl = tid & 3, rb = tid >> 2) sums rows rb, rb + 64, ..
__global__ __launch_bounds__(256) void k_pm_bwd_sums(const float* __restrict__ g, const float* __restrict__ ext, int B, long long R,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ scale, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ bvec, float* __restrict__ cvec) {
  __shared__ double s_r[2][256];
  const int cl = threadIdx.x & 3, rb = threadIdx.x >> 2, c = blockIdx.x * 4 + cl;
  const float mu = mean[c], is = invstd[c];
  double a0 = 0, a1 = 0;
  for (int b = rb; b < B; b += 64) {
    const float gv = g[(long long)b * PN_C3 + c];
    a0 += (double)gv;
    a1 += (double)(gv * ((ext[(long long)b * PN_C3 + c] - mu) * is));
  }
  s_r[0][threadIdx.x] = a0;
  s_r[1][threadIdx.x] = a1;
  __syncthreads();
  for (int o = 128; o >= 4; o >>= 1) {
    if ((int)threadIdx.x < o) { s_r[0][threadIdx.x] += s_r[0][threadIdx.x + o]; s_r[1][threadIdx.x] += s_r[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x < 4) {
    const float db = (float)s_r[0][threadIdx.x], dg = (float)s_r[1][threadIdx.x];
    dbeta[c] = db;
    dgamma[c] = dg;
    bvec[c] = scale[c] * db / (float)R;
    cvec[c] = scale[c] * is * dg / (float)R;
  }
}

// block j < 128: M[j][i] = sum_c W3[c][i] cvec[c] W3[c][j]; block 128: nv[i] = -sum_c (bvec[c] - cvec[c] mean[c]) W3[c][i]
__global__ __launch_bounds__(PN_C2) void k_pm_bwd_mats(const float* __restrict__ W3, const float* __restrict__ bvec,
                                                       const float* __restrict__ cvec, const float* __restrict__ mean,
                                                       float* __restrict__ M, float* __restrict__ nv) {
  const int jb = blockIdx.x, i = threadIdx.x;
  float acc = 0.f;
  if (jb < PN_C2) {
    for (int c = 0; c < PN_C3; ++c) acc = fmaf(W3[c * PN_C2 + i], cvec[c] * W3[c * PN_C2 + jb], acc);
    M[jb * PN_C2 + i] = acc;
  } else {
    for (int c = 0; c < PN_C3; ++c) acc = fmaf(bvec[c] - cvec[c] * mean[c], W3[c * PN_C2 + i], acc);
    nv[i] = -acc;
  }
}

// block c: dW3[c][k] = scale[c] T[c][k] - bvec[c] h[k] - cvec[c] ((W3 G)[c][k] - mean[c] h[k])
__global__ __launch_bounds__(PN_C2) void k_pm_bwd_dw(const float* __restrict__ W3, const double* __restrict__ G, const float* __restrict__ h,
                                                     const float* __restrict__ T, const float* __restrict__ scale,
                                                     const float* __restrict__ bvec, const float* __restrict__ cvec,
                                                     const float* __restrict__ mean, float* __restrict__ dW) {
  __shared__ float s_w[PN_C2];
  const int c = blockIdx.x, k = threadIdx.x;
  s_w[k] = W3[c * PN_C2 + k];
  __syncthreads();
  float wg = 0.f;
  for (int i = 0; i < PN_C2; ++i) wg = fmaf(s_w[i], (float)G[i * PN_C2 + k], wg);
  dW[c * PN_C2 + k] = scale[c] * T[c * PN_C2 + k] - bvec[c] * h[k] - cvec[c] * (wg - mean[c] * h[k]);
}

// vext (B, 512): what glx_pointmax_forward_f16x2 returned (max_p (sign(gamma) y)), REPLACED by ext = the extreme of y the BatchNorm
// uses; out (B, 512); mean / invstd / scale: 512 each (saved for the backward); running_*: NULL = not tracked.
extern "C" int glx_pointmax_bn_forward(const float* W3, const double* G, const float* h, long long R, float* vext, int B,
                                       const float* gamma, const float* beta, const float* bias, float eps, float momentum,
                                       float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* out,
                                       void* stream) {
  GLX_REQUIRE(W3 && G && h && vext && gamma && beta && mean && invstd && scale && out && R > 0 && B > 0, "glx_pointmax_bn_forward: bad arguments");
  GLX_REQUIRE(!running_mean == !running_var, "glx_pointmax_bn_forward: both running statistics or neither");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pm_stats, dim3(PN_C3), dim3(PN_C2), 0, st, W3, G, h, R, gamma, bias, eps, momentum, running_mean, running_var, mean,
                     invstd, scale);
  hipLaunchKernelGGL(k_pm_out, dim3((unsigned)(((long long)B * PN_C3 + 255) / 256)), dim3(256), 0, st, vext, B, gamma, (const float*)mean,
                     (const float*)scale, beta, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// g (B, 512): the gradient of out; ext / mean / invstd / scale: from glx_pointmax_bn_forward.  -> dgamma, dbeta, bvec, cvec (512 each),
// M (128 x 128) = W3^T diag(cvec) W3 and nv (128) = -(bvec - cvec mean) W3: the dense part of the input gradient is nv - h2 M.
extern "C" int glx_pointmax_bn_backward_sums(const float* g, const float* ext, int B, long long R, const float* W3, const float* mean,
                                             const float* invstd, const float* scale, float* dgamma, float* dbeta, float* bvec,
                                             float* cvec, float* M, float* nv, void* stream) {
  GLX_REQUIRE(g && ext && W3 && mean && invstd && scale && dgamma && dbeta && bvec && cvec && M && nv && B > 0 && R > 0,
              "glx_pointmax_bn_backward_sums: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_pm_bwd_sums, dim3(PN_C3 / 4), dim3(256), 0, st, g, ext, B, R, mean, invstd, scale, dgamma, dbeta, bvec, cvec);
  hipLaunchKernelGGL(k_pm_bwd_mats, dim3(PN_C2 + 1), dim3(PN_C2), 0, st, W3, (const float*)bvec, (const float*)cvec, mean, M, nv);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// dW3 (512, 128) from T (glx_pointmax_wsum of the UNSCALED g), the moments G (fp64) / h and the vectors of the sums call.
extern "C" int glx_pointmax_bn_backward_weight(const float* W3, const double* G, const float* h, const float* T, const float* scale,
                                               const float* bvec, const float* cvec, const float* mean, float* dW, void* stream) {
  GLX_REQUIRE(W3 && G && h && T && scale && bvec && cvec && mean && dW, "glx_pointmax_bn_backward_weight: null pointer");
  hipLaunchKernelGGL(k_pm_bwd_dw, dim3(PN_C3), dim3(PN_C2), 0, (hipStream_t)stream, W3, G, h, T, scale, bvec, cvec, mean, dW);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// glx_pointmax_scatter_add with the coefficients scaled per channel (coef[b, c] chan_scale[c]): the caller hands in g as it is
extern "C" int glx_pointmax_scatter_add_scaled(const int32_t* arg, const float* coef, const float* chan_scale, const float* W3, int B, int P,
                                               float* dh2, void* stream) {
  return pointmax_scatter(arg, coef, W3, nullptr, B, P, dh2, 1, stream, chan_scale);
}
extern "C" int glx_pointmax_scatter_add_scaled_sums(const int32_t* arg, const float* coef, const float* chan_scale, const float* W3, int B,
                                                    int P, float* dh2, const float* z, const float* pre_coef, float* bsum, void* stream) {
  return pointmax_scatter(arg, coef, W3, nullptr, B, P, dh2, 1, stream, chan_scale, z, pre_coef, bsum);
}

// T[c, :] = sum_b g[b, c] * h2[b * P + arg[b, c], :]   (512 x 128): the max's part of the weight gradient.  A wave per
// (channel, slice of the objects), 8 row gathers in flight; slices are summed in order by the second kernel.
#define PMW_SLICES 8
__global__ __launch_bounds__(PN_THREADS) void k_pointmax_wsum(const float* __restrict__ g, const int* __restrict__ arg,
                                                              const float* __restrict__ h2, int B, int P,
                                                              float* __restrict__ part, const float* __restrict__ pre) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // pre != NULL: the rows are relu(h2 scale + shift) on load
  const float sc0 = pre ? pre[lane] : 1.f, sc1 = pre ? pre[64 + lane] : 1.f;
  const float sh0 = pre ? pre[PN_C2 + lane] : 0.f, sh1 = pre ? pre[PN_C2 + 64 + lane] : 0.f;
  const int c = blockIdx.x * (PN_THREADS / 64) + wave, sl = blockIdx.y;
  const int per = (B + PMW_SLICES - 1) / PMW_SLICES, b0 = sl * per, b1 = min(B, b0 + per);
  float a0 = 0.f, a1 = 0.f;
  // the next batch's points and weights are requested a batch ahead (a row's address depends on arg: two latencies in a row otherwise)
  float wn[8];
  int an[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int bb = min(b0 + u, b1 - 1);
    const long long o = (long long)(b0 < b1 ? bb : 0) * PN_C3 + c;
    wn[u] = b0 + u < b1 ? g[o] : 0.f;
    an[u] = b0 < b1 ? arg[o] : 0;
  }
  for (int b = b0; b < b1; b += 8) {
    float w[8], x0[8], x1[8];
    int ap[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { w[u] = wn[u]; ap[u] = an[u]; }
    if (b + 8 < b1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int bb = min(b + 8 + u, b1 - 1);
        const long long o = (long long)bb * PN_C3 + c;
        wn[u] = b + 8 + u < b1 ? g[o] : 0.f;
        an[u] = arg[o];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int bb = min(b + u, b1 - 1);
      const float* row = h2 + ((long long)bb * P + ap[u]) * PN_C2;
      x0[u] = row[lane];
      x1[u] = row[64 + lane];
      if (pre) {
        x0[u] = fmaxf(__fmaf_rn(x0[u], sc0, sh0), 0.f);
        x1[u] = fmaxf(__fmaf_rn(x1[u], sc1, sh1), 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = fmaf(w[u], x0[u], a0);
      a1 = fmaf(w[u], x1[u], a1);
    }
  }
  float* dst = part + ((long long)sl * PN_C3 + c) * PN_C2;
  dst[lane] = a0;
  dst[64 + lane] = a1;
}

__global__ void k_pointmax_wsum_reduce(const float* __restrict__ part, float* __restrict__ T) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= PN_C3 * PN_C2) return;
  float s = 0.f;
#pragma unroll
  for (int sl = 0; sl < PMW_SLICES; ++sl) s += part[(long long)sl * PN_C3 * PN_C2 + e];
  T[e] = s;
}

extern "C" size_t glx_pointmax_wsum_workspace_bytes(void) { return (size_t)PMW_SLICES * PN_C3 * PN_C2 * sizeof(float); }

extern "C" int glx_pointmax_wsum_pre(const float* g, const int32_t* arg, const float* h2, const float* pre_coef, int B, int P, float* T,
                                     void* workspace, size_t workspace_bytes, void* stream);
extern "C" int glx_pointmax_wsum(const float* g, const int32_t* arg, const float* h2, int B, int P, float* T,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  return glx_pointmax_wsum_pre(g, arg, h2, nullptr, B, P, T, workspace, workspace_bytes, stream);
}

// ... of the rows relu(h2 scale + shift) (pre_coef = scale | shift, 128 each; NULL: of h2 itself)
extern "C" int glx_pointmax_wsum_pre(const float* g, const int32_t* arg, const float* h2, const float* pre_coef, int B, int P, float* T,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(g && arg && h2 && T && workspace && B >= 1 && P >= 1, "glx_pointmax_wsum: null pointer / empty batch");
  GLX_REQUIRE(workspace_bytes >= glx_pointmax_wsum_workspace_bytes(), "glx_pointmax_wsum: workspace too small");
  hipLaunchKernelGGL(k_pointmax_wsum, dim3(PN_C3 / (PN_THREADS / 64), PMW_SLICES), dim3(PN_THREADS), 0, (hipStream_t)stream, g,
                     (const int*)arg, h2, B, P, (float*)workspace, pre_coef);
  hipLaunchKernelGGL(k_pointmax_wsum_reduce, dim3(PN_C3 * PN_C2 / 256), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, T);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
