// Sparse convolution (SubMConv3d / SparseConv3d) on CDNA4: output-stationary implicit GEMM.
//
//   out[j, :] = sum_k in[nbr[j, k], :] @ W[k]            W: (K, Cin, Cout) fp32
//
// Semantics: spconv SubMConv3d / SparseConv3d forward as called from
// pcdet/models/backbones_3d/spconv_backbone.py:148-156 (third-party arithmetic; the
// algorithm is the published gather-GEMM-scatter, restated output-stationary so the
// scatter-add disappears).  One wave owns a 16-row output tile; for each kernel offset
// present in the tile it gathers the 16 neighbour rows straight into registers (each
// lane loads a contiguous Cin/4 slice, so a row is one or two full cache lines) and
// multiplies by W[k], which sits in LDS in MFMA-fragment order, with
// v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain).  Offsets absent from the
// whole tile are skipped, so the dense MFMA work tracks the rule count R.
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <type_traits>

#include "glx_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SC_MAXK 27

template <int CIN, int COUT>
struct SconvCfg {
  static constexpr int CQ = CIN / 4;            // input channels held per lane quad
  static constexpr int NT = COUT / 16;          // 16-wide output column tiles
  static constexpr int NC = NT >= 4 ? 4 : NT;   // floats per LDS B read
  static constexpr int NH = NT / NC;            // B reads per k-step
  static constexpr int QPAD = NC == 4 ? 0 : (NC == 2 ? 32 : 16);  // bank de-phasing
  static constexpr int QSTRIDE = CQ * 16 * NC + QPAD;             // dwords per (h,q)
  static constexpr int IMG = NH * 4 * QSTRIDE;                     // dwords per offset
};

// W (K, CIN, COUT) -> per-offset LDS images in MFMA fragment order:
//   img[(h*4+q)*QSTRIDE + (t*16+n)*NC + c] = W[k][q*CQ+t][(h*NC+c)*16+n]
template <int CIN, int COUT>
__global__ void k_pack_weights(const float* __restrict__ W, int K, float* __restrict__ Wp) {
  using C = SconvCfg<CIN, COUT>;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= K * CIN * COUT) return;
  int co = e % COUT;
  int ci = (e / COUT) % CIN;
  int k = e / (COUT * CIN);
  int q = ci / C::CQ, t = ci % C::CQ;
  int ct = co / 16, n = co % 16;
  int h = ct / C::NC, c = ct % C::NC;
  Wp[(size_t)k * C::IMG + (h * 4 + q) * C::QSTRIDE + (t * 16 + n) * C::NC + c] = W[e];
}

template <int N>
struct FVec;
template <>
struct FVec<1> { typedef float T; };
template <>
struct FVec<2> { typedef float2 T; };
template <>
struct FVec<4> { typedef float4 T; };

// Epilogue fused into the store of the output tile: y = relu?((acc + bias) * scale + shift).
struct SconvEpilogue {
  const float* bias;   // (Cout) or NULL
  const float* scale;  // (Cout) or NULL  (e.g. eval-mode BatchNorm folded: gamma / sqrt(var+eps))
  const float* shift;  // (Cout) or NULL
  int relu;
  const int* n_live;   // NULL, or device int32: rows >= *n_live are neither computed nor written
};

// Tile geometry: TR output rows per block, NW waves per block, NBUF LDS weight buffers.
template <int CIN, int COUT, int TR_, int NW_, int NBUF_>
struct SconvTile {
  using C = SconvCfg<CIN, COUT>;
  static constexpr int TR = TR_, NW = NW_, NBUF = NBUF_;
  static constexpr int THREADS = NW * 64;
  static constexpr int MAXC = (TR + 16 * NW - 1) / (16 * NW);   // chunks per wave per offset
  static constexpr int ACC_LD = COUT + 4;
  static constexpr int LW = (TR + 63) / 64;                     // waves that own row slots
  static constexpr size_t lds_bytes = (size_t)TR * ACC_LD * 4 + (size_t)SC_MAXK * TR * 5 +
                                      (TR + 32 + 32 * LW) * 4 + 64 + (size_t)NBUF * C::IMG * 4;
  static_assert(LW <= NW, "row-slot waves exceed block");
  static_assert(TR <= 256, "row slots are stored as bytes");
};

// gather the CQ-float slice of up to MAXC chunks of offset k owned by this wave
template <int CIN, int COUT, class T>
__device__ __forceinline__ void sc_gather(const float* __restrict__ in, const int* s_pin,
                                          int k, int cnt, int wave, int r, int q,
                                          float (&A)[T::MAXC][SconvCfg<CIN, COUT>::CQ],
                                          bool (&valid)[T::MAXC]) {
  using C = SconvCfg<CIN, COUT>;
#pragma unroll
  for (int j = 0; j < T::MAXC; ++j) {
    const int c = ((wave - k) & (T::NW - 1)) + j * T::NW;
    const int p = c * 16 + r;
    int irow = -1;
    if (p < cnt) irow = s_pin[k * T::TR + p];
    const float* ap = in + (long long)(irow < 0 ? 0 : irow) * CIN + q * C::CQ;
    if (c * 16 < cnt) {   // wave-uniform: skip the loads of an absent chunk
      if constexpr (C::CQ % 4 == 0) {
#pragma unroll
        for (int i = 0; i < C::CQ / 4; ++i) {
          f32x4 v = reinterpret_cast<const f32x4*>(ap)[i];
          A[j][4 * i + 0] = v[0]; A[j][4 * i + 1] = v[1]; A[j][4 * i + 2] = v[2]; A[j][4 * i + 3] = v[3];
        }
      } else {
#pragma unroll
        for (int i = 0; i < C::CQ; ++i) A[j][i] = ap[i];
      }
    }
    valid[j] = irow >= 0;   // zero-fill is applied at use: writing A here would stall on vmcnt(0)
  }
}

// MFMA the wave's chunks of offset k against the staged W[k] and add into the LDS tile
template <int CIN, int COUT, class T>
__device__ __forceinline__ void sc_compute(const float* s_w, float* s_acc,
                                           const unsigned char* s_pslot, int k, int cnt, int wave,
                                           int r, int q,
                                           const float (&A)[T::MAXC][SconvCfg<CIN, COUT>::CQ],
                                           const bool (&valid)[T::MAXC]) {
  using C = SconvCfg<CIN, COUT>;
#pragma unroll
  for (int j = 0; j < T::MAXC; ++j) {
    const int c = ((wave - k) & (T::NW - 1)) + j * T::NW;
    if (c * 16 >= cnt) continue;   // wave-uniform
    float Am[C::CQ];
#pragma unroll
    for (int i = 0; i < C::CQ; ++i) Am[i] = valid[j] ? A[j][i] : 0.f;
    // Operands swapped (W^T as the MFMA "A", gathered rows as "B"): D[i = cout][j = pair], so
    // lane (pair r, q) ends up with 4 CONSECUTIVE output channels 16ct + 4q .. +3 of its pair.
    f32x4 acc[C::NT];
#pragma unroll
    for (int ct = 0; ct < C::NT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < C::CQ; ++t) {
#pragma unroll
      for (int h = 0; h < C::NH; ++h) {
        const float* bp = s_w + (h * 4 + q) * C::QSTRIDE + (t * 16 + r) * C::NC;
        if constexpr (C::NC == 4) {
          f32x4 bv = *reinterpret_cast<const f32x4*>(bp);
          acc[h * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[0], Am[t], acc[h * 4 + 0], 0, 0, 0);
          acc[h * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[1], Am[t], acc[h * 4 + 1], 0, 0, 0);
          acc[h * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[2], Am[t], acc[h * 4 + 2], 0, 0, 0);
          acc[h * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[3], Am[t], acc[h * 4 + 3], 0, 0, 0);
        } else if constexpr (C::NC == 2) {
          float2 bv = *reinterpret_cast<const float2*>(bp);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv.x, Am[t], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv.y, Am[t], acc[1], 0, 0, 0);
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bp[0], Am[t], acc[0], 0, 0, 0);
        }
      }
    }
    // plain read-modify-write of the LDS tile (no atomics: rows of one offset are distinct and
    // offsets are separated by the block barrier; LDS float atomics are far slower than this)
    const int p = c * 16 + r;
    if (p < cnt) {
      float* dst = s_acc + (int)s_pslot[k * T::TR + p] * T::ACC_LD + 4 * q;
#pragma unroll
      for (int ct = 0; ct < C::NT; ++ct) {
        f32x4 v = *reinterpret_cast<f32x4*>(dst + ct * 16);
        v += acc[ct];
        *reinterpret_cast<f32x4*>(dst + ct * 16) = v;
      }
    }
  }
}

// Block = 8 waves, TR output rows.  Per kernel offset k the block compacts the rows that have
// a neighbour at k into a pair list; 16 pairs form one MFMA row tile (the matrix pipe only sees
// real rules), products are added into an fp32 accumulator tile in LDS.  Rows of one offset are
// distinct and offsets are separated by a barrier, so every output element is summed in a fixed
// order: bitwise reproducible, no global atomics.  W[k+1] streams into the second LDS buffer
// and the next offset's input rows into registers while offset k multiplies.
template <int CIN, int COUT, int TR_, int NW_, int NBUF_>
__global__ __launch_bounds__(NW_ * 64) void k_sconv_mfma(
    const float* __restrict__ in, const float* __restrict__ Wp, SconvEpilogue ep,
    const int* __restrict__ nbr, const int* __restrict__ tile_order, int N_out, int K,
    float* __restrict__ out) {
  if (ep.n_live) N_out = min(N_out, *ep.n_live);   // device-side row count (capacity launch)
  using C = SconvCfg<CIN, COUT>;
  using T = SconvTile<CIN, COUT, TR_, NW_, NBUF_>;
  constexpr int TR = T::TR, ACC_LD = T::ACC_LD, NBUF = T::NBUF, LW = T::LW;
  constexpr int SC_THREADS = T::THREADS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_acc = smem;                                        // TR * ACC_LD
  float* s_w = s_acc + TR * ACC_LD;                           // NBUF * IMG
  int* s_pin = reinterpret_cast<int*>(s_w + NBUF * C::IMG);   // SC_MAXK * TR
  int* s_rows = s_pin + SC_MAXK * TR;                         // TR
  int* s_cnt = s_rows + TR;                                   // 32
  int* s_wcnt = s_cnt + 32;                                   // LW * 32
  unsigned char* s_pslot = reinterpret_cast<unsigned char*>(s_wcnt + LW * 32);  // SC_MAXK * TR

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * TR;

  // ---- tile rows, zero accumulators, first weight image
  int my_row = -1;
  if (tid < TR) {
    int p = row0 + tid;
    my_row = (p < N_out) ? (tile_order ? tile_order[p] : p) : -1;
    s_rows[tid] = my_row;
  }
  for (int i = tid; i < TR * ACC_LD / 4; i += SC_THREADS)
    reinterpret_cast<f32x4*>(s_acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- neighbour row of every (slot, offset) straight into registers, then compaction
  int nb[SC_MAXK];
  if (tid < TR) {
#pragma unroll
    for (int k = 0; k < SC_MAXK; ++k)
      nb[k] = (k < K && my_row >= 0) ? nbr[(long long)my_row * K + k] : -1;
#pragma unroll
    for (int k = 0; k < SC_MAXK; ++k) {
      unsigned long long b = __ballot(nb[k] >= 0);
      if (lane == 0) s_wcnt[wave * 32 + k] = __popcll(b);
    }
  }
  __syncthreads();
  if (tid < TR) {
#pragma unroll
    for (int k = 0; k < SC_MAXK; ++k) {
      const bool v = nb[k] >= 0;
      unsigned long long b = __ballot(v);
      int base = 0;
#pragma unroll
      for (int w = 0; w < LW; ++w) base += (w < wave) ? s_wcnt[w * 32 + k] : 0;
      if (v) {
        int pos = k * TR + base + __popcll(b & ((1ull << lane) - 1ull));
        s_pin[pos] = nb[k];
        s_pslot[pos] = (unsigned char)tid;
      }
      if (wave == LW - 1 && lane == 0) s_cnt[k] = base + __popcll(b);
    }
  }
  __syncthreads();
  unsigned mask = 0;
#pragma unroll
  for (int k = 0; k < SC_MAXK; ++k)
    if (k < K && s_cnt[k] > 0) mask |= 1u << k;
  mask = __builtin_amdgcn_readfirstlane(mask);

  // ---- weight staging registers (next image) and input-row registers (ping-pong)
  constexpr int STAGE_F4 = C::IMG / 4;
  constexpr int SPT = (STAGE_F4 + SC_THREADS - 1) / SC_THREADS;
  f32x4 stage_regs[SPT];
#define SC_STAGE_LOAD(KK)                                                                   \
  {                                                                                         \
    const f32x4* src_ = reinterpret_cast<const f32x4*>(Wp + (size_t)(KK) * C::IMG);         \
    _Pragma("unroll") for (int i_ = 0; i_ < SPT; ++i_) {                                    \
      int e_ = tid + i_ * SC_THREADS;                                                       \
      stage_regs[i_] = src_[e_ < STAGE_F4 ? e_ : STAGE_F4 - 1];                             \
    }                                                                                       \
  }
#define SC_STAGE_STORE(BUF)                                                                 \
  {                                                                                         \
    f32x4* dst_ = reinterpret_cast<f32x4*>(s_w + (BUF) * C::IMG);                           \
    _Pragma("unroll") for (int i_ = 0; i_ < SPT; ++i_) {                                    \
      int e_ = tid + i_ * SC_THREADS;                                                       \
      if (STAGE_F4 % SC_THREADS == 0 || e_ < STAGE_F4) dst_[e_] = stage_regs[i_];           \
    }                                                                                       \
  }

  float A0[T::MAXC][C::CQ], A1[T::MAXC][C::CQ];
  bool V0[T::MAXC], V1[T::MAXC];
  unsigned rem = mask;
  int k = -1, cnt = 0;
  if (rem) {
    k = __builtin_ctz(rem);
    rem &= rem - 1;
    cnt = s_cnt[k];
    SC_STAGE_LOAD(k);
    SC_STAGE_STORE(0);
    sc_gather<CIN, COUT, T>(in, s_pin, k, cnt, wave, r, q, A0, V0);
  }
  __syncthreads();

  int buf = 0;
  // one phase: prefetch (W image + input rows) of the next offset, multiply the current one
#define SC_PHASE(CUR, NXT, VCUR, VNXT)                                                                 \
  {                                                                                         \
    int kn_ = -1, cntn_ = 0;                                                                \
    if (rem) {                                                                              \
      kn_ = __builtin_ctz(rem);                                                             \
      rem &= rem - 1;                                                                       \
      cntn_ = s_cnt[kn_];                                                                   \
      SC_STAGE_LOAD(kn_);                                                                   \
      sc_gather<CIN, COUT, T>(in, s_pin, kn_, cntn_, wave, r, q, NXT, VNXT);                         \
    }                                                                                       \
    sc_compute<CIN, COUT, T>(s_w + (NBUF == 2 ? buf : 0) * C::IMG, s_acc, s_pslot, k, cnt,     \
                          wave, r, q, CUR, VCUR);                                           \
    if (NBUF == 1) __syncthreads();                                                         \
    if (kn_ >= 0) SC_STAGE_STORE(NBUF == 2 ? (buf ^ 1) : 0);                                \
    __syncthreads();                                                                        \
    buf ^= 1;                                                                               \
    k = kn_;                                                                                \
    cnt = cntn_;                                                                            \
  }
  while (k >= 0) {
    SC_PHASE(A0, A1, V0, V1);
    if (k < 0) break;
    SC_PHASE(A1, A0, V1, V0);
  }
#undef SC_PHASE
#undef SC_STAGE_LOAD
#undef SC_STAGE_STORE

  // ---- epilogue: coalesced row stores with the fused pointwise tail
  constexpr int C4 = COUT / 4;
  for (int i = tid; i < TR * C4; i += SC_THREADS) {
    int rr = i / C4, c4 = i - rr * C4;
    int orow = s_rows[rr];
    if (orow < 0) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(s_acc + rr * ACC_LD + 4 * c4);
    const int co = 4 * c4;
    if (ep.bias) v += *reinterpret_cast<const f32x4*>(ep.bias + co);
    if (ep.scale) v *= *reinterpret_cast<const f32x4*>(ep.scale + co);
    if (ep.shift) v += *reinterpret_cast<const f32x4*>(ep.shift + co);
    if (ep.relu) {
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + (long long)orow * COUT + 4 * c4) = v;
  }
}

// ==================================================================== wave-private kernel
// Default sparse-conv kernel.  Every wave owns TRW consecutive output rows and everything
// about them (rule compaction, accumulator tile, epilogue), so the kernel has NO barriers and
// no shared state between waves: a wave streams W[k] fragments from L2 straight into
// registers (prefetched one phase ahead), gathers 16 rule pairs at a time (prefetched one
// chunk ahead), multiplies them on the matrix pipe and adds the 16x(16*NTG) result into its
// private LDS tile with plain ds_read/ds_write (in-order within a wave, so the summation order
// of every output element is fixed: bitwise reproducible).
template <int CIN, int COUT>
struct WpCfg {
  static constexpr int CQ = CIN / 4;
  static constexpr int CG0 = 4096 / CIN;                              // cols per W register set
  static constexpr int CG = COUT < CG0 ? COUT : (CG0 < 16 ? 16 : CG0);
  static constexpr int NG = COUT / CG;                                 // column groups
  static constexpr int NTG = CG / 16;                                  // 16-col tiles per group
  static constexpr int TRW = 48;                                       // rows per wave
  static constexpr int NWB = 4;                                        // waves per block
  static constexpr int ACC_LD = COUT + 4;
  static constexpr int WAVE_LDS_DW =
      TRW * ACC_LD + SC_MAXK * TRW + 32 + TRW + (SC_MAXK * TRW + 3) / 4;
  static constexpr size_t lds_bytes = (size_t)NWB * WAVE_LDS_DW * 4;
  static constexpr int IMG = CIN * COUT;                               // dwords per offset
};

// W (K, CIN, COUT) -> register-fragment order:
//   img[k][g][q][t][n][c] = W[k][q*CQ+t][g*CG + c*16 + n]
template <int CIN, int COUT>
__global__ void k_pack_weights_wp(const float* __restrict__ W, int K, float* __restrict__ Wp) {
  using C = WpCfg<CIN, COUT>;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= K * CIN * COUT) return;
  int co = e % COUT;
  int ci = (e / COUT) % CIN;
  int k = e / (COUT * CIN);
  int g = co / C::CG, cl = co % C::CG;
  int c = cl / 16, n = cl % 16;
  int q = ci / C::CQ, t = ci % C::CQ;
  Wp[(size_t)k * C::IMG + ((((size_t)g * 4 + q) * C::CQ + t) * 16 + n) * C::NTG + c] = W[e];
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_sconv_wp(
    const float* __restrict__ in, const float* __restrict__ Wp, SconvEpilogue ep,
    const int* __restrict__ nbr, const int* __restrict__ tile_order, int N_out, int K,
    float* __restrict__ out) {
  if (ep.n_live) N_out = min(N_out, *ep.n_live);   // device-side row count (capacity launch)
  using C = WpCfg<CIN, COUT>;
  constexpr int TRW = C::TRW, ACC_LD = C::ACC_LD, CQ = C::CQ, NTG = C::NTG, NG = C::NG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  float* s_acc = smem + (size_t)wave * C::WAVE_LDS_DW;                 // TRW * ACC_LD
  int* s_pin = reinterpret_cast<int*>(s_acc + TRW * ACC_LD);           // SC_MAXK * TRW
  int* s_cnt = s_pin + SC_MAXK * TRW;                                  // 32
  int* s_rows = s_cnt + 32;                                            // TRW
  unsigned char* s_pslot = reinterpret_cast<unsigned char*>(s_rows + TRW);  // SC_MAXK * TRW
  const long long row0 = ((long long)blockIdx.x * C::NWB + wave) * TRW;
  if (row0 >= N_out) return;     // whole wave idle (no barriers anywhere below)

  // ---- my rows, their neighbour lists (registers), per-offset compaction (wave ballots)
  int my_row = -1;
  if (lane < TRW && row0 + lane < N_out) my_row = tile_order ? tile_order[row0 + lane] : (int)(row0 + lane);
  if (lane < TRW) s_rows[lane] = my_row;
  for (int i = lane; i < TRW * ACC_LD / 4; i += 64)
    reinterpret_cast<f32x4*>(s_acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned mask = 0;
#pragma unroll
  for (int k = 0; k < SC_MAXK; ++k) {
    int nb = (k < K && my_row >= 0) ? nbr[(long long)my_row * K + k] : -1;
    unsigned long long b = __ballot(nb >= 0);
    if (nb >= 0) {
      int pos = k * TRW + __popcll(b & ((1ull << lane) - 1ull));
      s_pin[pos] = nb;
      s_pslot[pos] = (unsigned char)lane;
    }
    int n = __popcll(b);
    if (lane == 0) s_cnt[k] = n;
    if (n) mask |= 1u << k;
  }
  mask = __builtin_amdgcn_readfirstlane(mask);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- register sets: W fragment of the current / next phase, input rows of the current / next chunk
  float Wc[CQ][NTG], Wn[CQ][NTG];
  float Ac[CQ], An[CQ];
  bool an_valid = false;

#define WP_LOAD_W(KK, GG)                                                                    \
  {                                                                                          \
    const float* wp_ = Wp + (size_t)(KK) * C::IMG + (((size_t)(GG) * 4 + q) * CQ * 16 + r) * NTG; \
    _Pragma("unroll") for (int t_ = 0; t_ < CQ; ++t_) {                                      \
      if constexpr (NTG == 4) {                                                              \
        f32x4 v_ = *reinterpret_cast<const f32x4*>(wp_ + t_ * 16 * NTG);                     \
        Wn[t_][0] = v_[0]; Wn[t_][1] = v_[1]; Wn[t_][2] = v_[2]; Wn[t_][3] = v_[3];          \
      } else if constexpr (NTG == 2) {                                                       \
        float2 v_ = *reinterpret_cast<const float2*>(wp_ + t_ * 16 * NTG);                   \
        Wn[t_][0] = v_.x; Wn[t_][1] = v_.y;                                                  \
      } else {                                                                               \
        Wn[t_][0] = wp_[t_ * 16 * NTG];                                                      \
      }                                                                                      \
    }                                                                                        \
  }
#define WP_GATHER(KK, CC, CNT)                                                               \
  {                                                                                          \
    int p_ = (CC) * 16 + r;                                                                  \
    int irow_ = (p_ < (CNT)) ? s_pin[(KK) * TRW + p_] : -1;                                  \
    const float* ap_ = in + (long long)(irow_ < 0 ? 0 : irow_) * CIN + q * CQ;               \
    if constexpr (CQ % 4 == 0) {                                                             \
      _Pragma("unroll") for (int i_ = 0; i_ < CQ / 4; ++i_) {                                \
        f32x4 v_ = reinterpret_cast<const f32x4*>(ap_)[i_];                                  \
        An[4 * i_ + 0] = v_[0]; An[4 * i_ + 1] = v_[1]; An[4 * i_ + 2] = v_[2]; An[4 * i_ + 3] = v_[3]; \
      }                                                                                      \
    } else {                                                                                 \
      _Pragma("unroll") for (int i_ = 0; i_ < CQ; ++i_) An[i_] = ap_[i_];                    \
    }                                                                                        \
    an_valid = irow_ >= 0;                                                                   \
  }

  unsigned rem = mask;
  int k = -1, g = 0, cnt = 0;
  if (rem) {
    k = __builtin_ctz(rem);
    rem &= rem - 1;
    cnt = s_cnt[k];
    WP_LOAD_W(k, 0);
    WP_GATHER(k, 0, cnt);
  }
  while (k >= 0) {
    // ---- phase (k, g): promote the prefetched W set, start fetching the next one
#pragma unroll
    for (int t = 0; t < CQ; ++t)
#pragma unroll
      for (int c = 0; c < NTG; ++c) Wc[t][c] = Wn[t][c];
    int k2 = k, g2 = g + 1, cnt2 = cnt;
    if (g2 == NG) {
      g2 = 0;
      if (rem) {
        k2 = __builtin_ctz(rem);
        rem &= rem - 1;
        cnt2 = s_cnt[k2];
      } else {
        k2 = -1;
      }
    }
    if (k2 >= 0) WP_LOAD_W(k2, g2);
    const int nc = (cnt + 15) >> 4;
    for (int c = 0; c < nc; ++c) {
#pragma unroll
      for (int i = 0; i < CQ; ++i) Ac[i] = an_valid ? An[i] : 0.f;
      if (c + 1 < nc) {
        WP_GATHER(k, c + 1, cnt);
      } else if (k2 >= 0) {
        WP_GATHER(k2, 0, cnt2);
      }
      // operands swapped: D[i = cout][j = pair]; lane (pair r, q) gets channels 16ct+4q..+3
      f32x4 acc[NTG];
#pragma unroll
      for (int ct = 0; ct < NTG; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < CQ; ++t)
#pragma unroll
        for (int ct = 0; ct < NTG; ++ct)
          acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wc[t][ct], Ac[t], acc[ct], 0, 0, 0);
      const int p = c * 16 + r;
      if (p < cnt) {
        float* dst = s_acc + (int)s_pslot[k * TRW + p] * ACC_LD + g * C::CG + 4 * q;
#pragma unroll
        for (int ct = 0; ct < NTG; ++ct) {
          f32x4 v = *reinterpret_cast<f32x4*>(dst + ct * 16);
          v += acc[ct];
          *reinterpret_cast<f32x4*>(dst + ct * 16) = v;
        }
      }
    }
    k = k2; g = g2; cnt = cnt2;
  }
#undef WP_LOAD_W
#undef WP_GATHER
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- epilogue: this wave's rows, coalesced float4 stores with the fused pointwise tail
  constexpr int C4 = COUT / 4;
  for (int i = lane; i < TRW * C4; i += 64) {
    int rr = i / C4, c4 = i - rr * C4;
    int orow = s_rows[rr];
    if (orow < 0) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(s_acc + rr * ACC_LD + 4 * c4);
    const int co = 4 * c4;
    if (ep.bias) v += *reinterpret_cast<const f32x4*>(ep.bias + co);
    if (ep.scale) v *= *reinterpret_cast<const f32x4*>(ep.scale + co);
    if (ep.shift) v += *reinterpret_cast<const f32x4*>(ep.shift + co);
    if (ep.relu) {
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + (long long)orow * COUT + 4 * c4) = v;
  }
}

// ==================================================================== register-tile kernel
// Default sparse-conv kernel.  One wave (= one 64-thread block, so the hardware dispatcher
// load-balances the very uneven tiles) owns two 16-row output tiles whose accumulators live in
// registers for the whole kernel: no LDS accumulation, no barriers, no atomics.  For every
// kernel offset present in either tile the wave streams the W[k] fragment from L2 into
// registers (prefetched one offset ahead), gathers the 16 neighbour rows of a tile (prefetched
// one tile ahead; each load instruction reads one full 64-B segment per row) and issues
// CQ x NTG v_mfma_f32_16x16x4_f32.  Output channels wider than a register set are processed in
// column groups.  Summation order per output element is fixed: bitwise reproducible.
template <int CIN, int COUT>
struct RtCfg {
  static constexpr int CQ = CIN / 4;
  static constexpr int CG0 = 4096 / CIN;                              // cols per W register set
  static constexpr int CG = COUT < CG0 ? COUT : (CG0 < 16 ? 16 : CG0);
  static constexpr int NG = COUT / CG;                                 // column groups
  static constexpr int NTG = CG / 16;                                  // 16-col tiles per group
  static constexpr int IMG = CIN * COUT;                               // dwords per offset
  static constexpr int ROWS = 32;                                      // rows per wave (2 tiles)
};

// channel handled by lane quad q at k-step t: for CIN >= 16 step t = 4i+j reads channel
// 16i + 4q + j (so load i of the gather covers a contiguous 64-B segment of the row).
template <int CIN>
__host__ __device__ constexpr int rt_channel(int q, int t) {
  return CIN >= 16 ? 16 * (t / 4) + 4 * q + (t % 4) : q * (CIN / 4) + t;
}

// W (K, CIN, COUT) -> img[k][g][q][t][n][c] = W[k][rt_channel(q,t)][g*CG + c*16 + n]
template <int CIN, int COUT>
__global__ void k_pack_weights_rt(const float* __restrict__ W, int K, float* __restrict__ Wp) {
  using C = RtCfg<CIN, COUT>;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= K * CIN * COUT) return;
  int c = e % C::NTG;
  int n = (e / C::NTG) % 16;
  int t = (e / (C::NTG * 16)) % C::CQ;
  int q = (e / (C::NTG * 16 * C::CQ)) % 4;
  int g = (e / (C::NTG * 16 * C::CQ * 4)) % C::NG;
  int k = e / (C::NTG * 16 * C::CQ * 4 * C::NG);
  int ci = rt_channel<CIN>(q, t);
  int co = g * C::CG + c * 16 + n;
  Wp[e] = W[((size_t)k * CIN + ci) * COUT + co];
}

template <int CIN, int COUT>
__global__ __launch_bounds__(64, 2) void k_sconv_rt(
    const float* __restrict__ in, const float* __restrict__ Wp, SconvEpilogue ep,
    const int* __restrict__ nbr, const int* __restrict__ tile_order, int N_out, int K,
    float* __restrict__ out) {
  if (ep.n_live) N_out = min(N_out, *ep.n_live);   // device-side row count (capacity launch)
  using C = RtCfg<CIN, COUT>;
  constexpr int CQ = C::CQ, NTG = C::NTG, NG = C::NG;
  __shared__ int s_nbr[2 * 16 * (SC_MAXK + 1)];
  const int lane = threadIdx.x;
  const int r = lane & 15, q = lane >> 4;
  const long long row0 = (long long)blockIdx.x * C::ROWS;

  // ---- neighbour table of my 32 rows -> LDS; per-tile offset masks
  int my_row = -1;      // lanes 0..31: output row of slot `lane`
  if (lane < 32 && row0 + lane < N_out) my_row = tile_order ? tile_order[row0 + lane] : (int)(row0 + lane);
  unsigned m0 = 0, m1 = 0;
  for (int e = lane; e < 32 * K; e += 64) {
    int slot = e / K, kk = e - slot * K;
    int orow = __shfl(my_row, slot, 64);
    int v = orow >= 0 ? nbr[(long long)orow * K + kk] : -1;
    s_nbr[slot * (SC_MAXK + 1) + kk] = v;
    if (v >= 0) { if (slot < 16) m0 |= 1u << kk; else m1 |= 1u << kk; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m0 |= __shfl_xor(m0, o, 64);
    m1 |= __shfl_xor(m1, o, 64);
  }
  m0 = __builtin_amdgcn_readfirstlane(m0);
  m1 = __builtin_amdgcn_readfirstlane(m1);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int orow0 = __shfl(my_row, r, 64), orow1 = __shfl(my_row, 16 + r, 64);

  float Wc[CQ][NTG], Wn[CQ][NTG];
  float Ac[CQ], An[CQ];
  bool an_valid = false;

#define RT_LOAD_W(KK, GG)                                                                    \
  {                                                                                          \
    const float* wp_ = Wp + (size_t)(KK) * C::IMG + (((size_t)(GG) * 4 + q) * CQ * 16 + r) * NTG; \
    _Pragma("unroll") for (int t_ = 0; t_ < CQ; ++t_) {                                      \
      if constexpr (NTG == 4) {                                                              \
        f32x4 v_ = *reinterpret_cast<const f32x4*>(wp_ + t_ * 16 * NTG);                     \
        Wn[t_][0] = v_[0]; Wn[t_][1] = v_[1]; Wn[t_][2] = v_[2]; Wn[t_][3] = v_[3];          \
      } else if constexpr (NTG == 2) {                                                       \
        float2 v_ = *reinterpret_cast<const float2*>(wp_ + t_ * 16 * NTG);                   \
        Wn[t_][0] = v_.x; Wn[t_][1] = v_.y;                                                  \
      } else if constexpr (NTG == 8) {                                                       \
        f32x4 v_ = *reinterpret_cast<const f32x4*>(wp_ + t_ * 16 * NTG);                     \
        f32x4 u_ = *reinterpret_cast<const f32x4*>(wp_ + t_ * 16 * NTG + 4);                 \
        Wn[t_][0] = v_[0]; Wn[t_][1] = v_[1]; Wn[t_][2] = v_[2]; Wn[t_][3] = v_[3];          \
        Wn[t_][4] = u_[0]; Wn[t_][5] = u_[1]; Wn[t_][6] = u_[2]; Wn[t_][7] = u_[3];          \
      } else {                                                                               \
        Wn[t_][0] = wp_[t_ * 16 * NTG];                                                      \
      }                                                                                      \
    }                                                                                        \
  }
  // gather tile S (0/1) at offset KK: lane (r,q) reads 16-B pieces q of the row's 64-B segments
#define RT_GATHER(KK, S)                                                                     \
  {                                                                                          \
    int irow_ = s_nbr[((S) * 16 + r) * (SC_MAXK + 1) + (KK)];                                \
    const float* ap_ = in + (long long)(irow_ < 0 ? 0 : irow_) * CIN;                        \
    if constexpr (CIN >= 16) {                                                               \
      _Pragma("unroll") for (int i_ = 0; i_ < CQ / 4; ++i_) {                                \
        f32x4 v_ = *reinterpret_cast<const f32x4*>(ap_ + 16 * i_ + 4 * q);                   \
        An[4 * i_ + 0] = v_[0]; An[4 * i_ + 1] = v_[1]; An[4 * i_ + 2] = v_[2]; An[4 * i_ + 3] = v_[3]; \
      }                                                                                      \
    } else {                                                                                 \
      _Pragma("unroll") for (int i_ = 0; i_ < CQ; ++i_) An[i_] = ap_[q * CQ + i_];           \
    }                                                                                        \
    an_valid = irow_ >= 0;   /* zeroing happens at promotion time: a write to An here would  \
                                force a vmcnt(0) wait right behind the prefetch loads */     \
  }
  // next (offset, tile) item after (KK, S) in the order k-major, tile-minor
#define RT_NEXT(KK, S, REM, NK, NS)                                                          \
  {                                                                                          \
    NK = -1; NS = 0;                                                                         \
    if ((S) == 0 && ((m1 >> (KK)) & 1u)) { NK = (KK); NS = 1; }                              \
    else if (REM) { NK = __builtin_ctz(REM); NS = ((m0 >> NK) & 1u) ? 0 : 1; }              \
  }

  const unsigned mu = m0 | m1;
  for (int g = 0; g < NG; ++g) {
    f32x4 acc0[NTG], acc1[NTG];
#pragma unroll
    for (int ct = 0; ct < NTG; ++ct) { acc0[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[ct] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    unsigned rem = mu;
    int k = -1, sidx = 0;
    if (rem) {
      k = __builtin_ctz(rem);
      rem &= rem - 1;
      sidx = ((m0 >> k) & 1u) ? 0 : 1;
      RT_LOAD_W(k, g);
      if (sidx == 0) { RT_GATHER(k, 0); } else { RT_GATHER(k, 1); }
    }
    bool new_k = true;
    while (k >= 0) {
      if (new_k) {   // promote the prefetched W set, start fetching the next offset's
#pragma unroll
        for (int t = 0; t < CQ; ++t)
#pragma unroll
          for (int c = 0; c < NTG; ++c) Wc[t][c] = Wn[t][c];
        if (rem) RT_LOAD_W(__builtin_ctz(rem), g);
      }
#pragma unroll
      for (int i = 0; i < CQ; ++i) Ac[i] = an_valid ? An[i] : 0.f;
      int nk, ns;
      RT_NEXT(k, sidx, rem, nk, ns);
      if (nk >= 0) {
        if (ns == 0) { RT_GATHER(nk, 0); } else { RT_GATHER(nk, 1); }
      }
      // operands swapped: D[i = cout][j = row]; lane (row r, q) gets channels 16ct+4q..+3
      if (sidx == 0) {
#pragma unroll
        for (int t = 0; t < CQ; ++t)
#pragma unroll
          for (int ct = 0; ct < NTG; ++ct)
            acc0[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wc[t][ct], Ac[t], acc0[ct], 0, 0, 0);
      } else {
#pragma unroll
        for (int t = 0; t < CQ; ++t)
#pragma unroll
          for (int ct = 0; ct < NTG; ++ct)
            acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wc[t][ct], Ac[t], acc1[ct], 0, 0, 0);
      }
      new_k = nk != k;
      if (new_k && nk >= 0) rem &= rem - 1;
      k = nk; sidx = ns;
    }
    // ---- epilogue of this column group straight from registers
#pragma unroll
    for (int ct = 0; ct < NTG; ++ct) {
      const int co = g * C::CG + ct * 16 + 4 * q;
      f32x4 v0 = acc0[ct], v1 = acc1[ct];
      if (ep.bias) { f32x4 b = *reinterpret_cast<const f32x4*>(ep.bias + co); v0 += b; v1 += b; }
      if (ep.scale) { f32x4 b = *reinterpret_cast<const f32x4*>(ep.scale + co); v0 *= b; v1 *= b; }
      if (ep.shift) { f32x4 b = *reinterpret_cast<const f32x4*>(ep.shift + co); v0 += b; v1 += b; }
      if (ep.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] = fmaxf(v0[j], 0.f); v1[j] = fmaxf(v1[j], 0.f); }
      }
      if (orow0 >= 0) *reinterpret_cast<f32x4*>(out + (long long)orow0 * COUT + co) = v0;
      if (orow1 >= 0) *reinterpret_cast<f32x4*>(out + (long long)orow1 * COUT + co) = v1;
    }
  }
#undef RT_LOAD_W
#undef RT_GATHER
#undef RT_NEXT
}

// ==================================================================== column-owner kernel
// A block owns TR = 128 output rows and compacts them per kernel offset into pair lists (as
// the block kernel does), but the work is split by OUTPUT COLUMNS: wave (ct, slab) computes the
// 16-channel tile ct for every pair of its row slab.  Every wave therefore does the same amount
// of work, only ever touches its own columns of the LDS accumulator tile (no conflicts, no
// barriers in the main loop, fixed summation order), and needs just a (Cin x 16) slice of W[k],
// which it streams from L2 straight into registers one offset ahead.  The price is that the
// waves of a block gather the same input rows (served by L1/L2).
template <int CIN, int COUT>
struct CoCfg {
  static constexpr int CQ = CIN / 4;
  static constexpr int NT = COUT / 16;                 // column tiles = column owners
  static constexpr int NW = NT < 4 ? 4 : NT;           // waves per block
  static constexpr int RS = NW / NT;                   // row slabs
  static constexpr int TR = 128;
  static constexpr int TRS = TR / RS;                  // rows per slab (32, 64 or 128)
  static constexpr int ACC_LD = COUT + 4;
  static constexpr int IMG = CIN * COUT;
  static constexpr size_t lds_bytes = (size_t)TR * ACC_LD * 4 + (size_t)SC_MAXK * TR * 5 +
                                      (size_t)(TR + RS * 32 + 2 * 32) * 4 + 64;
};

// W (K, CIN, COUT) -> img[k][ct][q][n][t] = W[k][rt_channel(q,t)][ct*16 + n]
template <int CIN, int COUT>
__global__ void k_pack_weights_co(const float* __restrict__ W, int K, float* __restrict__ Wp) {
  using C = CoCfg<CIN, COUT>;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= K * CIN * COUT) return;
  int t = e % C::CQ;
  int n = (e / C::CQ) % 16;
  int q = (e / (C::CQ * 16)) % 4;
  int ct = (e / (C::CQ * 64)) % C::NT;
  int k = e / (C::CQ * 64 * C::NT);
  Wp[e] = W[((size_t)k * CIN + rt_channel<CIN>(q, t)) * COUT + ct * 16 + n];
}

template <int CIN, int COUT>
__global__ __launch_bounds__(512) void k_sconv_co(const float* __restrict__ in, const float* __restrict__ Wp,
                           SconvEpilogue ep, const int* __restrict__ nbr,
                           const int* __restrict__ tile_order, int N_out, int K,
                           float* __restrict__ out) {
  if (ep.n_live) N_out = min(N_out, *ep.n_live);   // device-side row count (capacity launch)
  using C = CoCfg<CIN, COUT>;
  constexpr int CQ = C::CQ, TR = C::TR, TRS = C::TRS, RS = C::RS, ACC_LD = C::ACC_LD;
  constexpr int THREADS = C::NW * 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_acc = smem;                                          // TR * ACC_LD
  int* s_pin = reinterpret_cast<int*>(s_acc + TR * ACC_LD);     // SC_MAXK * TR  ([k][slot-in-list])
  int* s_rows = s_pin + SC_MAXK * TR;                           // TR
  int* s_cnt = s_rows + TR;                                     // RS * 32
  int* s_wcnt = s_cnt + RS * 32;                                // 2 * 32 (setup only)
  unsigned char* s_pslot = reinterpret_cast<unsigned char*>(s_wcnt + 64);   // SC_MAXK * TR
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * TR;

  // ---- setup (threads < TR own one row slot each): rows, neighbour lists, per-slab compaction.
  // list of (slab, k) lives at s_pin[k * TR + slab * TRS ...]
  int my_row = -1;
  if (tid < TR) {
    int p = row0 + tid;
    my_row = (p < N_out) ? (tile_order ? tile_order[p] : p) : -1;
    s_rows[tid] = my_row;
  }
  for (int i = tid; i < TR * ACC_LD / 4; i += THREADS)
    reinterpret_cast<f32x4*>(s_acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int nb[SC_MAXK];
  const int slab = tid / TRS;                       // my slab (setup threads)
  // lanes of my wave that belong to my slab
  const unsigned long long slab_lanes =
      TRS >= 64 ? ~0ull : (((1ull << TRS) - 1ull) << ((lane / TRS) * TRS));
  if (tid < TR) {
#pragma unroll
    for (int k = 0; k < SC_MAXK; ++k)
      nb[k] = (k < K && my_row >= 0) ? nbr[(long long)my_row * K + k] : -1;
    if (TRS > 64) {
#pragma unroll
      for (int k = 0; k < SC_MAXK; ++k) {
        unsigned long long b = __ballot(nb[k] >= 0);
        if (lane == 0) s_wcnt[wave * 32 + k] = __popcll(b);
      }
    }
  }
  __syncthreads();
  if (tid < TR) {
#pragma unroll
    for (int k = 0; k < SC_MAXK; ++k) {
      const bool v = nb[k] >= 0;
      unsigned long long b = __ballot(v) & slab_lanes;
      int base = (TRS > 64 && (wave & 1)) ? s_wcnt[(wave - 1) * 32 + k] : 0;
      if (v) {
        int pos = k * TR + slab * TRS + base + __popcll(b & ((1ull << lane) - 1ull));
        s_pin[pos] = nb[k];
        s_pslot[pos] = (unsigned char)tid;
      }
      // the last lane group of a slab publishes the slab's count
      const bool closer = TRS > 64 ? ((wave & 1) && lane == 0) : ((lane % (TRS >= 64 ? 64 : TRS)) == 0);
      if (closer) s_cnt[slab * 32 + k] = base + __popcll(b);
    }
  }
  __syncthreads();

  // ---- main loop: wave (ct, wslab) walks every offset of its slab, no barriers
  const int ct = wave % C::NT, wslab = wave / C::NT;
  unsigned mask = 0;
#pragma unroll
  for (int k = 0; k < SC_MAXK; ++k)
    if (k < K && s_cnt[wslab * 32 + k] > 0) mask |= 1u << k;
  mask = __builtin_amdgcn_readfirstlane(mask);
  const int* pin = s_pin + wslab * TRS;
  const unsigned char* pslot = s_pslot + wslab * TRS;

  // Software pipeline: a wave has only ~16 MFMAs (512 cycles) of work per chunk, far less than
  // one L2 round trip, so 4 chunk gathers (ring A0..A3) and 2 weight slices (W1, W2) are kept in
  // flight ahead of the chunk being multiplied.
  float Wc[CQ], W1[CQ], W2[CQ], Ac[CQ];
  float A0[CQ], A1[CQ], A2[CQ], A3[CQ];
  bool v0 = false, v1 = false, v2 = false, v3 = false;
#define CO_LOAD_W(KK, WDST)                                                                  \
  {                                                                                          \
    const float* wp_ = Wp + (size_t)(KK) * C::IMG + (((size_t)ct * 4 + q) * 16 + r) * CQ;    \
    if constexpr (CQ % 4 == 0) {                                                             \
      _Pragma("unroll") for (int i_ = 0; i_ < CQ / 4; ++i_) {                                \
        f32x4 v_ = reinterpret_cast<const f32x4*>(wp_)[i_];                                  \
        WDST[4 * i_ + 0] = v_[0]; WDST[4 * i_ + 1] = v_[1]; WDST[4 * i_ + 2] = v_[2]; WDST[4 * i_ + 3] = v_[3]; \
      }                                                                                      \
    } else {                                                                                 \
      _Pragma("unroll") for (int i_ = 0; i_ < CQ; ++i_) WDST[i_] = wp_[i_];                  \
    }                                                                                        \
  }
#define CO_GATHER(KK, CC, CNT, ADST, VDST)                                                   \
  {                                                                                          \
    int p_ = (CC) * 16 + r;                                                                  \
    int irow_ = (p_ < (CNT)) ? pin[(KK) * TR + p_] : -1;                                     \
    const float* ap_ = in + (long long)(irow_ < 0 ? 0 : irow_) * CIN;                        \
    if constexpr (CIN >= 16) {                                                               \
      _Pragma("unroll") for (int i_ = 0; i_ < CQ / 4; ++i_) {                                \
        f32x4 v_ = *reinterpret_cast<const f32x4*>(ap_ + 16 * i_ + 4 * q);                   \
        ADST[4 * i_ + 0] = v_[0]; ADST[4 * i_ + 1] = v_[1]; ADST[4 * i_ + 2] = v_[2]; ADST[4 * i_ + 3] = v_[3]; \
      }                                                                                      \
    } else {                                                                                 \
      _Pragma("unroll") for (int i_ = 0; i_ < CQ; ++i_) ADST[i_] = ap_[q * CQ + i_];         \
    }                                                                                        \
    VDST = irow_ >= 0;                                                                       \
  }
  // cursors over the (offset, chunk) items of this slab: P = prefetch, X = compute
  struct Cur { unsigned rem; int k, c, cnt; };
  auto cur_init = [&](Cur& u) {
    u.rem = mask; u.k = -1; u.c = 0; u.cnt = 0;
    if (u.rem) { u.k = __builtin_ctz(u.rem); u.rem &= u.rem - 1; u.cnt = s_cnt[wslab * 32 + u.k]; }
  };
  auto cur_next = [&](Cur& u) {
    if (++u.c * 16 >= u.cnt) {
      u.c = 0;
      if (u.rem) { u.k = __builtin_ctz(u.rem); u.rem &= u.rem - 1; u.cnt = s_cnt[wslab * 32 + u.k]; }
      else u.k = -1;
    }
  };
  Cur P, X;
  cur_init(P);
  cur_init(X);
  if (P.k >= 0) { CO_GATHER(P.k, P.c, P.cnt, A0, v0); cur_next(P); }
  if (P.k >= 0) { CO_GATHER(P.k, P.c, P.cnt, A1, v1); cur_next(P); }
  if (P.k >= 0) { CO_GATHER(P.k, P.c, P.cnt, A2, v2); cur_next(P); }
  if (P.k >= 0) { CO_GATHER(P.k, P.c, P.cnt, A3, v3); cur_next(P); }
  // weight slices of the first two offsets
  {
    unsigned m2 = mask;
    if (m2) { CO_LOAD_W(__builtin_ctz(m2), W1); m2 &= m2 - 1; }
    if (m2) { CO_LOAD_W(__builtin_ctz(m2), W2); }
  }
  int curk = -1;
#define CO_STEP(ABUF, VBUF)                                                                  \
  if (X.k >= 0) {                                                                            \
    if (X.k != curk) {   /* new offset: rotate the weight slices, fetch the one 2 offsets ahead */ \
      curk = X.k;                                                                            \
      _Pragma("unroll") for (int t_ = 0; t_ < CQ; ++t_) { Wc[t_] = W1[t_]; W1[t_] = W2[t_]; } \
      unsigned m2_ = X.rem;                                                                  \
      if (m2_) { m2_ &= m2_ - 1; if (m2_) CO_LOAD_W(__builtin_ctz(m2_), W2); }               \
    }                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < CQ; ++i_) Ac[i_] = VBUF ? ABUF[i_] : 0.f;        \
    const int xk_ = X.k, xc_ = X.c, xcnt_ = X.cnt;                                           \
    cur_next(X);                                                                             \
    if (P.k >= 0) { CO_GATHER(P.k, P.c, P.cnt, ABUF, VBUF); cur_next(P); }                   \
    f32x4 acc0{0.f, 0.f, 0.f, 0.f}, acc1{0.f, 0.f, 0.f, 0.f};                                \
    _Pragma("unroll") for (int t_ = 0; t_ < CQ; t_ += 2) {                                   \
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Wc[t_], Ac[t_], acc0, 0, 0, 0);            \
      if (t_ + 1 < CQ) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Wc[t_ + 1], Ac[t_ + 1], acc1, 0, 0, 0); \
    }                                                                                        \
    acc0 += acc1;                                                                            \
    const int p_out_ = xc_ * 16 + r;                                                         \
    if (p_out_ < xcnt_) {                                                                    \
      float* dst_ = s_acc + (int)pslot[xk_ * TR + p_out_] * ACC_LD + ct * 16 + 4 * q;        \
      f32x4 o_ = *reinterpret_cast<f32x4*>(dst_);                                            \
      o_ += acc0;                                                                            \
      *reinterpret_cast<f32x4*>(dst_) = o_;                                                  \
    }                                                                                        \
  }
  while (X.k >= 0) {
    CO_STEP(A0, v0)
    CO_STEP(A1, v1)
    CO_STEP(A2, v2)
    CO_STEP(A3, v3)
  }
#undef CO_STEP
#undef CO_LOAD_W
#undef CO_GATHER
  __syncthreads();

  // ---- epilogue: coalesced row stores with the fused pointwise tail
  constexpr int C4 = COUT / 4;
  for (int i = tid; i < TR * C4; i += THREADS) {
    int rr = i / C4, c4 = i - rr * C4;
    int orow = s_rows[rr];
    if (orow < 0) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(s_acc + rr * ACC_LD + 4 * c4);
    const int co = 4 * c4;
    if (ep.bias) v += *reinterpret_cast<const f32x4*>(ep.bias + co);
    if (ep.scale) v *= *reinterpret_cast<const f32x4*>(ep.scale + co);
    if (ep.shift) v += *reinterpret_cast<const f32x4*>(ep.shift + co);
    if (ep.relu) {
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + (long long)orow * COUT + 4 * c4) = v;
  }
}

// ==================================================================== wave-private kernel, deep pipeline
// k_sconv_wq: every wave owns TRW = 48 consecutive output rows (compaction lists, fp32
// accumulator tile in LDS, epilogue) and never synchronises with another wave.  Per (offset,
// 16-pair chunk) item it issues CQ x NTG MFMAs against the full W[k] fragment held in registers.
// Latency hiding is explicit: the gathers of the next 4 items are always in flight (register
// ring A0..A3, filled straight by the loads -- absent pairs read a zero row, so there is no
// select or copy between load and MFMA) and W of the next offset streams into the second
// register set while the current one multiplies (the two sets swap roles, no copy).
template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void k_sconv_wq(
    const float* __restrict__ in, const float* __restrict__ Wp, SconvEpilogue ep,
    const int* __restrict__ nbr, const int* __restrict__ tile_order,
    const float* __restrict__ zero_row, int N_out, int K, float* __restrict__ out) {
  if (ep.n_live) N_out = min(N_out, *ep.n_live);   // device-side row count (capacity launch)
  using C = RtCfg<CIN, COUT>;
  using L = WpCfg<CIN, COUT>;   // LDS layout of the wave-private tile
  constexpr int TRW = L::TRW, ACC_LD = L::ACC_LD, CQ = C::CQ, NTG = C::NTG, NG = C::NG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  float* s_acc = smem + (size_t)wave * L::WAVE_LDS_DW;                 // TRW * ACC_LD
  int* s_pin = reinterpret_cast<int*>(s_acc + TRW * ACC_LD);           // SC_MAXK * TRW
  int* s_cnt = s_pin + SC_MAXK * TRW;                                  // 32
  int* s_rows = s_cnt + 32;                                            // TRW
  unsigned char* s_pslot = reinterpret_cast<unsigned char*>(s_rows + TRW);
  const long long row0 = ((long long)blockIdx.x * L::NWB + wave) * TRW;
  if (row0 >= N_out) return;

  int my_row = -1;
  if (lane < TRW && row0 + lane < N_out) my_row = tile_order ? tile_order[row0 + lane] : (int)(row0 + lane);
  if (lane < TRW) s_rows[lane] = my_row;
  for (int i = lane; i < TRW * ACC_LD / 4; i += 64)
    reinterpret_cast<f32x4*>(s_acc)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned mask = 0;
  int nbv[SC_MAXK];     // all 27 loads in flight at once
#pragma unroll
  for (int k = 0; k < SC_MAXK; ++k)
    nbv[k] = (k < K && my_row >= 0) ? nbr[(long long)my_row * K + k] : -1;
#pragma unroll
  for (int k = 0; k < SC_MAXK; ++k) {
    const int nb = nbv[k];
    unsigned long long b = __ballot(nb >= 0);
    if (nb >= 0) {
      int pos = k * TRW + __popcll(b & ((1ull << lane) - 1ull));
      s_pin[pos] = nb;
      s_pslot[pos] = (unsigned char)lane;
    }
    int n = __popcll(b);
    if (lane == 0) s_cnt[k] = n;
    if (n) mask |= 1u << k;
  }
  mask = __builtin_amdgcn_readfirstlane(mask);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  constexpr bool DEEP = CQ * NTG <= 32;   // ring of 4 when registers allow, else 2
  float Wa[CQ][NTG], Wb[CQ][NTG];
  float A0[CQ], A1[CQ], A2[DEEP ? CQ : 1], A3[DEEP ? CQ : 1];

#define WQ_LOAD_W(KK, GG, WDST)                                                              \
  {                                                                                          \
    const float* wp_ = Wp + (size_t)(KK) * C::IMG + (((size_t)(GG) * 4 + q) * CQ * 16 + r) * NTG; \
    _Pragma("unroll") for (int t_ = 0; t_ < CQ; ++t_) {                                      \
      if constexpr (NTG == 4) {                                                              \
        f32x4 v_ = *reinterpret_cast<const f32x4*>(wp_ + t_ * 16 * NTG);                     \
        WDST[t_][0] = v_[0]; WDST[t_][1] = v_[1]; WDST[t_][2] = v_[2]; WDST[t_][3] = v_[3];  \
      } else if constexpr (NTG == 2) {                                                       \
        float2 v_ = *reinterpret_cast<const float2*>(wp_ + t_ * 16 * NTG);                   \
        WDST[t_][0] = v_.x; WDST[t_][1] = v_.y;                                              \
      } else if constexpr (NTG == 8) {                                                       \
        f32x4 v_ = *reinterpret_cast<const f32x4*>(wp_ + t_ * 16 * NTG);                     \
        f32x4 u_ = *reinterpret_cast<const f32x4*>(wp_ + t_ * 16 * NTG + 4);                 \
        WDST[t_][0] = v_[0]; WDST[t_][1] = v_[1]; WDST[t_][2] = v_[2]; WDST[t_][3] = v_[3];  \
        WDST[t_][4] = u_[0]; WDST[t_][5] = u_[1]; WDST[t_][6] = u_[2]; WDST[t_][7] = u_[3];  \
      } else {                                                                               \
        WDST[t_][0] = wp_[t_ * 16 * NTG];                                                    \
      }                                                                                      \
    }                                                                                        \
  }
#define WQ_GATHER(KK, CC, CNT, ADST)                                                         \
  {                                                                                          \
    int p_ = (CC) * 16 + r;                                                                  \
    const float* ap_ = zero_row;                                                             \
    if (p_ < (CNT)) ap_ = in + (long long)s_pin[(KK) * TRW + p_] * CIN;                      \
    if constexpr (CIN >= 16) {                                                               \
      _Pragma("unroll") for (int i_ = 0; i_ < CQ / 4; ++i_) {                                \
        f32x4 v_ = *reinterpret_cast<const f32x4*>(ap_ + 16 * i_ + 4 * q);                   \
        ADST[4 * i_ + 0] = v_[0]; ADST[4 * i_ + 1] = v_[1]; ADST[4 * i_ + 2] = v_[2]; ADST[4 * i_ + 3] = v_[3]; \
      }                                                                                      \
    } else {                                                                                 \
      _Pragma("unroll") for (int i_ = 0; i_ < CQ; ++i_) ADST[i_] = ap_[q * CQ + i_];         \
    }                                                                                        \
  }
  struct Cur { unsigned rem; int k, c, cnt; };
#define WQ_CUR_INIT(U)                                                                       \
  { U.rem = mask; U.k = -1; U.c = 0; U.cnt = 0;                                              \
    if (U.rem) { U.k = __builtin_ctz(U.rem); U.rem &= U.rem - 1; U.cnt = s_cnt[U.k]; } }
#define WQ_CUR_NEXT(U)                                                                       \
  { if (++U.c * 16 >= U.cnt) { U.c = 0;                                                      \
      if (U.rem) { U.k = __builtin_ctz(U.rem); U.rem &= U.rem - 1; U.cnt = s_cnt[U.k]; }     \
      else U.k = -1; } }
  // one item: (new offset? swap W sets and prefetch the following offset's W), multiply,
  // refill this ring slot with the item 4 ahead, add into the LDS tile
#define WQ_MULT(ABUF, WSET)                                                                  \
  _Pragma("unroll") for (int t_ = 0; t_ < CQ; ++t_)                                          \
    _Pragma("unroll") for (int c_ = 0; c_ < NTG; ++c_)                                       \
      acc[c_] = __builtin_amdgcn_mfma_f32_16x16x4f32(WSET[t_][c_], ABUF[t_], acc[c_], 0, 0, 0);
#define WQ_STEP(ABUF)                                                                        \
  if (X.k >= 0) {                                                                            \
    if (X.k != curk) {                                                                       \
      curk = X.k;                                                                            \
      wpar ^= 1;                                                                             \
      if (X.rem) {                                                                           \
        if (wpar) { WQ_LOAD_W(__builtin_ctz(X.rem), g, Wa); } else { WQ_LOAD_W(__builtin_ctz(X.rem), g, Wb); } \
      }                                                                                      \
    }                                                                                        \
    f32x4 acc[NTG];                                                                          \
    _Pragma("unroll") for (int c_ = 0; c_ < NTG; ++c_) acc[c_] = f32x4{0.f, 0.f, 0.f, 0.f};  \
    if (wpar) { WQ_MULT(ABUF, Wb) } else { WQ_MULT(ABUF, Wa) }                               \
    const int xk_ = X.k, xc_ = X.c, xcnt_ = X.cnt;                                           \
    WQ_CUR_NEXT(X);                                                                          \
    if (P.k >= 0) { WQ_GATHER(P.k, P.c, P.cnt, ABUF); WQ_CUR_NEXT(P); }                      \
    const int p_out_ = xc_ * 16 + r;                                                         \
    if (p_out_ < xcnt_) {                                                                    \
      float* dst_ = s_acc + (int)s_pslot[xk_ * TRW + p_out_] * ACC_LD + g * C::CG + 4 * q;   \
      _Pragma("unroll") for (int c_ = 0; c_ < NTG; ++c_) {                                   \
        f32x4 o_ = *reinterpret_cast<f32x4*>(dst_ + c_ * 16);                                \
        o_ += acc[c_];                                                                       \
        *reinterpret_cast<f32x4*>(dst_ + c_ * 16) = o_;                                      \
      }                                                                                      \
    }                                                                                        \
  }

  for (int g = 0; g < NG; ++g) {
    Cur P, X;
    WQ_CUR_INIT(P);
    WQ_CUR_INIT(X);
    if (P.k >= 0) { WQ_GATHER(P.k, P.c, P.cnt, A0); WQ_CUR_NEXT(P); }
    if (P.k >= 0) { WQ_GATHER(P.k, P.c, P.cnt, A1); WQ_CUR_NEXT(P); }
    if constexpr (DEEP) {
      if (P.k >= 0) { WQ_GATHER(P.k, P.c, P.cnt, A2); WQ_CUR_NEXT(P); }
      if (P.k >= 0) { WQ_GATHER(P.k, P.c, P.cnt, A3); WQ_CUR_NEXT(P); }
    }
    // wpar == 1 means "current W is in Wb"; the first offset goes to Wb (wpar flips 0 -> 1)
    int wpar = 0, curk = -1;
    if (X.k >= 0) { WQ_LOAD_W(X.k, g, Wb); }
    while (X.k >= 0) {
      WQ_STEP(A0)
      WQ_STEP(A1)
      if constexpr (DEEP) {
        WQ_STEP(A2)
        WQ_STEP(A3)
      }
    }
  }
#undef WQ_STEP
#undef WQ_MULT
#undef WQ_CUR_NEXT
#undef WQ_CUR_INIT
#undef WQ_GATHER
#undef WQ_LOAD_W
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  constexpr int C4 = COUT / 4;
  for (int i = lane; i < TRW * C4; i += 64) {
    int rr = i / C4, c4 = i - rr * C4;
    int orow = s_rows[rr];
    if (orow < 0) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(s_acc + rr * ACC_LD + 4 * c4);
    const int co = 4 * c4;
    if (ep.bias) v += *reinterpret_cast<const f32x4*>(ep.bias + co);
    if (ep.scale) v *= *reinterpret_cast<const f32x4*>(ep.scale + co);
    if (ep.shift) v += *reinterpret_cast<const f32x4*>(ep.shift + co);
    if (ep.relu) {
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + (long long)orow * COUT + 4 * c4) = v;
  }
}

// ------------------------------------------------------------------ generic scalar kernel
__global__ void k_sconv_generic(const float* __restrict__ in, const float* __restrict__ W,
                                SconvEpilogue ep, const int* __restrict__ nbr, int N_out, int K,
                                int Cin, int Cout, float* __restrict__ out) {
  if (ep.n_live) N_out = min(N_out, *ep.n_live);   // device-side row count (capacity launch)
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N_out * Cout) return;
  int j = (int)(t / Cout);
  int co = (int)(t - (long long)j * Cout);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    int i = nbr[(long long)j * K + k];
    if (i < 0) continue;
    const float* ip = in + (long long)i * Cin;
    const float* wp = W + ((long long)k * Cin) * Cout + co;
    for (int ci = 0; ci < Cin; ++ci) acc = fmaf(ip[ci], wp[(long long)ci * Cout], acc);
  }
  if (ep.bias) acc += ep.bias[co];
  if (ep.scale) acc = acc * ep.scale[co] + (ep.shift ? ep.shift[co] : 0.f);
  else if (ep.shift) acc += ep.shift[co];
  if (ep.relu) acc = fmaxf(acc, 0.f);
  out[t] = acc;
}

extern "C" int glx_sconv_forward_generic(const float* in, int N_in, const float* W,
                                         const float* bias, const int32_t* nbr, int N_out, int K,
                                         int Cin, int Cout, float* out, void* stream) {
  (void)N_in;
  if (N_out == 0) return GLX_OK;
  GLX_REQUIRE(in && W && nbr && out && K > 0 && Cin > 0 && Cout > 0,
              "glx_sconv_forward_generic: bad arguments");
  long long total = (long long)N_out * Cout;
  SconvEpilogue ep{bias, nullptr, nullptr, 0, nullptr};
  hipLaunchKernelGGL(k_sconv_generic, dim3(glx_divup(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, W, ep, nbr, N_out, K, Cin, Cout, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ dispatch
static bool mfma_supported(int Cin, int Cout, int K) {
  auto okc = [](int c) { return c == 16 || c == 32 || c == 64 || c == 128; };
  return (okc(Cin) || Cin == 4 || Cin == 8) && okc(Cout) && K <= SC_MAXK;
}

template <int CIN, int COUT>
static size_t img_bytes() {   // block-kernel image + wave-private-kernel image, per offset
  return (size_t)(SconvCfg<CIN, COUT>::IMG + WpCfg<CIN, COUT>::IMG + RtCfg<CIN, COUT>::IMG +
                  CoCfg<CIN, COUT>::IMG) *
         sizeof(float);
}

template <class F>
static int sc_dispatch(int Cin, int Cout, F&& f) {
#define SC_CASE(A, B) \
  if (Cin == A && Cout == B) return f(std::integral_constant<int, A>{}, std::integral_constant<int, B>{});
  SC_CASE(4, 16) SC_CASE(4, 32) SC_CASE(8, 16) SC_CASE(8, 32)
  SC_CASE(16, 16) SC_CASE(16, 32) SC_CASE(16, 64) SC_CASE(16, 128)
  SC_CASE(32, 16) SC_CASE(32, 32) SC_CASE(32, 64) SC_CASE(32, 128)
  SC_CASE(64, 16) SC_CASE(64, 32) SC_CASE(64, 64) SC_CASE(64, 128)
  SC_CASE(128, 16) SC_CASE(128, 32) SC_CASE(128, 64) SC_CASE(128, 128)
#undef SC_CASE
  glx_set_error("sparse conv: no MFMA kernel for channels (%d,%d)", Cin, Cout);
  return GLX_EINVAL;
}

static size_t packed_bytes(int K, int Cin, int Cout) {
  size_t b = 0;
  sc_dispatch(Cin, Cout, [&](auto ci, auto co) {
    b = (size_t)K * img_bytes<decltype(ci)::value, decltype(co)::value>();
    return 0;
  });
  return b;
}

extern "C" size_t glx_sconv_workspace_bytes(int K, int Cin, int Cout) {
  if (!mfma_supported(Cin, Cout, K)) return 256;
  return glx_align(packed_bytes(K, Cin, Cout)) + 256;
}

// optional per-launch timing: the next sparse-conv launch on this host thread is bracketed by
// these two HIP events (hipExtLaunchKernelGGL start/stop = exactly the kernel's execution).
static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
extern "C" int glx_profile_next_sconv(void* start_event, void* stop_event) {
  g_prof_start = (hipEvent_t)start_event;
  g_prof_stop = (hipEvent_t)stop_event;
  return GLX_OK;
}

template <int CI, int CO>
static int pack_weights(const float* W, int K, float* Wp, hipStream_t st) {
  using C = SconvCfg<CI, CO>;
  size_t pbytes = (size_t)K * C::IMG * sizeof(float);
  if (C::QPAD) GLX_HIP(hipMemsetAsync(Wp, 0, pbytes, st));
  int nel = K * CI * CO;
  hipLaunchKernelGGL((k_pack_weights<CI, CO>), dim3(glx_divup(nel, 256)), dim3(256), 0, st, W, K,
                     Wp);
  hipLaunchKernelGGL((k_pack_weights_wp<CI, CO>), dim3(glx_divup(nel, 256)), dim3(256), 0, st, W,
                     K, Wp + (size_t)K * C::IMG);
  hipLaunchKernelGGL((k_pack_weights_rt<CI, CO>), dim3(glx_divup(nel, 256)), dim3(256), 0, st, W,
                     K, Wp + (size_t)K * (C::IMG + WpCfg<CI, CO>::IMG));
  hipLaunchKernelGGL((k_pack_weights_co<CI, CO>), dim3(glx_divup(nel, 256)), dim3(256), 0, st, W,
                     K, Wp + (size_t)K * (C::IMG + WpCfg<CI, CO>::IMG + RtCfg<CI, CO>::IMG));
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// experiment knob: glx_sconv_set_variant() or env GLX_SCONV_VARIANT; -1 = default kernel choice
static int env_variant() {
  const char* e = getenv("GLX_SCONV_VARIANT");
  return e ? atoi(e) : -1;
}
static int g_sconv_variant = env_variant();
extern "C" int glx_sconv_set_variant(int v) {
  g_sconv_variant = v;
  return GLX_OK;
}

template <int CI, int CO, int TR, int NW, int NBUF>
static int launch_tile(const float* in, const float* Wp, const SconvEpilogue& ep,
                       const int32_t* nbr, const int32_t* tile_order, int N_out, int K, float* out,
                       hipStream_t st) {
  using T = SconvTile<CI, CO, TR, NW, NBUF>;
  if constexpr (T::lds_bytes > 160 * 1024) {
    glx_set_error("sparse conv tile (%d,%d,TR=%d,NW=%d,NBUF=%d) needs %zu B of LDS", CI, CO, TR, NW,
                  NBUF, (size_t)T::lds_bytes);
    return GLX_EINVAL;
  } else {
    static bool attr_set = false;   // one per instantiation
    auto kern = k_sconv_mfma<CI, CO, TR, NW, NBUF>;
    const size_t lds = T::lds_bytes;
    if (!attr_set) {
      GLX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
      attr_set = true;
    }
    int nblocks = glx_divup(N_out, TR);
    if (g_prof_start && g_prof_stop) {
      hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(T::THREADS), lds, st, g_prof_start,
                            g_prof_stop, 0, in, Wp, ep, nbr, tile_order, N_out, K, out);
      g_prof_start = g_prof_stop = nullptr;
    } else {
      hipLaunchKernelGGL(kern, dim3(nblocks), dim3(T::THREADS), lds, st, in, Wp, ep, nbr,
                         tile_order, N_out, K, out);
    }
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
}

template <int CI, int CO>
static int launch_mfma(const float* in, const float* Wp, const SconvEpilogue& ep,
                       const int32_t* nbr, const int32_t* tile_order, int N_out, int K, float* out,
                       const float* zero_row, hipStream_t st) {
#define SC_GO(TR, NW, NBUF) \
  return launch_tile<CI, CO, TR, NW, NBUF>(in, Wp, ep, nbr, tile_order, N_out, K, out, st)
  constexpr bool big = CO >= 128;
  switch (g_sconv_variant) {
    case 1: SC_GO(128, 4, 1);
    case 2: SC_GO(128, 8, 1);
    case 3: SC_GO(128, 4, 2);
    case 4: SC_GO(128, 8, 2);
    case 5: SC_GO(64, 4, 1);
    case 6: SC_GO(64, 4, 2);
    case 7: if constexpr (!big) { SC_GO(256, 8, 1); } else { SC_GO(128, 8, 1); }
    case 10: SC_GO(32, 4, 1);
    case 11: SC_GO(32, 4, 2);
    case 12: SC_GO(64, 8, 1);
    case 13: SC_GO(32, 2, 1);
    case 0:
      if constexpr (big) {
        if constexpr (CI >= 128) { SC_GO(128, 8, 1); } else { SC_GO(128, 8, 2); }
      } else {
        SC_GO(256, 8, 2);
      }
    default: break;
  }
#undef SC_GO
  if (g_sconv_variant == 8) {   // wave-private LDS-accumulating kernel
    using C = WpCfg<CI, CO>;
    static bool attr_set = false;
    auto kern = k_sconv_wp<CI, CO>;
    if (!attr_set) {
      GLX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)C::lds_bytes));
      attr_set = true;
    }
    const float* Wp2 = Wp + (size_t)K * SconvCfg<CI, CO>::IMG;
    int nblocks = glx_divup(N_out, C::TRW * C::NWB);
    if (g_prof_start && g_prof_stop) {
      hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(C::NWB * 64), C::lds_bytes, st, g_prof_start,
                            g_prof_stop, 0, in, Wp2, ep, nbr, tile_order, N_out, K, out);
      g_prof_start = g_prof_stop = nullptr;
    } else {
      hipLaunchKernelGGL(kern, dim3(nblocks), dim3(C::NWB * 64), C::lds_bytes, st, in, Wp2, ep, nbr,
                         tile_order, N_out, K, out);
    }
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  if (g_sconv_variant == 15) {   // wave-private kernel with the deep software pipeline
    using L = WpCfg<CI, CO>;
    static bool attr_set = false;
    auto kern = k_sconv_wq<CI, CO>;
    if (!attr_set) {
      GLX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)L::lds_bytes));
      attr_set = true;
    }
    const float* Wp3 = Wp + (size_t)K * (SconvCfg<CI, CO>::IMG + WpCfg<CI, CO>::IMG);
    int nblocks = glx_divup(N_out, L::TRW * L::NWB);
    if (g_prof_start && g_prof_stop) {
      hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(L::NWB * 64), L::lds_bytes, st, g_prof_start,
                            g_prof_stop, 0, in, Wp3, ep, nbr, tile_order, zero_row, N_out, K, out);
      g_prof_start = g_prof_stop = nullptr;
    } else {
      hipLaunchKernelGGL(kern, dim3(nblocks), dim3(L::NWB * 64), L::lds_bytes, st, in, Wp3, ep, nbr,
                         tile_order, zero_row, N_out, K, out);
    }
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  if (g_sconv_variant == 14) {   // column-owner kernel
    using C = CoCfg<CI, CO>;
    static bool attr_set = false;
    auto kern = k_sconv_co<CI, CO>;
    if (!attr_set) {
      GLX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)C::lds_bytes));
      attr_set = true;
    }
    const float* Wp4 = Wp + (size_t)K * (SconvCfg<CI, CO>::IMG + WpCfg<CI, CO>::IMG + RtCfg<CI, CO>::IMG);
    int nblocks = glx_divup(N_out, C::TR);
    if (g_prof_start && g_prof_stop) {
      hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(C::NW * 64), C::lds_bytes, st, g_prof_start,
                            g_prof_stop, 0, in, Wp4, ep, nbr, tile_order, N_out, K, out);
      g_prof_start = g_prof_stop = nullptr;
    } else {
      hipLaunchKernelGGL(kern, dim3(nblocks), dim3(C::NW * 64), C::lds_bytes, st, in, Wp4, ep, nbr,
                         tile_order, N_out, K, out);
    }
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  if (g_sconv_variant == 9) {   // register-tile kernel, one wave per 32 output rows
    using R = RtCfg<CI, CO>;
    const float* Wp3 = Wp + (size_t)K * (SconvCfg<CI, CO>::IMG + WpCfg<CI, CO>::IMG);
    int nblocks = glx_divup(N_out, R::ROWS);
    auto kern = k_sconv_rt<CI, CO>;
    if (g_prof_start && g_prof_stop) {
      hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(64), 0, st, g_prof_start, g_prof_stop, 0, in,
                            Wp3, ep, nbr, tile_order, N_out, K, out);
      g_prof_start = g_prof_stop = nullptr;
    } else {
      hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64), 0, st, in, Wp3, ep, nbr, tile_order, N_out,
                         K, out);
    }
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  // default (measured best on the KITTI-shaped batch, tools/sconv_sweep.py): block kernel,
  // 64 rows x 4 waves for narrow outputs, 128 rows x 8 waves for 128 output channels
#define SC_GO(TR, NW, NBUF) \
  return launch_tile<CI, CO, TR, NW, NBUF>(in, Wp, ep, nbr, tile_order, N_out, K, out, st)
  if constexpr (CO >= 128) { SC_GO(128, 8, 1); } else { SC_GO(64, 4, 1); }
#undef SC_GO
}

extern "C" size_t glx_sconv_packed_bytes(int K, int Cin, int Cout) {
  if (!mfma_supported(Cin, Cout, K)) return 0;
  return packed_bytes(K, Cin, Cout);
}

extern "C" int glx_sconv_pack_weights(const float* W, int K, int Cin, int Cout, float* Wp,
                                      void* stream) {
  GLX_REQUIRE(W && Wp, "glx_sconv_pack_weights: null pointer");
  GLX_REQUIRE(mfma_supported(Cin, Cout, K),
              "glx_sconv_pack_weights: no MFMA kernel for (K=%d, Cin=%d, Cout=%d)", K, Cin, Cout);
  hipStream_t st = (hipStream_t)stream;
  return sc_dispatch(Cin, Cout, [&](auto ci, auto co) {
    return pack_weights<decltype(ci)::value, decltype(co)::value>(W, K, Wp, st);
  });
}

extern "C" int glx_sconv_forward(const float* in, int N_in, const float* W, const float* Wp,
                                 const float* bias, const float* scale, const float* shift,
                                 int relu, const int32_t* nbr, const int32_t* tile_order,
                                 int N_out, int K, int Cin, int Cout, float* out,
                                 const int32_t* n_out_live, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(K > 0 && Cin > 0 && Cout > 0 && N_out >= 0, "glx_sconv_forward: bad sizes");
  if (N_out == 0) return GLX_OK;
  GLX_REQUIRE(in && (W || Wp) && nbr && out, "glx_sconv_forward: null pointer");
  SconvEpilogue ep{bias, scale, shift, relu, n_out_live};
  if (!mfma_supported(Cin, Cout, K)) {
    GLX_REQUIRE(W, "glx_sconv_forward: raw weights required for channels (%d,%d)", Cin, Cout);
    long long total = (long long)N_out * Cout;
    hipLaunchKernelGGL(k_sconv_generic, dim3(glx_divup(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, in, W, ep, nbr, N_out, K, Cin, Cout, out);
    GLX_LAUNCH_CHECK();
    return GLX_OK;
  }
  hipStream_t st = (hipStream_t)stream;
  if (!Wp) {  // pack on the fly into the caller's workspace
    size_t need = packed_bytes(K, Cin, Cout);
    if (!workspace || workspace_bytes < need) {
      glx_set_error("glx_sconv_forward: workspace %zu < %zu bytes", workspace_bytes, need);
      return GLX_EWORKSPACE;
    }
    int rc = glx_sconv_pack_weights(W, K, Cin, Cout, (float*)workspace, stream);
    if (rc != GLX_OK) return rc;
    Wp = (const float*)workspace;
  }
  // first 1 KB of the workspace = the zero row that absent rule pairs gather from
  const float* zero_row = nullptr;
  if (g_sconv_variant == 15 && workspace && workspace_bytes >= 1024 &&
      Wp != (const float*)workspace) {
    GLX_HIP(hipMemsetAsync(workspace, 0, 1024, st));
    zero_row = (const float*)workspace;
  }
  if (g_sconv_variant == 15 && !zero_row) {
    glx_set_error("glx_sconv_forward: variant 15 needs pre-packed weights and >= 1 KB of workspace");
    return GLX_EWORKSPACE;
  }
  return sc_dispatch(Cin, Cout, [&](auto ci, auto co) {
    return launch_mfma<decltype(ci)::value, decltype(co)::value>(in, Wp, ep, nbr, tile_order,
                                                                 N_out, K, out, zero_row, st);
  });
}

// ------------------------------------------------------------------ weight transpose
__global__ void k_transpose_w(const float* __restrict__ W, int K, int Cin, int Cout,
                              float* __restrict__ Wt) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= K * Cin * Cout) return;
  int ci = e % Cin;
  int co = (e / Cin) % Cout;
  int k = e / (Cin * Cout);
  Wt[e] = W[((long long)k * Cin + ci) * Cout + co];
}

extern "C" int glx_sconv_transpose_weights(const float* W, int K, int Cin, int Cout, float* Wt,
                                           void* stream) {
  GLX_REQUIRE(W && Wt && K > 0 && Cin > 0 && Cout > 0, "glx_sconv_transpose_weights: bad args");
  int nel = K * Cin * Cout;
  hipLaunchKernelGGL(k_transpose_w, dim3(glx_divup(nel, 256)), dim3(256), 0, (hipStream_t)stream,
                     W, K, Cin, Cout, Wt);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ weight gradient
// dW[k][ci][co] = sum_j in[nbr[j,k]][ci] * gout[j][co].
// grid (chunks, K): each block reduces one offset over a chunk of output rows into a
// private slab, a second kernel sums the slabs in fixed order (bitwise reproducible,
// no float atomics: cdna_hip_programming.md Guideline 12).
#define WG_THREADS 256
#define WG_CHUNKS 64

__global__ __launch_bounds__(WG_THREADS) void k_wgrad_partial(
    const float* __restrict__ in, const float* __restrict__ gout, const int* __restrict__ nbr,
    int N_out, int K, int Cin, int Cout, int rows_per_chunk, float* __restrict__ slabs) {
  // thread owns elements e = tid, tid+256, ... of the Cin*Cout tile (<= 64 each for 128x128)
  const int k = blockIdx.y;
  const int chunk = blockIdx.x;
  const int j0 = chunk * rows_per_chunk;
  const int j1 = min(N_out, j0 + rows_per_chunk);
  const int nel = Cin * Cout;
  extern __shared__ float s[];  // [Cin] input row, [Cout] grad row
  float* s_in = s;
  float* s_g = s + Cin;
  float acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = 0.f;
  for (int j = j0; j < j1; ++j) {
    int i = nbr[(long long)j * K + k];
    if (i < 0) continue;  // block-uniform
    __syncthreads();
    for (int c = threadIdx.x; c < Cin; c += WG_THREADS) s_in[c] = in[(long long)i * Cin + c];
    for (int c = threadIdx.x; c < Cout; c += WG_THREADS) s_g[c] = gout[(long long)j * Cout + c];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      int e = threadIdx.x + u * WG_THREADS;
      if (e < nel) acc[u] = fmaf(s_in[e / Cout], s_g[e % Cout], acc[u]);
    }
  }
  float* dst = slabs + ((long long)chunk * K + k) * nel;
#pragma unroll
  for (int u = 0; u < 64; ++u) {
    int e = threadIdx.x + u * WG_THREADS;
    if (e < nel) dst[e] = acc[u];
  }
}

__global__ void k_wgrad_reduce(const float* __restrict__ slabs, int nchunks, long long nel_total,
                               float* __restrict__ dW) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nel_total) return;
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += slabs[(long long)c * nel_total + e];
  dW[e] = s;
}

extern "C" size_t glx_sconv_wgrad_workspace_bytes(int N_out, int K, int Cin, int Cout) {
  (void)N_out;
  return glx_align((size_t)WG_CHUNKS * K * Cin * Cout * sizeof(float)) + 256;
}

extern "C" int glx_sconv_wgrad(const float* in, int N_in, const float* grad_out,
                               const int32_t* nbr, int N_out, int K, int Cin, int Cout, float* dW,
                               void* workspace, size_t workspace_bytes, void* stream) {
  (void)N_in;
  GLX_REQUIRE(dW && (N_out == 0 || (in && grad_out && nbr)), "glx_sconv_wgrad: null pointer");
  GLX_REQUIRE(Cin * Cout <= 64 * WG_THREADS, "glx_sconv_wgrad: Cin*Cout=%d too large", Cin * Cout);
  hipStream_t st = (hipStream_t)stream;
  long long nel_total = (long long)K * Cin * Cout;
  if (N_out == 0) {
    GLX_HIP(hipMemsetAsync(dW, 0, nel_total * sizeof(float), st));
    return GLX_OK;
  }
  size_t need = (size_t)WG_CHUNKS * nel_total * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    glx_set_error("glx_sconv_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
    return GLX_EWORKSPACE;
  }
  int rows_per_chunk = glx_divup(N_out, WG_CHUNKS);
  hipLaunchKernelGGL(k_wgrad_partial, dim3(WG_CHUNKS, K), dim3(WG_THREADS),
                     (Cin + Cout) * sizeof(float), st, in, grad_out, nbr, N_out, K, Cin, Cout,
                     rows_per_chunk, (float*)workspace);
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(glx_divup(nel_total, 256)), dim3(256), 0, st,
                     (const float*)workspace, WG_CHUNKS, nel_total, dW);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ dense()
__global__ void k_dense_scatter(const float* __restrict__ f, const int4* __restrict__ idx, int N,
                                int C, int D, int H, int W, float* __restrict__ out,
                                const int* __restrict__ n_live) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_live) N = min(N, *n_live);
  if (t >= (long long)N * C) return;
  int row = (int)(t / C);
  int c = (int)(t - (long long)row * C);
  int4 p = idx[row];  // b z y x
  long long o = ((((long long)p.x * C + c) * D + p.y) * H + p.z) * W + p.w;
  out[o] = f[t];
}

extern "C" int glx_dense_scatter(const float* features, const int32_t* indices, int N, int C,
                                 int B, int D, int H, int W, float* out, const int32_t* n_live,
                                 void* stream) {
  (void)B;
  if (N == 0) return GLX_OK;
  GLX_REQUIRE(features && indices && out && C > 0, "glx_dense_scatter: bad arguments");
  long long total = (long long)N * C;
  hipLaunchKernelGGL(k_dense_scatter, dim3(glx_divup(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, features, (const int4*)indices, N, C, D, H, W, out,
                     n_live);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
