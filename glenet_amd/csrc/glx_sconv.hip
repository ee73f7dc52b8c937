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
};

#define SC_NW 8                      // waves per block
#define SC_THREADS (SC_NW * 64)

template <int CIN, int COUT>
struct SconvTile {
  using C = SconvCfg<CIN, COUT>;
  static constexpr int TR = COUT >= 128 ? 128 : 256;          // output rows per block
  static constexpr int MAXC = TR / (16 * SC_NW);              // chunks per wave per offset
  static constexpr int ACC_LD = COUT + 4;
  static constexpr size_t fixed_bytes =
      (size_t)TR * ACC_LD * 4 + (size_t)SC_MAXK * TR * 5 + (TR + 32 + 32 * (TR / 64)) * 4 + 64;
  static constexpr int NBUF = (fixed_bytes + 2 * (size_t)C::IMG * 4 <= 160 * 1024) ? 2 : 1;
  static constexpr size_t lds_bytes = fixed_bytes + (size_t)NBUF * C::IMG * 4;
};

// gather the CQ-float slice of up to MAXC chunks of offset k owned by this wave
template <int CIN, int COUT>
__device__ __forceinline__ void sc_gather(const float* __restrict__ in, const int* s_pin,
                                          int k, int cnt, int wave, int r, int q,
                                          float (&A)[SconvTile<CIN, COUT>::MAXC][SconvCfg<CIN, COUT>::CQ]) {
  using C = SconvCfg<CIN, COUT>;
  using T = SconvTile<CIN, COUT>;
#pragma unroll
  for (int j = 0; j < T::MAXC; ++j) {
    const int c = ((wave - k) & (SC_NW - 1)) + j * SC_NW;
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
      if (irow < 0) {
#pragma unroll
        for (int i = 0; i < C::CQ; ++i) A[j][i] = 0.f;
      }
    }
  }
}

// MFMA the wave's chunks of offset k against the staged W[k] and add into the LDS tile
template <int CIN, int COUT>
__device__ __forceinline__ void sc_compute(const float* s_w, float* s_acc,
                                           const unsigned char* s_pslot, int k, int cnt, int wave,
                                           int r, int q,
                                           const float (&A)[SconvTile<CIN, COUT>::MAXC][SconvCfg<CIN, COUT>::CQ]) {
  using C = SconvCfg<CIN, COUT>;
  using T = SconvTile<CIN, COUT>;
#pragma unroll
  for (int j = 0; j < T::MAXC; ++j) {
    const int c = ((wave - k) & (SC_NW - 1)) + j * SC_NW;
    if (c * 16 >= cnt) continue;   // wave-uniform
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
          acc[h * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[0], A[j][t], acc[h * 4 + 0], 0, 0, 0);
          acc[h * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[1], A[j][t], acc[h * 4 + 1], 0, 0, 0);
          acc[h * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[2], A[j][t], acc[h * 4 + 2], 0, 0, 0);
          acc[h * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[3], A[j][t], acc[h * 4 + 3], 0, 0, 0);
        } else if constexpr (C::NC == 2) {
          float2 bv = *reinterpret_cast<const float2*>(bp);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv.x, A[j][t], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv.y, A[j][t], acc[1], 0, 0, 0);
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bp[0], A[j][t], acc[0], 0, 0, 0);
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
template <int CIN, int COUT>
__global__ __launch_bounds__(SC_THREADS) void k_sconv_mfma(
    const float* __restrict__ in, const float* __restrict__ Wp, SconvEpilogue ep,
    const int* __restrict__ nbr, const int* __restrict__ tile_order, int N_out, int K,
    float* __restrict__ out) {
  using C = SconvCfg<CIN, COUT>;
  using T = SconvTile<CIN, COUT>;
  constexpr int TR = T::TR, ACC_LD = T::ACC_LD, NBUF = T::NBUF, LW = TR / 64;
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
  unsigned rem = mask;
  int k = -1, cnt = 0;
  if (rem) {
    k = __builtin_ctz(rem);
    rem &= rem - 1;
    cnt = s_cnt[k];
    SC_STAGE_LOAD(k);
    SC_STAGE_STORE(0);
    sc_gather<CIN, COUT>(in, s_pin, k, cnt, wave, r, q, A0);
  }
  __syncthreads();

  int buf = 0;
  // one phase: prefetch (W image + input rows) of the next offset, multiply the current one
#define SC_PHASE(CUR, NXT)                                                                  \
  {                                                                                         \
    int kn_ = -1, cntn_ = 0;                                                                \
    if (rem) {                                                                              \
      kn_ = __builtin_ctz(rem);                                                             \
      rem &= rem - 1;                                                                       \
      cntn_ = s_cnt[kn_];                                                                   \
      SC_STAGE_LOAD(kn_);                                                                   \
      sc_gather<CIN, COUT>(in, s_pin, kn_, cntn_, wave, r, q, NXT);                         \
    }                                                                                       \
    sc_compute<CIN, COUT>(s_w + (NBUF == 2 ? buf : 0) * C::IMG, s_acc, s_pslot, k, cnt,     \
                          wave, r, q, CUR);                                                 \
    if (NBUF == 1) __syncthreads();                                                         \
    if (kn_ >= 0) SC_STAGE_STORE(NBUF == 2 ? (buf ^ 1) : 0);                                \
    __syncthreads();                                                                        \
    buf ^= 1;                                                                               \
    k = kn_;                                                                                \
    cnt = cntn_;                                                                            \
  }
  while (k >= 0) {
    SC_PHASE(A0, A1);
    if (k < 0) break;
    SC_PHASE(A1, A0);
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

// ------------------------------------------------------------------ generic scalar kernel
__global__ void k_sconv_generic(const float* __restrict__ in, const float* __restrict__ W,
                                SconvEpilogue ep, const int* __restrict__ nbr, int N_out, int K,
                                int Cin, int Cout, float* __restrict__ out) {
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
  SconvEpilogue ep{bias, nullptr, nullptr, 0};
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
static size_t img_bytes() {
  return (size_t)SconvCfg<CIN, COUT>::IMG * sizeof(float);
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
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

template <int CI, int CO>
static int launch_mfma(const float* in, const float* Wp, const SconvEpilogue& ep,
                       const int32_t* nbr, const int32_t* tile_order, int N_out, int K, float* out,
                       hipStream_t st) {
  using T = SconvTile<CI, CO>;
  static bool attr_set = false;   // one instantiation per (CI, CO)
  auto kern = k_sconv_mfma<CI, CO>;
  const size_t lds = T::lds_bytes;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    attr_set = true;
  }
  int nblocks = glx_divup(N_out, T::TR);
  if (g_prof_start && g_prof_stop) {
    hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(SC_THREADS), lds, st, g_prof_start,
                          g_prof_stop, 0, in, Wp, ep, nbr, tile_order, N_out, K, out);
    g_prof_start = g_prof_stop = nullptr;
  } else {
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(SC_THREADS), lds, st, in, Wp, ep, nbr, tile_order,
                       N_out, K, out);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
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
                                 int N_out, int K, int Cin, int Cout, float* out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(K > 0 && Cin > 0 && Cout > 0 && N_out >= 0, "glx_sconv_forward: bad sizes");
  if (N_out == 0) return GLX_OK;
  GLX_REQUIRE(in && (W || Wp) && nbr && out, "glx_sconv_forward: null pointer");
  SconvEpilogue ep{bias, scale, shift, relu};
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
  return sc_dispatch(Cin, Cout, [&](auto ci, auto co) {
    return launch_mfma<decltype(ci)::value, decltype(co)::value>(in, Wp, ep, nbr, tile_order,
                                                                 N_out, K, out, st);
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
                                int C, int D, int H, int W, float* __restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * C) return;
  int row = (int)(t / C);
  int c = (int)(t - (long long)row * C);
  int4 p = idx[row];  // b z y x
  long long o = ((((long long)p.x * C + c) * D + p.y) * H + p.z) * W + p.w;
  out[o] = f[t];
}

extern "C" int glx_dense_scatter(const float* features, const int32_t* indices, int N, int C,
                                 int B, int D, int H, int W, float* out, void* stream) {
  (void)B;
  if (N == 0) return GLX_OK;
  GLX_REQUIRE(features && indices && out && C > 0, "glx_dense_scatter: bad arguments");
  long long total = (long long)N * C;
  hipLaunchKernelGGL(k_dense_scatter, dim3(glx_divup(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, features, (const int4*)indices, N, C, D, H, W, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
