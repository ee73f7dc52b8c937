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
#include <type_traits>

#include "glx_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SC_THREADS 256
#define SC_WAVES 4
#define SC_ROWS_PER_WAVE 16
#define SC_ROWS_PER_BLOCK (SC_WAVES * SC_ROWS_PER_WAVE)
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

template <int CIN, int COUT, bool RESIDENT>
__global__ __launch_bounds__(SC_THREADS) void k_sconv_mfma(
    const float* __restrict__ in, const float* __restrict__ Wp, const float* __restrict__ bias,
    const int* __restrict__ nbr, const int* __restrict__ tile_order, int N_out, int K,
    float* __restrict__ out) {
  using C = SconvCfg<CIN, COUT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // layout: [weights: RESIDENT ? K*IMG : 2*IMG] [s_nbr: 64*28 ints] [s_rows: 64 ints] [mask: 4]
  constexpr int WFLOATS_STAGE = C::IMG;
  float* s_w = smem;
  const int wfloats = RESIDENT ? K * WFLOATS_STAGE : 2 * WFLOATS_STAGE;
  int* s_nbr = reinterpret_cast<int*>(smem + wfloats);
  int* s_rows = s_nbr + SC_ROWS_PER_BLOCK * (SC_MAXK + 1);
  int* s_mask = s_rows + SC_ROWS_PER_BLOCK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  if (RESIDENT) {
    // whole filter bank stays in LDS for every tile this block walks
    const float4* src = reinterpret_cast<const float4*>(Wp);
    float4* dst = reinterpret_cast<float4*>(s_w);
    for (int i = tid; i < K * WFLOATS_STAGE / 4; i += SC_THREADS) dst[i] = src[i];
  }

  const int ntiles = (N_out + SC_ROWS_PER_BLOCK - 1) / SC_ROWS_PER_BLOCK;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * SC_ROWS_PER_BLOCK;
    __syncthreads();  // previous tile done with s_nbr / s_rows / s_mask (and weights landed)
    if (tid < SC_ROWS_PER_BLOCK) {
      int p = row0 + tid;
      s_rows[tid] = (p < N_out) ? (tile_order ? tile_order[p] : p) : -1;
    }
    if (tid == 0) s_mask[0] = 0;
    __syncthreads();
    // neighbour lists of the block's 64 rows -> LDS, and the per-wave / per-block offset masks
    unsigned my_mask = 0;
    for (int e = lane; e < SC_ROWS_PER_WAVE * K; e += 64) {
      int rr = e / K, kk = e - rr * K;
      int orow = s_rows[wave * 16 + rr];
      int v = (orow >= 0) ? nbr[(long long)orow * K + kk] : -1;
      s_nbr[(wave * 16 + rr) * (SC_MAXK + 1) + kk] = v;
      if (v >= 0) my_mask |= 1u << kk;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) my_mask |= __shfl_xor(my_mask, o, 64);
    const unsigned wave_mask = __builtin_amdgcn_readfirstlane(my_mask);
    unsigned block_mask = wave_mask;
    if (!RESIDENT) {
      if (lane == 0 && wave_mask) atomicOr(&s_mask[0], wave_mask);
      __syncthreads();
      block_mask = (unsigned)s_mask[0];
    }

    f32x4 acc[C::NT];
#pragma unroll
    for (int c = 0; c < C::NT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int my_nbr_base = (wave * 16 + r) * (SC_MAXK + 1);

    // ---- staged (double buffered) weights: prologue loads first offset
    constexpr int STAGE_F4 = WFLOATS_STAGE / 4;                       // float4 per image
    constexpr int STAGE_PER_THREAD = (STAGE_F4 + SC_THREADS - 1) / SC_THREADS;
    float4 stage_regs[STAGE_PER_THREAD];
    int buf = 0;
    unsigned todo = block_mask;
    if (!RESIDENT && todo) {
      int k0 = __builtin_ctz(todo);
      const float4* src = reinterpret_cast<const float4*>(Wp + (size_t)k0 * WFLOATS_STAGE);
      float4* dst = reinterpret_cast<float4*>(s_w);
#pragma unroll
      for (int i = 0; i < STAGE_PER_THREAD; ++i) {
        int e = tid + i * SC_THREADS;
        if (e < STAGE_F4) dst[e] = src[e];
      }
    }

    while (todo) {
      const int k = __builtin_ctz(todo);
      todo &= todo - 1;
      const float* wimg;
      if (RESIDENT) {
        wimg = s_w + (size_t)k * WFLOATS_STAGE;
      } else {
        __syncthreads();  // image k visible; other buffer free
        wimg = s_w + buf * WFLOATS_STAGE;
        if (todo) {  // issue global loads of the next image now, park them in registers
          int kn = __builtin_ctz(todo);
          const float4* src = reinterpret_cast<const float4*>(Wp + (size_t)kn * WFLOATS_STAGE);
#pragma unroll
          for (int i = 0; i < STAGE_PER_THREAD; ++i) {
            int e = tid + i * SC_THREADS;
            if (e < STAGE_F4) stage_regs[i] = src[e];
          }
        }
      }

      if (wave_mask & (1u << k)) {
        const int irow = s_nbr[my_nbr_base + k];
        float a[C::CQ];
        if (irow >= 0) {
          const float* ap = in + (long long)irow * CIN + q * C::CQ;
          if constexpr (C::CQ % 4 == 0) {
#pragma unroll
            for (int i = 0; i < C::CQ / 4; ++i) {
              float4 v = reinterpret_cast<const float4*>(ap)[i];
              a[4 * i + 0] = v.x; a[4 * i + 1] = v.y; a[4 * i + 2] = v.z; a[4 * i + 3] = v.w;
            }
          } else {
#pragma unroll
            for (int i = 0; i < C::CQ; ++i) a[i] = ap[i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < C::CQ; ++i) a[i] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < C::CQ; ++t) {
#pragma unroll
          for (int h = 0; h < C::NH; ++h) {
            typedef typename FVec<C::NC>::T BV;
            BV bv = *reinterpret_cast<const BV*>(wimg + (h * 4 + q) * C::QSTRIDE +
                                                  (t * 16 + r) * C::NC);
            const float* bp = reinterpret_cast<const float*>(&bv);
#pragma unroll
            for (int c = 0; c < C::NC; ++c) {
              acc[h * C::NC + c] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bp[c], acc[h * C::NC + c], 0, 0, 0);
            }
          }
        }
      }

      if (!RESIDENT) {
        if (todo) {
          float4* dst = reinterpret_cast<float4*>(s_w + (buf ^ 1) * WFLOATS_STAGE);
#pragma unroll
          for (int i = 0; i < STAGE_PER_THREAD; ++i) {
            int e = tid + i * SC_THREADS;
            if (e < STAGE_F4) dst[e] = stage_regs[i];
          }
        }
        buf ^= 1;
      }
    }

    // ---- epilogue: lane (n=r, q) holds rows 4q..4q+3, column 16*ct + n
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int orow = s_rows[wave * 16 + 4 * q + reg];
      if (orow >= 0) {
        float* op = out + (long long)orow * COUT + r;
#pragma unroll
        for (int ct = 0; ct < C::NT; ++ct) {
          float v = acc[ct][reg];
          if (bias) v += bias[ct * 16 + r];
          op[ct * 16] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------ generic scalar kernel
__global__ void k_sconv_generic(const float* __restrict__ in, const float* __restrict__ W,
                                const float* __restrict__ bias, const int* __restrict__ nbr,
                                int N_out, int K, int Cin, int Cout, float* __restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N_out * Cout) return;
  int j = (int)(t / Cout);
  int co = (int)(t - (long long)j * Cout);
  float acc = bias ? bias[co] : 0.f;
  for (int k = 0; k < K; ++k) {
    int i = nbr[(long long)j * K + k];
    if (i < 0) continue;
    const float* ip = in + (long long)i * Cin;
    const float* wp = W + ((long long)k * Cin) * Cout + co;
    for (int ci = 0; ci < Cin; ++ci) acc = fmaf(ip[ci], wp[(long long)ci * Cout], acc);
  }
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
  hipLaunchKernelGGL(k_sconv_generic, dim3(glx_divup(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, W, bias, nbr, N_out, K, Cin, Cout, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ dispatch
static bool mfma_supported(int Cin, int Cout, int K) {
  auto okc = [](int c) { return c == 16 || c == 32 || c == 64 || c == 128; };
  return okc(Cin) && okc(Cout) && K <= SC_MAXK;
}

template <int CIN, int COUT>
static size_t img_bytes() {
  return (size_t)SconvCfg<CIN, COUT>::IMG * sizeof(float);
}

template <class F>
static int sc_dispatch(int Cin, int Cout, F&& f) {
#define SC_CASE(A, B) \
  if (Cin == A && Cout == B) return f(std::integral_constant<int, A>{}, std::integral_constant<int, B>{});
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

#define SC_RESIDENT_LIMIT (72 * 1024)
#define SC_TILE_LDS ((SC_ROWS_PER_BLOCK * (SC_MAXK + 1) + SC_ROWS_PER_BLOCK + 4) * sizeof(int))

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
static int launch_mfma(const float* in, const float* Wp, const float* bias, const int32_t* nbr,
                       const int32_t* tile_order, int N_out, int K, float* out, hipStream_t st) {
  using C = SconvCfg<CI, CO>;
  size_t pbytes = (size_t)K * C::IMG * sizeof(float);
  int ntiles = glx_divup(N_out, SC_ROWS_PER_BLOCK);
  bool resident = pbytes <= SC_RESIDENT_LIMIT;
  size_t lds = (resident ? pbytes : 2 * (size_t)C::IMG * sizeof(float)) + SC_TILE_LDS;
  if (resident) {
    int grid = ntiles < 512 ? ntiles : 512;
    auto kern = k_sconv_mfma<CI, CO, true>;
    GLX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SC_THREADS), lds, st, in, Wp, bias, nbr, tile_order,
                       N_out, K, out);
  } else {
    auto kern = k_sconv_mfma<CI, CO, false>;
    GLX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(SC_THREADS), lds, st, in, Wp, bias, nbr,
                       tile_order, N_out, K, out);
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
                                 const float* bias, const int32_t* nbr, const int32_t* tile_order,
                                 int N_out, int K, int Cin, int Cout, float* out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(K > 0 && Cin > 0 && Cout > 0 && N_out >= 0, "glx_sconv_forward: bad sizes");
  if (N_out == 0) return GLX_OK;
  GLX_REQUIRE(in && (W || Wp) && nbr && out, "glx_sconv_forward: null pointer");
  if (!mfma_supported(Cin, Cout, K)) {
    GLX_REQUIRE(W, "glx_sconv_forward: raw weights required for channels (%d,%d)", Cin, Cout);
    return glx_sconv_forward_generic(in, N_in, W, bias, nbr, N_out, K, Cin, Cout, out, stream);
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
    return launch_mfma<decltype(ci)::value, decltype(co)::value>(in, Wp, bias, nbr, tile_order,
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
