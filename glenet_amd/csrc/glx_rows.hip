// Conv(kernel 1, no bias) + training-mode BatchNorm (+ ReLU) on (rows, C) matrices: the input and output MLPs of the RoI-grid
// pool (pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:70-130: mlps_in on the scale's voxels, mlps_out on the
// pooled grid points; Conv1d / Conv2d with a 1 x 1 kernel there) and any other tall-skinny x @ W^T with <= 64 channels.
//
// Shape of the work: 20 000 .. 110 000 rows, 16 .. 64 channels in, 16 .. 64 out: a stream of rows through an 8 KB weight
// matrix -- 28 .. 42 MB of traffic and 0.2 .. 0.9 GFLOP per call, i.e. ~5 us of HBM time and ~5 us of fp32 MFMA time.  The
// vendor GEMMs spend 17 .. 46 us on each of these products (32 x 32 macro tiles, one launch per product) and leave autograd
// a second product, a split-K batched product and its sum for the backward: 3 + 5 launches per layer with the BatchNorm's.
//
//   forward (1 launch + the transform):  z = x W^T, 16 rows per wave and trip: W (<= 64 x 64) lives in registers as MFMA A
//     operands, the rows stream through as B operands straight from global memory (16-byte loads), so a lane ends up with
//     FOUR CONSECUTIVE CHANNELS of one row (16-byte stores).  The BatchNorm statistics (sum z, sum z^2, fp64) are taken from
//     the accumulators; the block that draws the last ticket turns them into scale / shift, saved mean / invstd and the
//     running statistics (glx_bn_state.h: the scheme of k_bn_stats and the sparse convolutions' epilogue).
//   backward (the statistics launch of glx_bn.hip + 2 launches):  dz = a (dy [y > 0] - b - c xhat) is formed ON LOAD from the
//     3 C coefficients, then  dx = dz W  (k = output channels: the tile as loaded)  and  dW += dz^T x  (k = rows: the tile
//     transposed through a wave-private LDS patch) from the same registers; per-block partial dW, summed by a second launch in
//     a fixed order (bitwise reproducible).
// v_mfma_f32_16x16x4_f32 throughout: exact fp32 products, fp32 accumulation.
#include "glx_common.h"
#include "glx_bn_state.h"

typedef float ft4 __attribute__((ext_vector_type(4)));

#define RW_THREADS 256
#define RW_WAVES 4
#define RW_MAX_BLOCKS 512          // forward: two blocks per CU
#define RW_BWD_BLOCKS 256          // backward: one partial weight gradient per block

// ------------------------------------------------------------------------------------------------ forward
template <int K, int N, bool STATS>
__global__ __launch_bounds__(RW_THREADS) void k_rows_linear(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ z, int rows, const int* __restrict__ n_live,
                                                            BnState* __restrict__ st, BnFinalize f, int ldz) {
  // blockIdx.y: the block's N of the ldz output channels (a 128-wide layer = two column halves of 64: the weights of a half fit
  // the registers, the BatchNorm is per channel anyway)
  constexpr int KS = K / 16, NT = N / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int coff = blockIdx.y * N;
  w += (size_t)coff * K;
  z += coff;
  int n = rows;
  if (n_live) n = min(rows, *n_live);
  ft4 wv[NT][KS];          // A operand: output channel 16 nt + r, input channels 16 s + 4 q + e
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int s = 0; s < KS; ++s) wv[nt][s] = *reinterpret_cast<const ft4*>(w + (16 * nt + r) * K + 16 * s + 4 * q);
  double s0[NT][4], s1[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) s0[nt][i] = s1[nt][i] = 0;
  const int ntiles = (n + 15) >> 4, stride = gridDim.x * RW_WAVES;
  int tile = blockIdx.x * RW_WAVES + wave;
  ft4 xv[KS];
  if (tile < ntiles) {
    const int row = min(tile * 16 + r, n - 1);
#pragma unroll
    for (int s = 0; s < KS; ++s) xv[s] = *reinterpret_cast<const ft4*>(x + (long long)row * K + 16 * s + 4 * q);
  }
  while (tile < ntiles) {
    const int next = tile + stride;
    ft4 xn[KS];
    if (next < ntiles) {
      const int row = min(next * 16 + r, n - 1);
#pragma unroll
      for (int s = 0; s < KS; ++s) xn[s] = *reinterpret_cast<const ft4*>(x + (long long)row * K + 16 * s + 4 * q);
    }
    ft4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[nt] = ft4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[nt][s][e], xv[s][e], acc[nt], 0, 0, 0);
    }
    const int row = tile * 16 + r;
    if (row < n) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        *reinterpret_cast<ft4*>(z + (long long)row * ldz + 16 * nt + 4 * q) = acc[nt];
        if (STATS) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            s0[nt][i] += (double)acc[nt][i];
            s1[nt][i] += (double)acc[nt][i] * (double)acc[nt][i];
          }
        }
      }
    }
    if (next < ntiles) {
#pragma unroll
      for (int s = 0; s < KS; ++s) xv[s] = xn[s];
    }
    tile = next;
  }
  if (!STATS) return;
  // the 16 row lanes of a channel quad, then the block's four waves: thread t < N / 4 ends up with float4 column t
  __shared__ double s_red[RW_WAVES][N / 4][2][4];
  __shared__ int s_last;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        s0[nt][i] += __shfl_xor(s0[nt][i], o, 64);
        s1[nt][i] += __shfl_xor(s1[nt][i], o, 64);
      }
  if (r == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s_red[wave][4 * nt + q][0][i] = s0[nt][i];
        s_red[wave][4 * nt + q][1][i] = s1[nt][i];
      }
  }
  __syncthreads();
  double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
  if ((int)threadIdx.x < N / 4) {
#pragma unroll
    for (int wv_ = 0; wv_ < RW_WAVES; ++wv_)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0[i] += s_red[wv_][threadIdx.x][0][i];
        a1[i] += s_red[wv_][threadIdx.x][1][i];
      }
  }
  if (!bn_contribute(st, N, a0, a1, gridDim.x * gridDim.y, &s_last, coff)) return;
  __shared__ double s_fin[RW_THREADS][2];
  bn_finalize_sets<false, RW_THREADS>(st, f, ldz, n, s_fin);
}

// ------------------------------------------------------------------------------------------------ backward
struct RowsBwdBn {
  const float* coef_fwd;   // scale, shift of the forward transform (2 CO): the ReLU mask is re-derived from z
  const float* coef3;      // a, b, c of dz = a (g - b - c xhat) (3 CO), from the statistics launch
  const float* mean;
  const float* invstd;
  int relu;
};

// element ((a * NB + b) * 4 + i) * 64 + lane of a block's partial = dW[16 a + 4 (lane >> 4) + i][16 b + (lane & 15)]
template <int CI, int CO, bool BN>
__global__ __launch_bounds__(RW_THREADS) void k_rows_linear_bwd(const float* __restrict__ x, const float* __restrict__ z,
                                                                const float* __restrict__ dy, const float* __restrict__ w,
                                                                int rows, const int* __restrict__ n_live, RowsBwdBn bn,
                                                                float* __restrict__ gx, float* __restrict__ part, int ldo, int coff,
                                                                int accumulate) {
  // the launch's CO of the layer's ldo output channels start at `coff` (dy, z, w, the coefficients); accumulate: gx += (the
  // second half of a 128-wide layer, launched behind the first)
  constexpr int MT = CI / 16, KS = CO / 16, NA = CO / 16, NB = CI / 16, LD = CO + 16;
  w += (size_t)coff * CI;
  dy += coff;
  if (BN) z += coff;
  __shared__ float s_dz[RW_WAVES][16 * LD];
  __shared__ float s_acc[NA * NB * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  int n = rows;
  if (n_live) n = min(rows, *n_live);
  ft4 wt[MT][KS];          // input gradient, A operand: input channel 16 mt + r, k = output channels 16 s + 4 q + e
  if (gx) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) wt[mt][s][e] = w[(16 * s + 4 * q + e) * CI + 16 * mt + r];
  }
  ft4 c_sc[KS], c_sh[KS], c_mu[KS], c_is[KS], c_a[KS], c_b[KS], c_cc[KS];
  if (BN) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = coff + 16 * s + 4 * q;
      c_sc[s] = *reinterpret_cast<const ft4*>(bn.coef_fwd + c);
      c_sh[s] = *reinterpret_cast<const ft4*>(bn.coef_fwd + ldo + c);
      c_mu[s] = *reinterpret_cast<const ft4*>(bn.mean + c);
      c_is[s] = *reinterpret_cast<const ft4*>(bn.invstd + c);
      c_a[s] = *reinterpret_cast<const ft4*>(bn.coef3 + c);
      c_b[s] = *reinterpret_cast<const ft4*>(bn.coef3 + ldo + c);
      c_cc[s] = *reinterpret_cast<const ft4*>(bn.coef3 + 2 * ldo + c);
    }
  }
  ft4 accw[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) accw[a][b] = ft4{0.f, 0.f, 0.f, 0.f};
  float* dzs = s_dz[wave];
  const int ntiles = (n + 15) >> 4, stride = gridDim.x * RW_WAVES;
  for (int tile = blockIdx.x * RW_WAVES + wave; tile < ntiles; tile += stride) {
    const int row0 = tile * 16;
    // ---- dz of the tile, lane (r, q): row r, channels 16 s + 4 q ..
    const bool live = row0 + r < n;
    const long long ro = (long long)min(row0 + r, n - 1) * ldo + 4 * q;
    ft4 dzv[KS], zv[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      dzv[s] = *reinterpret_cast<const ft4*>(dy + ro + 16 * s);
      if (BN) zv[s] = *reinterpret_cast<const ft4*>(z + ro + 16 * s);
    }
    // the rows of x as the weight gradient's B operand: lane (c, k) = x[row 4 s4 + k][16 b + c]
    float xb[4][NB];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int rr = row0 + 4 * s4 + q;
      const float* xp = x + (long long)min(rr, n - 1) * CI + r;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float v = xp[16 * b];
        xb[s4][b] = rr < n ? v : 0.f;          // a dead row may hold anything (0 x NaN = NaN)
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g = dzv[s][e];
        if (BN) {
          if (bn.relu) g = bn_affine(zv[s][e], c_sc[s][e], c_sh[s][e]) > 0.f ? g : 0.f;
          const float xh = (zv[s][e] - c_mu[s][e]) * c_is[s][e];
          g = c_a[s][e] * (g - c_b[s][e] - xh * c_cc[s][e]);
        }
        dzv[s][e] = live ? g : 0.f;
      }
    // ---- input gradient: rows of dz as B operands (k = output channels)
    if (gx) {
      ft4 acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[mt] = ft4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[mt][s][e], dzv[s][e], acc[mt], 0, 0, 0);
      }
      if (live) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          ft4* dst = reinterpret_cast<ft4*>(gx + (long long)(row0 + r) * CI + 16 * mt + 4 * q);
          *dst = accumulate ? *dst + acc[mt] : acc[mt];
        }
      }
    }
    // ---- weight gradient: the tile transposed through the wave's LDS patch (k = rows)
    if (part) {
#pragma unroll
      for (int s = 0; s < KS; ++s) *reinterpret_cast<ft4*>(dzs + r * LD + 16 * s + 4 * q) = dzv[s];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        float da[NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) da[a] = dzs[(4 * s4 + q) * LD + 16 * a + r];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b) accw[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[a], xb[s4][b], accw[a][b], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  // rows past the live count: zero gradient
  if (gx && !accumulate) {
    const long long e0 = (long long)n * CI, e1 = (long long)rows * CI;
    for (long long e = e0 + ((long long)blockIdx.x * RW_THREADS + threadIdx.x) * 4; e < e1; e += (long long)gridDim.x * RW_THREADS * 4)
      *reinterpret_cast<ft4*>(gx + e) = ft4{0.f, 0.f, 0.f, 0.f};
  }
  if (!part) return;
  // ---- the block's partial: waves 1 .. 3 hand theirs to wave 0 through LDS, one after the other (fixed order)
  for (int wv_ = 1; wv_ < RW_WAVES; ++wv_) {
    __syncthreads();
    if (wave == wv_) {
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) s_acc[((a * NB + b) * 4 + i) * 64 + lane] = accw[a][b][i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) accw[a][b][i] += s_acc[((a * NB + b) * 4 + i) * 64 + lane];
    }
  }
  if (wave == 0) {
    float* dst = part + (size_t)blockIdx.x * (CO * CI);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[((a * NB + b) * 4 + i) * 64 + lane] = accw[a][b][i];
  }
}

// dW[co][ci] = sum over the blocks' partials: a block owns 16 elements, thread (element, group g of 16) adds partials g, g + 16, ..
// and the groups are combined in a fixed order (bitwise reproducible).  64 .. 256 blocks: the 1 .. 4 MB of partials are read by
// the whole chip (sixteen blocks of 64 elements with 64 loads in series per thread took 21 us).
#define RW_RED_EL 16
__global__ __launch_bounds__(256) void k_rows_wgrad_reduce(const float* __restrict__ part, int nparts, int CI, int CO,
                                                           float* __restrict__ gw) {
  __shared__ float s_red[16][RW_RED_EL];
  const int el = threadIdx.x & (RW_RED_EL - 1), g = threadIdx.x / RW_RED_EL;
  const int e = blockIdx.x * RW_RED_EL + el, total = CI * CO;
  float s = 0.f;
  if (e < total) {
    const float* p = part + e;
    int k = g;
#pragma unroll 1
    for (; k + 48 < nparts; k += 64) {
      const float v0 = p[(size_t)k * total], v1 = p[(size_t)(k + 16) * total], v2 = p[(size_t)(k + 32) * total],
                  v3 = p[(size_t)(k + 48) * total];
      s = (((s + v0) + v1) + v2) + v3;
    }
    for (; k < nparts; k += 16) s += p[(size_t)k * total];
  }
  s_red[g][el] = s;
  __syncthreads();
  if (g == 0 && e < total) {
    float v = s_red[0][el];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += s_red[k][el];
    const int NB = CI / 16;
    const int lane = e & 63, i = (e >> 6) & 3, ab = e >> 8;
    const int a = ab / NB, b = ab % NB;
    gw[(16 * a + 4 * (lane >> 4) + i) * CI + 16 * b + (lane & 15)] = v;
  }
}

// ------------------------------------------------------------------------------------------------ host
// Cout = 128: two column halves of 64 (forward: blockIdx.y; backward: two launches, the second adding its part of dX)
static bool rows_dims_ok(int Cin, int Cout) {
  return (Cin == 16 || Cin == 32 || Cin == 64) && (Cout == 16 || Cout == 32 || Cout == 64 || Cout == 128);
}
static int rows_cols_per_launch(int Cout) { return Cout > 64 ? 64 : Cout; }

extern "C" int glx_rows_linear_supported(int Cin, int Cout) { return rows_dims_ok(Cin, Cout) ? 1 : 0; }

extern "C" size_t glx_rows_linear_workspace_bytes(int Cin, int Cout) {
  return glx_align((size_t)RW_BWD_BLOCKS * (Cin > 0 ? Cin : 1) * (Cout > 0 ? rows_cols_per_launch(Cout) : 1) * sizeof(float));
}

static int rows_fwd_blocks(int rows) {
  const int want = glx_divup(glx_divup(rows, 16), RW_WAVES);
  return want < 1 ? 1 : (want > RW_MAX_BLOCKS ? RW_MAX_BLOCKS : want);
}

template <int K, int N>
static void rows_fwd_launch(const float* x, const float* w, float* z, int rows, const int32_t* n_live, BnState* st,
                            const BnFinalize& f, int Cout, hipStream_t stream) {
  const dim3 grid(rows_fwd_blocks(rows), Cout / N);
  if (st) hipLaunchKernelGGL((k_rows_linear<K, N, true>), grid, dim3(RW_THREADS), 0, stream, x, w, z, rows, (const int*)n_live, st, f, Cout);
  else hipLaunchKernelGGL((k_rows_linear<K, N, false>), grid, dim3(RW_THREADS), 0, stream, x, w, z, rows, (const int*)n_live, st, f, Cout);
}

#define RW_DISPATCH(CI_, CO_, CALL)                                                           \
  switch ((CI_) * 100 + (CO_)) {                                                              \
    case 1616: CALL(16, 16); break; case 1632: CALL(16, 32); break; case 1664: CALL(16, 64); break; \
    case 3216: CALL(32, 16); break; case 3232: CALL(32, 32); break; case 3264: CALL(32, 64); break; \
    case 6416: CALL(64, 16); break; case 6432: CALL(64, 32); break; case 6464: CALL(64, 64); break; \
    default: break;                                                                           \
  }

// z (rows, Cout) = x (rows, Cin) @ w^T (w: (Cout, Cin) row-major) on the first min(rows, *n_live) rows (the rest of z is left
// untouched and never read by the entry points below).  bn_state != NULL: the training-mode BatchNorm statistics of z in the
// same launch -- coef (scale, shift: 2 Cout floats, what glx_bn_apply_forward takes), save_mean, save_invstd, the running
// statistics (NULL: not tracked); gamma / beta NULL = 1 / 0.  bn_state: glx_bn_state_bytes() zero-filled once, shared with the
// other statistics kernels of the stream.  Replaces Conv1d(k = 1, bias = False) + the statistics half of BatchNorm1d
// (voxel_pool_modules.py:70-130).
extern "C" int glx_rows_linear_bn_forward(const float* x, int rows, int Cin, const float* w, int Cout, const int32_t* n_live,
                                          float* z, const float* gamma, const float* beta, float eps, float momentum,
                                          float* running_mean, float* running_var, float* coef, float* save_mean,
                                          float* save_invstd, void* bn_state, void* stream) {
  GLX_REQUIRE(rows_dims_ok(Cin, Cout), "glx_rows_linear_bn_forward: channels %d -> %d (16 / 32 / 64 in, 16 / 32 / 64 / 128 out)", Cin, Cout);
  GLX_REQUIRE(w && (rows == 0 || (x && z)), "glx_rows_linear_bn_forward: null pointer");
  GLX_REQUIRE(!bn_state || (coef && save_mean && save_invstd), "glx_rows_linear_bn_forward: statistics without their outputs");
  if (rows <= 0) return GLX_OK;
  BnFinalize f{gamma, beta, eps, momentum, coef, save_mean, save_invstd, running_mean, running_var, nullptr, nullptr, nullptr, 0};
#define RW_FWD(CI_, CO_) rows_fwd_launch<CI_, CO_>(x, w, z, rows, n_live, (BnState*)bn_state, f, Cout, (hipStream_t)stream)
  RW_DISPATCH(Cin, rows_cols_per_launch(Cout), RW_FWD)
#undef RW_FWD
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

template <int CI, int CO>
static void rows_bwd_launch(const float* x, const float* z, const float* dy, const float* w, int rows, const int32_t* n_live,
                            const RowsBwdBn& bn, float* gx, float* part, int blocks, int Cout, int coff, hipStream_t stream) {
  if (bn.coef3)
    hipLaunchKernelGGL((k_rows_linear_bwd<CI, CO, true>), dim3(blocks), dim3(RW_THREADS), 0, stream, x, z, dy, w, rows,
                       (const int*)n_live, bn, gx, part, Cout, coff, coff > 0 ? 1 : 0);
  else
    hipLaunchKernelGGL((k_rows_linear_bwd<CI, CO, false>), dim3(blocks), dim3(RW_THREADS), 0, stream, x, z, dy, w, rows,
                       (const int*)n_live, bn, gx, part, Cout, coff, coff > 0 ? 1 : 0);
}

// The backward of glx_rows_linear_bn_forward (+ glx_bn_apply_forward): dy = the gradient of the TRANSFORMED output; coef3 (from
// glx_bn_backward_sums) != NULL: the BatchNorm (+ ReLU, mask re-derived from z and coef_fwd) backward is applied to dy on load;
// coef3 == NULL: dy is the gradient of z itself (plain x @ w^T).  gx (rows, Cin): rows past the live count are zeroed; NULL =
// not wanted.  gw (Cout, Cin); NULL = not wanted.  workspace: glx_rows_linear_workspace_bytes(Cin, Cout).
extern "C" int glx_rows_linear_bn_backward(const float* x, const float* z, const float* dy, int rows, int Cin, const float* w,
                                           int Cout, const int32_t* n_live, const float* coef_fwd, int relu, const float* coef3,
                                           const float* mean, const float* invstd, float* gx, float* gw, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(rows_dims_ok(Cin, Cout), "glx_rows_linear_bn_backward: channels %d -> %d (16 / 32 / 64 in, 16 / 32 / 64 / 128 out)", Cin, Cout);
  GLX_REQUIRE(w && (rows == 0 || (x && dy)), "glx_rows_linear_bn_backward: null pointer");
  GLX_REQUIRE(!coef3 || (z && coef_fwd && mean && invstd), "glx_rows_linear_bn_backward: BatchNorm backward without z / coefficients");
  GLX_REQUIRE(!gw || (workspace && workspace_bytes >= glx_rows_linear_workspace_bytes(Cin, Cout)),
              "glx_rows_linear_bn_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 0) {                       // no rows: a zero weight gradient (the sum of no partials)
    if (gw) {
      hipLaunchKernelGGL(k_rows_wgrad_reduce, dim3(glx_divup(Cin * Cout, RW_RED_EL)), dim3(256), 0, st, (const float*)workspace, 0, Cin,
                         Cout, gw);
      GLX_LAUNCH_CHECK();
    }
    return GLX_OK;
  }
  const int want = glx_divup(glx_divup(rows, 16), RW_WAVES);
  const int blocks = want < 1 ? 1 : (want > RW_BWD_BLOCKS ? RW_BWD_BLOCKS : want);
  RowsBwdBn bn{coef_fwd, coef3, mean, invstd, relu};
  float* part = gw ? (float*)workspace : nullptr;
  const int cols = rows_cols_per_launch(Cout);
  for (int coff = 0; coff < Cout; coff += cols) {          // (the halves share the workspace: same stream, one after the other)
#define RW_BWD(CI_, CO_) rows_bwd_launch<CI_, CO_>(x, z, dy, w, rows, n_live, bn, gx, part, blocks, Cout, coff, st)
    RW_DISPATCH(Cin, cols, RW_BWD)
#undef RW_BWD
    GLX_LAUNCH_CHECK();
    if (gw) {
      hipLaunchKernelGGL(k_rows_wgrad_reduce, dim3(glx_divup(Cin * cols, RW_RED_EL)), dim3(256), 0, st, (const float*)part, blocks, Cin,
                         cols, gw + (size_t)coff * Cin);
      GLX_LAUNCH_CHECK();
    }
  }
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ several small dW = dY^T X
// The RoI head's FC towers (voxelrcnn_head.py:40-66: five 256 x 256 Linear layers behind the first) leave five weight gradients
// dW_l (256, 256) = dZ_l^T (256, R) H_(l-1) (R, 256) with R = 512 RoI rows: 67 MFLOP each.  As library products each was a batched
// split-K GEMM + a sum (ten launches, 17 - 110 us apiece beside the BEV backward); here ONE launch: blockIdx.y = the layer, a block
// owns a 64 x 64 tile of dW, its four waves take every fourth k-step (k = the R rows, four per step) and meet in LDS (fixed order:
// bitwise reproducible).  Operands straight from L2 as 16-byte loads: lane (j, q) reads channels 4 j .. 4 j + 3 of row 4 t + q of
// both matrices, which are the A operands of FOUR interleaved 16-channel tiles {4 m + e} and the B operands of four {4 n + e'}:
// sixteen MFMAs per pair of loads (a lane = output channel form with 4-byte loads needs twelve loads for eight).
#define RW_MJ_MAX 8
struct RowsWgradJobs {
  const float* x[RW_MJ_MAX];
  const float* gy[RW_MJ_MAX];
  float* out[RW_MJ_MAX];
};

#define RW_MJ_U 4          // k-steps per trip (their loads issued together)
__global__ __launch_bounds__(256) void k_rows_wgrad_multi(RowsWgradJobs jobs, int rows, int cin, int cout) {
  __shared__ float s_acc[16 * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
  const int tn = cin / 64, tile = blockIdx.x;
  const int co0 = (tile / tn) * 64, ci0 = (tile % tn) * 64;
  const float* __restrict__ gp = jobs.gy[blockIdx.y] + co0 + 4 * j;
  const float* __restrict__ xp = jobs.x[blockIdx.y] + ci0 + 4 * j;
  float* __restrict__ out = jobs.out[blockIdx.y];
  ft4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[e][f] = ft4{0.f, 0.f, 0.f, 0.f};
  const int nsteps = (rows + 3) >> 2;
  const ft4 zero = ft4{0.f, 0.f, 0.f, 0.f};
  ft4 a[RW_MJ_U], b[RW_MJ_U];
#define RW_MJ_LOAD(A, B, T0)                                                                     \
  _Pragma("unroll") for (int u_ = 0; u_ < RW_MJ_U; ++u_) {                                       \
    const int row_ = 4 * ((T0) + 4 * u_) + q;                                                    \
    const bool ok_ = row_ < rows;                                                                \
    const long long rc_ = ok_ ? row_ : rows - 1;                                                 \
    const ft4 av_ = *reinterpret_cast<const ft4*>(gp + rc_ * cout);                              \
    const ft4 bv_ = *reinterpret_cast<const ft4*>(xp + rc_ * cin);                               \
    A[u_] = ok_ ? av_ : zero;                                                                    \
    B[u_] = ok_ ? bv_ : zero;                                                                    \
  }
  int t = wave;                                   // this wave's k-steps: t, t + 4, ..
  if (t < nsteps) { RW_MJ_LOAD(a, b, t); }
  while (t < nsteps) {
    const int tnext = t + 4 * RW_MJ_U;
    ft4 an[RW_MJ_U], bn[RW_MJ_U];
    if (tnext < nsteps) { RW_MJ_LOAD(an, bn, tnext); }
#pragma unroll
    for (int u = 0; u < RW_MJ_U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][f], acc[e][f], 0, 0, 0);
    if (tnext < nsteps) {
#pragma unroll
      for (int u = 0; u < RW_MJ_U; ++u) { a[u] = an[u]; b[u] = bn[u]; }
    }
    t = tnext;
  }
#undef RW_MJ_LOAD
  // rows past the matrix were loaded as zeros (a step past nsteps inside a trip is all zeros too: row_ >= rows)
  for (int w = 1; w < RW_WAVES; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int i = 0; i < 4; ++i) s_acc[((e * 4 + f) * 4 + i) * 64 + lane] = acc[e][f][i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[e][f][i] += s_acc[((e * 4 + f) * 4 + i) * 64 + lane];
    }
  }
  if (wave == 0) {
    // acc[e][f][i] = dW[co0 + 4 (4 q + i) + e][ci0 + 4 j + f]
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ft4 v;
#pragma unroll
        for (int f = 0; f < 4; ++f) v[f] = acc[e][f][i];
        *reinterpret_cast<ft4*>(out + (long long)(co0 + 4 * (4 * q + i) + e) * cin + ci0 + 4 * j) = v;
      }
  }
}

// out_j (cout, cin) = gy_j (rows, cout)^T @ x_j (rows, cin) for j < njobs <= 8 problems of ONE shape, one launch.  cout % 64 == 0,
// cin % 64 == 0; everything row-major and dense; rows >= 1.  Replaces the weight gradient autograd derives for nn.Linear
// (voxelrcnn_head.py:40-66's towers) as library GEMMs.
extern "C" int glx_linear_wgrad_multi(int njobs, const float* const* x, const float* const* gy, float* const* out, int rows,
                                      int cin, int cout, void* stream) {
  GLX_REQUIRE(njobs >= 1 && njobs <= RW_MJ_MAX, "glx_linear_wgrad_multi: %d problems (1 .. %d)", njobs, RW_MJ_MAX);
  GLX_REQUIRE(x && gy && out, "glx_linear_wgrad_multi: null pointer");
  GLX_REQUIRE(rows >= 1 && cin > 0 && cout > 0 && cin % 64 == 0 && cout % 64 == 0,
              "glx_linear_wgrad_multi: rows %d, (%d, %d) filters: needs cout %% 64 == 0 and cin %% 64 == 0", rows, cout, cin);
  RowsWgradJobs jobs;
  for (int j = 0; j < RW_MJ_MAX; ++j) {
    const int s = j < njobs ? j : 0;
    GLX_REQUIRE(x[s] && gy[s] && out[s], "glx_linear_wgrad_multi: null pointer in problem %d", s);
    jobs.x[j] = x[s]; jobs.gy[j] = gy[s]; jobs.out[j] = out[s];
  }
  hipLaunchKernelGGL(k_rows_wgrad_multi, dim3((cout / 64) * (cin / 64), njobs), dim3(256), 0, (hipStream_t)stream, jobs, rows, cin,
                     cout);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ the wide first Linear
// VoxelRCNNHead's first shared layer (voxelrcnn_head.py:40-52: Linear(6^3 * 96 = 20 736, 256, bias=False)) on R = 512 RoI rows:
// 5.4 GFLOP per direction, the last library GEMMs of the training step until round 6.  Exact fp32 products on v_mfma_f32_16x16x4_f32
// (that instruction takes ONE scalar per lane and operand: any operand layout feeds it, no transposes), operands straight from L2
// as 16-byte loads, fixed summation order.
//   forward   y (R, N)  = x (R, K) w (N, K)^T     split along K over gridDim.y slabs -> partial tiles, summed by k_lin_reduce
//   input gr. gx (R, K) = dz (R, N) w (N, K)      a wave owns 16 rows x 64 columns, the contraction (N = 256) runs in the wave
//   weight gr. dW (N, K) = dz^T x                 glx_linear_wgrad_multi above (one job)
// Lane l = (i = l & 15, q = l >> 4).  A 16-byte load along the contraction index gives a lane the scalars of FOUR k-steps
// (k = k0 + 4 q + s, s = 0..3): both operands of the forward are read that way (rows of x, rows of w), and dz in the input
// gradient; w in the input gradient is read along its rows' K (four consecutive output columns = four interleaved column
// tiles), one load per k-step.
#define LIN_TRIPS 2        // 16-k trips whose loads are issued together (forward)
__global__ __launch_bounds__(256) void k_lin_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ part,
                                                 int R, int N, int K, int kslab) {
  __shared__ float s_acc[16 * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int tn = N / 64, tile = blockIdx.x;
  const int r0 = (tile / tn) * 64, n0 = (tile % tn) * 64;
  const int k_begin = blockIdx.y * kslab, k_end = min(K, k_begin + kslab);
  ft4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[e][f] = ft4{0.f, 0.f, 0.f, 0.f};
  const ft4 zero = ft4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ xp[4];
  const float* __restrict__ wp[4];
  bool xok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = r0 + 16 * e + i;
    xok[e] = row < R;
    xp[e] = x + (long long)(xok[e] ? row : R - 1) * K + 4 * q;
    wp[e] = w + (long long)(n0 + 16 * e + i) * K + 4 * q;
  }
  // this wave's trips: k_begin + 16 (wave + 4 t), a trip = 16 consecutive k (four MFMA k-steps)
  for (int k = k_begin + 16 * wave * LIN_TRIPS; k < k_end; k += 16 * 4 * LIN_TRIPS) {
    ft4 a[LIN_TRIPS][4], b[LIN_TRIPS][4];
#pragma unroll
    for (int u = 0; u < LIN_TRIPS; ++u) {
      const int kk = k + 16 * u;
      const bool kok = kk + 4 * q + 3 < k_end;            // K and the slabs are multiples of 16: a trip is whole or absent
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const ft4 av = *reinterpret_cast<const ft4*>(xp[e] + (kok ? kk : k_begin));
        const ft4 bv = *reinterpret_cast<const ft4*>(wp[e] + (kok ? kk : k_begin));
        a[u][e] = (kok && xok[e]) ? av : zero;
        b[u][e] = kok ? bv : zero;
      }
    }
#pragma unroll
    for (int u = 0; u < LIN_TRIPS; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int f = 0; f < 4; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e][s], b[u][f][s], acc[e][f], 0, 0, 0);
  }
  // the four waves' sums meet in LDS in wave order (bitwise reproducible)
  for (int wv = 1; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int t = 0; t < 4; ++t) s_acc[((e * 4 + f) * 4 + t) * 64 + lane] = acc[e][f][t];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[e][f][t] += s_acc[((e * 4 + f) * 4 + t) * 64 + lane];
    }
  }
  if (wave == 0) {
    // acc[e][f][t] = y[r0 + 16 e + 4 q + t][n0 + 16 f + i] (partial over this slab)
    float* __restrict__ dst = part + (long long)blockIdx.y * R * N;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = r0 + 16 * e + 4 * q + t;
        if (row < R) {
#pragma unroll
          for (int f = 0; f < 4; ++f) dst[(long long)row * N + n0 + 16 * f + i] = acc[e][f][t];
        }
      }
  }
}

__global__ __launch_bounds__(256) void k_lin_reduce(const float* __restrict__ part, int slabs, long long n, float* __restrict__ y) {
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= n) return;
  ft4 s = *reinterpret_cast<const ft4*>(part + e);
  for (int k = 1; k < slabs; ++k) s += *reinterpret_cast<const ft4*>(part + (long long)k * n + e);
  *reinterpret_cast<ft4*>(y + e) = s;
}

__global__ __launch_bounds__(256) void k_lin_dgrad(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ gx,
                                                   int R, int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int tk = K / 64, tile = blockIdx.x;
  const int r0 = (tile / tk) * 64 + 16 * wave, c0 = (tile % tk) * 64;
  const int row = r0 + i;
  const bool rok = row < R;
  const float* __restrict__ ap = dz + (long long)(rok ? row : R - 1) * N + 4 * q;      // dz[row][n0 + 4 q + s]
  const float* __restrict__ bp = w + c0 + 4 * i;                                       // w[n][c0 + 4 i .. + 3]: column tiles {4 i + f}
  ft4 acc[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) acc[f] = ft4{0.f, 0.f, 0.f, 0.f};
  const ft4 zero = ft4{0.f, 0.f, 0.f, 0.f};
  for (int n0 = 0; n0 < N; n0 += 32) {                       // two trips of 16 n per pass
    ft4 a[2], b[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const ft4 av = *reinterpret_cast<const ft4*>(ap + n0 + 16 * u);
      a[u] = rok ? av : zero;
#pragma unroll
      for (int s = 0; s < 4; ++s) b[u][s] = *reinterpret_cast<const ft4*>(bp + (long long)(n0 + 16 * u + 4 * q + s) * K);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], b[u][s][f], acc[f], 0, 0, 0);
  }
  // acc[f][t] = gx[r0 + 4 q + t][c0 + 4 i + f]: a lane's four column tiles are four consecutive floats
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int orow = r0 + 4 * q + t;
    if (orow < R) *reinterpret_cast<ft4*>(gx + (long long)orow * K + c0 + 4 * i) = ft4{acc[0][t], acc[1][t], acc[2][t], acc[3][t]};
  }
}

extern "C" size_t glx_linear_wide_workspace_bytes(int rows, int N, int K) {
  const int slabs = K >= 4096 ? (K % 27 == 0 && (K / 27) % 16 == 0 ? 27 : 16) : 1;
  return glx_align((size_t)slabs * rows * N * sizeof(float)) + 256;
}

// y (rows, N) = x (rows, K) @ w (N, K)^T.  N % 64 == 0, K % 16 == 0; workspace: glx_linear_wide_workspace_bytes.
extern "C" int glx_linear_wide_forward(const float* x, const float* w, float* y, int rows, int N, int K, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(x && w && y && workspace, "glx_linear_wide_forward: null pointer");
  GLX_REQUIRE(rows >= 1 && N > 0 && K > 0 && N % 64 == 0 && K % 16 == 0, "glx_linear_wide_forward: rows %d, N %d %% 64, K %d %% 16",
              rows, N, K);
  int slabs = K >= 4096 ? (K % 27 == 0 && (K / 27) % 16 == 0 ? 27 : 16) : 1;
  int kslab = glx_divup(glx_divup(K, slabs), 16) * 16;
  slabs = glx_divup(K, kslab);
  GLX_REQUIRE(workspace_bytes >= (size_t)slabs * rows * N * sizeof(float), "glx_linear_wide_forward: workspace %zu bytes", workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_lin_fwd, dim3(glx_divup(rows, 64) * (N / 64), slabs), dim3(256), 0, st, x, w, (float*)workspace, rows, N, K, kslab);
  GLX_LAUNCH_CHECK();
  const long long n = (long long)rows * N;
  hipLaunchKernelGGL(k_lin_reduce, dim3((unsigned)glx_divup(n, 1024)), dim3(256), 0, st, (const float*)workspace, slabs, n, y);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// gx (rows, K) = dz (rows, N) @ w (N, K).  K % 64 == 0, N % 32 == 0.
extern "C" int glx_linear_wide_input_grad(const float* dz, const float* w, float* gx, int rows, int N, int K, void* stream) {
  GLX_REQUIRE(dz && w && gx, "glx_linear_wide_input_grad: null pointer");
  GLX_REQUIRE(rows >= 1 && N > 0 && K > 0 && K % 64 == 0 && N % 32 == 0, "glx_linear_wide_input_grad: rows %d, N %d %% 32, K %d %% 64",
              rows, N, K);
  hipLaunchKernelGGL(k_lin_dgrad, dim3(glx_divup(rows, 64) * (K / 64)), dim3(256), 0, (hipStream_t)stream, dz, w, gx, rows, N, K);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
