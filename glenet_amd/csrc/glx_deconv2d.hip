// glx_deconv2d.hip -- the transposed convolutions of the BEV backbone's upsampling branch (SURVEY 8a row a21;
// pcdet/models/backbones_2d/base_bev_backbone.py:51-66: ConvTranspose2d(c, cu, u, stride=u), u = 1 or 2) on
// channels-last fp32 maps, with the split-bf16 arithmetic of glx_conv2d.hip (fp32 products as six bf16 MFMAs).
//
// With kernel = stride = u a transposed convolution has no overlapping taps: output pixel (u y + dy, u x + dx) is
// x[y][x] times the (Cin x Cout) slice W[:, :, dy, dx].  On the COARSE grid (the input's) that is
//   forward         y[(u y + dy, u x + dx)][co] = sum_ci        x[(y, x)][ci]              W[ci][co][dy][dx]
//   input gradient  gx[(y, x)][ci]              = sum_(dy,dx,co) gy[(u y + dy, u x + dx)][co] W[ci][co][dy][dx]
//   weight gradient dW[ci][co][dy][dx]          = sum_(y,x)     x[(y, x)][ci]              gy[(u y + dy, u x + dx)][co]
// i.e. pixel-row GEMMs whose rows are gathered from / scattered to a strided pixel set.  k_pconv runs the first two:
// 128 consecutive coarse pixels x 64 output channels per block, K = (input tap) x (32-channel chunk); the rows of a step
// are loaded one step ahead (fp32, 128-byte pieces), split into three bf16 planes in LDS (80-byte rows), the weight
// slices stream L2 -> registers -> LDS as in k_conv3x3.  k_pconv_wgrad runs the third like k_conv3x3_wgrad: gy (one
// alignment per tap) straight into registers, the x tile through the transposing LDS read, block partials + a reduce.
#include "glx_common.h"
#include <stdlib.h>
#include "glx_bf16x3.h"
#include "glx_bn_state.h"

#define PC_TM 128                        // coarse pixels per block
#ifndef PC_ROW
#define PC_ROW 80                        // (96-byte rows, conflict-free for the 3x3 kernels' operand map, measured no better here: r05)
#endif
#define PC_APLANE (PC_TM * PC_ROW)       // 10 240
#define PC_BN 64
#define PC_WPLANE (PC_BN * PC_ROW)       // 5 120
#define PC_WBUF (3 * PC_WPLANE)
#define PC_LDS (3 * PC_APLANE + 2 * PC_WBUF)   // 61 440: two blocks per CU

// W (Cin, Cout, u, u) with element strides ->
//   fwd [ntap = (dy, dx)][chunk of Cin][3][Cout][32]             (rows x, K = Cin, columns Cout, one slice per output tap)
//   bwd [ktap = (dy, dx)][chunk of Cout][3][Cin][32]             (rows gy at the tap's pixels, K = Cout per tap, columns Cin)
__global__ void k_pconv_pack(const float* __restrict__ W, long long s_ci, long long s_co, long long s_kh, long long s_kw,
                             int Cin, int Cout, int u, uint16_t* __restrict__ fwd, uint16_t* __restrict__ bwd) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Cin * Cout * u * u) return;
  const int co = e % Cout, ci = (e / Cout) % Cin, tap = e / (Cin * Cout);
  const float w = W[ci * s_ci + co * s_co + (tap / u) * s_kh + (tap % u) * s_kw];
  __bf16 p[3];
  cv_split(w, p[0], p[1], p[2]);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const uint16_t bits = __builtin_bit_cast(uint16_t, p[q]);
    if (fwd) fwd[((((size_t)tap * (Cin / 32) + ci / 32) * 3 + q) * Cout + co) * 32 + (ci & 31)] = bits;
    if (bwd) bwd[((((size_t)tap * (Cout / 32) + co / 32) * 3 + q) * Cin + ci) * 32 + (co & 31)] = bits;
  }
}

struct PconvArgs {
  const float* x;       // rows gathered from here: (B, Hc * ui, Wc * ui, Ck)
  const uint16_t* wp;   // [ntaps][ktaps][Ck / 32][3][N][32]
  float* y;             // rows scattered to here: (B, Hc * uo, Wc * uo, N)
  int B, Hc, Wc;        // the coarse grid
  int Ck, N;            // channels per input tap, output channels
  int ui, uo;           // pixel stride of the input / output map relative to the coarse grid (one of them is 1)
  int ktaps, ntaps;     // ui * ui, uo * uo
  int k3;               // 1: the input taps are a 3 x 3 window with zero padding 1 around (ui y, ui x) (strided convolution)
  int M, mtiles, nblk, nunits;
  const float* epi_scale;   // inference epilogue (glx_epilogue): y = relu?(acc * scale[c] + shift[c])
  const float* epi_shift;
  int epi_relu;
  int ldc, coff;        // floats between output pixels (>= N) and the first output channel's offset inside a pixel
  BnState* bn_state;    // second form, STATS: training-mode BatchNorm statistics of y taken in the epilogue (as k_conv3x3_v2)
  BnFinalize bn;
};

__global__ __launch_bounds__(256, 2) void k_pconv(PconvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sW = smem + 3 * PC_APLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int nch = a.Ck >> 5, nsteps = a.ktaps * nch;
  const size_t wslice = (size_t)a.N * 32;
  const int wdst = (tid >> 2) * PC_ROW + (tid & 3) * 16;
  const int hw = a.Hc * a.Wc;

  for (int unit = blockIdx.x; unit < a.nunits; unit += gridDim.x) {
    // unit -> (pixel tile, output tap, channel block): the taps and channel blocks of a tile run side by side
    int t = unit;
    const int n0 = (t % a.nblk) * PC_BN;
    t /= a.nblk;
    const int ntap = t % a.ntaps;
    const int m0 = (t / a.ntaps) * PC_TM;
    const uint16_t* wsrc = a.wp + ((size_t)ntap * nsteps * 3) * wslice + (size_t)n0 * 32 + tid * 8;

    // the thread's four 16-byte pieces of a step's rows: pixel m0 + (e >> 3), channels 4 (e & 7) ..
    long long abase[4];
    int iy0[4], ix0[4];       // k3: the window's centre in the input map
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, m = m0 + (e >> 3);
      if (m < a.M) {
        const int b = m / hw, rem = m - b * hw, y = rem / a.Wc, x = rem - y * a.Wc;
        abase[i] = (((long long)b * a.Hc * a.ui + (long long)y * a.ui) * (a.Wc * a.ui) + (long long)x * a.ui) * a.Ck + (e & 7) * 4;
        iy0[i] = y * a.ui;
        ix0[i] = x * a.ui;
      } else {
        abase[i] = -1;
        iy0[i] = ix0[i] = 0;
      }
    }
    f32x4 areg[4];
    uint4 wreg0, wreg1, wreg2;
#define PC_LOAD(STEP)                                                                                          \
  {                                                                                                            \
    const int kt_ = (STEP) / nch, ch_ = (STEP) - kt_ * nch;                                                    \
    const int dy_ = a.k3 ? kt_ / 3 - 1 : kt_ / a.ui, dx_ = a.k3 ? kt_ % 3 - 1 : kt_ % a.ui;                    \
    const long long off_ = ((long long)dy_ * (a.Wc * a.ui) + dx_) * a.Ck + ch_ * 32;                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                         \
      const bool ok_ = abase[i_] >= 0 && (!a.k3 || ((unsigned)(iy0[i_] + dy_) < (unsigned)(a.Hc * a.ui) &&     \
                                                    (unsigned)(ix0[i_] + dx_) < (unsigned)(a.Wc * a.ui)));     \
      areg[i_] = ok_ ? *reinterpret_cast<const f32x4*>(a.x + abase[i_] + off_) : f32x4{0.f, 0.f, 0.f, 0.f};    \
    }                                                                                                          \
    const uint16_t* s_ = wsrc + (size_t)(STEP) * 3 * wslice;                                                   \
    wreg0 = *reinterpret_cast<const uint4*>(s_);                                                               \
    wreg1 = *reinterpret_cast<const uint4*>(s_ + wslice);                                                      \
    wreg2 = *reinterpret_cast<const uint4*>(s_ + 2 * wslice);                                                  \
  }
#define PC_STORE(BUF)                                                                                          \
  {                                                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                         \
      const int e_ = tid + i_ * 256;                                                                           \
      char* d_ = sA + (e_ >> 3) * PC_ROW + (e_ & 7) * 8;                                                       \
      bf16x4 p0_, p1_, p2_;                                                                                    \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                       \
        __bf16 u_, v_, w_;                                                                                     \
        cv_split(areg[i_][j_], u_, v_, w_);                                                                    \
        p0_[j_] = u_; p1_[j_] = v_; p2_[j_] = w_;                                                              \
      }                                                                                                        \
      *reinterpret_cast<bf16x4*>(d_) = p0_;                                                                    \
      *reinterpret_cast<bf16x4*>(d_ + PC_APLANE) = p1_;                                                        \
      *reinterpret_cast<bf16x4*>(d_ + 2 * PC_APLANE) = p2_;                                                    \
    }                                                                                                          \
    char* w_ = sW + (BUF) * PC_WBUF + wdst;                                                                    \
    *reinterpret_cast<uint4*>(w_) = wreg0;                                                                     \
    *reinterpret_cast<uint4*>(w_ + PC_WPLANE) = wreg1;                                                         \
    *reinterpret_cast<uint4*>(w_ + 2 * PC_WPLANE) = wreg2;                                                     \
  }

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    PC_LOAD(0);
    __syncthreads();                      // the previous unit's reads of the images are done
    PC_STORE(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const int cur = s & 1;
      if (s + 1 < nsteps) { PC_LOAD(s + 1); }
      bf16x8 xa[2][3], wa[4][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          xa[i][q] = *reinterpret_cast<const bf16x8*>(sA + q * PC_APLANE + ((2 * wave + i) * 16 + r) * PC_ROW + kq * 16);
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          wa[n][q] = *reinterpret_cast<const bf16x8*>(sW + cur * PC_WBUF + q * PC_WPLANE + (n * 16 + r) * PC_ROW + kq * 16);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n) BF3_MFMA6(acc[i][n], wa[n], xa[i]);
      if (s + 1 < nsteps) {
        __syncthreads();                  // everyone has read the row image
        PC_STORE(cur ^ 1);
        __syncthreads();
      }
    }
#undef PC_LOAD
#undef PC_STORE
    // lane (r, kq) of accumulator (i, n) = coarse pixel m0 + (2 wave + i) * 16 + r, channels n0 + 16 n + 4 kq ..
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + (2 * wave + i) * 16 + r;
      if (m < a.M) {
        const int b = m / hw, rem = m - b * hw, y = rem / a.Wc, x = rem - y * a.Wc;
        float* dst = a.y + ((((long long)b * a.Hc + y) * a.uo + ntap / a.uo) * (a.Wc * a.uo) + (long long)x * a.uo + ntap % a.uo) * a.ldc +
                     a.coff + n0 + 4 * kq;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          f32x4 v = acc[i][n];
          if (a.epi_scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.epi_scale + n0 + 16 * n + 4 * kq);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(a.epi_shift + n0 + 16 * n + 4 * kq);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float t = __fmaf_rn(v[g], sc[g], sh[g]);
              v[g] = a.epi_relu ? fmaxf(t, 0.f) : t;
            }
          }
          *reinterpret_cast<f32x4*>(dst + 16 * n) = v;
        }
      }
    }
  }
}

// Second form, as k_conv3x3_v2: wave w owns ONE 16-channel tile for all 128 pixels of the block, its weight fragments
// (3 planes x 16 bytes per lane per step) come straight from L2 into the MFMA operands one step ahead -- no weight image
// in LDS; the row image alone (31 KB) lets three blocks share a CU.
#define PC2_LDS (3 * PC_APLANE)

template <bool STATS>
__global__ __launch_bounds__(256, 3) void k_pconv_v2(PconvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int nch = a.Ck >> 5, nsteps = a.ktaps * nch;
  const size_t wslice = (size_t)a.N * 32;
  const int hw = a.Hc * a.Wc;
  // STATS: the grid is a multiple of nblk, so every unit of this block has the same channel block
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  const int stats_n0 = ((int)blockIdx.x % a.nblk) * PC_BN;

  for (int unit = blockIdx.x; unit < a.nunits; unit += gridDim.x) {
    int t = unit;
    const int n0 = (t % a.nblk) * PC_BN;
    t /= a.nblk;
    const int ntap = t % a.ntaps;
    const int m0 = (t / a.ntaps) * PC_TM;
    const uint16_t* wsrc = a.wp + ((size_t)ntap * nsteps * 3) * wslice + (size_t)(n0 + 16 * wave + r) * 32 + kq * 8;

    long long abase[4];
    int iy0[4], ix0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, m = m0 + (e >> 3);
      if (m < a.M) {
        const int b = m / hw, rem = m - b * hw, y = rem / a.Wc, x = rem - y * a.Wc;
        abase[i] = (((long long)b * a.Hc * a.ui + (long long)y * a.ui) * (a.Wc * a.ui) + (long long)x * a.ui) * a.Ck + (e & 7) * 4;
        iy0[i] = y * a.ui;
        ix0[i] = x * a.ui;
      } else {
        abase[i] = -1;
        iy0[i] = ix0[i] = 0;
      }
    }
    f32x4 areg[4];
    bf16x8 wcur[3], wnxt[3];
#define PC2_LOAD(STEP)                                                                                         \
  {                                                                                                            \
    const int kt_ = (STEP) / nch, ch_ = (STEP) - kt_ * nch;                                                    \
    const int dy_ = a.k3 ? kt_ / 3 - 1 : kt_ / a.ui, dx_ = a.k3 ? kt_ % 3 - 1 : kt_ % a.ui;                    \
    const long long off_ = ((long long)dy_ * (a.Wc * a.ui) + dx_) * a.Ck + ch_ * 32;                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                         \
      const bool ok_ = abase[i_] >= 0 && (!a.k3 || ((unsigned)(iy0[i_] + dy_) < (unsigned)(a.Hc * a.ui) &&     \
                                                    (unsigned)(ix0[i_] + dx_) < (unsigned)(a.Wc * a.ui)));     \
      areg[i_] = ok_ ? *reinterpret_cast<const f32x4*>(a.x + abase[i_] + off_) : f32x4{0.f, 0.f, 0.f, 0.f};    \
    }                                                                                                          \
    const uint16_t* s_ = wsrc + (size_t)(STEP) * 3 * wslice;                                                   \
    wnxt[0] = *reinterpret_cast<const bf16x8*>(s_);                                                            \
    wnxt[1] = *reinterpret_cast<const bf16x8*>(s_ + wslice);                                                   \
    wnxt[2] = *reinterpret_cast<const bf16x8*>(s_ + 2 * wslice);                                               \
  }
#define PC2_STORE()                                                                                            \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                           \
    const int e_ = tid + i_ * 256;                                                                             \
    char* d_ = sA + (e_ >> 3) * PC_ROW + (e_ & 7) * 8;                                                         \
    bf16x4 p0_, p1_, p2_;                                                                                      \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                         \
      __bf16 u_, v_, w_;                                                                                       \
      cv_split(areg[i_][j_], u_, v_, w_);                                                                      \
      p0_[j_] = u_; p1_[j_] = v_; p2_[j_] = w_;                                                                \
    }                                                                                                          \
    *reinterpret_cast<bf16x4*>(d_) = p0_;                                                                      \
    *reinterpret_cast<bf16x4*>(d_ + PC_APLANE) = p1_;                                                          \
    *reinterpret_cast<bf16x4*>(d_ + 2 * PC_APLANE) = p2_;                                                      \
  }

    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    PC2_LOAD(0);
    for (int s = 0; s < nsteps; ++s) {
      __syncthreads();                  // everyone has read the previous row image
      PC2_STORE();
#pragma unroll
      for (int q = 0; q < 3; ++q) wcur[q] = wnxt[q];
      __syncthreads();
      if (s + 1 < nsteps) { PC2_LOAD(s + 1); }
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        bf16x8 xa[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q)
            xa[i][q] = *reinterpret_cast<const bf16x8*>(sA + q * PC_APLANE + ((2 * part + i) * 16 + r) * PC_ROW + kq * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) BF3_MFMA6(acc[2 * part + i], wcur, xa[i]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef PC2_LOAD
#undef PC2_STORE
    // accumulator i = coarse pixel m0 + 16 i + r; channels n0 + 16 wave + 4 kq ..
    const int ch0 = n0 + 16 * wave + 4 * kq;
    f32x4 sc = f32x4{1.f, 1.f, 1.f, 1.f}, sh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.epi_scale) {
      sc = *reinterpret_cast<const f32x4*>(a.epi_scale + ch0);
      sh = *reinterpret_cast<const f32x4*>(a.epi_shift + ch0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + 16 * i + r;
      if (m < a.M) {
        const int b = m / hw, rem = m - b * hw, y = rem / a.Wc, x = rem - y * a.Wc;
        f32x4 v = acc[i];
        if (a.epi_scale) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float tt = __fmaf_rn(v[g], sc[g], sh[g]);
            v[g] = a.epi_relu ? fmaxf(tt, 0.f) : tt;
          }
        }
        *reinterpret_cast<f32x4*>(a.y + ((((long long)b * a.Hc + y) * a.uo + ntap / a.uo) * (a.Wc * a.uo) + (long long)x * a.uo +
                                          ntap % a.uo) * a.ldc + a.coff + ch0) = v;
        if (STATS) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            ssum[g] += v[g];
            ssq[g] += v[g] * v[g];
          }
        }
      }
    }
  }
  if (STATS) {                                            // as k_conv3x3_v2: block sums -> accumulator set -> ticket -> finalize
    double* red = reinterpret_cast<double*>(smem);        // [moment][64 channels]
    __shared__ int s_last;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double d0 = (double)ssum[g], d1 = (double)ssq[g];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        d0 += __shfl_xor(d0, m);
        d1 += __shfl_xor(d1, m);
      }
      if (r == 0) {
        red[16 * wave + 4 * kq + g] = d0;
        red[64 + 16 * wave + 4 * kq + g] = d1;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int ch = tid & 63, mom = tid >> 6;
      double seen = unsafeAtomicAdd(a.bn_state->acc[blockIdx.x % BN_SETS] + mom * BN_MAXC + stats_n0 + ch, red[mom * 64 + ch]);
      asm volatile("" ::"v"(seen) : "memory");
    }
    __syncthreads();
    if (tid == 0)
      s_last = __hip_atomic_fetch_add(&a.bn_state->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (s_last) bn_finalize_sets<false, 256>(a.bn_state, a.bn, a.N, a.M * a.ntaps, reinterpret_cast<double(*)[2]>(smem));
  }
}

extern "C" size_t glx_deconv_packed_bytes(int Cin, int Cout, int u) {
  return glx_align((size_t)u * u * 3 * Cin * Cout * sizeof(uint16_t));
}

extern "C" int glx_deconv_pack(const float* W, long long s_ci, long long s_co, long long s_kh, long long s_kw, int Cin,
                               int Cout, int u, void* fwd, void* bwd, void* stream) {
  GLX_REQUIRE((u == 1 || u == 2) && Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0,
              "glx_deconv_pack: kernel = stride in {1, 2}, channels multiples of 64 (got u=%d, %d -> %d)", u, Cin, Cout);
  hipLaunchKernelGGL(k_pconv_pack, dim3(glx_divup(Cin * Cout * u * u, 256)), dim3(256), 0, (hipStream_t)stream, W, s_ci, s_co,
                     s_kh, s_kw, Cin, Cout, u, (uint16_t*)fwd, (uint16_t*)bwd);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// Epilogue and output placement of the NEXT glx_deconv_forward / glx_conv3x3s2_forward call of this host thread:
// y = relu?(acc * scale[c] + shift[c]) (a folded eval-mode BatchNorm), written with `ldc` floats between pixels starting at
// channel `coff` of each pixel -- a deblock's result straight into its slice of the concatenated map
// (base_bev_backbone.py:100-104).  ldc = 0: dense (ldc = Cout, coff = 0); scale = NULL: no transform.
// (glx_epilogue, an explicit argument of glx_deconv_forward_ex / glx_conv3x3s2_forward_ex)
static int pconv_launch(const float* x, const void* packed, float* y, int B, int Hc, int Wc, int Ck, int N, int ui, int uo,
                        hipStream_t st, int k3 = 0, const glx_epilogue* epi = nullptr, const glx_bn_stats* bnp = nullptr) {
  GLX_REQUIRE(!bnp || (!epi && bnp->state && bnp->coef && bnp->save_mean && bnp->save_invstd && N <= BN_MAXC),
              "glx_pconv: BatchNorm statistics: null pointer, more than %d channels, or combined with an epilogue", BN_MAXC);
  GLX_REQUIRE(!epi || ((epi->scale == nullptr) == (epi->shift == nullptr) && epi->ldc >= 0 && epi->coff >= 0 &&
                       (epi->ldc & 3) == 0 && (epi->coff & 3) == 0),
              "glx_pconv: bad epilogue (ldc %d, coff %d)", epi ? epi->ldc : 0, epi ? epi->coff : 0);
  static const int form = 2;
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_pconv, hipFuncAttributeMaxDynamicSharedMemorySize, PC_LDS));
    attr_set = true;
  }
  PconvArgs a;
  a.x = x; a.wp = (const uint16_t*)packed; a.y = y;
  a.B = B; a.Hc = Hc; a.Wc = Wc; a.Ck = Ck; a.N = N; a.ui = ui; a.uo = uo;
  a.ktaps = k3 ? 9 : ui * ui; a.ntaps = uo * uo;
  a.k3 = k3;
  a.epi_scale = a.epi_shift = nullptr;
  a.epi_relu = 0; a.ldc = N; a.coff = 0;
  a.bn_state = bnp ? (BnState*)bnp->state : nullptr;
  a.bn = BnFinalize{};
  if (bnp)
    a.bn = BnFinalize{bnp->gamma, bnp->beta, bnp->eps, bnp->momentum, bnp->coef, bnp->save_mean, bnp->save_invstd,
                      bnp->running_mean, bnp->running_var, nullptr, nullptr, nullptr};
  if (epi) {
    a.epi_scale = epi->scale; a.epi_shift = epi->shift; a.epi_relu = epi->relu;
    if (epi->ldc) { a.ldc = epi->ldc; a.coff = epi->coff; }
    GLX_REQUIRE(a.ldc >= a.coff + N, "glx_pconv: output slice [%d, %d) does not fit a pixel of %d floats", a.coff, a.coff + N, a.ldc);
  }
  a.M = B * Hc * Wc;
  a.mtiles = glx_divup(a.M, PC_TM);
  a.nblk = N / PC_BN;
  a.nunits = a.mtiles * a.ntaps * a.nblk;
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 0;
    GLX_HIP(hipGetDevice(&dev));
    GLX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    slots = 2 * (cus > 0 ? cus : 256);
  }
  if (form == 2 || bnp) {
    const int resident = slots / 2 * 3;
    int grid = a.nunits < resident ? a.nunits : resident;
    if (bnp) {
      grid = grid / a.nblk * a.nblk;                      // every block keeps one channel block (nunits is a multiple of nblk)
      hipLaunchKernelGGL(k_pconv_v2<true>, dim3(grid), dim3(256), PC2_LDS, st, a);
    } else {
      hipLaunchKernelGGL(k_pconv_v2<false>, dim3(grid), dim3(256), PC2_LDS, st, a);
    }
  } else {
    hipLaunchKernelGGL(k_pconv, dim3(a.nunits < slots ? a.nunits : slots), dim3(256), PC_LDS, st, a);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_deconv_forward_ex(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout, int u,
                                     float* y, const glx_epilogue* epilogue, void* stream) {
  GLX_REQUIRE(B > 0 && H > 0 && W > 0 && (u == 1 || u == 2) && Cin % 64 == 0 && Cout % 64 == 0,
              "glx_deconv_forward: bad sizes (%d, %d, %d), u=%d, %d -> %d", B, H, W, u, Cin, Cout);
  return pconv_launch(x, packed_fwd, y, B, H, W, Cin, Cout, 1, u, (hipStream_t)stream, 0, epilogue);
}
extern "C" int glx_deconv_forward(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout, int u,
                                  float* y, void* stream) {
  return glx_deconv_forward_ex(x, B, H, W, Cin, packed_fwd, Cout, u, y, nullptr, stream);
}
// ... with the training-mode BatchNorm statistics of y taken in the epilogue (contract of glx_bn_stats / glx_conv_opts.bn)
extern "C" int glx_deconv_forward_bn(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout, int u,
                                     float* y, const glx_bn_stats* bn, void* stream) {
  GLX_REQUIRE(B > 0 && H > 0 && W > 0 && (u == 1 || u == 2) && Cin % 64 == 0 && Cout % 64 == 0,
              "glx_deconv_forward_bn: bad sizes (%d, %d, %d), u=%d, %d -> %d", B, H, W, u, Cin, Cout);
  return pconv_launch(x, packed_fwd, y, B, H, W, Cin, Cout, 1, u, (hipStream_t)stream, 0, nullptr, bn);
}

extern "C" int glx_deconv_input_grad(const float* gy, int B, int H, int W, int Cin, const void* packed_bwd, int Cout, int u,
                                     float* gx, void* stream) {
  GLX_REQUIRE(B > 0 && H > 0 && W > 0 && (u == 1 || u == 2) && Cin % 64 == 0 && Cout % 64 == 0,
              "glx_deconv_input_grad: bad sizes (%d, %d, %d), u=%d, %d -> %d", B, H, W, u, Cin, Cout);
  return pconv_launch(gy, packed_bwd, gx, B, H, W, Cout, Cin, u, 1, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ weight gradient
#define PW_XPLANE (PC_TM * 64)           // 8 192: [pixel][32 ci] bf16, the two 32-byte halves swapped on odd 8-pixel groups
#define PW_LDS (3 * PW_XPLANE)

struct PwgradArgs {
  const float* x;     // (B, Hc, Wc, Cin)
  const float* gy;    // (B, Hc * u, Wc * u, Cout)
  float* ws;          // (blocks, taps * 2 * 4 * 256)
  int B, Hc, Wc, Cin, Cout, u, taps, M, mtiles, nq_ci, nq, P;
};

// A block owns (64 Cout x 32 Cin x u*u taps) over its coarse-pixel tiles; wave = one 16-channel tile of Cout.
// Needs Wc % 8 == 0 (a lane's 8 consecutive coarse pixels share a row).
template <int TAPS>
__global__ __launch_bounds__(256, 2) void k_pconv_wgrad(PwgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, kq = lane >> 4;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int q = j % a.nq, p = (j / a.nq) * 8 + xcd;
  const int ib = q % a.nq_ci, cb = q / a.nq_ci;
  const int hw = a.Hc * a.Wc, u = a.u;
  const long long gyrow = (long long)a.Wc * u * a.Cout;        // floats per fine row

  f32x4 acc[TAPS][2];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int qq = (lane & 15) >> 2, pp = lane & 3;
  int rbase[2];           // lane's part of a transposing read of k-step 0: pixel 8 kq + 4 h + qq, 4 channels at pp
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int px = 8 * kq + 4 * h + qq;
    rbase[h] = px * 64 + (((px >> 3) & 1) << 5) + pp * 8;
  }

  for (int tile = p; tile < a.mtiles; tile += a.P) {
    const int m0 = tile * PC_TM;
    // ---- x tile -> LDS (three planes)
    f32x4 areg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, m = m0 + (e >> 3);
      areg[i] = m < a.M ? *reinterpret_cast<const f32x4*>(a.x + (long long)m * a.Cin + ib * 32 + (e & 7) * 4)
                        : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the lane's gy pixels: k-step s -> coarse pixels m0 + 32 s + 8 kq + jj, jj = 0..7 (one row: Wc % 8 == 0)
    long long gbase[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int m = m0 + 32 * s + 8 * kq;
      if (m < a.M) {
        const int b = m / hw, rem = m - b * hw, y = rem / a.Wc, x = rem - y * a.Wc;
        gbase[s] = ((long long)b * a.Hc + y) * u * gyrow + (long long)x * u * a.Cout + cb * 64 + wave * 16 + c;
      } else {
        gbase[s] = -1;
      }
    }
    __syncthreads();          // the previous tile's reads of the image are done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, px = e >> 3, seg = e & 7;
      char* d = smem + px * 64 + (((seg >> 2) ^ ((px >> 3) & 1)) << 5) + (seg & 3) * 8;
      bf16x4 p0, p1, p2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        __bf16 uu, vv, ww;
        cv_split(areg[i][k], uu, vv, ww);
        p0[k] = uu; p1[k] = vv; p2[k] = ww;
      }
      *reinterpret_cast<bf16x4*>(d) = p0;
      *reinterpret_cast<bf16x4*>(d + PW_XPLANE) = p1;
      *reinterpret_cast<bf16x4*>(d + 2 * PW_XPLANE) = p2;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const long long toff = (long long)(t / u) * gyrow + (long long)(t % u) * a.Cout;
      bf16x8 ga[4][3];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const float g = gbase[s] >= 0 ? a.gy[gbase[s] + toff + (long long)jj * u * a.Cout] : 0.f;
          __bf16 uu, vv, ww;
          cv_split(g, uu, vv, ww);
          ga[s][0][jj] = uu; ga[s][1][jj] = vv; ga[s][2][jj] = ww;
        }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          bf16x8 xb[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            typedef i16x4 __attribute__((address_space(3))) * lds_p;
            const int off = pl * PW_XPLANE + s * 32 * 64;
            i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem + off + (rbase[0] ^ (n << 5))));
            i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem + off + (rbase[1] ^ (n << 5))));
            xb[pl] = wg_join(lo, hi);
          }
          BF3_MFMA6(acc[t][n], ga[s], xb);
        }
    }
  }
  // ---- the block's partial sums: element ((tap * 2 + n) * 4 + reg) * 256 + tid
  float* dst = a.ws + (size_t)(q * a.P + p) * (TAPS * 2 * 4 * 256) + tid;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) dst[((t * 2 + n) * 4 + rg) * 256] = acc[t][n][rg];
}

// 64 partial elements x 4 segments of the quadrant's blocks per 256 threads; dW (Cin, Cout, u, u) through its strides
__global__ __launch_bounds__(256) void k_pconv_wgrad_reduce(const float* __restrict__ ws, int P, int nq_ci, int part, int u,
                                                            float* __restrict__ dW, long long s_ci, long long s_co,
                                                            long long s_kh, long long s_kw) {
  __shared__ float psum[4][64];
  const int q = blockIdx.y, el = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const float* src = ws + (size_t)q * P * part + e;
  float sum = 0.f;
  for (int pb = seg; pb < P; pb += 4) sum += src[(size_t)pb * part];
  psum[seg][el] = sum;
  __syncthreads();
  if (seg == 0) {
    sum = (psum[0][el] + psum[1][el]) + (psum[2][el] + psum[3][el]);
    const int tap = e >> 11, n = (e >> 10) & 1, rg = (e >> 8) & 3, t = e & 255;
    const int wave = t >> 6, lane = t & 63;
    const int ib = q % nq_ci, cb = q / nq_ci;
    const int co = cb * 64 + wave * 16 + 4 * (lane >> 4) + rg, ci = ib * 32 + n * 16 + (lane & 15);
    dW[ci * s_ci + co * s_co + (tap / u) * s_kh + (tap % u) * s_kw] = sum;
  }
}

static int pwgrad_blocks(int Cin, int Cout, int* P_out) {
  const int nq = (Cin / 32) * (Cout / 64);
  int P = 512 / nq / 8 * 8;
  if (P < 8) P = 8;
  *P_out = P;
  return nq * P;
}

extern "C" size_t glx_deconv_wgrad_workspace_bytes(int Cin, int Cout, int u) {
  int P = 0;
  return glx_align((size_t)pwgrad_blocks(Cin, Cout, &P) * u * u * 2 * 4 * 256 * sizeof(float));
}

extern "C" int glx_deconv_wgrad(const float* x, const float* gy, int B, int H, int W, int Cin, int Cout, int u, float* dW,
                                long long s_ci, long long s_co, long long s_kh, long long s_kw, void* workspace,
                                size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(B > 0 && H > 0 && W > 0 && (u == 1 || u == 2) && Cin % 64 == 0 && Cout % 64 == 0 && W % 8 == 0,
              "glx_deconv_wgrad: bad sizes (%d, %d, %d), u=%d, %d -> %d (needs W %% 8 == 0)", B, H, W, u, Cin, Cout);
  GLX_REQUIRE(workspace_bytes >= glx_deconv_wgrad_workspace_bytes(Cin, Cout, u), "glx_deconv_wgrad: workspace too small");
  PwgradArgs a;
  a.x = x; a.gy = gy; a.ws = (float*)workspace;
  a.B = B; a.Hc = H; a.Wc = W; a.Cin = Cin; a.Cout = Cout; a.u = u; a.taps = u * u;
  a.M = B * H * W;
  a.mtiles = glx_divup(a.M, PC_TM);
  a.nq_ci = Cin / 32;
  a.nq = a.nq_ci * (Cout / 64);
  const int blocks = pwgrad_blocks(Cin, Cout, &a.P);
  hipStream_t st = (hipStream_t)stream;
  if (u == 1)
    hipLaunchKernelGGL(k_pconv_wgrad<1>, dim3(blocks), dim3(256), PW_LDS, st, a);
  else
    hipLaunchKernelGGL(k_pconv_wgrad<4>, dim3(blocks), dim3(256), PW_LDS, st, a);
  GLX_LAUNCH_CHECK();
  const int part = a.taps * 2 * 4 * 256;
  hipLaunchKernelGGL(k_pconv_wgrad_reduce, dim3(part / 64, a.nq), dim3(256), 0, st, a.ws, a.P, a.nq_ci, part, u, dW, s_ci,
                     s_co, s_kh, s_kw);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// The strided 3x3 convolution of the second BEV block (base_bev_backbone.py:33-36: ZeroPad2d(1) + Conv2d(c, 2c, 3, stride 2)):
// FORWARD only, on k_pconv with a 3 x 3 window of input taps around (2 y, 2 x) and the piece image glx_conv3x3_pack writes
// for the stride-1 layers ([tap][chunk][plane][Cout][32] is what k_pconv reads for nine input taps and one output tap).
// x (B, H, W, Cin), H and W even -> y (B, H / 2, W / 2, Cout).  Bit-reproducible, which the vendor kernel for this layer is
// not (split-K atomics in a FORWARD pass: every ReLU mask behind it can flip from run to run).
extern "C" int glx_conv3x3s2_forward_ex(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout,
                                        float* y, const glx_epilogue* epilogue, void* stream) {
  GLX_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cin % 32 == 0 && Cout % 64 == 0,
              "glx_conv3x3s2_forward: bad sizes (%d, %d, %d), %d -> %d (even maps, Cin %% 32, Cout %% 64)", B, H, W, Cin, Cout);
  return pconv_launch(x, packed_fwd, y, B, H / 2, W / 2, Cin, Cout, 2, 1, (hipStream_t)stream, 1, epilogue);
}
// ... with the training-mode BatchNorm statistics of y taken in the epilogue (contract of glx_bn_stats / glx_conv_opts.bn)
extern "C" int glx_conv3x3s2_forward_bn(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout,
                                        float* y, const glx_bn_stats* bn, void* stream) {
  GLX_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cin % 32 == 0 && Cout % 64 == 0 && bn,
              "glx_conv3x3s2_forward_bn: bad sizes (%d, %d, %d), %d -> %d (even maps, Cin %% 32, Cout %% 64)", B, H, W, Cin, Cout);
  return pconv_launch(x, packed_fwd, y, B, H / 2, W / 2, Cin, Cout, 2, 1, (hipStream_t)stream, 1, nullptr, bn);
}
extern "C" int glx_conv3x3s2_forward(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout, float* y,
                                     void* stream) {
  return glx_conv3x3s2_forward_ex(x, B, H, W, Cin, packed_fwd, Cout, y, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------ gradients of the strided layer
// ZeroPad2d(1) + Conv2d(c, 2c, 3, stride 2) (base_bev_backbone.py:33-38) is the stride-1 convolution sampled at the even pixels, so
// its gradients are the stride-1 kernels' (glx_conv3x3_forward on the flipped pack, glx_conv3x3_wgrad) on the output gradient
// spread back over the stride-1 map: out (B, 2H, 2W, C) = gy (B, H, W, C) at the even pixels, zero elsewhere.  One pass, 16-byte
// pieces (C % 4 == 0).
__global__ __launch_bounds__(256) void k_spread2(const float* __restrict__ gy, float* __restrict__ out, int B, int H, int W, int C) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;         // one 16-byte piece of `out`
  const int c4 = C >> 2;
  const long long n = (long long)B * 2 * H * 2 * W * c4;
  if (e >= n) return;
  const int c = (int)(e % c4);
  long long p = e / c4;
  const int x = (int)(p % (2 * W));
  p /= 2 * W;
  const int y = (int)(p % (2 * H));
  const int b = (int)(p / (2 * H));
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!((x | y) & 1)) v = reinterpret_cast<const float4*>(gy)[(((long long)b * H + (y >> 1)) * W + (x >> 1)) * c4 + c];
  reinterpret_cast<float4*>(out)[e] = v;
}

extern "C" int glx_spread_stride2(const float* gy, int B, int H, int W, int C, float* out, void* stream) {
  GLX_REQUIRE(gy && out, "glx_spread_stride2: null pointer");
  GLX_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "glx_spread_stride2: (%d, %d, %d, %d), C %% 4", B, H, W, C);
  const long long n = (long long)B * 2 * H * 2 * W * (C / 4);
  hipLaunchKernelGGL(k_spread2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gy, out, B, H, W, C);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
