// glx_conv2d.hip -- the 3x3 / stride 1 / pad 1 convolutions of the BEV backbone (SURVEY 8a row a21;
// pcdet/models/backbones_2d/base_bev_backbone.py:30-49) on channels-last fp32 maps.
//
// CDNA4 has no reduced-precision fp32 matrix mode and its fp32 MFMA (v_mfma_f32_16x16x4_f32, 256 flop / cycle / CU)
// runs at 1/16 of the bf16 rate (v_mfma_f32_16x16x32_bf16, 4096 flop / cycle / CU).  These kernels therefore compute
// the fp32 products ON THE BF16 PIPE, exactly enough: an fp32 value is the sum of three bf16 pieces
//     x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (8 + 8 + 8 significand bits),
// a product x * w is the sum of nine piece products, and the six with i + j <= 4 carry everything above 2^-23 |x w|
// (the dropped three are below one fp32 rounding of the product).  Six bf16 MFMAs with fp32 accumulation replace eight
// fp32 MFMAs of the same tile at a quarter of their cycles each: 6 / 16 of the matrix time at fp32 accuracy (piece
// products are exact in fp32; the accumulation is the fp32 accumulation of the matrix pipe, as in the fp32 form).
// That is the bf16 x 3 arithmetic: the weight gradient's, and the forward / input-gradient kernel's under
// GLX_CONV3X3_ARITH=bf16x3.  By default the forward / input-gradient kernel halves the instruction count once more with
// TWO fp16 pieces per operand and THREE products (f16 x 2, described in front of the pack kernels below).
//
// Implicit GEMM, output stationary: a block of 4 waves owns a TH x 16 pixel tile (TH = 7 or 8 rows, per launch) x 64 output
// channels; wave w owns ONE 16-channel tile for all pixel rows (TH accumulators).  K runs over (32-channel chunk of Cin) x
// (9 taps).  Per chunk the (TH + 2) x 18 halo of the tile is loaded once (fp32, coalesced 128-byte pieces), split into its
// fp16 / bf16 planes in registers and kept in LDS (96-byte pixel rows: the 16-lane ds_read_b128 of an operand is
// conflict-free); a tap is then only a shifted read of that image.  The weights are pre-split per step by k_conv3x3_pack into
// [tap][chunk][plane][Cout][32] pieces; a wave's fragments go L2 -> registers one tap ahead (no weight image in LDS, two
// barriers per nine taps).  The input gradient is the same kernel on the flipped, transposed pack (written by the same pack
// launch).
#include "glx_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "glx_bn_state.h"
#include "glx_bf16x3.h"

#define CV_TH 8
#define CV_TW 16
#define CV_HW (CV_TW + 2)
#define CV_HP ((CV_TH + 2) * CV_HW)      // 180 halo pixels
// Bytes per LDS pixel row of a plane: 32 bf16 + padding.  A ds_read_b128 is served in four groups of 16 lanes that are NOT
// consecutive lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS): with the operand
// map lane = (pixel column r = lane & 15, k quarter lane >> 4) a group holds all 16 pixels, eight with one quarter and eight with
// the next.  80-byte rows (what this file used through round 4, laid out for 16 CONSECUTIVE lanes) are 2-way conflicted on
// that map -- SQ_LDS_BANK_CONFLICT was half of the kernel's LDS-active cycles (profiles/r05_bev_mfma_start.md) -- and so are
// 64-, 112- and 144-byte rows; 96-byte rows are conflict-free for every start pixel (enumerated: tools/lds_rows.py).
// 50.1 -> 48.7 us on the 64 -> 64 layer, 54.5 -> 53.5 on 128 -> 128.  Three blocks of 3 x 162 x 96 = 46.7 KB still share a CU.
#ifndef CV_ROW
#define CV_ROW 96
#endif
#define CV_APLANE (CV_HP * CV_ROW)       // 17 280
#define CV_BN 64                         // output channels per block
#define CV_ALOADS ((CV_HP * 8 + 255) / 256)    // 16-byte pieces of the halo per thread (6)

// Arithmetic of the forward / input-gradient kernel (k_conv3x3_v2<.., F16>), chosen per process (GLX_CONV3X3_ARITH = f16x2 |
// bf16x3, glx_conv3x3_set_arith; the packs carry the layout of the arithmetic they were made under):
//   f16x2 (default): fp32 products from TWO fp16 pieces per operand and THREE MFMAs (a a', a b', b a').  fp16 has 11 significand
//     bits and a narrow exponent, so both operands are scaled by powers of two (exact) into fp16's range first: the filter per
//     OUTPUT channel once, by the pack kernels (exponent array behind the planes); the halo image per staged 32-channel chunk by
//     the maximum over the block's chunk, kept as a running exponent of the tile -- a later chunk with larger values lowers it
//     and the accumulators are rescaled (ldexp: exact) -- and taken out in the epilogue.  An operand carries 22 bits where its
//     second piece is a normal fp16 number (values within 2^-18 of the maximum of their 10 x 18 pixel x 32 channel chunk; below
//     that the absolute resolution is 2^-39 of that maximum), a product >= 20.5 bits (measured 2^-21.1 at worst on full-
//     significand operands): the error against an fp64 convolution stays that of the vendor's fp32 kernels
//     (tests/test_conv2d_gpu.py).  Half the matrix instructions, two LDS planes instead of three: 34 / 37 us where bf16x3 takes
//     49 / 54 (64 -> 64 @ 4 x 200 x 176 | 128 -> 128 @ 4 x 100 x 88), 5.70 against 6.01 ms on the training step.
//   bf16x3: three bf16 pieces (24 bits, fp32's own exponent range: no scaling), six MFMAs, products exact to 2^-22 whatever the
//     operands' magnitudes.  The weight gradient (k_conv3x3_wgrad) and glx_deconv2d.hip use this form regardless: their time
//     is not in the matrix pipe (profiles/r05_bev_mfma.md 3b).
static int env_conv_f16() {
  const char* e = getenv("GLX_CONV3X3_ARITH");
  return !(e && (!strcmp(e, "bf16x3") || !strcmp(e, "0")));
}
static int g_conv_f16 = env_conv_f16();
extern "C" int glx_conv3x3_set_arith(int f16x2) {        // returns the previous setting; packs made before are stale afterwards
  const int old = g_conv_f16;
  g_conv_f16 = f16x2 ? 1 : 0;
  return old;
}
extern "C" int glx_conv3x3_get_arith(void) { return g_conv_f16; }

// W (Cout, Cin, 3, 3) with element strides (s_co, s_ci, s_kh, s_kw) ->
//   fwd [tap][Cin/32][3][Cout][32]   (the conv itself)
//   bwd [tap'][Cout/32][3][Cin][32]  (its input gradient: a conv Cout -> Cin with W'[ci][co][kh'][kw'] = W[co][ci][2-kh'][2-kw'])
struct ConvPackJob {
  const float* W;
  long long s_co, s_ci, s_kh, s_kw;
  int Cin, Cout;
  uint16_t* fwd;
  uint16_t* bwd;
  int f16;               // 1: two fp16 planes + the exponent arrays, 0: three bf16 planes
};
#define CV_PACK_MAX_JOBS 16
struct ConvPackJobs { ConvPackJob j[CV_PACK_MAX_JOBS]; };

// the planes of an f16x2 pack are followed by its exponent array: int32 per output channel of the pack's convolution
__host__ __device__ inline size_t cv_pack_plane_elems(int Cin, int Cout, int npl) { return (size_t)9 * npl * Cin * Cout; }

// f16x2: e[c] = 14 - floor(log2 max |W[c]|) over the filter of output channel c -- of the convolution itself (block < Cout:
// c = co, maximum over ci and the taps) and of its input gradient (block >= Cout: c = ci, maximum over co and the taps)
__global__ __launch_bounds__(256) void k_conv3x3_wexp(ConvPackJobs jobs) {      // blockIdx.y = job, blockIdx.x = channel
  const ConvPackJob jb = jobs.j[blockIdx.y];
  const int Cin = jb.Cin, Cout = jb.Cout;
  if (!jb.f16 || (int)blockIdx.x >= Cin + Cout) return;
  const bool fwd = (int)blockIdx.x < Cout;
  const int c = fwd ? blockIdx.x : blockIdx.x - Cout, n = (fwd ? Cin : Cout) * 9;
  uint16_t* dst = fwd ? jb.fwd : jb.bwd;
  if (!dst) return;
  float m = 0.f;
  for (int e0 = threadIdx.x; e0 < n; e0 += 4 * 256) {       // four independent loads in flight per thread
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 256 < n ? e0 + u * 256 : e0;
      const int o = e / 9, tap = e % 9;
      const int co = fwd ? c : o, ci = fwd ? o : c;
      v[u] = jb.W[co * jb.s_co + ci * jb.s_ci + (tap / 3) * jb.s_kh + (tap % 3) * jb.s_kw];
    }
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  __shared__ float s_m[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int e = cv_block_exponent(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])));
    reinterpret_cast<int*>(dst + cv_pack_plane_elems(Cin, Cout, 2))[c] = e == 127 ? 0 : e;
  }
}

__global__ void k_conv3x3_pack(ConvPackJobs jobs) {      // blockIdx.y = job
  const ConvPackJob jb = jobs.j[blockIdx.y];
  const int Cin = jb.Cin, Cout = jb.Cout;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Cout * Cin * 9) return;
  const int ci = e % Cin, co = (e / Cin) % Cout, tap = e / (Cin * Cout);
  const int kh = tap / 3, kw = tap % 3;
  const float w = jb.W[co * jb.s_co + ci * jb.s_ci + kh * jb.s_kh + kw * jb.s_kw];
  if (jb.f16) {
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      uint16_t* dst = dir ? jb.bwd : jb.fwd;
      if (!dst) continue;
      const int ex = reinterpret_cast<const int*>(dst + cv_pack_plane_elems(Cin, Cout, 2))[dir ? ci : co];
      _Float16 p[2];
      cv_split2(ldexpf(w, ex), p[0], p[1]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint16_t bits = __builtin_bit_cast(uint16_t, p[q]);
        if (dir) dst[((((size_t)(8 - tap) * (Cout / 32) + co / 32) * 2 + q) * Cin + ci) * 32 + (co & 31)] = bits;
        else dst[((((size_t)tap * (Cin / 32) + ci / 32) * 2 + q) * Cout + co) * 32 + (ci & 31)] = bits;
      }
    }
    return;
  }
  __bf16 p[3];
  cv_split(w, p[0], p[1], p[2]);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const uint16_t bits = __builtin_bit_cast(uint16_t, p[q]);
    if (jb.fwd) jb.fwd[((((size_t)tap * (Cin / 32) + ci / 32) * 3 + q) * Cout + co) * 32 + (ci & 31)] = bits;
    if (jb.bwd) jb.bwd[((((size_t)(8 - tap) * (Cout / 32) + co / 32) * 3 + q) * Cin + ci) * 32 + (co & 31)] = bits;
  }
}

struct ConvArgs {
  const float* x;        // (B, H, W, Cin)
  const uint16_t* wp;    // packed pieces
  const int* wexp;       // f16x2: the pack's exponent per output channel
  float* y;              // (B, H, W, Cout)
  int B, H, W, Cin, Cout, tiles_x, tiles_y, nblk, ntiles;   // nblk = Cout / 64; ntiles = B * tiles_y * tiles_x * nblk
  int th;                // pixel rows per tile (CV_TH; the second form picks 6, 7 or 8 per launch)
  BnState* bn_state;     // STATS: training-mode BatchNorm statistics of y taken in the epilogue
  BnFinalize bn;
  const float* epi_scale;   // inference epilogue (glx_conv_opts.epilogue): y = relu?(conv * scale[c] + shift[c]), second form
  const float* epi_shift;
  int epi_relu;
  const float* pre_scale;   // input transform on load (glx_conv_opts.prologue): x' = relu?(x * scale[c] + shift[c]) at the pixels
  const float* pre_shift;   // of the map (the zero padding stays zero) -- the BatchNorm (+ ReLU) of the layer in front, second form
  int pre_relu;
  // BWD (glx_conv_opts.bn_bwd; the launch is the INPUT-GRADIENT convolution of a layer whose input was relu(bn(y_prev))):
  // the epilogue masks the gradient with the ReLU (re-derived from y_prev * scale + shift), writes dz and takes the two sums
  // of the BatchNorm backward (sum dz, sum dz * xhat) -- bn_state / bn then finalize them (coef (3 C), dgamma, dbeta)
  const float* bwd_y;       // (B, H, W, Cout) raw output of the convolution in front of that BatchNorm
  const float* bwd_coef;    // scale[Cout], shift[Cout]
  const float* bwd_mean;
  const float* bwd_invstd;
};

struct ConvTile {
  int b, y0, x0, n0;
};

__device__ __forceinline__ ConvTile cv_tile(const ConvArgs& a, int t) {
  ConvTile c;
  c.n0 = (t % a.nblk) * CV_BN;            // the channel blocks of one pixel tile run side by side (shared halo in L2)
  t /= a.nblk;
  c.x0 = (t % a.tiles_x) * CV_TW;
  t /= a.tiles_x;
  c.y0 = (t % a.tiles_y) * a.th;
  c.b = t / a.tiles_y;
  return c;
}

// ------------------------------------------------------------------------------------------------ forward
// (The first form of this kernel -- the tap's weight image staged in LDS, one barrier per tap, two blocks per CU -- was removed in
// round 6: profiles/LABBOOK_r01_r04.md has its measurements.)
// Persistent blocks: block i takes the tiles i, i + grid, ...; the first halo chunk and weight slice of the NEXT tile are requested
// during the last taps of the current one.  Tile = TH x 16 pixels x 64 channels, 4 waves: wave w owns ONE 16-channel tile for all
// pixel rows.  Its weight fragments (3 planes x 16 bytes per lane per step) then belong to it alone and come straight from
// L2 into the MFMA operand registers, one step ahead -- no weight image in LDS, no barrier per tap: the halo image only
// changes per 32-channel chunk, so a block synchronises twice per NINE taps instead of once per tap, and with 43 KB of
// LDS three blocks share a CU.  The price is 24 instead of 6 row-operand reads per wave and step (every wave reads the
// whole tile): 96 instead of 72 LDS reads per block and step, none of them weights.
// TH = pixel rows per tile (accumulators per wave), 7 or 8, picked per launch so that the tile count fits whole rounds of
// the 768 resident blocks: 200 rows x 176 columns x 4 frames are 1100 tiles of 8 rows (1.43 rounds: a second round with
// 43 % of the blocks) or 1276 of 7 (1.66); 100 x 88 x 4 with two channel blocks are 720 units of 7 rows, one round.
// (Six rows measured the same as seven and are instantiated for experiments only: GLX_CONV3X3_TH=6.)

// PRE: the input is transformed on load (a.pre_scale / a.pre_shift, staged in LDS behind the halo planes: the kernel sits
// at its 168-register budget, eight more live registers for the coefficients spilled 27 more -- as a template parameter the
// plain kernels compile as before).
template <bool STATS, int TH, bool PRE, bool BWD, bool F16>
__global__ __launch_bounds__(256, 3) void k_conv3x3_v2(ConvArgs a) {
  constexpr int NPL = F16 ? 2 : 3;      // operand planes
  static_assert(!(BWD && (STATS || PRE)), "BWD is a mode of the plain input-gradient launch");
  constexpr int HP = (TH + 2) * CV_HW, NL = (HP * 8 + 255) / 256, PLANE = HP * CV_ROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  float* s_max = reinterpret_cast<float*>(smem + NPL * PLANE);     // f16x2: the waves' maxima of the staged chunk
  float* s_pre = reinterpret_cast<float*>(smem + NPL * PLANE + 16);      // PRE: scale[Cin], shift[Cin]
  if constexpr (PRE) {
    for (int e = threadIdx.x; e < a.Cin; e += 256) {
      s_pre[e] = a.pre_scale[e];
      s_pre[a.Cin + e] = a.pre_shift[e];
    }
    __syncthreads();          // the first chunk is transformed in front of the loop's first barrier
  }
  if constexpr (BWD) {                                             // scale, shift, mean, invstd of the BatchNorm: 4 x Cout
    for (int e = threadIdx.x; e < a.Cout; e += 256) {
      s_pre[e] = a.bwd_coef[e];
      s_pre[a.Cout + e] = a.bwd_coef[a.Cout + e];
      s_pre[2 * a.Cout + e] = a.bwd_mean[e];
      s_pre[3 * a.Cout + e] = a.bwd_invstd[e];
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int nch = a.Cin >> 5;
  int tile = blockIdx.x;
  if (tile >= a.ntiles) return;
  ConvTile ct = cv_tile(a, tile);

  int aoff[NL];
  const int adst0 = (tid >> 3) * CV_ROW + (tid & 7) * 8;
  const size_t wslice = (size_t)a.Cout * 32;
  // the lane's 16 bytes of a plane of a weight slice: channel n0 + 16 wave + r, k = 8 kq ..
  const uint16_t* wsrc = a.wp + (size_t)(ct.n0 + 16 * wave + r) * 32 + kq * 8;
  f32x4 areg[NL];
  typedef typename std::conditional<F16, f16x8, bf16x8>::type cvop8;
  cvop8 wcur[NPL], wnxt[NPL];
#define V2_HALO(T)                                                                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < NL; ++i_) {                                                   \
    const int e_ = tid + i_ * 256;                                                                      \
    const int hp_ = e_ >> 3, seg_ = e_ & 7;                                                             \
    const int gy_ = (T).y0 - 1 + hp_ / CV_HW, gx_ = (T).x0 - 1 + hp_ % CV_HW;                           \
    const bool ok_ = e_ < HP * 8 && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W;                      \
    aoff[i_] = ok_ ? (((T).b * a.H + gy_) * a.W + gx_) * a.Cin + seg_ * 4 : -1;                         \
  }
#define V2_LOAD_A(CH)                                                                                   \
  _Pragma("unroll") for (int i_ = 0; i_ < NL; ++i_)                                                     \
    areg[i_] = aoff[i_] >= 0 ? *reinterpret_cast<const f32x4*>(a.x + aoff[i_] + (CH) * 32) : f32x4{0.f, 0.f, 0.f, 0.f};
// the staged chunk, transformed in place (PRE); f16x2: and the block's maximum of it on its way (s_max, read behind the barrier)
#define V2_PREP_A()                                                                                     \
  {                                                                                                     \
    float m_ = 0.f;                                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < NL; ++i_) {                                                 \
      if (PRE && aoff[i_] >= 0) {                                                                       \
        const f32x4 sc_ = *reinterpret_cast<const f32x4*>(s_pre + pre_ch * 32 + (tid & 7) * 4);         \
        const f32x4 sh_ = *reinterpret_cast<const f32x4*>(s_pre + a.Cin + pre_ch * 32 + (tid & 7) * 4); \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                              \
          const float t_ = __fmaf_rn(areg[i_][j_], sc_[j_], sh_[j_]);                                   \
          areg[i_][j_] = a.pre_relu ? fmaxf(t_, 0.f) : t_;                                              \
        }                                                                                               \
      }                                                                                                 \
      if (F16) {                                                                                        \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) m_ = fmaxf(m_, fabsf(areg[i_][j_]));           \
      }                                                                                                 \
    }                                                                                                   \
    if (F16) {                                                                                          \
      _Pragma("unroll") for (int o_ = 32; o_ > 0; o_ >>= 1) m_ = fmaxf(m_, __shfl_xor(m_, o_, 64));     \
      if (lane == 0) s_max[wave] = m_;                                                                  \
    }                                                                                                   \
  }
// f16x2: the running exponent of the tile follows the chunk's maximum down (accumulators rescaled: exact), the chunk is converted
// at it; bf16x3: three pieces as they are
#define V2_STORE_A()                                                                                    \
  if constexpr (F16)                                                                                    \
  {                                                                                                     \
    const float bm_ = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));                      \
    const int ex_ = __builtin_amdgcn_readfirstlane(cv_block_exponent(bm_));                             \
    if (ex_ < etile) {                                                                                  \
      if (etile != 127) {                                                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < TH; ++i_)                                               \
          _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) acc[i_][j_] = ldexpf(acc[i_][j_], ex_ - etile); \
      }                                                                                                 \
      etile = ex_;                                                                                      \
    }                                                                                                   \
    const int es_ = etile == 127 ? 0 : etile;                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < NL; ++i_) {                                                 \
      if (tid + i_ * 256 < HP * 8) {                                                                    \
        char* d_ = sA + adst0 + i_ * 32 * CV_ROW;                                                       \
        f16x4 p0_, p1_;                                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                              \
          _Float16 u_, v_;                                                                              \
          cv_split2(ldexpf(areg[i_][j_], es_), u_, v_);                                                 \
          p0_[j_] = u_; p1_[j_] = v_;                                                                   \
        }                                                                                               \
        *reinterpret_cast<f16x4*>(d_) = p0_;                                                            \
        *reinterpret_cast<f16x4*>(d_ + PLANE) = p1_;                                                    \
      }                                                                                                 \
    }                                                                                                   \
  }                                                                                                     \
  else                                                                                                  \
  {                                                                                                     \
  _Pragma("unroll") for (int i_ = 0; i_ < NL; ++i_) {                                                   \
    if (tid + i_ * 256 < HP * 8) {                                                                      \
      char* d_ = sA + adst0 + i_ * 32 * CV_ROW;                                                         \
      bf16x4 p0_, p1_, p2_;                                                                             \
      _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                \
        __bf16 u_, v_, w_;                                                                              \
        cv_split(areg[i_][j_], u_, v_, w_);                                                             \
        p0_[j_] = u_; p1_[j_] = v_; p2_[j_] = w_;                                                       \
      }                                                                                                 \
      *reinterpret_cast<bf16x4*>(d_) = p0_;                                                             \
      *reinterpret_cast<bf16x4*>(d_ + PLANE) = p1_;                                                     \
      *reinterpret_cast<bf16x4*>(d_ + 2 * PLANE) = p2_;                                                 \
    }                                                                                                   \
  }                                                                                                     \
  }
#define V2_LOAD_W(DST, SRC, TAP, CH)                                                                    \
  {                                                                                                     \
    const uint16_t* s_ = (SRC) + ((size_t)(TAP) * nch + (CH)) * NPL * wslice;                        \
    _Pragma("unroll") for (int q_ = 0; q_ < NPL; ++q_) DST[q_] = *reinterpret_cast<const cvop8*>(s_ + q_ * wslice); \
  }
  float ssum[4], ssq[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) ssum[g] = ssq[g] = 0.f;
  const int stats_n0 = ct.n0;

  V2_HALO(ct);
  V2_LOAD_A(0);
  V2_LOAD_W(wnxt, wsrc, 0, 0);
  const char* aBase = sA + kq * 16;
  while (true) {
    f32x4 acc[TH];
#pragma unroll
    for (int i = 0; i < TH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int etile = 127;          // f16x2: the exponent the tile's chunks are converted at (127: nothing but zeros so far)
    (void)etile;
    const int next = tile + gridDim.x;
    const bool has_next = next < a.ntiles;
    ConvTile nt = ct;
    const uint16_t* wnext_src = wsrc;
    if (has_next) {
      nt = cv_tile(a, next);
      wnext_src = a.wp + (size_t)(nt.n0 + 16 * wave + r) * 32 + kq * 8;
    }
    for (int ch = 0; ch < nch; ++ch) {
      const int pre_ch = ch;              // the thread's four channels of the staged chunk: 32 ch + 4 (tid & 7) ..
      (void)pre_ch;
      V2_PREP_A();
      __syncthreads();   // everyone is done reading the previous halo image
      V2_STORE_A();
      __syncthreads();
      const bool last = ch + 1 == nch;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int q = 0; q < NPL; ++q) wcur[q] = wnxt[q];
        if (tap < 8) {
          V2_LOAD_W(wnxt, wsrc, tap + 1, ch);
        } else if (!last) {
          V2_LOAD_W(wnxt, wsrc, 0, ch + 1);
        } else if (has_next) {
          V2_LOAD_W(wnxt, wnext_src, 0, 0);
        }
        if (tap == 6) {
          if (!last) {
            V2_LOAD_A(ch + 1);
          } else if (has_next) {
            V2_HALO(nt);
            V2_LOAD_A(0);
          }
        }
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int part = 0; part < (TH + 1) / 2; ++part) {     // two pixel rows at a time: 6 operand reads, 12 products
          cvop8 xa[2][NPL];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (2 * part + i < TH) {
              const int hp = (2 * part + i + dy) * CV_HW + r + dx;
#pragma unroll
              for (int q = 0; q < NPL; ++q) xa[i][q] = *reinterpret_cast<const cvop8*>(aBase + q * PLANE + hp * CV_ROW);
            }
          }
          __builtin_amdgcn_sched_barrier(0);       // this part's reads, then its products: keeps the scheduler from
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (2 * part + i < TH) {
              if constexpr (F16) {
                F2_MFMA3(acc[2 * part + i], wcur, xa[i]);
              } else {
                BF3_MFMA6(acc[2 * part + i], wcur, xa[i]);
              }
            }
          __builtin_amdgcn_sched_barrier(0);       // hoisting later parts' operands (170-register budget, 3 waves / SIMD)
        }
      }
    }
    // ---- epilogue: accumulator i = pixel row i, column r; channels n0 + 16 wave + 4 kq ..
    if constexpr (F16) {      // the two operands' exponents come out again (exact)
      const int et = etile == 127 ? 0 : etile;
      int ew[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) ew[g] = -(et + a.wexp[ct.n0 + 16 * wave + 4 * kq + g]);
#pragma unroll
      for (int i = 0; i < TH; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[i][g] = ldexpf(acc[i][g], ew[g]);
    }
#pragma unroll
    for (int i = 0; i < TH; ++i) {
      const int py = ct.y0 + i, px = ct.x0 + r;
      if (py < a.H && px < a.W) {
        f32x4 v = acc[i];
        if (!STATS && a.epi_scale) {           // folded eval-mode BatchNorm (+ ReLU)
          const f32x4 sc = *reinterpret_cast<const f32x4*>(a.epi_scale + ct.n0 + 16 * wave + 4 * kq);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(a.epi_shift + ct.n0 + 16 * wave + 4 * kq);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float t = __fmaf_rn(v[g], sc[g], sh[g]);
            v[g] = a.epi_relu ? fmaxf(t, 0.f) : t;
          }
        }
        const long long o_ = (((long long)ct.b * a.H + py) * a.W + px) * a.Cout + ct.n0 + 16 * wave + 4 * kq;
        if constexpr (BWD) {
          const int cb = ct.n0 + 16 * wave + 4 * kq;
          const f32x4 yv = *reinterpret_cast<const f32x4*>(a.bwd_y + o_);
          const f32x4 sc = *reinterpret_cast<const f32x4*>(s_pre + cb), sh = *reinterpret_cast<const f32x4*>(s_pre + a.Cout + cb);
          const f32x4 mu = *reinterpret_cast<const f32x4*>(s_pre + 2 * a.Cout + cb);
          const f32x4 is = *reinterpret_cast<const f32x4*>(s_pre + 3 * a.Cout + cb);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float dz = bn_affine(yv[g], sc[g], sh[g]) > 0.f ? v[g] : 0.f;
            v[g] = dz;
            ssum[g] += dz;
            ssq[g] += dz * ((yv[g] - mu[g]) * is[g]);
          }
        }
        *reinterpret_cast<f32x4*>(a.y + o_) = v;
        if (STATS) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            ssum[g] += acc[i][g];
            ssq[g] += acc[i][g] * acc[i][g];
          }
        }
      }
    }
    if (!has_next) break;
    tile = next;
    ct = nt;
    wsrc = wnext_src;
  }
#undef V2_LOAD_W
#undef V2_HALO
#undef V2_LOAD_A
#undef V2_STORE_A
  if (STATS || BWD) {
    double* red = reinterpret_cast<double*>(smem);          // [moment][64 channels]
    __shared__ int s_last2;
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
      s_last2 = __hip_atomic_fetch_add(&a.bn_state->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (s_last2) bn_finalize_sets<BWD, 256>(a.bn_state, a.bn, a.Cout, a.B * a.H * a.W, reinterpret_cast<double(*)[2]>(smem));
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][ci][dy][dx] = sum over pixels p of gy[p][co] * x[p + (dy-1, dx-1)][ci]: per tap a (Cout x pixels) x (pixels x
// Cin) product whose K index is the PIXEL, the strided index of a channels-last map.  A block owns a 64-channel block
// of Cout x a 32-channel block of Cin (all nine taps); a wave owns TWO 16-channel tiles of Cout x ONE of Cin (18
// accumulator tiles: every x fragment read from LDS feeds two tiles -- with one tile of Cout x two of Cin the LDS reads
// took as long as the MFMAs) and walks pixel tiles:
//   * gy (rows = co) is needed in ONE alignment only: every lane loads its 8 pixels of the two tiles straight from
//     global memory (16 lanes = 64 contiguous bytes) one k-step ahead and splits them in registers;
//   * x is needed in nine alignments: its halo is staged as in the forward kernel ([pixel][32 ci] rows, three planes)
//     and read with ds_read_b64_tr_b16, the transposing LDS read (a 16-lane group reads 4 pixels x 16 channels and
//     every lane receives ITS channel's 4 pixels).  64-byte pixel rows whose two 32-byte halves swap on odd 8-pixel
//     column groups make the reads of a 32-lane half (2 x 4 pixels, 8 columns apart) cover all 64 banks once.
// Every block ends with its partial sums (72 KB) in the workspace; k_conv3x3_wgrad_reduce adds the blocks of a
// (Cout block, Cin block) quadrant and writes dW through the weight's strides.
#define WG_XPLANE (CV_HP * 64)          // 11 520
#define WG_LDS (3 * WG_XPLANE)          // 34 560
#define WG_PART (9 * 2 * 4 * 256)       // floats per block partial

struct WgradArgs {
  const float* x;     // (B, H, W, Cin)
  const float* gy;    // (B, H, W, Cout)
  float* ws;          // (blocks, WG_PART)
  int B, H, W, Cin, Cout, tiles_x, tiles_y, ntiles, nq_ci, nq, P;
  const float* pre_scale;   // x' = relu?(x * scale[c] + shift[c]) on load (the BatchNorm + ReLU in front of the layer), or NULL
  const float* pre_shift;
  int pre_relu;
};

template <bool PIPE>
__global__ __launch_bounds__(256, 2) void k_conv3x3_wgrad(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, kq = lane >> 4;
  // blocks 8j .. 8j+7 land on the eight XCDs: the quadrants of one pixel-tile sequence share an XCD (gy, x in its L2)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int q = j % a.nq, p = (j / a.nq) * 8 + xcd;
  const int ib = q % a.nq_ci, cb = q / a.nq_ci;

  int adst[CV_ALOADS];
#pragma unroll
  for (int i = 0; i < CV_ALOADS; ++i) {
    const int e = tid + i * 256;
    const int hp = e >> 3, seg = e & 7, hx = hp % CV_HW;
    adst[i] = e < CV_HP * 8 ? hp * 64 + (((seg >> 2) ^ ((hx >> 3) & 1)) << 5) + (seg & 3) * 8 : -1;
  }
  f32x4 areg[CV_ALOADS];     // x halo of the next tile
  unsigned aok = 0;          // which of them are pixels of the map (the zero padding is not transformed)
  f32x4 pre_sc = f32x4{1.f, 1.f, 1.f, 1.f}, pre_sh = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.pre_scale) {         // the thread's four channels: 32 ib + 4 (tid & 7) ..
    pre_sc = *reinterpret_cast<const f32x4*>(a.pre_scale + ib * 32 + (tid & 7) * 4);
    pre_sh = *reinterpret_cast<const f32x4*>(a.pre_shift + ib * 32 + (tid & 7) * 4);
  }
  float graw[2][8];          // gy of the next k-step: channel tile 2 cp + a2, pixel 8 (kq & 1) + jj of tile row 2 s + (kq >> 1)
  const int cp = wave >> 1, nn = wave & 1;
#define WG_LOADX(T)                                                                                       \
  {                                                                                                       \
    int t_ = (T);                                                                                         \
    const int x0_ = (t_ % a.tiles_x) * CV_TW;                                                             \
    t_ /= a.tiles_x;                                                                                      \
    const int y0_ = (t_ % a.tiles_y) * CV_TH, b_ = t_ / a.tiles_y;                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < CV_ALOADS; ++i_) {                                            \
      const int e_ = tid + i_ * 256;                                                                      \
      const int hp_ = e_ >> 3, seg_ = e_ & 7;                                                             \
      const int gy_ = y0_ - 1 + hp_ / CV_HW, gx_ = x0_ - 1 + hp_ % CV_HW;                                 \
      const bool ok_ = e_ < CV_HP * 8 && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W;                  \
      aok = ok_ ? (aok | (1u << i_)) : (aok & ~(1u << i_));                                               \
      areg[i_] = ok_ ? *reinterpret_cast<const f32x4*>(a.x + (((long long)b_ * a.H + gy_) * a.W + gx_) * a.Cin + \
                                                       ib * 32 + seg_ * 4)                                 \
                     : f32x4{0.f, 0.f, 0.f, 0.f};                                                         \
    }                                                                                                     \
  }
#define WG_LOADG_(GR, T, S)                                                                               \
  {                                                                                                       \
    int t_ = (T);                                                                                         \
    const int x0_ = (t_ % a.tiles_x) * CV_TW;                                                             \
    t_ /= a.tiles_x;                                                                                      \
    const int y0_ = (t_ % a.tiles_y) * CV_TH, b_ = t_ / a.tiles_y;                                        \
    const int py_ = y0_ + 2 * (S) + (kq >> 1), px_ = x0_ + 8 * (kq & 1);                                  \
    const float* g_ = a.gy + (((long long)b_ * a.H + py_) * a.W + px_) * a.Cout + cb * 64 + cp * 32 + c;  \
    _Pragma("unroll") for (int a2_ = 0; a2_ < 2; ++a2_)                                                   \
      _Pragma("unroll") for (int jj_ = 0; jj_ < 8; ++jj_)                                                 \
        GR[a2_][jj_] = (py_ < a.H && px_ + jj_ < a.W) ? g_[(long long)jj_ * a.Cout + a2_ * 16] : 0.f;       \
  }
#define WG_LOADG(T, S) WG_LOADG_(graw, T, S)

  f32x4 acc[9][2];           // [tap][channel tile of the pair]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // lane's part of a transposed read: pixel column 8 (kq & 1) + 4 h + qq + dx of the halo, 4 channels at pp
  const int qq = (lane & 15) >> 2, pp = lane & 3;
  int rbase[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = 8 * (kq & 1) + 4 * h + qq + dx;
      rbase[h][dx] = (((kq >> 1) * CV_HW + hx) * 64 + (((hx >> 3) & 1) << 5) + pp * 8) ^ (nn << 5);
    }

  int tile = p;
  if (tile < a.ntiles) {
    WG_LOADX(tile);
    WG_LOADG(tile, 0);
  }
  while (tile < a.ntiles) {
    __syncthreads();          // the previous tile's reads of the image are done
#pragma unroll
    for (int i = 0; i < CV_ALOADS; ++i) {
      if (adst[i] >= 0) {
        bf16x4 p0, p1, p2;
        if (a.pre_scale && ((aok >> i) & 1u)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = __fmaf_rn(areg[i][e], pre_sc[e], pre_sh[e]);
            areg[i][e] = a.pre_relu ? fmaxf(t, 0.f) : t;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 u, v, w;
          cv_split(areg[i][e], u, v, w);
          p0[e] = u; p1[e] = v; p2[e] = w;
        }
        *reinterpret_cast<bf16x4*>(smem + adst[i]) = p0;
        *reinterpret_cast<bf16x4*>(smem + WG_XPLANE + adst[i]) = p1;
        *reinterpret_cast<bf16x4*>(smem + 2 * WG_XPLANE + adst[i]) = p2;
      }
    }
    __syncthreads();
    const int next = tile + a.P;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // this k-step's gy pieces (two channel tiles), then the next k-step's loads into the registers they leave
      bf16x8 ga[2][3];
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          __bf16 u, v, w;
          const float gval = graw[a2][jj];
          cv_split(gval, u, v, w);
          ga[a2][0][jj] = u; ga[a2][1][jj] = v; ga[a2][2][jj] = w;
        }
      if (s < 3) {
        WG_LOADG(tile, s + 1);
      } else if (next < a.ntiles) {
        WG_LOADG(next, 0);
      }
      if (s == 2 && next < a.ntiles) { WG_LOADX(next); }
#define WG_READX(XB, TAP)                                                                                  \
  _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_) {                                                    \
    const int off_ = pl_ * WG_XPLANE + (2 * s + (TAP) / 3) * CV_HW * 64;                                   \
    typedef i16x4 __attribute__((address_space(3))) * lds_p;                                               \
    i16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem + off_ + rbase[0][(TAP) % 3]));       \
    i16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(smem + off_ + rbase[1][(TAP) % 3]));       \
    XB[pl_] = wg_join(lo_, hi_);                                                                           \
  }
      bf16x8 xb[2][3];
      if (PIPE) { WG_READX(xb[0], 0); }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (PIPE) {
          if (tap < 8) { WG_READX(xb[(tap + 1) & 1], tap + 1); }
        } else {
          WG_READX(xb[tap & 1], tap);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
          f32x4 v = acc[tap][a2];
          BF3_MFMA6(v, ga[a2], xb[tap & 1]);
          acc[tap][a2] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef WG_READX
    }
    tile = next;
  }
#undef WG_LOADX
#undef WG_LOADG
#undef WG_LOADG_
  // ---- the block's partial sums: element ((tap * 2 + a2) * 4 + reg) * 256 + tid
  float* dst = a.ws + (size_t)(q * a.P + p) * WG_PART + tid;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) dst[((tap * 2 + n) * 4 + rg) * 256] = acc[tap][n][rg];
}

// ------------------------------------------------------------------------------------------------ weight gradient, second form
// The first form's time is in its operand paths, not in the matrix pipe (profiles/r05_bev_mfma.md 3b: one MFMA of six takes the
// same time; without either operand's global loads it lands at the MFMA floor): gy comes as sixteen 4-byte loads per lane and
// k-step, one k-step ahead, and is split by EVERY wave that needs it.  Here both operands take the path x always took: the
// block loads the tile's gy (8 x 16 pixels x 64 channels, 16-byte pieces, a tile ahead) like the x halo, converts each element
// ONCE and keeps it in LDS in the transposable layout ([pixel][32 channels] rows of 64 bytes, halves swapped on odd 8-pixel
// column groups); both MFMA operands are ds_read_b64_tr_b16 reads.  Two such images fit beside the halo only with two planes:
// the arithmetic is f16 x 2 (two fp16 pieces per operand, three products), scaled per TILE by the block's maxima of the staged
// x halo and gy tile; the accumulators carry a running exponent S = e_x + e_g across the block's tiles -- a tile whose product
// scale is smaller rescales them (ldexp: exact), a tile whose scale is larger converts its gy at S - e_x instead -- taken out
// when the partial sums are written.  Same partial-sum layout as the first form (k_conv3x3_wgrad_reduce).
#define WG2_GIMG (128 * 64)                        // one 32-channel image of the gy tile, one plane
#define WG2_GPLANE (2 * WG2_GIMG)                  // both channel halves of the block's 64 output channels
#define WG2_LDS (2 * WG_XPLANE + 2 * WG2_GPLANE + 64)   // 55 872 bytes: two blocks per CU
#define WG2_GLOADS 8                               // 16-byte pieces of the gy tile per thread

// k-steps (of four) at which the next tile's gy / x loads are issued
#ifndef W2_SG
#define W2_SG 1
#endif
#ifndef W2_SX
#define W2_SX 2
#endif
__global__ __launch_bounds__(256, 2) void k_conv3x3_wgrad2(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;                                  // [plane][halo pixel][64 B]
  char* sG = smem + 2 * WG_XPLANE;                  // [plane][channel half][pixel][64 B]
  float* s_max = reinterpret_cast<float*>(smem + 2 * WG_XPLANE + 2 * WG2_GPLANE);      // [wave][x | gy]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int q = j % a.nq, p = (j / a.nq) * 8 + xcd;
  const int ib = q % a.nq_ci, cb = q / a.nq_ci;
  const int cp = wave >> 1, nn = wave & 1;

  int adst[CV_ALOADS], gdst[WG2_GLOADS];
#pragma unroll
  for (int i = 0; i < CV_ALOADS; ++i) {
    const int e = tid + i * 256;
    const int hp = e >> 3, seg = e & 7, hx = hp % CV_HW;
    adst[i] = e < CV_HP * 8 ? hp * 64 + (((seg >> 2) ^ ((hx >> 3) & 1)) << 5) + (seg & 3) * 8 : -1;
  }
#pragma unroll
  for (int i = 0; i < WG2_GLOADS; ++i) {
    const int e = tid + i * 256;                    // pixel e >> 4 of the tile (row-major 8 x 16), channels 4 (e & 15) ..
    const int px = e >> 4, seg = e & 15, col = px & 15;
    gdst[i] = (seg >> 3) * WG2_GIMG + px * 64 + ((((seg & 7) >> 2) ^ ((col >> 3) & 1)) << 5) + (seg & 3) * 8;
  }
  f32x4 areg[CV_ALOADS], greg[WG2_GLOADS];
  unsigned aok = 0;
  f32x4 pre_sc = f32x4{1.f, 1.f, 1.f, 1.f}, pre_sh = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.pre_scale) {
    pre_sc = *reinterpret_cast<const f32x4*>(a.pre_scale + ib * 32 + (tid & 7) * 4);
    pre_sh = *reinterpret_cast<const f32x4*>(a.pre_shift + ib * 32 + (tid & 7) * 4);
  }
#define WG2_TILE(T, X0, Y0, BB)                                                                          \
  int X0, Y0, BB;                                                                                        \
  {                                                                                                      \
    int t_ = (T);                                                                                        \
    X0 = (t_ % a.tiles_x) * CV_TW;                                                                       \
    t_ /= a.tiles_x;                                                                                     \
    Y0 = (t_ % a.tiles_y) * CV_TH;                                                                       \
    BB = t_ / a.tiles_y;                                                                                 \
  }
#define WG2_LOADX(T)                                                                                     \
  {                                                                                                      \
    WG2_TILE(T, x0_, y0_, b_)                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < CV_ALOADS; ++i_) {                                           \
      const int e_ = tid + i_ * 256;                                                                     \
      const int hp_ = e_ >> 3, seg_ = e_ & 7;                                                            \
      const int gy_ = y0_ - 1 + hp_ / CV_HW, gx_ = x0_ - 1 + hp_ % CV_HW;                                \
      const bool ok_ = e_ < CV_HP * 8 && gy_ >= 0 && gy_ < a.H && gx_ >= 0 && gx_ < a.W;                 \
      aok = ok_ ? (aok | (1u << i_)) : (aok & ~(1u << i_));                                              \
      areg[i_] = ok_ ? *reinterpret_cast<const f32x4*>(a.x + (((long long)b_ * a.H + gy_) * a.W + gx_) * a.Cin + \
                                                       ib * 32 + seg_ * 4)                                \
                     : f32x4{0.f, 0.f, 0.f, 0.f};                                                        \
    }                                                                                                    \
  }
#define WG2_LOADG(T)                                                                                     \
  {                                                                                                      \
    WG2_TILE(T, x0_, y0_, b_)                                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < WG2_GLOADS; ++i_) {                                          \
      const int e_ = tid + i_ * 256;                                                                     \
      const int py_ = y0_ + (e_ >> 8), px_ = x0_ + ((e_ >> 4) & 15), seg_ = e_ & 15;                      \
      greg[i_] = (py_ < a.H && px_ < a.W)                                                                \
                     ? *reinterpret_cast<const f32x4*>(a.gy + (((long long)b_ * a.H + py_) * a.W + px_) * a.Cout + \
                                                       cb * 64 + seg_ * 4)                                \
                     : f32x4{0.f, 0.f, 0.f, 0.f};                                                        \
    }                                                                                                    \
  }

  f32x4 acc[9][2];           // [tap][channel tile of the pair]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  int esum = 254;            // the exponent e_x + e_g the accumulators are scaled by (254: nothing accumulated yet)

  // a lane's part of the transposing reads: pixel column 8 (kq & 1) + 4 h + qq (+ dx in the halo), 4 channels at pp
  const int qq = (lane & 15) >> 2, pp = lane & 3;
  int rbase[2][3], gbase[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hx = 8 * (kq & 1) + 4 * h + qq + dx;
      rbase[h][dx] = (((kq >> 1) * CV_HW + hx) * 64 + (((hx >> 3) & 1) << 5) + pp * 8) ^ (nn << 5);
    }
    const int col = 8 * (kq & 1) + 4 * h + qq;
    gbase[h] = cp * WG2_GIMG + ((kq >> 1) * 16 + col) * 64 + (((col >> 3) & 1) << 5) + pp * 8;
  }

  int tile = p;
  if (tile < a.ntiles) {
    WG2_LOADX(tile);
    WG2_LOADG(tile);
  }
  while (tile < a.ntiles) {
    // ---- the staged tile: x transformed in place, the block's maxima of both operands on their way
    float mx = 0.f, mg = 0.f;
#pragma unroll
    for (int i = 0; i < CV_ALOADS; ++i) {
      if (a.pre_scale && ((aok >> i) & 1u)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = __fmaf_rn(areg[i][e], pre_sc[e], pre_sh[e]);
          areg[i][e] = a.pre_relu ? fmaxf(t, 0.f) : t;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(areg[i][e]));
    }
#pragma unroll
    for (int i = 0; i < WG2_GLOADS; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) mg = fmaxf(mg, fabsf(greg[i][e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      mg = fmaxf(mg, __shfl_xor(mg, o, 64));
    }
    if (lane == 0) { s_max[2 * wave] = mx; s_max[2 * wave + 1] = mg; }
    __syncthreads();          // the previous tile's reads of the images are done; the maxima are there
    int ex = __builtin_amdgcn_readfirstlane(cv_block_exponent(fmaxf(fmaxf(s_max[0], s_max[2]), fmaxf(s_max[4], s_max[6]))));
    int eg = __builtin_amdgcn_readfirstlane(cv_block_exponent(fmaxf(fmaxf(s_max[1], s_max[3]), fmaxf(s_max[5], s_max[7]))));
    if (ex == 127) ex = 0;    // an operand of zeros: any scale
    if (eg == 127) eg = 0;
    if (ex + eg < esum) {
      if (esum != 254) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t][n][e] = ldexpf(acc[t][n][e], ex + eg - esum);
      }
      esum = ex + eg;
    }
    int egu = esum - ex;      // <= eg: a tile whose own product scale is larger takes the accumulators' scale
    egu = egu < -120 ? -120 : egu;
#pragma unroll
    for (int i = 0; i < CV_ALOADS; ++i) {
      if (adst[i] >= 0) {
        f16x4 p0, p1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          _Float16 u, v;
          cv_split2(ldexpf(areg[i][e], ex), u, v);
          p0[e] = u; p1[e] = v;
        }
        *reinterpret_cast<f16x4*>(sX + adst[i]) = p0;
        *reinterpret_cast<f16x4*>(sX + WG_XPLANE + adst[i]) = p1;
      }
    }
#pragma unroll
    for (int i = 0; i < WG2_GLOADS; ++i) {
      f16x4 p0, p1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 u, v;
        cv_split2(ldexpf(greg[i][e], egu), u, v);
        p0[e] = u; p1[e] = v;
      }
      *reinterpret_cast<f16x4*>(sG + gdst[i]) = p0;
      *reinterpret_cast<f16x4*>(sG + WG2_GPLANE + gdst[i]) = p1;
    }
    __syncthreads();
    const int next = tile + a.P;
    typedef i16x4 __attribute__((address_space(3))) * lds_p;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s == W2_SG && next < a.ntiles) { WG2_LOADG(next); }
      if (s == W2_SX && next < a.ntiles) { WG2_LOADX(next); }
      // this k-step's gy operands: two channel tiles x two planes
      f16x8 ga[2][2];
#pragma unroll
      for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const int off = pl * WG2_GPLANE + s * 2048;
          i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sG + off + (gbase[0] ^ (a2 << 5))));
          i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sG + off + (gbase[1] ^ (a2 << 5))));
          ga[a2][pl] = __builtin_bit_cast(f16x8, wg_join(lo, hi));
        }
#define WG2_READX(XB, TAP)                                                                                 \
  _Pragma("unroll") for (int pl_ = 0; pl_ < 2; ++pl_) {                                                    \
    const int off_ = pl_ * WG_XPLANE + (2 * s + (TAP) / 3) * CV_HW * 64;                                   \
    i16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sX + off_ + rbase[0][(TAP) % 3]));         \
    i16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(sX + off_ + rbase[1][(TAP) % 3]));         \
    XB[pl_] = __builtin_bit_cast(f16x8, wg_join(lo_, hi_));                                                \
  }
      f16x8 xb[2][2];
      WG2_READX(xb[0], 0);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap < 8) { WG2_READX(xb[(tap + 1) & 1], tap + 1); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
          f32x4 v = acc[tap][a2];
          F2_MFMA3(v, ga[a2], xb[tap & 1]);
          acc[tap][a2] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef WG2_READX
    }
    tile = next;
  }
#undef WG2_LOADX
#undef WG2_LOADG
#undef WG2_TILE
  // ---- the block's partial sums (the exponent taken out): element ((tap * 2 + a2) * 4 + reg) * 256 + tid
  const int eo = esum == 254 ? 0 : -esum;
  float* dst = a.ws + (size_t)(q * a.P + p) * WG_PART + tid;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[((tap * 2 + n) * 4 + e) * 256] = ldexpf(acc[tap][n][e], eo);
}

// 64 partial elements x 4 segments of the quadrant's blocks per 256 threads; dW through the weight's strides
__global__ __launch_bounds__(256) void k_conv3x3_wgrad_reduce(const float* __restrict__ ws, int P, int nq_ci,
                                                              float* __restrict__ dW, long long s_co, long long s_ci,
                                                              long long s_kh, long long s_kw) {
  __shared__ float part[4][64];
  const int q = blockIdx.y, el = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const float* src = ws + (size_t)q * P * WG_PART + e;
  float sum = 0.f;
  int pb = seg;
  for (; pb + 28 < P; pb += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(pb + 4 * u) * WG_PART];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; pb < P; pb += 4) sum += src[(size_t)pb * WG_PART];
  part[seg][el] = sum;
  __syncthreads();
  if (seg == 0) {
    sum = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
    const int tap = e >> 11, a2 = (e >> 10) & 1, rg = (e >> 8) & 3, t = e & 255;
    const int wave = t >> 6, lane = t & 63;
    const int ib = q % nq_ci, cb = q / nq_ci;
    const int co = cb * 64 + (wave >> 1) * 32 + a2 * 16 + 4 * (lane >> 4) + rg, ci = ib * 32 + (wave & 1) * 16 + (lane & 15);
    dW[co * s_co + ci * s_ci + (tap / 3) * s_kh + (tap % 3) * s_kw] = sum;
  }
}

static int wgrad_blocks(int Cin, int Cout, int* P_out) {
  const int nq = (Cin / 32) * (Cout / 64);
  // blocks per launch: every block leaves 72 KB of partial sums that the reduce reads back (ten layers: 2 x 377 MB per step at
  // 512 blocks); 384 measured 0.02-0.03 ms per step faster than 512 in three alternating series (256 / 320 / 448 / 768 no better)
  static const int env_slots = 384;
  int slots = env_slots >= 64 && env_slots <= 1024 ? env_slots : 384;
  int P = slots / nq;
  P = P / 8 * 8;
  if (P < 8) P = 8;
  *P_out = P;
  return nq * P;
}

// the weight gradient's kernel form: 2 (default) = k_conv3x3_wgrad2 (both operands through LDS, f16 x 2), 1 = k_conv3x3_wgrad
// (bf16 x 3, gy from global memory per k-step); GLX_WGRAD_FORM at load, glx_conv3x3_set_wgrad_form afterwards (returns the previous)
static int env_wgrad_form() {
  const char* e = getenv("GLX_WGRAD_FORM");
  return e && atoi(e) == 1 ? 1 : 2;
}
static int g_wgrad_form = env_wgrad_form();
extern "C" int glx_conv3x3_set_wgrad_form(int form) {
  const int old = g_wgrad_form;
  g_wgrad_form = form == 1 ? 1 : 2;
  return old;
}

extern "C" size_t glx_conv3x3_wgrad_workspace_bytes(int Cin, int Cout) {
  int P = 0;
  const int blocks = wgrad_blocks(Cin, Cout, &P);
  return glx_align((size_t)blocks * WG_PART * sizeof(float));
}

extern "C" int glx_conv3x3_wgrad_ex(const float* x, const float* gy, int B, int H, int W, int Cin, int Cout, float* dW,
                                    long long s_co, long long s_ci, long long s_kh, long long s_kw, const glx_epilogue* pre,
                                    void* workspace, size_t workspace_bytes, void* stream);
extern "C" int glx_conv3x3_wgrad(const float* x, const float* gy, int B, int H, int W, int Cin, int Cout, float* dW,
                                 long long s_co, long long s_ci, long long s_kh, long long s_kw, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  return glx_conv3x3_wgrad_ex(x, gy, B, H, W, Cin, Cout, dW, s_co, s_ci, s_kh, s_kw, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int glx_conv3x3_wgrad_ex(const float* x, const float* gy, int B, int H, int W, int Cin, int Cout, float* dW,
                                    long long s_co, long long s_ci, long long s_kh, long long s_kw, const glx_epilogue* pre,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(!pre || (pre->scale && pre->shift && pre->ldc == 0 && pre->coff == 0),
              "glx_conv3x3_wgrad_ex: the prologue needs scale and shift (Cin floats each)");
  GLX_REQUIRE(B > 0 && H > 0 && W > 0, "glx_conv3x3_wgrad: empty map (%d, %d, %d)", B, H, W);
  GLX_REQUIRE(Cin % 32 == 0 && Cout % CV_BN == 0, "glx_conv3x3_wgrad: needs Cin %% 32 == 0 and Cout %% 64 == 0 (got %d -> %d)",
              Cin, Cout);
  GLX_REQUIRE(workspace_bytes >= glx_conv3x3_wgrad_workspace_bytes(Cin, Cout), "glx_conv3x3_wgrad: workspace too small");
  WgradArgs a;
  a.x = x; a.gy = gy; a.ws = (float*)workspace;
  a.pre_scale = pre ? pre->scale : nullptr;
  a.pre_shift = pre ? pre->shift : nullptr;
  a.pre_relu = pre ? pre->relu : 0;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.tiles_x = glx_divup(W, CV_TW); a.tiles_y = glx_divup(H, CV_TH);
  a.ntiles = a.tiles_x * a.tiles_y * B;
  a.nq_ci = Cin / 32;
  a.nq = a.nq_ci * (Cout / CV_BN);
  const int blocks = wgrad_blocks(Cin, Cout, &a.P);
  static const bool pipe = true;
  if (g_wgrad_form == 2)
    hipLaunchKernelGGL(k_conv3x3_wgrad2, dim3(blocks), dim3(256), WG2_LDS, (hipStream_t)stream, a);
  else if (pipe)
    hipLaunchKernelGGL(k_conv3x3_wgrad<true>, dim3(blocks), dim3(256), WG_LDS, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_conv3x3_wgrad<false>, dim3(blocks), dim3(256), WG_LDS, (hipStream_t)stream, a);
  GLX_LAUNCH_CHECK();
  if (!dW) return GLX_OK;                            // the blocks' partial sums only: glx_conv3x3_wgrad_reduce finishes later
  return glx_conv3x3_wgrad_reduce(Cin, Cout, dW, s_co, s_ci, s_kh, s_kw, workspace, workspace_bytes, stream);
}

extern "C" int glx_conv3x3_wgrad_reduce(int Cin, int Cout, float* dW, long long s_co, long long s_ci, long long s_kh,
                                        long long s_kw, const void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(dW && workspace, "glx_conv3x3_wgrad_reduce: null pointer");
  GLX_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % CV_BN == 0,
              "glx_conv3x3_wgrad_reduce: needs Cin %% 32 == 0 and Cout %% 64 == 0 (got %d -> %d)", Cin, Cout);
  GLX_REQUIRE(workspace_bytes >= glx_conv3x3_wgrad_workspace_bytes(Cin, Cout), "glx_conv3x3_wgrad_reduce: workspace too small");
  int P = 0;
  wgrad_blocks(Cin, Cout, &P);
  const int nq_ci = Cin / 32, nq = nq_ci * (Cout / CV_BN);
  hipLaunchKernelGGL(k_conv3x3_wgrad_reduce, dim3(WG_PART / 64, nq), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, P,
                     nq_ci, dW, s_co, s_ci, s_kh, s_kw);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// The partial sums of SEVERAL layers in one launch (a training step defers them to the end of its backward pass): blockIdx.z = job,
// per element the same sums in the same order as k_conv3x3_wgrad_reduce.
#define WG_REDUCE_JOBS 16
struct WgReduceJob { const float* ws; float* dW; long long s_co, s_ci, s_kh, s_kw; int P, nq_ci, nq; };
struct WgReduceJobs { WgReduceJob j[WG_REDUCE_JOBS]; };
__global__ __launch_bounds__(256) void k_conv3x3_wgrad_reduce_multi(WgReduceJobs jobs) {
  __shared__ float part[4][64];
  const WgReduceJob jb = jobs.j[blockIdx.z];
  const int q = blockIdx.y, el = threadIdx.x & 63, seg = threadIdx.x >> 6;
  if (q >= jb.nq) return;                                        // block-uniform
  const int e = blockIdx.x * 64 + el;
  const float* src = jb.ws + (size_t)q * jb.P * WG_PART + e;
  float sum = 0.f;
  int pb = seg;
  for (; pb + 28 < jb.P; pb += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(pb + 4 * u) * WG_PART];
#pragma unroll
    for (int u = 0; u < 8; ++u) sum += v[u];
  }
  for (; pb < jb.P; pb += 4) sum += src[(size_t)pb * WG_PART];
  part[seg][el] = sum;
  __syncthreads();
  if (seg == 0) {
    sum = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
    const int tap = e >> 11, a2 = (e >> 10) & 1, rg = (e >> 8) & 3, t = e & 255;
    const int wave = t >> 6, lane = t & 63;
    const int ib = q % jb.nq_ci, cb = q / jb.nq_ci;
    const int co = cb * 64 + (wave >> 1) * 32 + a2 * 16 + 4 * (lane >> 4) + rg, ci = ib * 32 + (wave & 1) * 16 + (lane & 15);
    jb.dW[co * jb.s_co + ci * jb.s_ci + (tap / 3) * jb.s_kh + (tap % 3) * jb.s_kw] = sum;
  }
}

extern "C" int glx_conv3x3_wgrad_reduce_multi(int n, const int32_t* Cin, const int32_t* Cout, float* const* dW,
                                              const long long* strides, const void* const* workspace,
                                              const size_t* workspace_bytes, void* stream) {
  if (n <= 0) return GLX_OK;
  GLX_REQUIRE(Cin && Cout && dW && strides && workspace && workspace_bytes, "glx_conv3x3_wgrad_reduce_multi: null pointer");
  for (int i = 0; i < n; ++i) {
    GLX_REQUIRE(dW[i] && workspace[i], "glx_conv3x3_wgrad_reduce_multi: null pointer in job %d", i);
    GLX_REQUIRE(Cin[i] > 0 && Cout[i] > 0 && Cin[i] % 32 == 0 && Cout[i] % CV_BN == 0,
                "glx_conv3x3_wgrad_reduce_multi: job %d needs Cin %% 32 == 0 and Cout %% 64 == 0 (got %d -> %d)", i, Cin[i], Cout[i]);
    GLX_REQUIRE(workspace_bytes[i] >= glx_conv3x3_wgrad_workspace_bytes(Cin[i], Cout[i]),
                "glx_conv3x3_wgrad_reduce_multi: job %d: workspace too small", i);
  }
  for (int done = 0; done < n; done += WG_REDUCE_JOBS) {
    WgReduceJobs jobs;
    const int nj = n - done < WG_REDUCE_JOBS ? n - done : WG_REDUCE_JOBS;
    int max_nq = 0;
    for (int j = 0; j < nj; ++j) {
      const int i = done + j;
      int P = 0;
      wgrad_blocks(Cin[i], Cout[i], &P);
      const int nq_ci = Cin[i] / 32, nq = nq_ci * (Cout[i] / CV_BN);
      jobs.j[j] = WgReduceJob{(const float*)workspace[i], dW[i], strides[4 * i], strides[4 * i + 1], strides[4 * i + 2],
                              strides[4 * i + 3], P, nq_ci, nq};
      max_nq = nq > max_nq ? nq : max_nq;
    }
    hipLaunchKernelGGL(k_conv3x3_wgrad_reduce_multi, dim3(WG_PART / 64, max_nq, nj), dim3(256), 0, (hipStream_t)stream, jobs);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" size_t glx_conv3x3_packed_bytes(int Cin, int Cout) {
  return glx_align((size_t)9 * 3 * Cin * Cout * sizeof(uint16_t));
}

static int conv_pack_check(int Cin, int Cout, const void* fwd, const void* bwd) {
  GLX_REQUIRE(Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0,
              "glx_conv3x3_pack: channels must be multiples of 32 (got %d -> %d)", Cin, Cout);
  GLX_REQUIRE(fwd == nullptr || Cout % CV_BN == 0, "glx_conv3x3_pack: forward pack needs Cout %% 64 == 0 (got %d)", Cout);
  GLX_REQUIRE(bwd == nullptr || Cin % CV_BN == 0, "glx_conv3x3_pack: gradient pack needs Cin %% 64 == 0 (got %d)", Cin);
  return GLX_OK;
}

// f16x2: 1 / 0 = the layout of that arithmetic, -1 = the process setting (glx_conv3x3_set_arith).  The strided layer's forward
// (glx_conv3x3s2_forward*, csrc/glx_deconv2d.hip) reads the bf16x3 forward image whatever the process setting is.
extern "C" int glx_conv3x3_pack_arith(const float* W, long long s_co, long long s_ci, long long s_kh, long long s_kw,
                                      int Cin, int Cout, void* fwd, void* bwd, int f16x2, void* stream) {
  const int rc = conv_pack_check(Cin, Cout, fwd, bwd);
  if (rc != GLX_OK) return rc;
  const int f16 = f16x2 < 0 ? g_conv_f16 : (f16x2 ? 1 : 0);
  ConvPackJobs jobs;
  jobs.j[0] = ConvPackJob{W, s_co, s_ci, s_kh, s_kw, Cin, Cout, (uint16_t*)fwd, (uint16_t*)bwd, f16};
  if (f16) hipLaunchKernelGGL(k_conv3x3_wexp, dim3(Cin + Cout, 1), dim3(256), 0, (hipStream_t)stream, jobs);
  hipLaunchKernelGGL(k_conv3x3_pack, dim3(glx_divup(Cin * Cout * 9, 256), 1), dim3(256), 0, (hipStream_t)stream, jobs);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_conv3x3_pack(const float* W, long long s_co, long long s_ci, long long s_kh, long long s_kw,
                                int Cin, int Cout, void* fwd, void* bwd, void* stream) {
  return glx_conv3x3_pack_arith(W, s_co, s_ci, s_kh, s_kw, Cin, Cout, fwd, bwd, -1, stream);
}

// n weights in one launch (CV_PACK_MAX_JOBS per launch); arrays of length n on the HOST, strides[4 * i ..] = s_co, s_ci,
// s_kh, s_kw of weight i; f16x2: per weight 1 / 0 / -1 as above, NULL = the process setting for all
extern "C" int glx_conv3x3_pack_multi_arith(int n, const float* const* W, const long long* strides, const int32_t* Cin,
                                            const int32_t* Cout, void* const* fwd, void* const* bwd, const int32_t* f16x2,
                                            void* stream) {
  if (n <= 0) return GLX_OK;
  GLX_REQUIRE(W && strides && Cin && Cout && fwd && bwd, "glx_conv3x3_pack_multi: null pointer");
  for (int done = 0; done < n;) {
    ConvPackJobs jobs;
    int nj = 0, cover = 0, chans = 0, any16 = 0;
    for (; done < n && nj < CV_PACK_MAX_JOBS; ++done, ++nj) {
      const int rc = conv_pack_check(Cin[done], Cout[done], fwd[done], bwd[done]);
      if (rc != GLX_OK) return rc;
      const int f16 = (!f16x2 || f16x2[done] < 0) ? g_conv_f16 : (f16x2[done] ? 1 : 0);
      jobs.j[nj] = ConvPackJob{W[done], strides[4 * done], strides[4 * done + 1], strides[4 * done + 2], strides[4 * done + 3],
                               Cin[done], Cout[done], (uint16_t*)fwd[done], (uint16_t*)bwd[done], f16};
      cover = Cin[done] * Cout[done] * 9 > cover ? Cin[done] * Cout[done] * 9 : cover;
      if (f16) {
        any16 = 1;
        chans = Cin[done] + Cout[done] > chans ? Cin[done] + Cout[done] : chans;
      }
    }
    if (any16) hipLaunchKernelGGL(k_conv3x3_wexp, dim3(chans, nj), dim3(256), 0, (hipStream_t)stream, jobs);
    hipLaunchKernelGGL(k_conv3x3_pack, dim3(glx_divup(cover, 256), nj), dim3(256), 0, (hipStream_t)stream, jobs);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_conv3x3_pack_multi(int n, const float* const* W, const long long* strides, const int32_t* Cin,
                                      const int32_t* Cout, void* const* fwd, void* const* bwd, void* stream) {
  return glx_conv3x3_pack_multi_arith(n, W, strides, Cin, Cout, fwd, bwd, nullptr, stream);
}

static int conv_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n;
  }
  return cus;
}
static int g_conv_th = 0;       // experiments (glx_conv3x3_set_grid): rows per tile (0 = chosen per launch)
static int g_conv_grid = 0;     // experiments: blocks per launch (0 = as many as are meant to be resident)
// Blocks per CU of the persistent grid.  Three fit (168 registers
// each) and fill every SIMD's register file: no other kernel can start a wave while such a kernel runs.  Two per CU run
// the BEV layers within a few per cent of that and leave room for the kernels of a parallel graph branch
// (tools/conv_side_load.py: 600 small launches beside 40 convolutions, 3.28 -> 2.42 ms); inside the training step the
// two settings measured the same (7.79 / 7.83 ms), so the default stays three.
static int conv_per_cu() { return 3; }
extern "C" int glx_conv3x3_set_grid(int blocks, int rows_per_tile) {
  g_conv_grid = blocks;
  g_conv_th = rows_per_tile >= 6 && rows_per_tile <= 8 ? rows_per_tile : 0;
  return GLX_OK;
}

// the instantiations of the second form: (statistics epilogue | plain) x rows per tile x (plain | transform on load | bn_bwd) x arithmetic
template <bool F16>
static void (*conv_v2_kernel(bool stats, int th, bool pre, bool bwd))(ConvArgs) {
  if (bwd) return th == 8 ? k_conv3x3_v2<false, 8, false, true, F16> : th == 7 ? k_conv3x3_v2<false, 7, false, true, F16>
                                                                                : k_conv3x3_v2<false, 6, false, true, F16>;
  if (pre) {
    if (th == 8) return stats ? k_conv3x3_v2<true, 8, true, false, F16> : k_conv3x3_v2<false, 8, true, false, F16>;
    if (th == 7) return stats ? k_conv3x3_v2<true, 7, true, false, F16> : k_conv3x3_v2<false, 7, true, false, F16>;
    return stats ? k_conv3x3_v2<true, 6, true, false, F16> : k_conv3x3_v2<false, 6, true, false, F16>;
  }
  if (th == 8) return stats ? k_conv3x3_v2<true, 8, false, false, F16> : k_conv3x3_v2<false, 8, false, false, F16>;
  if (th == 7) return stats ? k_conv3x3_v2<true, 7, false, false, F16> : k_conv3x3_v2<false, 7, false, false, F16>;
  return stats ? k_conv3x3_v2<true, 6, false, false, F16> : k_conv3x3_v2<false, 6, false, false, F16>;
}


// opts->bn: training-mode BatchNorm statistics of y in the epilogue (same contract as glx_sconv_forward_ex); opts->epilogue:
// y = relu?(conv * scale[c] + shift[c]), an eval-mode BatchNorm folded behind the convolution (scale, shift: Cout device
// floats).  Explicit arguments: nothing is carried between calls.
extern "C" int glx_conv3x3_forward_ex(const float* x, int B, int H, int W, int Cin, const void* packed, int Cout,
                                      float* y, const glx_conv_opts* opts, void* stream) {
  const glx_bn_stats* bnp = opts ? opts->bn : nullptr;
  const glx_epilogue* epi = opts ? opts->epilogue : nullptr;
  const glx_epilogue* pre = opts ? opts->prologue : nullptr;
  const glx_bn_bwd_stats* bwd = opts ? opts->bn_bwd : nullptr;
  BnState* bn_state = bnp ? (BnState*)bnp->state : nullptr;
  BnFinalize bn_fin = {};
  if (bnp) {
    GLX_REQUIRE(bnp->state && bnp->coef && bnp->save_mean && bnp->save_invstd, "glx_conv3x3_forward_ex: BatchNorm statistics: null pointer");
    bn_fin = BnFinalize{bnp->gamma, bnp->beta, bnp->eps, bnp->momentum, bnp->coef, bnp->save_mean, bnp->save_invstd,
                        bnp->running_mean, bnp->running_var, nullptr, nullptr, nullptr};
  }
  const float* epi_scale = epi ? epi->scale : nullptr;
  const float* epi_shift = epi ? epi->shift : nullptr;
  const int epi_relu = epi ? epi->relu : 0;
  GLX_REQUIRE(!epi || (epi->scale && epi->shift && epi->ldc == 0 && epi->coff == 0),
              "glx_conv3x3_forward_ex: the epilogue needs scale and shift, dense placement");
  GLX_REQUIRE(B > 0 && H > 0 && W > 0, "glx_conv3x3_forward: empty map (%d, %d, %d)", B, H, W);
  GLX_REQUIRE(Cin % 32 == 0 && Cout % CV_BN == 0, "glx_conv3x3_forward: needs Cin %% 32 == 0 and Cout %% 64 == 0 (got %d -> %d)",
              Cin, Cout);
  GLX_REQUIRE((long long)B * H * W * (Cin > Cout ? Cin : Cout) < (1ll << 31), "glx_conv3x3_forward: map too large (2^31 elements)");
  GLX_REQUIRE(!(bn_state && epi_scale), "glx_conv3x3_forward: statistics and an inference epilogue in one call");
  GLX_REQUIRE(!bn_state || Cout <= BN_MAXC, "glx_conv3x3_forward: BatchNorm statistics for at most %d channels", BN_MAXC);
  void (*kern)(ConvArgs) = nullptr;
  GLX_REQUIRE(!pre || (pre->scale && pre->shift && pre->ldc == 0 && pre->coff == 0),
              "glx_conv3x3_forward_ex: the prologue needs scale and shift (Cin floats each)");
  if (bwd) {
    GLX_REQUIRE(!bnp && !epi && !pre, "glx_conv3x3_forward_ex: bn_bwd excludes the other options");
    GLX_REQUIRE(bwd->state && bwd->y && bwd->coef_fwd && bwd->mean && bwd->invstd && bwd->coef, "glx_conv3x3_forward_ex: bn_bwd: null pointer");
    GLX_REQUIRE(Cout <= BN_MAXC, "glx_conv3x3_forward_ex: bn_bwd for at most %d channels", BN_MAXC);
    bn_state = (BnState*)bwd->state;
    bn_fin = BnFinalize{bwd->gamma, nullptr, 0.f, 0.f, bwd->coef, nullptr, nullptr, nullptr, nullptr, bwd->invstd, bwd->dgamma, bwd->dbeta};
  }
  const bool f16 = g_conv_f16 != 0;
  int th = CV_TH;
  {
    // rows per tile: the fewest (rounds of the resident blocks) x (rows + a fixed cost per tile)
    const int resident = conv_per_cu() * conv_cus();
    long long best = -1;
    for (int t = 8; t >= 7; --t) {
      const long long units = (long long)glx_divup(W, CV_TW) * glx_divup(H, t) * B * (Cout / CV_BN);
      const long long cost = ((units + resident - 1) / resident) * (2 * t + 3);
      if (best < 0 || cost < best) { best = cost; th = t; }
    }
    if (g_conv_th >= 6 && g_conv_th <= 8) th = g_conv_th;
    // the BWD form (input gradient + ReLU mask + BatchNorm-backward sums in the epilogue) runs six rows per tile: at seven the
    // compiler spills 13 registers of its 168 (at eight, 23); 6.05 -> 6.02 ms per step (three alternating pairs, round 5)
    // (the f16x2 form of it has room for the cost model's seven or eight: 162 / 166 registers, nothing spilled)
    static const int th_bwd = -1;
    const int thb = th_bwd >= 0 ? th_bwd : (f16 ? 0 : 6);
    if (bwd && g_conv_th == 0 && thb >= 6 && thb <= 8) th = thb;
    kern = f16 ? conv_v2_kernel<true>(bn_state != nullptr, th, pre != nullptr, bwd != nullptr)
               : conv_v2_kernel<false>(bn_state != nullptr, th, pre != nullptr, bwd != nullptr);
  }
  const int lds_bytes = (f16 ? 2 : 3) * (th + 2) * CV_HW * CV_ROW + 16 + (pre ? 2 * Cin * (int)sizeof(float) : 0) +
                        (bwd ? 4 * Cout * (int)sizeof(float) : 0);
  {
    static int v2_set[2][3][2][9] = {};      // largest dynamic LDS size registered per instantiation
    int& reg = v2_set[f16 ? 1 : 0][bwd ? 2 : pre ? 1 : 0][(bn_state && !bwd) ? 1 : 0][th];
    if (reg < lds_bytes) {
      GLX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
      reg = lds_bytes;
    }
  }
  ConvArgs a;
  a.x = x; a.wp = (const uint16_t*)packed; a.y = y;
  a.wexp = reinterpret_cast<const int*>((const uint16_t*)packed + cv_pack_plane_elems(Cin, Cout, 2));
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.th = th;
  a.tiles_x = glx_divup(W, CV_TW); a.tiles_y = glx_divup(H, th);
  a.nblk = Cout / CV_BN;
  a.ntiles = a.tiles_x * a.tiles_y * B * a.nblk;
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 0;
    GLX_HIP(hipGetDevice(&dev));
    GLX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    slots = cus > 0 ? cus : 256;
  }
  a.bn_state = bn_state;
  a.bn = bn_fin;
  a.epi_scale = epi_scale;
  a.epi_shift = epi_shift;
  a.epi_relu = epi_relu;
  a.pre_scale = pre ? pre->scale : nullptr;
  a.pre_shift = pre ? pre->shift : nullptr;
  a.pre_relu = pre ? pre->relu : 0;
  a.bwd_y = bwd ? bwd->y : nullptr;
  a.bwd_coef = bwd ? bwd->coef_fwd : nullptr;
  a.bwd_mean = bwd ? bwd->mean : nullptr;
  a.bwd_invstd = bwd ? bwd->invstd : nullptr;
  const int resident = slots * conv_per_cu();
  int grid = g_conv_grid > 0 ? g_conv_grid : (a.ntiles < resident ? a.ntiles : resident);
  if (bn_state) grid = grid / a.nblk * a.nblk;   // every block keeps one channel block (ntiles is a multiple of nblk)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, a);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_conv3x3_forward(const float* x, int B, int H, int W, int Cin, const void* packed, int Cout,
                                   float* y, void* stream) {
  return glx_conv3x3_forward_ex(x, B, H, W, Cin, packed, Cout, y, nullptr, stream);
}
