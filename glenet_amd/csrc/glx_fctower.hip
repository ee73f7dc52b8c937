// The RoI head's FC towers behind the first (20 736 -> 256) Linear as ONE launch forward and ONE backward
// (pcdet/models/roi_heads/voxelrcnn_kl_label_iou_head.py:38-92 forward; voxelrcnn_head.py:40-66 the towers):
//
//   h0 = drop(relu(bn0(z0)))                         z0 = pooled features x W0^T (library split-K GEMM, outside)
//   h1 = relu(bn1(h0 W1^T))                          shared_fc_layer
//   h2 = drop(relu(bn2(h1 W2^T)))  h3 = relu(bn3(h2 W3^T))  ori_cls = h3 w_cls^T + b          cls_fc_layers, cls_pred_layer
//   h4 = drop(relu(bn4(h1 W4^T)))  h5 = relu(bn5(h4 W5^T))  rcnn_reg = h5 W_reg^T + b          reg_fc_layers, reg_pred_layer
//   rcnn_reg_std = h5 W_std^T + b ; s1 = relu(bn7(rcnn_reg_std)) ; s2 = relu(bn64(s1 W_fc1^T + b)) ; std_logit = s2 w_fc2^T + b
//
// with every BatchNorm1d in training mode (batch statistics over the R RoI rows, running statistics updated).  Through
// library GEMMs + one normalisation launch per layer this was 31 launches forward and 45 backward of 5-15 us each on the
// step's critical chain, every one of them queueing behind the BEV backward's blocks.
//
// Layout of the work: R is a few hundred rows, the layers are 256 wide.  A block owns a 16-COLUMN slab of a layer's output
// for ALL rows, so a layer's BatchNorm statistics (sums over rows) are block-local; 16 blocks per layer, the cls and reg
// towers side by side in two groups of 16 = 32 blocks, 8 waves each.  The slab's 16 x 256 weights sit in registers (64 per
// lane), the rows stream through as MFMA B operands straight from L2 (one 16-byte load per lane and k-step):
// v_mfma_f32_16x16x4_f32 with A = W slab, B = 16 rows, so that a lane ends up with FOUR CONSECUTIVE COLUMNS of one row
// (16-byte stores).  A layer's input is the previous layer's whole output, i.e. every other block's slab: between layers
// the blocks meet at a grid barrier (release / acquire at agent scope: the slabs cross XCDs).  Backward runs the same
// scheme: the BatchNorm/ReLU/dropout backward of layer L is column-local on dL/dh_L, dL/dh_{L-1} = dz_L W_L is a
// slab GEMM over W_L's INPUT columns -- which are layer L-1's output columns, so the two fuse in one block without a barrier.
// The Linear weight gradients dW_L = dz_L^T h_{L-1} stay with the caller (leaves of the backward pass, off the critical chain).
//
// `cooperative = 0` runs the same phases as one launch each (no co-residency assumption; the A/B for the barrier cost).
#include "glx_common.h"

typedef float ft4 __attribute__((ext_vector_type(4)));

#define FCT_THREADS 512
#define FCT_WAVES 8
#define FCT_W 256
#define FCT_NB 32
#define FCT_NS 7
#define FCT_NH 64
#define FCT_FWD_PHASES 5
#define FCT_BWD_PHASES 4

// ------------------------------------------------------------------------------------------------ grid barrier
// Producer: plain stores -> every wave's vmcnt(0) -> workgroup barrier -> lane 0: agent release, vmcnt(0), relaxed add.
// Consumer: relaxed poll -> agent acquire -> vmcnt(0) -> workgroup barrier -> plain loads.
__device__ __forceinline__ void fct_grid_barrier(unsigned* counter, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned polls = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++polls > GLX_SPIN_LIMIT) {          // the partner blocks are not resident: give up loudly instead of hanging
        __hip_atomic_store(counter + 24, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// after the last barrier: the block that arrives last at the exit counter zeroes both (the next launch starts from zero)
__device__ __forceinline__ void fct_grid_exit(unsigned* counter) {
  if (threadIdx.x == 0) {
    const unsigned n = __hip_atomic_fetch_add(counter + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == FCT_NB - 1) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(counter + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------------ slab pieces
// Column totals of a slab: v[i] = this lane's sum over its rows for column 4q + i; on return every lane holds the block's
// totals of its four columns.  Fixed order (lanes, then waves).  s_red: 2 x 8 x 16 floats, used alternately.
__device__ __forceinline__ void fct_colsum(float (&v)[4], float* s_red, int& flip) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) v[i] += __shfl_xor(v[i], m, 64);
  }
  float* buf = s_red + flip * (FCT_WAVES * 16);
  if (r == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) buf[wave * 16 + 4 * q + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < FCT_WAVES; ++w) s += buf[w * 16 + 4 * q + i];
    v[i] = s;
  }
  flip ^= 1;
}

// The slab GEMM.  fct_stage_w puts the slab's 16 x 256 weights in LDS as s_w[m][k] (row stride FCT_WLD): MODE 0: m = output
// feature cb + m of W (256 out, 256 in) -- the forward; MODE 1: m = INPUT feature cb + m, k = output feature (W read
// transposed) -- the input gradient; MODE 2: m = rows 0..6 of W, rows 0..6 of W2, two zero rows (the 7 + 7 outputs of
// reg_pred_layer and reg_std_layer).  fct_tile multiplies one 16-row tile of X (R, 256) by it: A = weights from LDS, B = the
// rows straight from global memory (one 16-byte load per lane and k-step, the second half of the tile in flight while the
// first multiplies); lane (q, r) gets out[row 16 tile + r][cb + 4 q + i], i = 0..3.
#define FCT_WLD (FCT_W + 4)
template <int MODE>
__device__ __forceinline__ void fct_stage_w(const float* __restrict__ W, const float* __restrict__ W2, int cb, float* s_w) {
  const int tid = threadIdx.x;
  __syncthreads();                                   // an earlier layer's readers of s_w
  if constexpr (MODE == 1) {
    for (int e0 = tid; e0 < FCT_W * 4; e0 += FCT_THREADS) {
      const int k = e0 >> 2, c4 = e0 & 3;
      const ft4 v = *reinterpret_cast<const ft4*>(W + k * FCT_W + cb + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) s_w[(4 * c4 + e) * FCT_WLD + k] = v[e];
    }
  } else {
    for (int e0 = tid; e0 < 16 * (FCT_W / 4); e0 += FCT_THREADS) {
      const int m = e0 >> 6, k4 = e0 & 63;
      ft4 v = ft4{0.f, 0.f, 0.f, 0.f};
      if constexpr (MODE == 0) v = *reinterpret_cast<const ft4*>(W + (cb + m) * FCT_W + 4 * k4);
      else if (m < 2 * FCT_NS) v = *reinterpret_cast<const ft4*>((m < FCT_NS ? W + m * FCT_W : W2 + (m - FCT_NS) * FCT_W) + 4 * k4);
      *reinterpret_cast<ft4*>(s_w + m * FCT_WLD + 4 * k4) = v;
    }
  }
  __syncthreads();
}

__device__ __forceinline__ ft4 fct_tile(const float* __restrict__ X, int tile, const float* s_w, ft4 acc) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const float* xp = X + tile * (16 * FCT_W) + r * FCT_W + 4 * q;
  const float* wl = s_w + r * FCT_WLD + 4 * q;
  ft4 xa[8], xb[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) xa[s] = *reinterpret_cast<const ft4*>(xp + 16 * s);
#pragma unroll
  for (int s = 0; s < 8; ++s) xb[s] = *reinterpret_cast<const ft4*>(xp + 128 + 16 * s);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const ft4 wv = *reinterpret_cast<const ft4*>(wl + 16 * s);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], xa[s][e], acc, 0, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const ft4 wv = *reinterpret_cast<const ft4*>(wl + 128 + 16 * s);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], xb[s][e], acc, 0, 0, 0);
  }
  return acc;
}

// The same with the first half of the NEXT tile of this wave (tile + 8) loaded behind the first half's multiplies: xa holds
// the first half of `tile` on entry (fct_tile_prime for the wave's first tile) and of tile + 8 on return.
__device__ __forceinline__ void fct_tile_prime(const float* __restrict__ X, int tile, int ntiles, ft4 (&xa)[8]) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const float* xp = X + (tile < ntiles ? tile : ntiles - 1) * (16 * FCT_W) + r * FCT_W + 4 * q;
#pragma unroll
  for (int s = 0; s < 8; ++s) xa[s] = *reinterpret_cast<const ft4*>(xp + 16 * s);
}

__device__ __forceinline__ ft4 fct_tile_pf(const float* __restrict__ X, int tile, int ntiles, const float* s_w, ft4 (&xa)[8], ft4 acc) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const float* xp = X + tile * (16 * FCT_W) + r * FCT_W + 4 * q;
  const float* wl = s_w + r * FCT_WLD + 4 * q;
  ft4 xb[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) xb[s] = *reinterpret_cast<const ft4*>(xp + 128 + 16 * s);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const ft4 wv = *reinterpret_cast<const ft4*>(wl + 16 * s);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], xa[s][e], acc, 0, 0, 0);
  }
  fct_tile_prime(X, tile + FCT_WAVES, ntiles, xa);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const ft4 wv = *reinterpret_cast<const ft4*>(wl + 128 + 16 * s);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[e], xb[s][e], acc, 0, 0, 0);
  }
  return acc;
}

// Training-mode BatchNorm1d + ReLU (+ dropout) of a slab whose Linear output produce(tile) delivers tile by tile: pass 1
// stores z and sums it, passes 2 and 3 read the block's own z back (L2) for the centred second moment and the transform.
// zbuf: where z lives afterwards (produce's values are stored there unless `stored` says they already are).
template <class Produce>
__device__ __forceinline__ void fct_bn_fwd(Produce&& produce, bool stored, int R, int cb, const glx_fc_bn& bn,
                                           const float* __restrict__ drop_u, float drop_p, float* zbuf, float* __restrict__ h_out,
                                           float* s_red, int& flip) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int ntiles = R >> 4, c0 = cb + 4 * q;
  const float invR = 1.f / (float)R;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int tile = wave; tile < ntiles; tile += FCT_WAVES) {
    const ft4 zt = produce(tile);
    if (!stored) *reinterpret_cast<ft4*>(zbuf + (tile * 16 + r) * FCT_W + c0) = zt;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += zt[i];
  }
  fct_colsum(v, s_red, flip);
  float mean[4], istd[4], var[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { mean[i] = v[i] * invR; v[i] = 0.f; }
#pragma unroll 1
  for (int tile = wave; tile < ntiles; tile += FCT_WAVES) {
    const ft4 zt = *reinterpret_cast<const ft4*>(zbuf + (tile * 16 + r) * FCT_W + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float d = zt[i] - mean[i]; v[i] += d * d; }
  }
  fct_colsum(v, s_red, flip);
  const ft4 gam = *reinterpret_cast<const ft4*>(bn.gamma + c0), bet = *reinterpret_cast<const ft4*>(bn.beta + c0);
#pragma unroll
  for (int i = 0; i < 4; ++i) { var[i] = v[i] * invR; istd[i] = 1.f / sqrtf(var[i] + bn.eps); }
  if (wave == 0 && r == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bn.save_mean[c0 + i] = mean[i];
      bn.save_invstd[c0 + i] = istd[i];
      if (bn.running_mean) {
        const float unb = R > 1 ? var[i] * ((float)R / (float)(R - 1)) : var[i];
        bn.running_mean[c0 + i] = (1.f - bn.momentum) * bn.running_mean[c0 + i] + bn.momentum * mean[i];
        bn.running_var[c0 + i] = (1.f - bn.momentum) * bn.running_var[c0 + i] + bn.momentum * unb;
      }
    }
  }
  const float keep = drop_u ? 1.f / (1.f - drop_p) : 1.f;
#pragma unroll 1
  for (int tile = wave; tile < ntiles; tile += FCT_WAVES) {
    const int off = (tile * 16 + r) * FCT_W + c0;
    const ft4 zt = *reinterpret_cast<const ft4*>(zbuf + off);
    ft4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = fmaxf(((zt[i] - mean[i]) * istd[i]) * gam[i] + bet[i], 0.f);
    if (drop_u) {
      const ft4 u = *reinterpret_cast<const ft4*>(drop_u + off);
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = u[i] >= drop_p ? y[i] * keep : 0.f;
    }
    *reinterpret_cast<ft4*>(h_out + off) = y;
  }
}

// Backward of the same for a slab: produce(tile) = dL/dh (this lane's four columns of a row) -> dz = dL/dz written, dgamma /
// dbeta.  Pass 1 parks the masked gradient in dz_out and takes the two sums, pass 2 turns it into dz.
template <class Produce>
__device__ __forceinline__ void fct_bn_bwd(Produce&& produce, int R, int cb, const glx_fc_bn& bn, const float* __restrict__ z,
                                           const float* __restrict__ drop_u, float drop_p, float* dz_out,
                                           float* __restrict__ dgamma, float* __restrict__ dbeta, float* s_red, int& flip) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  const int ntiles = R >> 4, c0 = cb + 4 * q;
  const float invR = 1.f / (float)R;
  const ft4 gam = *reinterpret_cast<const ft4*>(bn.gamma + c0), bet = *reinterpret_cast<const ft4*>(bn.beta + c0);
  const ft4 mean = *reinterpret_cast<const ft4*>(bn.save_mean + c0), istd = *reinterpret_cast<const ft4*>(bn.save_invstd + c0);
  const float keep = drop_u ? 1.f / (1.f - drop_p) : 1.f;
  float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int tile = wave; tile < ntiles; tile += FCT_WAVES) {
    const int off = (tile * 16 + r) * FCT_W + c0;
    ft4 gy = produce(tile);
    const ft4 zt = *reinterpret_cast<const ft4*>(z + off);
    ft4 u = ft4{1.f, 1.f, 1.f, 1.f};
    if (drop_u) u = *reinterpret_cast<const ft4*>(drop_u + off);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (zt[i] - mean[i]) * istd[i];
      float gv = gy[i];
      if (drop_u) gv = u[i] >= drop_p ? gv * keep : 0.f;
      if (!(xh * gam[i] + bet[i] > 0.f)) gv = 0.f;
      gy[i] = gv;
      a[i] += gv;
      b[i] += gv * xh;
    }
    *reinterpret_cast<ft4*>(dz_out + off) = gy;
  }
  fct_colsum(a, s_red, flip);
  fct_colsum(b, s_red, flip);
  if (wave == 0 && r == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { dbeta[c0 + i] = a[i]; dgamma[c0 + i] = b[i]; }
  }
#pragma unroll 1
  for (int tile = wave; tile < ntiles; tile += FCT_WAVES) {
    const int off = (tile * 16 + r) * FCT_W + c0;
    const ft4 gy = *reinterpret_cast<const ft4*>(dz_out + off);
    const ft4 zt = *reinterpret_cast<const ft4*>(z + off);
    ft4 d;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (zt[i] - mean[i]) * istd[i];
      d[i] = gam[i] * istd[i] * (gy[i] - a[i] * invR - xh * (b[i] * invR));
    }
    *reinterpret_cast<ft4*>(dz_out + off) = d;
  }
}

// D[m][n] = sum over rows A[row][acol0 + m] * B[row][bcol0 + n] (m, n < 16; columns >= avalid / bvalid read as zero), the
// rows split over the 8 waves, partial products summed through s_part (8 x 256 floats) in wave order.  Lane (q, r) returns
// D[4 q + i][r].  A / B may live in LDS or global memory.
__device__ __forceinline__ ft4 fct_outer(const float* A, int lda, int acol0, int avalid, const float* B, int ldb, int bcol0,
                                         int bvalid, int R, float* s_part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, q = lane >> 4;
  ft4 acc = ft4{0.f, 0.f, 0.f, 0.f};
  const int steps = R >> 2;
  // 8 k-steps (this wave's rows 4 ks + q) per batch: their 16 loads are in flight together
  for (int k0 = wave; k0 < steps; k0 += 8 * FCT_WAVES) {
    float av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int ks = k0 + u * FCT_WAVES;
      const int row = ks < steps ? 4 * ks + q : 0;
      av[u] = (r < avalid && ks < steps) ? A[(size_t)row * lda + acol0 + (r < avalid ? r : 0)] : 0.f;
      bv[u] = (r < bvalid && ks < steps) ? B[(size_t)row * ldb + bcol0 + (r < bvalid ? r : 0)] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  }
  __syncthreads();                                   // s_part may still be read from an earlier call
  *reinterpret_cast<ft4*>(s_part + wave * 256 + lane * 4) = acc;
  __syncthreads();
  ft4 s = ft4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < FCT_WAVES; ++w) s += *reinterpret_cast<const ft4*>(s_part + w * 256 + lane * 4);
  return s;
}

// sum over the block of n <= 16 per-thread values (fixed order); result in s_out[0..n) after the call's last barrier
template <int N>
__device__ __forceinline__ void fct_block_sum(float (&v)[N], float* s_buf /* 8 x 16 */, float* s_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v[i] += __shfl_xor(v[i], m, 64);
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) s_buf[wave * 16 + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < N) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < FCT_WAVES; ++w) s += s_buf[w * 16 + threadIdx.x];
    s_out[threadIdx.x] = s;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------ LDS of the small head
struct FctSmem {
  float red[2 * FCT_WAVES * 16];        // fct_colsum
  float part[FCT_WAVES * 256];          // fct_outer / column-thread partials (3 x 8 x 64 fit)
  float w[16 * FCT_WLD];                // fct_stage_w: the slab's weights
  float w2[16 * FCT_WLD];               // (a second slab: the shared layer's gradient sums two towers)
  float wf1[FCT_NH * 8];                // reg_std_fc1 weight, row stride 8
  float b1[FCT_NH], g64[FCT_NH], be64[FCT_NH], m64[FCT_NH], is64[FCT_NH], wf2[FCT_NH];
  float A64[FCT_NH], B64[FCT_NH];
  float g7[8], be7[8], m7[8], is7[8], A7[16];
  float sbuf[FCT_WAVES * 16];
};

__device__ __forceinline__ void fct_load_small(const glx_fc_tower& p, FctSmem& sm, bool stats) {
  const int tid = threadIdx.x;
  if (tid < FCT_NH * FCT_NS) sm.wf1[(tid / FCT_NS) * 8 + tid % FCT_NS] = p.w_fc1[tid];
  if (tid < FCT_NH) {
    sm.b1[tid] = p.b_fc1[tid];
    sm.g64[tid] = p.bn_s64.gamma[tid];
    sm.be64[tid] = p.bn_s64.beta[tid];
    sm.wf2[tid] = p.w_fc2[tid];
    if (stats) { sm.m64[tid] = p.bn_s64.save_mean[tid]; sm.is64[tid] = p.bn_s64.save_invstd[tid]; }
  }
  if (tid < FCT_NS) {
    sm.g7[tid] = p.bn_s7.gamma[tid];
    sm.be7[tid] = p.bn_s7.beta[tid];
    if (stats) { sm.m7[tid] = p.bn_s7.save_mean[tid]; sm.is7[tid] = p.bn_s7.save_invstd[tid]; }
  }
}

// layer L of the forward: h[L] = drop?(relu(bn_L(h[XL] W_L^T))) for this block's slab (constant indices: the descriptor
// stays in the kernel-argument segment)
template <int L, int XL, int DROP>
__device__ __forceinline__ void fct_fwd_layer(const glx_fc_tower& p, FctSmem& sm, int cb, int& flip) {
  const float* X = p.h[XL];
  const float* du = (DROP >= 0 && p.drop_u) ? p.drop_u + (size_t)DROP * p.R * FCT_W : nullptr;
  const ft4 zero4 = ft4{0.f, 0.f, 0.f, 0.f};
  fct_stage_w<0>(p.w[L], nullptr, cb, sm.w);
  ft4 xa[8];
  fct_tile_prime(X, threadIdx.x >> 6, p.R >> 4, xa);
  fct_bn_fwd([&](int tile) { return fct_tile_pf(X, tile, p.R >> 4, sm.w, xa, zero4); }, false, p.R, cb, p.bn[L], du, p.drop_p, p.z[L],
             p.h[L], sm.red, flip);
}

// backward through layer L's Linear into layer L - 1... : dL/dh[LB] = dz[L] W_L (+ dz[L2] W_L2), then BatchNorm / ReLU / dropout
// backward of layer LB for this block's slab
template <int L, int L2, int LB, int DROP>
__device__ __forceinline__ void fct_bwd_layer(const glx_fc_tower& p, const glx_fc_tower_grads& g, FctSmem& sm, int cb, int& flip) {
  const float* Xa = g.dz[L];
  const float* Xb = L2 >= 0 ? g.dz[L2 >= 0 ? L2 : 0] : nullptr;
  const float* du = (DROP >= 0 && p.drop_u) ? p.drop_u + (size_t)DROP * p.R * FCT_W : nullptr;
  const float* z = LB == 0 ? p.z0 : p.z[LB];
  const ft4 zero4 = ft4{0.f, 0.f, 0.f, 0.f};
  fct_stage_w<1>(p.w[L], nullptr, cb, sm.w);
  if constexpr (L2 >= 0) fct_stage_w<1>(p.w[L2 >= 0 ? L2 : 0], nullptr, cb, sm.w2);
  ft4 xa[8], xa2[L2 >= 0 ? 8 : 1];
  fct_tile_prime(Xa, threadIdx.x >> 6, p.R >> 4, xa);
  if constexpr (L2 >= 0) fct_tile_prime(Xb, threadIdx.x >> 6, p.R >> 4, xa2);
  fct_bn_bwd([&](int tile) {
    ft4 a4 = fct_tile_pf(Xa, tile, p.R >> 4, sm.w, xa, zero4);
    if constexpr (L2 >= 0) {
      __builtin_amdgcn_sched_barrier(0);             // the second product's loads stay behind the first's multiplies
      a4 = fct_tile_pf(Xb, tile, p.R >> 4, sm.w2, xa2, a4);
    }
    return a4;
  }, p.R, cb, p.bn[LB], z, du, p.drop_p, g.dz[LB], g.dgamma[LB], g.dbeta[LB], sm.red, flip);
}

// ------------------------------------------------------------------------------------------------ forward
template <int PH>
__device__ __forceinline__ void fct_fwd_phase(const glx_fc_tower& p, FctSmem& sm, float* s_dyn, int& flip) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int blk = blockIdx.x, grp = blk >> 4, cb = 16 * (blk & 15);
  const int R = p.R, ntiles = R >> 4;
  const ft4 zero4 = ft4{0.f, 0.f, 0.f, 0.f};
  (void)tid; (void)wave; (void)r; (void)q; (void)ntiles; (void)zero4; (void)grp; (void)cb;
  {
    if constexpr (PH == 0) {
      if (grp == 0) {
        float* z0 = const_cast<float*>(p.z0);
        fct_bn_fwd([&](int tile) { return *reinterpret_cast<const ft4*>(z0 + (tile * 16 + r) * FCT_W + cb + 4 * q); }, true, R, cb,
                   p.bn[0], p.drop_u, p.drop_p, z0, p.h[0], sm.red, flip);
      }
    } else if constexpr (PH == 1) {
      if (grp == 0) fct_fwd_layer<1, 0, -1>(p, sm, cb, flip);
    } else if constexpr (PH == 2) {
      if (grp == 0) fct_fwd_layer<2, 1, 1>(p, sm, cb, flip);
      else fct_fwd_layer<4, 1, 2>(p, sm, cb, flip);
    } else if constexpr (PH == 3) {
      if (grp == 0) fct_fwd_layer<3, 2, -1>(p, sm, cb, flip);
      else fct_fwd_layer<5, 4, -1>(p, sm, cb, flip);
    } else if (grp == 0) {
      // ---- (phase 4) ori_cls: a share of the rows per block, a wave per row
      const int per = (R + 15) / 16, r0 = (blk & 15) * per, r1 = min(R, r0 + per);
      const ft4 wv = *reinterpret_cast<const ft4*>(p.w_cls + 4 * lane);
      const float bias = p.b_cls[0];
      for (int row = r0 + wave; row < r1; row += FCT_WAVES) {
        const ft4 hv = *reinterpret_cast<const ft4*>(p.h[3] + row * FCT_W + 4 * lane);
        float s = hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + hv[3] * wv[3];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) p.ori_cls[row] = s + bias;
      }
    } else if (blk == 16) {
      // ---- rcnn_reg, rcnn_reg_std and the variance branch: one block (its BatchNorms need every row)
      fct_load_small(p, sm, false);
      fct_stage_w<2>(p.w_reg, p.w_std, 0, sm.w);
      float bias[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * q + i;
        bias[i] = c < FCT_NS ? p.b_reg[c] : (c < 2 * FCT_NS ? p.b_std[c - FCT_NS] : 0.f);
      }
      // the wave's (up to 8) row tiles stay in registers through the BatchNorm of the variance branch's first layer
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      ft4 vals[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int tile = wave + FCT_WAVES * t;
        vals[t] = zero4;
        if (tile < ntiles) {
          const ft4 o = fct_tile(p.h[5], tile, sm.w, zero4);
          const int row = tile * 16 + r;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = 4 * q + i;
            const float val = o[i] + bias[i];
            vals[t][i] = val;
            if (c < FCT_NS) p.rcnn_reg[row * FCT_NS + c] = val;
            else if (c < 2 * FCT_NS) { p.rcnn_reg_std[row * FCT_NS + c - FCT_NS] = val; v[i] += val; }
          }
        }
        __builtin_amdgcn_sched_barrier(0);           // one tile's 16 loads at a time
      }
      const float invR = 1.f / (float)R;
      fct_colsum(v, sm.red, flip);
      float mean[4], istd[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { mean[i] = v[i] * invR; v[i] = 0.f; }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (wave + FCT_WAVES * t < ntiles) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = 4 * q + i - FCT_NS;
            if (c >= 0 && c < FCT_NS) { const float d = vals[t][i] - mean[i]; v[i] += d * d; }
          }
        }
      }
      fct_colsum(v, sm.red, flip);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * q + i - FCT_NS;
        const float var = v[i] * invR;
        istd[i] = 1.f / sqrtf(var + p.bn_s7.eps);
        if (wave == 0 && r == 0 && c >= 0 && c < FCT_NS) {
          p.bn_s7.save_mean[c] = mean[i];
          p.bn_s7.save_invstd[c] = istd[i];
          if (p.bn_s7.running_mean) {
            const float unb = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
            p.bn_s7.running_mean[c] = (1.f - p.bn_s7.momentum) * p.bn_s7.running_mean[c] + p.bn_s7.momentum * mean[i];
            p.bn_s7.running_var[c] = (1.f - p.bn_s7.momentum) * p.bn_s7.running_var[c] + p.bn_s7.momentum * unb;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int tile = wave + FCT_WAVES * t;
        if (tile < ntiles) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = 4 * q + i - FCT_NS;
            if (c >= 0 && c < FCT_NS)
              s_dyn[(tile * 16 + r) * 8 + c] = fmaxf(((vals[t][i] - mean[i]) * istd[i]) * sm.g7[c] + sm.be7[c], 0.f);
          }
        }
      }
      __syncthreads();
      // t = s1 W_fc1^T + b (R, 64) is seven multiply-adds per element: recomputed from s1 (LDS) wherever it is needed instead
      // of making a round trip through memory.  Statistics: thread (part, j) over the rows part, part + 8, ...
      const int j = tid & 63, part = tid >> 6;
      float wj[FCT_NS];
#pragma unroll
      for (int o = 0; o < FCT_NS; ++o) wj[o] = sm.wf1[j * 8 + o];
      const float bj = sm.b1[j];
      auto tval = [&](int row) {
        const ft4 a = *reinterpret_cast<const ft4*>(s_dyn + row * 8), b = *reinterpret_cast<const ft4*>(s_dyn + row * 8 + 4);
        return bj + a[0] * wj[0] + a[1] * wj[1] + a[2] * wj[2] + a[3] * wj[3] + b[0] * wj[4] + b[1] * wj[5] + b[2] * wj[6];
      };
      float s = 0.f;
#pragma unroll 4
      for (int row = part; row < R; row += FCT_WAVES) s += tval(row);
      sm.part[part * 64 + j] = s;
      __syncthreads();
      float mean64 = 0.f;
#pragma unroll
      for (int w = 0; w < FCT_WAVES; ++w) mean64 += sm.part[w * 64 + j];
      mean64 *= invR;
      __syncthreads();
      s = 0.f;
#pragma unroll 4
      for (int row = part; row < R; row += FCT_WAVES) { const float d = tval(row) - mean64; s += d * d; }
      sm.part[part * 64 + j] = s;
      __syncthreads();
      if (tid < FCT_NH) {
        float var = 0.f;
#pragma unroll
        for (int w = 0; w < FCT_WAVES; ++w) var += sm.part[w * 64 + j];
        var *= invR;
        const float is = 1.f / sqrtf(var + p.bn_s64.eps);
        sm.m64[j] = mean64;
        sm.is64[j] = is;
        p.bn_s64.save_mean[j] = mean64;
        p.bn_s64.save_invstd[j] = is;
        if (p.bn_s64.running_mean) {
          const float unb = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
          p.bn_s64.running_mean[j] = (1.f - p.bn_s64.momentum) * p.bn_s64.running_mean[j] + p.bn_s64.momentum * mean64;
          p.bn_s64.running_var[j] = (1.f - p.bn_s64.momentum) * p.bn_s64.running_var[j] + p.bn_s64.momentum * unb;
        }
      }
      __syncthreads();
      // std_logit[row] = sum_j relu(bn64(t[row][j])) w_fc2[j] + b: a thread per row walks the 64 columns (parameters from
      // LDS, broadcast reads) -- a wave per row with a shuffle reduction was a chain of dependent cross-lane hops per row
      const float b2 = p.b_fc2[0];
      for (int row = tid; row < R; row += FCT_THREADS) {
        const ft4 a4 = *reinterpret_cast<const ft4*>(s_dyn + row * 8), b4 = *reinterpret_cast<const ft4*>(s_dyn + row * 8 + 4);
        float o = b2;
#pragma unroll 4
        for (int jx = 0; jx < FCT_NH; ++jx) {
          const ft4 w0 = *reinterpret_cast<const ft4*>(sm.wf1 + jx * 8), w1 = *reinterpret_cast<const ft4*>(sm.wf1 + jx * 8 + 4);
          const float t = sm.b1[jx] + a4[0] * w0[0] + a4[1] * w0[1] + a4[2] * w0[2] + a4[3] * w0[3] + b4[0] * w1[0] + b4[1] * w1[1] +
                          b4[2] * w1[2];
          o += fmaxf(((t - sm.m64[jx]) * sm.is64[jx]) * sm.g64[jx] + sm.be64[jx], 0.f) * sm.wf2[jx];
        }
        p.std_logit[row] = o;
      }
    }
  }
}


// PH >= 0: that phase alone (one launch per phase); PH < 0: all of them, the blocks meeting at grid barriers in between
template <int PH>
__global__ __launch_bounds__(FCT_THREADS) void k_fct_forward(glx_fc_tower p) {
  __shared__ FctSmem sm;
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];     // R x 8: s1 of the variance branch
  int flip = 0;
  if constexpr (PH >= 0) {
    fct_fwd_phase<PH>(p, sm, s_dyn, flip);
  } else {
    fct_fwd_phase<0>(p, sm, s_dyn, flip);
    fct_grid_barrier(p.barrier, 1 * FCT_NB);
    fct_fwd_phase<1>(p, sm, s_dyn, flip);
    fct_grid_barrier(p.barrier, 2 * FCT_NB);
    fct_fwd_phase<2>(p, sm, s_dyn, flip);
    fct_grid_barrier(p.barrier, 3 * FCT_NB);
    fct_fwd_phase<3>(p, sm, s_dyn, flip);
    fct_grid_barrier(p.barrier, 4 * FCT_NB);
    fct_fwd_phase<4>(p, sm, s_dyn, flip);
    fct_grid_exit(p.barrier);
  }
}

// ------------------------------------------------------------------------------------------------ backward
template <int PH>
__device__ __forceinline__ void fct_bwd_phase(const glx_fc_tower& p, const glx_fc_tower_grads& g, FctSmem& sm, float* s_g, int& flip) {
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int blk = blockIdx.x, grp = blk >> 4, cb = 16 * (blk & 15);
  const int R = p.R;
  const float invR = 1.f / (float)R;
  const ft4 zero4 = ft4{0.f, 0.f, 0.f, 0.f};
  (void)tid; (void)r; (void)q; (void)invR; (void)zero4; (void)grp; (void)cb;
  {
    if constexpr (PH == 0) {
      const bool writer = (blk & 15) == 0;
      if (grp == 0) {
        // ---- classification tower: dL/dh3 = g_cls w_cls
        for (int e = tid; e < R * 16; e += FCT_THREADS) s_g[e] = ((e & 15) == 0 && g.g_cls) ? g.g_cls[e >> 4] : 0.f;
        __syncthreads();
        const ft4 d = fct_outer(s_g, 16, 0, 1, p.h[3], FCT_W, cb, 16, R, sm.part);
        if (q == 0) g.dw_cls[cb + r] = d[0];
        if (writer) {
          float v[1] = {0.f};
          for (int row = tid; row < R; row += FCT_THREADS) v[0] += s_g[row * 16];
          fct_block_sum<1>(v, sm.sbuf, sm.A7);
          if (tid == 0) g.db_cls[0] = sm.A7[0];
        }
        const ft4 wv = *reinterpret_cast<const ft4*>(p.w_cls + cb + 4 * q);
        fct_bn_bwd([&](int tile) { return wv * s_g[(tile * 16 + r) * 16]; }, R, cb, p.bn[3], p.z[3], nullptr, 0.f, g.dz[3],
                   g.dgamma[3], g.dbeta[3], sm.red, flip);
      } else {
        // ---- variance branch backward (every block of the group computes it: no barrier in front of the regression
        // tower), then dL/dh5 = g_reg W_reg + d_std W_std
        fct_load_small(p, sm, true);
        // roles inside the group (every block computes the branch; the parameter gradients are spread so that no block
        // carries all the extra work): block 0 writes the BatchNorm / fc2 / bias gradients, blocks 1-4 one 16-row tile of
        // dW_fc1 each (block 1 also db_fc1) -- those four put dt in memory, a slice of the scratch each
        const int fc1_tile = (blk & 15) - 1;
        const bool fc1 = fc1_tile >= 0 && fc1_tile < 4;
        float* T = g.scratch + (size_t)(fc1 ? fc1_tile : 0) * R * FCT_NH;      // (R, 64): dt
        float* s_s1 = s_g + R * 16;                            // (R, 8) s1 = relu(bn7(rcnn_reg_std)), LDS
        float* s_gl = s_s1 + R * 8;                            // (R) dL/d std_logit
        __syncthreads();
        for (int row = tid; row < R; row += FCT_THREADS) {
#pragma unroll
          for (int o = 0; o < FCT_NS; ++o) {
            const float xh = (p.rcnn_reg_std[row * FCT_NS + o] - sm.m7[o]) * sm.is7[o];
            s_s1[row * 8 + o] = fmaxf(xh * sm.g7[o] + sm.be7[o], 0.f);
          }
          s_s1[row * 8 + 7] = 0.f;
          s_gl[row] = g.g_logit ? g.g_logit[row] : 0.f;
        }
        __syncthreads();
        // t = s1 W_fc1^T + b is seven multiply-adds per element: recomputed from s1 (LDS) wherever it is needed.
        // Column sums of the bn64 backward: thread (part, j) over the rows part, part + 8, ...
        const int j = tid & 63, part = tid >> 6;
        float wj[FCT_NS];
#pragma unroll
        for (int o = 0; o < FCT_NS; ++o) wj[o] = sm.wf1[j * 8 + o];
        const float bj = sm.b1[j], m64j = sm.m64[j], is64j = sm.is64[j], g64j = sm.g64[j], be64j = sm.be64[j], wf2j = sm.wf2[j];
        auto tval = [&](int row) {
          const ft4 a = *reinterpret_cast<const ft4*>(s_s1 + row * 8), b = *reinterpret_cast<const ft4*>(s_s1 + row * 8 + 4);
          return bj + a[0] * wj[0] + a[1] * wj[1] + a[2] * wj[2] + a[3] * wj[3] + b[0] * wj[4] + b[1] * wj[5] + b[2] * wj[6];
        };
        {
          float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll 4
          for (int row = part; row < R; row += FCT_WAVES) {
            const float xh = (tval(row) - m64j) * is64j;
            const float pre = xh * g64j + be64j;
            const float gl = s_gl[row];
            const float gq = pre > 0.f ? gl * wf2j : 0.f;
            a += gq;
            b += gq * xh;
            c += gl * fmaxf(pre, 0.f);
          }
          sm.part[part * 64 + j] = a;
          sm.part[512 + part * 64 + j] = b;
          sm.part[1024 + part * 64 + j] = c;
        }
        __syncthreads();
        if (tid < FCT_NH) {
          float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
          for (int w = 0; w < FCT_WAVES; ++w) { a += sm.part[w * 64 + j]; b += sm.part[512 + w * 64 + j]; c += sm.part[1024 + w * 64 + j]; }
          sm.A64[j] = a;
          sm.B64[j] = b;
          if (writer) { g.dbeta64[j] = a; g.dgamma64[j] = b; g.dw_fc2[j] = c; }
        }
        __syncthreads();
        const float A64j = sm.A64[j] * invR, B64j = sm.B64[j] * invR;
        if (fc1) {               // dt to memory (dW_fc1 contracts it with s1 below) and its column sums (db_fc1)
          float sdt = 0.f;
#pragma unroll 4
          for (int row = part; row < R; row += FCT_WAVES) {
            const float xh = (tval(row) - m64j) * is64j;
            const float gq = (xh * g64j + be64j > 0.f) ? s_gl[row] * wf2j : 0.f;
            const float dt = g64j * is64j * (gq - A64j - xh * B64j);
            T[row * FCT_NH + j] = dt;
            sdt += dt;
          }
          sm.part[part * 64 + j] = sdt;
          __syncthreads();
          if (tid < FCT_NH && fc1_tile == 0) {
            float sd = 0.f;
#pragma unroll
            for (int w = 0; w < FCT_WAVES; ++w) sd += sm.part[w * 64 + j];
            g.db_fc1[j] = sd;
          }
        }
        // a thread per row: ds1 = dt W_fc1 (dt recomputed column by column), through bn7 + ReLU (masked gradient and xhat
        // parked in s_g until the sums are known)
        float v14[14];
#pragma unroll
        for (int i = 0; i < 14; ++i) v14[i] = 0.f;
#pragma unroll 1
        for (int row = tid; row < R; row += FCT_THREADS) {
          float s1[FCT_NS], ds1[FCT_NS];
#pragma unroll
          for (int o = 0; o < FCT_NS; ++o) { s1[o] = s_s1[row * 8 + o]; ds1[o] = 0.f; }
          const float gl = s_gl[row];
#pragma unroll 2
          for (int jx = 0; jx < FCT_NH; ++jx) {
            const ft4 w0 = *reinterpret_cast<const ft4*>(sm.wf1 + jx * 8), w1 = *reinterpret_cast<const ft4*>(sm.wf1 + jx * 8 + 4);
            float t = sm.b1[jx] + s1[0] * w0[0] + s1[1] * w0[1] + s1[2] * w0[2] + s1[3] * w0[3] + s1[4] * w1[0] + s1[5] * w1[1] +
                      s1[6] * w1[2];
            const float xh = (t - sm.m64[jx]) * sm.is64[jx];
            const float gq = (xh * sm.g64[jx] + sm.be64[jx] > 0.f) ? gl * sm.wf2[jx] : 0.f;
            const float dt = sm.g64[jx] * sm.is64[jx] * (gq - sm.A64[jx] * invR - xh * (sm.B64[jx] * invR));
            ds1[0] += dt * w0[0]; ds1[1] += dt * w0[1]; ds1[2] += dt * w0[2]; ds1[3] += dt * w0[3];
            ds1[4] += dt * w1[0]; ds1[5] += dt * w1[1]; ds1[6] += dt * w1[2];
          }
#pragma unroll
          for (int o = 0; o < FCT_NS; ++o) {
            const float xh = (p.rcnn_reg_std[row * FCT_NS + o] - sm.m7[o]) * sm.is7[o];
            const float gsv = (xh * sm.g7[o] + sm.be7[o] > 0.f) ? ds1[o] : 0.f;
            s_g[row * 16 + o] = gsv;
            s_g[row * 16 + 7 + o] = xh;
            v14[o] += gsv;
            v14[7 + o] += gsv * xh;
          }
        }
        fct_block_sum<14>(v14, sm.sbuf, sm.A7);
        if (writer && tid < FCT_NS) { g.dbeta7[tid] = sm.A7[tid]; g.dgamma7[tid] = sm.A7[7 + tid]; }
#pragma unroll 1
        for (int row = tid; row < R; row += FCT_THREADS) {
#pragma unroll
          for (int o = 0; o < FCT_NS; ++o) {
            const float gsv = s_g[row * 16 + o], xh = s_g[row * 16 + 7 + o];
            const float d = sm.g7[o] * sm.is7[o] * (gsv - sm.A7[o] * invR - xh * (sm.A7[7 + o] * invR));
            s_g[row * 16 + o] = g.g_reg ? g.g_reg[row * FCT_NS + o] : 0.f;
            s_g[row * 16 + 7 + o] = d + (g.g_std ? g.g_std[row * FCT_NS + o] : 0.f);
          }
          s_g[row * 16 + 14] = 0.f;
          s_g[row * 16 + 15] = 0.f;
        }
        __syncthreads();
        if (fc1) {
          // dW_fc1[j][o] = sum_rows dt[row][j] s1[row][o]: this block's 16-row tile of j
          const int mt = fc1_tile;
          const ft4 d = fct_outer(T, FCT_NH, 16 * mt, 16, s_s1, 8, 0, FCT_NS, R, sm.part);
          if (r < FCT_NS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) g.dw_fc1[(16 * mt + 4 * q + i) * FCT_NS + r] = d[i];
          }
        }
        if (writer) {            // the prediction layers' bias gradients, db_fc2
          float v[1] = {0.f};
          for (int row = tid; row < R; row += FCT_THREADS) v[0] += g.g_logit ? g.g_logit[row] : 0.f;
          fct_block_sum<1>(v, sm.sbuf, sm.A7 + 14);
          if (tid == 0) g.db_fc2[0] = sm.A7[14];
          const int o = tid & 15, pp = tid >> 4;                    // bias gradients: 32 row partitions x 16 columns
          float s = 0.f;
          for (int row = pp; row < R; row += 32) s += s_g[row * 16 + o];
          __syncthreads();
          sm.part[pp * 16 + o] = s;
          __syncthreads();
          if (tid < 2 * FCT_NS) {
            float t2 = 0.f;
            for (int w = 0; w < 32; ++w) t2 += sm.part[w * 16 + tid];
            if (tid < FCT_NS) g.db_reg[tid] = t2; else g.db_std[tid - FCT_NS] = t2;
          }
        }
        {
          const ft4 d = fct_outer(s_g, 16, 0, 16, p.h[5], FCT_W, cb, 16, R, sm.part);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int o = 4 * q + i;
            if (o < FCT_NS) g.dw_reg[o * FCT_W + cb + r] = d[i];
            else if (o < 2 * FCT_NS) g.dw_std[(o - FCT_NS) * FCT_W + cb + r] = d[i];
          }
        }
        // dL/dh5 slab = G (R x 16) x [W_reg; W_std][:, slab]: four MFMA k-steps per row tile
        float wk[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int o = 4 * ks + q;
          wk[ks] = o < FCT_NS ? p.w_reg[o * FCT_W + cb + r] : (o < 2 * FCT_NS ? p.w_std[(o - FCT_NS) * FCT_W + cb + r] : 0.f);
        }
        fct_bn_bwd([&](int tile) {
          ft4 a4 = zero4;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(wk[ks], s_g[(tile * 16 + r) * 16 + 4 * ks + q], a4, 0, 0, 0);
          return a4;
        }, R, cb, p.bn[5], p.z[5], nullptr, 0.f, g.dz[5], g.dgamma[5], g.dbeta[5], sm.red, flip);
      }
    } else if constexpr (PH == 1) {
      if (grp == 0) fct_bwd_layer<3, -1, 2, 1>(p, g, sm, cb, flip);
      else fct_bwd_layer<5, -1, 4, 2>(p, g, sm, cb, flip);
    } else if constexpr (PH == 2) {
      if (grp == 0) fct_bwd_layer<2, 4, 1, -1>(p, g, sm, cb, flip);
    } else {
      if (grp == 0) fct_bwd_layer<1, -1, 0, 0>(p, g, sm, cb, flip);
    }
  }
}


template <int PH>
__global__ __launch_bounds__(FCT_THREADS) void k_fct_backward(glx_fc_tower p, glx_fc_tower_grads g) {
  __shared__ FctSmem sm;
  extern __shared__ __attribute__((aligned(16))) float s_dyn[];     // R x 16: the heads' output gradients per row
  int flip = 0;
  if constexpr (PH >= 0) {
    fct_bwd_phase<PH>(p, g, sm, s_dyn, flip);
  } else {
    fct_bwd_phase<0>(p, g, sm, s_dyn, flip);
    fct_grid_barrier(p.barrier, 1 * FCT_NB);
    fct_bwd_phase<1>(p, g, sm, s_dyn, flip);
    fct_grid_barrier(p.barrier, 2 * FCT_NB);
    fct_bwd_phase<2>(p, g, sm, s_dyn, flip);
    fct_grid_barrier(p.barrier, 3 * FCT_NB);
    fct_bwd_phase<3>(p, g, sm, s_dyn, flip);
    fct_grid_exit(p.barrier);
  }
}

// ------------------------------------------------------------------------------------------------ entry points
static int fct_check(const glx_fc_tower* t, const char* who) {
  GLX_REQUIRE(t, "%s: null descriptor", who);
  GLX_REQUIRE(t->R >= 16 && t->R <= 1024 && t->R % 16 == 0, "%s: R = %d rows (a multiple of 16 in 16..1024)", who, t->R);
  GLX_REQUIRE(t->drop_p >= 0.f && t->drop_p < 1.f && ((t->drop_p > 0.f) == (t->drop_u != nullptr)),
              "%s: dropout %g with%s draws", who, (double)t->drop_p, t->drop_u ? "" : "out");
  GLX_REQUIRE(t->z0 && t->barrier && t->scratch, "%s: null z0 / barrier / scratch", who);
  for (int l = 0; l < 6; ++l) {
    GLX_REQUIRE(l == 0 || (t->w[l] && t->z[l]), "%s: layer %d has no weight / output", who, l);
    GLX_REQUIRE(t->h[l] && t->bn[l].gamma && t->bn[l].beta && t->bn[l].save_mean && t->bn[l].save_invstd,
                "%s: layer %d lacks an output or BatchNorm tensor", who, l);
    GLX_REQUIRE((t->bn[l].running_mean == nullptr) == (t->bn[l].running_var == nullptr), "%s: layer %d running statistics", who, l);
  }
  GLX_REQUIRE(t->w_cls && t->b_cls && t->w_reg && t->b_reg && t->w_std && t->b_std && t->w_fc1 && t->b_fc1 && t->w_fc2 && t->b_fc2,
              "%s: a prediction layer is missing", who);
  GLX_REQUIRE(t->bn_s7.gamma && t->bn_s7.beta && t->bn_s7.save_mean && t->bn_s7.save_invstd && t->bn_s64.gamma && t->bn_s64.beta &&
              t->bn_s64.save_mean && t->bn_s64.save_invstd, "%s: a BatchNorm of the variance branch is missing", who);
  GLX_REQUIRE(t->ori_cls && t->rcnn_reg && t->rcnn_reg_std && t->std_logit, "%s: null output", who);
  return GLX_OK;
}

extern "C" size_t glx_fc_tower_scratch_bytes(int R) { return (size_t)16 * R * (FCT_NH + 8) * sizeof(float); }

static int fct_launch_fwd(const glx_fc_tower& t, hipStream_t st) {
  const size_t lds = (size_t)t.R * 8 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {          // static LDS (two weight slabs, partial sums) + the per-row array pass 64 KB beyond 512 rows
    GLX_HIP(hipFuncSetAttribute((const void*)k_fct_forward<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024));
    GLX_HIP(hipFuncSetAttribute((const void*)k_fct_forward<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024));
    attr_set = true;
  }
  // the one-launch form spins at grid barriers: only when the runtime confirms that all FCT_NB blocks are resident together
  // (a smaller partition, a CU mask or less LDS per CU: the per-phase launches below need no such guarantee)
  if (t.cooperative && glx_blocks_coresident((const void*)k_fct_forward<-1>, FCT_THREADS, lds, FCT_NB)) {
    hipLaunchKernelGGL(k_fct_forward<-1>, dim3(FCT_NB), dim3(FCT_THREADS), lds, st, t);
    GLX_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(k_fct_forward<0>, dim3(16), dim3(FCT_THREADS), (size_t)0, st, t);
    hipLaunchKernelGGL(k_fct_forward<1>, dim3(16), dim3(FCT_THREADS), (size_t)0, st, t);
    hipLaunchKernelGGL(k_fct_forward<2>, dim3(FCT_NB), dim3(FCT_THREADS), (size_t)0, st, t);
    hipLaunchKernelGGL(k_fct_forward<3>, dim3(FCT_NB), dim3(FCT_THREADS), (size_t)0, st, t);
    hipLaunchKernelGGL(k_fct_forward<4>, dim3(17), dim3(FCT_THREADS), lds, st, t);
    GLX_LAUNCH_CHECK();
  }
  return GLX_OK;
}

static int fct_launch_bwd(const glx_fc_tower& t, const glx_fc_tower_grads& g, hipStream_t st) {
  const size_t lds = (size_t)t.R * 25 * sizeof(float);          // per row: 16 output gradients, 8 s1, dL/d std_logit
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_fct_backward<-1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    GLX_HIP(hipFuncSetAttribute((const void*)k_fct_backward<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    attr_set = true;
  }
  if (t.cooperative && glx_blocks_coresident((const void*)k_fct_backward<-1>, FCT_THREADS, lds, FCT_NB)) {
    hipLaunchKernelGGL(k_fct_backward<-1>, dim3(FCT_NB), dim3(FCT_THREADS), lds, st, t, g);
    GLX_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(k_fct_backward<0>, dim3(FCT_NB), dim3(FCT_THREADS), lds, st, t, g);
    hipLaunchKernelGGL(k_fct_backward<1>, dim3(FCT_NB), dim3(FCT_THREADS), (size_t)0, st, t, g);
    hipLaunchKernelGGL(k_fct_backward<2>, dim3(16), dim3(FCT_THREADS), (size_t)0, st, t, g);
    hipLaunchKernelGGL(k_fct_backward<3>, dim3(16), dim3(FCT_THREADS), (size_t)0, st, t, g);
    GLX_LAUNCH_CHECK();
  }
  return GLX_OK;
}

// 1 when the launches of a tower of R rows can run on the current device at all (the LDS the phases ask for fits a block);
// *cooperative (may be NULL) = 1 when the one-launch forms with grid barriers will be used, 0 = one launch per phase.
extern "C" int glx_fc_tower_supported(int R, int* cooperative) {
  int dev = 0, lds_max = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess)
    return 0;
  hipFuncAttributes ff, fb;
  if (hipFuncGetAttributes(&ff, (const void*)k_fct_forward<4>) != hipSuccess) return 0;
  if (hipFuncGetAttributes(&fb, (const void*)k_fct_backward<0>) != hipSuccess) return 0;
  const size_t lf = (size_t)R * 8 * sizeof(float), lb = (size_t)R * 25 * sizeof(float);
  if (ff.sharedSizeBytes + lf > (size_t)lds_max || fb.sharedSizeBytes + lb > (size_t)lds_max) return 0;
  if (cooperative)
    *cooperative = glx_blocks_coresident((const void*)k_fct_forward<-1>, FCT_THREADS, lf, FCT_NB) &&
                   glx_blocks_coresident((const void*)k_fct_backward<-1>, FCT_THREADS, lb, FCT_NB);
  return 1;
}

// The error word of a barrier buffer (`barrier` of glx_fc_tower: 32 x u32): nonzero after a launch gave up waiting for its
// partner blocks (GLX_SPIN_LIMIT).  Host-synchronising read; clears the word.
extern "C" int glx_fc_tower_barrier_status(void* barrier, int* gave_up) {
  GLX_REQUIRE(barrier && gave_up, "glx_fc_tower_barrier_status: null pointer");
  unsigned w = 0;
  GLX_HIP(hipMemcpy(&w, (unsigned*)barrier + 24, sizeof(w), hipMemcpyDeviceToHost));
  *gave_up = (int)w;
  if (w) GLX_HIP(hipMemset((unsigned*)barrier + 24, 0, sizeof(w)));
  return GLX_OK;
}

extern "C" int glx_fc_tower_forward(const glx_fc_tower* t, void* stream) {
  int rc = fct_check(t, "glx_fc_tower_forward");
  if (rc != GLX_OK) return rc;
  return fct_launch_fwd(*t, (hipStream_t)stream);
}

extern "C" int glx_fc_tower_backward(const glx_fc_tower* t, const glx_fc_tower_grads* g, void* stream) {
  int rc = fct_check(t, "glx_fc_tower_backward");
  if (rc != GLX_OK) return rc;
  GLX_REQUIRE(g && g->scratch, "glx_fc_tower_backward: null gradients descriptor / scratch");
  for (int l = 0; l < 6; ++l)
    GLX_REQUIRE(g->dz[l] && g->dgamma[l] && g->dbeta[l], "glx_fc_tower_backward: layer %d lacks a gradient output", l);
  GLX_REQUIRE(g->dw_cls && g->db_cls && g->dw_reg && g->db_reg && g->dw_std && g->db_std && g->dgamma7 && g->dbeta7 && g->dw_fc1 &&
              g->db_fc1 && g->dgamma64 && g->dbeta64 && g->dw_fc2 && g->db_fc2, "glx_fc_tower_backward: null parameter gradient");
  return fct_launch_bwd(*t, *g, (hipStream_t)stream);
}
