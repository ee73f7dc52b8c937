// Shared by glx_bn.hip (k_bn_stats) and glx_sconv.hip (sparse-conv epilogue): the persistent accumulator of the
// training-mode BatchNorm statistics and its last-block finalize.  See the "statistics + finalize in one launch"
// comment in glx_bn.hip for the ordering argument (returning device-scope atomics, ticket, exchange-for-zero).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#define BN_MAXC 512
#ifndef BN_SETS
#define BN_SETS 16
#endif

struct BnState {
  double acc[BN_SETS][2 * BN_MAXC];
  unsigned ticket;
};

struct BnFinalize {
  const float* gamma; const float* beta; float eps, momentum;          // forward
  float* coef; float* save_mean; float* save_invstd; float* running_mean; float* running_var;
  const float* invstd; float* dgamma; float* dbeta;                     // backward
  long long count;      // > 0: the number of elements per channel the sums stand for (rows that are all zero are not summed:
                        // the first BEV layer as a sparse convolution, whose BatchNorm2d counts every pixel); 0: the rows
};

// y = x * scale + shift, written ONE way everywhere: the backward kernels re-derive the ReLU mask (y > 0) from x
// instead of reading y back (a third of their traffic), which only works if they round exactly like the forward.
__device__ __forceinline__ float bn_scale(float invstd, float gamma) { return __fmul_rn(invstd, gamma); }
__device__ __forceinline__ float bn_shift(float beta, float mean, float invstd, float gamma) {
  return __fsub_rn(beta, __fmul_rn(__fmul_rn(mean, invstd), gamma));
}
__device__ __forceinline__ float bn_affine(float x, float scale, float shift) { return __fmaf_rn(x, scale, shift); }

__device__ __forceinline__ double bn_take(double* p) {                  // read and clear
  return __longlong_as_double((long long)__hip_atomic_exchange(reinterpret_cast<unsigned long long*>(p), 0ull,
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Threads t < C / 4 of a block hold the block's sums (a0 = first moment, a1 = second) of float4 column t: add them to
// the block's accumulator set, take a ticket; returns true in EVERY thread of the block that drew the last one.
// `s_last` = one int of shared memory.  All threads of the block must call it.
// `coff`: the first of the block's C channels (a launch whose blocks own different channel ranges of one BatchNorm).
__device__ __forceinline__ bool bn_contribute(BnState* st, int C, const double (&a0)[4], const double (&a1)[4],
                                              unsigned nblocks, int* s_last, int coff = 0) {
  if ((int)threadIdx.x < (C >> 2)) {
    double* acc = st->acc[blockIdx.x % BN_SETS] + coff;
    double seen = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      seen += unsafeAtomicAdd(acc + 4 * threadIdx.x + i, a0[i]);
      seen += unsafeAtomicAdd(acc + BN_MAXC + 4 * threadIdx.x + i, a1[i]);
    }
    asm volatile("" ::"v"(seen) : "memory");                           // the atomics have returned: they are done
  }
  __syncthreads();
  if (threadIdx.x == 0)
    *s_last = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1;
  __syncthreads();
  return *s_last != 0;
}

// The last block: sum the sets (exchanging them for zero), finalize.  s_fin: THREADS x 2 doubles of shared memory.
template <bool BWD, int THREADS>
__device__ __forceinline__ void bn_finalize_sets(BnState* st, const BnFinalize& f, int C, int N, double (*s_fin)[2]) {
  if (f.count > 0) N = (int)(f.count < 2147483647LL ? f.count : 2147483647LL);
  const double cnt = N > 0 ? (double)N : 1.0;
  const int CB = C < THREADS ? C : THREADS, G = THREADS / CB;
  for (int cb = 0; cb < C; cb += CB) {
    const int c = cb + threadIdx.x % CB, g = threadIdx.x / CB;
    double s = 0, ss = 0;
    if (g < G)
      for (int k = g; k < BN_SETS; k += G) { s += bn_take(&st->acc[k][c]); ss += bn_take(&st->acc[k][BN_MAXC + c]); }
    __syncthreads();
    s_fin[threadIdx.x][0] = s;
    s_fin[threadIdx.x][1] = ss;
    __syncthreads();
    if (g != 0) continue;
    for (int k = 1; k < G && k < BN_SETS; ++k) { s += s_fin[k * CB + c - cb][0]; ss += s_fin[k * CB + c - cb][1]; }
    if (!BWD) {
      const double m = s / cnt;
      double var = ss / cnt - m * m;
      if (var < 0) var = 0;
      const float is = (float)(1.0 / sqrt(var + (double)f.eps));
      const float gm = f.gamma ? f.gamma[c] : 1.f, bt = f.beta ? f.beta[c] : 0.f;
      f.coef[c] = bn_scale(is, gm);
      f.coef[C + c] = bn_shift(bt, (float)m, is, gm);
      f.save_mean[c] = (float)m;
      f.save_invstd[c] = is;
      if (f.running_mean) {
        const double unb = N > 1 ? var * cnt / (cnt - 1.0) : var;
        f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)m;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unb;
      }
    } else {
      f.coef[c] = (f.gamma ? f.gamma[c] : 1.f) * f.invstd[c];
      f.coef[C + c] = (float)(s / cnt);
      f.coef[2 * C + c] = (float)(ss / cnt);
      if (f.dgamma) f.dgamma[c] = (float)ss;
      if (f.dbeta) f.dbeta[c] = (float)s;
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
