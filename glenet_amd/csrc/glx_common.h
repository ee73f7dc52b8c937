// glenet_amd HIP kernels -- shared device/host helpers (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/glenet_hip.h"

#define GLX_WAVE 64

void glx_set_error(const char* fmt, ...);

#define GLX_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      glx_set_error(__VA_ARGS__);         \
      return GLX_EINVAL;                  \
    }                                     \
  } while (0)

#define GLX_HIP(expr)                                                       \
  do {                                                                      \
    hipError_t e__ = (expr);                                                \
    if (e__ != hipSuccess) {                                                \
      glx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                    __FILE__, __LINE__);                                    \
      return GLX_EHIP;                                                      \
    }                                                                       \
  } while (0)

#define GLX_LAUNCH_CHECK()                                                   \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      glx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), \
                    __FILE__, __LINE__);                                     \
      return GLX_EHIP;                                                       \
    }                                                                        \
  } while (0)

// Kernels whose blocks meet at a spin barrier in global memory are only correct when ALL blocks of the launch are resident at
// once.  This asks the runtime (per call: the answer depends on the current device and on the dynamic LDS of the call) how
// many blocks of `func` fit per CU and whether `nblocks` fit the device -- half the device, in fact: such a launch runs beside
// other streams' kernels by design, so it may only claim a fraction of the chip.  Any runtime error counts as "no".
static inline bool glx_blocks_coresident(const void* func, int threads, size_t dyn_lds, int nblocks) {
  int dev = 0, cus = 0, lds_max = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return false;
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, func) != hipSuccess) return false;
  if (fa.sharedSizeBytes + dyn_lds > (size_t)lds_max) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, func, threads, dyn_lds) != hipSuccess) { (void)hipGetLastError(); return false; }
  return per_cu >= 1 && (long long)per_cu * cus >= 2LL * nblocks;
}

// Upper bound on the polls of a spin barrier (each poll is an agent-scope load behind an s_sleep: ~1 us): a launch whose
// partner blocks never arrive gives up after a few seconds, raises the error word next to the counter and runs on (its
// results are garbage, the host sees the word) instead of hanging the GPU.
#define GLX_SPIN_LIMIT (1u << 22)

static inline int glx_divup(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t glx_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- cell <-> linear index on a (B, D, H, W) grid, x fastest -------------
struct GlxGrid {
  int B, D, H, W;
  __host__ __device__ long long cells() const { return (long long)B * D * H * W; }
  __host__ __device__ long long words() const { return (cells() + 63) >> 6; }
  // one occupancy byte per chunk of 8 bitmap words (512 cells)
  __host__ __device__ long long chunks() const { return (words() + 7) >> 3; }
  __device__ long long lin(int b, int z, int y, int x) const {
    return (((long long)b * D + z) * H + y) * W + x;
  }
};

// rank dictionary lookup: row-order independent, collision free.
// bitmap has one bit per cell, prefix[w] = number of set bits in words < w.
__device__ __forceinline__ int glx_rank_lookup(const unsigned long long* __restrict__ bitmap,
                                               const int* __restrict__ prefix, long long lin) {
  long long w = lin >> 6;
  unsigned long long word = bitmap[w];
  unsigned long long bit = 1ull << (lin & 63);
  if (!(word & bit)) return -1;
  return prefix[w] + __popcll(word & (bit - 1));
}

__device__ __forceinline__ int glx_wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
