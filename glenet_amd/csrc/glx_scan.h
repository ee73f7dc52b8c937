// Device-wide exclusive scan (two launches: block sums, then a write pass in which every block
// re-derives its own offset from the block sums -- cheaper than a third launch for the <= a few
// thousand blocks of the cell bitmaps, and free of cross-block synchronisation) over a
// virtual int sequence v[i] = f(i).  Used for popcount ranks of cell bitmaps and for
// first-seen voxel numbering.  wave64 shuffles inside a wave, LDS across the 4 waves.
#pragma once
#include "glx_common.h"

#define SCAN_THREADS 256
#define SCAN_IPT 8  // items per thread
#define SCAN_IPB (SCAN_THREADS * SCAN_IPT)

__device__ __forceinline__ int glx_block_exclusive_scan_256(int v, int* total) {
  __shared__ int wsum[4];
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wid) base += wsum[w];
  }
  *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return base + inc - v;
}

struct PopcWords {
  const unsigned long long* words;
  __device__ int operator()(long long i) const { return __popcll(words[i]); }
};
struct IntArray {
  const int* a;
  __device__ int operator()(long long i) const { return a[i]; }
};

// skip (optional): one byte per thread chunk of SCAN_IPT items; 0 = the whole chunk is zero, so
// its items are neither read nor (in k_scan_write) given a prefix.  Lets the popcount scans of
// the mostly empty LiDAR cell bitmaps touch only the occupied 512-cell chunks.
template <class F>
__global__ void k_scan_block_sums(F f, long long n, const unsigned char* __restrict__ skip,
                                  int* __restrict__ block_sums) {
  long long i0 = (long long)blockIdx.x * SCAN_IPB + (long long)threadIdx.x * SCAN_IPT;
  int s = 0;
  if (!skip || (i0 < n && skip[i0 / SCAN_IPT])) {
#pragma unroll
    for (int i = 0; i < SCAN_IPT; ++i)
      if (i0 + i < n) s += f(i0 + i);
  }
  int total;
  glx_block_exclusive_scan_256(s, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

template <class F>
__global__ void k_scan_write(F f, long long n, const unsigned char* __restrict__ skip,
                             const int* __restrict__ block_sums, int* __restrict__ excl,
                             int* __restrict__ n_total) {
  long long i0 = (long long)blockIdx.x * SCAN_IPB + (long long)threadIdx.x * SCAN_IPT;
  const bool last = blockIdx.x == gridDim.x - 1;
  // with skip flags (cell bitmaps) a block whose own sum is zero has no prefix anyone reads
  // (the last block still reports the total); without them every item gets its prefix
  if (skip && !last && block_sums[blockIdx.x] == 0) return;
  // offset of this block = sum of the sums of the blocks before it
  __shared__ int s_off;
  {
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_THREADS) part += block_sums[b];
    int tot;
    glx_block_exclusive_scan_256(part, &tot);
    if (threadIdx.x == 0) s_off = tot;
    __syncthreads();
  }
  int v[SCAN_IPT];
  int s = 0;
  const bool live = !skip || (i0 < n && skip[i0 / SCAN_IPT]);
#pragma unroll
  for (int i = 0; i < SCAN_IPT; ++i) {
    v[i] = (live && i0 + i < n) ? f(i0 + i) : 0;
    s += v[i];
  }
  int total;
  int ex = glx_block_exclusive_scan_256(s, &total) + s_off;
  if (last && threadIdx.x == 0) *n_total = s_off + total;
  if (!live) return;
#pragma unroll
  for (int i = 0; i < SCAN_IPT; ++i) {
    if (i0 + i < n) excl[i0 + i] = ex;
    ex += v[i];
  }
}

static inline size_t glx_scan_workspace_bytes(long long n) {
  long long nblk = (n + SCAN_IPB - 1) / SCAN_IPB;
  return glx_align((size_t)(nblk > 0 ? nblk : 1) * sizeof(int));
}

// excl[i] = sum_{i'<i} f(i'), *n_total = sum of all.  workspace >= glx_scan_workspace_bytes(n).
template <class F>
static int glx_exclusive_scan(F f, long long n, int* excl, int* n_total, void* workspace,
                              size_t workspace_bytes, hipStream_t st,
                              const unsigned char* skip = nullptr) {
  int nblk = glx_divup(n, SCAN_IPB);
  if (nblk < 1) nblk = 1;
  if (workspace_bytes < (size_t)nblk * sizeof(int)) {
    glx_set_error("scan workspace too small: %zu < %zu", workspace_bytes,
                  (size_t)nblk * sizeof(int));
    return GLX_EWORKSPACE;
  }
  int* bsum = (int*)workspace;
  hipLaunchKernelGGL((k_scan_block_sums<F>), dim3(nblk), dim3(SCAN_THREADS), 0, st, f, n, skip,
                     bsum);
  hipLaunchKernelGGL((k_scan_write<F>), dim3(nblk), dim3(SCAN_THREADS), 0, st, f, n, skip,
                     (const int*)bsum, excl, n_total);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
