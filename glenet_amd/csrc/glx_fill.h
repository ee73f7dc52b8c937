// One launch that byte-fills up to GLX_FILL_MAX independent device regions (bitmaps, flag
// arrays, sentinel tables).  Replaces runs of small hipMemsetAsync calls: one dispatch instead
// of one per buffer, and a plain kernel node when the stream is being captured into a HIP graph.
#pragma once
#include "glx_common.h"

#define GLX_FILL_MAX 8
#define GLX_FILL_BLOCK_BYTES 16384   // 256 threads x 4 x 16 B

struct GlxFillJob {
  void* ptr;
  size_t bytes;
  unsigned char value;
};

struct GlxFillArgs {
  unsigned long long ptr[GLX_FILL_MAX];
  unsigned long long bytes[GLX_FILL_MAX];
  unsigned first_block[GLX_FILL_MAX + 1];
  unsigned pattern[GLX_FILL_MAX];
  int n;
};

static __global__ void k_fill_multi(GlxFillArgs a) {
  int r = 0;
#pragma unroll
  for (int i = 1; i < GLX_FILL_MAX; ++i)
    if (i < a.n && blockIdx.x >= a.first_block[i]) r = i;
  unsigned char* base = reinterpret_cast<unsigned char*>(a.ptr[r]);
  const unsigned long long bytes = a.bytes[r];
  const unsigned pat = a.pattern[r];
  const bool aligned = (a.ptr[r] & 15ull) == 0;
  unsigned long long off0 = (unsigned long long)(blockIdx.x - a.first_block[r]) * GLX_FILL_BLOCK_BYTES;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned long long off = off0 + (unsigned long long)(i * 256 + threadIdx.x) * 16;
    if (off >= bytes) break;
    if (aligned && off + 16 <= bytes) {
      *reinterpret_cast<uint4*>(base + off) = make_uint4(pat, pat, pat, pat);
    } else {
      for (unsigned long long b = off; b < off + 16 && b < bytes; ++b) base[b] = (unsigned char)pat;
    }
  }
}

static inline int glx_fill_multi(const GlxFillJob* jobs, int n, hipStream_t st) {
  GlxFillArgs a;
  memset(&a, 0, sizeof(a));
  unsigned blocks = 0;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (!jobs[i].ptr || jobs[i].bytes == 0) continue;
    if (m == GLX_FILL_MAX) {
      glx_set_error("glx_fill_multi: more than %d regions", GLX_FILL_MAX);
      return GLX_EINVAL;
    }
    a.ptr[m] = (unsigned long long)jobs[i].ptr;
    a.bytes[m] = jobs[i].bytes;
    a.pattern[m] = 0x01010101u * jobs[i].value;
    a.first_block[m] = blocks;
    blocks += (unsigned)((jobs[i].bytes + GLX_FILL_BLOCK_BYTES - 1) / GLX_FILL_BLOCK_BYTES);
    ++m;
  }
  a.n = m;
  a.first_block[m] = blocks;
  if (blocks == 0) return GLX_OK;
  hipLaunchKernelGGL(k_fill_multi, dim3(blocks), dim3(256), 0, st, a);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
