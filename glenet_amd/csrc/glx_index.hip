// Cell index (bitmap + popcount rank) and rule-table generation for sparse convolution.
//
// Semantics follow spconv's indice-pair generation as used by the reference at
// pcdet/models/backbones_3d/spconv_backbone.py:77-117 (SubMConv3d keeps the input
// set; SparseConv3d reaches every output cell touched by an active input); the data
// structure is ours: a succinct rank dictionary instead of a hash table, so lookups
// are collision free, deterministic and the active set enumerates in sorted order.
#include <stdarg.h>

#include "glx_common.h"
#include "glx_fill.h"
#include "glx_scan.h"

// ---------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
void glx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* glx_last_error(void) { return g_err; }
extern "C" int glx_abi_version(void) { return 7; }

// ---------------------------------------------------------------- timing events (bench)
extern "C" int glx_event_create(void** event) {
  GLX_REQUIRE(event, "glx_event_create: null");
  hipEvent_t e;
  GLX_HIP(hipEventCreate(&e));
  *event = (void*)e;
  return GLX_OK;
}
extern "C" int glx_event_destroy(void* event) {
  GLX_HIP(hipEventDestroy((hipEvent_t)event));
  return GLX_OK;
}
// blocks the host until `stop` has completed
extern "C" int glx_event_elapsed_ms(void* start, void* stop, float* ms) {
  GLX_REQUIRE(start && stop && ms, "glx_event_elapsed_ms: null");
  GLX_HIP(hipEventSynchronize((hipEvent_t)stop));
  GLX_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return GLX_OK;
}

// ---------------------------------------------------------------- bitmap + scan
__global__ void k_set_bits(const int4* __restrict__ idx, int N, GlxGrid g,
                           unsigned long long* __restrict__ bitmap,
                           unsigned char* __restrict__ chunk_flags, int* __restrict__ status) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int4 c = idx[i];  // b, z, y, x
  if ((unsigned)c.x >= (unsigned)g.B || (unsigned)c.y >= (unsigned)g.D ||
      (unsigned)c.z >= (unsigned)g.H || (unsigned)c.w >= (unsigned)g.W) {
    *status = 1;
    return;
  }
  long long l = g.lin(c.x, c.y, c.z, c.w);
  atomicOr(&bitmap[l >> 6], 1ull << (l & 63));
  chunk_flags[l >> 9] = 1;
}

__global__ void k_scatter_perm(const int4* __restrict__ idx, int N, GlxGrid g,
                               const unsigned long long* __restrict__ bitmap,
                               const int* __restrict__ prefix, int* __restrict__ rank_to_row,
                               int* __restrict__ row_to_rank) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int4 c = idx[i];
  if ((unsigned)c.x >= (unsigned)g.B || (unsigned)c.y >= (unsigned)g.D ||
      (unsigned)c.z >= (unsigned)g.H || (unsigned)c.w >= (unsigned)g.W) {
    if (row_to_rank) row_to_rank[i] = -1;
    return;
  }
  int r = glx_rank_lookup(bitmap, prefix, g.lin(c.x, c.y, c.z, c.w));
  if (r >= 0 && r < N) rank_to_row[r] = i;
  if (row_to_rank) row_to_rank[i] = r;
}

extern "C" int64_t glx_index_words(int B, int D, int H, int W) {
  GlxGrid g{B, D, H, W};
  return g.words();
}

// workspace: block sums of the popcount scan
extern "C" size_t glx_index_workspace_bytes(int B, int D, int H, int W) {
  GlxGrid g{B, D, H, W};
  return glx_scan_workspace_bytes(g.words()) + 256;
}

int glx_scan_bitmap(const GlxGrid& g, uint64_t* bitmap, const uint8_t* chunk_flags,
                    int32_t* prefix, int32_t* n_total, void* workspace, size_t workspace_bytes,
                    hipStream_t st) {
  PopcWords f{(const unsigned long long*)bitmap};
  return glx_exclusive_scan(f, g.words(), prefix, n_total, workspace, workspace_bytes, st,
                            chunk_flags);
}

extern "C" int glx_index_build(const int32_t* indices, int N, int B, int D, int H, int W,
                               uint64_t* bitmap, uint8_t* chunk_flags, int32_t* prefix,
                               int32_t* rank_to_row, int32_t* row_to_rank, int32_t* n_unique,
                               int32_t* status, void* workspace, size_t workspace_bytes,
                               void* stream) {
  GLX_REQUIRE(N >= 0 && B > 0 && D > 0 && H > 0 && W > 0, "glx_index_build: bad sizes");
  GLX_REQUIRE(bitmap && chunk_flags && prefix && n_unique && status, "glx_index_build: null output");
  hipStream_t st = (hipStream_t)stream;
  GlxGrid g{B, D, H, W};
  {
    GlxFillJob jobs[3] = {{bitmap, (size_t)g.words() * 8, 0}, {chunk_flags, (size_t)g.chunks(), 0},
                          {status, sizeof(int), 0}};
    int rc = glx_fill_multi(jobs, 3, st);
    if (rc != GLX_OK) return rc;
  }
  if (N > 0) {
    hipLaunchKernelGGL(k_set_bits, dim3(glx_divup(N, 256)), dim3(256), 0, st,
                       (const int4*)indices, N, g, (unsigned long long*)bitmap, chunk_flags,
                       status);
  }
  int rc = glx_scan_bitmap(g, bitmap, chunk_flags, prefix, n_unique, workspace, workspace_bytes, st);
  if (rc != GLX_OK) return rc;
  if (N > 0 && rank_to_row) {
    hipLaunchKernelGGL(k_scatter_perm, dim3(glx_divup(N, 256)), dim3(256), 0, st,
                       (const int4*)indices, N, g, (const unsigned long long*)bitmap,
                       (const int*)prefix, rank_to_row, row_to_rank);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ---------------------------------------------------------------- SubM rules
// one thread per (output row j, kz, ky): probes the kw cells of one x-row of the window.
__global__ void k_rules_subm(const int4* __restrict__ idx, int N, GlxGrid g,
                             const unsigned long long* __restrict__ bitmap,
                             const int* __restrict__ prefix, const int* __restrict__ rank_to_row,
                             int kd, int kh, int kw, int dd, int dh, int dw, int* __restrict__ nbr,
                             int* __restrict__ pair_count, const int* __restrict__ n_live) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int zy = kd * kh;
  int hits = 0;
  if (n_live) N = min(N, *n_live);
  int s = (int)(t / zy);   // walk rows in cell order: neighbouring threads probe the same words
  int j = -1;
  if (t < (long long)N * zy) j = rank_to_row ? rank_to_row[s] : s;
  if (j >= 0 && j < N) {   // (a rank without a row only exists when max_voxels dropped cells)
    int r = (int)(t - (long long)s * zy);
    int kz = r / kh, ky = r - kz * kh;
    int4 c = idx[j];
    int z = c.y + (kz - kd / 2) * dd, y = c.z + (ky - kh / 2) * dh;
    int* dst = nbr + (long long)j * (zy * kw) + (long long)r * kw;
    bool row_ok = (unsigned)z < (unsigned)g.D && (unsigned)y < (unsigned)g.H;
    for (int kx = 0; kx < kw; ++kx) {
      int x = c.w + (kx - kw / 2) * dw;
      int v = -1;
      if (row_ok && (unsigned)x < (unsigned)g.W) {
        int rk = glx_rank_lookup(bitmap, prefix, g.lin(c.x, z, y, x));
        if (rk >= 0) {
          v = rank_to_row ? rank_to_row[rk] : rk;
          if (v >= N) v = -1;   // beyond the capacity of a shape-static set
          hits += v >= 0;
        }
      }
      dst[kx] = v;
    }
  }
  if (pair_count) {  // optional: a same-address atomic per wave serialises (~12 ns each)
    hits = glx_wave_sum(hits);
    if ((threadIdx.x & 63) == 0 && hits) atomicAdd(pair_count, hits);
  }
}

extern "C" int glx_rules_subm(const int32_t* indices, int N, int B, int D, int H, int W,
                              const uint64_t* bitmap, const int32_t* prefix,
                              const int32_t* rank_to_row, int kd, int kh, int kw, int32_t* nbr,
                              int32_t* pair_count, const int32_t* n_live, void* stream) {
  return glx_rules_subm_dilated(indices, N, B, D, H, W, bitmap, prefix, rank_to_row, kd, kh, kw, 1, 1, 1, nbr, pair_count,
                                n_live, stream);
}

extern "C" int glx_rules_subm_dilated(const int32_t* indices, int N, int B, int D, int H, int W,
                                      const uint64_t* bitmap, const int32_t* prefix,
                                      const int32_t* rank_to_row, int kd, int kh, int kw, int dd, int dh, int dw,
                                      int32_t* nbr, int32_t* pair_count, const int32_t* n_live, void* stream) {
  GLX_REQUIRE(kd > 0 && kh > 0 && kw > 0 && (kd & 1) && (kh & 1) && (kw & 1),
              "glx_rules_subm: kernel size must be odd, got (%d,%d,%d)", kd, kh, kw);
  GLX_REQUIRE(dd > 0 && dh > 0 && dw > 0, "glx_rules_subm: dilation must be positive, got (%d,%d,%d)", dd, dh, dw);
  if (N == 0) return GLX_OK;
  GLX_REQUIRE(indices && bitmap && prefix && nbr, "glx_rules_subm: null pointer");
  GlxGrid g{B, D, H, W};
  long long total = (long long)N * kd * kh;
  hipLaunchKernelGGL(k_rules_subm, dim3(glx_divup(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const int4*)indices, N, g,
                     (const unsigned long long*)bitmap, (const int*)prefix, rank_to_row, kd, kh,
                     kw, dd, dh, dw, nbr, pair_count, n_live);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ---------------------------------------------------------------- strided conv
struct ConvGeom {
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int dd = 1, dh = 1, dw = 1;     // dilation (spconv's SparseConv3d(dilation=...): input cell = o * s - p + k * d)
};

// one thread per input row: the outputs that reach input cell c along one axis are
// o in [ceil((c + p - (k-1)) / s), floor((c + p) / s)] clipped to the grid -- at most
// ceil(k/s) per axis (2 x 2 x 2 for the 3x3x3 stride-2 convs), found with six divisions instead
// of 27 modulo tests.  Rows are walked in cell order (in_rank_to_row) so neighbouring threads hit
// the same output words.
__device__ __forceinline__ void outset_axis_range(int c, int k, int s, int p, int O, int& lo, int& hi) {
  int a = c + p - (k - 1);
  lo = a <= 0 ? 0 : (a + s - 1) / s;
  hi = (c + p) / s;        // c + p >= 0
  if (hi > O - 1) hi = O - 1;
}

__global__ void k_outset_mark(const int4* __restrict__ idx, const int* __restrict__ in_rank_to_row,
                              int N, ConvGeom cg, GlxGrid og,
                              unsigned long long* __restrict__ obitmap,
                              unsigned char* __restrict__ oflags, const int* __restrict__ n_live) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (n_live) N = min(N, *n_live);
  if (s >= N) return;
  int i = in_rank_to_row ? in_rank_to_row[s] : s;
  if (i < 0 || i >= N) return;
  int4 c = idx[i];
  int z0, z1, y0, y1, x0, x1;
  outset_axis_range(c.y, cg.kd, cg.sd, cg.pd, og.D, z0, z1);
  outset_axis_range(c.z, cg.kh, cg.sh, cg.ph, og.H, y0, y1);
  outset_axis_range(c.w, cg.kw, cg.sw, cg.pw, og.W, x0, x1);
  for (int oz = z0; oz <= z1; ++oz)
    for (int oy = y0; oy <= y1; ++oy)
      for (int ox = x0; ox <= x1; ++ox) {
        long long l = og.lin(c.x, oz, oy, ox);
        unsigned long long bit = 1ull << (l & 63);
        // cheap pre-test avoids most redundant atomics (each output is reached several times;
        // same-address atomics serialise in L2)
        if (!(obitmap[l >> 6] & bit)) {
          atomicOr(&obitmap[l >> 6], bit);
          oflags[l >> 9] = 1;
        }
      }
}

// Dilated geometry (any dilation != 1): the outputs an input cell reaches are not a contiguous range -- one test per tap:
// o = (c + p - k d) / s where that is a non-negative multiple of s inside the output grid.
__global__ void k_outset_mark_dilated(const int4* __restrict__ idx, const int* __restrict__ in_rank_to_row,
                                      int N, ConvGeom cg, GlxGrid og,
                                      unsigned long long* __restrict__ obitmap,
                                      unsigned char* __restrict__ oflags, const int* __restrict__ n_live) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (n_live) N = min(N, *n_live);
  if (s >= N) return;
  int i = in_rank_to_row ? in_rank_to_row[s] : s;
  if (i < 0 || i >= N) return;
  int4 c = idx[i];
  for (int kz = 0; kz < cg.kd; ++kz) {
    const int az = c.y + cg.pd - kz * cg.dd;
    if (az < 0 || az % cg.sd || az / cg.sd >= og.D) continue;
    for (int ky = 0; ky < cg.kh; ++ky) {
      const int ay = c.z + cg.ph - ky * cg.dh;
      if (ay < 0 || ay % cg.sh || ay / cg.sh >= og.H) continue;
      for (int kx = 0; kx < cg.kw; ++kx) {
        const int ax = c.w + cg.pw - kx * cg.dw;
        if (ax < 0 || ax % cg.sw || ax / cg.sw >= og.W) continue;
        long long l = og.lin(c.x, az / cg.sd, ay / cg.sh, ax / cg.sw);
        unsigned long long bit = 1ull << (l & 63);
        if (!(obitmap[l >> 6] & bit)) {
          atomicOr(&obitmap[l >> 6], bit);
          oflags[l >> 9] = 1;
        }
      }
    }
  }
}

// Variant for the common geometry where every axis range has at most 2 cells: the lanes of a
// wave walk the input rows in cell order, so neighbouring lanes target the same output word.
// Their bits are OR-ed across the wave first (segmented by word: equal words are contiguous in
// lane order) and only the first lane of each segment issues the atomic.
__global__ void k_outset_mark_agg(const int4* __restrict__ idx,
                                  const int* __restrict__ in_rank_to_row, int N, ConvGeom cg,
                                  GlxGrid og, unsigned long long* __restrict__ obitmap,
                                  unsigned char* __restrict__ oflags,
                                  const int* __restrict__ n_live) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (n_live) N = min(N, *n_live);
  int i = -1;
  if (s < N) i = in_rank_to_row ? in_rank_to_row[s] : s;
  const bool act = i >= 0 && i < N;
  int4 c = act ? idx[i] : make_int4(0, 0, 0, 0);
  int z0, z1, y0, y1, x0, x1;
  outset_axis_range(c.y, cg.kd, cg.sd, cg.pd, og.D, z0, z1);
  outset_axis_range(c.z, cg.kh, cg.sh, cg.ph, og.H, y0, y1);
  outset_axis_range(c.w, cg.kw, cg.sw, cg.pw, og.W, x0, x1);
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int dz = 0; dz < 2; ++dz) {
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int oz = z0 + dz, oy = y0 + dy;
      const bool on = act && oz <= z1 && oy <= y1 && x0 <= x1;
      // up to two cells x0, x0+1 of one output x-row: one or two words
      long long l0 = on ? og.lin(c.x, oz, oy, x0) : -1;
      long long w = on ? (l0 >> 6) : -1 - lane;          // inactive lanes never match a neighbour
      unsigned long long m = 0, m2 = 0;
      if (on) {
        m = 1ull << (l0 & 63);
        if (x1 > x0) {
          if (((l0 + 1) >> 6) == w) m |= 1ull << ((l0 + 1) & 63);
          else m2 = 1ull;                                // spills into bit 0 of the next word
        }
      }
      if (m2) {   // rare: handle the spill directly
        if (!(obitmap[w + 1] & 1ull)) { atomicOr(&obitmap[w + 1], 1ull); oflags[(w + 1) >> 3] = 1; }
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        long long wn = __shfl_down(w, o, 64);
        unsigned long long mn = __shfl_down(m, o, 64);
        if (lane + o < 64 && wn == w) m |= mn;
      }
      long long wp = __shfl_up(w, 1, 64);
      const bool head = on && (lane == 0 || wp != w);
      if (head && (obitmap[w] & m) != m) {
        atomicOr(&obitmap[w], m);
        oflags[w >> 3] = 1;
      }
    }
  }
}

extern "C" int glx_outset_build(const int32_t* indices_in, int N_in, int B, int D, int H, int W,
                                const int32_t* in_rank_to_row, int kd, int kh, int kw, int sd,
                                int sh, int sw, int pd, int ph, int pw, int oD, int oH, int oW,
                                uint64_t* out_bitmap, uint8_t* out_chunk_flags,
                                int32_t* out_prefix, int32_t* n_out, const int32_t* n_in_live,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return glx_outset_build_dilated(indices_in, N_in, B, D, H, W, in_rank_to_row, kd, kh, kw, sd, sh, sw, pd, ph, pw, 1, 1, 1,
                                  oD, oH, oW, out_bitmap, out_chunk_flags, out_prefix, n_out, n_in_live, workspace,
                                  workspace_bytes, stream);
}

extern "C" int glx_outset_build_dilated(const int32_t* indices_in, int N_in, int B, int D, int H, int W,
                                        const int32_t* in_rank_to_row, int kd, int kh, int kw, int sd,
                                        int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw,
                                        int oD, int oH, int oW, uint64_t* out_bitmap, uint8_t* out_chunk_flags,
                                        int32_t* out_prefix, int32_t* n_out, const int32_t* n_in_live,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(kd > 0 && kh > 0 && kw > 0 && sd > 0 && sh > 0 && sw > 0 && dd > 0 && dh > 0 && dw > 0,
              "glx_outset_build: bad geometry");
  GLX_REQUIRE(oD > 0 && oH > 0 && oW > 0, "glx_outset_build: empty output grid");
  (void)D; (void)H; (void)W;
  hipStream_t st = (hipStream_t)stream;
  GlxGrid og{B, oD, oH, oW};
  GLX_REQUIRE(out_bitmap && out_chunk_flags && out_prefix && n_out, "glx_outset_build: null output");
  {
    GlxFillJob jobs[2] = {{out_bitmap, (size_t)og.words() * 8, 0},
                          {out_chunk_flags, (size_t)og.chunks(), 0}};
    int rc = glx_fill_multi(jobs, 2, st);
    if (rc != GLX_OK) return rc;
  }
  if (N_in > 0) {
    ConvGeom cg{kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw};
    auto reach = [](int k, int s) { return (k + s - 1) / s; };
    if (dd != 1 || dh != 1 || dw != 1) {
      hipLaunchKernelGGL(k_outset_mark_dilated, dim3(glx_divup(N_in, 256)), dim3(256), 0, st,
                         (const int4*)indices_in, in_rank_to_row, N_in, cg, og,
                         (unsigned long long*)out_bitmap, out_chunk_flags, n_in_live);
    } else if (reach(kd, sd) <= 2 && reach(kh, sh) <= 2 && reach(kw, sw) <= 2) {
      hipLaunchKernelGGL(k_outset_mark_agg, dim3(glx_divup(N_in, 256)), dim3(256), 0, st,
                         (const int4*)indices_in, in_rank_to_row, N_in, cg, og,
                         (unsigned long long*)out_bitmap, out_chunk_flags, n_in_live);
    } else {
      hipLaunchKernelGGL(k_outset_mark, dim3(glx_divup(N_in, 256)), dim3(256), 0, st,
                         (const int4*)indices_in, in_rank_to_row, N_in, cg, og,
                         (unsigned long long*)out_bitmap, out_chunk_flags, n_in_live);
    }
  }
  return glx_scan_bitmap(og, out_bitmap, out_chunk_flags, out_prefix, n_out, workspace,
                         workspace_bytes, st);
}

// one thread per bitmap word of an occupied chunk: decode the word's base cell once, then walk
// its set bits carrying x / y / z / b like an odometer.  Rows >= capacity are not written.
template <class IDX>
__global__ void k_outset_emit(const unsigned long long* __restrict__ bitmap,
                              const unsigned char* __restrict__ chunk_flags,
                              const int* __restrict__ prefix, long long nwords, GlxGrid g,
                              int capacity, int4* __restrict__ out) {
  long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords || !chunk_flags[w >> 3]) return;
  unsigned long long word = bitmap[w];
  if (!word) return;
  int base = prefix[w];
  IDX l0 = (IDX)(w << 6);
  int x = (int)(l0 % (IDX)g.W);
  IDX q = l0 / (IDX)g.W;
  int y = (int)(q % (IDX)g.H);
  q /= (IDX)g.H;
  int z = (int)(q % (IDX)g.D);
  int b = (int)(q / (IDX)g.D);
  int done = 0;   // bits of this word already stepped over
  while (word) {
    int bit = __ffsll((long long)word) - 1;
    word &= word - 1;
    x += bit - done;
    done = bit;
    while (x >= g.W) { x -= g.W; if (++y == g.H) { y = 0; if (++z == g.D) { z = 0; ++b; } } }
    if (base < capacity) out[base] = make_int4(b, z, y, x);
    ++base;
  }
}

extern "C" int glx_outset_emit(const uint64_t* bitmap, const uint8_t* chunk_flags,
                               const int32_t* prefix, int B, int D, int H, int W, int capacity,
                               int32_t* indices_out, void* stream) {
  GlxGrid g{B, D, H, W};
  long long nwords = g.words();
  if (capacity <= 0) return GLX_OK;
  GLX_REQUIRE(bitmap && chunk_flags && prefix && indices_out, "glx_outset_emit: null pointer");
  if (g.cells() + 64 < (1ll << 32)) {   // 32-bit divisions are several times cheaper
    hipLaunchKernelGGL((k_outset_emit<unsigned>), dim3(glx_divup(nwords, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned long long*)bitmap, chunk_flags,
                       (const int*)prefix, nwords, g, capacity, (int4*)indices_out);
  } else {
    hipLaunchKernelGGL((k_outset_emit<long long>), dim3(glx_divup(nwords, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned long long*)bitmap, chunk_flags,
                       (const int*)prefix, nwords, g, capacity, (int4*)indices_out);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

__global__ void k_rules_strided(const int4* __restrict__ oidx, int N_out, int N_in, GlxGrid ig,
                                const unsigned long long* __restrict__ ibitmap,
                                const int* __restrict__ iprefix,
                                const int* __restrict__ irank_to_row, ConvGeom cg,
                                int* __restrict__ nbr, int* __restrict__ pair_count,
                                const int* __restrict__ n_live) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int zy = cg.kd * cg.kh;
  int hits = 0;
  if (n_live) N_out = min(N_out, *n_live);
  if (t < (long long)N_out * zy) {
    int j = (int)(t / zy);
    int r = (int)(t - (long long)j * zy);
    int kz = r / cg.kh, ky = r - kz * cg.kh;
    int4 c = oidx[j];
    int z = c.y * cg.sd - cg.pd + kz * cg.dd, y = c.z * cg.sh - cg.ph + ky * cg.dh;
    bool row_ok = (unsigned)z < (unsigned)ig.D && (unsigned)y < (unsigned)ig.H;
    int* dst = nbr + (long long)j * (zy * cg.kw) + (long long)r * cg.kw;
    for (int kx = 0; kx < cg.kw; ++kx) {
      int x = c.w * cg.sw - cg.pw + kx * cg.dw;
      int v = -1;
      if (row_ok && (unsigned)x < (unsigned)ig.W) {
        int rk = glx_rank_lookup(ibitmap, iprefix, ig.lin(c.x, z, y, x));
        if (rk >= 0) {
          v = irank_to_row ? irank_to_row[rk] : rk;
          if (v >= N_in) v = -1;   // beyond the capacity of a shape-static input set
          hits += v >= 0;
        }
      }
      dst[kx] = v;
    }
  }
  if (pair_count) {  // optional: a same-address atomic per wave serialises (~12 ns each)
    hits = glx_wave_sum(hits);
    if ((threadIdx.x & 63) == 0 && hits) atomicAdd(pair_count, hits);
  }
}

extern "C" int glx_rules_strided(const int32_t* indices_out, int N_out, int N_in, int B, int D,
                                 int H, int W, const uint64_t* in_bitmap, const int32_t* in_prefix,
                                 const int32_t* in_rank_to_row, int kd, int kh, int kw, int sd,
                                 int sh, int sw, int pd, int ph, int pw, int32_t* nbr,
                                 int32_t* pair_count, const int32_t* n_out_live, void* stream) {
  return glx_rules_strided_dilated(indices_out, N_out, N_in, B, D, H, W, in_bitmap, in_prefix, in_rank_to_row, kd, kh, kw,
                                   sd, sh, sw, pd, ph, pw, 1, 1, 1, nbr, pair_count, n_out_live, stream);
}

extern "C" int glx_rules_strided_dilated(const int32_t* indices_out, int N_out, int N_in, int B, int D,
                                         int H, int W, const uint64_t* in_bitmap, const int32_t* in_prefix,
                                         const int32_t* in_rank_to_row, int kd, int kh, int kw, int sd,
                                         int sh, int sw, int pd, int ph, int pw, int dd, int dh, int dw, int32_t* nbr,
                                         int32_t* pair_count, const int32_t* n_out_live, void* stream) {
  if (N_out == 0) return GLX_OK;
  GLX_REQUIRE(indices_out && in_bitmap && in_prefix && nbr, "glx_rules_strided: null pointer");
  GLX_REQUIRE(dd > 0 && dh > 0 && dw > 0, "glx_rules_strided: dilation must be positive");
  GlxGrid ig{B, D, H, W};
  ConvGeom cg{kd, kh, kw, sd, sh, sw, pd, ph, pw, dd, dh, dw};
  long long total = (long long)N_out * kd * kh;
  hipLaunchKernelGGL(k_rules_strided, dim3(glx_divup(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const int4*)indices_out, N_out, N_in, ig,
                     (const unsigned long long*)in_bitmap, (const int*)in_prefix, in_rank_to_row,
                     cg, nbr, pair_count, n_out_live);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

__global__ void k_rules_invert(const int* __restrict__ nbr, long long total, int K,
                               int* __restrict__ nbr_in, const int* __restrict__ n_live) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_live) total = min(total, (long long)*n_live * K);   // rows past the live count are undefined
  if (t >= total) return;
  int i = nbr[t];
  if (i < 0) return;
  int j = (int)(t / K);
  int k = (int)(t - (long long)j * K);
  nbr_in[(long long)i * K + k] = j;
}

extern "C" int glx_rules_invert(const int32_t* nbr, int N_out, int K, int N_in, int32_t* nbr_in,
                                const int32_t* n_out_live, void* stream) {
  GLX_REQUIRE(K > 0, "glx_rules_invert: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (N_in > 0) {
    GlxFillJob job{nbr_in, (size_t)N_in * K * sizeof(int), 0xFF};
    int rc = glx_fill_multi(&job, 1, st);
    if (rc != GLX_OK) return rc;
  }
  long long total = (long long)N_out * K;
  if (total > 0) {
    hipLaunchKernelGGL(k_rules_invert, dim3(glx_divup(total, 256)), dim3(256), 0, st, nbr, total,
                       K, nbr_in, n_out_live);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ several copy-then-fill regions, one launch
// Loading the next batch into the static input buffers of a recorded step is seven small launches as tensor ops
// (points, frame ids + their padding value, the zero-padded ground-truth / label-variance blocks of every frame).
// Region i: dst[i][0 .. copy_words) = src[i][...], dst[i][copy_words .. total_words) = fill[i]; 32-bit words.
#define GLX_CF_MAX 48
#define GLX_CF_BLOCK_WORDS 4096   // 256 threads x 16 words
struct GlxCopyFillArgs {
  unsigned long long dst[GLX_CF_MAX], src[GLX_CF_MAX];
  unsigned copy_words[GLX_CF_MAX], total_words[GLX_CF_MAX], fill[GLX_CF_MAX];
  unsigned first_block[GLX_CF_MAX + 1];
  int n;
};

__global__ __launch_bounds__(256) void k_copy_fill_multi(GlxCopyFillArgs a) {
  int r = 0;
  for (int i = 1; i < a.n; ++i)
    if (blockIdx.x >= a.first_block[i]) r = i;
  unsigned* dst = reinterpret_cast<unsigned*>(a.dst[r]);
  const unsigned* src = reinterpret_cast<const unsigned*>(a.src[r]);
  const unsigned nc = a.copy_words[r], nt = a.total_words[r], fv = a.fill[r];
  const unsigned base = (blockIdx.x - a.first_block[r]) * GLX_CF_BLOCK_WORDS;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned w = base + i * 256 + threadIdx.x;
    if (w < nt) dst[w] = w < nc ? src[w] : fv;
  }
}

extern "C" int glx_copy_fill_multi(int n, void* const* dst, const void* const* src, const uint32_t* copy_words,
                                   const uint32_t* total_words, const uint32_t* fill, void* stream) {
  if (n <= 0) return GLX_OK;
  GLX_REQUIRE(dst && src && copy_words && total_words && fill, "glx_copy_fill_multi: null pointer");
  GLX_REQUIRE(n <= GLX_CF_MAX, "glx_copy_fill_multi: %d regions (at most %d)", n, GLX_CF_MAX);
  GlxCopyFillArgs a;
  memset(&a, 0, sizeof(a));
  unsigned blocks = 0;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (total_words[i] == 0) continue;
    GLX_REQUIRE(dst[i] && (copy_words[i] == 0 || src[i]) && copy_words[i] <= total_words[i],
                "glx_copy_fill_multi: region %d", i);
    a.dst[m] = (unsigned long long)dst[i];
    a.src[m] = (unsigned long long)src[i];
    a.copy_words[m] = copy_words[i];
    a.total_words[m] = total_words[i];
    a.fill[m] = fill[i];
    a.first_block[m] = blocks;
    blocks += (total_words[i] + GLX_CF_BLOCK_WORDS - 1) / GLX_CF_BLOCK_WORDS;
    ++m;
  }
  a.n = m;
  a.first_block[m] = blocks;
  if (blocks == 0) return GLX_OK;
  hipLaunchKernelGGL(k_copy_fill_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
