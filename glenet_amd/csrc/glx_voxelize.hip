// Point-cloud voxelization on device.
//
// Hard voxelization restates the CPU generators the reference calls through
// VoxelGeneratorWrapper (pcdet/datasets/processor/data_processor.py:15-60 ->
// spconv.utils.VoxelGeneratorV2 / Point2VoxelCPU3d, third-party): a sequential loop over
// points with a cell -> voxel-id map.  The sequential "first seen" semantics are
// reproduced in parallel without a sort:
//   1. every point sets its cell's bit; a popcount scan ranks the distinct cells;
//   2. atomicMin per cell finds the first point of each cell;
//   3. an exclusive scan of "I am my cell's first point" flags over the points numbers the
//      voxels in first-seen order (frames are stacked, so the scan is per frame after
//      subtracting the frame base); ids >= max_voxels are dropped exactly like the
//      generator's `continue`;
//   4. the first max_points points of a voxel = the max_points smallest point ids of its
//      cell: an atomicMin cascade over sorted slots.
// Dynamic voxelization follows DynamicMeanVFE.forward
// (pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:53-72): sorted-unique over the key
// b*XYZ + x*YZ + y*Z + z is exactly the rank order of a bitmap laid out (b, x, y, z).
#include "glx_common.h"
#include "glx_fill.h"
#include "glx_scan.h"

int glx_scan_bitmap(const GlxGrid& g, uint64_t* bitmap, const uint8_t* chunk_flags,
                    int32_t* prefix, int32_t* n_total, void* workspace, size_t workspace_bytes,
                    hipStream_t st);

#define VOX_SENT 0x7F7F7F7F

struct VoxGeom {
  float xmin, ymin, zmin;
  float vx, vy, vz;
  int gx, gy, gz;
  int depth;   // z extent of the bitmap layout (>= gz)
};

// cell of a point, reference arithmetic: floor((p - min) / size) in fp32, per axis.
__device__ __forceinline__ bool vox_cell(const float* __restrict__ p, const VoxGeom& vg, int& cx,
                                         int& cy, int& cz) {
  float fx = floorf((p[0] - vg.xmin) / vg.vx);
  float fy = floorf((p[1] - vg.ymin) / vg.vy);
  float fz = floorf((p[2] - vg.zmin) / vg.vz);
  if (!(fx >= 0.f && fx < (float)vg.gx && fy >= 0.f && fy < (float)vg.gy && fz >= 0.f &&
        fz < (float)vg.gz))
    return false;
  cx = (int)fx; cy = (int)fy; cz = (int)fz;
  return true;
}

// XMAJOR=false: lin = ((b*gz + z)*gy + y)*gx + x  (hard: coords [b,z,y,x] ascending)
// XMAJOR=true : lin = ((b*gx + x)*gy + y)*gz + z  (dynamic: DynamicMeanVFE key order)
template <bool XMAJOR>
__global__ void k_vox_mark(const float* __restrict__ pts, const int* __restrict__ pbatch, int P,
                           int C, int B, VoxGeom vg, unsigned long long* __restrict__ bitmap,
                           unsigned char* __restrict__ chunk_flags,
                           long long* __restrict__ cell_lin, int* __restrict__ frame_start) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int b = pbatch ? pbatch[p] : 0;
  if (frame_start && (unsigned)b < (unsigned)B) {
    if (p == 0 || pbatch[p - 1] != b) frame_start[b] = p;  // pbatch != NULL whenever B > 1
  }
  int cx, cy, cz;
  long long l = -1;
  if ((unsigned)b < (unsigned)B && vox_cell(pts + (long long)p * C, vg, cx, cy, cz)) {
    l = XMAJOR ? (((long long)b * vg.gx + cx) * vg.gy + cy) * vg.gz + cz
               : (((long long)b * vg.depth + cz) * vg.gy + cy) * vg.gx + cx;
    atomicOr(&bitmap[l >> 6], 1ull << (l & 63));
    chunk_flags[l >> 9] = 1;
  }
  cell_lin[p] = l;
}

__global__ void k_vox_first_point(const long long* __restrict__ cell_lin, int P,
                                  const unsigned long long* __restrict__ bitmap,
                                  const int* __restrict__ prefix, int* __restrict__ cell_rank,
                                  int* __restrict__ first_pt) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  long long l = cell_lin[p];
  int r = -1;
  if (l >= 0) {
    r = glx_rank_lookup(bitmap, prefix, l);
    atomicMin(&first_pt[r], p);
  }
  cell_rank[p] = r;
}

__global__ void k_vox_flags(const int* __restrict__ cell_rank, const int* __restrict__ first_pt,
                            int P, int* __restrict__ flags) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int r = cell_rank[p];
  flags[p] = (r >= 0 && first_pt[r] == p) ? 1 : 0;
}

// single thread: frame bases in the global first-seen numbering and capped output offsets
__global__ void k_vox_frames(const int* __restrict__ excl, const int* __restrict__ total, int P,
                             int B, int max_voxels, int* __restrict__ frame_start,
                             int* __restrict__ frame_base, int* __restrict__ voxel_offset) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int next = P;
  for (int b = B - 1; b >= 0; --b) {  // empty frames inherit the next frame's start
    if (frame_start[b] < 0) frame_start[b] = next;
    next = frame_start[b];
  }
  int tot = *total;
  for (int b = 0; b < B; ++b) frame_base[b] = frame_start[b] < P ? excl[frame_start[b]] : tot;
  frame_base[B] = tot;
  int off = 0;
  for (int b = 0; b < B; ++b) {
    voxel_offset[b] = off;
    int cnt = frame_base[b + 1] - frame_base[b];
    off += cnt < max_voxels ? cnt : max_voxels;
  }
  voxel_offset[B] = off;
}

__global__ void k_vox_assign_rows(const int* __restrict__ flags, const int* __restrict__ excl,
                                  const int* __restrict__ cell_rank,
                                  const long long* __restrict__ cell_lin,
                                  const int* __restrict__ pbatch, int P, VoxGeom vg, int max_voxels,
                                  const int* __restrict__ frame_base,
                                  const int* __restrict__ voxel_offset,
                                  int* __restrict__ row_of_rank, int4* __restrict__ coords) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P || !flags[p]) return;
  int b = pbatch ? pbatch[p] : 0;
  int local = excl[p] - frame_base[b];
  int row = -1;
  if (local < max_voxels) {
    row = voxel_offset[b] + local;
    long long l = cell_lin[p];
    int x = (int)(l % vg.gx);
    long long q = l / vg.gx;
    int y = (int)(q % vg.gy);
    int z = (int)((q / vg.gy) % vg.depth);
    coords[row] = make_int4(b, z, y, x);
  }
  row_of_rank[cell_rank[p]] = row;
}

__global__ void k_vox_slots(const int* __restrict__ cell_rank, const int* __restrict__ row_of_rank,
                            int P, int max_points, int* __restrict__ slots) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int r = cell_rank[p];
  if (r < 0) return;
  int row = row_of_rank[r];
  if (row < 0) return;
  int* s = slots + (long long)row * max_points;
  int v = p;
  for (int i = 0; i < max_points; ++i) {  // keeps the max_points smallest ids, sorted
    int old = atomicMin(&s[i], v);
    if (old == VOX_SENT) break;
    v = old > v ? old : v;
  }
}

__global__ void k_vox_fill(const float* __restrict__ pts, const int* __restrict__ slots,
                           const int* __restrict__ n_rows_dev, int max_points, int C,
                           float* __restrict__ voxels, int* __restrict__ num_points) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int n_rows = *n_rows_dev;
  if (t >= (long long)n_rows * max_points) return;
  int row = (int)(t / max_points);
  int s = (int)(t - (long long)row * max_points);
  int pidx = slots[t];
  float* dst = voxels + t * C;
  if (pidx != VOX_SENT) {
    const float* src = pts + (long long)pidx * C;
    for (int c = 0; c < C; ++c) dst[c] = src[c];
  } else {
    for (int c = 0; c < C; ++c) dst[c] = 0.f;
  }
  if (s == 0) {
    int n = 0;
    for (int i = 0; i < max_points; ++i) n += slots[(long long)row * max_points + i] != VOX_SENT;
    num_points[row] = n;
  }
}

struct HardWs {
  uint64_t* bitmap;
  uint8_t* cflags;
  int32_t* prefix;
  long long* cell_lin;
  int *cell_rank, *first_pt, *flags, *excl, *row_of_rank, *slots, *frame_start, *frame_base,
      *total, *n_unique;
  void* scan_ws;
  size_t scan_ws_bytes;
  size_t bytes;
};

static HardWs hard_ws_layout(void* base, int P, int B, int gx, int gy, int gz, int max_points,
                             int max_voxels) {
  HardWs w;
  GlxGrid g{B, gz + 1, gy, gx};   // sized for index_depth up to gz + 1 (the backbone's sparse shape)
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t n) {
    void* r = p ? p + off : nullptr;
    off += glx_align(n);
    return r;
  };
  size_t Pn = P > 0 ? P : 1;
  w.bitmap = (uint64_t*)take((size_t)g.words() * 8);
  w.cflags = (uint8_t*)take((size_t)g.chunks());
  w.prefix = (int32_t*)take((size_t)g.words() * 4);
  w.cell_lin = (long long*)take(Pn * 8);
  w.cell_rank = (int*)take(Pn * 4);
  w.first_pt = (int*)take(Pn * 4);
  w.flags = (int*)take(Pn * 4);
  w.excl = (int*)take(Pn * 4);
  w.row_of_rank = (int*)take(Pn * 4);
  w.slots = (int*)take((size_t)B * max_voxels * max_points * 4);
  w.frame_start = (int*)take((size_t)(B + 1) * 4);
  w.frame_base = (int*)take((size_t)(B + 1) * 4);
  w.total = (int*)take(4);
  w.n_unique = (int*)take(4);
  size_t s1 = glx_scan_workspace_bytes(g.words()), s2 = glx_scan_workspace_bytes(P);
  w.scan_ws_bytes = s1 > s2 ? s1 : s2;
  w.scan_ws = take(w.scan_ws_bytes);
  w.bytes = off;
  return w;
}

extern "C" size_t glx_voxelize_hard_workspace_bytes(int P, int B, int gx, int gy, int gz,
                                                    int max_points, int max_voxels) {
  return hard_ws_layout(nullptr, P, B, gx, gy, gz, max_points, max_voxels).bytes + 256;
}

extern "C" int glx_voxelize_hard(const float* points, const int32_t* point_batch, int P, int C,
                                 int B, const float* vrange, const float* vsize, int gx, int gy,
                                 int gz, int max_points, int max_voxels, float* voxels,
                                 int32_t* coords, int32_t* num_points, int32_t* voxel_offset,
                                 int index_depth, uint64_t* idx_bitmap, uint8_t* idx_flags,
                                 int32_t* idx_prefix, int32_t* idx_rank_to_row,
                                 int32_t* idx_n_unique, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  // an empty cloud (P == 0: no point buffer to speak of) is a valid input: zero voxels, offsets all zero
  GLX_REQUIRE((points || P == 0) && vrange && vsize && voxels && coords && num_points && voxel_offset,
              "glx_voxelize_hard: null pointer");
  GLX_REQUIRE(P >= 0 && C >= 3 && B >= 1 && gx > 0 && gy > 0 && gz > 0 && max_points > 0 &&
                  max_voxels > 0,
              "glx_voxelize_hard: bad sizes");
  GLX_REQUIRE(B == 1 || point_batch || P == 0, "glx_voxelize_hard: point_batch required when B > 1");
  HardWs w = hard_ws_layout(workspace, P, B, gx, gy, gz, max_points, max_voxels);
  if (!workspace || workspace_bytes < w.bytes) {
    glx_set_error("glx_voxelize_hard: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
    return GLX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  // The cell bitmap may be laid out for a deeper grid (index_depth >= gz) and written into
  // caller buffers: it is then directly the cell index of the SparseConvTensor built from these
  // voxels (spatial_shape [index_depth, gy, gx], spconv_backbone.py:75), saving a second build.
  const int depth = index_depth > 0 ? index_depth : gz;
  GLX_REQUIRE(depth >= gz && depth <= gz + 1, "glx_voxelize_hard: index_depth must be gz or gz+1");
  if (idx_bitmap) {
    GLX_REQUIRE(idx_flags && idx_prefix && idx_rank_to_row && idx_n_unique,
                "glx_voxelize_hard: incomplete index outputs");
    w.bitmap = idx_bitmap; w.cflags = idx_flags; w.prefix = idx_prefix;
    w.row_of_rank = idx_rank_to_row; w.n_unique = idx_n_unique;
  }
  GlxGrid g{B, depth, gy, gx};
  VoxGeom vg{vrange[0], vrange[1], vrange[2], vsize[0], vsize[1], vsize[2], gx, gy, gz, depth};
  const int nb = glx_divup(P > 0 ? P : 1, 256);
  {
    const size_t Pn = (size_t)(P > 0 ? P : 1);
    GlxFillJob jobs[6] = {{w.bitmap, (size_t)g.words() * 8, 0},
                          {w.cflags, (size_t)g.chunks(), 0},
                          {w.first_pt, Pn * 4, 0x7F},
                          {w.row_of_rank, Pn * 4, 0xFF},
                          {w.slots, (size_t)B * max_voxels * max_points * 4, 0x7F},
                          {w.frame_start, (size_t)(B + 1) * 4, 0xFF}};
    int frc = glx_fill_multi(jobs, 6, st);
    if (frc != GLX_OK) return frc;
  }
  if (P > 0) {
    hipLaunchKernelGGL((k_vox_mark<false>), dim3(nb), dim3(256), 0, st, points, point_batch, P, C,
                       B, vg, (unsigned long long*)w.bitmap, w.cflags, w.cell_lin,
                       point_batch ? w.frame_start : nullptr);
  }
  int rc = glx_scan_bitmap(g, w.bitmap, w.cflags, w.prefix, w.n_unique, w.scan_ws, w.scan_ws_bytes,
                           st);
  if (rc != GLX_OK) return rc;
  if (P > 0) {
    hipLaunchKernelGGL(k_vox_first_point, dim3(nb), dim3(256), 0, st, w.cell_lin, P,
                       (const unsigned long long*)w.bitmap, (const int*)w.prefix, w.cell_rank,
                       w.first_pt);
    hipLaunchKernelGGL(k_vox_flags, dim3(nb), dim3(256), 0, st, (const int*)w.cell_rank,
                       (const int*)w.first_pt, P, w.flags);
  }
  IntArray fa{w.flags};
  rc = glx_exclusive_scan(fa, P, w.excl, w.total, w.scan_ws, w.scan_ws_bytes, st);
  if (rc != GLX_OK) return rc;
  if (!point_batch) {  // single frame starts at 0
    GlxFillJob job{w.frame_start, 4, 0};
    rc = glx_fill_multi(&job, 1, st);
    if (rc != GLX_OK) return rc;
  }
  hipLaunchKernelGGL(k_vox_frames, dim3(1), dim3(64), 0, st, (const int*)w.excl,
                     (const int*)w.total, P, B, max_voxels, w.frame_start, w.frame_base,
                     voxel_offset);
  if (P > 0) {
    hipLaunchKernelGGL(k_vox_assign_rows, dim3(nb), dim3(256), 0, st, (const int*)w.flags,
                       (const int*)w.excl, (const int*)w.cell_rank, (const long long*)w.cell_lin,
                       point_batch, P, vg, max_voxels, (const int*)w.frame_base,
                       (const int*)voxel_offset, w.row_of_rank, (int4*)coords);
    hipLaunchKernelGGL(k_vox_slots, dim3(nb), dim3(256), 0, st, (const int*)w.cell_rank,
                       (const int*)w.row_of_rank, P, max_points, w.slots);
    long long cap = (long long)B * max_voxels * max_points;
    hipLaunchKernelGGL(k_vox_fill, dim3(glx_divup(cap, 256)), dim3(256), 0, st, points,
                       (const int*)w.slots, (const int*)(voxel_offset + B), max_points, C, voxels,
                       num_points);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ dynamic + mean
__global__ void k_dyn_accumulate(const float* __restrict__ pts, const long long* __restrict__ cell_lin,
                                 int P, int C, const unsigned long long* __restrict__ bitmap,
                                 const int* __restrict__ prefix, float* __restrict__ sums,
                                 int* __restrict__ counts) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)P * C) return;
  int p = (int)(t / C);
  int c = (int)(t - (long long)p * C);
  long long l = cell_lin[p];
  if (l < 0) return;
  int r = glx_rank_lookup(bitmap, prefix, l);
  atomicAdd(&sums[(long long)r * C + c], pts[t]);
  if (c == 0) atomicAdd(&counts[r], 1);
}

__global__ void k_dyn_emit(const unsigned long long* __restrict__ bitmap,
                           const int* __restrict__ prefix, long long nwords, int gx, int gy, int gz,
                           int C, const int* __restrict__ counts, float* __restrict__ feats,
                           int4* __restrict__ coords) {
  long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  unsigned long long word = bitmap[w];
  if (!word) return;
  int r = prefix[w];
  while (word) {
    int bit = __ffsll((long long)word) - 1;
    word &= word - 1;
    long long l = (w << 6) + bit;  // ((b*gx + x)*gy + y)*gz + z
    int z = (int)(l % gz);
    long long q = l / gz;
    int y = (int)(q % gy);
    q /= gy;
    int x = (int)(q % gx);
    int b = (int)(q / gx);
    coords[r] = make_int4(b, z, y, x);
    float inv = (float)counts[r];
    for (int c = 0; c < C; ++c) feats[(long long)r * C + c] = feats[(long long)r * C + c] / inv;
    ++r;
  }
}

struct DynWs {
  uint64_t* bitmap;
  uint8_t* flags;
  int32_t* prefix;
  long long* cell_lin;
  int* counts;
  void* scan_ws;
  size_t scan_ws_bytes, bytes;
};

static DynWs dyn_ws_layout(void* base, int P, int B, int gx, int gy, int gz) {
  DynWs w;
  GlxGrid g{B, gx, gy, gz};
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t n) {
    void* r = p ? p + off : nullptr;
    off += glx_align(n);
    return r;
  };
  size_t Pn = P > 0 ? P : 1;
  w.bitmap = (uint64_t*)take((size_t)g.words() * 8);
  w.flags = (uint8_t*)take((size_t)g.chunks());
  w.prefix = (int32_t*)take((size_t)g.words() * 4);
  w.cell_lin = (long long*)take(Pn * 8);
  w.counts = (int*)take(Pn * 4);
  w.scan_ws_bytes = glx_scan_workspace_bytes(g.words());
  w.scan_ws = take(w.scan_ws_bytes);
  w.bytes = off;
  return w;
}

// ------------------------------------------------------------------ device data step
// DataProcessor.mask_points_and_boxes_outside_range + shuffle_points (data_processor.py:78-105) on the stacked,
// capacity-sized point buffer of a shape-static step: no boolean indexing (variable shapes, a host read-back), no
// sort and no library call (rocPRIM's sort puts memset nodes into a captured graph, which ROCm 7.2 replays
// unreliably -- DESIGN 3a).  Four launches:
//   scan (2)    excl[i] = kept points before row i (keep = frame id valid and x / y inside the CLOSED range,
//               common_utils.py:60-63), total kept
//   frames      frame f's first row by binary search in the non-decreasing frame ids -> kept offsets ko[0..B];
//               the step's permutation key = seed ^ f(call counter), counter += 1
//   scatter     kept row i of frame f with in-frame rank r goes to ko[f] + perm_f(r): a keyed BIJECTION of
//               [0, n_f) -- a 6-round Feistel network on the next even power of two with cycle walking -- i.e. a
//               pseudo-random permutation of the frame's kept points, evaluated independently per point.
//               Rows behind the kept ones get frame id B (padding for glx_voxelize_hard) and zeros.
struct DsKeep {
  const float* pts; const int* batch; int C, B; float x0, y0, x1, y1;
  __device__ int operator()(long long i) const {
    const int b = batch ? batch[i] : 0;
    const float x = pts[i * C], y = pts[i * C + 1];
    return (b >= 0 && b < B && x >= x0 && x <= x1 && y >= y0 && y <= y1) ? 1 : 0;
  }
};

__device__ __forceinline__ unsigned ds_mix(unsigned x) {        // a 32-bit finaliser (murmur3)
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}

// bijection of [0, n): Feistel rounds on 2 * half_bits bits, walked until the value falls below n
__device__ __forceinline__ unsigned ds_perm(unsigned r, unsigned n, unsigned long long key) {
  if (n <= 1) return 0;
  int bits = 32 - __clz(n - 1);
  bits += bits & 1;
  if (bits < 2) bits = 2;
  const int hb = bits >> 1;
  const unsigned hmask = (1u << hb) - 1u;
  unsigned v = r;
  do {
    unsigned L = v >> hb, R = v & hmask;
#pragma unroll
    for (int round = 0; round < 6; ++round) {
      const unsigned k = (unsigned)(key >> ((round & 1) * 32)) + 0x9E3779B9u * (unsigned)(round + 1);
      const unsigned F = ds_mix(R ^ k) & hmask;
      const unsigned t = L ^ F;
      L = R; R = t;
    }
    v = (L << hb) | R;
  } while (v >= n);
  return v;
}

__global__ void k_ds_frames(const int* __restrict__ batch, const int* __restrict__ excl,
                            const int* __restrict__ n_kept, int P, int B, int* __restrict__ ko,
                            unsigned long long* __restrict__ counter, unsigned long long* __restrict__ key) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f <= B) {
    // first row whose frame id is >= f (ids non-decreasing; padding rows carry B)
    int lo = 0, hi = P;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((batch ? batch[mid] : 0) < f) lo = mid + 1; else hi = mid;
    }
    ko[f] = (f == B || lo >= P) ? *n_kept : excl[lo];
  }
  if (f == 0) {
    const unsigned long long c = counter[1];
    // splitmix64 of (seed, call number): a fresh key every call, reproducible from the seed
    unsigned long long z = counter[0] + 0x9E3779B97F4A7C15ull * (c + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    *key = z ^ (z >> 31);
    counter[1] = c + 1;
  }
}

__global__ void k_ds_scatter(DsKeep keep, const int* __restrict__ excl, const int* __restrict__ ko,
                             const int* __restrict__ n_kept, const unsigned long long* __restrict__ key, int P,
                             int shuffle, float* __restrict__ out, int* __restrict__ out_batch,
                             int* __restrict__ order) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int C = keep.C;
  if (i >= *n_kept) {                                   // the tail of the output: padding rows
    out_batch[i] = keep.B;
    if (order) order[i] = -1;
    for (int c = 0; c < C; ++c) out[(long long)i * C + c] = 0.f;
  }
  if (!keep(i)) return;
  const int f = keep.batch ? keep.batch[i] : 0;
  const unsigned r = (unsigned)(excl[i] - ko[f]), n = (unsigned)(ko[f + 1] - ko[f]);
  const unsigned long long k = *key ^ (0xD1B54A32D192ED03ull * (unsigned long long)(f + 1));
  const int dest = ko[f] + (int)(shuffle ? ds_perm(r, n, k) : r);
  for (int c = 0; c < C; ++c) out[(long long)dest * C + c] = keep.pts[(long long)i * C + c];
  out_batch[dest] = f;
  if (order) order[dest] = i;
}

extern "C" size_t glx_mask_shuffle_workspace_bytes(int P, int B) {
  return glx_align((size_t)(P > 0 ? P : 1) * sizeof(int)) + glx_align((size_t)(B + 2) * sizeof(int)) +
         glx_scan_workspace_bytes(P > 0 ? P : 1) + 512;
}

extern "C" int glx_mask_shuffle(const float* points, const int32_t* point_batch, int P, int C, int B,
                                const float* range_xy, int shuffle, uint64_t* seed_and_calls, float* out_points,
                                int32_t* out_batch, int32_t* order, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (P <= 0) return GLX_OK;
  GLX_REQUIRE(points && range_xy && seed_and_calls && out_points && out_batch && C >= 2 && B >= 1,
              "glx_mask_shuffle: bad arguments");
  GLX_REQUIRE(out_points != points, "glx_mask_shuffle: in-place operation is not supported");
  if (!workspace || workspace_bytes < glx_mask_shuffle_workspace_bytes(P, B) - 256) {
    glx_set_error("glx_mask_shuffle: workspace %zu < %zu bytes", workspace_bytes, glx_mask_shuffle_workspace_bytes(P, B));
    return GLX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)workspace;
  int* excl = (int*)w; w += glx_align((size_t)P * sizeof(int));
  int* ko = (int*)w; w += glx_align((size_t)(B + 2) * sizeof(int));     // ko[0..B], then n_kept
  int* n_kept = ko + B + 1;
  unsigned long long* key = (unsigned long long*)w; w += 256;
  void* scan_ws = w;
  const size_t scan_bytes = glx_scan_workspace_bytes(P);
  DsKeep keep{points, point_batch, C, B, range_xy[0], range_xy[1], range_xy[2], range_xy[3]};
  int rc = glx_exclusive_scan(keep, P, excl, n_kept, scan_ws, scan_bytes, st);
  if (rc != GLX_OK) return rc;
  hipLaunchKernelGGL(k_ds_frames, dim3(glx_divup(B + 1, 64)), dim3(64), 0, st, point_batch, (const int*)excl,
                     (const int*)n_kept, P, B, ko, (unsigned long long*)seed_and_calls, key);
  hipLaunchKernelGGL(k_ds_scatter, dim3(glx_divup(P, 256)), dim3(256), 0, st, keep, (const int*)excl, (const int*)ko,
                     (const int*)n_kept, (const unsigned long long*)key, P, shuffle ? 1 : 0, out_points, out_batch,
                     order);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" size_t glx_voxelize_dynamic_workspace_bytes(int P, int B, int gx, int gy, int gz) {
  return dyn_ws_layout(nullptr, P, B, gx, gy, gz).bytes + 256;
}

extern "C" int glx_voxelize_dynamic_mean(const float* points, const int32_t* point_batch, int P,
                                         int C, int B, const float* vrange, const float* vsize,
                                         int gx, int gy, int gz, float* features, int32_t* coords,
                                         int32_t* n_voxels, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  GLX_REQUIRE((points || P == 0) && vrange && vsize && features && coords && n_voxels,
              "glx_voxelize_dynamic_mean: null pointer");
  GLX_REQUIRE(P >= 0 && C >= 3 && B >= 1 && gx > 0 && gy > 0 && gz > 0,
              "glx_voxelize_dynamic_mean: bad sizes");
  GLX_REQUIRE(B == 1 || point_batch || P == 0, "glx_voxelize_dynamic_mean: point_batch required when B > 1");
  DynWs w = dyn_ws_layout(workspace, P, B, gx, gy, gz);
  if (!workspace || workspace_bytes < w.bytes) {
    glx_set_error("glx_voxelize_dynamic_mean: workspace %zu < %zu bytes", workspace_bytes, w.bytes);
    return GLX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  GlxGrid g{B, gx, gy, gz};
  VoxGeom vg{vrange[0], vrange[1], vrange[2], vsize[0], vsize[1], vsize[2], gx, gy, gz, gz};
  {
    GlxFillJob jobs[4] = {{w.bitmap, (size_t)g.words() * 8, 0},
                          {w.flags, (size_t)g.chunks(), 0},
                          {P > 0 ? features : nullptr, (size_t)P * C * 4, 0},
                          {P > 0 ? w.counts : nullptr, (size_t)P * 4, 0}};
    int frc = glx_fill_multi(jobs, 4, st);
    if (frc != GLX_OK) return frc;
  }
  if (P > 0) {
    hipLaunchKernelGGL((k_vox_mark<true>), dim3(glx_divup(P, 256)), dim3(256), 0, st, points,
                       point_batch, P, C, B, vg, (unsigned long long*)w.bitmap, w.flags,
                       w.cell_lin, (int*)nullptr);
  }
  int rc = glx_scan_bitmap(g, w.bitmap, w.flags, w.prefix, n_voxels, w.scan_ws, w.scan_ws_bytes, st);
  if (rc != GLX_OK) return rc;
  if (P > 0) {
    long long total = (long long)P * C;
    hipLaunchKernelGGL(k_dyn_accumulate, dim3(glx_divup(total, 256)), dim3(256), 0, st, points,
                       (const long long*)w.cell_lin, P, C, (const unsigned long long*)w.bitmap,
                       (const int*)w.prefix, features, w.counts);
    long long nwords = g.words();
    hipLaunchKernelGGL(k_dyn_emit, dim3(glx_divup(nwords, 256)), dim3(256), 0, st,
                       (const unsigned long long*)w.bitmap, (const int*)w.prefix, nwords, gx, gy,
                       gz, C, (const int*)w.counts, features, (int4*)coords);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ MeanVFE
__global__ void k_mean_vfe(const float* __restrict__ voxels, const int* __restrict__ num, int Nv,
                           int mp, int C, float* __restrict__ out, const int* __restrict__ n_live) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_live) Nv = min(Nv, *n_live);
  if (t >= (long long)Nv * C) return;
  int v = (int)(t / C);
  int c = (int)(t - (long long)v * C);
  const float* src = voxels + (long long)v * mp * C + c;
  float s = 0.f;
  for (int p = 0; p < mp; ++p) s += src[(long long)p * C];
  float n = (float)num[v];
  out[t] = s / (n < 1.f ? 1.f : n);
}

extern "C" int glx_mean_vfe(const float* voxels, const int32_t* num_points, int Nv, int max_points,
                            int C, float* out, const int32_t* n_live, void* stream) {
  GLX_REQUIRE(Nv >= 0 && max_points > 0 && C > 0, "glx_mean_vfe: bad sizes");
  if (Nv == 0) return GLX_OK;   // no voxels: the (empty) buffers may be null
  GLX_REQUIRE(voxels && num_points && out, "glx_mean_vfe: null pointer");
  long long total = (long long)Nv * C;
  hipLaunchKernelGGL(k_mean_vfe, dim3(glx_divup(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     voxels, num_points, Nv, max_points, C, out, n_live);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
