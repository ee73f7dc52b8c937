// Point / box operators of pcdet.ops: points_in_boxes, RoI-aware pooling, RoI point pooling,
// voxel query, ball query, grouping.  Arithmetic restates the reference CUDA kernels cited at
// each function (fp32, contraction off); the parallel decomposition is ours.
#include "glx_common.h"
#include "glx_fill.h"
#include "glx_scan.h"
#include "glx_libm.h"

// ------------------------------------------------------------------ inside test
// check_pt_in_box3d, pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:23-36
// (== roipoint_pool3d_kernel.cu:22-35): z test in double (dz / 2.0), xy in double vs MARGIN.
struct BoxT {
  float cx, cy, cz, dx, dy, dz, cosa, sina;
  __device__ void load(const float* b) {
    cx = b[0]; cy = b[1]; cz = b[2]; dx = b[3]; dy = b[4]; dz = b[5];
    // The reference calls the float overloads (lidar_to_local_coords, roiaware_pool3d_kernel.cu:23-27; cosf / sinf of
    // glibc in roiaware_pool3d.cpp:119-123).  glx_libm.h returns glibc's bits (verified over all finite floats), which
    // is what the CPU oracle / libglenet_host.so hold; the device's own cosf / sinf (ocml) are different ~1 ulp
    // routines and would flip boundary points against those.  A CUDA build of the reference (libdevice float trig)
    // can differ from either by an ulp of the rotation: points within ~1e-6 m of a box face may be classified
    // differently there.
    cosa = glxm::cosf_(-b[6]);
    sina = glxm::sinf_(-b[6]);
  }
  __device__ __forceinline__ int contains(float x, float y, float z, float margin, float& lx,
                                          float& ly) const {
    if (fabsf(z - cz) > dz / 2.0) return 0;
    float sx = x - cx, sy = y - cy;
    lx = sx * cosa + sy * (-sina);
    ly = sx * sina + sy * cosa;
    return (int)(fabsf(lx) < dx / 2.0 + margin) & (int)(fabsf(ly) < dy / 2.0 + margin);   // branch-free, as the reference's
  }
};

#define PIB_MARGIN 1e-5f

// points_in_boxes_kernel, roiaware_pool3d_kernel.cu:313-336: first containing box or -1.
// boxes of the frame are transformed once into LDS, points stream through.
__global__ void k_points_in_boxes(int B, int T, int P, const float* __restrict__ boxes,
                                  const float* __restrict__ pts, int* __restrict__ out) {
  extern __shared__ BoxT sbox[];
  const int b = blockIdx.y;
  for (int k = threadIdx.x; k < T; k += blockDim.x) sbox[k].load(boxes + ((long long)b * T + k) * 7);
  __syncthreads();
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float* q = pts + ((long long)b * P + p) * 3;
  float x = q[0], y = q[1], z = q[2], lx, ly;
  int r = -1;
  for (int k = 0; k < T; ++k)
    if (sbox[k].contains(x, y, z, PIB_MARGIN, lx, ly)) { r = k; break; }
  out[(long long)b * P + p] = r;
}

extern "C" int glx_points_in_boxes(const float* boxes, const float* pts, int B, int T, int P,
                                   int32_t* box_idx_of_points, void* stream) {
  if (B == 0 || P == 0) return GLX_OK;
  GLX_REQUIRE(pts && box_idx_of_points && (T == 0 || boxes), "glx_points_in_boxes: null pointer");
  GLX_REQUIRE((size_t)T * sizeof(BoxT) <= 64 * 1024, "glx_points_in_boxes: %d boxes per frame", T);
  hipLaunchKernelGGL(k_points_in_boxes, dim3(glx_divup(P, 256), B), dim3(256), T * sizeof(BoxT),
                     (hipStream_t)stream, B, T, P, boxes, pts, box_idx_of_points);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ RoI-aware pooling
// generate_pts_mask_for_box3d + collect_inside_pts_for_box3d (kernel.cu:39-108): per box, the
// inside points are appended to their voxel's list IN POINT ORDER, at most max_pts-1 each.
// One wave per box walks the points 64 at a time; inside a batch the rank of a point among
// earlier lanes of the same voxel keeps the serial order without a serial loop.
__global__ void k_roiaware_collect(int N, int P, int ox, int oy, int oz, int maxpts,
                                   const float* __restrict__ rois, const float* __restrict__ pts,
                                   int* __restrict__ pts_idx_of_voxels) {
  extern __shared__ int s_cnt[];   // per-voxel fill count of this box (wave-private block)
  const int box = blockIdx.x;
  const int lane = threadIdx.x;
  const int nvox = ox * oy * oz;
  BoxT bx;
  bx.load(rois + (long long)box * 7);
  int* lists = pts_idx_of_voxels + (long long)box * nvox * maxpts;
  for (int v = lane; v < nvox; v += 64) s_cnt[v] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float x_res = bx.dx / ox, y_res = bx.dy / oy, z_res = bx.dz / oz;
  for (int base = 0; base < P; base += 64) {
    int k = base + lane;
    int vox = -1;
    if (k < P) {
      float x = pts[(long long)k * 3], y = pts[(long long)k * 3 + 1], z = pts[(long long)k * 3 + 2];
      float lx = 0.f, ly = 0.f;
      if (bx.contains(x, y, z, PIB_MARGIN, lx, ly)) {
        float lz = z - bx.cz;
        unsigned xi = (unsigned)(int)((lx + bx.dx / 2) / x_res);
        unsigned yi = (unsigned)(int)((ly + bx.dy / 2) / y_res);
        unsigned zi = (unsigned)(int)((lz + bx.dz / 2) / z_res);
        xi = min(xi, (unsigned)(ox - 1)) & 0xFF;   // min(max(x,0),o-1) on unsigned, 8-bit fields
        yi = min(yi, (unsigned)(oy - 1)) & 0xFF;
        zi = min(zi, (unsigned)(oz - 1)) & 0xFF;
        vox = (int)((xi * oy + yi) * oz + zi);
      }
    }
    unsigned long long inside = __ballot(vox >= 0);
    if (!inside) continue;
    // rank among earlier lanes with the same voxel, and whether I am the last of my voxel
    int rank = 0;
    bool last = true;
    for (unsigned long long m = inside; m;) {
      int l = __ffsll((long long)m) - 1;
      m &= m - 1;
      int v = __shfl(vox, l, 64);
      if (v == vox) {
        if (l < lane) ++rank;
        if (l > lane) last = false;
      }
    }
    int cnt = vox >= 0 ? s_cnt[vox] : 0;      // count before this batch (all lanes read first)
    __builtin_amdgcn_wave_barrier();
    if (vox >= 0) {
      int slot = cnt + rank;
      if (slot < maxpts - 1) lists[(long long)vox * maxpts + slot + 1] = k;
      if (last) s_cnt[vox] = min(cnt + rank + 1, maxpts - 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  for (int v = lane; v < nvox; v += 64) lists[(long long)v * maxpts] = s_cnt[v];
}

// roiaware_maxpool3d / roiaware_avgpool3d, kernel.cu:111-190: thread per (box, voxel, channel)
__global__ void k_roiaware_pool(int N, int C, int maxpts, int nvox, int method,
                                const float* __restrict__ feat, const int* __restrict__ lists,
                                float* __restrict__ pooled, int* __restrict__ argmax) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * nvox * C) return;
  int c = (int)(t % C);
  long long v = t / C;
  const int* lst = lists + v * maxpts;
  int total = lst[0];
  if (method == 0) {
    int am = -1;
    float mv = -INFINITY;
    for (int k = 1; k <= total; ++k) {
      float f = feat[(long long)lst[k] * C + c];
      if (f > mv) { mv = f; am = lst[k]; }
    }
    pooled[t] = am != -1 ? mv : 0.f;          // every element written: the caller need not zero-fill 180 MB
    argmax[t] = am;
  } else {
    float s = 0.f;
    for (int k = 1; k <= total; ++k) s += feat[(long long)lst[k] * C + c];
    pooled[t] = total > 0 ? s / total : 0.f;
  }
}

extern "C" int glx_roiaware_pool3d_forward(const float* rois, int N, const float* pts, int P,
                                           const float* pts_feature, int C, int ox, int oy, int oz,
                                           int max_pts, int pool_method, int32_t* argmax,
                                           int32_t* pts_idx_of_voxels, float* pooled,
                                           void* stream) {
  if (N == 0) return GLX_OK;
  GLX_REQUIRE(rois && pts_idx_of_voxels && pooled && argmax, "glx_roiaware_pool3d_forward: null pointer");
  GLX_REQUIRE(ox > 0 && oy > 0 && oz > 0 && ox < 256 && oy < 256 && oz < 256 && max_pts >= 2,
              "glx_roiaware_pool3d_forward: out size must be < 256 per axis, max_pts >= 2");
  hipStream_t st = (hipStream_t)stream;
  GLX_REQUIRE((size_t)ox * oy * oz * 4 <= 64 * 1024, "glx_roiaware_pool3d_forward: %d voxels per box",
              ox * oy * oz);
  if (P > 0)
    hipLaunchKernelGGL(k_roiaware_collect, dim3(N), dim3(64), (size_t)ox * oy * oz * 4, st, N, P,
                       ox, oy, oz, max_pts, rois, pts, pts_idx_of_voxels);
  long long total = (long long)N * ox * oy * oz * C;
  hipLaunchKernelGGL(k_roiaware_pool, dim3(glx_divup(total, 256)), dim3(256), 0, st, N, C, max_pts,
                     ox * oy * oz, pool_method, pts_feature, (const int*)pts_idx_of_voxels, pooled,
                     argmax);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// roiaware_{max,avg}pool3d_backward, kernel.cu:236-286 (float atomics, like the reference)
__global__ void k_roiaware_backward(long long nvoxC, int C, int maxpts, int method,
                                    const int* __restrict__ lists, const int* __restrict__ argmax,
                                    const float* __restrict__ grad_out, float* __restrict__ grad_in) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nvoxC) return;
  int c = (int)(t % C);
  long long v = t / C;
  float g = grad_out[t];
  if (method == 0) {
    int a = argmax[t];
    if (a == -1) return;
    atomicAdd(grad_in + (long long)a * C + c, g * 1);
  } else {
    const int* lst = lists + v * maxpts;
    int total = lst[0];
    float cur = 1 / fmaxf((float)total, 1.0f);
    for (int k = 1; k <= total; ++k) atomicAdd(grad_in + (long long)lst[k] * C + c, g * cur);
  }
}

extern "C" int glx_roiaware_pool3d_backward(const int32_t* pts_idx_of_voxels,
                                            const int32_t* argmax, const float* grad_out, int N,
                                            int ox, int oy, int oz, int C, int max_pts,
                                            int pool_method, float* grad_in, void* stream) {
  if (N == 0) return GLX_OK;
  GLX_REQUIRE(pts_idx_of_voxels && argmax && grad_out && grad_in, "glx_roiaware_pool3d_backward: null");
  long long total = (long long)N * ox * oy * oz * C;
  hipLaunchKernelGGL(k_roiaware_backward, dim3(glx_divup(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, total, C, max_pts, pool_method,
                     (const int*)pts_idx_of_voxels, (const int*)argmax, grad_out, grad_in);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ RoI point pooling
// assign_pts_to_box3d + get_pooled_idx + roipool3d_forward (roipoint_pool3d_kernel.cu:38-134):
// one wave per (frame, box) collects the first S inside points in point order by ballot
// compaction, wraps them around if fewer, then gathers xyz + features.
__global__ void k_roipoint_pool(int B, int Np, int M, int C, int S, const float* __restrict__ xyz,
                                const float* __restrict__ boxes, const float* __restrict__ feat,
                                float* __restrict__ pooled, int* __restrict__ empty_flag) {
  extern __shared__ int s_idx[];   // S
  __shared__ int s_cnt;
  const int m = blockIdx.x, b = blockIdx.y, lane = threadIdx.x & 63;
  const float* X = xyz + (long long)b * Np * 3;
  if (threadIdx.x < 64) {          // the ordered scan is one wave's work; the gather below is the block's
    BoxT bx;
    bx.load(boxes + ((long long)b * M + m) * 7);
    int cnt = 0;
    for (int base = 0; base < Np && cnt < S; base += 64) {
      int k = base + lane;
      bool in = false;
      if (k < Np) {
        float lx, ly;
        in = bx.contains(X[(long long)k * 3], X[(long long)k * 3 + 1], X[(long long)k * 3 + 2],
                         PIB_MARGIN, lx, ly);
      }
      unsigned long long bal = __ballot(in);
      int pos = cnt + __popcll(bal & ((1ull << lane) - 1ull));
      if (in && pos < S) s_idx[pos] = k;
      cnt += __popcll(bal);
    }
    if (lane == 0) s_cnt = cnt;
  }
  __syncthreads();
  int cnt = s_cnt;
  if (cnt == 0) {
    if (threadIdx.x == 0) empty_flag[(long long)b * M + m] = 1;
    return;
  }
  if (cnt > S) cnt = S;
  const int W = 3 + C;
  float* dst = pooled + ((long long)b * M + m) * S * W;
  for (int e = threadIdx.x; e < S * W; e += blockDim.x) {
    int s = e / W, j = e - s * W;
    int src = s_idx[s < cnt ? s : s % cnt];
    dst[e] = j < 3 ? X[(long long)src * 3 + j] : feat[((long long)b * Np + src) * C + (j - 3)];
  }
}

extern "C" int glx_roipoint_pool3d(const float* xyz, const float* boxes3d, const float* pts_feature,
                                   int B, int Np, int M, int C, int S, float* pooled,
                                   int32_t* empty_flag, void* stream) {
  if (B == 0 || M == 0) return GLX_OK;
  GLX_REQUIRE(xyz && boxes3d && pooled && empty_flag && (C == 0 || pts_feature),
              "glx_roipoint_pool3d: null pointer");
  GLX_REQUIRE(S > 0 && (size_t)S * 4 <= 64 * 1024, "glx_roipoint_pool3d: bad sample count %d", S);
  hipLaunchKernelGGL(k_roipoint_pool, dim3(M, B), dim3(256), S * sizeof(int), (hipStream_t)stream,
                     B, Np, M, C, S, xyz, boxes3d, pts_feature, pooled, empty_flag);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ voxel query
// voxel_query_kernel_stack, pointnet2_stack/src/voxel_query_gpu.cu:10-89: the neighbour window is
// scanned z, y, x ascending, the first nsample cells whose centre lies within the radius are kept,
// unused slots repeat the first hit, -1 in slot 0 marks an empty ball.  MAP = dense (B,Z,Y,X)
// int32 map (reference API) or the rank dictionary of the sparse tensor (no 189 MB map to build
// and clear per scale).
// The reference walks the (2r+1)^3 window with one thread per grid point: 729 dependent lookups in
// a row and 1.7 waves per SIMD at the Voxel-RCNN sizes -- latency-bound.  Here G lanes share a grid
// point and test G consecutive window cells per step; the scan order is kept by ranking the hits
// of a step with a ballot (slot = hits so far + hits in lower lanes), so the output is identical.
// CEN: the neighbour positions are not read from an xyz array but rebuilt from the sparse tensor's
// own (N,4) [b,z,y,x] indices, rounding step by step like get_voxel_centers
// (pcdet/utils/common_utils.py:66-82): (i + 0.5) * (voxel * stride) + range_min -- one 16-byte
// gather instead of three 4-byte ones and no centres tensor.  new_coords are then the STRIDE-1
// voxel coordinates of the grid points and are floor-divided by coord_stride here
// (voxelrcnn_head.py:167).
struct VoxelCentres {
  const int* indices;
  float vsx, vsy, vsz, r0x, r0y, r0z;
};

__device__ __forceinline__ void glx_voxel_centre(const VoxelCentres& g, long long row, float& x,
                                                 float& y, float& z) {
  const int4 c = reinterpret_cast<const int4*>(g.indices)[row];      // b z y x
  x = __fadd_rn(__fmul_rn(__fadd_rn((float)c.w, 0.5f), g.vsx), g.r0x);
  y = __fadd_rn(__fmul_rn(__fadd_rn((float)c.z, 0.5f), g.vsy), g.r0y);
  z = __fadd_rn(__fmul_rn(__fadd_rn((float)c.y, 0.5f), g.vsz), g.r0z);
}

__device__ __forceinline__ int glx_floordiv(int a, int b) {          // b > 0
  return a >= 0 ? a / b : -((-a + b - 1) / b);
}

template <bool DENSE, int G, bool CEN>
__global__ __launch_bounds__(256) void k_voxel_query(
    int M, int R1, int R2, int R3, int nsample, float radius2, int zr, int yr, int xr,
    const float* __restrict__ new_xyz, const float* __restrict__ xyz, const int* __restrict__ new_coords,
    const int* __restrict__ point_indices, const unsigned long long* __restrict__ bitmap,
    const int* __restrict__ prefix, const int* __restrict__ rank_to_row, int* __restrict__ idx,
    VoxelCentres cen, int coord_stride) {
  const int lane = threadIdx.x & 63;
  const int g = lane % G, sub = lane / G;
  const long long pt = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 / G) + sub;
  const bool live = pt < M;
  const int wy = 2 * yr + 1, wx = 2 * xr + 1;
  const int W = (2 * zr + 1) * wy * wx;
  const float inv_x = 1.f / wx, inv_y = 1.f / wy;
  float nx = 0.f, ny = 0.f, nz = 0.f;
  int4 nc = make_int4(0, 0, 0, 0);                                   // b z y x
  if (live) {
    nx = new_xyz[pt * 3], ny = new_xyz[pt * 3 + 1], nz = new_xyz[pt * 3 + 2];
    nc = reinterpret_cast<const int4*>(new_coords)[pt];
    if (CEN && coord_stride > 1) {
      nc.y = glx_floordiv(nc.y, coord_stride);
      nc.z = glx_floordiv(nc.z, coord_stride);
      nc.w = glx_floordiv(nc.w, coord_stride);
    }
  }
  int* o = idx + (live ? pt : 0) * nsample;
  int cnt = live ? 0 : nsample, first = -1;
  for (int base = 0; base < W; base += G) {
    if (!__any(cnt < nsample)) break;
    const int w = base + g;
    int nb = -1;
    if (cnt < nsample && w < W) {
      const int t = (int)((w + 0.5f) * inv_x);                       // w / wx (exact: w < 2^20)
      const int q = (int)((t + 0.5f) * inv_y);
      const int z = nc.y + q - zr, y = nc.z + (t - q * wy) - yr, x = nc.w + (w - t * wx) - xr;
      if (z >= 0 && z < R1 && y >= 0 && y < R2 && x >= 0 && x < R3) {
        const long long lin = (((long long)nc.x * R1 + z) * R2 + y) * R3 + x;
        if (DENSE) {
          nb = point_indices[lin];
        } else {
          nb = glx_rank_lookup(bitmap, prefix, lin);
          if (nb >= 0 && rank_to_row) nb = rank_to_row[nb];
        }
        if (nb >= 0) {
          float xp, yp, zp;
          if (CEN) {
            glx_voxel_centre(cen, nb, xp, yp, zp);
          } else {
            xp = xyz[(long long)nb * 3], yp = xyz[(long long)nb * 3 + 1], zp = xyz[(long long)nb * 3 + 2];
          }
          const float d2 = (xp - nx) * (xp - nx) + (yp - ny) * (yp - ny) + (zp - nz) * (zp - nz);
          if (d2 > radius2) nb = -1;
        }
      }
    }
    const unsigned long long hits = __ballot(nb >= 0);
    const unsigned bits = (unsigned)(hits >> (sub * G)) & ((1u << G) - 1u);
    const int lead = __shfl(nb, bits ? sub * G + __ffs(bits) - 1 : lane, 64);
    if (bits) {
      const int pos = cnt + __popc(bits & ((1u << g) - 1u));
      if (nb >= 0 && pos < nsample) o[pos] = nb;
      if (first < 0) first = lead;
      cnt += __popc(bits);
    }
  }
  if (!live) return;
  if (first < 0) {
    if (g == 0) o[0] = -1;
  } else {
    for (int l = cnt + g; l < nsample; l += G) o[l] = first;
  }
}

// Row-wise scan over the rank dictionary: the (2xr+1) x-cells of one (dz, dy) window row are
// consecutive bits of the occupancy bitmap, so a lane fetches a whole row with one or two 64-bit
// loads instead of 2xr+1 dictionary probes, and the (mostly empty) rows cost nothing further.
// G = 8 lanes share a grid point and take 8 consecutive rows per step; the reference's z, y, x
// scan order is kept by ranking a step's hits with a prefix sum of the per-row hit counts over the
// 8 lanes.  The radius test runs on the occupied cells only; with CEN the centre follows from the
// probed (z, y, x) itself -- the dictionary is keyed by the tensor's own indices, so it is the value
// glx_voxel_centre would gather -- and the rank -> row lookup is done just for the hits that are
// written.  Same output as k_voxel_query, 81 row fetches instead of 729 probes at the Voxel-RCNN
// ranges [4, 4, 4].
constexpr int VQ_ROW_MAX_WX = 32;

template <bool CEN>
__device__ __forceinline__ void vq_rows_body(
    int M, int R1, int R2, int R3, int nsample, float radius2, int zr, int yr, int xr,
    const float* __restrict__ new_xyz, const float* __restrict__ xyz, const int* __restrict__ new_coords,
    const unsigned long long* __restrict__ bitmap, const int* __restrict__ prefix,
    const int* __restrict__ rank_to_row, int* __restrict__ idx, VoxelCentres cen, int coord_stride) {
  constexpr int G = 8;
  const int lane = threadIdx.x & 63;
  const int g = lane % G, sub = lane / G;
  const long long pt = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 / G) + sub;
  const bool live = pt < M;
  const int wy = 2 * yr + 1;
  const int rows = (2 * zr + 1) * wy;
  const float inv_y = 1.f / wy;
  float nx = 0.f, ny = 0.f, nz = 0.f;
  int4 nc = make_int4(0, 0, 0, 0);                                   // b z y x
  if (live) {
    nx = new_xyz[pt * 3], ny = new_xyz[pt * 3 + 1], nz = new_xyz[pt * 3 + 2];
    nc = reinterpret_cast<const int4*>(new_coords)[pt];
    if (CEN && coord_stride > 1) {
      nc.y = glx_floordiv(nc.y, coord_stride);
      nc.z = glx_floordiv(nc.z, coord_stride);
      nc.w = glx_floordiv(nc.w, coord_stride);
    }
  }
  int* o = idx + (live ? pt : 0) * nsample;
  const int x_lo = max(nc.w - xr, 0), x_hi = min(nc.w + xr, R3 - 1);
  int cnt = live ? 0 : nsample, first = -1;
  for (int base = 0; base < rows; base += G) {
    if (!__any(cnt < nsample)) break;
    const int r = base + g;
    unsigned mask = 0;
    long long lin0 = 0;
    if (cnt < nsample && r < rows && x_lo <= x_hi) {
      const int q = (int)((r + 0.5f) * inv_y);                       // r / wy
      const int z = nc.y + q - zr, y = nc.z + (r - q * wy) - yr;
      if (z >= 0 && z < R1 && y >= 0 && y < R2) {
        lin0 = (((long long)nc.x * R1 + z) * R2 + y) * R3 + x_lo;
        const int n = x_hi - x_lo + 1, s = (int)(lin0 & 63);
        unsigned long long bits = bitmap[lin0 >> 6] >> s;
        if (s + n > 64) bits |= bitmap[(lin0 >> 6) + 1] << (64 - s);
        mask = (unsigned)bits & (n >= 32 ? 0xffffffffu : ((1u << n) - 1u));
        float yp = 0.f, zp = 0.f;
        if (CEN) {
          yp = __fadd_rn(__fmul_rn(__fadd_rn((float)y, 0.5f), cen.vsy), cen.r0y);
          zp = __fadd_rn(__fmul_rn(__fadd_rn((float)z, 0.5f), cen.vsz), cen.r0z);
        }
        for (unsigned m = mask; m; m &= m - 1) {
          const int j = __ffs(m) - 1;
          float xp;
          if (CEN) {
            xp = __fadd_rn(__fmul_rn(__fadd_rn((float)(x_lo + j), 0.5f), cen.vsx), cen.r0x);
          } else {
            int nb = glx_rank_lookup(bitmap, prefix, lin0 + j);
            if (rank_to_row) nb = rank_to_row[nb];
            xp = xyz[(long long)nb * 3], yp = xyz[(long long)nb * 3 + 1], zp = xyz[(long long)nb * 3 + 2];
          }
          const float d2 = (xp - nx) * (xp - nx) + (yp - ny) * (yp - ny) + (zp - nz) * (zp - nz);
          if (d2 > radius2) mask &= ~(1u << j);
        }
      }
    }
    const int c = __popc(mask);
    int incl = c;
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
      const int t = __shfl_up(incl, d, G);
      if (g >= d) incl += t;
    }
    const int total = __shfl(incl, G - 1, G);
    const int excl = incl - c;
    int lead = -1;                                                   // first hit of the scan, for the padding
    if (first < 0 && c > 0 && excl == 0) {
      lead = glx_rank_lookup(bitmap, prefix, lin0 + __ffs(mask) - 1);
      if (rank_to_row) lead = rank_to_row[lead];
    }
#pragma unroll
    for (int d = 1; d < G; d <<= 1) lead = max(lead, __shfl_xor(lead, d, G));
    if (first < 0) first = lead;
    int pos = cnt + excl;
    for (unsigned m = mask; m && pos < nsample; m &= m - 1, ++pos) {
      int nb = glx_rank_lookup(bitmap, prefix, lin0 + __ffs(m) - 1);
      if (rank_to_row) nb = rank_to_row[nb];
      o[pos] = nb;
    }
    cnt += total;
  }
  if (!live) return;
  if (first < 0) {
    if (g == 0) o[0] = -1;
  } else {
    for (int l = cnt + g; l < nsample; l += G) o[l] = first;
  }
}

template <bool CEN>
__global__ __launch_bounds__(256) void k_voxel_query_rows(
    int M, int R1, int R2, int R3, int nsample, float radius2, int zr, int yr, int xr,
    const float* __restrict__ new_xyz, const float* __restrict__ xyz, const int* __restrict__ new_coords,
    const unsigned long long* __restrict__ bitmap, const int* __restrict__ prefix,
    const int* __restrict__ rank_to_row, int* __restrict__ idx, VoxelCentres cen, int coord_stride) {
  vq_rows_body<CEN>(M, R1, R2, R3, nsample, radius2, zr, yr, xr, new_xyz, xyz, new_coords, bitmap, prefix, rank_to_row, idx, cen,
                    coord_stride);
}

// The RoI grid's queries of several feature scales (same grid points, each scale its own tensor) in ONE launch:
// blockIdx.y = scale.  The scales' launches were a serial chain of 43-46 us each on the RoI branch.
#define VQ_MAX_SCALES 4
struct VqScale {
  int R1, R2, R3, nsample, zr, yr, xr, stride;
  float radius2;
  const unsigned long long* bitmap;
  const int* prefix;
  const int* rank_to_row;
  int* idx;
  VoxelCentres cen;
};
struct VqScales { VqScale s[VQ_MAX_SCALES]; };

__global__ __launch_bounds__(256) void k_voxel_query_rows_multi(int M, const float* __restrict__ new_xyz,
                                                                const int* __restrict__ new_coords, VqScales q) {
  const VqScale& a = q.s[blockIdx.y];
  vq_rows_body<true>(M, a.R1, a.R2, a.R3, a.nsample, a.radius2, a.zr, a.yr, a.xr, new_xyz, nullptr, new_coords, a.bitmap,
                     a.prefix, a.rank_to_row, a.idx, a.cen, a.stride);
}

extern "C" int glx_voxel_query(int M, int Z, int Y, int X, int nsample, float radius, int z_range,
                               int y_range, int x_range, const float* new_xyz, const float* xyz,
                               const int32_t* new_coords, const int32_t* point_indices,
                               int32_t* idx, void* stream) {
  if (M == 0) return GLX_OK;
  GLX_REQUIRE(new_xyz && xyz && new_coords && point_indices && idx, "glx_voxel_query: null pointer");
  hipLaunchKernelGGL((k_voxel_query<true, 8, false>), dim3(glx_divup(M, 32)), dim3(256), 0,
                     (hipStream_t)stream, M, Z, Y, X, nsample, radius * radius, z_range, y_range,
                     x_range, new_xyz, xyz, new_coords, point_indices,
                     (const unsigned long long*)nullptr, (const int*)nullptr, (const int*)nullptr,
                     idx, VoxelCentres{}, 1);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_voxel_query_index(int M, int Z, int Y, int X, int nsample, float radius,
                                     int z_range, int y_range, int x_range, const float* new_xyz,
                                     const float* xyz, const int32_t* new_coords,
                                     const uint64_t* bitmap, const int32_t* prefix,
                                     const int32_t* rank_to_row, int32_t* idx, void* stream) {
  if (M == 0) return GLX_OK;
  GLX_REQUIRE(new_xyz && xyz && new_coords && bitmap && prefix && idx, "glx_voxel_query_index: null");
  if (2 * x_range + 1 <= VQ_ROW_MAX_WX)
    hipLaunchKernelGGL((k_voxel_query_rows<false>), dim3(glx_divup(M, 32)), dim3(256), 0,
                       (hipStream_t)stream, M, Z, Y, X, nsample, radius * radius, z_range, y_range,
                       x_range, new_xyz, xyz, new_coords, (const unsigned long long*)bitmap,
                       (const int*)prefix, rank_to_row, idx, VoxelCentres{}, 1);
  else
    hipLaunchKernelGGL((k_voxel_query<false, 8, false>), dim3(glx_divup(M, 32)), dim3(256), 0,
                       (hipStream_t)stream, M, Z, Y, X, nsample, radius * radius, z_range, y_range,
                       x_range, new_xyz, xyz, new_coords, (const int*)nullptr,
                       (const unsigned long long*)bitmap, (const int*)prefix, rank_to_row, idx,
                       VoxelCentres{}, 1);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ ball query
// ball_query_kernel_stack, pointnet2_stack/src/ball_query_gpu.cu:16-65 (strict d2 < r2; indices
// local to the query's frame).
__device__ __forceinline__ int batch_of(int pt, const int* cnt, int B, int& start_other,
                                        const int* other_cnt) {
  int bs = 0, pc = cnt[0];
  for (int k = 1; k < B; k++) {
    if (pt < pc) break;
    pc += cnt[k];
    bs = k;
  }
  start_other = 0;
  for (int k = 0; k < bs; k++) start_other += other_cnt[k];
  return bs;
}

// A wave per query: the lanes test 64 points at a time, a ballot keeps the hits in index order (the first
// nsample points inside the radius, remaining slots = the first hit, idx[m,0] = -1 for an empty ball:
// ball_query_gpu.cu:16-67).  The reference's thread-per-query scan leaves 2048 x B queries on 128
// waves, each walking 16 K points one by one (2.07 ms at 4 x 2048 x 16384; this form: see tools/ops_time.py).
__global__ __launch_bounds__(256) void k_ball_query(int B, int M, float radius2, int nsample,
                                                    const float* __restrict__ new_xyz, const int* __restrict__ new_cnt,
                                                    const float* __restrict__ xyz, const int* __restrict__ xyz_cnt,
                                                    int* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int pt = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pt >= M) return;
  int start;
  int bs = batch_of(pt, new_cnt, B, start, xyz_cnt);
  const float* X = xyz + (long long)start * 3;
  const float nx = new_xyz[(long long)pt * 3], ny = new_xyz[(long long)pt * 3 + 1],
              nz = new_xyz[(long long)pt * 3 + 2];
  int* o = idx + (long long)pt * nsample;
  const int n = xyz_cnt[bs];
  int cnt = 0, first = -1;
  for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
    const int k = k0 + lane;
    bool in = false;
    if (k < n) {
      float x = X[(long long)k * 3], y = X[(long long)k * 3 + 1], z = X[(long long)k * 3 + 2];
      float d2 = (nx - x) * (nx - x) + (ny - y) * (ny - y) + (nz - z) * (nz - z);
      in = d2 < radius2;
    }
    const unsigned long long hits = __ballot(in);
    if (hits) {
      if (first < 0) first = k0 + __ffsll((long long)hits) - 1;
      const int pos = cnt + __popcll(hits & ((1ull << lane) - 1ull));
      if (in && pos < nsample) o[pos] = k;
      cnt += __popcll(hits);
    }
  }
  if (first < 0) {
    if (lane == 0) o[0] = -1;
  } else {
    for (int l = (cnt < nsample ? cnt : nsample) + lane; l < nsample; l += 64) o[l] = first;
  }
}

extern "C" int glx_ball_query(int B, int M, float radius, int nsample, const float* new_xyz,
                              const int32_t* new_xyz_batch_cnt, const float* xyz,
                              const int32_t* xyz_batch_cnt, int32_t* idx, void* stream) {
  if (M == 0) return GLX_OK;
  GLX_REQUIRE(new_xyz && new_xyz_batch_cnt && xyz && xyz_batch_cnt && idx, "glx_ball_query: null");
  hipLaunchKernelGGL(k_ball_query, dim3(glx_divup(M, 4)), dim3(256), 0, (hipStream_t)stream, B,
                     M, radius * radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt,
                     idx);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ grouping
// group_points_kernel_stack, pointnet2_stack/src/group_points_gpu.cu:71-101:
//   out[m, c, s] = features[start(batch of m) + idx[m, s], c]
// one wave per query: feature rows are read as whole rows (coalesced), transposed through LDS,
// and written as whole (C, nsample) tiles.
__global__ void k_group_points(int B, int M, int C, int ns, const float* __restrict__ feat,
                               const int* __restrict__ feat_cnt, const int* __restrict__ idx,
                               const int* __restrict__ idx_cnt, float* __restrict__ out) {
  extern __shared__ float tile[];   // per wave: ns * (C + 1)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * (blockDim.x >> 6) + wave;
  if (m >= M) return;
  float* tl = tile + (size_t)wave * ns * (C + 1);
  int start;
  batch_of(m, idx_cnt, B, start, feat_cnt);
  const int* id = idx + (long long)m * ns;
  for (int e = lane; e < ns * C; e += 64) {
    int s = e / C, c = e - s * C;
    tl[s * (C + 1) + c] = feat[((long long)start + id[s]) * C + c];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float* o = out + (long long)m * C * ns;
  for (int e = lane; e < ns * C; e += 64) {
    int c = e / ns, s = e - c * ns;
    o[e] = tl[s * (C + 1) + c];
  }
}

extern "C" int glx_group_points(int B, int M, int C, int nsample, const float* features,
                                const int32_t* features_batch_cnt, const int32_t* idx,
                                const int32_t* idx_batch_cnt, float* out, void* stream) {
  if (M == 0 || C == 0) return GLX_OK;
  GLX_REQUIRE(features && features_batch_cnt && idx && idx_batch_cnt && out, "glx_group_points: null");
  size_t per_wave = (size_t)nsample * (C + 1) * sizeof(float);
  GLX_REQUIRE(per_wave <= 40 * 1024, "glx_group_points: nsample*C = %d too large", nsample * C);
  int waves = per_wave * 4 <= 64 * 1024 ? 4 : 1;
  hipLaunchKernelGGL(k_group_points, dim3(glx_divup(M, waves)), dim3(64 * waves), per_wave * waves,
                     (hipStream_t)stream, B, M, C, nsample, features, features_batch_cnt, idx,
                     idx_batch_cnt, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// group_points_grad_kernel_stack, group_points_gpu.cu:15-44: atomicAdd scatter into (N, C);
// lanes walk channels fastest so each atomic wave-instruction covers contiguous channels of
// one feature row (cdna_hip_programming.md Guideline 12).
__global__ void k_group_points_grad(int B, int M, int C, int ns, const float* __restrict__ grad_out,
                                    const int* __restrict__ idx, const int* __restrict__ idx_cnt,
                                    const int* __restrict__ feat_cnt,
                                    float* __restrict__ grad_features) {
  extern __shared__ float tile[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * (blockDim.x >> 6) + wave;
  if (m >= M) return;
  float* tl = tile + (size_t)wave * ns * (C + 1);
  int start;
  batch_of(m, idx_cnt, B, start, feat_cnt);
  const float* g = grad_out + (long long)m * C * ns;
  for (int e = lane; e < ns * C; e += 64) {
    int c = e / ns, s = e - c * ns;
    tl[s * (C + 1) + c] = g[e];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int* id = idx + (long long)m * ns;
  for (int e = lane; e < ns * C; e += 64) {
    int s = e / C, c = e - s * C;
    atomicAdd(grad_features + ((long long)start + id[s]) * C + c, tl[s * (C + 1) + c]);
  }
}

extern "C" int glx_group_points_grad(int B, int M, int C, int N, int nsample, const float* grad_out,
                                     const int32_t* idx, const int32_t* idx_batch_cnt,
                                     const int32_t* features_batch_cnt, float* grad_features,
                                     void* stream) {
  (void)N;
  if (M == 0 || C == 0) return GLX_OK;
  GLX_REQUIRE(grad_out && idx && idx_batch_cnt && features_batch_cnt && grad_features,
              "glx_group_points_grad: null");
  size_t per_wave = (size_t)nsample * (C + 1) * sizeof(float);
  GLX_REQUIRE(per_wave <= 40 * 1024, "glx_group_points_grad: nsample*C = %d too large", nsample * C);
  int waves = per_wave * 4 <= 64 * 1024 ? 4 : 1;
  hipLaunchKernelGGL(k_group_points_grad, dim3(glx_divup(M, waves)), dim3(64 * waves),
                     per_wave * waves, (hipStream_t)stream, B, M, C, nsample, grad_out, idx,
                     idx_batch_cnt, features_batch_cnt, grad_features);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// Gather form of the same gradient.  The atomic scatter above collapses on the RoI-grid pooling of
// Voxel-RCNN: 100+ RoIs per frame sit on the same few objects, every voxel row there receives
// thousands of contributions and the memory-side atomic unit serialises them (20.8 ms for 86 400
// grid points x 16 x 32 channels = 44 M atomic adds; the atomic-rate table of MI355X_MICROARCH.md has
// the same collapse for one hot row).  Here the (grid point, slot) references are bucketed by feature
// row first -- a count, an exclusive scan and a fill, 2 x M*ns integer atomics instead of M*ns*C
// float ones -- and each row then sums its references itself: a wave per row, lanes = channels x
// reference slots, no float atomic at all.  (The order of a row's references comes from integer
// atomics, so sums may differ in the last bits between runs, as with the atomic scatter.)
// count / fill: the 64 references of a wave (4 grid points x 16 slots) name few distinct rows (unused
// slots repeat the first hit, neighbouring grid points share voxels), so equal rows are combined
// inside the wave first -- one integer atomic per distinct row and wave instead of one per
// reference (the per-reference version spent 7 / 12 ms in these two kernels on the hot rows).
// idx_cnt == NULL: idx holds GLOBAL rows as the voxel query leaves them (idx[m,0] < 0 = empty ball,
// whose slots reference nothing); else per-frame rows as grouping_operation takes them.
__global__ void k_gp_count(int B, int M, int ns, const int* __restrict__ idx, const int* __restrict__ idx_cnt,
                           const int* __restrict__ feat_cnt, int* __restrict__ cnt) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = t < (long long)M * ns;
  int row = -1;
  if (valid) {
    if (idx_cnt) {
      int start;
      batch_of((int)(t / ns), idx_cnt, B, start, feat_cnt);
      row = start + idx[t];
    } else {
      valid = idx[t / ns * ns] >= 0;
      row = idx[t];
    }
  }
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int r = __shfl(row, leader, 64);
    const unsigned long long same = __ballot(valid && row == r);
    if (lane == leader) atomicAdd(&cnt[r], __popcll(same));
    todo &= ~same;
  }
}

__global__ void k_gp_fill(int B, int M, int ns, const int* __restrict__ idx, const int* __restrict__ idx_cnt,
                          const int* __restrict__ feat_cnt, const int* __restrict__ offs,
                          int* __restrict__ cursor, int* __restrict__ refs) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = t < (long long)M * ns;
  int row = -1;
  if (valid) {
    if (idx_cnt) {
      int start;
      batch_of((int)(t / ns), idx_cnt, B, start, feat_cnt);
      row = start + idx[t];
    } else {
      valid = idx[t / ns * ns] >= 0;
      row = idx[t];
    }
  }
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int r = __shfl(row, leader, 64);
    const unsigned long long same = __ballot(valid && row == r);
    int base = 0;
    if (lane == leader) base = atomicAdd(&cursor[r], __popcll(same));
    base = __shfl(base, leader, 64);
    if (valid && row == r)
      refs[offs[r] + base + __popcll(same & ((1ull << lane) - 1ull))] = (int)t;   // reference = m * ns + s
    todo &= ~same;
  }
}

// One wave per CHUNK of 64 consecutive references (they are sorted by row): uniform work per wave
// whatever the distribution -- a wave per row would leave the hot rows (tens of thousands of
// references) to a single wave.  Inside a chunk the references of one row are summed in registers
// (lane = channel x reference slot) and flushed with one atomic add per (row, channel); a row that
// spans k chunks receives k adds instead of one per reference.
#define GP_CHUNK 64
__global__ __launch_bounds__(256) void k_gp_gather(int N, int C, int ns, int rows_layout,
                                                   const float* __restrict__ grad_out,
                                                   const int* __restrict__ offs, const int* __restrict__ refs,
                                                   float* __restrict__ grad_features) {
  const int lane = threadIdx.x & 63;
  const long long p0 = ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * GP_CHUNK;
  const long long T = offs[N];                 // references actually bucketed (empty balls hold none)
  if (p0 >= T) return;
  const long long p1 = p0 + GP_CHUNK < T ? p0 + GP_CHUNK : T;
  const int CL = (C < 64 && (C & (C - 1)) == 0) ? C : 64;   // channels covered by one pass of the wave
  const int J = 64 / CL;                                     // references in flight per pass
  const int c0 = lane % CL, j = lane / CL;
  int lo = 0, hi = N;                                        // row of reference p0: last row with offs[row] <= p0
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((long long)offs[mid] <= p0) lo = mid; else hi = mid;
  }
  int row = lo;
  long long e0 = p0;
  while (e0 < p1) {
    while ((long long)offs[row + 1] <= e0) ++row;            // skip rows without references
    const long long e1 = (long long)offs[row + 1] < p1 ? (long long)offs[row + 1] : p1;
    for (int cb = 0; cb < C; cb += CL) {
      const int c = cb + c0;
      float acc = 0.f;
      if (c < C && j < J) {
        for (long long e = e0 + j; e < e1; e += J) {
          const int r = refs[e];
          if (rows_layout) {                    // grad_out (M, ns, C): a reference's channels are contiguous
            acc += grad_out[(long long)r * C + c];
          } else {                              // grad_out (M, C, ns) as grouping_operation produces it
            const int m = r / ns, sl = r - m * ns;
            acc += grad_out[((long long)m * C + c) * ns + sl];
          }
        }
      }
      for (int d = CL; d < 64; d <<= 1) acc += __shfl_xor(acc, d, 64);
      if (j == 0 && c < C) atomicAdd(grad_features + (long long)row * C + c, acc);
    }
    e0 = e1;
  }
}

extern "C" size_t glx_group_points_grad_workspace_bytes(int M, int N, int nsample) {
  return glx_align((size_t)(N + 1) * 4) * 3 + glx_align((size_t)M * nsample * 4) +
         glx_scan_workspace_bytes(N + 1) + 256;
}

static int gp_grad_gather_impl(int B, int M, int C, int N, int nsample, const float* grad_out,
                               const int32_t* idx, const int32_t* idx_batch_cnt,
                               const int32_t* features_batch_cnt, float* grad_features, int rows_layout,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (N <= 0 || C <= 0) return GLX_OK;
  GLX_REQUIRE(grad_features && (M == 0 || (grad_out && idx)), "glx_group_points_grad_gather: null");
  const size_t need = glx_group_points_grad_workspace_bytes(M, N, nsample) - 256;
  if (!workspace || workspace_bytes < need) {
    glx_set_error("glx_group_points_grad_gather: workspace %zu < %zu bytes", workspace_bytes, need);
    return GLX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t rowb = glx_align((size_t)(N + 1) * 4);
  int* cnt = (int*)workspace;
  int* cursor = (int*)((char*)workspace + rowb);
  int* offs = (int*)((char*)workspace + 2 * rowb);
  int* refs = (int*)((char*)workspace + 3 * rowb);
  void* scan_ws = (char*)refs + glx_align((size_t)M * nsample * 4);
  GlxFillJob job{cnt, 2 * rowb, 0};                           // counts and cursors
  int rc = glx_fill_multi(&job, 1, st);
  if (rc != GLX_OK) return rc;
  const long long total = (long long)M * nsample;
  if (total > 0)
    hipLaunchKernelGGL(k_gp_count, dim3((unsigned)glx_divup(total, 256)), dim3(256), 0, st, B, M, nsample, idx,
                       idx_batch_cnt, features_batch_cnt, cnt);
  rc = glx_exclusive_scan(IntArray{cnt}, (long long)N, offs, offs + N, scan_ws, glx_scan_workspace_bytes(N + 1), st);
  if (rc != GLX_OK) return rc;
  if (total > 0)
    hipLaunchKernelGGL(k_gp_fill, dim3((unsigned)glx_divup(total, 256)), dim3(256), 0, st, B, M, nsample, idx,
                       idx_batch_cnt, features_batch_cnt, (const int*)offs, cursor, refs);
  GlxFillJob zj{grad_features, (size_t)N * C * sizeof(float), 0};
  rc = glx_fill_multi(&zj, 1, st);
  if (rc != GLX_OK) return rc;
  if (total > 0)
    hipLaunchKernelGGL(k_gp_gather, dim3((unsigned)glx_divup(glx_divup(total, (long long)GP_CHUNK), 4LL)), dim3(256),
                       0, st, N, C, nsample, rows_layout, grad_out, (const int*)offs, (const int*)refs,
                       grad_features);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_group_points_grad_gather(int B, int M, int C, int N, int nsample, const float* grad_out,
                                            const int32_t* idx, const int32_t* idx_batch_cnt,
                                            const int32_t* features_batch_cnt, float* grad_features,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(M == 0 || (idx_batch_cnt && features_batch_cnt), "glx_group_points_grad_gather: null counts");
  return gp_grad_gather_impl(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt,
                             grad_features, 0, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------ row-major grouping
// out[m, s, :] = features[idx[m, s], :] with idx GLOBAL rows as the voxel query leaves them
// (idx[m,0] < 0: empty ball -> zeros): the (M, ns, C) layout keeps a neighbour's channels contiguous
// (128-byte row reads and writes; the reference's (M, C, ns) layout transposes on the way out and
// again in the gradient).  Gradient: the gather form above on contiguous rows.
__global__ void k_group_rows(const float* __restrict__ feats, const int* __restrict__ idx, long long total,
                             int C, int ns, float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const long long ms = e / C;
  const int c = (int)(e - ms * C);
  const long long m = ms / ns;
  out[e] = idx[m * ns] < 0 ? 0.f : feats[(long long)idx[ms] * C + c];
}

extern "C" int glx_group_rows(const float* features, const int32_t* idx, int M, int nsample, int C,
                              float* out, void* stream) {
  const long long total = (long long)M * nsample * C;
  if (total <= 0) return GLX_OK;
  GLX_REQUIRE(features && idx && out, "glx_group_rows: null pointer");
  hipLaunchKernelGGL(k_group_rows, dim3((unsigned)glx_divup(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     features, idx, total, C, nsample, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_group_rows_grad(const float* grad_out, const int32_t* idx, int M, int nsample, int C, int N,
                                   float* grad_features, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  return gp_grad_gather_impl(1, M, C, N, nsample, grad_out, idx, nullptr, nullptr, grad_features, 1, workspace,
                             workspace_bytes, stream);
}

// ------------------------------------------------------------------ (boxes x points) mask
// points_in_boxes_cpu, pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:143-168: out (N,P)
// 0/1 with the CPU variant's MARGIN (1e-2); margin is a parameter so the GPU value also fits.
__global__ void k_points_in_boxes_mask(int N, int P, float margin, const float* __restrict__ boxes,
                                       const float* __restrict__ pts, int* __restrict__ out) {
  const int i = blockIdx.y;
  BoxT bx;
  bx.load(boxes + (long long)i * 7);
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float lx, ly;
  out[(long long)i * P + p] =
      bx.contains(pts[(long long)p * 3], pts[(long long)p * 3 + 1], pts[(long long)p * 3 + 2], margin, lx, ly);
}

extern "C" int glx_points_in_boxes_mask(const float* boxes, int N, const float* pts, int P,
                                        float margin, int32_t* out, void* stream) {
  if (N == 0 || P == 0) return GLX_OK;
  GLX_REQUIRE(boxes && pts && out, "glx_points_in_boxes_mask: null pointer");
  hipLaunchKernelGGL(k_points_in_boxes_mask, dim3(glx_divup(P, 256), N), dim3(256), 0,
                     (hipStream_t)stream, N, P, margin, boxes, pts, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ====================================================================================
// PV-RCNN set-abstraction operators
// ====================================================================================

// Farthest point sampling, one block of 1024 threads per frame.  Every thread keeps its points
// (k = t, t+1024, ...: the assignment of the reference kernel, which fixes the tie rule) and their
// running min-distances in REGISTERS for clouds up to 16 K points (no global traffic inside the
// m-1 dependent iterations; the reference re-reads xyz and temp from global memory each time);
// larger clouds fall back to the caller's temp buffer.  Reduction: (value, thread) pairs, greater
// value wins, equal values keep the lower thread -- the net effect of the reference's tree
// (__update keeps idx1 on ties, sampling_gpu.cu:16-21).  The winner publishes its coordinates
// through LDS, so the next iteration starts without a global load.
#define FPS_THREADS 1024
#define FPS_DPT 16

struct FpsBest {
  float v;
  int t;   // owning thread
  int i;   // point index inside the frame
};

__device__ __forceinline__ FpsBest fps_pick(FpsBest a, FpsBest b) {
  return (b.v > a.v || (b.v == a.v && b.t < a.t)) ? b : a;
}

// Frames of a launch: stacked (per-frame counts on the device, GLOBAL output indices) or the batch layout of
// pointnet2_batch (B equal frames of uni_n points, uni_m samples each, indices LOCAL to the frame).
// tie_mod = the block size the reference would have launched (opt_n_threads(n), cuda_utils.h:9-13): among equal
// maxima its winner has the smallest k mod tie_mod, then the smallest k.
struct FpsFrames {
  const int* cnt;
  const int* num;
  int uni_n, uni_m, tie_mod;
};

template <bool REGS>
__global__ __launch_bounds__(FPS_THREADS) void k_stack_fps(
    int B, const float* __restrict__ xyz, FpsFrames fr,
    float* __restrict__ temp, int* __restrict__ idxs) {
  __shared__ float s_v[FPS_THREADS / 64];
  __shared__ int s_i[FPS_THREADS / 64];
  __shared__ int s_key[FPS_THREADS / 64];
  __shared__ float s_x[FPS_THREADS / 64], s_y[FPS_THREADS / 64], s_z[FPS_THREADS / 64];
  __shared__ float s_pt[3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long start = 0, ostart = 0;
  int n, m;
  if (fr.cnt) {
    for (int k = 0; k < b; ++k) { start += fr.cnt[k]; ostart += fr.num[k]; }
    n = fr.cnt[b]; m = fr.num[b];
  } else {
    start = (long long)b * fr.uni_n; ostart = (long long)b * fr.uni_m;
    n = fr.uni_n; m = fr.uni_m;
  }
  const int obase = fr.cnt ? (int)start : 0;
  const bool plain_ties = fr.tie_mod >= FPS_THREADS;
  const float* X = xyz + start * 3;
  float* T = temp + start;
  int* O = idxs + ostart;
  if (m <= 0 || n <= 0) return;
  // the register-resident form holds FPS_THREADS * FPS_DPT points: a frame with more (the host's max_points hint
  // was stale or too small) takes the global-memory loop instead of silently ignoring its tail
  const bool regs = REGS && n <= FPS_THREADS * FPS_DPT;
  float px[FPS_DPT], py[FPS_DPT], pz[FPS_DPT], pt[FPS_DPT];
  if (regs) {
#pragma unroll
    for (int j = 0; j < FPS_DPT; ++j) {
      int k = tid + j * FPS_THREADS;
      bool ok = k < n;
      px[j] = ok ? X[k * 3] : 0.f; py[j] = ok ? X[k * 3 + 1] : 0.f; pz[j] = ok ? X[k * 3 + 2] : 0.f;
      pt[j] = ok ? T[k] : -2.f;   // never selected
    }
  }
  if (tid == 0) {
    O[0] = obase;
    s_pt[0] = X[0]; s_pt[1] = X[1]; s_pt[2] = X[2];
  }
  __syncthreads();
  for (int s = 1; s < m; ++s) {
    const float x1 = s_pt[0], y1 = s_pt[1], z1 = s_pt[2];
    FpsBest best{-1.f, tid, 0};
    if (regs) {
#pragma unroll
      for (int j = 0; j < FPS_DPT; ++j) {
        float dx = px[j] - x1, dy = py[j] - y1, dz = pz[j] - z1;
        float d = dx * dx + dy * dy + dz * dz;
        float d2 = fminf(d, pt[j]);
        if (pt[j] >= 0.f) pt[j] = d2; else d2 = -2.f;
        if (d2 > best.v) { best.v = d2; best.i = tid + j * FPS_THREADS; }
      }
    } else {
      for (int k = tid; k < n; k += FPS_THREADS) {
        float dx = X[k * 3] - x1, dy = X[k * 3 + 1] - y1, dz = X[k * 3 + 2] - z1;
        float d = dx * dx + dy * dy + dz * dz;
        float d2 = fminf(d, T[k]);
        T[k] = d2;
        if (d2 > best.v) { best.v = d2; best.i = k; }
      }
    }
    // wave maximum of the value alone (6 cross-lane steps), then the lowest lane that holds it -- lower
    // lane = lower thread, the tie rule -- publishes index and coordinates from its registers
    float vmax = best.v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    int win, wkey = tid;
    if (plain_ties) {
      win = __ffsll((long long)__ballot(best.v == vmax)) - 1;
    } else {
      // fewer reference threads than ours (n < 1024: one point per thread here): order the tied threads as
      // the reference's block would -- (thread of the point there = tid mod tie_mod, then the point index)
      wkey = best.v == vmax ? (((tid % fr.tie_mod) << 10) | tid) : 0x7fffffff;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wkey = min(wkey, __shfl_xor(wkey, o, 64));
      win = wkey & 63;
    }
    if (lane == win) {
      float cx, cy, cz;
      if (regs) {
        const int jw = best.i / FPS_THREADS;
        cx = px[0]; cy = py[0]; cz = pz[0];
#pragma unroll
        for (int j = 1; j < FPS_DPT; ++j)
          if (jw == j) { cx = px[j]; cy = py[j]; cz = pz[j]; }
      } else {
        cx = X[best.i * 3]; cy = X[best.i * 3 + 1]; cz = X[best.i * 3 + 2];
      }
      s_v[wave] = best.v; s_i[wave] = best.i; s_key[wave] = wkey;
      s_x[wave] = cx; s_y[wave] = cy; s_z[wave] = cz;
    }
    __syncthreads();
    if (wave == 0) {
      float wv = lane < FPS_THREADS / 64 ? s_v[lane] : -3.f;
      float wmax = wv;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
      wmax = __shfl(wmax, 0, 64);
      int win2;
      if (plain_ties) {
        win2 = __ffsll((long long)__ballot(lane < FPS_THREADS / 64 && wv == wmax)) - 1;   // lower wave = lower threads
      } else {
        int k2 = (lane < FPS_THREADS / 64 && wv == wmax) ? s_key[lane] : 0x7fffffff;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) k2 = min(k2, __shfl_xor(k2, o, 64));
        k2 = __shfl(k2, 0, 64);
        win2 = (k2 & 1023) >> 6;
      }
      if (lane == win2) {
        O[s] = s_i[lane] + obase;
        s_pt[0] = s_x[lane]; s_pt[1] = s_y[lane]; s_pt[2] = s_z[lane];
      }
    }
    __syncthreads();
  }
}

extern "C" int glx_stack_fps(const float* xyz, const int32_t* xyz_batch_cnt, int B, int max_points,
                             const int32_t* num_sampled, float* temp, int32_t* idxs, void* stream) {
  if (B <= 0) return GLX_OK;
  GLX_REQUIRE(xyz && xyz_batch_cnt && num_sampled && temp && idxs, "glx_stack_fps: null pointer");
  hipStream_t st = (hipStream_t)stream;
  FpsFrames fr{xyz_batch_cnt, num_sampled, 0, 0, FPS_THREADS};
  if (max_points > 0 && max_points <= FPS_THREADS * FPS_DPT) {
    hipLaunchKernelGGL((k_stack_fps<true>), dim3(B), dim3(FPS_THREADS), 0, st, B, xyz, fr, temp, idxs);
  } else {
    hipLaunchKernelGGL((k_stack_fps<false>), dim3(B), dim3(FPS_THREADS), 0, st, B, xyz, fr, temp, idxs);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// farthest_point_sampling_kernel<block_size>, pointnet2_batch/src/sampling_gpu.cu:97-230 (and its twin in
// pointnet2_stack/src/sampling_gpu.cu:15-185): B frames of N points, idx (B, m) LOCAL indices; the block
// size the reference picks (largest power of two <= N, at most 1024) only shows in its tie rule.
extern "C" int glx_batch_fps(int B, int N, int m, const float* xyz, float* temp, int32_t* idx, void* stream) {
  if (B <= 0 || m <= 0) return GLX_OK;
  GLX_REQUIRE(xyz && temp && idx, "glx_batch_fps: null pointer");
  GLX_REQUIRE(N >= 1, "glx_batch_fps: N = %d", N);
  // opt_n_threads(N), cuda_utils.h:9-13, evaluated the same way (double log ratio, truncated)
  const int pow_2 = (int)(log((double)N) / log(2.0));
  int tie = 1 << pow_2;
  tie = tie > FPS_THREADS ? FPS_THREADS : (tie < 1 ? 1 : tie);
  hipStream_t st = (hipStream_t)stream;
  FpsFrames fr{nullptr, nullptr, N, m, tie};
  if (N <= FPS_THREADS * FPS_DPT) {
    hipLaunchKernelGGL((k_stack_fps<true>), dim3(B), dim3(FPS_THREADS), 0, st, B, xyz, fr, temp, idx);
  } else {
    hipLaunchKernelGGL((k_stack_fps<false>), dim3(B), dim3(FPS_THREADS), 0, st, B, xyz, fr, temp, idx);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// three nearest known points of the same frame: grid (blocks over the frame's queries, frame);
// the frame's known points stream through LDS in tiles shared by the 256 queries of the block.
#define TNN_THREADS 256
#define TNN_TILE 1024
#define TNN_SPLIT 8

__global__ __launch_bounds__(TNN_THREADS) void k_three_nn(
    int B, const float* __restrict__ unknown, const int* __restrict__ unknown_batch_cnt,
    const float* __restrict__ known, const int* __restrict__ known_batch_cnt, int uni_nu, int uni_nk,
    float* __restrict__ dist2, int* __restrict__ idx) {
  __shared__ float s_k[TNN_TILE * 3];
  const int b = blockIdx.y;
  long long ustart = 0, kstart = 0;
  int nu, nk;
  if (unknown_batch_cnt) {
    for (int k = 0; k < b; ++k) { ustart += unknown_batch_cnt[k]; kstart += known_batch_cnt[k]; }
    nu = unknown_batch_cnt[b]; nk = known_batch_cnt[b];
  } else {   // batch layout (pointnet2_batch): B equal frames, indices local to the frame
    ustart = (long long)b * uni_nu; kstart = (long long)b * uni_nk;
    nu = uni_nu; nk = uni_nk;
  }
  const int ibase = unknown_batch_cnt ? (int)kstart : 0;
  // TNN_SPLIT lanes per query, each scanning every TNN_SPLIT-th known point: 8x the waves of a thread per
  // query (65 K queries were 1024 waves = one per SIMD, nothing to hide the LDS latency behind)
  if ((long long)blockIdx.x * (TNN_THREADS / TNN_SPLIT) >= nu) return;   // block-uniform
  const int q = blockIdx.x * (TNN_THREADS / TNN_SPLIT) + threadIdx.x / TNN_SPLIT;
  const int sub = threadIdx.x % TNN_SPLIT;
  const bool act = q < nu;
  const float* up = unknown + (ustart + (act ? q : 0)) * 3;
  const float ux = up[0], uy = up[1], uz = up[2];
  // the reference's running minima start at 1e40 in double = above every float: +inf in float decides alike
  float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
  int i1 = 0, i2 = 0, i3 = 0;
  const float* Kp = known + kstart * 3;
  for (int t0 = 0; t0 < nk; t0 += TNN_TILE) {
    const int tn = min(TNN_TILE, nk - t0);
    __syncthreads();
    for (int e = threadIdx.x; e < tn * 3; e += TNN_THREADS) s_k[e] = Kp[(long long)t0 * 3 + e];
    __syncthreads();
    if (act) {
#pragma unroll 4
      for (int k = sub; k < tn; k += TNN_SPLIT) {
        float x = s_k[k * 3], y = s_k[k * 3 + 1], z = s_k[k * 3 + 2];
        float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
        if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = t0 + k; }
        else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = t0 + k; }
        else if (d < b3) { b3 = d; i3 = t0 + k; }
      }
    }
  }
  // merge the TNN_SPLIT sorted triples: three rounds of "smallest head by (distance, index)" -- the order
  // the reference's strict comparisons produce (equal distances fill the slots in index order)
  float od[3];
  int oi[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float hd = b1;
    int hi = i1;
#pragma unroll
    for (int o = TNN_SPLIT / 2; o > 0; o >>= 1) {
      const float xd = __shfl_xor(hd, o, 64);
      const int xi = __shfl_xor(hi, o, 64);
      if (xd < hd || (xd == hd && xi < hi)) { hd = xd; hi = xi; }
    }
    od[r] = hd; oi[r] = hi;
    if (b1 == hd && i1 == hi) { b1 = b2; i1 = i2; b2 = b3; i2 = i3; b3 = INFINITY; i3 = 0; }
  }
  if (act && sub == 0) {
    float* dp = dist2 + (ustart + q) * 3;
    int* ip = idx + (ustart + q) * 3;
    dp[0] = od[0]; dp[1] = od[1]; dp[2] = od[2];
    ip[0] = oi[0] + ibase; ip[1] = oi[1] + ibase; ip[2] = oi[2] + ibase;
  }
}

extern "C" int glx_three_nn(int B, int N, int max_queries_per_frame, const float* unknown,
                            const int32_t* unknown_batch_cnt, const float* known,
                            const int32_t* known_batch_cnt, float* dist2, int32_t* idx, void* stream) {
  if (N <= 0 || B <= 0) return GLX_OK;
  GLX_REQUIRE(unknown && unknown_batch_cnt && known && known_batch_cnt && dist2 && idx,
              "glx_three_nn: null pointer");
  const int per = max_queries_per_frame > 0 ? max_queries_per_frame : N;
  hipLaunchKernelGGL(k_three_nn, dim3(glx_divup(per, TNN_THREADS / TNN_SPLIT), B), dim3(TNN_THREADS), 0,
                     (hipStream_t)stream, B, unknown, unknown_batch_cnt, known, known_batch_cnt, 0, 0,
                     dist2, idx);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// three_nn_kernel_fast, pointnet2_batch/src/interpolate_gpu.cu:15-60: unknown (B, n, 3), known (B, m, 3) ->
// dist2 (B, n, 3), idx (B, n, 3) LOCAL to the frame.  With m < 3 the unfilled slots keep the reference's
// initial values narrowed to float (1e40 -> +inf) and index 0.
extern "C" int glx_batch_three_nn(int B, int n, int m, const float* unknown, const float* known, float* dist2,
                                  int32_t* idx, void* stream) {
  if (n <= 0 || B <= 0) return GLX_OK;
  GLX_REQUIRE(unknown && known && dist2 && idx, "glx_batch_three_nn: null pointer");
  hipLaunchKernelGGL(k_three_nn, dim3(glx_divup(n, TNN_THREADS / TNN_SPLIT), B), dim3(TNN_THREADS), 0,
                     (hipStream_t)stream, B, unknown, (const int*)nullptr, known, (const int*)nullptr, n, m,
                     dist2, idx);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

__global__ void k_three_interpolate(int N, int C, const float* __restrict__ f,
                                    const int* __restrict__ idx, const float* __restrict__ w,
                                    float* __restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * C) return;
  int p = (int)(t / C), c = (int)(t - (long long)p * C);
  const int* ip = idx + (long long)p * 3;
  const float* wp = w + (long long)p * 3;
  out[t] = wp[0] * f[(long long)ip[0] * C + c] + wp[1] * f[(long long)ip[1] * C + c] +
           wp[2] * f[(long long)ip[2] * C + c];
}

__global__ void k_three_interpolate_grad(int N, int C, const float* __restrict__ g,
                                         const int* __restrict__ idx, const float* __restrict__ w,
                                         float* __restrict__ gf) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * C) return;
  int p = (int)(t / C), c = (int)(t - (long long)p * C);
  const int* ip = idx + (long long)p * 3;
  const float* wp = w + (long long)p * 3;
  const float v = g[t];
  atomicAdd(gf + (long long)ip[0] * C + c, v * wp[0]);
  atomicAdd(gf + (long long)ip[1] * C + c, v * wp[1]);
  atomicAdd(gf + (long long)ip[2] * C + c, v * wp[2]);
}

extern "C" int glx_three_interpolate(int N, int C, const float* features, const int32_t* idx,
                                     const float* weight, float* out, void* stream) {
  if (N <= 0 || C <= 0) return GLX_OK;
  GLX_REQUIRE(features && idx && weight && out, "glx_three_interpolate: null pointer");
  hipLaunchKernelGGL(k_three_interpolate, dim3(glx_divup((long long)N * C, 256)), dim3(256), 0,
                     (hipStream_t)stream, N, C, features, idx, weight, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_three_interpolate_grad(int N, int C, const float* grad_out, const int32_t* idx,
                                          const float* weight, float* grad_features, void* stream) {
  if (N <= 0 || C <= 0) return GLX_OK;
  GLX_REQUIRE(grad_out && idx && weight && grad_features, "glx_three_interpolate_grad: null pointer");
  hipLaunchKernelGGL(k_three_interpolate_grad, dim3(glx_divup((long long)N * C, 256)), dim3(256), 0,
                     (hipStream_t)stream, N, C, grad_out, idx, weight, grad_features);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ====================================================================================
// RoI-grid pooling aggregation (Voxel-RCNN), everything after the voxel query of one scale in ONE
// kernel (inference): NeighborVoxelSAModuleMSG.forward, voxel_pool_modules.py:88-108 --
//   v[c]   = max_s relu( feats[idx[m,s], c] + Wpos[c,:] . (xyz[idx[m,s]] - new_xyz[m]) + bpos[c] )
//   out[o] = relu( Wout[o,:] . v + bout[o] )
// with an empty ball contributing feats = 0, offset = 0 (so v = relu(bpos)); (Wpos, bpos) and
// (Wout, bout) are the 1x1 convs with their eval-mode BatchNorms folded.  The PyTorch path
// materialises (M, C, ns) tensors four times per scale (group, mask, pos-MLP, ReLU) before the
// max -- 177 MB each at the GLENet-VR sizes; here a grid point's 16 neighbours are reduced in
// registers: lanes = channels (coalesced 128-B row gathers), a wave holds 64/LP grid points.
template <int LP, bool CEN>
__global__ __launch_bounds__(256) void k_voxel_pool_agg(
    const float* __restrict__ feats, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ idx, const unsigned char* __restrict__ empty, int M, int ns, int Cm, int Co,
    const float* __restrict__ Wpos, const float* __restrict__ bpos, const float* __restrict__ Wout,
    const float* __restrict__ bout, float* __restrict__ out, int out_stride, VoxelCentres cen) {
  extern __shared__ float s_wout[];            // [c][o] : Cm * Co
  for (int e = threadIdx.x; e < Cm * Co; e += blockDim.x) {
    int o = e / Cm, c = e - o * Cm;           // Wout is (Co, Cm) row-major
    s_wout[c * Co + o] = Wout[e];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int PPW = 64 / LP;                 // grid points per wave
  const int c = lane % LP, sub = lane / LP;
  const long long m = ((long long)blockIdx.x * (blockDim.x >> 6) + wave) * PPW + sub;
  const bool live = m < M;
  const bool cok = c < Cm;
  float wx = 0.f, wy = 0.f, wz = 0.f, bp = 0.f;
  if (cok) { wx = Wpos[c * 3]; wy = Wpos[c * 3 + 1]; wz = Wpos[c * 3 + 2]; bp = bpos[c]; }
  float v = 0.f;
  if (live) {
    if (empty ? empty[m] != 0 : idx[m * ns] < 0) {
      v = fmaxf(bp, 0.f);
    } else {
      const float qx = new_xyz[m * 3], qy = new_xyz[m * 3 + 1], qz = new_xyz[m * 3 + 2];
      v = -3.0e38f;
      for (int s = 0; s < ns; ++s) {
        const long long i = idx[m * ns + s];
        const float f = cok ? feats[i * Cm + c] : 0.f;
        float px, py, pz;
        if (CEN) {
          glx_voxel_centre(cen, i, px, py, pz);
        } else {
          px = xyz[i * 3], py = xyz[i * 3 + 1], pz = xyz[i * 3 + 2];
        }
        const float rx = px - qx, ry = py - qy, rz = pz - qz;
        float t = wx * rx;                       // conv (no bias) then the folded BN shift
        t = fmaf(wy, ry, t);
        t = fmaf(wz, rz, t);
        v = fmaxf(v, fmaxf(f + (t + bp), 0.f));
      }
    }
  }
  // second 1x1 conv (+BN+ReLU): lane o needs every v[c] of its grid point
  float acc = (c < Co) ? bout[c < Co ? c : 0] : 0.f;
  for (int k = 0; k < Cm; ++k) {
    const float vk = __shfl(v, sub * LP + k, 64);
    if (c < Co) acc = fmaf(s_wout[k * Co + c], vk, acc);
  }
  if (live && c < Co) out[m * out_stride + c] = fmaxf(acc, 0.f);
}

static int voxel_pool_agg_launch(const float* feats, const float* xyz, const float* new_xyz,
                                 const int32_t* idx, const uint8_t* empty, int M, int nsample, int Cm,
                                 int Co, const float* Wpos, const float* bpos, const float* Wout,
                                 const float* bout, float* out, int out_stride, const VoxelCentres* cen,
                                 void* stream) {
  const int LP = (Cm <= 32 && Co <= 32) ? 32 : 64;
  const dim3 grid(glx_divup(M, 4 * (64 / LP))), block(256);
  const size_t lds = (size_t)Cm * Co * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define GLX_AGG(LPV, CENV)                                                                        \
  hipLaunchKernelGGL((k_voxel_pool_agg<LPV, CENV>), grid, block, lds, st, feats, xyz, new_xyz, idx, \
                     empty, M, nsample, Cm, Co, Wpos, bpos, Wout, bout, out, out_stride,          \
                     cen ? *cen : VoxelCentres{})
  if (LP == 32) { if (cen) GLX_AGG(32, true); else GLX_AGG(32, false); }
  else          { if (cen) GLX_AGG(64, true); else GLX_AGG(64, false); }
#undef GLX_AGG
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_voxel_pool_agg(const float* feats, const float* xyz, const float* new_xyz,
                                  const int32_t* idx, const uint8_t* empty, int M, int nsample, int Cm,
                                  int Co, const float* Wpos, const float* bpos, const float* Wout,
                                  const float* bout, float* out, void* stream) {
  if (M <= 0) return GLX_OK;
  GLX_REQUIRE(feats && xyz && new_xyz && idx && Wpos && bpos && Wout && bout && out,
              "glx_voxel_pool_agg: null pointer");
  GLX_REQUIRE(Cm >= 1 && Cm <= 64 && Co >= 1 && Co <= 64 && nsample >= 1,
              "glx_voxel_pool_agg: channel widths must be 1..64");
  return voxel_pool_agg_launch(feats, xyz, new_xyz, idx, empty, M, nsample, Cm, Co, Wpos, bpos, Wout,
                               bout, out, Co, nullptr, stream);
}

// ------------------------------------------------------------------ RoI-grid pooling, whole stage
// VoxelRCNNHead.roi_grid_pool (pcdet/models/roi_heads/voxelrcnn_head.py:106-191) for inference in
// 1 + 2 x scales launches: grid points + their stride-1 voxel coordinates from the RoIs, then per
// scale a query and an aggregation that rebuild voxel centres from the sparse tensor's indices.
//
// Grid points: get_dense_grid_points + rotate_points_along_z + centre (voxelrcnn_head.py:191-215,
// common_utils.py:35-57), rounding step by step in fp32: ((i + 0.5) * (1/G)) * size - size * 0.5,
// x' = x cos - y sin, y' = x sin + y cos, + centre.  (The reference rotates with a batched 3x3
// matmul whose summation may contract to fma: 1 ulp.)  Voxel coordinates: the reference's float
// floor division `(p - range_min) // voxel` restated from ATen's div_floor kernel for a scalar
// divisor: fmod, (a - mod) * (1/b), sign fix, floor with the > 0.5 round-up.
__device__ __forceinline__ float glx_div_floor(float a, float b, float inv_b) {
  const float mod = fmodf(a, b);
  float div = __fmul_rn(__fsub_rn(a, mod), inv_b);
  if (mod != 0.f && ((b < 0.f) != (mod < 0.f))) div = __fsub_rn(div, 1.f);
  if (div == 0.f) return copysignf(0.f, __fmul_rn(a, inv_b));
  float fl = floorf(div);
  if (__fsub_rn(div, fl) > 0.5f) fl = __fadd_rn(fl, 1.f);
  return fl;
}

struct RoiGridGeom {
  float r0x, r0y, r0z, vx, vy, vz, ivx, ivy, ivz;
};

__global__ void k_roi_grid_points(const float* __restrict__ rois, int n_rois, int cols,
                                  int rois_per_frame, int G, float inv_g, RoiGridGeom gm,
                                  float* __restrict__ grid_xyz, int* __restrict__ coords) {
  const int G3 = G * G * G;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_rois * G3) return;
  const int r = (int)(t / G3), k = (int)(t - (long long)r * G3);
  const int ix = k / (G * G), iy = (k / G) % G, iz = k % G;          // nonzero() order: x-major
  const float* b = rois + (long long)r * cols;
  const float dx = b[3], dy = b[4], dz = b[5];
  float sn, cs;
  sn = sinf(b[6]);
  cs = cosf(b[6]);
  const float lx = __fsub_rn(__fmul_rn(__fmul_rn(__fadd_rn((float)ix, 0.5f), inv_g), dx), __fmul_rn(dx, 0.5f));
  const float ly = __fsub_rn(__fmul_rn(__fmul_rn(__fadd_rn((float)iy, 0.5f), inv_g), dy), __fmul_rn(dy, 0.5f));
  const float lz = __fsub_rn(__fmul_rn(__fmul_rn(__fadd_rn((float)iz, 0.5f), inv_g), dz), __fmul_rn(dz, 0.5f));
  const float gx = __fadd_rn(__fsub_rn(__fmul_rn(lx, cs), __fmul_rn(ly, sn)), b[0]);
  const float gy = __fadd_rn(__fadd_rn(__fmul_rn(lx, sn), __fmul_rn(ly, cs)), b[1]);
  const float gz = __fadd_rn(lz, b[2]);
  grid_xyz[t * 3] = gx, grid_xyz[t * 3 + 1] = gy, grid_xyz[t * 3 + 2] = gz;
  int4 c;
  c.x = r / rois_per_frame;
  c.y = (int)glx_div_floor(__fsub_rn(gz, gm.r0z), gm.vz, gm.ivz);
  c.z = (int)glx_div_floor(__fsub_rn(gy, gm.r0y), gm.vy, gm.ivy);
  c.w = (int)glx_div_floor(__fsub_rn(gx, gm.r0x), gm.vx, gm.ivx);
  reinterpret_cast<int4*>(coords)[t] = c;                             // b z y x at stride 1
}

extern "C" int glx_roi_grid_points(const float* rois, int n_rois, int cols, int rois_per_frame,
                                   int grid_size, const float* range_min, const float* voxel_size,
                                   float* grid_xyz, int32_t* coords, void* stream) {
  if (n_rois <= 0) return GLX_OK;
  GLX_REQUIRE(rois && range_min && voxel_size && grid_xyz && coords, "glx_roi_grid_points: null pointer");
  GLX_REQUIRE(cols >= 7 && rois_per_frame >= 1 && grid_size >= 1 && grid_size <= 32,
              "glx_roi_grid_points: rois need >= 7 columns, grid_size 1..32");
  RoiGridGeom gm{range_min[0], range_min[1], range_min[2], voxel_size[0], voxel_size[1], voxel_size[2],
                 1.f / voxel_size[0], 1.f / voxel_size[1], 1.f / voxel_size[2]};
  const long long total = (long long)n_rois * grid_size * grid_size * grid_size;
  hipLaunchKernelGGL(k_roi_grid_points, dim3((unsigned)glx_divup(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, rois, n_rois, cols, rois_per_frame, grid_size,
                     1.f / (float)grid_size, gm, grid_xyz, coords);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

static VoxelCentres make_centres(const int32_t* indices, const float* range_min, const float* voxel_size,
                                 int stride) {
  // tensor(voxel_size).float() * downsample_times, as get_voxel_centers forms it
  return VoxelCentres{indices, voxel_size[0] * (float)stride, voxel_size[1] * (float)stride,
                      voxel_size[2] * (float)stride, range_min[0], range_min[1], range_min[2]};
}

__global__ __launch_bounds__(256) void k_voxel_centres(VoxelCentres cen, int N, float* __restrict__ xyz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float x, y, z;
  glx_voxel_centre(cen, i, x, y, z);
  xyz[3 * (long long)i] = x, xyz[3 * (long long)i + 1] = y, xyz[3 * (long long)i + 2] = z;
}

extern "C" int glx_voxel_centers(const int32_t* indices, int N, int stride, const float* range_min,
                                 const float* voxel_size, float* xyz, void* stream) {
  if (N <= 0) return GLX_OK;
  GLX_REQUIRE(indices && range_min && voxel_size && xyz, "glx_voxel_centers: null pointer");
  GLX_REQUIRE(stride >= 1, "glx_voxel_centers: stride must be >= 1");
  hipLaunchKernelGGL(k_voxel_centres, dim3(glx_divup(N, 256)), dim3(256), 0, (hipStream_t)stream,
                     make_centres(indices, range_min, voxel_size, stride), N, xyz);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_roi_grid_query(int M, int Z, int Y, int X, int nsample, float radius, int z_range,
                                  int y_range, int x_range, const float* grid_xyz,
                                  const int32_t* coords, int stride, const int32_t* indices,
                                  const float* range_min, const float* voxel_size,
                                  const uint64_t* bitmap, const int32_t* prefix,
                                  const int32_t* rank_to_row, int32_t* idx, void* stream) {
  if (M == 0) return GLX_OK;
  GLX_REQUIRE(grid_xyz && coords && indices && range_min && voxel_size && bitmap && prefix && idx,
              "glx_roi_grid_query: null pointer");
  GLX_REQUIRE(stride >= 1 && nsample >= 1, "glx_roi_grid_query: stride and nsample must be >= 1");
  if (2 * x_range + 1 <= VQ_ROW_MAX_WX)
    hipLaunchKernelGGL((k_voxel_query_rows<true>), dim3(glx_divup(M, 32)), dim3(256), 0,
                       (hipStream_t)stream, M, Z, Y, X, nsample, radius * radius, z_range, y_range,
                       x_range, grid_xyz, (const float*)nullptr, coords,
                       (const unsigned long long*)bitmap, (const int*)prefix, rank_to_row, idx,
                       make_centres(indices, range_min, voxel_size, stride), stride);
  else
    hipLaunchKernelGGL((k_voxel_query<false, 8, true>), dim3(glx_divup(M, 32)), dim3(256), 0,
                       (hipStream_t)stream, M, Z, Y, X, nsample, radius * radius, z_range, y_range,
                       x_range, grid_xyz, (const float*)nullptr, coords, (const int*)nullptr,
                       (const unsigned long long*)bitmap, (const int*)prefix, rank_to_row, idx,
                       make_centres(indices, range_min, voxel_size, stride), stride);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_roi_grid_query_multi(int n_scales, const glx_roi_query* scales, int M, const float* grid_xyz,
                                        const int32_t* coords, const float* range_min, const float* voxel_size,
                                        void* stream) {
  if (M == 0 || n_scales == 0) return GLX_OK;
  GLX_REQUIRE(scales && grid_xyz && coords && range_min && voxel_size, "glx_roi_grid_query_multi: null pointer");
  GLX_REQUIRE(n_scales >= 1 && n_scales <= VQ_MAX_SCALES, "glx_roi_grid_query_multi: %d scales (1..%d)", n_scales, VQ_MAX_SCALES);
  VqScales q;
  for (int k = 0; k < n_scales; ++k) {
    const glx_roi_query& a = scales[k];
    GLX_REQUIRE(a.indices && a.bitmap && a.prefix && a.idx && a.stride >= 1 && a.nsample >= 1,
                "glx_roi_grid_query_multi: scale %d: null pointer / bad stride or nsample", k);
    GLX_REQUIRE(2 * a.x_range + 1 <= VQ_ROW_MAX_WX, "glx_roi_grid_query_multi: scale %d: x range %d", k, a.x_range);
    q.s[k] = VqScale{a.Z, a.Y, a.X, a.nsample, a.z_range, a.y_range, a.x_range, a.stride, a.radius * a.radius,
                     (const unsigned long long*)a.bitmap, (const int*)a.prefix, (const int*)a.rank_to_row, (int*)a.idx,
                     make_centres(a.indices, range_min, voxel_size, a.stride)};
  }
  hipLaunchKernelGGL(k_voxel_query_rows_multi, dim3(glx_divup(M, 32), n_scales), dim3(256), 0, (hipStream_t)stream, M, grid_xyz,
                     coords, q);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_roi_grid_agg(const float* feats, const int32_t* indices, int stride,
                                const float* range_min, const float* voxel_size, const float* grid_xyz,
                                const int32_t* idx, int M, int nsample, int Cm, int Co,
                                const float* Wpos, const float* bpos, const float* Wout,
                                const float* bout, float* out, int out_stride, void* stream) {
  if (M <= 0) return GLX_OK;
  GLX_REQUIRE(feats && indices && range_min && voxel_size && grid_xyz && idx && Wpos && bpos && Wout &&
                  bout && out, "glx_roi_grid_agg: null pointer");
  GLX_REQUIRE(Cm >= 1 && Cm <= 64 && Co >= 1 && Co <= 64 && nsample >= 1 && out_stride >= Co,
              "glx_roi_grid_agg: channel widths must be 1..64, out_stride >= Co");
  const VoxelCentres cen = make_centres(indices, range_min, voxel_size, stride);
  return voxel_pool_agg_launch(feats, nullptr, grid_xyz, idx, nullptr, M, nsample, Cm, Co, Wpos, bpos,
                               Wout, bout, out, out_stride, &cen, stream);
}

// ------------------------------------------------------------------ relu(a + b) max over the neighbours
// Training path of the RoI-grid pooling MLP (voxel_pool_modules.py:96-104: relu(grouped + position
// features), max_pool over nsample) on row-major (M, ns, C) tensors in one pass: out[m, c] =
// max_s relu(a[m,s,c] + b[m,s,c]) and the slot that attains it (first on ties, as torch.max).
// Gradient: the (M, ns, C) tensor that is grad_out at the winning slot where the maximum is
// positive and 0 elsewhere -- the same tensor for a and b.  Replaces add + relu + max (6 tensor
// passes forward, 4 backward) by 2 + 1.
__global__ void k_relu_add_max(const float* __restrict__ a, const float* __restrict__ b, long long MC, int ns,
                               int C, float* __restrict__ out, int* __restrict__ arg) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= MC) return;
  const long long m = t / C;
  const int c = (int)(t - m * C);
  const float* pa = a + m * ns * C + c;
  const float* pb = b + m * ns * C + c;
  float best = -1.f;
  int bi = 0;
  for (int s = 0; s < ns; ++s) {
    const float v = fmaxf(pa[(long long)s * C] + pb[(long long)s * C], 0.f);
    if (v > best) { best = v; bi = s; }
  }
  out[t] = best;
  arg[t] = bi;
}

__global__ void k_relu_add_max_grad(const float* __restrict__ grad_out, const float* __restrict__ out,
                                    const int* __restrict__ arg, long long total, int ns, int C,
                                    float* __restrict__ grad_in) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const long long ms = e / C;
  const int c = (int)(e - ms * C);
  const long long m = ms / ns;
  const int s = (int)(ms - m * ns);
  const long long t = m * C + c;
  grad_in[e] = (arg[t] == s && out[t] > 0.f) ? grad_out[t] : 0.f;
}

extern "C" int glx_relu_add_max(const float* a, const float* b, int M, int nsample, int C, float* out,
                                int32_t* arg, void* stream) {
  const long long mc = (long long)M * C;
  if (mc <= 0) return GLX_OK;
  GLX_REQUIRE(a && b && out && arg && nsample > 0, "glx_relu_add_max: null pointer or no samples");
  hipLaunchKernelGGL(k_relu_add_max, dim3((unsigned)glx_divup(mc, 256)), dim3(256), 0, (hipStream_t)stream, a, b,
                     mc, nsample, C, out, arg);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_relu_add_max_grad(const float* grad_out, const float* out, const int32_t* arg, int M,
                                     int nsample, int C, float* grad_in, void* stream) {
  const long long total = (long long)M * nsample * C;
  if (total <= 0) return GLX_OK;
  GLX_REQUIRE(grad_out && out && arg && grad_in, "glx_relu_add_max_grad: null pointer");
  hipLaunchKernelGGL(k_relu_add_max_grad, dim3((unsigned)glx_divup(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     grad_out, out, arg, total, nsample, C, grad_in);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
