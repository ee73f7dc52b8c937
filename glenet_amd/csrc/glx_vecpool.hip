// VectorPool family of PV-RCNN++: the four remaining exports of
// pcdet.ops.pointnet2.pointnet2_stack.pointnet2_stack_cuda --
//   query_stacked_local_neighbor_idxs_wrapper_stack, query_three_nn_by_stacked_local_idxs_wrapper_stack,
//   vector_pool_wrapper, vector_pool_grad_wrapper   (pointnet2_stack/src/vector_pool_gpu.cu:19-480).
// The reference walks a frame's support points with ONE thread per new point (n = 16 K+ sequential distance
// tests per thread) and hands out output slots with atomicAdd, which makes the order of the per-point
// segments run-dependent.  Here a WAVE owns a new point: its lanes test 64 support points per step and a
// ballot keeps the hits in index order, so the sequential semantics (first nsample hits, first hit per
// sub-voxel, sums in ascending k) are reproduced exactly; slots come from an exclusive scan of per-point
// counts, i.e. segments are laid out in ascending new-point order (one of the orders the reference's
// atomics can produce) and results are bitwise reproducible.
#include "glx_common.h"
#include "glx_scan.h"

#define VP_WAVES 4

__device__ __forceinline__ int vp_frame(int pt, const int* __restrict__ new_cnt, const int* __restrict__ xyz_cnt, int B,
                                        int& start, int& n) {
  int bs = 0, pc = new_cnt[0];
  for (int k = 1; k < B; k++) {
    if (pt < pc) break;
    pc += new_cnt[k];
    bs = k;
  }
  start = 0;
  for (int k = 0; k < bs; k++) start += xyz_cnt[k];
  n = xyz_cnt[bs];
  return bs;
}

__device__ __forceinline__ bool vp_in_range(float lx, float ly, float lz, float dist, int neighbor_type) {
  if (neighbor_type == 1) return !(lx * lx + ly * ly + lz * lz > dist * dist);
  return !((int)(fabsf(lx) > dist) | (int)(fabsf(ly) > dist) | (int)(fabsf(lz) > dist));
}

// ---- local neighbour lists (vector_pool_gpu.cu:122-200).  EMIT = false: counts[pt]; true: the lists.
template <bool EMIT>
__global__ __launch_bounds__(64 * VP_WAVES) void k_vp_neighbors(
    const float* __restrict__ support_xyz, const int* __restrict__ xyz_cnt, const float* __restrict__ new_xyz,
    const int* __restrict__ new_cnt, int B, int M, float dist, int nsample, int neighbor_type, int max_thresh,
    int* __restrict__ counts, const int* __restrict__ offs, int* __restrict__ stack, int* __restrict__ start_len) {
  const int pt = blockIdx.x * VP_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pt >= M) return;
  int start, n;
  vp_frame(pt, new_cnt, xyz_cnt, B, start, n);
  const float qx = new_xyz[(long long)pt * 3], qy = new_xyz[(long long)pt * 3 + 1], qz = new_xyz[(long long)pt * 3 + 2];
  const int cap = (nsample > 0 && nsample < 1000) ? nsample : 1000;    // temp_idxs[1000], then the nsample break
  const int s0 = EMIT ? offs[pt] : 0;
  int cnt = 0;
  for (int k0 = 0; k0 < n && cnt < cap; k0 += 64) {
    const int k = k0 + lane;
    bool hit = false;
    if (k < n) {
      const float* p = support_xyz + (long long)(start + k) * 3;
      hit = vp_in_range(p[0] - qx, p[1] - qy, p[2] - qz, dist, neighbor_type);
    }
    const unsigned long long mask = __ballot(hit);
    const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
    if (EMIT && hit && pos < cap && s0 < max_thresh && s0 + pos < max_thresh) stack[s0 + pos] = start + k;
    cnt += __popcll(mask);
  }
  cnt = cnt < cap ? cnt : cap;
  if (lane == 0) {
    if (EMIT) {
      start_len[pt * 2] = s0;
      start_len[pt * 2 + 1] = cnt;
    } else {
      counts[pt] = cnt;
    }
  }
}

// ---- three nearest of a point's neighbour list per grid centre (vector_pool_gpu.cu:19-85)
__global__ void k_vp_three_nn(const float* __restrict__ support_xyz, const float* __restrict__ centers,
                              int* __restrict__ grid_idxs, float* __restrict__ grid_dist2,
                              const int* __restrict__ stack, const int* __restrict__ start_len, int M, int G) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)M * G) return;
  const int pt = (int)(t / G);
  const float cx = centers[t * 3], cy = centers[t * 3 + 1], cz = centers[t * 3 + 2];
  const int* nb = stack + start_len[pt * 2];
  const int len = start_len[pt * 2 + 1];
  double b1 = 1e40, b2 = 1e40, b3 = 1e40;
  int i1 = -1, i2 = -1, i3 = -1;
  for (int k = 0; k < len; ++k) {
    const int j = nb[k];
    const float x = support_xyz[(long long)j * 3], y = support_xyz[(long long)j * 3 + 1], z = support_xyz[(long long)j * 3 + 2];
    const float d = (cx - x) * (cx - x) + (cy - y) * (cy - y) + (cz - z) * (cz - z);
    if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = j; }
    else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = j; }
    else if (d < b3) { b3 = d; i3 = j; }
  }
  if (i2 == -1) { i2 = i1; b2 = b1; }
  if (i3 == -1) { i3 = i1; b3 = b1; }
  grid_dist2[t * 3] = (float)b1; grid_dist2[t * 3 + 1] = (float)b2; grid_dist2[t * 3 + 2] = (float)b3;
  grid_idxs[t * 3] = i1; grid_idxs[t * 3 + 1] = i2; grid_idxs[t * 3 + 2] = i3;
}

// ---- vector pooling (vector_pool_gpu.cu:243-375).  WRITE = false: counts[pt] = rows of grouped_idxs the point
// produces; true: pooled sums, point counts, local xyz sums and the rows at offs[pt].
template <bool WRITE>
__global__ __launch_bounds__(64 * VP_WAVES) void k_vp_pool(
    const float* __restrict__ support_xyz, const float* __restrict__ support_features, const int* __restrict__ xyz_cnt,
    const float* __restrict__ new_xyz, const int* __restrict__ new_cnt, int B, int M, int cin, int cout, int gx, int gy,
    int gz, float dist, int use_xyz, int nsample, int neighbor_type, int pooling_type, int* __restrict__ counts,
    const int* __restrict__ offs, float* __restrict__ new_features, float* __restrict__ new_local_xyz,
    int* __restrict__ point_cnt, int* __restrict__ grouped_idxs) {
  const int pt = blockIdx.x * VP_WAVES + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pt >= M) return;
  int start, n;
  vp_frame(pt, new_cnt, xyz_cnt, B, start, n);
  const int G = gx * gy * gz, cg = cout / G, folds = cin / cg;
  const float sx = dist * 2 / gx, sy = dist * 2 / gy, sz = dist * 2 / gz;
  const float qx = new_xyz[(long long)pt * 3], qy = new_xyz[(long long)pt * 3 + 1], qz = new_xyz[(long long)pt * 3 + 2];
  float* nf = new_features + (long long)pt * cout;
  float* nl = new_local_xyz + (long long)pt * 3 * G;
  int* pc = point_cnt + (long long)pt * G;
  const int s0 = WRITE ? offs[pt] : 0;
  unsigned long long taken0 = 0, taken1 = 0;        // pooling_type 1, count pass: sub-voxels already filled
  int rec = 0;
  bool done = false;
  for (int k0 = 0; k0 < n && !done; k0 += 64) {
    const int k = k0 + lane;
    bool hit = false;
    float lx = 0.f, ly = 0.f, lz = 0.f;
    int g = 0;
    if (k < n) {
      const float* p = support_xyz + (long long)(start + k) * 3;
      lx = p[0] - qx; ly = p[1] - qy; lz = p[2] - qz;
      hit = vp_in_range(lx, ly, lz, dist, neighbor_type);
      if (hit) {
        const int ix = (int)floorf((lx + dist) / sx), iy = (int)floorf((ly + dist) / sy), iz = (int)floorf((lz + dist) / sz);
        g = ix * gy * gz + iy * gz + iz;
        g = g < 0 ? 0 : (g > G - 1 ? G - 1 : g);
      }
    }
    unsigned long long mask = __ballot(hit);
    if (pooling_type == 0 && !WRITE) {              // every hit is a row, up to nsample
      rec += __popcll(mask);
      if (nsample > 0 && rec >= nsample) { rec = nsample; done = true; }
      continue;
    }
    while (mask && !done) {                          // hits in ascending k, one at a time (wave-uniform)
      const int b = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int gu = __shfl(g, b, 64);
      const int kk = k0 + b;
      bool take = true;
      if (pooling_type != 0) {
        if (WRITE) {
          take = pc[gu] == 0;
        } else {
          const unsigned long long bit = 1ull << (gu & 63);
          take = !((gu < 64 ? taken0 : taken1) & bit);
          if (take) { if (gu < 64) taken0 |= bit; else taken1 |= bit; }
        }
      }
      if (!take) continue;
      if (WRITE) {
        const float* f = support_features + (long long)(start + kk) * cin;
        for (int c = lane; c < cg; c += 64) {
          if (pooling_type == 0) {
            float acc = nf[gu * cg + c];
            for (int q = 0; q < folds; ++q) acc += f[c + q * cg];   // i ascending: the folds of channel c in order
            nf[gu * cg + c] = acc;
          } else {
            nf[gu * cg + c] = f[c + (folds - 1) * cg];              // plain assignments: the last fold stays
          }
        }
        const float lxu = __shfl(lx, b, 64), lyu = __shfl(ly, b, 64), lzu = __shfl(lz, b, 64);
        if (lane == 0) {
          pc[gu] += 1;
          if (use_xyz) {
            if (pooling_type == 0) { nl[gu * 3] += lxu; nl[gu * 3 + 1] += lyu; nl[gu * 3 + 2] += lzu; }
            else { nl[gu * 3] = lxu; nl[gu * 3 + 1] = lyu; nl[gu * 3 + 2] = lzu; }
          }
          int* row = grouped_idxs + (long long)(s0 + rec) * 3;
          row[0] = start + kk; row[1] = pt; row[2] = gu;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // pc / nf of this point are re-read by the wave
      }
      ++rec;
      if (pooling_type == 0) done = nsample > 0 && rec >= nsample;
      else done = (nsample > 0 && rec >= nsample) || rec >= G;
    }
  }
  if (!WRITE && lane == 0) counts[pt] = rec;
}

// ---- gradient of the average pooling (vector_pool_gpu.cu:433-460)
__global__ void k_vp_grad(const float* __restrict__ grad_new, const int* __restrict__ point_cnt,
                          const int* __restrict__ grouped_idxs, long long total, int cin, int cout, int G,
                          float* __restrict__ grad_support) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const long long e = t / cin;
  const int c = (int)(t - e * cin), cg = cout / G;
  const int s = grouped_idxs[e * 3], p = grouped_idxs[e * 3 + 1], g = grouped_idxs[e * 3 + 2];
  const float w = 1 / fmaxf((float)point_cnt[(long long)p * G + g], 1.0f);
  atomicAdd(grad_support + (long long)s * cin + c, grad_new[(long long)p * cout + g * cg + c % cg] * w);
}

extern "C" size_t glx_vector_pool_workspace_bytes(int M) {
  return 2 * glx_align((size_t)(M + 1) * sizeof(int)) + glx_scan_workspace_bytes(M + 1) + 256;
}

static int vp_ws(void* workspace, size_t bytes, int M, int** counts, int** offs, void** scan_ws, const char* who) {
  if (!workspace || bytes < glx_vector_pool_workspace_bytes(M) - 256) {
    glx_set_error("%s: workspace %zu < %zu bytes", who, bytes, glx_vector_pool_workspace_bytes(M) - 256);
    return GLX_EWORKSPACE;
  }
  const size_t rowb = glx_align((size_t)(M + 1) * sizeof(int));
  *counts = (int*)workspace;
  *offs = (int*)((char*)workspace + rowb);
  *scan_ws = (char*)workspace + 2 * rowb;
  return GLX_OK;
}

extern "C" int glx_query_stacked_local_neighbor_idxs(const float* support_xyz, const int32_t* xyz_batch_cnt,
                                                     const float* new_xyz, const int32_t* new_xyz_batch_cnt, int B,
                                                     int M, int32_t* stack_neighbor_idxs, int32_t* start_len,
                                                     int32_t* cumsum, int avg_length, float max_dist, int nsample,
                                                     int neighbor_type, void* workspace, size_t workspace_bytes,
                                                     void* stream) {
  GLX_REQUIRE(cumsum && B > 0 && (M == 0 || (support_xyz && xyz_batch_cnt && new_xyz && new_xyz_batch_cnt &&
                                             stack_neighbor_idxs && start_len)),
              "glx_query_stacked_local_neighbor_idxs: null pointer");
  int *counts, *offs;
  void* scan_ws;
  int rc = vp_ws(workspace, workspace_bytes, M, &counts, &offs, &scan_ws, "glx_query_stacked_local_neighbor_idxs");
  if (rc != GLX_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    GLX_HIP(hipMemsetAsync(cumsum, 0, sizeof(int), st));
    return GLX_OK;
  }
  const dim3 grid(glx_divup(M, VP_WAVES)), block(64 * VP_WAVES);
  hipLaunchKernelGGL((k_vp_neighbors<false>), grid, block, 0, st, support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                     B, M, max_dist, nsample, neighbor_type, 0, counts, (const int*)nullptr, (int*)nullptr, (int*)nullptr);
  rc = glx_exclusive_scan(IntArray{counts}, (long long)M, offs, cumsum, scan_ws, glx_scan_workspace_bytes(M + 1), st);
  if (rc != GLX_OK) return rc;
  hipLaunchKernelGGL((k_vp_neighbors<true>), grid, block, 0, st, support_xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt,
                     B, M, max_dist, nsample, neighbor_type, avg_length * M, counts, (const int*)offs, stack_neighbor_idxs,
                     start_len);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_query_three_nn_by_stacked_local_idxs(const float* support_xyz, const float* new_xyz_grid_centers,
                                                        int32_t* new_xyz_grid_idxs, float* new_xyz_grid_dist2,
                                                        const int32_t* stack_neighbor_idxs, const int32_t* start_len,
                                                        int M, int num_total_grids, void* stream) {
  const long long total = (long long)M * num_total_grids;
  if (total <= 0) return GLX_OK;
  GLX_REQUIRE(support_xyz && new_xyz_grid_centers && new_xyz_grid_idxs && new_xyz_grid_dist2 && stack_neighbor_idxs &&
                  start_len, "glx_query_three_nn_by_stacked_local_idxs: null pointer");
  hipLaunchKernelGGL(k_vp_three_nn, dim3((unsigned)glx_divup(total, 256)), dim3(256), 0, (hipStream_t)stream, support_xyz,
                     new_xyz_grid_centers, new_xyz_grid_idxs, new_xyz_grid_dist2, stack_neighbor_idxs, start_len, M,
                     num_total_grids);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_vector_pool(const float* support_xyz, const float* support_features, const int32_t* xyz_batch_cnt,
                               const float* new_xyz, const int32_t* new_xyz_batch_cnt, int B, int M, int num_c_in,
                               int num_c_out, int num_grid_x, int num_grid_y, int num_grid_z, float max_dist,
                               int use_xyz, int num_max_sum_points, int nsample, int neighbor_type, int pooling_type,
                               float* new_features, float* new_local_xyz, int32_t* point_cnt_of_grid,
                               int32_t* grouped_idxs, int32_t* cum_sum, void* workspace, size_t workspace_bytes,
                               void* stream) {
  const int G = num_grid_x * num_grid_y * num_grid_z;
  GLX_REQUIRE(G > 0 && G <= 128 && num_c_out % G == 0 && num_c_in % (num_c_out / G) == 0,
              "glx_vector_pool: grids %d (<= 128), channels in %d out %d", G, num_c_in, num_c_out);
  GLX_REQUIRE(cum_sum && B > 0 && (M == 0 || (support_xyz && support_features && xyz_batch_cnt && new_xyz &&
                                              new_xyz_batch_cnt && new_features && new_local_xyz && point_cnt_of_grid &&
                                              grouped_idxs)),
              "glx_vector_pool: null pointer");
  int *counts, *offs;
  void* scan_ws;
  int rc = vp_ws(workspace, workspace_bytes, M, &counts, &offs, &scan_ws, "glx_vector_pool");
  if (rc != GLX_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    GLX_HIP(hipMemsetAsync(cum_sum, 0, sizeof(int), st));
    return GLX_OK;
  }
  const dim3 grid(glx_divup(M, VP_WAVES)), block(64 * VP_WAVES);
  hipLaunchKernelGGL((k_vp_pool<false>), grid, block, 0, st, support_xyz, support_features, xyz_batch_cnt, new_xyz,
                     new_xyz_batch_cnt, B, M, num_c_in, num_c_out, num_grid_x, num_grid_y, num_grid_z, max_dist, use_xyz,
                     nsample, neighbor_type, pooling_type, counts, (const int*)nullptr, new_features, new_local_xyz,
                     point_cnt_of_grid, grouped_idxs);
  rc = glx_exclusive_scan(IntArray{counts}, (long long)M, offs, cum_sum, scan_ws, glx_scan_workspace_bytes(M + 1), st);
  if (rc != GLX_OK) return rc;
  // the caller's retry loop (pointnet2_utils.py:399-416) re-allocates when the rows do not fit: find that out
  // before writing anything (the reference writes a truncated result the caller then throws away)
  int total = 0;
  GLX_HIP(hipMemcpyAsync(&total, cum_sum, sizeof(int), hipMemcpyDeviceToHost, st));
  GLX_HIP(hipStreamSynchronize(st));
  if (total > num_max_sum_points) return GLX_OK;
  hipLaunchKernelGGL((k_vp_pool<true>), grid, block, 0, st, support_xyz, support_features, xyz_batch_cnt, new_xyz,
                     new_xyz_batch_cnt, B, M, num_c_in, num_c_out, num_grid_x, num_grid_y, num_grid_z, max_dist, use_xyz,
                     nsample, neighbor_type, pooling_type, counts, (const int*)offs, new_features, new_local_xyz,
                     point_cnt_of_grid, grouped_idxs);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_vector_pool_grad(const float* grad_new_features, const int32_t* point_cnt_of_grid,
                                    const int32_t* grouped_idxs, int num_idxs, int num_c_in, int num_c_out,
                                    int num_total_grids, float* grad_support_features, void* stream) {
  const long long total = (long long)num_idxs * num_c_in;
  if (total <= 0) return GLX_OK;
  GLX_REQUIRE(grad_new_features && point_cnt_of_grid && grouped_idxs && grad_support_features && num_total_grids > 0,
              "glx_vector_pool_grad: null pointer");
  hipLaunchKernelGGL(k_vp_grad, dim3((unsigned)glx_divup(total, 256)), dim3(256), 0, (hipStream_t)stream, grad_new_features,
                     point_cnt_of_grid, grouped_idxs, total, num_c_in, num_c_out, num_total_grids, grad_support_features);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
