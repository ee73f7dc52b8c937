// pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda on CDNA4: the batch-layout PointNet++ operators
// (PointNet2MSG backbone, PointRCNN head; pointnet2_batch/src/pointnet2_api.cpp:10-24).  Layout differs from
// the stacked family in glx_points.hip: B equal frames, features CHANNEL-major (B, C, N), indices local to the
// frame.  Farthest point sampling and three_nn are front ends of the stacked kernels (glx_batch_fps,
// glx_batch_three_nn in glx_points.hip); the channel-major gathers are written here.
//
// Common shape of the gather kernels: a thread owns one output column (a query point, or a (query, sample)
// pair), keeps its index (and weights) in registers and walks the C channel planes -- index traffic is paid
// once instead of C times (the reference launches a thread per (b, c, column) and re-reads idx for every
// channel), stores are coalesced along the column axis, the random reads stay inside one N-float plane
// (<= 64 KB, L2-resident).  The frame is a grid axis, so no thread divides by N.
#include "glx_common.h"

// ------------------------------------------------------------------ ball query
// ball_query_kernel_fast, pointnet2_batch/src/ball_query_gpu.cu:15-51: the first nsample points with
// d2 < r2 (strict) in index order, unused slots = the first hit; a ball without hits leaves its row as the
// caller passed it (BallQuery.forward zero-fills it, pointnet2_utils.py:236) -- no -1 sentinel here.
// A wave per query: 64 distance tests per step, a ballot keeps the hits in index order.
__global__ __launch_bounds__(256) void k_bq_batch(int n, int m, float radius2, int nsample,
                                                  const float* __restrict__ new_xyz,
                                                  const float* __restrict__ xyz, int* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int pt = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (pt >= m) return;
  const float* X = xyz + (long long)b * n * 3;
  const float* q = new_xyz + ((long long)b * m + pt) * 3;
  const float nx = q[0], ny = q[1], nz = q[2];
  int* o = idx + ((long long)b * m + pt) * nsample;
  int cnt = 0, first = -1;
  for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
    const int k = k0 + lane;
    bool in = false;
    if (k < n) {
      const float x = X[(long long)k * 3], y = X[(long long)k * 3 + 1], z = X[(long long)k * 3 + 2];
      const float d2 = (nx - x) * (nx - x) + (ny - y) * (ny - y) + (nz - z) * (nz - z);
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
  if (first >= 0)
    for (int l = (cnt < nsample ? cnt : nsample) + lane; l < nsample; l += 64) o[l] = first;
}

extern "C" int glx_batch_ball_query(int B, int n, int m, float radius, int nsample, const float* new_xyz,
                                    const float* xyz, int32_t* idx, void* stream) {
  if (B <= 0 || m <= 0 || nsample <= 0) return GLX_OK;
  GLX_REQUIRE(new_xyz && xyz && idx, "glx_batch_ball_query: null pointer");
  hipLaunchKernelGGL(k_bq_batch, dim3(glx_divup(m, 4), B), dim3(256), 0, (hipStream_t)stream, n, m,
                     radius * radius, nsample, new_xyz, xyz, idx);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ grouping / gathering
// group_points_kernel_fast (group_points_gpu.cu:57-78): out[b, c, p, s] = points[b, c, idx[b, p, s]];
// gather_points_kernel_fast (sampling_gpu.cu:14-33) is the same statement with nsample = 1, so both entry
// points run this kernel on `cols` = npoints * nsample index columns per frame.
__global__ __launch_bounds__(256) void k_cm_gather(int C, int n, long long cols, const float* __restrict__ points,
                                                   const int* __restrict__ idx, float* __restrict__ out) {
  const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (col >= cols) return;
  const int i = idx[(long long)b * cols + col];
  const float* P = points + (long long)b * C * n + i;
  float* O = out + (long long)b * C * cols + col;
  int c = 0;
  for (; c + 4 <= C; c += 4) {   // four independent gathers in flight
    const float v0 = P[(long long)c * n], v1 = P[(long long)(c + 1) * n], v2 = P[(long long)(c + 2) * n],
                v3 = P[(long long)(c + 3) * n];
    O[(long long)c * cols] = v0; O[(long long)(c + 1) * cols] = v1;
    O[(long long)(c + 2) * cols] = v2; O[(long long)(c + 3) * cols] = v3;
  }
  for (; c < C; ++c) O[(long long)c * cols] = P[(long long)c * n];
}

// group_points_grad_kernel_fast (group_points_gpu.cu:14-32) / gather_points_grad_kernel_fast
// (sampling_gpu.cu:52-71): grad_points[b, c, idx[b, col]] += grad_out[b, c, col]; grad_points arrives zeroed.
__global__ __launch_bounds__(256) void k_cm_scatter_add(int C, int n, long long cols,
                                                        const float* __restrict__ grad_out,
                                                        const int* __restrict__ idx, float* __restrict__ grad_points) {
  const long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (col >= cols) return;
  const int i = idx[(long long)b * cols + col];
  const float* G = grad_out + (long long)b * C * cols + col;
  float* P = grad_points + (long long)b * C * n + i;
  for (int c = 0; c < C; ++c) atomicAdd(P + (long long)c * n, G[(long long)c * cols]);
}

static int cm_gather(const char* who, int B, int C, int n, long long cols, const float* points, const int32_t* idx,
                     float* out, void* stream) {
  if (B <= 0 || C <= 0 || cols <= 0) return GLX_OK;
  GLX_REQUIRE(points && idx && out, "%s: null pointer", who);
  GLX_REQUIRE(B <= 65535, "%s: B = %d exceeds the grid's y extent", who, B);
  hipLaunchKernelGGL(k_cm_gather, dim3(glx_divup(cols, 256), B), dim3(256), 0, (hipStream_t)stream, C, n, cols,
                     points, idx, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

static int cm_scatter(const char* who, int B, int C, int n, long long cols, const float* grad_out,
                      const int32_t* idx, float* grad_points, void* stream) {
  if (B <= 0 || C <= 0 || cols <= 0) return GLX_OK;
  GLX_REQUIRE(grad_out && idx && grad_points, "%s: null pointer", who);
  GLX_REQUIRE(B <= 65535, "%s: B = %d exceeds the grid's y extent", who, B);
  hipLaunchKernelGGL(k_cm_scatter_add, dim3(glx_divup(cols, 256), B), dim3(256), 0, (hipStream_t)stream, C, n, cols,
                     grad_out, idx, grad_points);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_batch_group_points(int B, int C, int n, int npoints, int nsample, const float* points,
                                      const int32_t* idx, float* out, void* stream) {
  return cm_gather("glx_batch_group_points", B, C, n, (long long)npoints * nsample, points, idx, out, stream);
}

extern "C" int glx_batch_group_points_grad(int B, int C, int n, int npoints, int nsample, const float* grad_out,
                                           const int32_t* idx, float* grad_points, void* stream) {
  return cm_scatter("glx_batch_group_points_grad", B, C, n, (long long)npoints * nsample, grad_out, idx,
                    grad_points, stream);
}

extern "C" int glx_batch_gather_points(int B, int C, int n, int npoints, const float* points, const int32_t* idx,
                                       float* out, void* stream) {
  return cm_gather("glx_batch_gather_points", B, C, n, npoints, points, idx, out, stream);
}

extern "C" int glx_batch_gather_points_grad(int B, int C, int n, int npoints, const float* grad_out,
                                            const int32_t* idx, float* grad_points, void* stream) {
  return cm_scatter("glx_batch_gather_points_grad", B, C, n, npoints, grad_out, idx, grad_points, stream);
}

// ------------------------------------------------------------------ three-point interpolation
// three_interpolate_kernel_fast (interpolate_gpu.cu:84-106):
//   out[b, c, p] = w0 * points[b, c, i0] + w1 * points[b, c, i1] + w2 * points[b, c, i2]
// evaluated left to right as two rounded products-and-sums (contraction is off for this library), the
// reference's expression order.
__global__ __launch_bounds__(256) void k_cm_interp(int C, int m, int n, const float* __restrict__ points,
                                                   const int* __restrict__ idx, const float* __restrict__ weight,
                                                   float* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= n) return;
  const int* ip = idx + ((long long)b * n + p) * 3;
  const float* wp = weight + ((long long)b * n + p) * 3;
  const int i0 = ip[0], i1 = ip[1], i2 = ip[2];
  const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
  const float* P = points + (long long)b * C * m;
  float* O = out + (long long)b * C * n + p;
  for (int c = 0; c < C; ++c, P += m, O += n) *O = w0 * P[i0] + w1 * P[i1] + w2 * P[i2];
}

// three_interpolate_grad_kernel_fast (interpolate_gpu.cu:130-153); grad_points (B, C, m) arrives zeroed.
__global__ __launch_bounds__(256) void k_cm_interp_grad(int C, int n, int m, const float* __restrict__ grad_out,
                                                        const int* __restrict__ idx, const float* __restrict__ weight,
                                                        float* __restrict__ grad_points) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= n) return;
  const int* ip = idx + ((long long)b * n + p) * 3;
  const float* wp = weight + ((long long)b * n + p) * 3;
  const int i0 = ip[0], i1 = ip[1], i2 = ip[2];
  const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
  const float* G = grad_out + (long long)b * C * n + p;
  float* P = grad_points + (long long)b * C * m;
  for (int c = 0; c < C; ++c, G += n, P += m) {
    const float g = *G;
    atomicAdd(P + i0, g * w0);
    atomicAdd(P + i1, g * w1);
    atomicAdd(P + i2, g * w2);
  }
}

extern "C" int glx_batch_three_interpolate(int B, int C, int m, int n, const float* points, const int32_t* idx,
                                           const float* weight, float* out, void* stream) {
  if (B <= 0 || C <= 0 || n <= 0) return GLX_OK;
  GLX_REQUIRE(points && idx && weight && out, "glx_batch_three_interpolate: null pointer");
  hipLaunchKernelGGL(k_cm_interp, dim3(glx_divup(n, 256), B), dim3(256), 0, (hipStream_t)stream, C, m, n, points,
                     idx, weight, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_batch_three_interpolate_grad(int B, int C, int n, int m, const float* grad_out,
                                                const int32_t* idx, const float* weight, float* grad_points,
                                                void* stream) {
  if (B <= 0 || C <= 0 || n <= 0) return GLX_OK;
  GLX_REQUIRE(grad_out && idx && weight && grad_points, "glx_batch_three_interpolate_grad: null pointer");
  hipLaunchKernelGGL(k_cm_interp_grad, dim3(glx_divup(n, 256), B), dim3(256), 0, (hipStream_t)stream, C, n, m,
                     grad_out, idx, weight, grad_points);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
