// Rotated-box BEV overlap / IoU, rotated NMS (bit-mask + on-device sweep) and GLENet's
// variance-voting NMS.
//
// Arithmetic restates, operation for operation in fp32 with contraction off:
//   pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:35-234 (== iou3d_cpu.cpp:59-229)   [NMSCONV]
//   pcdet/ops/iou3d/src/iou3d_kernel.cu:50-268 (== iou3d/src/iou3d_cpu.cpp:36-253) [OLDCONV]
// sin/cos/atan2 are evaluated in double and rounded once to float (the CPU reference calls
// glibc's float routines, which are correctly rounded for all but ~1e-3 of inputs).
// NMS: the reference computes the full N x ceil(N/64) suppression matrix, copies it to the
// host and sweeps it serially (iou3d_nms.cpp:90-136).  Here only the upper triangle is
// computed and the sweep runs on the device in one wave, 64 boxes per step.
#include "glx_common.h"
#include "glx_fill.h"

struct P2 {
  float x, y;
};

__device__ __forceinline__ float f_cos(float a) { return (float)cos((double)a); }
__device__ __forceinline__ float f_sin(float a) { return (float)sin((double)a); }
__device__ __forceinline__ float f_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }

__device__ __forceinline__ float cross2(P2 a, P2 b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float cross3(P2 p1, P2 p2, P2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
__device__ __forceinline__ float mn(float a, float b) { return a > b ? b : a; }
__device__ __forceinline__ float mx(float a, float b) { return a > b ? a : b; }

__device__ __forceinline__ int check_rect_cross(P2 p1, P2 p2, P2 q1, P2 q2) {
  return mn(p1.x, p2.x) <= mx(q1.x, q2.x) && mn(q1.x, q2.x) <= mx(p1.x, p2.x) &&
         mn(p1.y, p2.y) <= mx(q1.y, q2.y) && mn(q1.y, q2.y) <= mx(p1.y, p2.y);
}

#define IOU_EPS 1e-8f

__device__ __forceinline__ int seg_intersection(P2 p1, P2 p0, P2 q1, P2 q0, P2& ans) {
  if (check_rect_cross(p0, p1, q0, q1) == 0) return 0;
  float s1 = cross3(q0, p1, p0);
  float s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0);
  float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > IOU_EPS) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

// Box in the form both conventions reduce to: axis-aligned corners before rotation, centre,
// cos/sin of the heading.  OLD = iou3d library ([x1,y1,x2,y2,ry], +sin rotation, margin 1e-5),
// otherwise iou3d_nms ([x,y,z,dx,dy,dz,heading], (cos,-sin;sin,cos), margin 1e-2).
template <bool OLD>
struct RBox {
  float x1, y1, x2, y2, cx, cy, c, s;      // c,s = cos/sin(angle)
  float nc, ns;                            // cos/sin(-angle) for the inside test
  float hx, hy;                            // NMS convention: dx/2, dy/2
  __device__ void load(const float* b) {
    float ang;
    if (OLD) {
      x1 = b[0]; y1 = b[1]; x2 = b[2]; y2 = b[3]; ang = b[4];
      cx = (x1 + x2) / 2; cy = (y1 + y2) / 2;
      hx = hy = 0.f;
    } else {
      ang = b[6];
      hx = b[3] / 2; hy = b[4] / 2;
      x1 = b[0] - hx; y1 = b[1] - hy; x2 = b[0] + hx; y2 = b[1] + hy;
      cx = b[0]; cy = b[1];
    }
    c = f_cos(ang); s = f_sin(ang);
    nc = f_cos(-ang); ns = f_sin(-ang);
  }
  __device__ P2 rotate(P2 p) const {
    P2 r;
    if (OLD) {   // iou3d_kernel.cu:115-119
      r.x = (p.x - cx) * c + (p.y - cy) * s + cx;
      r.y = -(p.x - cx) * s + (p.y - cy) * c + cy;
    } else {     // iou3d_nms_kernel.cu:94-98
      r.x = (p.x - cx) * c + (p.y - cy) * (-s) + cx;
      r.y = (p.x - cx) * s + (p.y - cy) * c + cy;
    }
    return r;
  }
  __device__ int contains(P2 p) const {
    if (OLD) {   // iou3d_kernel.cu:50-65, MARGIN 1e-5
      const float MARGIN = 1e-5f;
      float rx = (p.x - cx) * nc + (p.y - cy) * ns + cx;
      float ry = -(p.x - cx) * ns + (p.y - cy) * nc + cy;
      return (rx > x1 - MARGIN && rx < x2 + MARGIN && ry > y1 - MARGIN && ry < y2 + MARGIN);
    } else {     // iou3d_nms_kernel.cu:51-61, MARGIN 1e-2
      const float MARGIN = 1e-2f;
      float rx = (p.x - cx) * nc + (p.y - cy) * (-ns);
      float ry = (p.x - cx) * ns + (p.y - cy) * nc;
      return (fabsf(rx) < hx + MARGIN && fabsf(ry) < hy + MARGIN);
    }
  }
};

template <bool OLD>
__device__ float box_overlap(const RBox<OLD>& A, const RBox<OLD>& B) {
  P2 ca[5], cb[5];
  ca[0] = A.rotate(P2{A.x1, A.y1}); ca[1] = A.rotate(P2{A.x2, A.y1});
  ca[2] = A.rotate(P2{A.x2, A.y2}); ca[3] = A.rotate(P2{A.x1, A.y2}); ca[4] = ca[0];
  cb[0] = B.rotate(P2{B.x1, B.y1}); cb[1] = B.rotate(P2{B.x2, B.y1});
  cb[2] = B.rotate(P2{B.x2, B.y2}); cb[3] = B.rotate(P2{B.x1, B.y2}); cb[4] = cb[0];
  P2 pts[16];
  P2 ctr{0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      P2 ip;
      if (seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], ip)) {
        ctr.x = ctr.x + ip.x; ctr.y = ctr.y + ip.y;
        pts[cnt++] = ip;
      }
    }
  for (int k = 0; k < 4; k++) {
    if (A.contains(cb[k])) { ctr.x = ctr.x + cb[k].x; ctr.y = ctr.y + cb[k].y; pts[cnt++] = cb[k]; }
    if (B.contains(ca[k])) { ctr.x = ctr.x + ca[k].x; ctr.y = ctr.y + ca[k].y; pts[cnt++] = ca[k]; }
  }
  if (cnt == 0) return 0.f;   // reference divides by zero then loops over nothing: area 0
  ctr.x /= cnt; ctr.y /= cnt;
  // bubble sort by polar angle about the centroid (same comparisons, same tie behaviour)
  float ang[16];
  for (int i = 0; i < cnt; i++) ang[i] = f_atan2(pts[i].y - ctr.y, pts[i].x - ctr.x);
  for (int j = 0; j < cnt - 1; j++)
    for (int i = 0; i < cnt - j - 1; i++)
      if (ang[i] > ang[i + 1]) {
        P2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
        float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; k++) {
    P2 a{pts[k].x - pts[0].x, pts[k].y - pts[0].y};
    P2 b{pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y};
    area += cross2(a, b);
  }
  return (float)(fabs((double)area) / 2.0);
}

template <bool OLD>
__device__ __forceinline__ float box_area(const float* b) {
  return OLD ? (b[2] - b[0]) * (b[3] - b[1]) : b[3] * b[4];
}

// mode 0: overlap area, 1: IoU
template <bool OLD, int STRIDE>
__global__ void k_pairwise(const float* __restrict__ a, int N, const float* __restrict__ b, int M,
                           int mode, float* __restrict__ out) {
  // 16x16 tile of (a row, b col); b boxes of the tile are prepared once in LDS
  __shared__ float sb[16 * STRIDE];
  const int j0 = blockIdx.x * 16, i0 = blockIdx.y * 16;
  const int tid = threadIdx.y * 16 + threadIdx.x;
  for (int e = tid; e < 16 * STRIDE; e += 256) {
    int jj = j0 + e / STRIDE;
    sb[e] = jj < M ? b[(long long)jj * STRIDE + e % STRIDE] : 0.f;
  }
  __syncthreads();
  const int i = i0 + threadIdx.y, j = j0 + threadIdx.x;
  if (i >= N || j >= M) return;
  RBox<OLD> A, B;
  A.load(a + (long long)i * STRIDE);
  B.load(sb + threadIdx.x * STRIDE);
  float s = box_overlap<OLD>(A, B);
  if (mode == 1) {
    float sa = box_area<OLD>(a + (long long)i * STRIDE), sbb = box_area<OLD>(sb + threadIdx.x * STRIDE);
    s = s / fmaxf(sa + sbb - s, IOU_EPS);
  }
  out[(long long)i * M + j] = s;
}

extern "C" int glx_boxes_overlap_bev(const float* boxes_a, int N, const float* boxes_b, int M,
                                     int iou, float* out, void* stream) {
  if (N == 0 || M == 0) return GLX_OK;
  GLX_REQUIRE(boxes_a && boxes_b && out, "glx_boxes_overlap_bev: null pointer");
  dim3 grid(glx_divup(M, 16), glx_divup(N, 16)), block(16, 16);
  hipLaunchKernelGGL((k_pairwise<false, 7>), grid, block, 0, (hipStream_t)stream, boxes_a, N,
                     boxes_b, M, iou ? 1 : 0, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_iou3d_boxes_overlap_bev(const float* boxes_a, int N, const float* boxes_b, int M,
                                           int iou, float* out, void* stream) {
  if (N == 0 || M == 0) return GLX_OK;
  GLX_REQUIRE(boxes_a && boxes_b && out, "glx_iou3d_boxes_overlap_bev: null pointer");
  dim3 grid(glx_divup(M, 16), glx_divup(N, 16)), block(16, 16);
  hipLaunchKernelGGL((k_pairwise<true, 5>), grid, block, 0, (hipStream_t)stream, boxes_a, N,
                     boxes_b, M, iou ? 1 : 0, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// boxes_aligned_overlap_kernel, iou3d_kernel.cu:284-293: out[i] = overlap(a[i], b[i])
__global__ void k_aligned_overlap(const float* __restrict__ a, const float* __restrict__ b, int N,
                                  float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  RBox<true> A, B;
  A.load(a + (long long)i * 5);
  B.load(b + (long long)i * 5);
  out[i] = box_overlap<true>(A, B);
}

extern "C" int glx_iou3d_boxes_aligned_overlap_bev(const float* boxes_a, const float* boxes_b,
                                                   int N, float* out, void* stream) {
  if (N == 0) return GLX_OK;
  GLX_REQUIRE(boxes_a && boxes_b && out, "glx_iou3d_boxes_aligned_overlap_bev: null pointer");
  hipLaunchKernelGGL(k_aligned_overlap, dim3(glx_divup(N, 64)), dim3(64), 0, (hipStream_t)stream,
                     boxes_a, boxes_b, N, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ NMS
__device__ __forceinline__ float iou_normal(const float* a, const float* b) {
  // iou3d_nms_kernel.cu:314-325
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  float interS = width * height;
  float Sa = a[3] * a[4], Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, IOU_EPS);
}

// Upper-triangular suppression matrix: block (col tile c >= row tile r), 64 threads = 64 rows.
// mask[i][c] bit j = iou(box i, box 64c+j) > thresh for 64c+j > i  (nms_kernel, :267-311).
template <bool NORMAL>
__global__ void k_nms_mask(const float* __restrict__ boxes, int N, float thresh, int col_blocks,
                           unsigned long long* __restrict__ mask) {
  // linear block id -> (r, c) with c >= r
  int b = blockIdx.x;
  int r = 0;
  // rows r has (col_blocks - r) tiles; find r by subtraction (col_blocks <= ~1000: cheap)
  int rem = b;
  while (rem >= col_blocks - r) { rem -= col_blocks - r; ++r; }
  const int c = r + rem;
  __shared__ float sb[64 * 7];
  const int t = threadIdx.x;
  const int jbase = c * 64;
  const int col_size = min(N - jbase, 64);
  if (t < col_size)
    for (int e = 0; e < 7; ++e) sb[t * 7 + e] = boxes[(long long)(jbase + t) * 7 + e];
  __syncthreads();
  const int i = r * 64 + t;
  if (i >= N) return;
  unsigned long long bits = 0;
  const float* bi = boxes + (long long)i * 7;
  int start = (r == c) ? t + 1 : 0;
  if (NORMAL) {
    for (int j = start; j < col_size; ++j)
      if (iou_normal(bi, sb + j * 7) > thresh) bits |= 1ull << j;
  } else {
    RBox<false> A;
    A.load(bi);
    float sa = bi[3] * bi[4];
    for (int j = start; j < col_size; ++j) {
      RBox<false> B;
      B.load(sb + j * 7);
      float s = box_overlap<false>(A, B);
      float iou = s / fmaxf(sa + sb[j * 7 + 3] * sb[j * 7 + 4] - s, IOU_EPS);
      if (iou > thresh) bits |= 1ull << j;
    }
  }
  mask[(long long)i * col_blocks + c] = bits;
}

// One wave sweeps the matrix 64 boxes at a time (host loop of iou3d_nms.cpp:119-132):
// lane j holds the removed-bits of column word j, j+64, ... ; inside a 64-box word the greedy
// chain runs on the diagonal tile only.
__global__ void k_nms_sweep(const unsigned long long* __restrict__ mask, int N, int col_blocks,
                            long long* __restrict__ keep, int* __restrict__ num_out) {
  extern __shared__ unsigned long long remv[];   // col_blocks words
  const int lane = threadIdx.x;
  for (int w = lane; w < col_blocks; w += 64) remv[w] = 0ull;
  __syncthreads();
  int num = 0;
  for (int b = 0; b < col_blocks; ++b) {
    const int base = b * 64;
    const int nb = min(64, N - base);
    // diagonal tile: lane j holds row (base+j)'s word b (bits of later boxes in this word)
    unsigned long long diag = (lane < nb) ? mask[(long long)(base + lane) * col_blocks + b] : 0ull;
    unsigned long long cur = remv[b];
    unsigned long long kept = 0ull;
    for (int j = 0; j < nb; ++j) {
      unsigned long long dj = __shfl(diag, j, 64);
      if (!((cur >> j) & 1ull)) {
        kept |= 1ull << j;
        cur |= dj;
      }
    }
    // emit kept indices in order, OR their rows into the later words
    unsigned long long k2 = kept;
    while (k2) {
      int j = __ffsll((long long)k2) - 1;
      k2 &= k2 - 1;
      if (lane == 0) keep[num] = base + j;
      ++num;
      const unsigned long long* row = mask + (long long)(base + j) * col_blocks;
      for (int w = b + 1 + lane; w < col_blocks; w += 64) remv[w] |= row[w];
    }
    __syncthreads();
  }
  if (lane == 0) *num_out = num;
}

extern "C" size_t glx_nms_workspace_bytes(int N) {
  size_t cb = (size_t)((N + 63) / 64);
  return glx_align((size_t)(N > 0 ? N : 1) * cb * 8) + 256;
}

extern "C" int glx_nms(const float* boxes_sorted, int N, float thresh, int normal, int64_t* keep,
                       int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(keep && num_out, "glx_nms: null output");
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    GlxFillJob job{num_out, sizeof(int), 0};
    int frc = glx_fill_multi(&job, 1, st);
    if (frc != GLX_OK) return frc;
    return GLX_OK;
  }
  GLX_REQUIRE(boxes_sorted, "glx_nms: null boxes");
  int col_blocks = (N + 63) / 64;
  size_t need = (size_t)N * col_blocks * 8;
  if (!workspace || workspace_bytes < need) {
    glx_set_error("glx_nms: workspace %zu < %zu bytes", workspace_bytes, need);
    return GLX_EWORKSPACE;
  }
  unsigned long long* mask = (unsigned long long*)workspace;
  // lower-triangle words are never written by the mask kernel but the sweep only reads
  // words >= the row's own block, so no clearing is needed.
  int ntiles = col_blocks * (col_blocks + 1) / 2;
  if (normal)
    hipLaunchKernelGGL((k_nms_mask<true>), dim3(ntiles), dim3(64), 0, st, boxes_sorted, N, thresh,
                       col_blocks, mask);
  else
    hipLaunchKernelGGL((k_nms_mask<false>), dim3(ntiles), dim3(64), 0, st, boxes_sorted, N, thresh,
                       col_blocks, mask);
  hipLaunchKernelGGL(k_nms_sweep, dim3(1), dim3(64), (size_t)col_blocks * 8, st,
                     (const unsigned long long*)mask, N, col_blocks, (long long*)keep, num_out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ GLENet variance-voting NMS
// nms_func, pcdet/ops/iou3d_nms/iou3d_nms_utils.py:227-273, one block, fp32 like numpy.
// ious (N,N) precomputed on the ORIGINAL boxes (:235); boxes/scores are updated in place.
#define VOTE_THREADS 1024

__device__ __forceinline__ float block_sum(float v, float* red) {
  // fixed-shape tree: deterministic
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = VOTE_THREADS / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  float r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(VOTE_THREADS) void k_nms_vote(
    float* __restrict__ boxes, float* __restrict__ scores, const float* __restrict__ variance,
    int var_stride, const float* __restrict__ ious, int N, float iou_thr, float score_thr) {
  __shared__ float red[VOTE_THREADS];
  __shared__ int redi[VOTE_THREADS];
  __shared__ int s_idx;
  const int t = threadIdx.x;
  const float PI = 3.14159265358979323846f;          // float32(np.pi)
  const float PI_3_2 = (float)(3.14159265358979323846 * 3 / 2);
  const float PI_2x = (float)(3.14159265358979323846 * 2);
  const float PI_4 = (float)(3.14159265358979323846 / 4);
  (void)PI;
  // undone mask lives in registers: each thread owns boxes t, t+1024, ...
  constexpr int PER = 4;   // N <= 4096
  bool undone[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    int i = t + u * VOTE_THREADS;
    undone[u] = i < N && scores[i] >= score_thr;
  }
  while (true) {
    // argmax of scores over undone boxes, first index on ties (np.argmax)
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      int i = t + u * VOTE_THREADS;
      if (undone[u]) {
        float s = scores[i];
        if (s > best || (s == best && i < bi)) { best = s; bi = i; }
      }
    }
    red[t] = best; redi[t] = bi;
    __syncthreads();
    for (int s = VOTE_THREADS / 2; s > 0; s >>= 1) {
      if (t < s) {
        float o = red[t + s]; int oi = redi[t + s];
        if (oi != 0x7fffffff && (redi[t] == 0x7fffffff || o > red[t] || (o == red[t] && oi < redi[t]))) {
          red[t] = o; redi[t] = oi;
        }
      }
      __syncthreads();
    }
    if (t == 0) s_idx = redi[0];
    __syncthreads();
    const int idx = s_idx;
    __syncthreads();
    if (idx == 0x7fffffff) break;   // undone_mask.sum() == 0

    if (variance) {
      const float top_h = boxes[(long long)idx * 7 + 6];
      // per-thread partial sums of pi (7) and pi * box (7)
      float sp[7], sb[7];
#pragma unroll
      for (int c = 0; c < 7; ++c) { sp[c] = 0.f; sb[c] = 0.f; }
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        int i = t + u * VOTE_THREADS;
        if (!undone[u]) continue;
        float iou = ious[(long long)i * N + idx];
        if (!(iou > iou_thr)) continue;
        float bx[7];
#pragma unroll
        for (int c = 0; c < 7; ++c) bx[c] = boxes[(long long)i * 7 + c];
        if (fabsf(bx[6] - top_h) >= PI_3_2) bx[6] = top_h > 0.f ? bx[6] + PI_2x : bx[6] - PI_2x;
        float d = 1.f - iou;
        float p = expf(-1.f * (d * d) / 0.05f);
        bool far = fabsf(bx[6] - top_h) >= PI_4;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          float w = p / variance[(long long)i * var_stride + c];
          if (c == 6 && far) w = 0.f;
          sp[c] += w;
          sb[c] += w * bx[c];
        }
      }
      float nb[7];
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        float tp = block_sum(sp[c], red);
        float tb = block_sum(sb[c], red);
        nb[c] = tb / tp;    // == sum((pi / pi.sum) * box)
      }
      if (t == 0) {
#pragma unroll
        for (int c = 0; c < 7; ++c) boxes[(long long)idx * 7 + c] = nb[c];
      }
    }
    // undone[idx] = False; scores[undone] *= (iou < thr); undone[scores < score_thr] = False
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      int i = t + u * VOTE_THREADS;
      if (i == idx) undone[u] = false;
      if (undone[u]) {
        float s = scores[i] * ((ious[(long long)i * N + idx] < iou_thr) ? 1.f : 0.f);
        scores[i] = s;
      }
      if (i < N && scores[i] < score_thr) undone[u] = false;
    }
    __syncthreads();
  }
}

extern "C" int glx_nms_vote(float* boxes, float* scores, const float* variance, int var_stride,
                            const float* ious, int N, float iou_thr, float score_thr,
                            void* stream) {
  if (N == 0) return GLX_OK;
  GLX_REQUIRE(boxes && scores && ious, "glx_nms_vote: null pointer");
  GLX_REQUIRE(N <= 4 * VOTE_THREADS, "glx_nms_vote: N=%d exceeds %d (NMS_PRE_MAXSIZE)", N,
              4 * VOTE_THREADS);
  GLX_REQUIRE(!variance || var_stride >= 7, "glx_nms_vote: variance needs >= 7 columns");
  hipLaunchKernelGGL(k_nms_vote, dim3(1), dim3(VOTE_THREADS), 0, (hipStream_t)stream, boxes,
                     scores, variance, var_stride, ious, N, iou_thr, score_thr);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
