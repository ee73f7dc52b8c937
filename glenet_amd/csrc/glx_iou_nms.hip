// Rotated-box BEV overlap / IoU, rotated NMS (bit-mask + on-device sweep) and GLENet's
// variance-voting NMS.
//
// Arithmetic restates, operation for operation in fp32 with contraction off:
//   pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:35-234 (== iou3d_cpu.cpp:59-229)   [NMSCONV]
//   pcdet/ops/iou3d/src/iou3d_kernel.cu:50-268 (== iou3d/src/iou3d_cpu.cpp:36-253) [OLDCONV]
// sin / cos / atan2 are glx_libm.h's: glibc's float routines restated (the CPU reference calls glibc's), verified
// against the host libm over all finite floats -- so an overlap is the reference's CPU value bit for bit.
// NMS: the reference computes the full N x ceil(N/64) suppression matrix, copies it to the
// host and sweeps it serially (iou3d_nms.cpp:90-136).  Here only the upper triangle is
// computed and the sweep runs on the device in one wave, 64 boxes per step.
#include "glx_common.h"
#include "glx_fill.h"
#include "glx_libm.h"

struct P2 {
  float x, y;
};

__device__ __forceinline__ float f_cos(float a) { return glxm::cosf_(a); }
__device__ __forceinline__ float f_sin(float a) { return glxm::sinf_(a); }
__device__ __forceinline__ float f_atan2(float y, float x) { return glxm::atan2f_(y, x); }

// Test hook: the routines above applied elementwise (fn 0 sinf, 1 cosf, 2 atanf, 3 atan2f(x, y)) so that
// tests/test_libm_gpu.py can compare the DEVICE build with the host libm bit for bit.
__global__ void k_libm_eval(int fn, const float* __restrict__ x, const float* __restrict__ y, long long n,
                            float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[i];
  float r;
  if (fn == 0) r = glxm::sinf_(a);
  else if (fn == 1) r = glxm::cosf_(a);
  else if (fn == 2) r = glxm::atanf_(a);
  else r = glxm::atan2f_(a, y[i]);
  out[i] = r;
}

extern "C" int glx_libm_eval(int fn, const float* x, const float* y, int64_t n, float* out, void* stream) {
  if (n <= 0) return GLX_OK;
  GLX_REQUIRE(fn >= 0 && fn <= 3 && x && out && (fn != 3 || y), "glx_libm_eval: bad arguments");
  hipLaunchKernelGGL(k_libm_eval, dim3(glx_divup(n, 256)), dim3(256), 0, (hipStream_t)stream, fn, x, y,
                     (long long)n, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

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

// ---- prepared boxes: the per-box part of the rotated overlap (corners' frame, heading trig in
// double rounded once, bounding radius) computed ONCE per box instead of once per pair.
struct PBox {
  float cx, cy, hx, hy, c, s, rad, area;
};

__global__ void k_nms_prepare(const float* __restrict__ boxes, int N, PBox* __restrict__ pb) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  boxes += (long long)blockIdx.y * N * 7;     // blockIdx.y = frame of a batched call
  pb += (long long)blockIdx.y * N;
  const float* b = boxes + (long long)i * 7;
  PBox p;
  p.cx = b[0]; p.cy = b[1];
  p.hx = b[3] / 2; p.hy = b[4] / 2;
  p.c = f_cos(b[6]); p.s = f_sin(b[6]);
  // cos(-a) == cos(a), sin(-a) == -sin(a) exactly (even / odd, also after rounding), so the inverse
  // rotation of the reference's inside test needs no second evaluation
  p.rad = sqrtf(p.hx * p.hx + p.hy * p.hy);
  p.area = b[3] * b[4];
  pb[i] = p;
}

__device__ __forceinline__ RBox<false> rbox_from(const PBox& p) {
  RBox<false> r;
  r.hx = p.hx; r.hy = p.hy;
  r.x1 = p.cx - p.hx; r.y1 = p.cy - p.hy; r.x2 = p.cx + p.hx; r.y2 = p.cy + p.hy;
  r.cx = p.cx; r.cy = p.cy;
  r.c = p.c; r.s = p.s; r.nc = p.c; r.ns = -p.s;
  return r;
}

// Boxes whose centres are further apart than the sum of their bounding radii (+ the 1e-2 margin
// of the corner-inside test, + slack for rounding) share no intersection point and no contained
// corner: the reference routine returns exactly 0 for them, so they can be skipped.
__device__ __forceinline__ bool pbox_far(const PBox& a, const PBox& b) {
  const float dx = a.cx - b.cx, dy = a.cy - b.cy, reach = a.rad + b.rad + 0.05f;
  return dx * dx + dy * dy > reach * reach;
}

// NMS only asks whether IoU > thresh.  A rigorous upper bound of the IoU that costs ~40 flops: along the direction u
// of the centre difference the intersection cannot be longer than the overlap of the two boxes' projection intervals,
// across it not wider than the narrower box's projection; so inter <= L * W, and IoU <= inter / (Aa + Ab - inter) is
// increasing in inter.  What has to be bounded is the area the exact routine COMPUTES, not the true intersection: its
// polygon also takes corners that lie up to MARGIN = 1e-2 outside the other box (the inside test of
// iou3d_nms_kernel.cu:51-61), so for small or thin boxes it exceeds the true intersection by far more than a rounding
// error.  Every vertex it can use -- true edge crossings, corners of b inside a grown by the margin, corners
// of a inside b grown by the margin -- lies in (a grown) n (b grown), a convex set that contains the polygon; the bound
// is therefore taken on the boxes GROWN by the margin on every side, while the union keeps the true areas, exactly as
// the reference's `sa + sb - s` does.  Pairs whose bound stays below the threshold (0.1 % on top for the bound's own
// rounding) cannot set a bit of the suppression matrix and skip the ~600-flop rotated overlap -- at thresh 0.8 that is
// nine of ten pairs that pass the far-apart test.  Box axes: heading rotates (x, y) to (x c - y s, x s + y c).
__device__ __forceinline__ bool pbox_iou_below(const PBox& a, const PBox& b, float thresh) {
  const float MARGIN = 1e-2f;
  float ux = b.cx - a.cx, uy = b.cy - a.cy;
  const float d2 = ux * ux + uy * uy;
  float dist = 0.f;
  if (d2 > 1e-12f) {
    const float inv = rsqrtf(d2);
    dist = d2 * inv;
    ux *= inv; uy *= inv;
  } else {
    ux = 1.f; uy = 0.f;
  }
  const float ahx = a.hx + MARGIN, ahy = a.hy + MARGIN, bhx = b.hx + MARGIN, bhy = b.hy + MARGIN;
  // half extents along u (h) and across it (w): |u . ex| hx + |u . ey| hy with ex = (c, s), ey = (-s, c)
  const float pa = fabsf(ux * a.c + uy * a.s), qa = fabsf(uy * a.c - ux * a.s);
  const float pb_ = fabsf(ux * b.c + uy * b.s), qb = fabsf(uy * b.c - ux * b.s);
  const float ha = pa * ahx + qa * ahy, wa = qa * ahx + pa * ahy;
  const float hb = pb_ * bhx + qb * bhy, wb = qb * bhx + pb_ * bhy;
  const float L = fminf(fmaxf(ha + hb - dist, 0.f), 2.f * fminf(ha, hb));
  float inter = L * 2.f * fminf(wa, wb);
  inter = fminf(inter, 4.f * fminf(ahx * ahy, bhx * bhy)) * 1.001f;
  const float un = a.area + b.area - inter;
  return un > 0.f && inter < thresh * un;       // bound / union < thresh  (NaN boxes: false -> exact routine decides)
}

// (N, M) overlap / IoU matrix in the iou3d_nms convention, one block per 64 x 64 tile: the 128 boxes
// of the tile are prepared once in LDS (trig per box, not per pair), pairs that are exactly zero by
// the far-apart test are written as 0 straight away, the survivors are queued and evaluated
// densely (a thread per surviving pair) -- same two-phase scheme as k_nms_mask.
__global__ __launch_bounds__(256) void k_pairwise_tile(const float* __restrict__ a, int N,
                                                       const float* __restrict__ b, int M, int mode,
                                                       float* __restrict__ out, long long si, long long sj,
                                                       long long a_frame, long long b_frame, long long out_frame,
                                                       const int* __restrict__ counts) {
  // blockIdx.z = frame of a batch of independent box lists (strides in elements); counts: live boxes per frame
  // (rows and columns past it are left untouched)
  a += a_frame * blockIdx.z; b += b_frame * blockIdx.z; out += out_frame * blockIdx.z;
  if (counts) { N = min(N, counts[blockIdx.z]); M = min(M, counts[blockIdx.z]); }
  __shared__ PBox s_row[64], s_col[64];
  __shared__ unsigned short s_q[64 * 64];
  __shared__ int s_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
  if (tid == 0) s_n = 0;
  if (tid < 128) {
    const bool isrow = tid < 64;
    const int idx = isrow ? i0 + tid : j0 + tid - 64;
    const int lim = isrow ? N : M;
    if (idx < lim) {
      const float* bx = (isrow ? a : b) + (long long)idx * 7;
      PBox p;
      p.cx = bx[0]; p.cy = bx[1];
      p.hx = bx[3] / 2; p.hy = bx[4] / 2;
      p.c = f_cos(bx[6]); p.s = f_sin(bx[6]);
      p.rad = sqrtf(p.hx * p.hx + p.hy * p.hy);
      p.area = bx[3] * bx[4];
      (isrow ? s_row : s_col)[isrow ? tid : tid - 64] = p;
    }
  }
  __syncthreads();
  const int j = j0 + lane;
  const bool jok = j < M;
  const PBox B = s_col[jok ? lane : 0];
  for (int it = 0; it < 16; ++it) {
    const int rl = wave * 16 + it, i = i0 + rl;
    if (i >= N) break;   // wave-uniform
    const PBox A = s_row[rl];
    const bool pass = jok && !pbox_far(A, B);
    if (jok && !pass) out[(long long)i * si + (long long)j * sj] = 0.f;
    const unsigned long long bal = __ballot(pass);
    if (bal) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_n, __popcll(bal));
      base = __shfl(base, 0, 64);
      if (pass) s_q[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)((rl << 6) | lane);
    }
  }
  __syncthreads();
  const int n = s_n;
  for (int e = tid; e < n; e += 256) {
    const int rl = s_q[e] >> 6, cl = s_q[e] & 63;
    const PBox A = s_row[rl], Bq = s_col[cl];
    float v = box_overlap<false>(rbox_from(A), rbox_from(Bq));
    if (mode == 1) v = v / fmaxf(A.area + Bq.area - v, IOU_EPS);
    out[(long long)(i0 + rl) * si + (long long)(j0 + cl) * sj] = v;
  }
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
  hipLaunchKernelGGL(k_pairwise_tile, dim3(glx_divup(M, 64), glx_divup(N, 64)), dim3(256), 0,
                     (hipStream_t)stream, boxes_a, N, boxes_b, M, iou ? 1 : 0, out, (long long)M, 1ll, 0ll, 0ll, 0ll,
                     (const int*)nullptr);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// `frames` independent lists of N boxes (frames, N, 7): out (frames, N, N) with out[f][j][i] = IoU(box i, box j) when
// transposed (what glx_nms_vote_batch reads), out[f][i][j] otherwise.  counts (frames) int32 device or NULL: only the
// leading counts[f] boxes of a frame are live, the rest of its matrix is not written.
extern "C" int glx_boxes_iou_bev_self_batch(const float* boxes, int frames, int N, const int32_t* counts, int transposed,
                                            float* out, void* stream) {
  if (frames <= 0 || N <= 0) return GLX_OK;
  GLX_REQUIRE(boxes && out, "glx_boxes_iou_bev_self_batch: null pointer");
  GLX_REQUIRE(frames <= 65535, "glx_boxes_iou_bev_self_batch: %d frames", frames);
  hipLaunchKernelGGL(k_pairwise_tile, dim3(glx_divup(N, 64), glx_divup(N, 64), frames), dim3(256), 0, (hipStream_t)stream,
                     boxes, N, boxes, N, 1, out, transposed ? 1ll : (long long)N, transposed ? (long long)N : 1ll,
                     (long long)N * 7, (long long)N * 7, (long long)N * N, counts);
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

// Upper-triangular suppression matrix, one block per 64 x 64 tile (col tile c >= row tile r)
// (nms_kernel, iou3d_nms_kernel.cu:267-311).  Two phases, because a wave that runs the rotated
// overlap for one surviving lane out of 64 wastes the other 63:
//   1. lane = column box, every wave walks 16 rows: pairs that pass the exact far-apart test are
//      appended to a queue in LDS (ballot + one LDS atomic per wave-row);
//   2. the queue is processed densely, a thread per surviving pair; hits are OR-ed into the
//      tile's 64 mask words in LDS.
// Layout: maskT[c][i] (column-block major) -- the sweep reads one column block for many rows.
template <bool NORMAL>
__global__ __launch_bounds__(256) void k_nms_mask(const float* __restrict__ boxes,
                                                  const PBox* __restrict__ pb, int N, float thresh,
                                                  int col_blocks,
                                                  unsigned long long* __restrict__ maskT) {
  int rem = blockIdx.x, r = 0;
  while (rem >= col_blocks - r) { rem -= col_blocks - r; ++r; }   // linear id -> (r, c >= r)
  const int c = r + rem;
  boxes += (long long)blockIdx.y * N * 7;                         // blockIdx.y = frame
  if (!NORMAL) pb += (long long)blockIdx.y * N;
  maskT += (long long)blockIdx.y * N * col_blocks;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ unsigned long long s_bits[64];
  __shared__ unsigned short s_q[64 * 64];
  __shared__ int s_n;
  __shared__ PBox s_col[64];
  if (tid < 64) s_bits[tid] = 0ull;
  if (tid == 0) s_n = 0;
  const int j = c * 64 + lane;
  const bool jok = j < N;
  if (NORMAL) {
    __syncthreads();
    float bj[7];
    if (jok) {
#pragma unroll
      for (int e = 0; e < 7; ++e) bj[e] = boxes[(long long)j * 7 + e];
    }
    for (int it = 0; it < 16; ++it) {
      const int rl = wave * 16 + it, i = r * 64 + rl;
      if (i >= N) break;
      const bool hit = jok && j > i && iou_normal(boxes + (long long)i * 7, bj) > thresh;
      const unsigned long long bits = __ballot(hit);
      if (lane == 0) s_bits[rl] = bits;
    }
  } else {
    if (wave == 0 && jok) s_col[lane] = pb[j];
    __syncthreads();
    const PBox B = s_col[jok ? lane : 0];
    for (int it = 0; it < 16; ++it) {
      const int rl = wave * 16 + it, i = r * 64 + rl;
      if (i >= N) break;   // wave-uniform
      const PBox A = pb[i];   // uniform address: one broadcast load
      const bool pass = jok && j > i && (thresh < 0.f || (!pbox_far(A, B) && !pbox_iou_below(A, B, thresh)));
      const unsigned long long bal = __ballot(pass);
      if (bal) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_n, __popcll(bal));
        base = __shfl(base, 0, 64);
        if (pass) s_q[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)((rl << 6) | lane);
      }
    }
    __syncthreads();
    const int n = s_n;
    for (int e = tid; e < n; e += 256) {
      const int rl = s_q[e] >> 6, cl = s_q[e] & 63;
      const PBox A = pb[r * 64 + rl], Bq = s_col[cl];
      const float sgm = box_overlap<false>(rbox_from(A), rbox_from(Bq));
      if (sgm / fmaxf(A.area + Bq.area - sgm, IOU_EPS) > thresh) atomicOr(&s_bits[rl], 1ull << cl);
    }
  }
  __syncthreads();
  if (tid < 64 && r * 64 + tid < N) maskT[(long long)c * N + r * 64 + tid] = s_bits[tid];
}

// ---- broad phase for large box sets.  The tiles above test all N^2 / 2 pairs against the far-apart rule
// (40 M pairs for 9000 proposals: 0.67 ms for the 4 frames of a training batch, nearly all of it on pairs that
// are metres apart).  Boxes are score-sorted, so tiles have no spatial coherence; a uniform BEV grid has:
// cell size >= the largest box diameter + the far-apart slack, so every pair that survives the far-apart test
// lies in the same or a neighbouring cell.  k_nms_bins (one block per frame): extent and largest radius, cell
// of every box, counting sort by cell in LDS.  k_nms_pairs (a wave per box): the boxes of the 3 x 3 cells
// around it, lanes = candidates, only pairs (i, j > i); survivors of the far-apart test get the exact
// rotated overlap and hits are OR-ed into the (zero-filled) matrix.  Same predicate, same bits as k_nms_mask.
#define NMS_GRID 64                      // cells per axis (at most)
#define NMS_BIN_THREADS 1024
struct NmsGrid {
  float x0, y0, inv;                     // cell = floor((c - origin) * inv)
  int gx, gy;
};

// pb_stride: boxes per frame in `pb` (>= N: a launch may cover the leading N boxes of longer lists); done / max_keep: a frame
// whose earlier pass over a prefix already kept max_keep boxes is skipped (see glx_nms_batch).
__global__ __launch_bounds__(NMS_BIN_THREADS) void k_nms_bins(const PBox* __restrict__ pb, int N, int pb_stride,
                                                              NmsGrid* __restrict__ grids, int* __restrict__ cell_off,
                                                              int* __restrict__ order, const int* __restrict__ done,
                                                              int max_keep) {
  const int f = blockIdx.x, tid = threadIdx.x;
  if (done && done[f] >= max_keep) return;
  pb += (long long)f * pb_stride;
  cell_off += (long long)f * (NMS_GRID * NMS_GRID + 1);
  order += (long long)f * N;
  __shared__ float s_red[5][NMS_BIN_THREADS / 64];
  __shared__ NmsGrid s_g;
  __shared__ int s_cnt[NMS_GRID * NMS_GRID + 1];
  float mnx = 3.0e38f, mny = 3.0e38f, mxx = -3.0e38f, mxy = -3.0e38f, mxr = 0.f;
  for (int i = tid; i < N; i += NMS_BIN_THREADS) {
    const PBox p = pb[i];
    if (!(p.cx == p.cx) || !(p.cy == p.cy)) continue;     // NaN boxes overlap nothing: they go to cell 0
    mnx = fminf(mnx, p.cx); mxx = fmaxf(mxx, p.cx);
    mny = fminf(mny, p.cy); mxy = fmaxf(mxy, p.cy);
    if (p.rad == p.rad) mxr = fmaxf(mxr, p.rad);
  }
  for (int o = 32; o > 0; o >>= 1) {
    mnx = fminf(mnx, __shfl_xor(mnx, o, 64)); mny = fminf(mny, __shfl_xor(mny, o, 64));
    mxx = fmaxf(mxx, __shfl_xor(mxx, o, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, o, 64));
    mxr = fmaxf(mxr, __shfl_xor(mxr, o, 64));
  }
  if ((tid & 63) == 0) {
    const int w = tid >> 6;
    s_red[0][w] = mnx; s_red[1][w] = mny; s_red[2][w] = mxx; s_red[3][w] = mxy; s_red[4][w] = mxr;
  }
  for (int c = tid; c <= NMS_GRID * NMS_GRID; c += NMS_BIN_THREADS) s_cnt[c] = 0;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < NMS_BIN_THREADS / 64; ++w) {
      s_red[0][0] = fminf(s_red[0][0], s_red[0][w]); s_red[1][0] = fminf(s_red[1][0], s_red[1][w]);
      s_red[2][0] = fmaxf(s_red[2][0], s_red[2][w]); s_red[3][0] = fmaxf(s_red[3][0], s_red[3][w]);
      s_red[4][0] = fmaxf(s_red[4][0], s_red[4][w]);
    }
    NmsGrid g;
    g.x0 = s_red[0][0]; g.y0 = s_red[1][0];
    const float ex = fmaxf(s_red[2][0] - g.x0, 0.f), ey = fmaxf(s_red[3][0] - g.y0, 0.f);
    // a pair passes the far-apart test only if its centres are within 2 * max radius + 0.05: one cell apart
    float cell = 2.f * s_red[4][0] + 0.06f;
    cell = fmaxf(cell, fmaxf(ex, ey) / (float)NMS_GRID * 1.0001f + 1e-6f);
    if (!(cell < 3.0e38f)) cell = 3.0e38f;                 // infinite radius: one cell
    g.inv = 1.f / cell;
    g.gx = min(NMS_GRID, (int)(ex * g.inv) + 1);
    g.gy = min(NMS_GRID, (int)(ey * g.inv) + 1);
    s_g = g;
    grids[f] = g;
  }
  __syncthreads();
  const NmsGrid g = s_g;
  auto cell_of = [&](const PBox& p) {
    if (!(p.cx == p.cx) || !(p.cy == p.cy)) return 0;
    const int ix = min(g.gx - 1, max(0, (int)((p.cx - g.x0) * g.inv)));
    const int iy = min(g.gy - 1, max(0, (int)((p.cy - g.y0) * g.inv)));
    return iy * g.gx + ix;
  };
  for (int i = tid; i < N; i += NMS_BIN_THREADS) atomicAdd(&s_cnt[cell_of(pb[i])], 1);
  __syncthreads();
  {                                                        // exclusive scan of the <= 4096 counts, 4 cells per thread
    const int nc = g.gx * g.gy, c0 = tid * 4;
    int v[4], sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = c0 + q < nc ? s_cnt[c0 + q] : 0; sum += v[q]; }
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if ((tid & 63) >= o) inc += t; }
    __shared__ int s_wave[NMS_BIN_THREADS / 64];
    if ((tid & 63) == 63) s_wave[tid >> 6] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += s_wave[w];
    int run = base + inc - sum;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (c0 + q < nc) { s_cnt[c0 + q] = run; cell_off[c0 + q] = run; }
      run += v[q];
    }
    if (tid == NMS_BIN_THREADS - 1) cell_off[nc] = N;
  }
  __syncthreads();
  for (int i = tid; i < N; i += NMS_BIN_THREADS) order[atomicAdd(&s_cnt[cell_of(pb[i])], 1)] = i;
}

__global__ __launch_bounds__(256) void k_nms_pairs(const PBox* __restrict__ pb, int N, int pb_stride, float thresh,
                                                   int col_blocks, const NmsGrid* __restrict__ grids,
                                                   const int* __restrict__ cell_off, const int* __restrict__ order,
                                                   unsigned long long* __restrict__ maskT, const int* __restrict__ done,
                                                   int max_keep) {
  const int f = blockIdx.y, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  if (done && done[f] >= max_keep) return;
  pb += (long long)f * pb_stride;
  cell_off += (long long)f * (NMS_GRID * NMS_GRID + 1);
  order += (long long)f * N;
  maskT += (long long)f * N * col_blocks;
  const NmsGrid g = grids[f];
  const PBox A = pb[i];
  if (!(A.cx == A.cx) || !(A.cy == A.cy)) return;
  const int ix = min(g.gx - 1, max(0, (int)((A.cx - g.x0) * g.inv)));
  const int iy = min(g.gy - 1, max(0, (int)((A.cy - g.y0) * g.inv)));
  // survivors of the far-apart test are queued per wave and evaluated 64 at a time, a lane per pair: the rotated
  // overlap is ~600 branchy flops, run on the few passing lanes of a candidate batch it would idle the others
  __shared__ int s_q[4][128];
  int* q = s_q[threadIdx.x >> 6];
  int qn = 0;
  auto flush = [&](int n) {                                // evaluate q[0..n), n <= 64
    if (lane < n) {
      const int j = q[lane];
      const PBox B = pb[j];
      const float sgm = box_overlap<false>(rbox_from(A), rbox_from(B));
      if (sgm / fmaxf(A.area + B.area - sgm, IOU_EPS) > thresh)
        atomicOr(&maskT[(long long)(j >> 6) * N + i], 1ull << (j & 63));
    }
  };
  // the candidates: up to three runs of the cell-sorted list (the cells of a grid row are contiguous), walked as ONE
  // sequence so that the indices of batch t + 2 and the boxes of batch t + 1 are in flight while batch t is tested
  // (a batch is two dependent gathers: with one batch at a time the kernel was their latency, 36 000 waves x ~12 batches)
  int seg0[3], len[3];
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int cy = iy + dy;
    int e0 = 0, e1 = 0;
    if (cy >= 0 && cy < g.gy) {
      e0 = cell_off[cy * g.gx + max(ix - 1, 0)];
      e1 = cell_off[cy * g.gx + min(ix + 1, g.gx - 1) + 1];
    }
    seg0[dy + 1] = e0;
    len[dy + 1] = e1 - e0;
  }
  const int T = len[0] + len[1] + len[2];
  auto cand = [&](int t) {                                 // index of candidate t of the sequence, -1 past its end
    if (t >= T) return -1;
    int e = seg0[0] + t;
    if (t >= len[0]) e = seg0[1] + (t - len[0]);
    if (t >= len[0] + len[1]) e = seg0[2] + (t - len[0] - len[1]);
    return order[e];
  };
  int jn = cand(lane);
  PBox Bn = pb[jn > i ? jn : i];
  int jnn = cand(64 + lane);
  for (int tb = 0; tb < T; tb += 64) {                     // wave-uniform trip count
    const int j = jn;
    const PBox Bc = Bn;
    jn = jnn;
    Bn = pb[jn > i ? jn : i];                              // next batch's boxes (own box where there is no candidate)
    jnn = cand(tb + 128 + lane);
    const bool pass = j > i && !pbox_far(A, Bc) && !pbox_iou_below(A, Bc, thresh);
    const unsigned long long bal = __ballot(pass);
    if (pass) q[qn + __popcll(bal & ((1ull << lane) - 1ull))] = j;
    qn += __popcll(bal);
    if (qn >= 64) {
      flush(64);
      const int moved = lane + 64 < qn ? q[lane + 64] : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (lane + 64 < qn) q[lane] = moved;
      qn -= 64;
    }
  }
  flush(qn);
}

static size_t nms_grid_bytes(int N) {
  return glx_align(sizeof(NmsGrid)) + glx_align((size_t)(NMS_GRID * NMS_GRID + 1) * sizeof(int)) +
         glx_align((size_t)(N > 0 ? N : 1) * sizeof(int));
}
#define NMS_BROAD_MIN 1024               // below this the tiles are few and the dense matrix is cheaper
#define NMS_PREFIX 2048                  // leading boxes of the first pass of the early-terminating NMS (glx_nms_batch)

// Greedy sweep of the matrix, 64 boxes per step (host loop of iou3d_nms.cpp:119-132), one block of
// 256 threads.  The removed-word of column block b is evaluated ON DEMAND: OR over the boxes kept
// so far of maskT[b][box] -- all threads gather from one contiguous column block (L2-friendly) and
// combine with a wave reduction + LDS; then wave 0 resolves the diagonal tile and appends the newly
// kept boxes.  Same greedy order as the reference, so the keep list is identical.
#define SWEEP_THREADS 1024
__global__ __launch_bounds__(SWEEP_THREADS) void k_nms_sweep(
    const unsigned long long* __restrict__ maskT, int N, int col_blocks,
    long long* __restrict__ keep, int keep_stride, int* __restrict__ num_out, int max_keep, int skip_if_done) {
  maskT += (long long)blockIdx.x * N * col_blocks;               // blockIdx.x = frame
  keep += (long long)blockIdx.x * keep_stride;
  num_out += blockIdx.x;
  if (skip_if_done && *num_out >= max_keep) return;              // the pass over the prefix has the whole answer already
  extern __shared__ int s_keep[];   // kept boxes so far (the gather list of every later step)
  __shared__ unsigned long long s_part[SWEEP_THREADS / 64];
  __shared__ int s_num;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_num = 0;
  // diagonal word of block 0, one step ahead from here on (its load latency hides behind the gather)
  unsigned long long diag = (wave == 0 && lane < min(64, N)) ? maskT[lane] : 0ull;
  __syncthreads();
  for (int b = 0; b < col_blocks; ++b) {
    const int base = b * 64;
    const int nb = min(64, N - base);
    const int num = s_num;
    if (num >= max_keep) break;      // uniform: the caller only wants the first max_keep survivors
    // removed bits of this block from every box kept so far (all of them precede `base`):
    // independent gathers from one contiguous column block, up to 4 in flight per thread
    const unsigned long long* col = maskT + (long long)b * N;
    unsigned long long acc = 0ull;
    int e = tid;
    for (; e + 3 * SWEEP_THREADS < num; e += 4 * SWEEP_THREADS) {
      const unsigned long long v0 = col[s_keep[e]], v1 = col[s_keep[e + SWEEP_THREADS]],
                               v2 = col[s_keep[e + 2 * SWEEP_THREADS]],
                               v3 = col[s_keep[e + 3 * SWEEP_THREADS]];
      acc |= (v0 | v1) | (v2 | v3);
    }
    for (; e < num; e += SWEEP_THREADS) acc |= col[s_keep[e]];
    unsigned long long diag_next = 0ull;
    if (wave == 0 && b + 1 < col_blocks && base + 64 + lane < N)
      diag_next = maskT[(long long)(b + 1) * N + base + 64 + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc |= __shfl_xor(acc, o, 64);
    if (lane == 0) s_part[wave] = acc;
    __syncthreads();
    if (wave == 0) {
      unsigned long long cur = (lane < SWEEP_THREADS / 64) ? s_part[lane] : 0ull;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) cur |= __shfl_xor(cur, o, 64);
      // greedy chain inside the 64-box word on the scalar unit: lane jj's diagonal word is read
      // with v_readlane (uniform index), the removed set lives in an SGPR pair
      const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
      unsigned clo = __builtin_amdgcn_readfirstlane((unsigned)cur);
      unsigned chi = __builtin_amdgcn_readfirstlane((unsigned)(cur >> 32));
      unsigned klo = 0u, khi = 0u;
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) {
        const unsigned rl = __builtin_amdgcn_readlane(dlo, jj), rh = __builtin_amdgcn_readlane(dhi, jj);
        const bool fr = !((clo >> jj) & 1u);
        klo |= fr ? (1u << jj) : 0u;
        clo |= fr ? rl : 0u;
        chi |= fr ? rh : 0u;
      }
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) {
        const unsigned rh = __builtin_amdgcn_readlane(dhi, 32 + jj);   // bits above 32+jj only
        const bool fr = !((chi >> jj) & 1u);
        khi |= fr ? (1u << jj) : 0u;
        chi |= fr ? rh : 0u;
      }
      unsigned long long kept = ((unsigned long long)khi << 32) | klo;
      if (nb < 64) kept &= (1ull << nb) - 1ull;
      if ((kept >> lane) & 1ull) {
        const int pos = num + __popcll(kept & ((1ull << lane) - 1ull));
        s_keep[pos] = base + lane;
        keep[pos] = base + lane;
      }
      if (lane == 0) s_num = num + __popcll(kept);
      diag = diag_next;
    }
    __syncthreads();
  }
  if (tid == 0) *num_out = s_num;
}

// zero the suppression matrices of the frames that still need the full pass
__global__ void k_nms_zero_unless_done(unsigned long long* __restrict__ mask, long long words_per_frame,
                                       const int* __restrict__ done, int max_keep) {
  if (done[blockIdx.y] >= max_keep) return;
  unsigned long long* m = mask + (long long)blockIdx.y * words_per_frame;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < words_per_frame; e += (long long)gridDim.x * blockDim.x)
    m[e] = 0ull;
}

static size_t nms_mask_bytes(int N) {
  size_t cb = (size_t)((N + 63) / 64);
  return glx_align((size_t)(N > 0 ? N : 1) * cb * 8);
}
extern "C" size_t glx_nms_workspace_bytes(int N) {   // suppression matrix + prepared boxes + broad-phase grid
  return nms_mask_bytes(N) + glx_align((size_t)(N > 0 ? N : 1) * sizeof(PBox)) + nms_grid_bytes(N) + 256;
}

// Batched form: `frames` independent box lists of N boxes each (boxes (F,N,7), keep (F,N),
// num_out (F)), every kernel launched once with the frame on a grid axis -- the single-block sweeps
// of the frames run side by side.  max_keep > 0 stops a frame's sweep once that many boxes are
// kept (the caller truncates to NMS_POST_MAXSIZE anyway; the kept prefix is unchanged).
extern "C" int glx_nms_batch(const float* boxes_sorted, int frames, int N, float thresh, int normal,
                             int max_keep, int64_t* keep, int32_t* num_out, void* workspace,
                             size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(keep && num_out, "glx_nms: null output");
  GLX_REQUIRE(frames >= 1 && frames <= 65535, "glx_nms: frames must be 1..65535");
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) {
    GlxFillJob job{num_out, sizeof(int) * (size_t)frames, 0};
    return glx_fill_multi(&job, 1, st);
  }
  GLX_REQUIRE(boxes_sorted, "glx_nms: null boxes");
  int col_blocks = (N + 63) / 64;
  const size_t mask_b = nms_mask_bytes(N), pb_b = glx_align((size_t)N * sizeof(PBox)), grid_b = nms_grid_bytes(N);
  size_t need = (size_t)frames * (mask_b + pb_b + grid_b);
  if (!workspace || workspace_bytes < need) {
    glx_set_error("glx_nms: workspace %zu < %zu bytes", workspace_bytes, need);
    return GLX_EWORKSPACE;
  }
  // frame f: mask words at f * N * col_blocks, prepared boxes at f * N (dense, as the kernels index)
  unsigned long long* mask = (unsigned long long*)workspace;
  PBox* pb = (PBox*)((char*)workspace + (size_t)frames * mask_b);
  // words of the lower triangle (row block > column block) are never written and never read:
  // the sweep ORs column block b only over boxes kept BEFORE block b, plus the diagonal tile.
  int ntiles = col_blocks * (col_blocks + 1) / 2;
  if (normal) {
    hipLaunchKernelGGL((k_nms_mask<true>), dim3(ntiles, frames), dim3(256), 0, st, boxes_sorted,
                       (const PBox*)nullptr, N, thresh, col_blocks, mask);
  } else {
    hipLaunchKernelGGL(k_nms_prepare, dim3(glx_divup(N, 256), frames), dim3(256), 0, st, boxes_sorted,
                       N, pb);
    if (N >= NMS_BROAD_MIN && thresh >= 0.f) {     // thresh < 0 suppresses disjoint boxes too: every pair counts
      char* gbase = (char*)workspace + (size_t)frames * (mask_b + pb_b);
      NmsGrid* grids = (NmsGrid*)gbase;
      int* cell_off = (int*)(gbase + glx_align((size_t)frames * sizeof(NmsGrid)));
      int* order = cell_off + (size_t)frames * (NMS_GRID * NMS_GRID + 1);
      // Early termination (max_keep > 0, long lists).  The greedy decision for box i depends on boxes before i only, so
      // the keep list of the leading N1 boxes is a PREFIX of the full list; when it already holds max_keep boxes -- the
      // caller wants no more: keep[:NMS_POST_MAXSIZE] of class_agnostic_nms -- the full pass has nothing to add.  Score-
      // sorted proposal lists end their first 512 survivors early (the bulk of 9000 candidates are background boxes that
      // rarely suppress one another at 0.8), so: a pass over the first NMS_PREFIX boxes, then the full pass whose every
      // kernel returns at once for a frame that is done (device-side flag: no read-back, capturable).  Same keep list,
      // bit for bit, either way; 40 M candidate pairs per frame shrink to 2 M when the prefix suffices.
      const bool two_stage = max_keep > 0 && N >= 2 * NMS_PREFIX && max_keep <= NMS_PREFIX;
      static bool sweep_attr = false;
      if (!sweep_attr) {
        GLX_HIP(hipFuncSetAttribute((const void*)k_nms_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        sweep_attr = true;
      }
      GLX_REQUIRE((size_t)N * 4 <= 150 * 1024, "glx_nms: N = %d exceeds the %d boxes the sweep keeps in LDS", N,
                  150 * 1024 / 4);
      if (two_stage) {
        const int N1 = NMS_PREFIX, cb1 = (N1 + 63) / 64;
        GlxFillJob zj{mask, (size_t)frames * (size_t)N1 * cb1 * 8, 0};
        int rc = glx_fill_multi(&zj, 1, st);
        if (rc != GLX_OK) return rc;
        hipLaunchKernelGGL(k_nms_bins, dim3(frames), dim3(NMS_BIN_THREADS), 0, st, (const PBox*)pb, N1, N, grids, cell_off,
                           order, (const int*)nullptr, 0);
        hipLaunchKernelGGL(k_nms_pairs, dim3(glx_divup(N1, 4), frames), dim3(256), 0, st, (const PBox*)pb, N1, N, thresh,
                           cb1, (const NmsGrid*)grids, (const int*)cell_off, (const int*)order, mask, (const int*)nullptr, 0);
        hipLaunchKernelGGL(k_nms_sweep, dim3(frames), dim3(SWEEP_THREADS), (size_t)N1 * 4, st,
                           (const unsigned long long*)mask, N1, cb1, (long long*)keep, N, num_out, max_keep, 0);
        hipLaunchKernelGGL(k_nms_zero_unless_done, dim3(256, frames), dim3(256), 0, st, mask, (long long)N * col_blocks,
                           (const int*)num_out, max_keep);
      } else {
        GlxFillJob zj{mask, (size_t)frames * (size_t)N * col_blocks * 8, 0};
        int rc = glx_fill_multi(&zj, 1, st);
        if (rc != GLX_OK) return rc;
      }
      const int* done = two_stage ? (const int*)num_out : (const int*)nullptr;
      hipLaunchKernelGGL(k_nms_bins, dim3(frames), dim3(NMS_BIN_THREADS), 0, st, (const PBox*)pb, N, N, grids, cell_off,
                         order, done, max_keep);
      hipLaunchKernelGGL(k_nms_pairs, dim3(glx_divup(N, 4), frames), dim3(256), 0, st, (const PBox*)pb, N, N, thresh,
                         col_blocks, (const NmsGrid*)grids, (const int*)cell_off, (const int*)order, mask, done, max_keep);
      hipLaunchKernelGGL(k_nms_sweep, dim3(frames), dim3(SWEEP_THREADS), (size_t)N * 4, st,
                         (const unsigned long long*)mask, N, col_blocks, (long long*)keep, N, num_out,
                         max_keep > 0 ? max_keep : 0x7fffffff, two_stage ? 1 : 0);
      GLX_LAUNCH_CHECK();
      return GLX_OK;
    } else {
      hipLaunchKernelGGL((k_nms_mask<false>), dim3(ntiles, frames), dim3(256), 0, st, boxes_sorted,
                         (const PBox*)pb, N, thresh, col_blocks, mask);
    }
  }
  GLX_REQUIRE((size_t)N * 4 <= 150 * 1024, "glx_nms: N = %d exceeds the %d boxes the sweep keeps in LDS", N,
              150 * 1024 / 4);
  static bool sweep_attr = false;
  if (!sweep_attr) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_nms_sweep, hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024));
    sweep_attr = true;
  }
  hipLaunchKernelGGL(k_nms_sweep, dim3(frames), dim3(SWEEP_THREADS), (size_t)N * 4, st,
                     (const unsigned long long*)mask, N, col_blocks, (long long*)keep, N, num_out,
                     max_keep > 0 ? max_keep : 0x7fffffff, 0);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_nms(const float* boxes_sorted, int N, float thresh, int normal, int64_t* keep,
                       int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream) {
  return glx_nms_batch(boxes_sorted, 1, N, thresh, normal, 0, keep, num_out, workspace,
                       workspace_bytes, stream);
}

// ------------------------------------------------------------------ GLENet variance-voting NMS
// nms_func, pcdet/ops/iou3d_nms/iou3d_nms_utils.py:227-273, one block, fp32 like numpy.
// ious (N,N) precomputed on the ORIGINAL boxes (:235); boxes/scores are updated in place.
#define VOTE_THREADS 1024

// Three barriers per round: (1) arg-max of the undone scores -- wave shuffles, 16 wave results in LDS,
// every thread folds them; (2) the 14 weighted sums -- wave shuffles, 16 x 14 partials in LDS, lanes
// 0..6 of wave 0 fold them and write the voted box; (3) end of round (scores / box visible).  The
// reference loop runs once per box when score_threshold is 0 (suppressed boxes keep score 0 >= 0 and
// are visited too), so the round cost is the whole cost: 4096 boxes took 70 ms with tree reductions
// (~180 barriers per round).  Sum order: deterministic, differs from numpy's pairwise sum like any
// fp32 reduction (tolerance 1e-4 on the voted boxes, tests/test_ops_gpu.py).

// State in registers: a thread owns boxes t, t+1024, ... -- their scores (only the owner ever changes a
// score), coordinates and variances; the voted box reaches its owner through LDS.  The only memory
// access on a round's critical path is one coalesced row of the TRANSPOSED IoU matrix.
// THREADS: 1024 for up to 4096 boxes, 256 for up to 1024 (4 waves: cheaper barriers and folds -- the
// two-stage models vote <= 100 RoIs per frame).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_nms_vote(
    float* __restrict__ boxes, float* __restrict__ scores, const float* __restrict__ variance,
    int var_stride, const float* __restrict__ iousT, int N, float iou_thr, float score_thr,
    float* __restrict__ scratch, const int* __restrict__ counts) {
  // blockIdx.x = frame; N = row pitch of every per-frame array; counts[frame] (<= N) live boxes when given
  const int P = N;
  {
    const long long f = blockIdx.x;
    boxes += f * P * 7; scores += f * P; iousT += f * P * P;
    if (variance) variance += f * P * var_stride;
    if (scratch) scratch += f * P * 8;
    if (counts) N = min(P, counts[f]);
  }
  constexpr int VOTE_WAVES = THREADS / 64;
  __shared__ float s_best[VOTE_WAVES], s_head[VOTE_WAVES];
  bool tail = false;
  __shared__ int s_bi[VOTE_WAVES];
  __shared__ float s_part[VOTE_WAVES][16];
  __shared__ float s_new[7];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float PI_3_2 = (float)(3.14159265358979323846 * 3 / 2);
  const float PI_2x = (float)(3.14159265358979323846 * 2);
  const float PI_4 = (float)(3.14159265358979323846 / 4);
  constexpr int PER = 4;   // N <= 4 * THREADS
  bool undone[PER];
  float sc[PER], bxr[PER][7], vr[PER][7];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = t + u * THREADS;
    sc[u] = i < N ? scores[i] : 0.f;
    undone[u] = i < N && sc[u] >= score_thr;
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      bxr[u][c] = i < N ? boxes[(long long)i * 7 + c] : 0.f;
      vr[u][c] = (variance && i < N) ? variance[(long long)i * var_stride + c] : 1.f;
    }
  }
  while (true) {
    // argmax of scores over undone boxes, first index on ties (np.argmax); its heading rides along
    float best = -INFINITY, head = 0.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = t + u * THREADS;
      if (undone[u] && (bi == 0x7fffffff || sc[u] > best || (sc[u] == best && i < bi))) {
        best = sc[u]; bi = i; head = bxr[u][6];
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float ob = __shfl_xor(best, d, 64), oh = __shfl_xor(head, d, 64);
      const int oi = __shfl_xor(bi, d, 64);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; head = oh; }
    }
    if (lane == 0) { s_best[wave] = best; s_bi[wave] = bi; s_head[wave] = head; }
    __syncthreads();                                                     // (1)
    best = s_best[0]; bi = s_bi[0]; head = s_head[0];
#pragma unroll
    for (int w = 1; w < VOTE_WAVES; ++w) {
      const float ob = s_best[w];
      const int oi = s_bi[w];
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; head = s_head[w]; }
    }
    const int idx = bi;
    if (idx == 0x7fffffff) break;   // undone_mask.sum() == 0
    if (best == 0.f && score_thr <= 0.f) {
      // Every undone box has score 0 (suppressed boxes stay "undone" under a threshold of 0): from here
      // on the reference visits them in index order, each round votes box j from the still-undone
      // boxes i >= j -- all untouched originals -- and changes no score.  The rounds are independent:
      // they run in parallel in k_nms_vote_tail instead of one after the other (4096 boxes: ~3900 rounds).
      bool odd = false;
#pragma unroll
      for (int u = 0; u < PER; ++u) odd |= undone[u] && sc[u] != 0.f;
      if (!__syncthreads_or(odd)) { tail = true; break; }
    }

    float iouv[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = t + u * THREADS;
      iouv[u] = undone[u] ? iousT[(long long)idx * P + i] : 0.f;        // == ious[i][idx]
    }
    if (variance) {
      const float top_h = head;
      float acc[16];                 // sums of pi (7) and of pi * box (7), padded to 16
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = 0.f;
      bool cand = false;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        if (!undone[u]) continue;
        const float iou = iouv[u];
        if (!(iou > iou_thr)) continue;
        cand = true;
        float h = bxr[u][6];
        if (fabsf(h - top_h) >= PI_3_2) h = top_h > 0.f ? h + PI_2x : h - PI_2x;
        float d = 1.f - iou;
        float p = expf(-1.f * (d * d) / 0.05f);
        bool far = fabsf(h - top_h) >= PI_4;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          float w = p / vr[u][c];
          if (c == 6 && far) w = 0.f;
          acc[c] += w;
          acc[7 + c] += w * (c == 6 ? h : bxr[u][c]);
        }
      }
      // Wave sums of the 16 values as a reduce-scatter butterfly: at distance 32 / 16 / 8 / 4 a lane
      // keeps one half of its values and trades the other (8 + 4 + 2 + 1 shuffles), then two plain
      // steps -- 17 cross-lane operations instead of 6 per value (the LDS crossbar, shared by the 16
      // waves, was the round's bottleneck).  A wave without a candidate (most of them) skips it.
      if (__ballot(cand) == 0ull) {
        if ((lane & 3) == 0) s_part[wave][lane >> 2] = 0.f;
      } else {
        float r8[8], r4[4], r2[2], r1;
        {
          const bool up = lane & 32;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float keep = up ? acc[8 + j] : acc[j], give = up ? acc[j] : acc[8 + j];
            r8[j] = keep + __shfl_xor(give, 32, 64);
          }
        }
        {
          const bool up = lane & 16;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float keep = up ? r8[4 + j] : r8[j], give = up ? r8[j] : r8[4 + j];
            r4[j] = keep + __shfl_xor(give, 16, 64);
          }
        }
        {
          const bool up = lane & 8;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const float keep = up ? r4[2 + j] : r4[j], give = up ? r4[j] : r4[2 + j];
            r2[j] = keep + __shfl_xor(give, 8, 64);
          }
        }
        {
          const bool up = lane & 4;
          const float keep = up ? r2[1] : r2[0], give = up ? r2[0] : r2[1];
          r1 = keep + __shfl_xor(give, 4, 64);
        }
        r1 += __shfl_xor(r1, 2, 64);
        r1 += __shfl_xor(r1, 1, 64);
        // lane bits 32/16/8/4 select value 8/4/2/1 of the index
        if ((lane & 3) == 0) s_part[wave][((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1)] = r1;
      }
      __syncthreads();                                                   // (2)
      if (t < 7) {
        float tp = 0.f, tb = 0.f;
#pragma unroll
        for (int w = 0; w < VOTE_WAVES; ++w) { tp += s_part[w][t]; tb += s_part[w][7 + t]; }
        const float nb = tb / tp;                   // == sum((pi / pi.sum) * box)
        s_new[t] = nb;
        boxes[(long long)idx * 7 + t] = nb;
      }
    }
    // undone[idx] = False; scores[undone] *= (iou < thr); undone[scores < score_thr] = False
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = t + u * THREADS;
      if (i == idx) undone[u] = false;
      if (undone[u]) {
        sc[u] = sc[u] * ((iouv[u] < iou_thr) ? 1.f : 0.f);
        if (sc[u] < score_thr) undone[u] = false;
      }
    }
    __syncthreads();                                                     // (3)
    if (variance && (idx % THREADS) == t) {                   // the owner takes the voted box
#pragma unroll
      for (int u = 0; u < PER; ++u)
        if ((idx / THREADS) == u) {
#pragma unroll
          for (int c = 0; c < 7; ++c) bxr[u][c] = s_new[c];
        }
    }
  }
  if (variance) {                       // hand the independent rounds to k_nms_vote_tail
    int* flags = (int*)(scratch + (long long)P * 7);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = t + u * THREADS;
      if (i < N) flags[i] = (tail && undone[u]) ? 1 : 0;
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = t + u * THREADS;
    if (i < N) scores[i] = sc[u];
  }
}

// The independent rounds of the zero-score tail (see k_nms_vote): a wave per flagged box j votes it from
// the flagged boxes i >= j, all of them untouched originals; results go to scratch and are copied
// over the boxes by k_nms_vote_copy once every wave has read what it needs.
__global__ __launch_bounds__(256) void k_nms_vote_tail(const float* __restrict__ boxes,
                                                       const float* __restrict__ variance, int var_stride,
                                                       const float* __restrict__ iousT, int N, float iou_thr,
                                                       float* __restrict__ scratch, const int* __restrict__ counts) {
  const int P = N;
  {
    const long long f = blockIdx.y;
    boxes += f * P * 7; iousT += f * P * P; variance += f * P * var_stride; scratch += f * P * 8;
    if (counts) N = min(P, counts[f]);
  }
  const int* flags = (const int*)(scratch + (long long)P * 7);
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= N || !flags[j]) return;
  const float PI_3_2 = (float)(3.14159265358979323846 * 3 / 2);
  const float PI_2x = (float)(3.14159265358979323846 * 2);
  const float PI_4 = (float)(3.14159265358979323846 / 4);
  const float top_h = boxes[(long long)j * 7 + 6];
  float acc[14];
#pragma unroll
  for (int c = 0; c < 14; ++c) acc[c] = 0.f;
  for (int i = (j & ~63) + lane; i < N; i += 64) {
    if (i < j || !flags[i]) continue;
    const float iou = iousT[(long long)j * P + i];
    if (!(iou > iou_thr)) continue;
    float bx[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) bx[c] = boxes[(long long)i * 7 + c];
    if (fabsf(bx[6] - top_h) >= PI_3_2) bx[6] = top_h > 0.f ? bx[6] + PI_2x : bx[6] - PI_2x;
    const float d = 1.f - iou;
    const float p = expf(-1.f * (d * d) / 0.05f);
    const bool far = fabsf(bx[6] - top_h) >= PI_4;
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      float w = p / variance[(long long)i * var_stride + c];
      if (c == 6 && far) w = 0.f;
      acc[c] += w;
      acc[7 + c] += w * bx[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 14; ++c) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc[c] += __shfl_xor(acc[c], d, 64);
  }
  if (lane < 7) {
    float tp = acc[0], tb = acc[7];
#pragma unroll
    for (int c = 1; c < 7; ++c)
      if (lane == c) { tp = acc[c]; tb = acc[7 + c]; }
    scratch[(long long)j * 7 + lane] = tb / tp;
  }
}

__global__ void k_nms_vote_copy(float* __restrict__ boxes, const float* __restrict__ scratch, int N,
                                const int* __restrict__ counts) {
  const int P = N;
  boxes += (long long)blockIdx.y * P * 7; scratch += (long long)blockIdx.y * P * 8;
  if (counts) N = min(P, counts[blockIdx.y]);
  const int* flags = (const int*)(scratch + (long long)P * 7);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < N * 7 && flags[e / 7]) boxes[e] = scratch[e];
}

extern "C" int glx_nms_vote_batch(float* boxes, float* scores, const float* variance, int var_stride,
                                  const float* ious_t, int frames, int N, const int32_t* counts, float iou_thr,
                                  float score_thr, float* scratch, void* stream) {
  if (N == 0 || frames == 0) return GLX_OK;
  GLX_REQUIRE(boxes && scores && ious_t, "glx_nms_vote: null pointer");
  GLX_REQUIRE(N <= 4 * VOTE_THREADS, "glx_nms_vote: N=%d exceeds %d (NMS_PRE_MAXSIZE)", N,
              4 * VOTE_THREADS);
  GLX_REQUIRE(frames > 0 && frames <= 65535, "glx_nms_vote: %d frames", frames);
  GLX_REQUIRE(!variance || var_stride >= 7, "glx_nms_vote: variance needs >= 7 columns");
  GLX_REQUIRE(!variance || scratch, "glx_nms_vote: voting needs the N*8-float scratch buffer");
  if (N <= 1024)
    hipLaunchKernelGGL(k_nms_vote<256>, dim3(frames), dim3(256), 0, (hipStream_t)stream, boxes, scores, variance,
                       var_stride, ious_t, N, iou_thr, score_thr, scratch, counts);
  else
    hipLaunchKernelGGL(k_nms_vote<VOTE_THREADS>, dim3(frames), dim3(VOTE_THREADS), 0, (hipStream_t)stream, boxes,
                       scores, variance, var_stride, ious_t, N, iou_thr, score_thr, scratch, counts);
  if (variance) {
    hipLaunchKernelGGL(k_nms_vote_tail, dim3(glx_divup(N, 4), frames), dim3(256), 0, (hipStream_t)stream,
                       (const float*)boxes, variance, var_stride, ious_t, N, iou_thr, scratch, counts);
    hipLaunchKernelGGL(k_nms_vote_copy, dim3(glx_divup(N * 7, 256), frames), dim3(256), 0, (hipStream_t)stream, boxes,
                       (const float*)scratch, N, counts);
  }
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_nms_vote(float* boxes, float* scores, const float* variance, int var_stride,
                            const float* ious_t, int N, float iou_thr, float score_thr,
                            float* scratch, void* stream) {
  return glx_nms_vote_batch(boxes, scores, variance, var_stride, ious_t, 1, N, nullptr, iou_thr, score_thr, scratch,
                            stream);
}

// Stand-in for a trained first stage (bench.py, tests): the first G proposal slots of every frame take the frame's ground
// truth + a fixed offset wherever a ground-truth row is live (class > 0) -- eleven elementwise launches as tensor
// statements (compare, add, two where, casts, copies), one here.
__global__ void k_seed_rois(float* __restrict__ rois, int64_t* __restrict__ labels, const float* __restrict__ gt,
                            const float* __restrict__ offset, int B, int R, int ld, int G, int gld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * G) return;
  const int b = i / G, g = i - b * G;
  const float* q = gt + ((long long)b * G + g) * gld;
  if (!(q[7] > 0.f)) return;
  float* r = rois + ((long long)b * R + g) * ld;
  for (int k = 0; k < 7; ++k) r[k] = q[k] + offset[k];
  labels[(long long)b * R + g] = (int64_t)q[7];
}

extern "C" int glx_seed_rois(float* rois, int64_t* roi_labels, const float* gt_boxes, const float* offset7, int B, int R,
                             int roi_ld, int G, int gt_ld, void* stream) {
  if (B <= 0 || G <= 0) return GLX_OK;
  GLX_REQUIRE(rois && roi_labels && gt_boxes && offset7, "glx_seed_rois: null pointer");
  GLX_REQUIRE(G <= R && roi_ld >= 7 && gt_ld >= 8, "glx_seed_rois: %d ground-truth rows for %d RoI slots", G, R);
  hipLaunchKernelGGL(k_seed_rois, dim3(glx_divup(B * G, 256)), dim3(256), 0, (hipStream_t)stream, rois, roi_labels, gt_boxes,
                     offset7, B, R, roi_ld, G, gt_ld);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ inference post-processing
// The two ends of Detector3DTemplate.post_processing + class_agnostic_nms (pcdet/models/detectors/detector3d_template.py:
// 179-317, pcdet/models/model_utils/model_nms_utils.py:6-62) around the top-k and the voting NMS, one launch each.

// k_det_candidates: a thread per (frame, top-k slot).  order[f][j] = index of the j-th best score of the frame among
// the boxes that passed SCORE_THRESH (slots >= counts[f] are padding): gathers the box with its heading wrapped to
// [-pi, pi) (new_nms_gpu, iou3d_nms_utils.py:212-214: limit_period(offset 0.5, period 2 pi)) and exp(log-variance).
__global__ void k_det_candidates(const float* __restrict__ box_preds, const float* __restrict__ std_preds, int ld_box,
                                 int ld_std, const int64_t* __restrict__ order, const int* __restrict__ counts, int F,
                                 int R, int K, float* __restrict__ cand, float* __restrict__ var) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * K) return;
  const int f = i / K, j = i - f * K;
  const bool live = j < counts[f];
  const long long src = (long long)f * R + (live ? order[i] : 0);
  float* c = cand + (long long)i * 7;
  for (int k = 0; k < 7; ++k) c[k] = live ? box_preds[src * ld_box + k] : 0.f;
  if (live) {
    const float period = (float)(3.14159265358979323846 * 2);
    c[6] = c[6] - floorf(c[6] / period + 0.5f) * period;
  }
  if (var) {
    float* v = var + (long long)i * 7;
    for (int k = 0; k < 7; ++k) v[k] = live ? expf(std_preds[src * ld_std + k]) : 1.f;
  }
}

extern "C" int glx_det_candidates(const float* box_preds, const float* std_preds, int ld_box, int ld_std,
                                  const int64_t* order, const int32_t* counts, int F, int R, int K, float* cand,
                                  float* var, void* stream) {
  if (F <= 0 || K <= 0) return GLX_OK;
  GLX_REQUIRE(box_preds && order && counts && cand, "glx_det_candidates: null pointer");
  GLX_REQUIRE(ld_box >= 7 && (!var || (std_preds && ld_std >= 7)), "glx_det_candidates: boxes / variances need >= 7 columns");
  hipLaunchKernelGGL(k_det_candidates, dim3(glx_divup(F * K, 256)), dim3(256), 0, (hipStream_t)stream, box_preds,
                     std_preds, ld_box, ld_std, order, counts, F, R, K, cand, var);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// k_det_gather: one block per frame.  Survivors of the voting NMS are the slots whose score is still > 0
// (iou3d_nms_utils.py:219-220), in slot order = descending score; the first `P` of them (NMS_POST_MAXSIZE) whose ORIGINAL
// score also exceeds post_thr (POST_SCORE_THRESH, detector3d_template.py:295-300; scores descend along the keep list, so
// that mask is a prefix) are written with the voted box, the original score, the label of the source box and its index.
__global__ __launch_bounds__(256) void k_det_gather(const float* __restrict__ new_scores, const float* __restrict__ top,
                                                    const float* __restrict__ cand, const int64_t* __restrict__ order,
                                                    const int64_t* __restrict__ labels, const int* __restrict__ counts,
                                                    int R, int K, int P, float post_thr, int use_post,
                                                    float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                    int64_t* __restrict__ out_labels, int64_t* __restrict__ out_index,
                                                    int* __restrict__ out_num) {
  __shared__ int s_wave[4];
  __shared__ int s_base, s_cnt;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = min(K, counts[f]);
  if (tid == 0) { s_base = 0; s_cnt = 0; }
  for (int e = tid; e < P * 7; e += 256) out_boxes[(long long)f * P * 7 + e] = 0.f;
  for (int e = tid; e < P; e += 256) {
    out_scores[(long long)f * P + e] = 0.f;
    out_labels[(long long)f * P + e] = 0;
    out_index[(long long)f * P + e] = -1;
  }
  __syncthreads();
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + tid;
    const long long g = (long long)f * K + j;
    const bool alive = j < n && new_scores[g] > 0.f;
    const bool pass = alive && (!use_post || top[g] > post_thr);
    const unsigned long long bal = __ballot(alive);
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    const int slot = before + __popcll(bal & ((1ull << lane) - 1ull));
    if (pass && slot < P) {
      const long long o = (long long)f * P + slot;
      for (int k = 0; k < 7; ++k) out_boxes[o * 7 + k] = cand[g * 7 + k];
      out_scores[o] = top[g];
      out_labels[o] = labels ? labels[(long long)f * R + order[g]] : 1;
      out_index[o] = order[g];
      atomicAdd(&s_cnt, 1);
    }
    __syncthreads();
    if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (tid == 0) out_num[f] = s_cnt;      // the written slots are the prefix [0, s_cnt): scores descend along the survivors
}

extern "C" int glx_det_gather(const float* new_scores, const float* top, const float* cand, const int64_t* order,
                              const int64_t* labels, const int32_t* counts, int F, int R, int K, int P, float post_thr,
                              int use_post, float* out_boxes, float* out_scores, int64_t* out_labels,
                              int64_t* out_index, int32_t* out_num, void* stream) {
  if (F <= 0) return GLX_OK;
  GLX_REQUIRE(new_scores && top && cand && order && counts && out_boxes && out_scores && out_labels && out_index && out_num,
              "glx_det_gather: null pointer");
  GLX_REQUIRE(K > 0 && P > 0 && R > 0, "glx_det_gather: bad sizes");
  hipLaunchKernelGGL(k_det_gather, dim3(F), dim3(256), 0, (hipStream_t)stream, new_scores, top, cand, order, labels, counts,
                     R, K, P, post_thr, use_post, out_boxes, out_scores, out_labels, out_index, out_num);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ RoI targets (training)
// ProposalTargetLayer (pcdet/models/roi_heads/target_assigner/proposal_target_layer.py:13-239) for the
// whole batch in two launches and no host round trip; the reference loops over frames and classes
// in Python with an (R, G) IoU matrix, masks, nonzero() and host-side random draws for each.
//
// k_roi_match: a wave per RoI, a lane per ground truth of its frame (trailing all-zero rows trimmed as
// :98-102), 3-D IoU exactly as boxes_iou3d_gpu composes it (BEV overlap x height overlap / clamped
// union volume, iou3d_nms_utils.py:88-121), running maximum with the smallest index on ties
// (torch.max).  same_class: only ground truths of the RoI's label compete (:209-238); a RoI without
// one keeps overlap 0 and assignment 0.
#define ROI_MAX_GT 256
__global__ __launch_bounds__(256) void k_roi_match(const float* __restrict__ rois, const int64_t* __restrict__ roi_labels,
                                                   int R, int roi_ld, const float* __restrict__ gt, int G, int gt_ld,
                                                   int same_class, float* __restrict__ max_overlaps,
                                                   int* __restrict__ assignment, int* __restrict__ n_gt_out) {
  __shared__ PBox s_gt[ROI_MAX_GT];
  __shared__ float s_lo[ROI_MAX_GT], s_hi[ROI_MAX_GT], s_vol[ROI_MAX_GT];
  __shared__ int s_label[ROI_MAX_GT];
  __shared__ int s_n;
  const int b = blockIdx.y, tid = threadIdx.x;
  gt += (long long)b * G * gt_ld;
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int g = tid; g < G; g += blockDim.x) {
    const float* bx = gt + (long long)g * gt_ld;
    float sum = 0.f;
    for (int c = 0; c < gt_ld; ++c) sum += bx[c];
    if (sum != 0.f) atomicMax(&s_n, g + 1);
    PBox p;
    p.cx = bx[0]; p.cy = bx[1];
    p.hx = bx[3] / 2; p.hy = bx[4] / 2;
    p.c = f_cos(bx[6]); p.s = f_sin(bx[6]);
    p.rad = sqrtf(p.hx * p.hx + p.hy * p.hy);
    p.area = bx[3] * bx[4];
    s_gt[g] = p;
    s_lo[g] = bx[2] - bx[5] / 2;
    s_hi[g] = bx[2] + bx[5] / 2;
    s_vol[g] = bx[3] * bx[4] * bx[5];
    s_label[g] = (int)(long long)bx[gt_ld - 1];
  }
  __syncthreads();
  const int n = s_n;
  if (tid == 0 && blockIdx.x == 0) n_gt_out[b] = n;
  const int r = blockIdx.x * (blockDim.x >> 6) + (tid >> 6);      // a wave per RoI, lane = ground truth
  if (r >= R) return;
  const int lane = tid & 63;
  const float* bx = rois + ((long long)b * R + r) * roi_ld;
  PBox A;
  A.cx = bx[0]; A.cy = bx[1];
  A.hx = bx[3] / 2; A.hy = bx[4] / 2;
  A.c = f_cos(bx[6]); A.s = f_sin(bx[6]);
  A.rad = sqrtf(A.hx * A.hx + A.hy * A.hy);
  A.area = bx[3] * bx[4];
  const float lo = bx[2] - bx[5] / 2, hi = bx[2] + bx[5] / 2, vol = bx[3] * bx[4] * bx[5];
  const int label = (int)roi_labels[(long long)b * R + r];
  float best = -1.f;
  int arg = 0x7fffffff;
  for (int g = lane; g < n; g += 64) {
    if (same_class && s_label[g] != label) continue;
    float bev = 0.f;
    if (!pbox_far(A, s_gt[g])) bev = box_overlap<false>(rbox_from(A), rbox_from(s_gt[g]));
    const float h = fmaxf(mn(hi, s_hi[g]) - mx(lo, s_lo[g]), 0.f);
    const float inter = bev * h;
    const float v = inter / fmaxf(vol + s_vol[g] - inter, 1e-6f);
    if (v > best) { best = v; arg = g; }                           // ascending g within a lane: first maximum
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {                              // maximum, smallest index on ties
    const float ob = __shfl_xor(best, d, 64);
    const int oa = __shfl_xor(arg, d, 64);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
  }
  if (lane == 0) {
    max_overlaps[(long long)b * R + r] = best < 0.f ? 0.f : best;
    assignment[(long long)b * R + r] = best < 0.f ? 0 : arg;
  }
}

// k_roi_sample: one block per frame.  Lists by overlap (:128-137): foreground >= fg_thresh, easy background
// < bg_lo, hard background in [bg_lo, REG_FG_THRESH); each in ascending RoI order (= nonzero()).
// The random draws come in as uniform numbers so that the caller owns the generator:
//   key (B, R)  -- foreground RoIs are taken in ascending key order (a uniform random subset and order:
//                  the reference's permutation, :147-148);
//   pick (B, P) -- slot q draws list[floor(pick[q] * len(list))] with replacement (torch.randint, :186-207;
//                  the all-foreground case :158-162).
// Emits the sampled RoI index and its ground-truth index (-1: the frame has none -> zero row, :103).
struct RoiSampleCfg {
  int P, fg_per_image;
  float fg_thresh, bg_lo, reg_fg;
  double hard_ratio;
};

__global__ __launch_bounds__(256) void k_roi_sample(const float* __restrict__ max_overlaps, const int* __restrict__ assignment,
                                                    const int* __restrict__ n_gt, const float* __restrict__ key,
                                                    const float* __restrict__ pick, int R, RoiSampleCfg cfg,
                                                    int* __restrict__ sampled, int* __restrict__ sampled_gt) {
  extern __shared__ int s_mem[];
  int* s_list[3] = {s_mem, s_mem + R, s_mem + 2 * R};          // fg, hard, easy
  __shared__ int s_cnt[3][257];
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  max_overlaps += (long long)b * R;
  key += (long long)b * R;
  pick += (long long)b * cfg.P;
  const int per = (R + nt - 1) / nt, r0 = tid * per, r1 = min(R, r0 + per);
  // three independent predicates, as the reference's three nonzero() calls (:128-137): with CLS_FG_THRESH <
  // REG_FG_THRESH an RoI in [fg_thresh, REG_FG) is foreground AND hard background; a NaN overlap is in no list
  auto member = [&](float v, int k) {
    return k == 0 ? v >= cfg.fg_thresh : (k == 2 ? v < cfg.bg_lo : (v < cfg.reg_fg && v >= cfg.bg_lo));
  };
  int c3[3] = {0, 0, 0};
  for (int r = r0; r < r1; ++r) {
    const float v = max_overlaps[r];
    for (int k = 0; k < 3; ++k) c3[k] += member(v, k) ? 1 : 0;
  }
  for (int k = 0; k < 3; ++k) s_cnt[k][tid] = c3[k];
  __syncthreads();
  if (tid < 3) {                                               // exclusive scan over the threads' chunks
    int run = 0;
    for (int t = 0; t < nt; ++t) { const int v = s_cnt[tid][t]; s_cnt[tid][t] = run; run += v; }
    s_cnt[tid][256] = run;
  }
  __syncthreads();
  int w3[3] = {s_cnt[0][tid], s_cnt[1][tid], s_cnt[2][tid]};
  for (int r = r0; r < r1; ++r) {
    const float v = max_overlaps[r];
    for (int k = 0; k < 3; ++k)
      if (member(v, k)) s_list[k][w3[k]++] = r;
  }
  __syncthreads();
  const int nfg = s_cnt[0][256], nhard = s_cnt[1][256], neasy = s_cnt[2][256], nbg = nhard + neasy;
  const int P = cfg.P;
  int* out = sampled + (long long)b * P;
  int take = 0;
  if (nfg > 0 && nbg > 0) {
    take = min(cfg.fg_per_image, nfg);
    for (int i = tid; i < nfg; i += nt) {                      // rank of every foreground RoI by (key, index)
      const int ri = s_list[0][i];
      const float ki = key[ri];
      int rank = 0;
      for (int j = 0; j < nfg; ++j) {
        const int rj = s_list[0][j];
        const float kj = key[rj];
        rank += (kj < ki || (kj == ki && rj < ri)) ? 1 : 0;
      }
      if (rank < take) out[rank] = ri;
    }
  } else if (nfg > 0) {                                        // only foreground: P draws with replacement
    for (int q = tid; q < P; q += nt) out[q] = s_list[0][min((int)(pick[q] * (float)nfg), nfg - 1)];
    take = P;
  }
  const int nslots = P - take;
  if (nslots > 0 && nbg > 0) {
    int hard_num;
    if (nhard > 0 && neasy > 0) hard_num = min((int)((double)nslots * cfg.hard_ratio), nhard);
    else hard_num = nhard > 0 ? nslots : 0;
    for (int q = tid; q < nslots; q += nt) {
      const bool hard = q < hard_num;
      const int len = hard ? nhard : neasy;
      out[take + q] = s_list[hard ? 1 : 2][min((int)(pick[take + q] * (float)len), len - 1)];
    }
  }
  __syncthreads();
  const int n = n_gt[b];
  for (int q = tid; q < P; q += nt)
    sampled_gt[(long long)b * P + q] = n > 0 ? assignment[(long long)b * R + out[q]] : -1;
}

extern "C" int glx_roi_targets(const float* rois, const int64_t* roi_labels, int B, int R, int roi_ld,
                               const float* gt_boxes, int G, int gt_ld, int same_class, const float* key,
                               const float* pick, int P, int fg_per_image, float fg_thresh, float bg_lo,
                               float reg_fg, double hard_ratio, float* max_overlaps, int32_t* assignment,
                               int32_t* n_gt, int32_t* sampled, int32_t* sampled_gt, void* stream) {
  if (B <= 0 || P <= 0) return GLX_OK;
  GLX_REQUIRE(rois && roi_labels && gt_boxes && key && pick && max_overlaps && assignment && n_gt && sampled &&
              sampled_gt, "glx_roi_targets: null pointer");
  GLX_REQUIRE(R > 0 && R <= 8192, "glx_roi_targets: %d RoIs per frame (1..8192)", R);
  GLX_REQUIRE(G > 0 && G <= ROI_MAX_GT, "glx_roi_targets: %d ground-truth rows per frame (1..%d)", G, ROI_MAX_GT);
  GLX_REQUIRE(roi_ld >= 7 && gt_ld >= 8, "glx_roi_targets: rois need >= 7 columns, ground truths >= 8");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_roi_match, dim3(glx_divup(R, 4), B), dim3(256), 0, st, rois, roi_labels, R, roi_ld, gt_boxes, G,
                     gt_ld, same_class, max_overlaps, assignment, n_gt);
  RoiSampleCfg cfg{P, fg_per_image, fg_thresh, bg_lo, reg_fg, hard_ratio};
  static bool lds_opt_in = false;              // 3 * R ints of dynamic LDS: 96 KB at the largest R
  if (!lds_opt_in) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_roi_sample, hipFuncAttributeMaxDynamicSharedMemorySize,
                                3 * 8192 * (int)sizeof(int)));
    lds_opt_in = true;
  }
  hipLaunchKernelGGL(k_roi_sample, dim3(B), dim3(256), (size_t)3 * R * sizeof(int), st, (const float*)max_overlaps,
                     (const int*)assignment, (const int*)n_gt, key, pick, R, cfg, sampled, sampled_gt);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ RoI targets: the sampled rows
// ProposalTargetLayer.forward after the sampling (proposal_target_layer.py:13-63): the sampled RoIs and their
// matched ground truths gathered, regression mask and classification labels derived from the IoUs -- two dozen
// gather / where / compare launches as tensor ops, one block row here.  A frame without ground truth
// (sampled_gt < 0) gets zero rows.  roi_iou labels keep the tensor expression's rounding: the division by the
// Python scalar (fg - bg) is ATen's multiplication by the float reciprocal.
__global__ __launch_bounds__(256) void k_roi_target_gather(
    const float* __restrict__ rois, const long long* __restrict__ roi_labels,
    const float* __restrict__ roi_scores, int R, int ld, const float* __restrict__ gt_boxes, int G, int gld,
    const float* __restrict__ gt_unc, int ud, const float* __restrict__ max_overlaps,
    const int* __restrict__ sampled, const int* __restrict__ sampled_gt, int P, int total, float reg_fg,
    float cls_fg, float cls_bg, float cls_inv_span, int score_type, float* __restrict__ o_rois, float* __restrict__ o_gt,
    float* __restrict__ o_iou, float* __restrict__ o_scores, long long* __restrict__ o_labels,
    float* __restrict__ o_unc, long long* __restrict__ o_valid, void* __restrict__ o_cls) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = i / P;
  const int s = sampled[i], g = sampled_gt[i];
  const float* r = rois + ((long long)b * R + s) * ld;
  for (int c = 0; c < ld; ++c) o_rois[(long long)i * ld + c] = r[c];
  const float* gb = gt_boxes + ((long long)b * G + (g < 0 ? 0 : g)) * gld;
  for (int c = 0; c < gld; ++c) o_gt[(long long)i * gld + c] = g < 0 ? 0.f : gb[c];
  if (gt_unc) {
    const float* u = gt_unc + ((long long)b * G + (g < 0 ? 0 : g)) * ud;
    for (int c = 0; c < ud; ++c) o_unc[(long long)i * ud + c] = g < 0 ? 0.f : u[c];
  }
  const float iou = max_overlaps[(long long)b * R + s];
  o_iou[i] = iou;
  if (roi_scores) o_scores[i] = roi_scores[(long long)b * R + s];
  o_labels[i] = roi_labels[(long long)b * R + s];
  o_valid[i] = iou > reg_fg ? 1 : 0;
  if (score_type == 0) {
    long long lab = iou > cls_fg ? 1 : 0;
    if (iou > cls_bg && iou < cls_fg) lab = -1;
    ((long long*)o_cls)[i] = lab;
  } else {
    const bool fg = iou > cls_fg, bg = iou < cls_bg;
    ((float*)o_cls)[i] = (fg || bg) ? (fg ? 1.f : 0.f) : __fmul_rn(__fsub_rn(iou, cls_bg), cls_inv_span);
  }
}

extern "C" int glx_roi_target_gather(const float* rois, const int64_t* roi_labels, const float* roi_scores,
                                     int B, int R, int roi_ld, const float* gt_boxes, int G, int gt_ld,
                                     const float* gt_unc, int unc_ld, const float* max_overlaps,
                                     const int32_t* sampled, const int32_t* sampled_gt, int P, float reg_fg,
                                     float cls_fg, float cls_bg, float cls_inv_span, int score_type, float* out_rois,
                                     float* out_gt, float* out_iou, float* out_scores, int64_t* out_labels,
                                     float* out_unc, int64_t* out_reg_valid, void* out_cls_labels,
                                     void* stream) {
  if (B <= 0 || P <= 0) return GLX_OK;
  GLX_REQUIRE(rois && roi_labels && gt_boxes && max_overlaps && sampled && sampled_gt && out_rois && out_gt &&
              out_iou && out_labels && out_reg_valid && out_cls_labels, "glx_roi_target_gather: null pointer");
  GLX_REQUIRE(!roi_scores || out_scores, "glx_roi_target_gather: scores without an output");
  GLX_REQUIRE(!gt_unc || out_unc, "glx_roi_target_gather: uncertainties without an output");
  GLX_REQUIRE(score_type == 0 || score_type == 1, "glx_roi_target_gather: score_type 0 (cls) or 1 (roi_iou)");
  const int total = B * P;
  hipLaunchKernelGGL(k_roi_target_gather, dim3(glx_divup(total, 256)), dim3(256), 0, (hipStream_t)stream, rois,
                     (const long long*)roi_labels, roi_scores, R, roi_ld, gt_boxes, G, gt_ld, gt_unc, unc_ld,
                     max_overlaps, sampled, sampled_gt, P, total, reg_fg, cls_fg, cls_bg, cls_inv_span, score_type, out_rois,
                     out_gt, out_iou, out_scores, (long long*)out_labels, out_unc, (long long*)out_reg_valid,
                     out_cls_labels);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ top-K of the proposal scores, sorted
// RoIHeadTemplate.proposal_layer (roi_head_template.py:81-88): `torch.topk(scores, k = NMS_PRE_MAXSIZE)` of the
// 70 400 anchor scores of a frame, 9 000 kept in descending order for the NMS.  torch runs it as a multi-block radix
// select plus a segmented sort -- 45 launches, 0.23 ms of launch latency in a training step.  Here one block per
// frame does the whole thing in LDS: (1) 8-bit MSB radix select of the K-th largest key; (2) ordered compaction of
// the keys above it, plus as many of the keys equal to it as are still needed, lowest index first; (3) a stable LSD
// radix sort (4-bit digits, 8 passes) of the K survivors.  Ties: lower index first (torch leaves the order of
// equal scores unspecified); keys are the floats' bit patterns, so NaN sorts above +inf as in torch.
// Measured (tools/topk_time.py, 4 x 70 400 -> 9 000): 124 us against torch's 171 us eager / ~230 us of launch latency
// inside a HIP graph; 36 us are fixed (launch + the 8 sort passes), the six streaming passes cost ~10 us each (ballots
// and LDS histogram atomics of one CU, not load latency: 24 loads in flight per thread changed nothing).
#define TK_THREADS 1024
#define TK_MAXK 10240
#define TK_E ((TK_MAXK + TK_THREADS - 1) / TK_THREADS)      // survivors per thread in the sort, at most
#define TK_U 8                                              // loads in flight per thread while streaming the frame

__device__ __forceinline__ unsigned tk_key(float v) {       // ascending key <=> descending score
  const unsigned u = __float_as_uint(v);
  return ~((u & 0x80000000u) ? ~u : (u | 0x80000000u));
}
__device__ __forceinline__ float tk_unkey(unsigned d) {
  const unsigned a = ~d;
  return __uint_as_float((a & 0x80000000u) ? (a & 0x7fffffffu) : ~a);
}

// Step (3) of the top-K kernels: stable LSD radix sort (4-bit digits, 8 passes) of the K survivors in keys0 / idx0 through a
// permutation (keys stay where they are); writes `top` / `order` of frame `frame`.  All TK_THREADS threads of the block.
__device__ __forceinline__ void tk_sort_and_write(unsigned* keys0, unsigned* idx0, unsigned short* perm_a, unsigned short* perm_b,
                                                  unsigned short* cnt, unsigned* s_wsum, int K, int frame,
                                                  float* __restrict__ top, long long* __restrict__ order) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int E = (K + TK_THREADS - 1) / TK_THREADS;
  unsigned short* pin = perm_a;
  unsigned short* pout = perm_b;
  for (int shift = 0; shift < 32; shift += 4) {
    unsigned long long packed = 0;               // 16 counters of 4 bits: a thread holds at most TK_E = 10 keys
    for (int e = 0; e < E; ++e) {
      const int p = tid * E + e;
      if (p < K) packed += 1ull << (4 * ((keys0[pin[p]] >> shift) & 15u));
    }
#pragma unroll
    for (int dg = 0; dg < 16; ++dg) cnt[dg * TK_THREADS + tid] = (unsigned short)((packed >> (4 * dg)) & 15u);
    __syncthreads();
    // exclusive scan of the 16 x 1024 counters in (digit, thread) order: thread t owns entries [16 t, 16 t + 16)
    unsigned loc[16], sum = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { loc[j] = cnt[tid * 16 + j]; sum += loc[j]; }
    unsigned inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned v = __shfl_up(inc, off, 64);
      if (lane >= off) inc += v;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    unsigned base = inc - sum;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
#pragma unroll
    for (int j = 0; j < 16; ++j) { cnt[tid * 16 + j] = (unsigned short)base; base += loc[j]; }
    __syncthreads();
    for (int e = 0; e < E; ++e) {
      const int p = tid * E + e;
      if (p < K) {
        const unsigned short slot = pin[p];
        const unsigned dg = (keys0[slot] >> shift) & 15u;
        const unsigned short pos = cnt[dg * TK_THREADS + tid];
        cnt[dg * TK_THREADS + tid] = pos + 1;
        pout[pos] = slot;
      }
    }
    __syncthreads();
    unsigned short* t = pin; pin = pout; pout = t;
  }
  for (int p = tid; p < K; p += TK_THREADS) {
    const unsigned short slot = pin[p];
    top[(size_t)frame * K + p] = tk_unkey(keys0[slot]);
    order[(size_t)frame * K + p] = (long long)idx0[slot];
  }
}

__global__ __launch_bounds__(TK_THREADS) void k_topk_desc(const float* __restrict__ scores, int A, int K, int Kp,
                                                          float* __restrict__ top,
                                                          long long* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tk_smem[];
  unsigned* keys0 = reinterpret_cast<unsigned*>(tk_smem);                  // Kp
  unsigned* idx0 = keys0 + Kp;                                             // Kp
  unsigned short* perm_a = reinterpret_cast<unsigned short*>(idx0 + Kp);   // Kp
  unsigned short* perm_b = perm_a + Kp;                                    // Kp
  unsigned short* cnt = perm_b + Kp;                                       // 16 * TK_THREADS
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_bin, s_rem;
  __shared__ unsigned s_wlt[TK_THREADS / 64], s_weq[TK_THREADS / 64], s_wsum[TK_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* s = scores + (size_t)blockIdx.x * A;
  const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;

  // (1) the K-th smallest key, one byte at a time
  unsigned prefix = 0, mask = 0;
  unsigned remaining = (unsigned)K;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    // one block streams the frame: 8 loads in flight per thread, or the pass is one L2 latency per 4 KB
    for (int i0 = 0; i0 < A; i0 += TK_U * TK_THREADS) {
      float v[TK_U];
#pragma unroll
      for (int j = 0; j < TK_U; ++j) {
        const int i = i0 + j * TK_THREADS + tid;
        v[j] = i < A ? s[i] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < TK_U; ++j) {
        const int i = i0 + j * TK_THREADS + tid;
        const unsigned d = tk_key(v[j]);
        const bool in = i < A && (d & mask) == prefix;
        const unsigned dg = (d >> shift) & 255u;
        // scores of one frame share their high bytes: lanes that agree with the wave's first digit are added as one
        const unsigned long long act = __ballot(in);
        if (act) {
          const int first = __ffsll((long long)act) - 1;
          const unsigned lead = __shfl(dg, first, 64);
          const unsigned long long same = __ballot(in && dg == lead);
          const bool bulk = __popcll(same) >= 16;
          if (bulk && lane == first) atomicAdd(&s_hist[lead], (unsigned)__popcll(same));
          if (in && !(bulk && dg == lead)) atomicAdd(&s_hist[dg], 1u);
        }
      }
    }
    __syncthreads();
    if (wave == 0) {                               // the bin where the running count reaches `remaining`: 4 bins per lane
      unsigned c[4], sum = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { c[j] = s_hist[lane * 4 + j]; sum += c[j]; }
      unsigned inc = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned u = __shfl_up(inc, off, 64);
        if (lane >= off) inc += u;
      }
      const unsigned long long hit = __ballot(inc >= remaining);
      const int owner = hit ? __ffsll((long long)hit) - 1 : 63;
      if (lane == owner) {
        unsigned cum = inc - sum;
        int j = 0;
        for (; j < 3; ++j) {
          if (cum + c[j] >= remaining) break;
          cum += c[j];
        }
        s_bin = (unsigned)(lane * 4 + j);
        s_rem = remaining - cum;
      }
    }
    __syncthreads();
    prefix |= s_bin << shift;
    mask |= 255u << shift;
    remaining = s_rem;
    __syncthreads();
  }
  const unsigned kth = prefix;                   // keys < kth all survive, `remaining` of the keys == kth do

  // (2) ordered compaction: a wave owns a contiguous slice of the frame
  const int per = ((A + TK_THREADS / 64 - 1) / (TK_THREADS / 64) + 63) / 64 * 64;
  const int lo = wave * per, hi = min(A, lo + per);
  unsigned n_lt = 0, n_eq = 0;
  for (int i0 = lo; i0 < hi; i0 += TK_U * 64) {
    float v[TK_U];
#pragma unroll
    for (int j = 0; j < TK_U; ++j) {
      const int i = i0 + j * 64 + lane;
      v[j] = i < hi ? s[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TK_U; ++j) {
      const int i = i0 + j * 64 + lane;
      const unsigned d = tk_key(v[j]);
      n_lt += (unsigned)__popcll(__ballot(i < hi && d < kth));
      n_eq += (unsigned)__popcll(__ballot(i < hi && d == kth));
    }
  }
  if (lane == 0) { s_wlt[wave] = n_lt; s_weq[wave] = n_eq; }
  __syncthreads();
  unsigned lt_run = 0, eq_run = 0;
  for (int w = 0; w < wave; ++w) { lt_run += s_wlt[w]; eq_run += s_weq[w]; }
  for (int i0 = lo; i0 < hi; i0 += TK_U * 64) {
    float v[TK_U];
#pragma unroll
    for (int j = 0; j < TK_U; ++j) {
      const int i = i0 + j * 64 + lane;
      v[j] = i < hi ? s[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TK_U; ++j) {
      const int i = i0 + j * 64 + lane;
      const unsigned d = tk_key(v[j]);
      const bool isl = i < hi && d < kth, ise = i < hi && d == kth;
      const unsigned long long bl = __ballot(isl), be = __ballot(ise);
      const unsigned lt_before = lt_run + (unsigned)__popcll(bl & lt_mask);
      const unsigned eq_before = eq_run + (unsigned)__popcll(be & lt_mask);
      int pos = -1;
      if (isl) pos = (int)(lt_before + min(eq_before, remaining));
      else if (ise && eq_before < remaining) pos = (int)(lt_before + eq_before);
      if (pos >= 0) { keys0[pos] = d; idx0[pos] = (unsigned)i; perm_a[pos] = (unsigned short)pos; }
      lt_run += (unsigned)__popcll(bl);
      eq_run += (unsigned)__popcll(be);
    }
  }
  __syncthreads();

  tk_sort_and_write(keys0, idx0, perm_a, perm_b, cnt, s_wsum, K, (int)blockIdx.x, top, order);
}

// ---- the same result from many blocks per frame.  One block streaming a 70 400-score frame six times is what the
// single-launch kernel above spends 60 of its 124 us on; here TKM_NB blocks of a frame each keep a 1 / TKM_NB slice of the
// scores IN REGISTERS and cooperate through global memory: per radix pass a block histograms its slice in LDS, adds the
// non-empty bins to the frame's global histogram and waits at a per-frame barrier (a counter the blocks spin on -- frames x
// TKM_NB blocks are a fraction of the chip, all resident); every block then finds the bin on its own.  After four passes
// the blocks exchange their (below, equal) counts and write their survivors in index order to the frame's unsorted list;
// k_topk_sort (one block per frame, the LDS radix sort above) finishes and ZEROES the workspace for the next call --
// the workspace must be zero when the first call starts (glx_topk_workspace_bytes of zeros, owned by ONE stream).
#define TKM_NB 32
#define TKM_THREADS 256
#define TKM_E 16                                   // scores per thread at most: A <= TKM_NB * TKM_THREADS * TKM_E = 131 072
struct TkFrameWs {
  unsigned hist[4][256];
  unsigned cnt[TKM_NB][2];
  unsigned bar[8];
};

__device__ __forceinline__ void tkm_barrier(unsigned* counter) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned polls = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < TKM_NB) {
      __builtin_amdgcn_s_sleep(2);
      if (++polls > GLX_SPIN_LIMIT) break;     // partner blocks not resident (the entry point checks): do not hang the GPU
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(TKM_THREADS) void k_topk_select(const float* __restrict__ scores, int A, int K,
                                                             TkFrameWs* __restrict__ ws, unsigned* __restrict__ keys_out,
                                                             unsigned* __restrict__ idx_out) {
  const int f = blockIdx.x / TKM_NB, b = blockIdx.x - f * TKM_NB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TkFrameWs* w = ws + f;
  __shared__ unsigned s_hist[256];
  __shared__ unsigned s_bin, s_rem;
  __shared__ unsigned s_w[TKM_THREADS / 64][2];
  // the block's slice, a contiguous run per thread (index order = thread order, then element order)
  const int per_blk = ((A + TKM_NB - 1) / TKM_NB + TKM_THREADS - 1) / TKM_THREADS * TKM_THREADS;
  const int E = per_blk / TKM_THREADS;                               // <= TKM_E (checked by the host)
  const int i0 = b * per_blk + tid * E;
  const float* s = scores + (size_t)f * A;
  unsigned key[TKM_E];
#pragma unroll
  for (int e = 0; e < TKM_E; ++e) {
    const int i = i0 + e;
    key[e] = (e < E && i < A) ? tk_key(s[i]) : 0xFFFFFFFFu;          // the largest key: never among the K smallest ...
  }
  // ... unless the frame really holds NaN-free 0xFFFFFFFF keys, which tk_key never produces for a float
  unsigned prefix = 0, mask = 0, remaining = (unsigned)K;
#pragma unroll 1
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < TKM_E; ++e) {
      const int i = i0 + e;
      if (e < E && i < A && (key[e] & mask) == prefix) atomicAdd(&s_hist[(key[e] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 256 && s_hist[tid]) atomicAdd(&w->hist[pass][tid], s_hist[tid]);
    tkm_barrier(&w->bar[pass]);
    if (wave == 0) {                               // the bin where the running count reaches `remaining`: 4 bins per lane
      unsigned c[4], sum = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c[j] = __hip_atomic_load(&w->hist[pass][lane * 4 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sum += c[j];
      }
      unsigned inc = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned u = __shfl_up(inc, off, 64);
        if (lane >= off) inc += u;
      }
      const unsigned long long hit = __ballot(inc >= remaining);
      const int owner = hit ? __ffsll((long long)hit) - 1 : 63;
      if (lane == owner) {
        unsigned cum = inc - sum;
        int j = 0;
        for (; j < 3; ++j) {
          if (cum + c[j] >= remaining) break;
          cum += c[j];
        }
        s_bin = (unsigned)(lane * 4 + j);
        s_rem = remaining - cum;
      }
    }
    __syncthreads();
    prefix |= s_bin << shift;
    mask |= 255u << shift;
    remaining = s_rem;
    __syncthreads();
  }
  const unsigned kth = prefix;                   // keys < kth all survive, `remaining` of the keys == kth do (lowest index first)
  // ---- counts of the block, exchanged; position of every survivor in the frame's index order
  unsigned my_lt = 0, my_eq = 0;
#pragma unroll
  for (int e = 0; e < TKM_E; ++e) {
    const int i = i0 + e;
    if (e < E && i < A) { my_lt += key[e] < kth; my_eq += key[e] == kth; }
  }
  unsigned inc_lt = my_lt, inc_eq = my_eq;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned u = __shfl_up(inc_lt, off, 64), v = __shfl_up(inc_eq, off, 64);
    if (lane >= off) { inc_lt += u; inc_eq += v; }
  }
  if (lane == 63) { s_w[wave][0] = inc_lt; s_w[wave][1] = inc_eq; }
  __syncthreads();
  unsigned lt_run = inc_lt - my_lt, eq_run = inc_eq - my_eq, blk_lt = 0, blk_eq = 0;
#pragma unroll
  for (int ww = 0; ww < TKM_THREADS / 64; ++ww) {
    if (ww < wave) { lt_run += s_w[ww][0]; eq_run += s_w[ww][1]; }
    blk_lt += s_w[ww][0];
    blk_eq += s_w[ww][1];
  }
  if (tid == 0) {
    __hip_atomic_store(&w->cnt[b][0], blk_lt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&w->cnt[b][1], blk_eq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  tkm_barrier(&w->bar[4]);
  for (int bb = 0; bb < b; ++bb) {               // block-uniform: <= 31 pairs of loads
    lt_run += __hip_atomic_load(&w->cnt[bb][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    eq_run += __hip_atomic_load(&w->cnt[bb][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int e = 0; e < TKM_E; ++e) {
    const int i = i0 + e;
    if (!(e < E && i < A)) continue;
    const bool isl = key[e] < kth, ise = key[e] == kth;
    int pos = -1;
    if (isl) pos = (int)(lt_run + min(eq_run, remaining));
    else if (ise && eq_run < remaining) pos = (int)(lt_run + eq_run);
    if (pos >= 0) { keys_out[(size_t)f * K + pos] = key[e]; idx_out[(size_t)f * K + pos] = (unsigned)i; }
    lt_run += isl;
    eq_run += ise;
  }
}

__global__ __launch_bounds__(TK_THREADS) void k_topk_sort(const unsigned* __restrict__ keys_in, const unsigned* __restrict__ idx_in,
                                                          int K, int Kp, TkFrameWs* __restrict__ ws, float* __restrict__ top,
                                                          long long* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tk_smem[];
  unsigned* keys0 = reinterpret_cast<unsigned*>(tk_smem);                  // Kp
  unsigned* idx0 = keys0 + Kp;                                             // Kp
  unsigned short* perm_a = reinterpret_cast<unsigned short*>(idx0 + Kp);   // Kp
  unsigned short* perm_b = perm_a + Kp;                                    // Kp
  unsigned short* cnt = perm_b + Kp;                                       // 16 * TK_THREADS
  __shared__ unsigned s_wsum[TK_THREADS / 64];
  const int f = blockIdx.x, tid = threadIdx.x;
  for (int p = tid; p < K; p += TK_THREADS) {
    keys0[p] = keys_in[(size_t)f * K + p];
    idx0[p] = idx_in[(size_t)f * K + p];
    perm_a[p] = (unsigned short)p;
  }
  // the select kernel is done with this frame's workspace: clean for the next call
  for (int e = tid; e < (int)(sizeof(TkFrameWs) / 4); e += TK_THREADS) reinterpret_cast<unsigned*>(ws + f)[e] = 0u;
  __syncthreads();
  tk_sort_and_write(keys0, idx0, perm_a, perm_b, cnt, s_wsum, K, f, top, order);
}

extern "C" size_t glx_topk_workspace_bytes(int frames, int K) {
  return glx_align((size_t)(frames > 0 ? frames : 1) * sizeof(TkFrameWs)) + 2 * glx_align((size_t)(frames > 0 ? frames : 1) * K * 4);
}

// The multi-block form: `workspace` = glx_topk_workspace_bytes(frames, K) bytes that were ZERO before the first call and are
// only ever touched by these calls (they leave it zero), used by one stream at a time.
extern "C" int glx_topk_desc_ws(const float* scores, int frames, int A, int K, float* top, int64_t* order, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (frames <= 0 || K <= 0) return GLX_OK;
  GLX_REQUIRE(scores && top && order && workspace, "glx_topk_desc_ws: null pointer");
  GLX_REQUIRE(K <= A && K <= TK_MAXK, "glx_topk_desc_ws: K = %d (1..min(A = %d, %d))", K, A, TK_MAXK);
  GLX_REQUIRE(A <= TKM_NB * TKM_THREADS * TKM_E, "glx_topk_desc_ws: A = %d (<= %d)", A, TKM_NB * TKM_THREADS * TKM_E);
  GLX_REQUIRE(frames * TKM_NB <= 512, "glx_topk_desc_ws: %d frames (the cooperating blocks must all be resident)", frames);
  GLX_REQUIRE(workspace_bytes >= glx_topk_workspace_bytes(frames, K), "glx_topk_desc_ws: workspace too small");
  // the cooperating blocks spin on a per-frame counter: without the runtime's word that they are all resident together
  // (smaller partition, CU mask) the single-block form does the same job
  if (!glx_blocks_coresident((const void*)k_topk_select, TKM_THREADS, 0, frames * TKM_NB))
    return glx_topk_desc(scores, frames, A, K, top, order, stream);
  char* base = (char*)workspace;
  TkFrameWs* ws = (TkFrameWs*)base;
  unsigned* keys = (unsigned*)(base + glx_align((size_t)frames * sizeof(TkFrameWs)));
  unsigned* idx = (unsigned*)((char*)keys + glx_align((size_t)frames * K * 4));
  const int Kp = (K + 63) / 64 * 64;
  const size_t lds = (size_t)Kp * 8 + (size_t)Kp * 4 + (size_t)16 * TK_THREADS * 2;
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_topk_sort, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)TK_MAXK * 12 + 16 * TK_THREADS * 2)));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_topk_select, dim3(frames * TKM_NB), dim3(TKM_THREADS), 0, (hipStream_t)stream, scores, A, K, ws, keys, idx);
  hipLaunchKernelGGL(k_topk_sort, dim3(frames), dim3(TK_THREADS), lds, (hipStream_t)stream, (const unsigned*)keys,
                     (const unsigned*)idx, K, Kp, ws, top, (long long*)order);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_topk_max_k(void) { return TK_MAXK; }

extern "C" int glx_topk_desc(const float* scores, int frames, int A, int K, float* top, int64_t* order,
                             void* stream) {
  if (frames <= 0 || K <= 0) return GLX_OK;
  GLX_REQUIRE(scores && top && order, "glx_topk_desc: null pointer");
  GLX_REQUIRE(K <= A && K <= TK_MAXK, "glx_topk_desc: K = %d (1..min(A = %d, %d))", K, A, TK_MAXK);
  const int Kp = (K + 63) / 64 * 64;
  const size_t lds = (size_t)Kp * 8 + (size_t)Kp * 4 + (size_t)16 * TK_THREADS * 2;
  static bool attr_set = false;
  if (!attr_set) {
    GLX_HIP(hipFuncSetAttribute((const void*)k_topk_desc, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)TK_MAXK * 12 + 16 * TK_THREADS * 2)));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_topk_desc, dim3(frames), dim3(TK_THREADS), lds, (hipStream_t)stream, scores, A, K, Kp, top,
                     (long long*)order);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ proposals
// The tail of the proposal layer (pcdet/models/roi_heads/roi_head_template.py:63-126) as one launch: slot j of frame
// b takes candidate keep[b][j] of the top-k list when j < num[b], zeros otherwise; the label comes through the top-k
// order from the per-anchor class index.  Replaces arange / compare / where / three gathers / three masks / add.
__global__ void k_gather_proposals(const float* __restrict__ cand, const float* __restrict__ top,
                                   const int64_t* __restrict__ lab, const int64_t* __restrict__ order,
                                   const int64_t* __restrict__ keep, const int* __restrict__ num, int F, int A, int K,
                                   int keep_stride, int P, int C, float* __restrict__ rois, float* __restrict__ scores,
                                   int64_t* __restrict__ labels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * P) return;
  const int b = i / P, j = i - b * P;
  const bool valid = j < num[b] && j < keep_stride;
  const long long sel = valid ? keep[(size_t)b * keep_stride + j] : 0;
  const float* src = cand + ((size_t)b * K + sel) * C;
  float* dst = rois + (size_t)i * C;
  for (int c = 0; c < C; ++c) dst[c] = valid ? src[c] : 0.f;
  scores[i] = valid ? top[(size_t)b * K + sel] : 0.f;
  labels[i] = (valid ? lab[(size_t)b * A + order[(size_t)b * K + sel]] : 0) + 1;
}

extern "C" int glx_gather_proposals(const float* cand, const float* top, const int64_t* lab, const int64_t* order,
                                    const int64_t* keep, const int* num, int F, int A, int K, int keep_stride, int P,
                                    int C, float* rois, float* scores, int64_t* labels, void* stream) {
  GLX_REQUIRE(F >= 0 && A > 0 && K > 0 && P > 0 && C > 0 && keep_stride > 0, "glx_gather_proposals: bad sizes");
  if (F == 0) return GLX_OK;
  hipLaunchKernelGGL(k_gather_proposals, dim3(glx_divup(F * P, 256)), dim3(256), 0, (hipStream_t)stream, cand, top, lab,
                     order, keep, num, F, A, K, keep_stride, P, C, rois, scores, labels);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
