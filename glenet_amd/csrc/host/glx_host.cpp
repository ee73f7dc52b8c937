// libglenet_host.so -- the host-memory entry points of the hot path (include/glenet_host.h).
// No HIP, no torch, no globals: safe in forked DataLoader workers.  Build: g++ -O2 -ffp-contract=off (the
// float expressions below are written in the order the reference evaluates them; contraction or fast-math
// would change bits).
#include "../../../include/glenet_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace {

struct V2 {
  float x, y;
};

inline float fmin2(float a, float b) { return a > b ? b : a; }
inline float fmax2(float a, float b) { return a > b ? a : b; }
inline float area2(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
// twice the signed area of (p0, p1, p2)
inline float turn(V2 p1, V2 p2, V2 p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

// Two conventions share the polygon routine:
//   NMS  (pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp): boxes (7) centre + size + heading, corners rotated by
//        (cos, -sin; sin, cos), containment margin 1e-2 on half sizes;
//   OLD  (pcdet/ops/iou3d/src/iou3d_cpu.cpp): boxes (5) [x1,y1,x2,y2,ry], corners rotated by (cos, sin; -sin, cos)
//        about the rectangle's centre, containment compares the back-rotated point with the rectangle +- 1e-5.
struct NmsBox {
  static const int LD = 7;
  const float* b;
  V2 c;
  float hx, hy;       // half sizes
  float x1, y1, x2, y2;
  explicit NmsBox(const float* p) : b(p) {
    hx = p[3] / 2;
    hy = p[4] / 2;
    x1 = p[0] - hx; y1 = p[1] - hy;
    x2 = p[0] + hx; y2 = p[1] + hy;
    c.x = p[0]; c.y = p[1];
  }
  float angle() const { return b[6]; }
  float area() const { return b[3] * b[4]; }
  static V2 spin(V2 c, float co, float si, V2 p) {
    V2 r;
    r.x = (p.x - c.x) * co + (p.y - c.y) * (-si) + c.x;
    r.y = (p.x - c.x) * si + (p.y - c.y) * co + c.y;
    return r;
  }
  bool holds(V2 p) const {
    const float margin = 1e-2f;
    const float co = cosf(-b[6]), si = sinf(-b[6]);
    const float rx = (p.x - b[0]) * co + (p.y - b[1]) * (-si);
    const float ry = (p.x - b[0]) * si + (p.y - b[1]) * co;
    return fabsf(rx) < b[3] / 2 + margin && fabsf(ry) < b[4] / 2 + margin;
  }
};

struct OldBox {
  static const int LD = 5;
  const float* b;
  V2 c;
  float x1, y1, x2, y2;
  explicit OldBox(const float* p) : b(p) {
    x1 = p[0]; y1 = p[1]; x2 = p[2]; y2 = p[3];
    c.x = (x1 + x2) / 2;
    c.y = (y1 + y2) / 2;
  }
  float angle() const { return b[4]; }
  float area() const { return (b[2] - b[0]) * (b[3] - b[1]); }
  static V2 spin(V2 c, float co, float si, V2 p) {
    V2 r;
    r.x = (p.x - c.x) * co + (p.y - c.y) * si + c.x;
    r.y = -(p.x - c.x) * si + (p.y - c.y) * co + c.y;
    return r;
  }
  bool holds(V2 p) const {
    const float margin = 1e-5f;
    const float cx = (b[0] + b[2]) / 2, cy = (b[1] + b[3]) / 2;
    const float co = cosf(-b[4]), si = sinf(-b[4]);
    const float rx = (p.x - cx) * co + (p.y - cy) * si + cx;
    const float ry = -(p.x - cx) * si + (p.y - cy) * co + cy;
    return rx > b[0] - margin && rx < b[2] + margin && ry > b[1] - margin && ry < b[3] + margin;
  }
};

const float kEps = 1e-8f;

// proper crossing of segments (p0,p1) and (q0,q1): bounding boxes meet and each segment's ends lie strictly on
// opposite sides of the other; the crossing point by the area ratio, or by the two line equations when the
// ratio's denominator vanishes.
inline bool cross_point(V2 p1, V2 p0, V2 q1, V2 q0, V2* out) {
  const bool boxes_meet = fmin2(p0.x, p1.x) <= fmax2(q0.x, q1.x) && fmin2(q0.x, q1.x) <= fmax2(p0.x, p1.x) &&
                          fmin2(p0.y, p1.y) <= fmax2(q0.y, q1.y) && fmin2(q0.y, q1.y) <= fmax2(p0.y, p1.y);
  if (!boxes_meet) return false;
  const float s1 = turn(q0, p1, p0), s2 = turn(p1, q1, p0);
  const float s3 = turn(p0, q1, q0), s4 = turn(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = turn(q1, p1, p0);
  if (fabsf(s5 - s1) > kEps) {
    out->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    out->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float det = a0 * b1 - a1 * b0;
    out->x = (b0 * c1 - b1 * c0) / det;
    out->y = (a1 * c0 - a0 * c1) / det;
  }
  return true;
}

template <class Box>
inline void corners_of(const Box& bx, V2 out[5]) {
  const float co = cosf(bx.angle()), si = sinf(bx.angle());
  const V2 raw[4] = {{bx.x1, bx.y1}, {bx.x2, bx.y1}, {bx.x2, bx.y2}, {bx.x1, bx.y2}};
  for (int k = 0; k < 4; ++k) out[k] = Box::spin(bx.c, co, si, raw[k]);
  out[4] = out[0];
}

// area of the intersection polygon: edge crossings + contained corners, ordered by angle about their mean
// (adjacent swaps, as many passes as points), summed as a fan from the first vertex
template <class Box>
float overlap_area(const float* pa, const float* pb) {
  const Box A(pa), B(pb);
  V2 ca[5], cb[5];
  // the reference rotates corner k of A then corner k of B; the order does not matter to the values
  corners_of(A, ca);
  corners_of(B, cb);
  V2 poly[16];
  V2 mean = {0.f, 0.f};
  int n = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (cross_point(ca[i + 1], ca[i], cb[j + 1], cb[j], &poly[n])) {
        mean.x = mean.x + poly[n].x;
        mean.y = mean.y + poly[n].y;
        ++n;
      }
  for (int k = 0; k < 4; ++k) {
    if (A.holds(cb[k])) {
      mean.x = mean.x + cb[k].x;
      mean.y = mean.y + cb[k].y;
      poly[n++] = cb[k];
    }
    if (B.holds(ca[k])) {
      mean.x = mean.x + ca[k].x;
      mean.y = mean.y + ca[k].y;
      poly[n++] = ca[k];
    }
  }
  mean.x /= n;   // n == 0: NaN, and no vertex is looked at below
  mean.y /= n;
  for (int pass = 0; pass < n - 1; ++pass)
    for (int i = 0; i < n - pass - 1; ++i)
      if (atan2f(poly[i].y - mean.y, poly[i].x - mean.x) > atan2f(poly[i + 1].y - mean.y, poly[i + 1].x - mean.x)) {
        const V2 t = poly[i];
        poly[i] = poly[i + 1];
        poly[i + 1] = t;
      }
  float fan = 0;
  for (int k = 0; k < n - 1; ++k) {
    const V2 u = {poly[k].x - poly[0].x, poly[k].y - poly[0].y};
    const V2 w = {poly[k + 1].x - poly[0].x, poly[k + 1].y - poly[0].y};
    fan += area2(u, w);
  }
  return (float)(fabsf(fan) / 2.0);
}

template <class Box>
inline float iou_of(const float* pa, const float* pb) {
  const float sa = Box(pa).area(), sb = Box(pb).area();
  const float s = overlap_area<Box>(pa, pb);
  return s / fmaxf(sa + sb - s, kEps);
}

template <class Box, bool IOU>
int pairwise(const float* a, int N, const float* b, int M, float* out) {
  if (N < 0 || M < 0 || ((N > 0 && M > 0) && (!a || !b || !out))) return -22;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j)
      out[(size_t)i * M + j] = IOU ? iou_of<Box>(a + (size_t)i * Box::LD, b + (size_t)j * Box::LD)
                                   : overlap_area<Box>(a + (size_t)i * Box::LD, b + (size_t)j * Box::LD);
  return 0;
}

// open-addressing map cell -> voxel id for the voxelizer (a dense table would be 225 MB per KITTI frame)
struct CellMap {
  int64_t* key;
  int32_t* val;
  size_t mask;
  explicit CellMap(size_t n) {
    size_t cap = 64;
    while (cap < 2 * n + 2) cap <<= 1;
    mask = cap - 1;
    key = (int64_t*)malloc(cap * sizeof(int64_t));
    val = (int32_t*)malloc(cap * sizeof(int32_t));
    if (key) memset(key, 0xff, cap * sizeof(int64_t));
  }
  ~CellMap() { free(key); free(val); }
  bool ok() const { return key && val; }
  size_t slot(int64_t k) const {
    size_t h = ((uint64_t)k * 0x9E3779B97F4A7C15ull) >> 20 & mask;
    while (key[h] != -1 && key[h] != k) h = (h + 1) & mask;
    return h;
  }
};

}  // namespace

extern "C" {

int glxh_abi_version(void) { return 1; }

int glxh_boxes_iou_bev(const float* a, int N, const float* b, int M, float* out) {
  return pairwise<NmsBox, true>(a, N, b, M, out);
}
int glxh_iou3d_boxes_overlap_bev(const float* a, int N, const float* b, int M, float* out) {
  return pairwise<OldBox, false>(a, N, b, M, out);
}
int glxh_iou3d_boxes_iou_bev(const float* a, int N, const float* b, int M, float* out) {
  return pairwise<OldBox, true>(a, N, b, M, out);
}

int glxh_points_in_boxes(const float* boxes, int N, const float* pts, int P, int32_t* out) {
  if (N < 0 || P < 0 || ((N > 0 && P > 0) && (!boxes || !pts || !out))) return -22;
  const float margin = 1e-2f;
  for (int i = 0; i < N; ++i) {
    const float* bx = boxes + (size_t)i * 7;
    const float co = cosf(-bx[6]), si = sinf(-bx[6]);
    for (int j = 0; j < P; ++j) {
      const float* p = pts + (size_t)j * 3;
      int in = 0;
      if (!(fabsf(p[2] - bx[2]) > bx[5] / 2.0)) {
        const float sx = p[0] - bx[0], sy = p[1] - bx[1];
        const float lx = sx * co + sy * (-si);
        const float ly = sx * si + sy * co;
        in = (fabsf(lx) < bx[3] / 2.0 + margin) & (fabsf(ly) < bx[4] / 2.0 + margin);
      }
      out[(size_t)i * P + j] = in;
    }
  }
  return 0;
}

int glxh_voxelize_hard(const float* points, int P, int C, const float* range, const float* vsize, const int* grid,
                       int max_points, int max_voxels, float* voxels, int32_t* coords, int32_t* num_points,
                       int* num_voxels) {
  if (P < 0 || C < 3 || max_points <= 0 || max_voxels <= 0 || !range || !vsize || !grid || !voxels || !coords ||
      !num_points || !num_voxels || (P > 0 && !points))
    return -22;
  CellMap map((size_t)(P < max_voxels ? P : max_voxels) + 1);
  if (!map.ok()) return -12;
  memset(voxels, 0, (size_t)max_voxels * max_points * C * sizeof(float));
  memset(num_points, 0, (size_t)max_voxels * sizeof(int32_t));
  int made = 0;
  for (int i = 0; i < P; ++i) {
    const float* p = points + (size_t)i * C;
    int c[3];
    bool inside = true;
    for (int a = 0; a < 3 && inside; ++a) {
      const float f = floorf((p[a] - range[a]) / vsize[a]);
      inside = f >= 0.f && f < (float)grid[a];
      c[a] = inside ? (int)f : 0;
    }
    if (!inside) continue;
    const int64_t cell = ((int64_t)c[2] * grid[1] + c[1]) * grid[0] + c[0];
    const size_t h = map.slot(cell);
    int v;
    if (map.key[h] == cell) {
      v = map.val[h];
    } else {
      if (made >= max_voxels) continue;       // the generator's `continue`: later points of NEW cells are dropped
      v = made++;
      map.key[h] = cell;
      map.val[h] = v;
      coords[(size_t)v * 3 + 0] = c[2];
      coords[(size_t)v * 3 + 1] = c[1];
      coords[(size_t)v * 3 + 2] = c[0];
    }
    const int n = num_points[v];
    if (n < max_points) {
      memcpy(voxels + ((size_t)v * max_points + n) * C, p, (size_t)C * sizeof(float));
      num_points[v] = n + 1;
    }
  }
  *num_voxels = made;
  return 0;
}

}  // extern "C"
