/*
 * glenet_oracle.c -- CPU restatement of GLENet's detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under glenet_amd/ may import, link or call this
 * file; it is the checker used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg, never the thing shipped or measured as the product.
 *
 * Each function cites the reference source (path:line under the GLENet tree) whose
 * arithmetic it restates.  Pinning status (see DESIGN.md "Oracle"):
 *   - rotated BEV overlap / IoU / NMS sweep: pinned.  The iou3d-convention variant is
 *     checked bit-for-bit against the reference's own pcdet/ops/iou3d/src/iou3d_cpu.cpp
 *     compiled as oracle/_ref (tests/golden/iou3d_ref_*.npz); the iou3d_nms-convention
 *     variant shares every helper and differs only in corner construction, rotation sign
 *     and MARGIN, cited line by line below.
 *   - voxelization, sparse convolution, dense(): PARITY UNPINNED by the reference (the
 *     arithmetic lives in third-party spconv, not vendored, no version pinned, not
 *     installed here); restated from the published algorithm + the reference call
 *     sites, and cross-checked against torch.nn.functional.conv3d in tests/.
 *   - voxel_query / ball_query / group_points / roiaware / roipoint / points_in_boxes:
 *     restated from the CUDA kernels cited below (GPU-only code, cannot run here);
 *     PARITY UNPINNED by any reference test, cross-checked by independent numpy
 *     formulations in tests/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* Threads for the CPU-baseline timing legs of bench.py (OpenMP; default 1 = the plain serial loops every
 * test checks against).  With n > 1 the loops below whose iterations are independent run in parallel --
 * same arithmetic per element, same results; the one exception is the weight gradient of
 * orc_sconv_backward, which then sums per-thread partial slabs (order of the fp32 additions differs). */
static int orc_nthreads = 1;
ORC_API void orc_set_threads(int n) { orc_nthreads = n > 1 ? n : 1; }
ORC_API int orc_get_threads(void) { return orc_nthreads; }
#define ORC_PAR_FOR _Pragma("omp parallel for schedule(static) if(orc_nthreads > 1) num_threads(orc_nthreads)")

/* ======================================================================== voxelization */

/* Hard voxelization.  Restates spconv's CPU generator (points_to_voxel_3d_np /
 * Point2VoxelCPU3d::point_to_voxel), called at
 * pcdet/datasets/processor/data_processor.py:44-60.  Per point: c = floor((p - min)/size)
 * per axis in fp32, reject if outside the grid; look the cell up in a cell->voxel map;
 * a new cell becomes voxel #voxel_num unless voxel_num >= max_voxels (then the point is
 * skipped, `continue`); the point is appended if the voxel holds < max_points.
 * coords are written [z, y, x] (data_processor.py:145, dataset.py:192-197).
 * Returns the number of voxels. */
ORC_API int orc_voxelize_hard(const float* points, int P, int C, const float* range,
                              const float* vsize, const int* grid /*gx,gy,gz*/, int max_points,
                              int max_voxels, float* voxels, int32_t* coords,
                              int32_t* num_points) {
  int gx = grid[0], gy = grid[1], gz = grid[2];
  size_t ncell = (size_t)gx * gy * gz;
  int32_t* cell2vox = (int32_t*)calloc(ncell, sizeof(int32_t)); /* 0 = empty, else id+1 */
  if (!cell2vox) return -1;
  int voxel_num = 0;
  memset(voxels, 0, (size_t)max_voxels * max_points * C * sizeof(float));
  memset(num_points, 0, (size_t)max_voxels * sizeof(int32_t));
  for (int i = 0; i < P; ++i) {
    const float* p = points + (size_t)i * C;
    int c[3];
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      float f = floorf((p[j] - range[j]) / vsize[j]);
      if (!(f >= 0.f && f < (float)grid[j])) { failed = 1; break; }
      c[j] = (int)f;
    }
    if (failed) continue;
    size_t cell = ((size_t)c[2] * gy + c[1]) * gx + c[0];
    int vid = cell2vox[cell] - 1;
    if (vid == -1) {
      if (voxel_num >= max_voxels) continue;
      vid = voxel_num++;
      cell2vox[cell] = vid + 1;
      coords[vid * 3 + 0] = c[2];
      coords[vid * 3 + 1] = c[1];
      coords[vid * 3 + 2] = c[0];
    }
    int n = num_points[vid];
    if (n < max_points) {
      memcpy(voxels + ((size_t)vid * max_points + n) * C, p, C * sizeof(float));
      num_points[vid] = n + 1;
    }
  }
  free(cell2vox);
  return voxel_num;
}

/* MeanVFE.forward, pcdet/models/backbones_3d/vfe/mean_vfe.py:26-29. */
ORC_API void orc_mean_vfe(const float* voxels, const int32_t* num_points, int Nv, int mp, int C,
                          float* out) {
  for (int v = 0; v < Nv; ++v)
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int p = 0; p < mp; ++p) s += voxels[((size_t)v * mp + p) * C + c];
      float n = (float)num_points[v];
      out[(size_t)v * C + c] = s / (n < 1.f ? 1.f : n);
    }
}

/* ======================================================================== sparse conv */

typedef struct {
  int64_t* keys;
  int32_t* vals;
  size_t cap; /* power of two */
} orc_map;

static void map_init(orc_map* m, size_t n) {
  size_t cap = 16;
  while (cap < n * 2 + 1) cap <<= 1;
  m->cap = cap;
  m->keys = (int64_t*)malloc(cap * sizeof(int64_t));
  m->vals = (int32_t*)malloc(cap * sizeof(int32_t));
  for (size_t i = 0; i < cap; ++i) m->keys[i] = -1;
}
static void map_free(orc_map* m) { free(m->keys); free(m->vals); }
static size_t map_slot(const orc_map* m, int64_t key) {
  uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ull;
  size_t s = (size_t)(h >> 20) & (m->cap - 1);
  while (m->keys[s] != -1 && m->keys[s] != key) s = (s + 1) & (m->cap - 1);
  return s;
}
static int map_get(const orc_map* m, int64_t key) {
  size_t s = map_slot(m, key);
  return m->keys[s] == key ? m->vals[s] : -1;
}
static void map_put(orc_map* m, int64_t key, int32_t v) {
  size_t s = map_slot(m, key);
  m->keys[s] = key;
  m->vals[s] = v;
}

static int64_t lin4(int b, int z, int y, int x, int D, int H, int W) {
  return (((int64_t)b * D + z) * H + y) * W + x;
}

static int cmp_i64(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

/* Output index set of a regular (strided) sparse conv: every output cell o with
 * o*stride - pad + k == i for some active input i and kernel offset k
 * (spconv getIndicePairs / get_indice_pairs, called for spconv_backbone.py:90,97,104,113).
 * spconv leaves the output ORDER implementation-defined; this build defines it as
 * ascending linear (b, z, y, x).  out_indices must hold N_in*K rows; returns N_out. */
ORC_API int orc_outset_strided_dil(const int32_t* in_idx, int N_in, const int* shape /*D,H,W*/,
                                   const int* ksize, const int* stride, const int* pad, const int* dil,
                                   const int* oshape, int32_t* out_indices);
ORC_API int orc_outset_strided(const int32_t* in_idx, int N_in, const int* shape /*D,H,W*/,
                               const int* ksize, const int* stride, const int* pad,
                               const int* oshape, int32_t* out_indices) {
  const int one[3] = {1, 1, 1};
  return orc_outset_strided_dil(in_idx, N_in, shape, ksize, stride, pad, one, oshape, out_indices);
}
/* ... with dilation (spconv SparseConv3d(dilation=...)): output o is reached through offset k when o * s - p + k * d == c */
ORC_API int orc_outset_strided_dil(const int32_t* in_idx, int N_in, const int* shape /*D,H,W*/,
                                   const int* ksize, const int* stride, const int* pad, const int* dil,
                                   const int* oshape, int32_t* out_indices) {
  int K = ksize[0] * ksize[1] * ksize[2];
  int64_t* cand = (int64_t*)malloc((size_t)(N_in > 0 ? N_in : 1) * K * sizeof(int64_t));
  size_t nc = 0;
  (void)shape;
  for (int i = 0; i < N_in; ++i) {
    const int32_t* c = in_idx + (size_t)i * 4;
    for (int kz = 0; kz < ksize[0]; ++kz)
      for (int ky = 0; ky < ksize[1]; ++ky)
        for (int kx = 0; kx < ksize[2]; ++kx) {
          int n[3] = {c[1] + pad[0] - kz * dil[0], c[2] + pad[1] - ky * dil[1], c[3] + pad[2] - kx * dil[2]};
          int ok = 1, o[3];
          for (int d = 0; d < 3; ++d) {
            if (n[d] < 0 || n[d] % stride[d]) { ok = 0; break; }
            o[d] = n[d] / stride[d];
            if (o[d] >= oshape[d]) { ok = 0; break; }
          }
          if (ok) cand[nc++] = lin4(c[0], o[0], o[1], o[2], oshape[0], oshape[1], oshape[2]);
        }
  }
  qsort(cand, nc, sizeof(int64_t), cmp_i64);
  int n_out = 0;
  for (size_t i = 0; i < nc; ++i) {
    if (i && cand[i] == cand[i - 1]) continue;
    int64_t l = cand[i];
    int x = (int)(l % oshape[2]); l /= oshape[2];
    int y = (int)(l % oshape[1]); l /= oshape[1];
    int z = (int)(l % oshape[0]);
    int b = (int)(l / oshape[0]);
    int32_t* o = out_indices + (size_t)n_out * 4;
    o[0] = b; o[1] = z; o[2] = y; o[3] = x;
    ++n_out;
  }
  free(cand);
  return n_out;
}

/* Rule ("indice pair") generation, classic per-offset pair lists:
 *   pairs_in[k][p], pairs_out[k][p], p < n_pairs[k]; arrays are (K, max(N_in,N_out)).
 * subm != 0: output set == input set, neighbour = cell + (k - ksize/2)   (SubMConv3d)
 * subm == 0: input cell = out*stride - pad + k                           (SparseConv3d)
 * Cross-correlation convention, identical to torch.nn.functional.conv3d. */
ORC_API int orc_build_rules_dil(const int32_t* in_idx, int N_in, const int32_t* out_idx, int N_out,
                                const int* shape, const int* ksize, const int* stride,
                                const int* pad, const int* dil, int subm, int32_t* pairs_in, int32_t* pairs_out,
                                int32_t* n_pairs);
ORC_API int orc_build_rules(const int32_t* in_idx, int N_in, const int32_t* out_idx, int N_out,
                            const int* shape, const int* ksize, const int* stride,
                            const int* pad, int subm, int32_t* pairs_in, int32_t* pairs_out,
                            int32_t* n_pairs) {
  const int one[3] = {1, 1, 1};
  return orc_build_rules_dil(in_idx, N_in, out_idx, N_out, shape, ksize, stride, pad, one, subm, pairs_in, pairs_out, n_pairs);
}
ORC_API int orc_build_rules_dil(const int32_t* in_idx, int N_in, const int32_t* out_idx, int N_out,
                                const int* shape, const int* ksize, const int* stride,
                                const int* pad, const int* dil, int subm, int32_t* pairs_in, int32_t* pairs_out,
                                int32_t* n_pairs) {
  int K = ksize[0] * ksize[1] * ksize[2];
  int D = shape[0], H = shape[1], W = shape[2];
  size_t ld = (size_t)(N_in > N_out ? N_in : N_out);
  orc_map m;
  map_init(&m, (size_t)N_in);
  for (int i = 0; i < N_in; ++i) {
    const int32_t* c = in_idx + (size_t)i * 4;
    map_put(&m, lin4(c[0], c[1], c[2], c[3], D, H, W), i);
  }
  long total = 0;
  for (int k = 0; k < K; ++k) n_pairs[k] = 0;
  for (int j = 0; j < N_out; ++j) {
    const int32_t* o = out_idx + (size_t)j * 4;
    int k = 0;
    for (int kz = 0; kz < ksize[0]; ++kz)
      for (int ky = 0; ky < ksize[1]; ++ky)
        for (int kx = 0; kx < ksize[2]; ++kx, ++k) {
          int z, y, x;
          if (subm) {
            z = o[1] + (kz - ksize[0] / 2) * dil[0]; y = o[2] + (ky - ksize[1] / 2) * dil[1];
            x = o[3] + (kx - ksize[2] / 2) * dil[2];
          } else {
            z = o[1] * stride[0] - pad[0] + kz * dil[0];
            y = o[2] * stride[1] - pad[1] + ky * dil[1];
            x = o[3] * stride[2] - pad[2] + kx * dil[2];
          }
          if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) continue;
          int i = map_get(&m, lin4(o[0], z, y, x, D, H, W));
          if (i < 0) continue;
          int p = n_pairs[k]++;
          pairs_in[(size_t)k * ld + p] = i;
          pairs_out[(size_t)k * ld + p] = j;
          ++total;
        }
  }
  map_free(&m);
  return (int)total;
}

/* Forward: gather -> GEMM -> scatter-add per kernel offset, fp32 accumulate, weights
 * (K, Cin, Cout) (spconv 1.x layout, detector3d_template.py:377-384).  out is overwritten. */
ORC_API void orc_sconv_forward(const float* in, const float* W, const float* bias,
                               const int32_t* pairs_in, const int32_t* pairs_out,
                               const int32_t* n_pairs, int K, int ld, int N_out, int Cin,
                               int Cout, float* out) {
  for (int j = 0; j < N_out; ++j)
    for (int c = 0; c < Cout; ++c) out[(size_t)j * Cout + c] = bias ? bias[c] : 0.f;
  for (int k = 0; k < K; ++k) {
    const float* Wk = W + (size_t)k * Cin * Cout;
    const int np_k = n_pairs[k];
    ORC_PAR_FOR   /* the outputs of one offset are distinct rows */
    for (int p = 0; p < np_k; ++p) {
      const float* x = in + (size_t)pairs_in[(size_t)k * ld + p] * Cin;
      float* y = out + (size_t)pairs_out[(size_t)k * ld + p] * Cout;
      for (int ci = 0; ci < Cin; ++ci) {
        float xv = x[ci];
        const float* w = Wk + (size_t)ci * Cout;
        for (int co = 0; co < Cout; ++co) y[co] += xv * w[co];
      }
    }
  }
}

/* Backward of the above: din (N_in,Cin), dW (K,Cin,Cout) both overwritten. */
ORC_API void orc_sconv_backward(const float* in, const float* W, const float* dout,
                                const int32_t* pairs_in, const int32_t* pairs_out,
                                const int32_t* n_pairs, int K, int ld, int N_in, int Cin,
                                int Cout, float* din, float* dW) {
  memset(din, 0, (size_t)N_in * Cin * sizeof(float));
  memset(dW, 0, (size_t)K * Cin * Cout * sizeof(float));
  if (orc_nthreads > 1) { /* baseline timing only: inputs of one offset are distinct rows; dW in per-thread slabs */
    for (int k = 0; k < K; ++k) {
      const float* Wk = W + (size_t)k * Cin * Cout;
      float* dWk = dW + (size_t)k * Cin * Cout;
      const int np_k = n_pairs[k];
      _Pragma("omp parallel num_threads(orc_nthreads)")
      {
        float* part = (float*)calloc((size_t)Cin * Cout, sizeof(float));
        _Pragma("omp for schedule(static)")
        for (int p = 0; p < np_k; ++p) {
          int i = pairs_in[(size_t)k * ld + p], j = pairs_out[(size_t)k * ld + p];
          const float* x = in + (size_t)i * Cin;
          const float* g = dout + (size_t)j * Cout;
          float* dx = din + (size_t)i * Cin;
          for (int ci = 0; ci < Cin; ++ci) {
            const float* w = Wk + (size_t)ci * Cout;
            float* dw = part + (size_t)ci * Cout;
            float acc = 0.f, xv = x[ci];
            for (int co = 0; co < Cout; ++co) {
              acc += g[co] * w[co];
              dw[co] += xv * g[co];
            }
            dx[ci] += acc;
          }
        }
        _Pragma("omp critical")
        for (int e = 0; e < Cin * Cout; ++e) dWk[e] += part[e];
        free(part);
      }
    }
    return;
  }
  for (int k = 0; k < K; ++k) {
    const float* Wk = W + (size_t)k * Cin * Cout;
    float* dWk = dW + (size_t)k * Cin * Cout;
    for (int p = 0; p < n_pairs[k]; ++p) {
      int i = pairs_in[(size_t)k * ld + p], j = pairs_out[(size_t)k * ld + p];
      const float* x = in + (size_t)i * Cin;
      const float* g = dout + (size_t)j * Cout;
      float* dx = din + (size_t)i * Cin;
      for (int ci = 0; ci < Cin; ++ci) {
        const float* w = Wk + (size_t)ci * Cout;
        float* dw = dWk + (size_t)ci * Cout;
        float acc = 0.f, xv = x[ci];
        for (int co = 0; co < Cout; ++co) {
          acc += g[co] * w[co];
          dw[co] += xv * g[co];
        }
        dx[ci] += acc;
      }
    }
  }
}

/* SparseConvTensor.dense(): (B, C, D, H, W), zero background (height_compression.py:21). */
ORC_API void orc_dense(const float* f, const int32_t* idx, int N, int C, int B, int D, int H,
                       int W, float* out) {
  memset(out, 0, (size_t)B * C * D * H * W * sizeof(float));
  for (int n = 0; n < N; ++n) {
    const int32_t* p = idx + (size_t)n * 4;
    for (int c = 0; c < C; ++c)
      out[((((size_t)p[0] * C + c) * D + p[1]) * H + p[2]) * W + p[3]] = f[(size_t)n * C + c];
  }
}

/* ======================================================================== rotated IoU */

typedef struct { float x, y; } pt2;

/* pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:59-66 (== iou3d_nms_kernel.cu:35-42) */
static float cross2(pt2 a, pt2 b) { return a.x * b.y - a.y * b.x; }
static float cross3(pt2 p1, pt2 p2, pt2 p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static float fminf_(float a, float b) { return a > b ? b : a; } /* iou3d_cpu.cpp:30-36 */
static float fmaxf_(float a, float b) { return a > b ? a : b; }

/* iou3d_cpu.cpp:68-74 */
static int check_rect_cross(pt2 p1, pt2 p2, pt2 q1, pt2 q2) {
  return fminf_(p1.x, p2.x) <= fmaxf_(q1.x, q2.x) && fminf_(q1.x, q2.x) <= fmaxf_(p1.x, p2.x) &&
         fminf_(p1.y, p2.y) <= fmaxf_(q1.y, q2.y) && fminf_(q1.y, q2.y) <= fmaxf_(p1.y, p2.y);
}

#define ORC_EPS 1e-8f

/* iou3d_cpu.cpp:88-117: segment intersection, strict crossing test, line-equation fallback */
static int intersection(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2* ans) {
  if (check_rect_cross(p0, p1, q0, q1) == 0) return 0;
  float s1 = cross3(q0, p1, p0);
  float s2 = cross3(p1, q1, p0);
  float s3 = cross3(p0, q1, q0);
  float s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > ORC_EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

/* iou3d_cpu.cpp:125-127 */
static int point_cmp(pt2 a, pt2 b, pt2 c) {
  return atan2f(a.y - c.y, a.x - c.x) > atan2f(b.y - c.y, b.x - c.x);
}

/* Shared tail of box_overlap: corner tests were done by the caller (they differ between
 * the two libraries); bubble sort by angle and shoelace area.
 * iou3d_cpu.cpp:196-224 == iou3d/src/iou3d_cpu.cpp:214-243. */
static float polygon_area(pt2* cross_points, int cnt, pt2 poly_center) {
  poly_center.x /= cnt;
  poly_center.y /= cnt;
  for (int j = 0; j < cnt - 1; j++)
    for (int i = 0; i < cnt - j - 1; i++)
      if (point_cmp(cross_points[i], cross_points[i + 1], poly_center)) {
        pt2 t = cross_points[i];
        cross_points[i] = cross_points[i + 1];
        cross_points[i + 1] = t;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; k++) {
    pt2 a = {cross_points[k].x - cross_points[0].x, cross_points[k].y - cross_points[0].y};
    pt2 b = {cross_points[k + 1].x - cross_points[0].x, cross_points[k + 1].y - cross_points[0].y};
    area += cross2(a, b);
  }
  return (float)(fabs(area) / 2.0);
}

/* ---- iou3d_nms convention: boxes [x,y,z,dx,dy,dz,heading] ------------------------- */

/* iou3d_cpu.cpp:76-86: MARGIN 1e-2, point rotated by -heading about the centre */
static int check_in_box2d_nms(const float* box, pt2 p) {
  const float MARGIN = 1e-2f;
  float center_x = box[0], center_y = box[1];
  float angle_cos = cosf(-box[6]), angle_sin = sinf(-box[6]);
  float rot_x = (p.x - center_x) * angle_cos + (p.y - center_y) * (-angle_sin);
  float rot_y = (p.x - center_x) * angle_sin + (p.y - center_y) * angle_cos;
  /* `box[3] / 2 + MARGIN` is float/int -> float, + float MARGIN (declared const float) */
  return (fabsf(rot_x) < box[3] / 2 + MARGIN && fabsf(rot_y) < box[4] / 2 + MARGIN);
}

/* iou3d_cpu.cpp:119-123: (cos, -sin; sin, cos) */
static pt2 rotate_nms(pt2 c, float ac, float as, pt2 p) {
  pt2 r;
  r.x = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
  r.y = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
  return r;
}

/* iou3d_cpu.cpp:129-225 (box_overlap) */
static float box_overlap_nms(const float* box_a, const float* box_b) {
  float a_angle = box_a[6], b_angle = box_b[6];
  float a_dx_half = box_a[3] / 2, b_dx_half = box_b[3] / 2, a_dy_half = box_a[4] / 2,
        b_dy_half = box_b[4] / 2;
  float a_x1 = box_a[0] - a_dx_half, a_y1 = box_a[1] - a_dy_half;
  float a_x2 = box_a[0] + a_dx_half, a_y2 = box_a[1] + a_dy_half;
  float b_x1 = box_b[0] - b_dx_half, b_y1 = box_b[1] - b_dy_half;
  float b_x2 = box_b[0] + b_dx_half, b_y2 = box_b[1] + b_dy_half;
  pt2 center_a = {box_a[0], box_a[1]}, center_b = {box_b[0], box_b[1]};
  pt2 ca[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  pt2 cb[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
  float a_cos = cosf(a_angle), a_sin = sinf(a_angle);
  float b_cos = cosf(b_angle), b_sin = sinf(b_angle);
  for (int k = 0; k < 4; k++) {
    ca[k] = rotate_nms(center_a, a_cos, a_sin, ca[k]);
    cb[k] = rotate_nms(center_b, b_cos, b_sin, cb[k]);
  }
  ca[4] = ca[0];
  cb[4] = cb[0];
  pt2 cross_points[16];
  pt2 poly_center = {0, 0};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      int flag = intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &cross_points[cnt]);
      if (flag) {
        poly_center.x += cross_points[cnt].x;
        poly_center.y += cross_points[cnt].y;
        cnt++;
      }
    }
  for (int k = 0; k < 4; k++) {
    if (check_in_box2d_nms(box_a, cb[k])) {
      poly_center.x += cb[k].x; poly_center.y += cb[k].y;
      cross_points[cnt++] = cb[k];
    }
    if (check_in_box2d_nms(box_b, ca[k])) {
      poly_center.x += ca[k].x; poly_center.y += ca[k].y;
      cross_points[cnt++] = ca[k];
    }
  }
  return polygon_area(cross_points, cnt, poly_center);
}

/* iou3d_cpu.cpp:227-234 */
static float iou_bev_nms(const float* a, const float* b) {
  float sa = a[3] * a[4], sb = b[3] * b[4];
  float s = box_overlap_nms(a, b);
  return s / fmaxf(sa + sb - s, ORC_EPS);
}

/* boxes_overlap_bev (iou3d_nms.cpp:49-68 -> iou3d_nms_kernel.cu:236-249), CPU restatement */
ORC_API void orc_boxes_overlap_bev(const float* a, int N, const float* b, int M, float* out) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) out[(size_t)i * M + j] = box_overlap_nms(a + i * 7, b + j * 7);
}
/* boxes_iou_bev_cpu, iou3d_cpu.cpp:232-252 */
ORC_API void orc_boxes_iou_bev(const float* a, int N, const float* b, int M, float* out) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) out[(size_t)i * M + j] = iou_bev_nms(a + i * 7, b + j * 7);
}

/* IoU of listed pairs (a = boxes[pairs[2p]], b = boxes[pairs[2p+1]]): the census of tools/nms_census.py evaluates
 * the candidate pairs of 9000-box frames (everything else is provably 0) instead of all 81 M. */
ORC_API void orc_iou_bev_pairs(const float* boxes, const int32_t* pairs, long long P, float* out) {
  ORC_PAR_FOR
  for (long long p = 0; p < P; ++p)
    out[p] = iou_bev_nms(boxes + (size_t)pairs[2 * p] * 7, boxes + (size_t)pairs[2 * p + 1] * 7);
}

/* iou_normal, iou3d_nms_kernel.cu:314-325 */
static float iou_normal(const float* a, const float* b) {
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  float interS = width * height;
  float Sa = a[3] * a[4], Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, ORC_EPS);
}

/* nms_gpu = nms_kernel bit matrix (iou3d_nms_kernel.cu:267-311: upper triangle, strict
 * `iou > thresh`) + the host sweep of iou3d_nms.cpp:116-132.  boxes already sorted by score
 * (iou3d_nms_utils.py:190-194).  normal != 0 selects nms_normal (kernel.cu:328-372).
 * keep[] receives indices into boxes; returns the count. */
ORC_API int orc_nms(const float* boxes, int N, float thresh, int normal, int64_t* keep) {
  int col_blocks = (N + 63) / 64;
  uint64_t* remv = (uint64_t*)calloc(col_blocks > 0 ? col_blocks : 1, sizeof(uint64_t));
  uint64_t* row = (uint64_t*)malloc((col_blocks > 0 ? col_blocks : 1) * sizeof(uint64_t));
  int num = 0;
  for (int i = 0; i < N; ++i) {
    int nblock = i / 64, inblock = i % 64;
    if (remv[nblock] & (1ULL << inblock)) continue;
    keep[num++] = i;
    /* row i of the mask, computed lazily: bit j set iff j > i and iou(i,j) > thresh */
    memset(row, 0, col_blocks * sizeof(uint64_t));
    ORC_PAR_FOR   /* one 64-column word per iteration: words are written by one thread each */
    for (int jw = nblock; jw < col_blocks; ++jw) {
      int j0 = jw * 64 > i + 1 ? jw * 64 : i + 1, j1 = (jw + 1) * 64 < N ? (jw + 1) * 64 : N;
      for (int j = j0; j < j1; ++j) {
        float v = normal ? iou_normal(boxes + i * 7, boxes + j * 7)
                         : iou_bev_nms(boxes + i * 7, boxes + j * 7);
        if (v > thresh) row[jw] |= 1ULL << (j % 64);
      }
    }
    for (int j = nblock; j < col_blocks; ++j) remv[j] |= row[j];
  }
  free(remv);
  free(row);
  return num;
}

/* ---- iou3d (older) convention: boxes [x1,y1,x2,y2,ry] ------------------------------ */

/* pcdet/ops/iou3d/src/iou3d_cpu.cpp:56-71: MARGIN 1e-5, +sin rotation, compare to corners */
static int check_in_box2d_old(const float* box, pt2 p) {
  const float MARGIN = 1e-5f;
  float center_x = (box[0] + box[2]) / 2;
  float center_y = (box[1] + box[3]) / 2;
  float angle_cos = cosf(-box[4]), angle_sin = sinf(-box[4]);
  float rot_x = (p.x - center_x) * angle_cos + (p.y - center_y) * angle_sin + center_x;
  float rot_y = -(p.x - center_x) * angle_sin + (p.y - center_y) * angle_cos + center_y;
  return (rot_x > box[0] - MARGIN && rot_x < box[2] + MARGIN && rot_y > box[1] - MARGIN &&
          rot_y < box[3] + MARGIN);
}

/* iou3d/src/iou3d_cpu.cpp:121-125 */
static pt2 rotate_old(pt2 c, float ac, float as, pt2 p) {
  pt2 r;
  r.x = (p.x - c.x) * ac + (p.y - c.y) * as + c.x;
  r.y = -(p.x - c.x) * as + (p.y - c.y) * ac + c.y;
  return r;
}

/* iou3d/src/iou3d_cpu.cpp:127-244 with input_2d == 1 */
static float box_overlap_old(const float* box_a, const float* box_b) {
  float a_x1 = box_a[0], a_y1 = box_a[1], a_x2 = box_a[2], a_y2 = box_a[3], a_angle = box_a[4];
  float b_x1 = box_b[0], b_y1 = box_b[1], b_x2 = box_b[2], b_y2 = box_b[3], b_angle = box_b[4];
  pt2 center_a = {(a_x1 + a_x2) / 2, (a_y1 + a_y2) / 2};
  pt2 center_b = {(b_x1 + b_x2) / 2, (b_y1 + b_y2) / 2};
  pt2 ca[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  pt2 cb[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
  float a_cos = cosf(a_angle), a_sin = sinf(a_angle);
  float b_cos = cosf(b_angle), b_sin = sinf(b_angle);
  for (int k = 0; k < 4; k++) {
    ca[k] = rotate_old(center_a, a_cos, a_sin, ca[k]);
    cb[k] = rotate_old(center_b, b_cos, b_sin, cb[k]);
  }
  ca[4] = ca[0];
  cb[4] = cb[0];
  pt2 cross_points[16];
  pt2 poly_center = {0, 0};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      int flag = intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &cross_points[cnt]);
      if (flag) {
        poly_center.x += cross_points[cnt].x;
        poly_center.y += cross_points[cnt].y;
        cnt++;
      }
    }
  for (int k = 0; k < 4; k++) {
    if (check_in_box2d_old(box_a, cb[k])) {
      poly_center.x += cb[k].x; poly_center.y += cb[k].y;
      cross_points[cnt++] = cb[k];
    }
    if (check_in_box2d_old(box_b, ca[k])) {
      poly_center.x += ca[k].x; poly_center.y += ca[k].y;
      cross_points[cnt++] = ca[k];
    }
  }
  return polygon_area(cross_points, cnt, poly_center);
}

/* boxes_overlap_bev_cpu, iou3d/src/iou3d_cpu.cpp:258-280 */
ORC_API void orc_iou3d_boxes_overlap_bev(const float* a, int N, const float* b, int M,
                                         float* out) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) out[(size_t)i * M + j] = box_overlap_old(a + i * 5, b + j * 5);
}
/* boxes_iou_bev_cpu, iou3d/src/iou3d_cpu.cpp:246-253,283-304 */
ORC_API void orc_iou3d_boxes_iou_bev(const float* a, int N, const float* b, int M, float* out) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < M; ++j) {
      const float* A = a + i * 5;
      const float* B = b + j * 5;
      float sa = (A[2] - A[0]) * (A[3] - A[1]);
      float sb = (B[2] - B[0]) * (B[3] - B[1]);
      float s = box_overlap_old(A, B);
      out[(size_t)i * M + j] = s / fmaxf(sa + sb - s, ORC_EPS);
    }
}
/* boxes_aligned_overlap_kernel, iou3d/src/iou3d_kernel.cu:284-293: out[i] = overlap(a[i], b[i]) */
ORC_API void orc_iou3d_boxes_aligned_overlap_bev(const float* a, const float* b, int N,
                                                 float* out) {
  for (int i = 0; i < N; ++i) out[i] = box_overlap_old(a + i * 5, b + i * 5);
}

/* ======================================================================== point / box ops */

/* lidar_to_local_coords + check_pt_in_box3d.
 * GPU variant: roiaware_pool3d_kernel.cu:15-36 (MARGIN 1e-5);
 * CPU variant: roiaware_pool3d.cpp:120-139 (MARGIN 1e-2).  z test `fabsf(z-cz) > dz/2.0`
 * promotes to double (dz / 2.0), xy tests compare fp32 local coords with a double bound. */
static int pt_in_box3d(const float* pt, const float* box3d, double margin, float* lx, float* ly) {
  float x = pt[0], y = pt[1], z = pt[2];
  float cx = box3d[0], cy = box3d[1], cz = box3d[2];
  float dx = box3d[3], dy = box3d[4], dz = box3d[5], rz = box3d[6];
  if (fabsf(z - cz) > dz / 2.0) return 0;
  float cosa = cosf(-rz), sina = sinf(-rz);
  float sx = x - cx, sy = y - cy;
  *lx = sx * cosa + sy * (-sina);
  *ly = sx * sina + sy * cosa;
  /* `const float MARGIN = 1e-5; ... dx / 2.0 + MARGIN`: double + (float->double) */
  float m = (float)margin;
  return (fabsf(*lx) < dx / 2.0 + m) & (fabsf(*ly) < dy / 2.0 + m);
}

/* points_in_boxes_cpu, roiaware_pool3d.cpp:143-168: (N boxes, P points) 0/1, MARGIN 1e-2 */
ORC_API void orc_points_in_boxes_cpu(const float* boxes, int N, const float* pts, int P,
                                     int32_t* out) {
  float lx, ly;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < P; ++j) out[(size_t)i * P + j] = pt_in_box3d(pts + j * 3, boxes + i * 7, 1e-2, &lx, &ly);
}

/* points_in_boxes_kernel, roiaware_pool3d_kernel.cu:313-336: first containing box or -1 */
ORC_API void orc_points_in_boxes_gpu(const float* boxes, int B, int T, const float* pts, int P,
                                     int32_t* out) {
  float lx, ly;
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < P; ++j) {
      int r = -1;
      for (int k = 0; k < T; ++k)
        if (pt_in_box3d(pts + ((size_t)b * P + j) * 3, boxes + ((size_t)b * T + k) * 7, 1e-5, &lx, &ly)) {
          r = k;
          break;
        }
      out[(size_t)b * P + j] = r;
    }
}

/* RoI-aware pooling forward, roiaware_pool3d_kernel.cu:39-190 + launcher :194-228.
 * pts_idx_of_voxels (N,ox,oy,oz,maxpts) slot 0 = count; argmax (N,ox,oy,oz,C);
 * pooled (N,ox,oy,oz,C); all three must be zero-filled by the caller (utils.py:80-82). */
ORC_API void orc_roiaware_pool3d_forward(const float* rois, int N, const float* pts, int P,
                                         const float* feat, int C, int ox, int oy, int oz,
                                         int maxpts, int method, int32_t* argmax,
                                         int32_t* pts_idx, float* pooled) {
  for (int b = 0; b < N; ++b) {
    const float* roi = rois + (size_t)b * 7;
    int32_t* pv = pts_idx + (size_t)b * ox * oy * oz * maxpts;
    for (int k = 0; k < P; ++k) { /* generate_pts_mask + collect_inside_pts, serial per box */
      float lx = 0, ly = 0;
      if (!pt_in_box3d(pts + (size_t)k * 3, roi, 1e-5, &lx, &ly)) continue;
      float lz = pts[(size_t)k * 3 + 2] - roi[2];
      float dx = roi[3], dy = roi[4], dz = roi[5];
      float x_res = dx / ox, y_res = dy / oy, z_res = dz / oz;
      unsigned xi = (unsigned)(int)((lx + dx / 2) / x_res);
      unsigned yi = (unsigned)(int)((ly + dy / 2) / y_res);
      unsigned zi = (unsigned)(int)((lz + dz / 2) / z_res);
      /* min(max(x_idx, 0), out_x - 1) on unsigned (kernel.cu:69-71) */
      xi = xi < (unsigned)(ox - 1) ? xi : (unsigned)(ox - 1);
      yi = yi < (unsigned)(oy - 1) ? yi : (unsigned)(oy - 1);
      zi = zi < (unsigned)(oz - 1) ? zi : (unsigned)(oz - 1);
      /* encode/decode through 8-bit fields (kernel.cu:73,92-94) */
      xi &= 0xFF; yi &= 0xFF; zi &= 0xFF;
      size_t base = ((size_t)xi * oy * oz + (size_t)yi * oz + zi) * maxpts;
      unsigned cnt = (unsigned)pv[base];
      if (cnt < (unsigned)(maxpts - 1)) {
        pv[base + cnt + 1] = k;
        pv[base]++;
      }
    }
    for (int v = 0; v < ox * oy * oz; ++v) {
      const int32_t* lst = pv + (size_t)v * maxpts;
      int total = lst[0];
      for (int c = 0; c < C; ++c) {
        size_t o = ((size_t)b * ox * oy * oz + v) * C + c;
        if (method == 0) { /* roiaware_maxpool3d, kernel.cu:111-157 */
          int am = -1;
          float mv = -INFINITY; /* kernel.cu:137 `float max_val = -1e50` narrows to -inf */
          for (int k = 1; k <= total; ++k) {
            float fv = feat[(size_t)lst[k] * C + c];
            if (fv > mv) { mv = fv; am = lst[k]; }
          }
          if (am != -1) pooled[o] = mv;
          argmax[o] = am;
        } else { /* roiaware_avgpool3d, kernel.cu:160-190 */
          float s = 0;
          for (int k = 1; k <= total; ++k) s += feat[(size_t)lst[k] * C + c];
          if (total > 0) pooled[o] = s / total;
        }
      }
    }
  }
}

/* RoI-aware pooling backward, roiaware_pool3d_kernel.cu:236-286; grad_in zero-filled by caller */
ORC_API void orc_roiaware_pool3d_backward(const int32_t* pts_idx, const int32_t* argmax,
                                          const float* grad_out, int N, int ox, int oy, int oz,
                                          int C, int maxpts, int method, float* grad_in) {
  size_t nvox = (size_t)N * ox * oy * oz;
  for (size_t v = 0; v < nvox; ++v)
    for (int c = 0; c < C; ++c) {
      float g = grad_out[v * C + c];
      if (method == 0) {
        int a = argmax[v * C + c];
        if (a == -1) continue;
        grad_in[(size_t)a * C + c] += g * 1;
      } else {
        const int32_t* lst = pts_idx + v * maxpts;
        int total = lst[0];
        float cur = 1 / fmaxf((float)total, 1.0f);
        for (int k = 1; k <= total; ++k) grad_in[(size_t)lst[k] * C + c] += g * cur;
      }
    }
}

/* RoI point pooling, roipoint_pool3d_kernel.cu:38-134 (boxes already enlarged by the
 * wrapper, roipoint_pool3d_utils.py:51).  pooled (B,M,S,3+C) and empty_flag (B,M) zeroed
 * by the caller. */
ORC_API void orc_roipoint_pool3d(const float* xyz, const float* boxes, const float* feat, int B,
                                 int Np, int M, int C, int S, float* pooled,
                                 int32_t* empty_flag) {
  int32_t* idx = (int32_t*)malloc((size_t)S * sizeof(int32_t));
  float lx, ly;
  for (int b = 0; b < B; ++b)
    for (int m = 0; m < M; ++m) {
      const float* box = boxes + ((size_t)b * M + m) * 7;
      int cnt = 0;
      for (int k = 0; k < Np && cnt < S; ++k)
        if (pt_in_box3d(xyz + ((size_t)b * Np + k) * 3, box, 1e-5, &lx, &ly)) idx[cnt++] = k;
      if (cnt == 0) {
        empty_flag[(size_t)b * M + m] = 1;
        continue;
      }
      for (int k = cnt; k < S; ++k) idx[k] = idx[k % cnt];
      for (int s = 0; s < S; ++s) {
        float* dst = pooled + (((size_t)b * M + m) * S + s) * (3 + C);
        const float* p = xyz + ((size_t)b * Np + idx[s]) * 3;
        dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2];
        memcpy(dst + 3, feat + ((size_t)b * Np + idx[s]) * C, C * sizeof(float));
      }
    }
  free(idx);
}

/* voxel_query_kernel_stack, pointnet2_stack/src/voxel_query_gpu.cu:10-89.
 * idx (M, nsample) must arrive zero-filled (voxel_query_utils.py:33); returns raw kernel
 * output (idx[0] = -1 marks an empty ball). */
ORC_API void orc_voxel_query(int M, int R1, int R2, int R3, int nsample, float radius,
                             int z_range, int y_range, int x_range, const float* new_xyz,
                             const float* xyz, const int32_t* new_coords,
                             const int32_t* point_indices, int32_t* idx) {
  float radius2 = radius * radius;
  ORC_PAR_FOR
  for (int pt = 0; pt < M; ++pt) {
    const float* q = new_xyz + (size_t)pt * 3;
    const int32_t* nc = new_coords + (size_t)pt * 4;
    int32_t* o = idx + (size_t)pt * nsample;
    int cnt = 0;
    for (int dz = -z_range; dz <= z_range; ++dz) {
      int z = nc[1] + dz;
      if (z < 0 || z >= R1) continue;
      for (int dy = -y_range; dy <= y_range; ++dy) {
        int y = nc[2] + dy;
        if (y < 0 || y >= R2) continue;
        for (int dx = -x_range; dx <= x_range; ++dx) {
          int x = nc[3] + dx;
          if (x < 0 || x >= R3) continue;
          size_t index = (size_t)nc[0] * R1 * R2 * R3 + (size_t)z * R2 * R3 + (size_t)y * R3 + x;
          int nb = point_indices[index];
          if (nb < 0) continue;
          float xp = xyz[(size_t)nb * 3 + 0], yp = xyz[(size_t)nb * 3 + 1], zp = xyz[(size_t)nb * 3 + 2];
          float d2 = (xp - q[0]) * (xp - q[0]) + (yp - q[1]) * (yp - q[1]) + (zp - q[2]) * (zp - q[2]);
          if (d2 > radius2) continue;
          if (cnt < nsample) {
            if (cnt == 0)
              for (int l = 0; l < nsample; ++l) o[l] = nb;
            o[cnt] = nb;
            ++cnt;
          }
        }
      }
    }
    if (cnt == 0) o[0] = -1;
  }
}

/* ball_query_kernel_stack, pointnet2_stack/src/ball_query_gpu.cu:16-65 (strict d2 < r2,
 * indices local to the query's batch). */
ORC_API void orc_ball_query(int B, int M, float radius, int nsample, const float* new_xyz,
                            const int32_t* new_xyz_batch_cnt, const float* xyz,
                            const int32_t* xyz_batch_cnt, int32_t* idx) {
  float radius2 = radius * radius;
  for (int pt = 0; pt < M; ++pt) {
    int bs = 0, pc = new_xyz_batch_cnt[0];
    for (int k = 1; k < B; k++) {
      if (pt < pc) break;
      pc += new_xyz_batch_cnt[k];
      bs = k;
    }
    size_t start = 0;
    for (int k = 0; k < bs; k++) start += xyz_batch_cnt[k];
    const float* q = new_xyz + (size_t)pt * 3;
    const float* X = xyz + start * 3;
    int32_t* o = idx + (size_t)pt * nsample;
    int n = xyz_batch_cnt[bs], cnt = 0;
    for (int k = 0; k < n; ++k) {
      float x = X[k * 3], y = X[k * 3 + 1], z = X[k * 3 + 2];
      float d2 = (q[0] - x) * (q[0] - x) + (q[1] - y) * (q[1] - y) + (q[2] - z) * (q[2] - z);
      if (d2 < radius2) {
        if (cnt == 0)
          for (int l = 0; l < nsample; ++l) o[l] = k;
        o[cnt] = k;
        ++cnt;
        if (cnt >= nsample) break;
      }
    }
    if (cnt == 0) o[0] = -1;
  }
}

/* group_points_kernel_stack, pointnet2_stack/src/group_points_gpu.cu:71-101 */
ORC_API void orc_group_points(int B, int M, int C, int nsample, const float* features,
                              const int32_t* features_batch_cnt, const int32_t* idx,
                              const int32_t* idx_batch_cnt, float* out) {
  ORC_PAR_FOR
  for (int pt = 0; pt < M; ++pt) {
    int bs = 0, pc = idx_batch_cnt[0];
    for (int k = 1; k < B; k++) {
      if (pt < pc) break;
      pc += idx_batch_cnt[k];
      bs = k;
    }
    size_t start = 0;
    for (int k = 0; k < bs; k++) start += features_batch_cnt[k];
    for (int c = 0; c < C; ++c)
      for (int s = 0; s < nsample; ++s)
        out[((size_t)pt * C + c) * nsample + s] =
            features[(start + idx[(size_t)pt * nsample + s]) * C + c];
  }
}

/* group_points_grad_kernel_stack, group_points_gpu.cu:15-44; grad_features zeroed by caller */
ORC_API void orc_group_points_grad(int B, int M, int C, int N, int nsample, const float* grad_out,
                                   const int32_t* idx, const int32_t* idx_batch_cnt,
                                   const int32_t* features_batch_cnt, float* grad_features) {
  (void)N;
  for (int pt = 0; pt < M; ++pt) {
    int bs = 0, pc = idx_batch_cnt[0];
    for (int k = 1; k < B; k++) {
      if (pt < pc) break;
      pc += idx_batch_cnt[k];
      bs = k;
    }
    size_t start = 0;
    for (int k = 0; k < bs; k++) start += features_batch_cnt[k];
    for (int c = 0; c < C; ++c)
      for (int s = 0; s < nsample; ++s)
        grad_features[(start + idx[(size_t)pt * nsample + s]) * C + c] +=
            grad_out[((size_t)pt * C + c) * nsample + s];
  }
}

/* ------------------------------------------------------------------------------------------
 * PV-RCNN set-abstraction operators.
 * ------------------------------------------------------------------------------------------ */

/* stack_farthest_point_sampling_kernel<1024>, pointnet2_stack/src/sampling_gpu.cu:187-302.
 * Frame b: sample 0 is its first point; then m-1 times: temp[k] = min(temp[k], d(k, last)),
 * next = argmax temp.  Tie rule of the 1024-thread reduction restated: thread t scans
 * k = t, t+1024, ... with a strict '>' (first maximum of its sequence), the tree keeps the LOWER
 * thread on equal values (__update :16-21) -> among equal maxima the winner has the smallest
 * k mod 1024, then the smallest k.  temp arrives filled with 1e10 (pointnet2_utils.py:214). */
ORC_API void orc_stack_fps(int B, const float* xyz, const int32_t* xyz_batch_cnt, float* temp,
                           const int32_t* num_sampled, int32_t* idxs) {
  size_t start = 0, ostart = 0;
  for (int b = 0; b < B; ++b) {
    const float* X = xyz + start * 3;
    float* T = temp + start;
    int32_t* O = idxs + ostart;
    int n = xyz_batch_cnt[b], m = num_sampled[b];
    int old = 0;
    if (m > 0) O[0] = (int32_t)start;
    for (int j = 1; j < m; ++j) {
      float x1 = X[old * 3], y1 = X[old * 3 + 1], z1 = X[old * 3 + 2];
      float best = -1.f;
      int besti = 0, bestt = 1 << 30;
      for (int t = 0; t < 1024 && t < n; ++t) {
        float tb = -1.f;
        int ti = 0;
        for (int k = t; k < n; k += 1024) {
          float x2 = X[k * 3], y2 = X[k * 3 + 1], z2 = X[k * 3 + 2];
          float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
          float d2 = d < T[k] ? d : T[k];
          T[k] = d2;
          if (d2 > tb) { tb = d2; ti = k; }
        }
        if (tb > best || (tb == best && t < bestt)) { best = tb; besti = ti; bestt = t; }
      }
      old = besti;
      O[j] = (int32_t)(old + start);
    }
    start += n;
    ostart += m;
  }
}

/* three_nn_kernel_stack, pointnet2_stack/src/interpolate_gpu.cu:16-76: squared distances and
 * GLOBAL indices of the three nearest known points of the same frame (strict '<', ascending k). */
ORC_API void orc_three_nn(int B, int N, const float* unknown, const int32_t* unknown_batch_cnt,
                          const float* known, const int32_t* known_batch_cnt, float* dist2,
                          int32_t* idx) {
  for (int pt = 0; pt < N; ++pt) {
    int bs = 0, pc = unknown_batch_cnt[0];
    for (int k = 1; k < B; k++) {
      if (pt < pc) break;
      pc += unknown_batch_cnt[k];
      bs = k;
    }
    int start = 0;
    for (int k = 0; k < bs; k++) start += known_batch_cnt[k];
    const float* Kp = known + (size_t)start * 3;
    int n = known_batch_cnt[bs];
    float ux = unknown[pt * 3], uy = unknown[pt * 3 + 1], uz = unknown[pt * 3 + 2];
    double b1 = 1e40, b2 = 1e40, b3 = 1e40;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k = 0; k < n; ++k) {
      float x = Kp[k * 3], y = Kp[k * 3 + 1], z = Kp[k * 3 + 2];
      float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
      if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
      else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
      else if (d < b3) { b3 = d; i3 = k; }
    }
    dist2[pt * 3] = (float)b1; dist2[pt * 3 + 1] = (float)b2; dist2[pt * 3 + 2] = (float)b3;
    idx[pt * 3] = i1 + start; idx[pt * 3 + 1] = i2 + start; idx[pt * 3 + 2] = i3 + start;
  }
}

/* The host C library's float routines, elementwise (fn 0 sinf, 1 cosf, 2 atanf, 3 atan2f(x, y)): what
 * iou3d_cpu.cpp's cos / sin / atan2 calls resolve to.  Checker for the device restatement in csrc/glx_libm.h. */
ORC_API void orc_libm_eval(int fn, const float* x, const float* y, long long n, float* out) {
  ORC_PAR_FOR
  for (long long i = 0; i < n; ++i)
    out[i] = fn == 0 ? sinf(x[i]) : fn == 1 ? cosf(x[i]) : fn == 2 ? atanf(x[i]) : atan2f(x[i], y[i]);
}

/* ------------------------------------------------------------------------------------------
 * Batch-layout PointNet++ operators (pcdet/ops/pointnet2/pointnet2_batch/src/): B equal frames, indices
 * local to the frame.  GPU-only CUDA upstream, no reference test: parity unpinned, restated line by line.
 * ------------------------------------------------------------------------------------------ */

/* ball_query_kernel_fast, pointnet2_batch/src/ball_query_gpu.cu:15-51: idx (B, m, nsample) arrives as the
 * caller filled it (zeros, pointnet2_utils.py:236) and a ball without hits leaves its row untouched. */
ORC_API void orc_batch_ball_query(int B, int n, int m, float radius, int nsample, const float* new_xyz,
                                  const float* xyz, int32_t* idx) {
  float radius2 = radius * radius;
  for (int b = 0; b < B; ++b)
    for (int pt = 0; pt < m; ++pt) {
      const float* q = new_xyz + ((size_t)b * m + pt) * 3;
      const float* X = xyz + (size_t)b * n * 3;
      int32_t* o = idx + ((size_t)b * m + pt) * nsample;
      int cnt = 0;
      for (int k = 0; k < n; ++k) {
        float x = X[k * 3], y = X[k * 3 + 1], z = X[k * 3 + 2];
        float d2 = (q[0] - x) * (q[0] - x) + (q[1] - y) * (q[1] - y) + (q[2] - z) * (q[2] - z);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[l] = k;
          o[cnt] = k;
          ++cnt;
          if (cnt >= nsample) break;
        }
      }
    }
}

/* farthest_point_sampling_kernel<block_size>, pointnet2_batch/src/sampling_gpu.cu:97-230 with
 * block_size = opt_n_threads(n) (cuda_utils.h:9-13: 2^floor(log n / log 2) clamped to [1, 1024]).  Thread t
 * scans k = t, t + block_size, ... with a strict '>' (first maximum of its sequence), the tree keeps the
 * LOWER thread on equal values (__update :82-87): among equal maxima the winner has the smallest
 * k mod block_size, then the smallest k.  idxs (B, m) are LOCAL; temp (B, n) arrives filled with 1e10. */
ORC_API void orc_batch_fps(int B, int n, int m, const float* xyz, float* temp, int32_t* idxs) {
  int pow_2 = (int)(log((double)n) / log(2.0));
  int bs = 1 << pow_2;
  if (bs > 1024) bs = 1024;
  if (bs < 1) bs = 1;
  for (int b = 0; b < B; ++b) {
    const float* X = xyz + (size_t)b * n * 3;
    float* T = temp + (size_t)b * n;
    int32_t* O = idxs + (size_t)b * m;
    int old = 0;
    if (m > 0) O[0] = 0;
    for (int j = 1; j < m; ++j) {
      float x1 = X[old * 3], y1 = X[old * 3 + 1], z1 = X[old * 3 + 2];
      float best = -1.f;
      int besti = 0, bestt = 1 << 30;
      for (int t = 0; t < bs; ++t) {
        float tb = -1.f;
        int ti = 0;
        for (int k = t; k < n; k += bs) {
          float x2 = X[k * 3], y2 = X[k * 3 + 1], z2 = X[k * 3 + 2];
          float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
          float d2 = d < T[k] ? d : T[k];
          T[k] = d2;
          if (d2 > tb) { tb = d2; ti = k; }
        }
        if (tb > best || (tb == best && t < bestt)) { best = tb; besti = ti; bestt = t; }
      }
      old = besti;
      O[j] = old;
    }
  }
}

/* three_nn_kernel_fast, pointnet2_batch/src/interpolate_gpu.cu:15-60: running minima in double starting at
 * 1e40, strict '<', ascending k; outputs narrowed to float (1e40 -> +inf when m < 3). */
ORC_API void orc_batch_three_nn(int B, int n, int m, const float* unknown, const float* known, float* dist2,
                                int32_t* idx) {
  for (int b = 0; b < B; ++b)
    for (int pt = 0; pt < n; ++pt) {
      const float* u = unknown + ((size_t)b * n + pt) * 3;
      const float* Kp = known + (size_t)b * m * 3;
      float ux = u[0], uy = u[1], uz = u[2];
      double b1 = 1e40, b2 = 1e40, b3 = 1e40;
      int i1 = 0, i2 = 0, i3 = 0;
      for (int k = 0; k < m; ++k) {
        float x = Kp[k * 3], y = Kp[k * 3 + 1], z = Kp[k * 3 + 2];
        float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
        if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
        else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
        else if (d < b3) { b3 = d; i3 = k; }
      }
      float* dp = dist2 + ((size_t)b * n + pt) * 3;
      int32_t* ip = idx + ((size_t)b * n + pt) * 3;
      dp[0] = (float)b1; dp[1] = (float)b2; dp[2] = (float)b3;
      ip[0] = i1; ip[1] = i2; ip[2] = i3;
    }
}

/* three_interpolate_kernel_stack / _grad_, interpolate_gpu.cu:100-160 */
ORC_API void orc_three_interpolate(int N, int C, const float* features, const int32_t* idx,
                                   const float* weight, float* out) {
  for (int p = 0; p < N; ++p)
    for (int c = 0; c < C; ++c)
      out[(size_t)p * C + c] = weight[p * 3] * features[(size_t)idx[p * 3] * C + c] +
                               weight[p * 3 + 1] * features[(size_t)idx[p * 3 + 1] * C + c] +
                               weight[p * 3 + 2] * features[(size_t)idx[p * 3 + 2] * C + c];
}

ORC_API void orc_three_interpolate_grad(int N, int C, const float* grad_out, const int32_t* idx,
                                        const float* weight, float* grad_features) {
  for (int p = 0; p < N; ++p)
    for (int c = 0; c < C; ++c)
      for (int j = 0; j < 3; ++j)
        grad_features[(size_t)idx[p * 3 + j] * C + c] += grad_out[(size_t)p * C + c] * weight[p * 3 + j];
}

/* ------------------------------------------------------------------------------------------
 * VectorPool family of PV-RCNN++: pointnet2_stack/src/vector_pool_gpu.cu.
 * The CUDA kernels hand out buffer slots with atomicAdd, so the ORDER of the per-point segments
 * (start offsets, rows of grouped_idxs) is implementation-defined upstream; this restatement -- and the
 * HIP build -- define it as ascending new-point index (one of the orders the reference can produce).
 * Everything else (which neighbours, their order inside a point's segment, pooled sums) is fixed by the
 * sequential scan over k of each kernel thread.
 * ------------------------------------------------------------------------------------------ */

static int orc_frame_of(int pt, const int32_t* cnt, int B) {
  int bs = 0, pc = cnt[0];
  for (int k = 1; k < B; k++) {
    if (pt < pc) break;
    pc += cnt[k];
    bs = k;
  }
  return bs;
}

static int orc_in_range(float lx, float ly, float lz, float dist, int neighbor_type) {
  if (neighbor_type == 1) return !(lx * lx + ly * ly + lz * lz > dist * dist);
  return !((fabs(lx) > dist) | (fabs(ly) > dist) | (fabs(lz) > dist));
}

/* query_stacked_local_neighbor_idxs_kernel, vector_pool_gpu.cu:122-200.  Returns the total (cumsum).
 * A point keeps at most 1000 neighbours (the kernel's temp_idxs) and stops at nsample when nsample > 0;
 * segments beyond avg_length * M are dropped / truncated exactly as lines 191-199 do. */
ORC_API int orc_query_stacked_local_neighbor_idxs(const float* support_xyz, const int32_t* xyz_batch_cnt,
                                                  const float* new_xyz, const int32_t* new_xyz_batch_cnt,
                                                  int B, int M, int32_t* stack_neighbor_idxs, int32_t* start_len,
                                                  int avg_length, float max_dist, int nsample, int neighbor_type) {
  int cumsum = 0;
  int max_thresh = avg_length * M;
  int* temp = (int*)malloc(1000 * sizeof(int));
  for (int pt = 0; pt < M; ++pt) {
    int bs = orc_frame_of(pt, new_xyz_batch_cnt, B);
    int start = 0;
    for (int k = 0; k < bs; k++) start += xyz_batch_cnt[k];
    const float* X = support_xyz + (size_t)start * 3;
    const float* q = new_xyz + (size_t)pt * 3;
    int n = xyz_batch_cnt[bs], cnt = 0;
    for (int k = 0; k < n; ++k) {
      float lx = X[k * 3] - q[0], ly = X[k * 3 + 1] - q[1], lz = X[k * 3 + 2] - q[2];
      if (!orc_in_range(lx, ly, lz, max_dist, neighbor_type)) continue;
      if (cnt < 1000) temp[cnt] = k; else break;
      cnt++;
      if (nsample > 0 && cnt >= nsample) break;
    }
    start_len[pt * 2] = cumsum;
    start_len[pt * 2 + 1] = cnt;
    int s0 = cumsum;
    cumsum += cnt;
    if (s0 >= max_thresh) continue;
    int w = cnt;
    if (s0 + w >= max_thresh) w = max_thresh - s0;
    for (int k = 0; k < w; ++k) stack_neighbor_idxs[s0 + k] = temp[k] + start;
  }
  free(temp);
  return cumsum;
}

/* query_three_nn_by_stacked_local_idxs_kernel, vector_pool_gpu.cu:19-85 */
ORC_API void orc_query_three_nn_by_stacked_local_idxs(const float* support_xyz, const float* grid_centers,
                                                      int32_t* grid_idxs, float* grid_dist2,
                                                      const int32_t* stack_neighbor_idxs, const int32_t* start_len,
                                                      int M, int num_total_grids) {
  for (int pt = 0; pt < M; ++pt)
    for (int g = 0; g < num_total_grids; ++g) {
      const float* c = grid_centers + ((size_t)pt * num_total_grids + g) * 3;
      const int32_t* nb = stack_neighbor_idxs + start_len[pt * 2];
      int len = start_len[pt * 2 + 1];
      double b1 = 1e40, b2 = 1e40, b3 = 1e40;
      int i1 = -1, i2 = -1, i3 = -1;
      for (int k = 0; k < len; ++k) {
        int j = nb[k];
        float x = support_xyz[(size_t)j * 3], y = support_xyz[(size_t)j * 3 + 1], z = support_xyz[(size_t)j * 3 + 2];
        float d = (c[0] - x) * (c[0] - x) + (c[1] - y) * (c[1] - y) + (c[2] - z) * (c[2] - z);
        if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = j; }
        else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = j; }
        else if (d < b3) { b3 = d; i3 = j; }
      }
      if (i2 == -1) { i2 = i1; b2 = b1; }
      if (i3 == -1) { i3 = i1; b3 = b1; }
      size_t o = ((size_t)pt * num_total_grids + g) * 3;
      grid_dist2[o] = (float)b1; grid_dist2[o + 1] = (float)b2; grid_dist2[o + 2] = (float)b3;
      grid_idxs[o] = i1; grid_idxs[o + 1] = i2; grid_idxs[o + 2] = i3;
    }
}

/* vector_pool_kernel_stack, vector_pool_gpu.cu:243-375.  Outputs arrive zero-filled; returns cum_sum.
 * Rows of grouped_idxs beyond num_max_sum_points are counted, not stored (and, as in the kernel, such a
 * hit does not advance sample_cnt). */
ORC_API int orc_vector_pool(const float* support_xyz, const float* support_features, const int32_t* xyz_batch_cnt,
                            const float* new_xyz, const int32_t* new_xyz_batch_cnt, int B, int M, int num_c_in,
                            int num_c_out, int gx, int gy, int gz, float max_dist, int use_xyz,
                            int num_max_sum_points, int nsample, int neighbor_type, int pooling_type,
                            float* new_features, float* new_local_xyz, int32_t* point_cnt_of_grid,
                            int32_t* grouped_idxs) {
  int G = gx * gy * gz, cg = num_c_out / G;
  float sx = max_dist * 2 / gx, sy = max_dist * 2 / gy, sz = max_dist * 2 / gz;
  int cum = 0;
  for (int pt = 0; pt < M; ++pt) {
    int bs = orc_frame_of(pt, new_xyz_batch_cnt, B);
    int start = 0;
    for (int k = 0; k < bs; k++) start += xyz_batch_cnt[k];
    const float* X = support_xyz + (size_t)start * 3;
    const float* F = support_features + (size_t)start * num_c_in;
    const float* q = new_xyz + (size_t)pt * 3;
    float* nf = new_features + (size_t)pt * num_c_out;
    float* nl = new_local_xyz + (size_t)pt * 3 * G;
    int32_t* pc = point_cnt_of_grid + (size_t)pt * G;
    int n = xyz_batch_cnt[bs], sample_cnt = 0;
    for (int k = 0; k < n; ++k) {
      float lx = X[k * 3] - q[0], ly = X[k * 3 + 1] - q[1], lz = X[k * 3 + 2] - q[2];
      if (!orc_in_range(lx, ly, lz, max_dist, neighbor_type)) continue;
      int ix = (int)floorf((lx + max_dist) / sx), iy = (int)floorf((ly + max_dist) / sy);
      int iz = (int)floorf((lz + max_dist) / sz);
      int g = ix * gy * gz + iy * gz + iz;
      g = g < 0 ? 0 : (g > G - 1 ? G - 1 : g);
      if (pooling_type == 0) {
        pc[g]++;
        for (int i = 0; i < num_c_in; ++i) nf[g * cg + i % cg] += F[(size_t)k * num_c_in + i];
        if (use_xyz) { nl[g * 3] += lx; nl[g * 3 + 1] += ly; nl[g * 3 + 2] += lz; }
      } else {
        if (pc[g] != 0) continue;
        pc[g]++;
        for (int i = 0; i < num_c_in; ++i) nf[g * cg + i % cg] = F[(size_t)k * num_c_in + i];
        if (use_xyz) { nl[g * 3] = lx; nl[g * 3 + 1] = ly; nl[g * 3 + 2] = lz; }
      }
      int cnt = cum++;
      if (cnt >= num_max_sum_points) continue;
      grouped_idxs[cnt * 3] = start + k;
      grouped_idxs[cnt * 3 + 1] = pt;
      grouped_idxs[cnt * 3 + 2] = g;
      sample_cnt++;
      if (pooling_type == 0) { if (nsample > 0 && sample_cnt >= nsample) break; }
      else if ((nsample > 0 && sample_cnt >= nsample) || sample_cnt >= G) break;
    }
  }
  return cum;
}

/* vector_pool_grad_kernel_stack, vector_pool_gpu.cu:433-460; grad_support_features zero-filled by the caller */
ORC_API void orc_vector_pool_grad(const float* grad_new_features, const int32_t* point_cnt_of_grid,
                                  const int32_t* grouped_idxs, int num_idxs, int num_c_in, int num_c_out,
                                  int num_total_grids, float* grad_support_features) {
  int cg = num_c_out / num_total_grids;
  for (int e = 0; e < num_idxs; ++e) {
    int s = grouped_idxs[e * 3], p = grouped_idxs[e * 3 + 1], g = grouped_idxs[e * 3 + 2];
    int tot = point_cnt_of_grid[(size_t)p * num_total_grids + g];
    float w = 1 / fmaxf((float)tot, 1.0);
    for (int c = 0; c < num_c_in; ++c)
      grad_support_features[(size_t)s * num_c_in + c] += grad_new_features[(size_t)p * num_c_out + g * cg + c % cg] * w;
  }
}
