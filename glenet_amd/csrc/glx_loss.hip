// GLENet's KL regression loss of the RoI head (VoxelRCNNKLLabelIoUHead.get_box_reg_layer_loss,
// pcdet/models/roi_heads/voxelrcnn_kl_label_iou_head.py:93-138; "ad hoc, in fact it's kl loss"):
//   t     = ResidualCoder.encode_torch(gt in the RoI frame, RoI moved to the origin with heading 0)
//           (box_coder_utils.py:13-43: sizes clamped at 1e-5, xy by the BEV diagonal, z by the height,
//            log size ratios, heading difference)
//   src   = smooth_l1((reg - t) * code_weight, beta)            (loss_utils.py:100-130; NaN targets ignored)
//   s     = max(reg_std, -50),  lv = log(label_variance + 1e-10)
//   loss  = sum_fg [ exp(-s) * src + exp(lv - s) - 0.5 * (lv - s) ] / max(#fg, 1) * weight
// The reference evaluates this with ~40 tensor kernels and three host read-backs (`.item()`) per
// step; here one block computes the loss, its three reported parts and both gradients (d/dreg,
// d/dreg_std) in two passes over the (R, 7) elements, sums in fp64 in a fixed tree order.
#include "glx_common.h"
#include "glx_fill.h"

#define KL_THREADS 256

template <int NT>
__device__ __forceinline__ double blk_sum(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int s = NT / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ double kl_block_sum(double v, double* red) { return blk_sum<KL_THREADS>(v, red); }

struct KlCodeWeights { float w[7]; };

// One block of NT threads; the bodies below are shared by the stand-alone kernels (NT = KL_THREADS, float masks, dense
// 7-column rows) and by k_roi_head_losses (all three RoI-head terms in one launch, int64 mask, strided ground-truth rows).
// `fg`: > 0 = foreground (float or int64).  Returns the loss (every thread).
template <int NT, class M>
__device__ __forceinline__ float kl_reg_body(
    const float* __restrict__ reg, const float* __restrict__ reg_std, const float* __restrict__ rois,
    const float* __restrict__ gt, int gt_ld, const float* __restrict__ label_var, const M* __restrict__ fg,
    int R, const KlCodeWeights& cw, float beta, float weight, float* __restrict__ out,
    float* __restrict__ grad_reg, float* __restrict__ grad_std, double* red) {
  double c = 0;
  for (int i = threadIdx.x; i < R; i += NT) c += fg[i] > 0 ? 1.0 : 0.0;
  const double nfg = blk_sum<NT>(c, red);
  const float scale = weight / (float)(nfg > 1.0 ? nfg : 1.0);
  double s_src = 0, s_sq = 0, s_log = 0;
  for (int e = threadIdx.x; e < R * 7; e += NT) {
    const int i = e / 7, k = e - i * 7;
    const float* a = rois + (long long)i * 7;
    const float* g = gt + (long long)i * gt_ld;
    const float dxa = fmaxf(a[3], 1e-5f), dya = fmaxf(a[4], 1e-5f), dza = fmaxf(a[5], 1e-5f);
    float t;
    if (k < 2) t = g[k] / sqrtf(dxa * dxa + dya * dya);            // anchor centre is the origin
    else if (k == 2) t = g[2] / dza;
    else if (k < 6) t = logf(fmaxf(g[k], 1e-5f) / (k == 3 ? dxa : (k == 4 ? dya : dza)));
    else t = g[6];                                                  // anchor heading is 0
    const float x = reg[e];
    if (isnan(t)) t = x;
    const float diff = (x - t) * cw.w[k];
    const float n = fabsf(diff);
    const float src = beta < 1e-5f ? n : (n < beta ? 0.5f * n * n / beta : n - 0.5f * beta);
    const float dsrc = beta < 1e-5f ? (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f))
                                    : (n < beta ? diff / beta : (diff > 0.f ? 1.f : -1.f));
    const float sraw = reg_std[e];
    const bool clamped = sraw < -50.f;
    const float s = clamped ? -50.f : sraw;
    const float lv = logf(label_var[e] + 1e-10f);
    const float m = fg[i] > 0 ? 1.f : 0.f;
    const float es = expf(-s), sq = expf(lv - s);
    s_src += (double)(es * src * m);
    s_sq += (double)(sq * m);
    s_log += (double)(-0.5f * (lv - s) * m);
    if (grad_reg) grad_reg[e] = m * es * dsrc * cw.w[k] * scale;
    if (grad_std) grad_std[e] = clamped ? 0.f : m * (-es * src - sq + 0.5f) * scale;
  }
  const double a_src = blk_sum<NT>(s_src, red), a_sq = blk_sum<NT>(s_sq, red), a_log = blk_sum<NT>(s_log, red);
  const float o1 = (float)a_src * scale, o2 = (float)a_sq * scale, o3 = (float)a_log * scale;
  const float total = o1 + o2 + o3;
  if (threadIdx.x == 0) {
    out[1] = o1;
    out[2] = o2;
    out[3] = o3;
    out[0] = total;
    out[4] = (float)nfg;
  }
  return total;
}

__global__ __launch_bounds__(KL_THREADS) void k_kl_reg_loss(
    const float* __restrict__ reg, const float* __restrict__ reg_std, const float* __restrict__ rois,
    const float* __restrict__ gt, const float* __restrict__ label_var, const float* __restrict__ fg,
    int R, KlCodeWeights cw, float beta, float weight, float* __restrict__ out,
    float* __restrict__ grad_reg, float* __restrict__ grad_std) {
  __shared__ double red[KL_THREADS];
  kl_reg_body<KL_THREADS>(reg, reg_std, rois, gt, 7, label_var, fg, R, cw, beta, weight, out, grad_reg, grad_std, red);
}

extern "C" int glx_kl_reg_loss(const float* rcnn_reg, const float* rcnn_reg_std, const float* rois,
                               const float* gt_of_rois, const float* gt_uncertainty,
                               const float* fg_mask, int R, const float* code_weights, float beta,
                               float weight, float* out5, float* grad_reg, float* grad_std,
                               void* stream) {
  GLX_REQUIRE(out5, "glx_kl_reg_loss: null output");
  GLX_REQUIRE(R == 0 || (rcnn_reg && rcnn_reg_std && rois && gt_of_rois && gt_uncertainty && fg_mask),
              "glx_kl_reg_loss: null pointer");
  KlCodeWeights cw;
  for (int k = 0; k < 7; ++k) cw.w[k] = code_weights ? code_weights[k] : 1.f;
  hipLaunchKernelGGL(k_kl_reg_loss, dim3(1), dim3(KL_THREADS), 0, (hipStream_t)stream, rcnn_reg,
                     rcnn_reg_std, rois, gt_of_rois, gt_uncertainty, fg_mask, R, cw, beta, weight, out5,
                     grad_reg, grad_std);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ corner loss of the RoI head
// The CORNER_LOSS_REGULARIZATION tail of get_box_reg_layer_loss (voxelrcnn_kl_label_iou_head.py:148-172):
// foreground RoIs only -- decode the regression against the RoI at the origin (ResidualCoder.decode_torch,
// box_coder_utils.py:45-78), rotate the centre by the RoI heading and translate
// (common_utils.rotate_points_along_z), then get_corner_loss_lidar (loss_utils.py:210-233): 8 corners
// (box_utils.boxes_to_corners_3d template order), distance to the ground-truth corners and to those of
// the ground truth turned by pi, the smaller one through smooth-L1 (beta 1), mean over corners, mean
// over the foreground RoIs.  One thread per RoI computes the row loss and d loss / d rcnn_reg[row, 0:7]
// analytically (the reference: ~60 tensor kernels forward, as many backward).
__device__ __forceinline__ void corner_of(int j, float dx, float dy, float dz, float ch, float sh, float cx,
                                          float cy, float cz, float& x, float& y, float& z, float& lx,
                                          float& ly) {
  // template rows of boxes_to_corners_3d: (1,1,-1),(1,-1,-1),(-1,-1,-1),(-1,1,-1),(1,1,1),(1,-1,1),(-1,-1,1),(-1,1,1), halved
  const float tx = (j & 3) < 2 ? 0.5f : -0.5f;
  const float ty = ((j & 3) == 0 || (j & 3) == 3) ? 0.5f : -0.5f;
  const float tz = j < 4 ? -0.5f : 0.5f;
  lx = dx * tx;
  ly = dy * ty;
  x = lx * ch - ly * sh + cx;
  y = lx * sh + ly * ch + cy;
  z = dz * tz + cz;
}

// ACC: the row gradients are ADDED to what grad_reg holds (the KL term's, written by this block before a barrier).
template <int NT, bool ACC, class M>
__device__ __forceinline__ float corner_loss_body(
    const float* __restrict__ reg, const float* __restrict__ rois, const float* __restrict__ gt, int gt_ld,
    const M* __restrict__ fg, int R, float weight, float* __restrict__ out, float* __restrict__ grad_reg, double* red) {
  double c = 0;
  for (int i = threadIdx.x; i < R; i += NT) c += fg[i] > 0 ? 1.0 : 0.0;
  const double nfg = blk_sum<NT>(c, red);
  const float scale = nfg > 0.0 ? weight / (float)nfg : 0.f;
  double acc = 0;
  for (int i = threadIdx.x; i < R; i += NT) {
    float g7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (fg[i] > 0) {
      const float* a = rois + (long long)i * 7;
      const float* r = reg + (long long)i * 7;
      const float* q = gt + (long long)i * gt_ld;
      const float dxa = a[3], dya = a[4], dza = a[5], ra = a[6];
      const float diag = sqrtf(dxa * dxa + dya * dya);
      const float xl = r[0] * diag, yl = r[1] * diag, zl = r[2] * dza;
      const float dx = expf(r[3]) * dxa, dy = expf(r[4]) * dya, dz = expf(r[5]) * dza;
      const float h = r[6] + ra;
      const float ca = cosf(ra), sa = sinf(ra);
      const float cx = xl * ca - yl * sa + a[0], cy = xl * sa + yl * ca + a[1], cz = zl + a[2];
      const float ch = cosf(h), sh = sinf(h);
      const float gch = cosf(q[6]), gsh = sinf(q[6]);
      const float fch = cosf(q[6] + 3.14159265358979323846f), fsh = sinf(q[6] + 3.14159265358979323846f);
      float row = 0.f, gcx = 0.f, gcy = 0.f, gcz = 0.f, gdx = 0.f, gdy = 0.f, gdz = 0.f, gh = 0.f;
      for (int j = 0; j < 8; ++j) {
        float px, py, pz, lx, ly, qx, qy, qz, fx, fy, fz, t0, t1;
        corner_of(j, dx, dy, dz, ch, sh, cx, cy, cz, px, py, pz, lx, ly);
        corner_of(j, q[3], q[4], q[5], gch, gsh, q[0], q[1], q[2], qx, qy, qz, t0, t1);
        corner_of(j, q[3], q[4], q[5], fch, fsh, q[0], q[1], q[2], fx, fy, fz, t0, t1);
        const float d1 = sqrtf((px - qx) * (px - qx) + (py - qy) * (py - qy) + (pz - qz) * (pz - qz));
        const float d2 = sqrtf((px - fx) * (px - fx) + (py - fy) * (py - fy) + (pz - fz) * (pz - fz));
        const bool first = d1 <= d2;
        const float d = first ? d1 : d2;
        const float ex = px - (first ? qx : fx), ey = py - (first ? qy : fy), ez = pz - (first ? qz : fz);
        row += d < 1.f ? 0.5f * d * d : d - 0.5f;
        const float k = d < 1.f ? 1.f : 1.f / d;                // d smooth_l1 / d d, times 1/d of the norm
        const float Gx = k * ex, Gy = k * ey, Gz = k * ez;
        const float tx = (j & 3) < 2 ? 0.5f : -0.5f;
        const float ty = ((j & 3) == 0 || (j & 3) == 3) ? 0.5f : -0.5f;
        const float tz = j < 4 ? -0.5f : 0.5f;
        gcx += Gx; gcy += Gy; gcz += Gz;
        gdx += Gx * tx * ch + Gy * tx * sh;
        gdy += -Gx * ty * sh + Gy * ty * ch;
        gdz += Gz * tz;
        gh += Gx * (-lx * sh - ly * ch) + Gy * (lx * ch - ly * sh);
      }
      acc += (double)(row * 0.125f);
      const float s8 = scale * 0.125f;
      const float gxl = gcx * ca + gcy * sa, gyl = -gcx * sa + gcy * ca;
      g7[0] = gxl * diag * s8; g7[1] = gyl * diag * s8; g7[2] = gcz * dza * s8;
      g7[3] = gdx * dx * s8; g7[4] = gdy * dy * s8; g7[5] = gdz * dz * s8; g7[6] = gh * s8;
    }
    if (grad_reg) {
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        if constexpr (ACC) grad_reg[(long long)i * 7 + k] += g7[k];
        else grad_reg[(long long)i * 7 + k] = g7[k];
      }
    }
  }
  const double total = blk_sum<NT>(acc, red);
  const float loss = (float)total * scale;
  if (threadIdx.x == 0) {
    out[0] = loss;
    out[1] = (float)nfg;
  }
  return loss;
}

__global__ __launch_bounds__(KL_THREADS) void k_corner_loss(
    const float* __restrict__ reg, const float* __restrict__ rois, const float* __restrict__ gt,
    const float* __restrict__ fg, int R, float weight, float* __restrict__ out,
    float* __restrict__ grad_reg) {
  __shared__ double red[KL_THREADS];
  corner_loss_body<KL_THREADS, false>(reg, rois, gt, 7, fg, R, weight, out, grad_reg, red);
}

extern "C" int glx_corner_loss(const float* rcnn_reg, const float* rois, const float* gt_of_rois_src,
                               const float* fg_mask, int R, float weight, float* out2, float* grad_reg,
                               void* stream) {
  GLX_REQUIRE(out2, "glx_corner_loss: null output");
  GLX_REQUIRE(R == 0 || (rcnn_reg && rois && gt_of_rois_src && fg_mask), "glx_corner_loss: null pointer");
  hipLaunchKernelGGL(k_corner_loss, dim3(1), dim3(KL_THREADS), 0, (hipStream_t)stream, rcnn_reg, rois,
                     gt_of_rois_src, fg_mask, R, weight, out2, grad_reg);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ canonical transformation
// RoIHeadTemplate.assign_targets, lines 140-159 (pcdet/models/roi_heads/roi_head_template.py): the
// matched ground-truth box of every RoI expressed in the RoI's frame -- centre difference rotated by
// -roi_ry (common_utils.rotate_points_along_z), heading difference folded into [-pi/2, pi/2] (a
// ground truth pointing the other way round is turned by pi).  Extra channels (class id, ...) pass
// through.  ~25 tensor kernels in the reference, one launch here.
__device__ __forceinline__ float glx_pymod(float a, float m) {    // python / torch `%` for m > 0
  float r = fmodf(a, m);
  if (r != 0.f && r < 0.f) r += m;
  return r;
}

__global__ void k_roi_canonical_gt(const float* __restrict__ rois, int roi_cols,
                                   const float* __restrict__ gt, int gt_cols, int R,
                                   float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const float PI = 3.14159265358979323846f, TWO_PI = (float)(2.0 * 3.14159265358979323846);
  const float* a = rois + (long long)i * roi_cols;
  const float* g = gt + (long long)i * gt_cols;
  float* o = out + (long long)i * gt_cols;
  const float ry = glx_pymod(a[6], TWO_PI);
  const float x = g[0] - a[0], y = g[1] - a[1], z = g[2] - a[2];
  const float cs = cosf(-ry), sn = sinf(-ry);
  o[0] = x * cs - y * sn;
  o[1] = x * sn + y * cs;
  o[2] = z;
  o[3] = g[3]; o[4] = g[4]; o[5] = g[5];
  float h = glx_pymod(g[6] - ry, TWO_PI);
  if (h > (float)(3.14159265358979323846 * 0.5) && h < (float)(3.14159265358979323846 * 1.5)) h = glx_pymod(h + PI, TWO_PI);
  if (h > PI) h = h - TWO_PI;
  h = fminf(fmaxf(h, (float)(-3.14159265358979323846 / 2)), (float)(3.14159265358979323846 / 2));
  o[6] = h;
  for (int k = 7; k < gt_cols; ++k) o[k] = g[k];
}

extern "C" int glx_roi_canonical_gt(const float* rois, int roi_cols, const float* gt_of_rois, int gt_cols,
                                    int R, float* out, void* stream) {
  if (R <= 0) return GLX_OK;
  GLX_REQUIRE(rois && gt_of_rois && out && roi_cols >= 7 && gt_cols >= 7,
              "glx_roi_canonical_gt: need >= 7 columns and non-null pointers");
  hipLaunchKernelGGL(k_roi_canonical_gt, dim3(glx_divup(R, 256)), dim3(256), 0, (hipStream_t)stream, rois,
                     roi_cols, gt_of_rois, gt_cols, R, out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ anchor target assignment
// AxisAlignedTargetAssigner.assign_targets_single (pcdet/models/dense_heads/target_assigner/
// axis_aligned_target_assigner.py:133-213) for one anchor class, all frames of the batch, without
// sampling (POS_FRACTION < 0) and with the nearest-BEV IoU (match_height False):
//   iou(a, g) = box_utils.boxes3d_nearest_bev_iou (box_utils.py:249-298): both boxes snapped to the
//               nearer axis (|limit_period(heading, 0.5, pi)| < pi/4 keeps (dx,dy), else swaps them),
//               axis-aligned IoU with the union clamped at 1e-6;
//   an anchor is positive when its best IoU >= matched_threshold or when it attains some ground
//   truth's best IoU (ties included; ground truths whose best IoU is 0 match nothing); it is
//   background when its best IoU < unmatched_threshold and it is not such a forced match; the rest
//   is "don't care" (-1).  Positives get the residual encoding of their best ground truth
//   (ResidualCoder.encode_torch) and weight 1 / #(labels >= 0) (NORM_BY_NUM_EXAMPLES) or 1.
// The reference runs this per frame and class in ~70 tensor kernels over (N x M) matrices with
// `.nonzero()` / boolean-index host synchronisations; here: 3 launches per class for the whole batch.
struct AlignedBev { float x1, y1, x2, y2, area; };

__device__ __forceinline__ AlignedBev aligned_bev(const float* b) {
  const float PI = (float)3.14159265358979323846;
  const float rot = fabsf(b[6] - floorf(b[6] / PI + 0.5f) * PI);
  const bool keep = rot < (float)(3.14159265358979323846 / 4);
  const float dx = keep ? b[3] : b[4], dy = keep ? b[4] : b[3];
  AlignedBev r;
  r.x1 = b[0] - dx / 2; r.y1 = b[1] - dy / 2; r.x2 = b[0] + dx / 2; r.y2 = b[1] + dy / 2;
  r.area = (r.x2 - r.x1) * (r.y2 - r.y1);
  return r;
}

__device__ __forceinline__ float bev_iou(const AlignedBev& a, const AlignedBev& b) {
  const float xl = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.f);
  const float yl = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.f);
  const float inter = xl * yl;
  return inter / fmaxf(a.area + b.area - inter, 1e-6f);
}

#define TA_MAXGT 128

// per frame: rows of gt considered = up to the last row whose 8 values do not sum to 0 (at least
// row 0), as the reference trims its zero padding; also clears the per-frame scratch
__global__ void k_assign_prepare(const float* __restrict__ gt, int B, int M, int gt_cols,
                                 int* __restrict__ n_valid, int* __restrict__ gmax,
                                 int* __restrict__ num_examples) {
  const int b = blockIdx.x;
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    const float* g = gt + ((long long)b * M + j) * gt_cols;
    float s = 0.f;
    for (int c = 0; c < gt_cols; ++c) s += g[c];
    if (s != 0.f) atomicMax(&s_last, j);
    gmax[b * TA_MAXGT + (j < TA_MAXGT ? j : 0)] = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) { n_valid[b] = s_last + 1; num_examples[b] = 0; }
}

__global__ __launch_bounds__(256) void k_assign_iou_max(
    const float* __restrict__ anchors, int N, const float* __restrict__ gt, int M, int gt_cols, int cls,
    const int* __restrict__ n_valid, float* __restrict__ amax, int* __restrict__ aarg,
    int* __restrict__ gmax) {
  const int b = blockIdx.y;
  __shared__ AlignedBev s_g[TA_MAXGT];
  __shared__ int s_ok[TA_MAXGT], s_gmax[TA_MAXGT];
  const int nv = min(n_valid[b], M);
  for (int j = threadIdx.x; j < nv; j += blockDim.x) {
    const float* g = gt + ((long long)b * M + j) * gt_cols;
    s_g[j] = aligned_bev(g);
    s_ok[j] = (int)g[gt_cols - 1] == cls;
    s_gmax[j] = 0;
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    const AlignedBev a = aligned_bev(anchors + (long long)i * 7);
    float best = -1.f;
    int arg = -1;
    for (int j = 0; j < nv; ++j) {
      if (!s_ok[j]) continue;
      const float v = bev_iou(a, s_g[j]);
      if (v > best) { best = v; arg = j; }                 // first maximum, as torch.argmax
      if (v > 0.f) atomicMax(&s_gmax[j], __float_as_int(v));
    }
    amax[(long long)b * N + i] = best;
    aarg[(long long)b * N + i] = arg;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < nv; j += blockDim.x)
    if (s_gmax[j] > 0) atomicMax(&gmax[b * TA_MAXGT + j], s_gmax[j]);
}

__global__ __launch_bounds__(256) void k_assign_labels(
    const float* __restrict__ anchors, int N, const float* __restrict__ gt, int M, int gt_cols, int cls,
    float matched, float unmatched, const int* __restrict__ n_valid, const float* __restrict__ amax,
    const int* __restrict__ aarg, const int* __restrict__ gmax, int* __restrict__ labels,
    float* __restrict__ targets, int* __restrict__ num_examples, int* __restrict__ unc_gt) {
  const int b = blockIdx.y;
  __shared__ AlignedBev s_g[TA_MAXGT];
  __shared__ int s_ok[TA_MAXGT];
  __shared__ float s_gm[TA_MAXGT];
  __shared__ int s_cnt;
  const int nv = min(n_valid[b], M);
  if (threadIdx.x == 0) s_cnt = 0;
  for (int j = threadIdx.x; j < nv; j += blockDim.x) {
    const float* g = gt + ((long long)b * M + j) * gt_cols;
    s_g[j] = aligned_bev(g);
    s_ok[j] = (int)g[gt_cols - 1] == cls;
    const int gm = gmax[b * TA_MAXGT + j];
    s_gm[j] = gm == 0 ? -1.f : __int_as_float(gm);          // a ground truth nobody overlaps matches nothing
  }
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int lab = -2;
  if (i < N) {
    const float* an = anchors + (long long)i * 7;
    const float best = amax[(long long)b * N + i];
    const int arg = aarg[(long long)b * N + i];
    float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (arg < 0) {
      lab = 0;                                               // no ground truth of this class in the frame
    } else {
      const AlignedBev a = aligned_bev(an);
      bool forced = false;
      for (int j = 0; j < nv; ++j)
        if (s_ok[j] && bev_iou(a, s_g[j]) == s_gm[j]) forced = true;
      lab = (forced || best >= matched) ? cls : (best < unmatched ? 0 : -1);
      // WeightedAxisAlignedTargetAssigner (weighted_axis_aligned_target_assigner.py:166-173): a positive anchor carries
      // the label uncertainty of ITS arg-max ground truth in both branches -- gt_inds_force =
      // anchor_to_gt_argmax[anchors_with_max_overlap] for the anchors a ground truth forced, anchor_to_gt_argmax[pos_inds]
      // for those over the threshold -- i.e. of the box its regression targets are encoded from.
      if (unc_gt) unc_gt[(long long)b * N + i] = lab > 0 ? arg : -1;
      if (lab > 0) {                                         // ResidualCoder.encode_torch(gt[arg], anchor)
        const float* g = gt + ((long long)b * M + arg) * gt_cols;
        const float dxa = fmaxf(an[3], 1e-5f), dya = fmaxf(an[4], 1e-5f), dza = fmaxf(an[5], 1e-5f);
        const float dxg = fmaxf(g[3], 1e-5f), dyg = fmaxf(g[4], 1e-5f), dzg = fmaxf(g[5], 1e-5f);
        const float diag = sqrtf(dxa * dxa + dya * dya);
        t[0] = (g[0] - an[0]) / diag; t[1] = (g[1] - an[1]) / diag; t[2] = (g[2] - an[2]) / dza;
        t[3] = logf(dxg / dxa); t[4] = logf(dyg / dya); t[5] = logf(dzg / dza);
        t[6] = g[6] - an[6];
      }
    }
    if (unc_gt && arg < 0) unc_gt[(long long)b * N + i] = -1;
    labels[(long long)b * N + i] = lab;
#pragma unroll
    for (int k = 0; k < 7; ++k) targets[((long long)b * N + i) * 7 + k] = t[k];
  }
  const unsigned long long bal = __ballot(lab >= 0);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&s_cnt, __popcll(bal));
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(&num_examples[b], s_cnt);
}

__global__ void k_assign_weights(const int* __restrict__ labels, int N, const int* __restrict__ num_examples,
                                 int norm, float* __restrict__ w) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int n = num_examples[b];
  const float pos = norm ? 1.0f / (float)(n > 1 ? n : 1) : 1.0f;
  w[(long long)b * N + i] = labels[(long long)b * N + i] > 0 ? pos : 0.f;
}

extern "C" size_t glx_assign_targets_workspace_bytes(int B, int N) {
  return glx_align((size_t)B * N * 8) + glx_align((size_t)B * (TA_MAXGT + 2) * 4) + 256;
}

extern "C" int glx_assign_targets_ex(const float* anchors, int N, const float* gt_boxes, int B, int M,
                                     int gt_cols, int class_id, float matched_threshold,
                                     float unmatched_threshold, int norm_by_num_examples,
                                     int32_t* box_cls_labels, float* box_reg_targets, float* reg_weights,
                                     int32_t* uncertainty_gt_index, void* workspace, size_t workspace_bytes,
                                     void* stream);

extern "C" int glx_assign_targets(const float* anchors, int N, const float* gt_boxes, int B, int M,
                                  int gt_cols, int class_id, float matched_threshold,
                                  float unmatched_threshold, int norm_by_num_examples,
                                  int32_t* box_cls_labels, float* box_reg_targets, float* reg_weights,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  return glx_assign_targets_ex(anchors, N, gt_boxes, B, M, gt_cols, class_id, matched_threshold, unmatched_threshold,
                               norm_by_num_examples, box_cls_labels, box_reg_targets, reg_weights, nullptr, workspace,
                               workspace_bytes, stream);
}

extern "C" int glx_assign_targets_ex(const float* anchors, int N, const float* gt_boxes, int B, int M,
                                     int gt_cols, int class_id, float matched_threshold,
                                     float unmatched_threshold, int norm_by_num_examples,
                                     int32_t* box_cls_labels, float* box_reg_targets, float* reg_weights,
                                     int32_t* uncertainty_gt_index, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  if (B <= 0 || N <= 0) return GLX_OK;
  GLX_REQUIRE(anchors && gt_boxes && box_cls_labels && box_reg_targets && reg_weights,
              "glx_assign_targets: null pointer");
  GLX_REQUIRE(M >= 1 && M <= TA_MAXGT && gt_cols >= 8 && B <= 65535,
              "glx_assign_targets: 1..%d ground-truth rows of >= 8 columns (box + class)", TA_MAXGT);
  const size_t need = glx_assign_targets_workspace_bytes(B, N) - 256;
  if (!workspace || workspace_bytes < need) {
    glx_set_error("glx_assign_targets: workspace %zu < %zu bytes", workspace_bytes, need);
    return GLX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* amax = (float*)workspace;
  int* aarg = (int*)(amax + (size_t)B * N);
  int* gmax = (int*)((char*)workspace + glx_align((size_t)B * N * 8));
  int* n_valid = gmax + (size_t)B * TA_MAXGT;
  int* num_examples = n_valid + B;
  hipLaunchKernelGGL(k_assign_prepare, dim3(B), dim3(128), 0, st, gt_boxes, B, M, gt_cols, n_valid, gmax,
                     num_examples);
  const dim3 grid(glx_divup(N, 256), B);
  hipLaunchKernelGGL(k_assign_iou_max, grid, dim3(256), 0, st, anchors, N, gt_boxes, M, gt_cols, class_id,
                     (const int*)n_valid, amax, aarg, gmax);
  hipLaunchKernelGGL(k_assign_labels, grid, dim3(256), 0, st, anchors, N, gt_boxes, M, gt_cols, class_id,
                     matched_threshold, unmatched_threshold, (const int*)n_valid, (const float*)amax,
                     (const int*)aarg, (const int*)gmax, box_cls_labels, box_reg_targets, num_examples,
                     uncertainty_gt_index);
  hipLaunchKernelGGL(k_assign_weights, grid, dim3(256), 0, st, (const int*)box_cls_labels, N,
                     (const int*)num_examples, norm_by_num_examples, reg_weights);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ dense (anchor) head loss
// AnchorHeadTemplate.get_loss (pcdet/models/dense_heads/anchor_head_template.py:108-232) for a
// class-count C head with NUM_DIR_BINS = 2, all three terms and their gradients:
//   cls: SigmoidFocalClassificationLoss(alpha, gamma = 2) on the logits against the one-hot of the
//        anchor's label (don't-care anchors have weight 0), weights 1 / max(#positives of the frame, 1);
//   loc: add_sin_difference on the heading channel, code-weighted smooth-L1 (beta 1/9), positives only,
//        same normaliser;
//   dir: 2-bin softmax cross-entropy of the direction logits against
//        floor(limit_period(target_heading + anchor_heading - dir_offset, 0, 2 pi) / pi), positives only.
// Each term is summed over anchors and frames and divided by the batch size, then weighted.
// Pass 1 counts the positives per frame, pass 2 writes per-block partial sums (fp64) and the three
// gradients, pass 3 reduces the partials in a fixed order.
#define RPN_THREADS 256

__global__ void k_rpn_count_pos(const int* __restrict__ labels, int A, int* __restrict__ npos) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool pos = i < A && labels[(long long)b * A + i] > 0;
  const unsigned long long bal = __ballot(pos);
  if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&npos[b], __popcll(bal));
}

struct RpnParams {
  float alpha, beta, dir_offset, cls_w, loc_w, dir_w;
  float cw[7];
  int num_class, class_agnostic;
};

__global__ __launch_bounds__(RPN_THREADS) void k_rpn_loss(
    const float* __restrict__ cls_preds, const float* __restrict__ box_preds,
    const float* __restrict__ dir_preds, const int* __restrict__ labels,
    const float* __restrict__ reg_targets, const float* __restrict__ anchors, int B, int A,
    const int* __restrict__ npos, RpnParams p, double* __restrict__ partial,
    float* __restrict__ g_cls, float* __restrict__ g_box, float* __restrict__ g_dir) {
  __shared__ double red[RPN_THREADS];
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int C = p.num_class;
  double l_cls = 0, l_loc = 0, l_dir = 0;
  if (i < A) {
    const long long e = (long long)b * A + i;
    int lab = labels[e];
    const float norm = 1.0f / fmaxf((float)npos[b], 1.0f);
    const bool pos = lab > 0;
    const float w_cls = lab >= 0 ? norm : 0.f;               // positives and negatives, not don't-care
    if (p.class_agnostic && pos) lab = 1;
    const float inv_b = 1.0f / (float)B;
    // ---- classification
    for (int c = 0; c < C; ++c) {
      const float x = cls_preds[e * C + c];
      const float z = (lab == c + 1) ? 1.f : 0.f;
      const float ps = 1.f / (1.f + expf(-x));
      const float aw = z * p.alpha + (1.f - z) * (1.f - p.alpha);
      const float pt = z * (1.f - ps) + (1.f - z) * ps;
      const float bce = fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
      l_cls += (double)(aw * (pt * pt) * bce * w_cls);
      if (g_cls) {
        const float dpt = (1.f - 2.f * z) * ps * (1.f - ps);
        g_cls[e * C + c] = w_cls * aw * (2.f * pt * dpt * bce + pt * pt * (ps - z)) * inv_b * p.cls_w;
      }
    }
    // ---- localisation + direction (positives only)
    float gb[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gd0 = 0.f, gd1 = 0.f;
    if (pos) {
      const float* bp = box_preds + e * 7;
      const float* tg = reg_targets + e * 7;
      for (int k = 0; k < 7; ++k) {
        float diff, dd = 1.f;
        if (k == 6) {
          const float sa = sinf(bp[6]), ca = cosf(bp[6]), sb = sinf(tg[6]), cb = cosf(tg[6]);
          diff = sa * cb - ca * sb;
          dd = ca * cb + sa * sb;
        } else {
          float t = tg[k];
          if (isnan(t)) t = bp[k];
          diff = bp[k] - t;
        }
        diff *= p.cw[k];
        const float n = fabsf(diff);
        l_loc += (double)((n < p.beta ? 0.5f * n * n / p.beta : n - 0.5f * p.beta) * norm);
        const float ds = n < p.beta ? diff / p.beta : (diff > 0.f ? 1.f : -1.f);
        gb[k] = ds * p.cw[k] * dd * norm * inv_b * p.loc_w;
      }
      if (dir_preds) {
        const float TWO_PI = (float)(2.0 * 3.14159265358979323846);
        const float rot = tg[6] + anchors[(long long)i * 7 + 6];
        const float v = rot - p.dir_offset;
        const float off = v - floorf(v / TWO_PI + 0.f) * TWO_PI;
        int bin = (int)floorf(off / (float)(2.0 * 3.14159265358979323846 / 2));
        bin = bin < 0 ? 0 : (bin > 1 ? 1 : bin);
        const float d0 = dir_preds[e * 2], d1 = dir_preds[e * 2 + 1];
        const float m = fmaxf(d0, d1);
        const float lse = m + logf(expf(d0 - m) + expf(d1 - m));
        l_dir += (double)((lse - (bin ? d1 : d0)) * norm);
        const float s0 = expf(d0 - lse), s1 = expf(d1 - lse);
        gd0 = (s0 - (bin == 0 ? 1.f : 0.f)) * norm * inv_b * p.dir_w;
        gd1 = (s1 - (bin == 1 ? 1.f : 0.f)) * norm * inv_b * p.dir_w;
      }
    }
    if (g_box) {
#pragma unroll
      for (int k = 0; k < 7; ++k) g_box[e * 7 + k] = gb[k];
    }
    if (g_dir) { g_dir[e * 2] = gd0; g_dir[e * 2 + 1] = gd1; }
  }
  const long long blk = (long long)blockIdx.y * gridDim.x + blockIdx.x;
  const double a = kl_block_sum(l_cls, red), c = kl_block_sum(l_loc, red), d = kl_block_sum(l_dir, red);
  if (threadIdx.x == 0) { partial[blk * 3] = a; partial[blk * 3 + 1] = c; partial[blk * 3 + 2] = d; }
}

__global__ __launch_bounds__(RPN_THREADS) void k_rpn_finish(const double* __restrict__ partial, int nblk,
                                                            int B, RpnParams p, float* __restrict__ out) {
  __shared__ double red[RPN_THREADS];
  double s[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < nblk; i += RPN_THREADS)
    for (int k = 0; k < 3; ++k) s[k] += partial[(long long)i * 3 + k];
  const double a = kl_block_sum(s[0], red), c = kl_block_sum(s[1], red), d = kl_block_sum(s[2], red);
  if (threadIdx.x == 0) {
    out[1] = (float)(a / B) * p.cls_w;
    out[2] = (float)(c / B) * p.loc_w;
    out[3] = (float)(d / B) * p.dir_w;
    out[0] = out[1] + out[2] + out[3];
  }
}

extern "C" size_t glx_rpn_loss_workspace_bytes(int B, int A) {
  return glx_align((size_t)B * glx_divup(A, RPN_THREADS) * 3 * sizeof(double)) + glx_align((size_t)B * 4) + 256;
}

extern "C" int glx_rpn_loss(const float* cls_preds, const float* box_preds, const float* dir_preds,
                            const int32_t* box_cls_labels, const float* box_reg_targets,
                            const float* anchors, int B, int A, int num_class, int class_agnostic,
                            float alpha, float beta, const float* code_weights, float dir_offset,
                            float cls_weight, float loc_weight, float dir_weight, float* out4,
                            float* grad_cls, float* grad_box, float* grad_dir, void* workspace,
                            size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(out4, "glx_rpn_loss: null output");
  if (B <= 0 || A <= 0) return GLX_OK;
  GLX_REQUIRE(cls_preds && box_preds && box_cls_labels && box_reg_targets && (anchors || !dir_preds),
              "glx_rpn_loss: null pointer");
  GLX_REQUIRE(num_class >= 1 && B <= 65535, "glx_rpn_loss: bad sizes");
  const size_t need = glx_rpn_loss_workspace_bytes(B, A) - 256;
  if (!workspace || workspace_bytes < need) {
    glx_set_error("glx_rpn_loss: workspace %zu < %zu bytes", workspace_bytes, need);
    return GLX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int nbx = glx_divup(A, RPN_THREADS);
  double* partial = (double*)workspace;
  int* npos = (int*)((char*)workspace + glx_align((size_t)B * nbx * 3 * sizeof(double)));
  GlxFillJob job{npos, (size_t)B * sizeof(int), 0};
  int rc = glx_fill_multi(&job, 1, st);
  if (rc != GLX_OK) return rc;
  RpnParams p;
  p.alpha = alpha; p.beta = beta; p.dir_offset = dir_offset;
  p.cls_w = cls_weight; p.loc_w = loc_weight; p.dir_w = dir_weight;
  for (int k = 0; k < 7; ++k) p.cw[k] = code_weights ? code_weights[k] : 1.f;
  p.num_class = num_class; p.class_agnostic = class_agnostic;
  const dim3 grid(nbx, B);
  hipLaunchKernelGGL(k_rpn_count_pos, grid, dim3(RPN_THREADS), 0, st, box_cls_labels, A, npos);
  hipLaunchKernelGGL(k_rpn_loss, grid, dim3(RPN_THREADS), 0, st, cls_preds, box_preds, dir_preds,
                     box_cls_labels, box_reg_targets, anchors, B, A, (const int*)npos, p, partial, grad_cls,
                     grad_box, grad_dir);
  hipLaunchKernelGGL(k_rpn_finish, dim3(1), dim3(RPN_THREADS), 0, st, (const double*)partial, nbx * B, B, p,
                     out4);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ RoI classification loss
// RoIHeadTemplate.get_box_cls_layer_loss, CLS_LOSS = BinaryCrossEntropy (roi_head_template.py:246-272):
// F.binary_cross_entropy(sigmoid(rcnn_cls), soft IoU labels) over the RoIs whose label is >= 0,
// divided by max(#valid, 1); torch's conventions kept: log terms clamped at -100, the gradient's
// p(1-p) denominator clamped at 1e-12.  One block, loss + gradient.
__global__ __launch_bounds__(KL_THREADS) void k_rcnn_cls_loss(const float* __restrict__ logits,
                                                              const float* __restrict__ labels, int R,
                                                              float weight, float* __restrict__ out,
                                                              float* __restrict__ grad) {
  __shared__ double red[KL_THREADS];
  double c = 0;
  for (int i = threadIdx.x; i < R; i += KL_THREADS) c += labels[i] >= 0.f ? 1.0 : 0.0;
  const double nv = kl_block_sum(c, red);
  const float scale = weight / (float)(nv > 1.0 ? nv : 1.0);
  double acc = 0;
  for (int i = threadIdx.x; i < R; i += KL_THREADS) {
    const float y = labels[i];
    const float p = 1.f / (1.f + expf(-logits[i]));
    const float m = y >= 0.f ? 1.f : 0.f;
    const float l = -(y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
    acc += (double)(l * m);
    if (grad) grad[i] = m * (p - y) / fmaxf((1.f - p) * p, 1e-12f) * (p * (1.f - p)) * scale;
  }
  const double total = kl_block_sum(acc, red);
  if (threadIdx.x == 0) { out[0] = (float)total * scale; out[1] = (float)nv; }
}

extern "C" int glx_rcnn_cls_loss(const float* rcnn_cls, const float* rcnn_cls_labels, int R, float weight,
                                 float* out2, float* grad, void* stream) {
  GLX_REQUIRE(out2 && (R == 0 || (rcnn_cls && rcnn_cls_labels)), "glx_rcnn_cls_loss: null pointer");
  hipLaunchKernelGGL(k_rcnn_cls_loss, dim3(1), dim3(KL_THREADS), 0, (hipStream_t)stream, rcnn_cls,
                     rcnn_cls_labels, R, weight, out2, grad);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ GLENet's score rescaling + classification loss
// VoxelRCNNKLLabelIoUHead.forward (voxelrcnn_kl_label_iou_head.py:70-76): the classification logit is rescaled by the
// predicted localisation certainty before it is used, p = sigmoid(ori_cls) * sigmoid(std_logit),
// rcnn_cls = log((p + 1e-6) / (1 - p + 1e-6)) -- nine elementwise launches forward and fourteen backward as tensor
// ops.  Here the rescaling, the BinaryCrossEntropy loss above and the chain rule down to the two logits are ONE
// block: rcnn_cls (R) is written for the callers that read it, grad_ori / grad_std hold d loss / d logit.
template <int NT>
__device__ __forceinline__ float cls_rescale_body(
    const float* __restrict__ ori_cls, const float* __restrict__ std_logit,
    const float* __restrict__ labels, int R, float weight, float* __restrict__ rcnn_cls,
    float* __restrict__ out, float* __restrict__ grad_ori, float* __restrict__ grad_std, double* red) {
  double c = 0;
  if (labels)
    for (int i = threadIdx.x; i < R; i += NT) c += labels[i] >= 0.f ? 1.0 : 0.0;
  const double nv = labels ? blk_sum<NT>(c, red) : 0.0;
  const float scale = weight / (float)(nv > 1.0 ? nv : 1.0);
  double acc = 0;
  for (int i = threadIdx.x; i < R; i += NT) {
    const float sa = 1.f / (1.f + expf(-ori_cls[i])), sb = 1.f / (1.f + expf(-std_logit[i]));
    const float pr = sa * sb;
    const float num = pr + 1e-6f, den = (1.f - pr) + 1e-6f;
    const float z = logf(num / den);
    rcnn_cls[i] = z;
    if (!labels) continue;
    const float y = labels[i];
    const float p = 1.f / (1.f + expf(-z));
    const float m = y >= 0.f ? 1.f : 0.f;
    const float l = -(y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
    acc += (double)(l * m);
    const float gz = m * (p - y) / fmaxf((1.f - p) * p, 1e-12f) * (p * (1.f - p)) * scale;   // d loss / d rcnn_cls
    const float gp = gz * (1.f / num + 1.f / den);                                            // d rcnn_cls / d pr
    if (grad_ori) grad_ori[i] = gp * sb * sa * (1.f - sa);
    if (grad_std) grad_std[i] = gp * sa * sb * (1.f - sb);
  }
  if (!labels) return 0.f;
  const double total = blk_sum<NT>(acc, red);
  const float loss = (float)total * scale;
  if (threadIdx.x == 0) { out[0] = loss; out[1] = (float)nv; }
  return loss;
}

__global__ __launch_bounds__(KL_THREADS) void k_cls_rescale_loss(
    const float* __restrict__ ori_cls, const float* __restrict__ std_logit,
    const float* __restrict__ labels, int R, float weight, float* __restrict__ rcnn_cls,
    float* __restrict__ out, float* __restrict__ grad_ori, float* __restrict__ grad_std) {
  __shared__ double red[KL_THREADS];
  cls_rescale_body<KL_THREADS>(ori_cls, std_logit, labels, R, weight, rcnn_cls, out, grad_ori, grad_std, red);
}

// ------------------------------------------------------------------ the three RoI-head terms in ONE launch
// VoxelRCNNKLLabelIoUHead.get_loss (roi_head_template.py get_loss -> voxelrcnn_kl_label_iou_head.py:93-172 + the
// classification term, roi_head_template.py:246-272) behind GLENet's score rescaling: the three blocks above ran as
// three launches with seven tensor launches between them (mask compare / cast, slices of the 8-column ground-truth
// rows, the sum of the terms) and autograd added the two gradients of rcnn_reg in another.  Here one block of 1024
// threads runs the three bodies back to back: ground-truth rows through their stride, the int64 mask as it is,
// d loss / d rcnn_reg = KL term + corner term, out9[10] = { total, cls, #valid, KL, src, square, log, #fg, corner, #fg }.
#define RHL_THREADS 1024
struct RoiHeadLossArgs {
  const float *ori_cls, *std_logit, *cls_labels, *rcnn_reg, *rcnn_reg_std, *rois, *gt_ct, *gt_src, *label_var;
  const long long* reg_valid;
  int R, gt_ct_ld, gt_src_ld;
  KlCodeWeights cw;
  float beta, w_cls, w_reg, w_corner;
  float *rcnn_cls, *out9, *grad_ori, *grad_std_logit, *grad_reg, *grad_reg_std;
};

__global__ __launch_bounds__(RHL_THREADS) void k_roi_head_losses(RoiHeadLossArgs a) {
  __shared__ double red[RHL_THREADS];
  const float l_cls = cls_rescale_body<RHL_THREADS>(a.ori_cls, a.std_logit, a.cls_labels, a.R, a.w_cls, a.rcnn_cls, a.out9 + 1,
                                                    a.grad_ori, a.grad_std_logit, red);
  const float l_kl = kl_reg_body<RHL_THREADS>(a.rcnn_reg, a.rcnn_reg_std, a.rois, a.gt_ct, a.gt_ct_ld, a.label_var, a.reg_valid,
                                              a.R, a.cw, a.beta, a.w_reg, a.out9 + 3, a.grad_reg, a.grad_reg_std, red);
  __threadfence_block();          // the KL term's rows of grad_reg, written by other threads of this block
  __syncthreads();
  const float l_cor = corner_loss_body<RHL_THREADS, true>(a.rcnn_reg, a.rois, a.gt_src, a.gt_src_ld, a.reg_valid, a.R,
                                                          a.w_corner, a.out9 + 8, a.grad_reg, red);
  if (threadIdx.x == 0) a.out9[0] = (l_cls + l_kl) + l_cor;
}

extern "C" int glx_cls_rescale_loss(const float* ori_cls, const float* std_logit,
                                    const float* rcnn_cls_labels, int R, float weight, float* rcnn_cls,
                                    float* out2, float* grad_ori, float* grad_std, void* stream) {
  GLX_REQUIRE(R == 0 || (ori_cls && std_logit && rcnn_cls), "glx_cls_rescale_loss: null pointer");
  GLX_REQUIRE(!rcnn_cls_labels || out2, "glx_cls_rescale_loss: labels without an output for the loss");
  hipLaunchKernelGGL(k_cls_rescale_loss, dim3(1), dim3(KL_THREADS), 0, (hipStream_t)stream, ori_cls,
                     std_logit, rcnn_cls_labels, R, weight, rcnn_cls, out2, grad_ori, grad_std);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

extern "C" int glx_roi_head_losses(const glx_roi_head_losses_args* args, void* stream) {
  GLX_REQUIRE(args && args->out, "glx_roi_head_losses: null arguments / output");
  const glx_roi_head_losses_args& g = *args;
  GLX_REQUIRE(g.R == 0 || (g.ori_cls && g.std_logit && g.cls_labels && g.rcnn_reg && g.rcnn_reg_std && g.rois && g.gt_ct &&
                           g.gt_src && g.label_var && g.reg_valid && g.rcnn_cls),
              "glx_roi_head_losses: null pointer");
  GLX_REQUIRE(g.gt_ct_ld >= 7 && g.gt_src_ld >= 7, "glx_roi_head_losses: ground-truth rows need >= 7 columns (%d, %d)",
              g.gt_ct_ld, g.gt_src_ld);
  RoiHeadLossArgs a;
  a.ori_cls = g.ori_cls; a.std_logit = g.std_logit; a.cls_labels = g.cls_labels; a.rcnn_reg = g.rcnn_reg;
  a.rcnn_reg_std = g.rcnn_reg_std; a.rois = g.rois; a.gt_ct = g.gt_ct; a.gt_src = g.gt_src; a.label_var = g.label_var;
  a.reg_valid = (const long long*)g.reg_valid;
  a.R = g.R; a.gt_ct_ld = g.gt_ct_ld; a.gt_src_ld = g.gt_src_ld;
  for (int k = 0; k < 7; ++k) a.cw.w[k] = g.code_weights[k];
  a.beta = g.beta; a.w_cls = g.w_cls; a.w_reg = g.w_reg; a.w_corner = g.w_corner;
  a.rcnn_cls = g.rcnn_cls; a.out9 = g.out; a.grad_ori = g.grad_ori; a.grad_std_logit = g.grad_std_logit;
  a.grad_reg = g.grad_reg; a.grad_reg_std = g.grad_reg_std;
  hipLaunchKernelGGL(k_roi_head_losses, dim3(1), dim3(RHL_THREADS), 0, (hipStream_t)stream, a);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------ anchor head: predicted boxes
// AnchorHeadTemplate.generate_predicted_boxes (anchor_head_template.py:222-271): ResidualCoder.decode_torch
// (box_coder_utils.py:44-69) of every anchor's residuals and the heading put into the predicted direction bin
// (limit_period of common_utils.py:35-38) -- thirty elementwise launches as tensor ops.  The arithmetic keeps the
// tensor expression's rounding: products and sums are rounded separately (no fused multiply-add), a division by a
// Python scalar is ATen's multiplication by the float reciprocal.
__global__ __launch_bounds__(256) void k_predicted_boxes(
    const float* __restrict__ box_preds, const float* __restrict__ dir_preds,
    const float* __restrict__ anchors, long long total, int A, int ndir, float dir_offset,
    float dir_limit_offset, float* __restrict__ boxes) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float* a = anchors + (i % A) * 7;
  const float* e = box_preds + i * 7;
  float* o = boxes + i * 7;
  const float diag = sqrtf(__fadd_rn(__fmul_rn(a[3], a[3]), __fmul_rn(a[4], a[4])));
  o[0] = __fadd_rn(__fmul_rn(e[0], diag), a[0]);
  o[1] = __fadd_rn(__fmul_rn(e[1], diag), a[1]);
  o[2] = __fadd_rn(__fmul_rn(e[2], a[5]), a[2]);
  o[3] = __fmul_rn(expf(e[3]), a[3]);
  o[4] = __fmul_rn(expf(e[4]), a[4]);
  o[5] = __fmul_rn(expf(e[5]), a[5]);
  float r = __fadd_rn(e[6], a[6]);
  if (dir_preds) {
    const float* d = dir_preds + i * ndir;
    int label = 0;
    float best = d[0];
    for (int k = 1; k < ndir; ++k)
      if (d[k] > best) { best = d[k]; label = k; }
    const float period = (float)(2.0 * 3.14159265358979323846 / ndir);
    const float inv_period = 1.f / period;
    const float val = __fsub_rn(r, dir_offset);
    const float rot = __fsub_rn(val, __fmul_rn(floorf(__fadd_rn(__fmul_rn(val, inv_period), dir_limit_offset)), period));
    r = __fadd_rn(__fadd_rn(rot, dir_offset), __fmul_rn(period, (float)label));
  }
  o[6] = r;
}

extern "C" int glx_predicted_boxes(const float* box_preds, const float* dir_preds, const float* anchors,
                                   int B, int A, int num_dir_bins, float dir_offset,
                                   float dir_limit_offset, float* boxes, void* stream) {
  if (B <= 0 || A <= 0) return GLX_OK;
  GLX_REQUIRE(box_preds && anchors && boxes, "glx_predicted_boxes: null pointer");
  GLX_REQUIRE(!dir_preds || num_dir_bins >= 1, "glx_predicted_boxes: direction bins");
  const long long total = (long long)B * A;
  hipLaunchKernelGGL(k_predicted_boxes, dim3((unsigned)glx_divup(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, box_preds, dir_preds, anchors, total, A, num_dir_bins, dir_offset,
                     dir_limit_offset, boxes);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
