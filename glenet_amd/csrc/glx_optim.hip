// Gradient-norm clipping + AdamW over ONE flat fp32 parameter buffer: the update of the training step
// (tools/train_utils/train_utils.py:38-39: clip_grad_norm_(model.parameters(), GRAD_NORM_CLIP) then
// optimizer.step(); optimiser = adam_onecycle, tools/train_utils/optimization/__init__.py:29-53: Adam with
// true weight decay, betas (0.9, 0.99), learning rate and beta1 driven by the one-cycle schedule).
// torch.optim.AdamW(capturable, foreach) spends ~300 launches per step on the model's 143 parameter tensors
// (1.1 ms of a 24 ms step); with parameters, gradients and both moments in flat buffers the update is two
// launches that move 7 x 4 bytes per element once (HBM-bound, ~210 MB for GLENet-VR).
//   k_gradnorm_partial : per-block fp64 sums of g^2 in a fixed order (deterministic); block 0 bumps the step
//   k_adamw            : every block folds the partials in the same order -> the same clip coefficient, then
//                        the elementwise update with torch.optim.AdamW's arithmetic (fp32, one rounding per op)
#include "glx_common.h"

typedef float of32x4 __attribute__((ext_vector_type(4)));

#define OPT_THREADS 256
#define OPT_MAX_BLOCKS 1024

// grad_scale: the gradient that is clipped and applied is g * grad_scale, each element rounded to float first -- the
// 1 / world_size of a data-parallel SUM all-reduce folded into the update (it was a separate pass over the buffer).
__global__ __launch_bounds__(OPT_THREADS) void k_gradnorm_partial(const float* __restrict__ g, long long n,
                                                                  float grad_scale, double* __restrict__ partial,
                                                                  int* __restrict__ step) {
  const long long n4 = n >> 2;
  double s = 0;
  for (long long e = (long long)blockIdx.x * OPT_THREADS + threadIdx.x; e < n4; e += (long long)gridDim.x * OPT_THREADS) {
    of32x4 v = reinterpret_cast<const of32x4*>(g)[e] * grad_scale;
    s += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    float v = g[(n4 << 2) + threadIdx.x] * grad_scale;
    s += (double)v * v;
  }
  __shared__ double red[OPT_THREADS];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = OPT_THREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = red[0];
    if (blockIdx.x == 0 && step) *step += 1;
  }
}

// hyper[0] = learning rate, hyper[1] = beta1 (device scalars: the one-cycle schedule writes them, a recorded
// HIP graph reads them).  max_norm <= 0: no clipping (the norm is still reported).
__global__ __launch_bounds__(OPT_THREADS) void k_adamw(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, long long n,
                                                       const float* __restrict__ hyper, float beta2, float eps,
                                                       float wd, float max_norm, float grad_scale,
                                                       const int* __restrict__ step,
                                                       const double* __restrict__ partial, int nparts,
                                                       float* __restrict__ norm_out) {
  __shared__ float s_coef;
  __shared__ double s_wave[OPT_THREADS / 64];
  // every block folds the partials the same way (thread t takes parts t, t + 256, ...; waves by shuffles, then in
  // order): the same clip coefficient everywhere, and not a 900-step serial chain in front of every block's work
  double mine = 0;
  for (int i = threadIdx.x; i < nparts; i += OPT_THREADS) mine += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int i = 0; i < OPT_THREADS / 64; ++i) s += s_wave[i];
    const float total = (float)sqrt(s);
    float coef = 1.f;
    if (max_norm > 0.f) {                       // clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
      coef = max_norm / (total + 1e-6f);
      coef = coef > 1.f ? 1.f : coef;
    }
    s_coef = coef;
    if (blockIdx.x == 0 && norm_out) *norm_out = total;
  }
  __syncthreads();
  const float coef = s_coef;
  const float lr = hyper[0], b1 = hyper[1];
  const int t = *step;
  // torch.optim.adam._single_tensor_adam (capturable branch) in fp32
  const float bc1 = 1.f - powf(b1, (float)t), bc2 = 1.f - powf(beta2, (float)t);
  const float step_size = lr / bc1;
  const float bc2_sqrt = sqrtf(bc2);
  const float decay = 1.f - lr * wd;
  const long long n4 = n >> 2;
  for (long long e = (long long)blockIdx.x * OPT_THREADS + threadIdx.x; e < n4; e += (long long)gridDim.x * OPT_THREADS) {
    of32x4 pv = reinterpret_cast<of32x4*>(p)[e], gv = reinterpret_cast<const of32x4*>(g)[e];
    of32x4 mv = reinterpret_cast<of32x4*>(m)[e], vv = reinterpret_cast<of32x4*>(v)[e];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float gi = (gv[i] * grad_scale) * coef;
      pv[i] *= decay;
      mv[i] = mv[i] + (gi - mv[i]) * (1.f - b1);              // exp_avg.lerp_(grad, 1 - beta1)
      vv[i] = vv[i] * beta2 + (1.f - beta2) * gi * gi;        // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
      const float denom = sqrtf(vv[i]) / bc2_sqrt + eps;
      pv[i] -= step_size * (mv[i] / denom);
    }
    reinterpret_cast<of32x4*>(p)[e] = pv;
    reinterpret_cast<of32x4*>(m)[e] = mv;
    reinterpret_cast<of32x4*>(v)[e] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long e = (n4 << 2) + threadIdx.x;
    const float gi = (g[e] * grad_scale) * coef;
    float pe = p[e] * decay;
    const float me = m[e] + (gi - m[e]) * (1.f - b1);
    const float ve = v[e] * beta2 + (1.f - beta2) * gi * gi;
    pe -= step_size * (me / (sqrtf(ve) / bc2_sqrt + eps));
    p[e] = pe; m[e] = me; v[e] = ve;
  }
}

static int opt_blocks(long long n) {
  long long b = (n / 4 + OPT_THREADS * 8 - 1) / (OPT_THREADS * 8);
  return (int)(b < 1 ? 1 : (b > OPT_MAX_BLOCKS ? OPT_MAX_BLOCKS : b));
}

extern "C" size_t glx_adamw_workspace_bytes(void) { return OPT_MAX_BLOCKS * sizeof(double) + 256; }

extern "C" int glx_adamw_clip_step_scaled(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                          int64_t n, const float* hyper, float beta2, float eps, float weight_decay,
                                          float max_norm, float grad_scale, int32_t* step, float* norm_out,
                                          void* workspace, size_t workspace_bytes, void* stream);

extern "C" int glx_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                   int64_t n, const float* hyper, float beta2, float eps, float weight_decay,
                                   float max_norm, int32_t* step, float* norm_out, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return glx_adamw_clip_step_scaled(params, grads, exp_avg, exp_avg_sq, n, hyper, beta2, eps, weight_decay, max_norm,
                                    1.0f, step, norm_out, workspace, workspace_bytes, stream);
}

extern "C" int glx_adamw_clip_step_scaled(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                          int64_t n, const float* hyper, float beta2, float eps, float weight_decay,
                                          float max_norm, float grad_scale, int32_t* step, float* norm_out,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  GLX_REQUIRE(n >= 0 && (n == 0 || (params && grads && exp_avg && exp_avg_sq)) && hyper && step,
              "glx_adamw_clip_step: null pointer");
  GLX_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
              "glx_adamw_clip_step: buffers must be 16-byte aligned");
  if (!workspace || workspace_bytes < glx_adamw_workspace_bytes() - 256) {
    glx_set_error("glx_adamw_clip_step: workspace %zu < %zu bytes", workspace_bytes, glx_adamw_workspace_bytes() - 256);
    return GLX_EWORKSPACE;
  }
  if (n == 0) return GLX_OK;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = opt_blocks(n);
  hipLaunchKernelGGL(k_gradnorm_partial, dim3(blocks), dim3(OPT_THREADS), 0, st, grads, (long long)n, grad_scale,
                     (double*)workspace, step);
  hipLaunchKernelGGL(k_adamw, dim3(blocks), dim3(OPT_THREADS), 0, st, params, grads, exp_avg, exp_avg_sq,
                     (long long)n, hyper, beta2, eps, weight_decay, max_norm, grad_scale, (const int*)step,
                     (const double*)workspace, blocks, norm_out);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ sum of the tensors' 2-norms
// The regulariser of cvae_uncertainty/model.py:20-28 (`l2_regularisation`: the SUM over a module's parameter tensors of their 2-norms)
// on a flat parameter buffer whose tensors are the segments segs[i] = (start, length): one block per tensor for the norms (fixed
// summation order), one block for their sum; and its gradient  coef scale p / |p|  (0 for a zero tensor, as torch's norm backward)
// added into the flat gradient buffer in one launch -- autograd's form is four elementwise launches per tensor and one more to add
// the result to the tensor's other gradient.
#define SEG_SPLIT 8      // blocks per tensor (the largest tensor sets the launch's length)
__global__ __launch_bounds__(256) void k_seg_norms(const float* __restrict__ p, const long long* __restrict__ segs, double* __restrict__ part) {
  __shared__ double red[256];
  const long long start = segs[2 * blockIdx.x], len = segs[2 * blockIdx.x + 1];
  const long long per = (len + SEG_SPLIT - 1) / SEG_SPLIT, lo = per * blockIdx.y, hi = lo + per < len ? lo + per : len;
  double s = 0.0;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const double v = (double)p[start + i];
    s += v * v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x * SEG_SPLIT + blockIdx.y] = red[0];
}
__global__ void k_seg_norms_sum(const double* __restrict__ part, int nseg, float scale, float* __restrict__ norms, float* __restrict__ total) {
  for (int i = threadIdx.x; i < nseg; i += blockDim.x) {
    double s = 0.0;
    for (int k = 0; k < SEG_SPLIT; ++k) s += part[i * SEG_SPLIT + k];
    norms[i] = (float)sqrt(s);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < nseg; ++i) s += norms[i];          // torch.stack(norms).sum() adds floats too; the order is this one, always
    total[0] = s * scale;
  }
}
__global__ __launch_bounds__(256) void k_seg_norm_grad(const float* __restrict__ p, const long long* __restrict__ segs, int nseg,
                                                       const float* __restrict__ norms, const float* __restrict__ coef, float scale,
                                                       float* __restrict__ grads, long long n) {
  const float c = (coef ? coef[0] : 1.f) * scale;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    int lo = 0, hi = nseg - 1;                              // the last segment that starts at or before i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (segs[2 * mid] <= i) lo = mid; else hi = mid - 1;
    }
    const long long start = segs[2 * lo], len = segs[2 * lo + 1];
    if (i < start || i >= start + len) continue;            // alignment padding between tensors
    const float nm = norms[lo];
    if (nm > 0.f) grads[i] += c * (p[i] / nm);
  }
}
extern "C" size_t glx_flat_l2_workspace_bytes(int nseg) { return glx_align((size_t)(nseg > 0 ? nseg : 1) * SEG_SPLIT * sizeof(double)); }
extern "C" int glx_flat_l2_norms(const float* params, const int64_t* segs, int nseg, float scale, float* norms, float* total,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  if (nseg <= 0) return GLX_OK;
  GLX_REQUIRE(params && segs && norms && total && workspace, "glx_flat_l2_norms: null pointer");
  GLX_REQUIRE(workspace_bytes >= glx_flat_l2_workspace_bytes(nseg), "glx_flat_l2_norms: workspace too small");
  hipLaunchKernelGGL(k_seg_norms, dim3(nseg, SEG_SPLIT), dim3(256), 0, (hipStream_t)stream, params, (const long long*)segs, (double*)workspace);
  hipLaunchKernelGGL(k_seg_norms_sum, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nseg, scale, norms, total);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
extern "C" int glx_flat_l2_norm_grad_add(const float* params, const int64_t* segs, int nseg, const float* norms, const float* coef,
                                         float scale, float* grads, int64_t n, void* stream) {
  if (nseg <= 0 || n <= 0) return GLX_OK;
  GLX_REQUIRE(params && segs && norms && grads, "glx_flat_l2_norm_grad_add: null pointer");
  const int blocks = opt_blocks(n);
  hipLaunchKernelGGL(k_seg_norm_grad, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, (const long long*)segs, nseg, norms,
                     coef, scale, grads, (long long)n);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}

// ------------------------------------------------------------------------------------------------ stage stamps
// One-thread launch that stores the constant 100 MHz wall clock (s_memrealtime) into stamps[slot].  Recorded into a
// captured step at its stage boundaries it times the stages INSIDE graph replays, where events and the profiler's
// per-kernel signals would change what is measured (bench.py `stages_ms`).
__global__ void k_stamp(unsigned long long* stamps, int slot) { stamps[slot] = wall_clock64(); }

extern "C" int glx_stamp(unsigned long long* stamps, int slot, void* stream) {
  GLX_REQUIRE(stamps != nullptr && slot >= 0, "glx_stamp: bad arguments");
  hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, stamps, slot);
  GLX_LAUNCH_CHECK();
  return GLX_OK;
}
