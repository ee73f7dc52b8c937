/*
 * glenet_hip.h -- C ABI of libglenet_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for GLENet's data-parallel detection hot path:
 * every entry point takes plain device pointers + sizes + an explicit HIP stream
 * (passed as void*, NULL = default stream), never allocates, never synchronises
 * the host unless its comment says so, and returns GLX_OK (0) or a negative GLX_E*
 * code; glx_last_error() holds the message.  Temporary memory is caller-owned: each
 * op that needs scratch has a *_workspace_bytes() query.
 *
 * Each declaration cites the reference interface (path:line under the GLENet tree)
 * that it replaces.  "spconv" means the third-party traveller59/spconv package that
 * the reference imports at pcdet/utils/spconv_utils.py:3-6 (not vendored, unpinned).
 *
 * Conventions
 *   - all tensors row-major contiguous, float32 / int32 unless stated;
 *   - sparse indices are (N,4) int32 [batch, z, y, x]  (spconv_backbone.py:141-146);
 *   - boxes are (N,7) float32 [x, y, z, dx, dy, dz, heading] (iou3d_nms_kernel.cu:104);
 *   - a "grid" is (B, D, H, W) = (batch, z, y, x) cell counts;
 *   - "n_live" arguments (NULL, or a device int32[1]) make an op shape-static: the row count N
 *     given by the host is then only the CAPACITY of the buffers and the launch, the number of
 *     live rows is read on the device (min(N, *n_live)) and rows beyond it are neither read nor
 *     written.  A whole frame (voxelize -> rules -> convs -> dense) then runs without a single
 *     host synchronisation and can be captured in a HIP graph.
 */
#ifndef GLENET_HIP_H_
#define GLENET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLX_OK 0
#define GLX_EINVAL (-1)     /* bad argument (shape, alignment, unsupported channel count) */
#define GLX_EHIP (-2)       /* HIP runtime / launch failure */
#define GLX_EWORKSPACE (-3) /* workspace too small */

/* Message of the last failing call on this host thread ("" if none). */
const char* glx_last_error(void);
/* Library ABI version (bumped on any signature change). */
int glx_abi_version(void);

/* ---- per-call options: explicit ARGUMENTS of the *_ex entry points (NULL = none).  There is no "applies to the next
 * call" state anywhere in this ABI: two host threads driving two streams share nothing. ----
 *
 * glx_bn_stats: a TRAINING-mode BatchNorm behind a convolution (spconv_backbone.py:21-25: conv -> BatchNorm1d -> ReLU;
 * base_bev_backbone.py:37-49: Conv2d -> BatchNorm2d -> ReLU).  The per-channel statistics of the convolution's output are
 * taken in its epilogue (one launch) and finalized by the last block: coef (2 * Cout floats) = scale | shift for
 * glx_bn_apply_forward, save_mean / save_invstd (Cout) for glx_bn_relu_backward, running statistics updated with
 * nn.BatchNorm's semantics (NULL: not tracked).  state: glx_bn_state_bytes() device bytes, zero-filled once, shared by
 * the calls of ONE stream.
 * glx_epilogue: an inference epilogue y = relu?(acc * scale[c] + shift[c]) (scale = shift = NULL: none) -- an eval-mode
 * BatchNorm (+ ReLU) folded behind the convolution -- and, for the deconv / strided entry points, the placement of the
 * result: `ldc` floats between output pixels, from channel `coff` of a pixel on (ldc = 0: dense). */
typedef struct glx_bn_stats {
  void* state;
  const float* gamma;
  const float* beta;
  float eps, momentum;
  float* coef;
  float* save_mean;
  float* save_invstd;
  float* running_mean;
  float* running_var;
  int64_t count;   /* glx_sconv_forward_ex only: > 0 = elements per channel the statistics stand for when that is more than the
                    * output rows (the rest are zeros nobody stores: BaseBEVBackbone's first layer run as a sparse convolution,
                    * whose BatchNorm2d counts every pixel of the dense map); 0: the live output rows */
} glx_bn_stats;
typedef struct glx_epilogue {
  const float* scale;
  const float* shift;
  int relu, ldc, coff;
} glx_epilogue;
/* Backward statistics of a BatchNorm (+ ReLU) in the epilogue of the INPUT-GRADIENT convolution that produces the gradient
 * of its output (glx_conv3x3_forward_ex / glx_sconv_forward_ex on the adjoint filters; for the sparse form y has the rows of the gradient): the epilogue masks the gradient with the ReLU (re-derived
 * from y * scale + shift), writes dz, and takes sum dz and sum dz * xhat per channel; the last block writes coef (3 C floats:
 * a = gamma * invstd, b = mean dz, cc = mean dz * xhat -- what glx_bn_backward_apply reads), dgamma and dbeta.
 * y: the raw output (B, H, W, C) of the convolution in front of the BatchNorm; coef_fwd: its scale[C], shift[C];
 * mean / invstd: the saved batch statistics; state: as glx_bn_stats. */
typedef struct glx_bn_bwd_stats {
  void* state;
  const float* y;
  const float* coef_fwd;
  const float* mean;
  const float* invstd;
  const float* gamma;
  float* coef;
  float* dgamma;
  float* dbeta;
} glx_bn_bwd_stats;
/* tile_map: glx_sconv_tile_map of the rule table (NULL = built-in map); profile_start / profile_stop: two HIP events
 * that bracket the MFMA launch of the call (hipExtLaunchKernelGGL: kernel-only duration for bench.py's roofline). */
typedef struct glx_sconv_opts {
  const int32_t* tile_map;
  const glx_bn_stats* bn;
  void* profile_start;
  void* profile_stop;
  const glx_bn_bwd_stats* bn_bwd;   /* see glx_bn_bwd_stats; excludes bn and the bias / scale / shift / relu arguments.  NULL: none */
  /* transform of the INPUT rows on load, x' = relu(x * scale[c] + shift[c]) (scale, shift: Cin device floats, relu = 1, ldc =
   * coff = 0): the training-mode BatchNorm1d + ReLU of the layer in front (spconv_backbone.py:21-25 post_act_block) applied
   * without writing the normalised features; MFMA tile kernels with Cin >= 16 in one launch.  NULL: none. */
  const glx_epilogue* prologue;
} glx_sconv_opts;
typedef struct glx_conv_opts {
  const glx_bn_stats* bn;
  const glx_epilogue* epilogue;
  /* glx_conv3x3_forward_ex only: transform of the INPUT on load, x' = relu?(x * scale[c] + shift[c]) at the pixels of the map
   * (the zero padding stays zero; scale, shift: Cin device floats, ldc = coff = 0) -- the training-mode BatchNorm (+ ReLU) of
   * the layer in front applied without writing the normalised map (base_bev_backbone.py:36-49).  NULL: none. */
  const glx_epilogue* prologue;
  /* glx_conv3x3_forward_ex only, excludes the three above: see glx_bn_bwd_stats.  NULL: none. */
  const glx_bn_bwd_stats* bn_bwd;
} glx_conv_opts;

/* ROCm 7.2 workaround (csrc/glx_graph.hip): a hipMemsetAsync recorded into a HIP graph is replayed with a stale pattern from
 * the graph's second launch on (tools/graph_memset_repro.py).  Replaces every memset node of a CAPTURED, not yet instantiated
 * hipGraph_t by a kernel node doing the same fill (same dependencies and dependents); *n_replaced (host, may be NULL) =
 * how many.  No reference counterpart: the reference records no graphs. */
int glx_graph_replace_memsets(void* graph, int* n_replaced);

/* HIP events for kernel-only timing (bench.py); glx_event_elapsed_ms blocks until `stop`. */
int glx_event_create(void** event);
int glx_event_destroy(void* event);
int glx_event_elapsed_ms(void* start, void* stop, float* ms);

/* ------------------------------------------------------------------------------------
 * Cell index ("rank dictionary"): one bit per grid cell + per-word exclusive popcount.
 * rank(cell) = prefix[word] + popc(bits below) enumerates active cells in ascending
 * linear order (b, z, y, x); it replaces spconv's hash table for rule generation and
 * the dense (B,Z,Y,X) int32 voxel->point map of pcdet/utils/common_utils.py:226-243.
 * ------------------------------------------------------------------------------------ */

/* words = ceil(B*D*H*W / 64).  bitmap: words*8 bytes, prefix: words*4 bytes, chunk_flags:
 * ceil(words/8) bytes (one occupancy byte per 512 cells: scans and enumerations skip empty
 * chunks; prefix[] is only defined for words of occupied chunks). */
int64_t glx_index_words(int B, int D, int H, int W);
size_t glx_index_workspace_bytes(int B, int D, int H, int W);

/* Build the index of N active cells `indices` (N,4).  Outputs:
 *   bitmap[words] u64, prefix[words] i32, rank_to_row[N] i32, row_to_rank[N] i32,
 *   n_unique (device int32[1]): number of distinct cells (== N iff no duplicates),
 *   status (device int32[1]): set non-zero if any index is outside the grid.
 * Replaces: spconv indice hash build (call sites spconv_backbone.py:78-114). */
int glx_index_build(const int32_t* indices, int N, int B, int D, int H, int W,
                    uint64_t* bitmap, uint8_t* chunk_flags, int32_t* prefix, int32_t* rank_to_row,
                    int32_t* row_to_rank, int32_t* n_unique, int32_t* status,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Submanifold rule table: nbr[j*K + k] = input row at offset k of output row j, or -1.
 * K = kd*kh*kw, k = (kz*kh + ky)*kw + kx, neighbour cell = cell(j) + (k - ksize/2).
 * pair_count: NULL, or device int32[1] (caller zeroes it) accumulating the number of valid
 * pairs R with one atomic per wave (slow: diagnostics only).
 * Replaces: spconv SubMConv3d indice-pair generation (spconv_backbone.py:12,78,85). */
int glx_rules_subm(const int32_t* indices, int N, int B, int D, int H, int W,
                   const uint64_t* bitmap, const int32_t* prefix, const int32_t* rank_to_row,
                   int kd, int kh, int kw, int32_t* nbr, int32_t* pair_count,
                   const int32_t* n_live, void* stream);

/* ... with dilation (dd, dh, dw): neighbour of offset k at cell + (k - K/2) * d (spconv's SubMConv3d(dilation=...);
 * glx_rules_subm = dilation 1). */
int glx_rules_subm_dilated(const int32_t* indices, int N, int B, int D, int H, int W,
                           const uint64_t* bitmap, const int32_t* prefix, const int32_t* rank_to_row,
                           int kd, int kh, int kw, int dd, int dh, int dw, int32_t* nbr, int32_t* pair_count,
                           const int32_t* n_live, void* stream);

/* Strided (regular) sparse conv output set: marks every output cell reached by >=1 active
 * input in out_bitmap, scans it, writes n_out (device int32[1]).  Output rows are
 * enumerated in ascending linear (b,z,y,x) order.  Out grid = (B, oD, oH, oW).
 * Replaces: spconv SparseConv3d output index generation (spconv_backbone.py:14,90,97,104,113). */
int glx_outset_build(const int32_t* indices_in, int N_in, int B, int D, int H, int W,
                     const int32_t* in_rank_to_row, int kd, int kh, int kw, int sd, int sh, int sw,
                     int pd, int ph, int pw, int oD, int oH, int oW, uint64_t* out_bitmap,
                     uint8_t* out_chunk_flags, int32_t* out_prefix, int32_t* n_out,
                     const int32_t* n_in_live, void* workspace, size_t workspace_bytes,
                     void* stream);
/* ... with dilation: output o is reached from input cell c through offset k where o * s - p + k * d == c
 * (spconv's SparseConv3d(dilation=...); the callers compute oD/oH/oW = (size + 2 p - d (k - 1) - 1) / s + 1). */
int glx_outset_build_dilated(const int32_t* indices_in, int N_in, int B, int D, int H, int W,
                             const int32_t* in_rank_to_row, int kd, int kh, int kw, int sd, int sh, int sw,
                             int pd, int ph, int pw, int dd, int dh, int dw, int oD, int oH, int oW,
                             uint64_t* out_bitmap, uint8_t* out_chunk_flags, int32_t* out_prefix, int32_t* n_out,
                             const int32_t* n_in_live, void* workspace, size_t workspace_bytes, void* stream);
/* Decode the set bits of an index into (n,4) indices [b,z,y,x], ascending order; indices_out
 * has room for `capacity` rows, further cells are dropped (compare n_out with the capacity). */
int glx_outset_emit(const uint64_t* bitmap, const uint8_t* chunk_flags, const int32_t* prefix,
                    int B, int D, int H, int W, int capacity, int32_t* indices_out, void* stream);
/* Rule table of the strided conv: nbr[j*K+k] = input row at cell(j)*stride - pad + k, or -1.
 * in_rank_to_row may be NULL when input rows are already in ascending cell order; N_in = rows
 * of the input set (input rows >= N_in are treated as absent). */
int glx_rules_strided(const int32_t* indices_out, int N_out, int N_in, int B, int D, int H, int W,
                      const uint64_t* in_bitmap, const int32_t* in_prefix,
                      const int32_t* in_rank_to_row, int kd, int kh, int kw, int sd, int sh,
                      int sw, int pd, int ph, int pw, int32_t* nbr, int32_t* pair_count,
                      const int32_t* n_out_live, void* stream);
/* ... with dilation: nbr[j*K+k] = input row at cell(j)*stride - pad + k*dilation. */
int glx_rules_strided_dilated(const int32_t* indices_out, int N_out, int N_in, int B, int D, int H, int W,
                              const uint64_t* in_bitmap, const int32_t* in_prefix,
                              const int32_t* in_rank_to_row, int kd, int kh, int kw, int sd, int sh,
                              int sw, int pd, int ph, int pw, int dd, int dh, int dw, int32_t* nbr,
                              int32_t* pair_count, const int32_t* n_out_live, void* stream);
/* Input-major inverse: nbr_in[i*K+k] = output row j with nbr[j*K+k]==i, else -1.
 * n_out_live (device int32, may be NULL): live output rows of a shape-static rule set. */
int glx_rules_invert(const int32_t* nbr, int N_out, int K, int N_in, int32_t* nbr_in,
                     const int32_t* n_out_live, void* stream);

/* ------------------------------------------------------------------------------------
 * Sparse convolution  out[j,:] = sum_k in[nbr[j,k],:] @ W[k]   (+ bias)
 * W is (K, Cin, Cout) float32, i.e. spconv-1.x layout (kd,kh,kw,Cin,Cout) flattened
 * (pcdet/models/detectors/detector3d_template.py:377-384).  tile_order (optional, N_out)
 * lists output rows in spatial order for gather locality; NULL = identity.
 * Replaces: spconv SubMConv3d/SparseConv3d forward (spconv_backbone.py:148-156).
 * The same entry point computes input gradients: pass grad_out as `in`, nbr_in as `nbr`
 * and the transposed weights (glx_sconv_transpose_weights).
 * ------------------------------------------------------------------------------------ */
size_t glx_sconv_workspace_bytes(int K, int Cin, int Cout);
/* Bytes of the MFMA-fragment-ordered copy of W (0 when (Cin,Cout) has no MFMA kernel, i.e.
 * channels outside {16,32,64,128} or K > 27: those run the scalar kernel on raw W). */
size_t glx_sconv_packed_bytes(int K, int Cin, int Cout);
/* Work-balanced block -> tile map of a rule table for the 64-row tile kernels: tile_map
 * (int32[ceil(N_out/64)]) is a permutation of the tiles such that the blocks the hardware places on
 * one CU (b, b+256, b+512 ...) carry similar numbers of 16-pair MFMA chunks.  Results do not depend
 * on it (tiles are independent).  Handed to a convolution as glx_sconv_opts.tile_map.  Built once per rule table,
 * shared by the convs of an indice_key. */
size_t glx_sconv_tile_map_workspace_bytes(int N_out);
int glx_sconv_tile_map(const int32_t* nbr, const int32_t* tile_order, int N_out, int K,
                       const int32_t* n_out_live, int32_t* tile_map, void* workspace,
                       size_t workspace_bytes, void* stream);
/* Arithmetic of the block kernels behind glx_sconv_forward[_ex] for channels Cin in {32, 64, 128} -> Cout in {64, 128} (every other
 * layer multiplies in fp32 either way), per process: 1 = f16x2 (default: two fp16 pieces per operand -- the packed filter scaled by
 * one power of two, every gathered row by its own -- and three v_mfma_f32_16x16x32_f16 per product tile, >= 20.4 bits per product,
 * fp32 sums), 0 = fp32 (v_mfma_f32_16x16x4_f32, exact products).  env GLX_SCONV_ARITH=fp32 selects 0 at load.  A packed filter
 * holds the images of both, so the switch needs no re-pack.  spconv's SubMConv3d / SparseConv3d (spconv_backbone.py:30-75). */
int glx_sconv_set_arith(int f16x2);
int glx_sconv_get_arith(void);
/* Re-order W (K,Cin,Cout) into Wp (glx_sconv_packed_bytes); do it once per weight update. */
int glx_sconv_pack_weights(const float* W, int K, int Cin, int Cout, float* Wp, void* stream);
/* Same for the ADJOINT conv, straight from the forward weights: (Cin, Cout) are the dimensions of
 * the conv that will run; transposed != 0: W is stored (K, Cout, Cin) (the weights of the conv this
 * one is the input gradient of); flip_taps != 0: tap k reads W[K-1-k] (input gradient of a
 * submanifold conv on its own rule table).  One launch instead of flip + transpose + copy + pack. */
int glx_sconv_pack_weights_view(const float* W, int K, int Cin, int Cout, int transposed,
                                int flip_taps, float* Wp, void* stream);
/* n weight images in one launch (a training step packs every layer's forward and adjoint image: 24 launches
 * otherwise); arrays of length n on the HOST, job i as glx_sconv_pack_weights_view(W[i], K[i], Cin[i], Cout[i],
 * transposed[i], flip_taps[i], Wp[i]). */
int glx_sconv_pack_weights_multi(int n, const float* const* W, const int32_t* K, const int32_t* Cin,
                                 const int32_t* Cout, const int32_t* transposed, const int32_t* flip_taps,
                                 float* const* Wp, void* stream);
/* W: raw weights (may be NULL when Wp is given); Wp: packed weights or NULL (then W is packed
 * into `workspace`, >= glx_sconv_workspace_bytes).  Fused pointwise tail on the output tile:
 * y = relu?((acc + bias) * scale + shift); bias/scale/shift are (Cout) or NULL -- this is how
 * an eval-mode BatchNorm1d + ReLU that follows the conv (spconv_backbone.py:21-25) is folded. */
int glx_sconv_forward(const float* in, int N_in, const float* W, const float* Wp,
                      const float* bias, const float* scale, const float* shift, int relu,
                      const int32_t* nbr, const int32_t* tile_order, int N_out, int K, int Cin,
                      int Cout, float* out, const int32_t* n_out_live, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same with per-call options (tile map of the rule table, training-mode BatchNorm statistics in the epilogue -- MFMA
 * tile kernels, one launch --, profiling events): see glx_sconv_opts above. */
int glx_sconv_forward_ex(const float* in, int N_in, const float* W, const float* Wp,
                         const float* bias, const float* scale, const float* shift, int relu,
                         const int32_t* nbr, const int32_t* tile_order, int N_out, int K, int Cin,
                         int Cout, float* out, const int32_t* n_out_live, void* workspace,
                         size_t workspace_bytes, const glx_sconv_opts* opts, void* stream);
/* Reference-quality scalar kernel (any channel count); used for tiny Cin and as a
 * device-side cross-check of the MFMA kernel. */
int glx_sconv_forward_generic(const float* in, int N_in, const float* W, const float* bias,
                              const int32_t* nbr, int N_out, int K, int Cin, int Cout,
                              float* out, void* stream);
/* Wt[k][co][ci] = W[k][ci][co]. */
int glx_sconv_transpose_weights(const float* W, int K, int Cin, int Cout, float* Wt,
                                void* stream);
/* Weight gradient dW[k] = sum_j in[nbr[j,k],:]^T @ grad_out[j,:]   (K, Cin, Cout).
 * Replaces: spconv backward wgrad (autograd of spconv_backbone.py convs). */
size_t glx_sconv_wgrad_workspace_bytes(int N_out, int K, int Cin, int Cout);
int glx_sconv_wgrad(const float* in, int N_in, const float* grad_out, const int32_t* nbr,
                    int N_out, int K, int Cin, int Cout, float* dW, const int32_t* n_out_live,
                    int submanifold, void* workspace, size_t workspace_bytes, void* stream);
/* submanifold != 0: nbr is a submanifold table (its centre offset K/2 pairs every row): the centre
 * gets shorter row slices so that its blocks do not outlast the others. */

/* Per-offset pair lists of a rule table -- spconv's "indice pairs" (the (K, 2, N) indice_pairs + indice_pair_num that
 * spconv.ops.get_indice_pairs returns and the reference's convs carry in indice_dict, spconv_backbone.py:77-117): for every
 * offset k the (input row, output row) pairs with nbr[j, k] >= 0 in ascending j.  Built once per table (two launches), shared
 * by every weight gradient over it.  `lists` is an opaque device buffer of glx_pair_lists_bytes(N_out, K) bytes; rows
 * >= *n_live (device, may be NULL) are ignored. */
size_t glx_pair_lists_bytes(int N_out, int K);
int glx_pair_lists_build(const int32_t* nbr, int N_out, int K, const int32_t* n_live, void* lists,
                         size_t lists_bytes, void* stream);
/* Weight gradient (as glx_sconv_wgrad) over pair lists: every offset's list is cut into equal chunks sized on the device
 * from the table's pair count, a block per chunk, slabs summed per offset in chunk order (bitwise reproducible).
 * N_out / K: the values the lists were built with. */
size_t glx_sconv_wgrad_pairs_workspace_bytes(int N_out, int K, int Cin, int Cout);
int glx_sconv_wgrad_pairs(const float* in, const float* grad_out, const void* lists, int N_out, int K,
                          int Cin, int Cout, float* dW, void* workspace, size_t workspace_bytes, void* stream);
/* The arithmetic glx_sconv_wgrad_pairs[_ex] uses for these channels: 1 = the f16 x 2 form (glx_sconv_set_arith's arithmetic with
 * the pairs as contraction index: a 32-pair panel scaled by the block's two maxima, running exponent over a chunk; taken where it
 * measured faster -- Cin >= 64 and Cout >= 64 except 128 -> 64; env GLX_SCONV_WGRAD_F16=0 never, 2 wherever Cin, Cout are
 * multiples of 32), 0 = fp32 MFMAs. */
int glx_sconv_wgrad_arith(int Cin, int Cout);
/* The same for input rows that are transformed on load (`pre` as glx_sconv_opts.prologue; NULL = glx_sconv_wgrad_pairs). */
int glx_sconv_wgrad_pairs_ex(const float* in, const float* grad_out, const void* lists, int N_out, int K, int Cin, int Cout,
                             float* dW, const glx_epilogue* pre, void* workspace, size_t workspace_bytes, void* stream);
/* The two launches apart: glx_sconv_wgrad_pairs with dW == NULL leaves the chunk products in `workspace` (which the caller
 * then keeps), glx_sconv_wgrad_pairs_reduce sums them into dW -- a leaf of the backward pass that a training step can run
 * later on another stream (the sum is a pass over memory, the products fill the matrix pipe: they overlap). */
int glx_sconv_wgrad_pairs_reduce(const void* lists, int N_out, int K, int Cin, int Cout, float* dW, const void* workspace,
                                 size_t workspace_bytes, void* stream);
/* The sums of n layers in ONE launch (arrays of n: a training step keeps every layer's chunk products in a buffer of its own and
 * runs the sums at the end of its backward pass -- thirteen ~8 us launches on the step's main chain become one).  Per element the
 * same sums in the same order as glx_sconv_wgrad_pairs_reduce.  spconv's SparseConvolution weight gradient (ops.py indice_conv_backward). */
int glx_sconv_wgrad_pairs_reduce_multi(int n, const void* const* lists, const int32_t* N_out, const int32_t* K, const int32_t* Cin,
                                       const int32_t* Cout, float* const* dW, const void* const* workspace,
                                       const size_t* workspace_bytes, void* stream);

/* SparseConvTensor.dense(): out (B, C, D, H, W) must be zero-filled by the caller.
 * Replaces: spconv dense() (height_compression.py:21). */
int glx_dense_scatter(const float* features, const int32_t* indices, int N, int C, int B,
                      int D, int H, int W, float* out, const int32_t* n_live, void* stream);

/* Adjoint of dense(): grad_features (N, C) gathered from grad_dense (B, C, D, H, W) at the rows'
 * cells; rows >= *n_live (shape-static tensors) are left untouched.
 * Replaces: the autograd of spconv's dense() (advanced indexing in torch). */
int glx_dense_gather(const float* grad_dense, const int32_t* indices, int N, int C, int B, int D,
                     int H, int W, float* grad_features, const int32_t* n_live, void* stream);

/* Same result without the caller's zero fill: every element of out is written once, rows are
 * found through the tensor's cell index (bitmap / prefix / rank_to_row or NULL when rows are in
 * cell order); N = rows of `features` (cells whose row is >= N read as empty). */
int glx_dense_from_index(const float* features, int N, int C, const uint64_t* bitmap,
                         const int32_t* prefix, const int32_t* rank_to_row, int B, int D, int H,
                         int W, float* out, void* stream);
/* The same tensor as HeightCompression's view (B, C * D, H, W) of dense() (height_compression.py:21-25), written
 * in channels-last memory: out (B, H, W, C * D), channel index c * D + z, C * D a multiple of 4.  glx_dense_gather_nhwc
 * is its adjoint (grad_features[row, c] = grad[b, y, x, c * D + z]), honouring n_live like glx_dense_gather. */
int glx_dense_from_index_nhwc(const float* features, int N, int C, const uint64_t* bitmap, const int32_t* prefix,
                              const int32_t* rank_to_row, int B, int D, int H, int W, float* out, void* stream);
int glx_dense_gather_nhwc(const float* grad_dense, const int32_t* indices, int N, int C, int B, int D, int H, int W,
                          float* grad_features, const int32_t* n_live, void* stream);

/* ------------------------------------------------------------------------------------
 * Voxelization.
 * ------------------------------------------------------------------------------------ */

/* Hard voxelization of a stacked batch of point clouds, on device, with the exact
 * first-seen ordering of the CPU generators the reference calls
 * (pcdet/datasets/processor/data_processor.py:15-60 -> spconv.utils.VoxelGeneratorV2 /
 * Point2VoxelCPU3d; third-party, restated in oracle/):
 *   per point c = floor((p - range_min) / voxel_size) per axis (fp32), dropped when outside
 *   [0, grid); a cell becomes voxel #v at its first point in input order, while fewer than
 *   max_voxels voxels exist in that frame; each voxel keeps its first max_points points.
 * points (P, C) fp32 with xyz in columns 0..2; point_batch (P) int32 frame id in [0,B) or
 * NULL when B == 1; frames must be stacked contiguously (frame ids non-decreasing).
 * vrange = {xmin,ymin,zmin,xmax,ymax,zmax}, vsize = {vx,vy,vz} are HOST arrays; grid
 * (gx,gy,gz) cells.
 * Outputs (capacity B*max_voxels rows, rows of frame b start at voxel_offset[b]):
 *   voxels (cap, max_points, C) zero padded, coords (cap, 4) int32 [b,z,y,x],
 *   num_points (cap) int32, voxel_offset (B+1) int32 (device; [B] = total voxels). */
size_t glx_voxelize_hard_workspace_bytes(int P, int B, int gx, int gy, int gz, int max_points,
                                         int max_voxels);
int glx_voxelize_hard(const float* points, const int32_t* point_batch, int P, int C, int B,
                      const float* vrange, const float* vsize, int gx, int gy, int gz,
                      int max_points, int max_voxels, float* voxels, int32_t* coords,
                      int32_t* num_points, int32_t* voxel_offset,
                      /* optional: also hand out the voxelizer's cell bitmap as the cell index of
                       * the sparse tensor (B, index_depth, gy, gx), index_depth in {gz, gz+1};
                       * idx_rank_to_row has P entries (-1 for cells dropped by max_voxels),
                       * idx_n_unique device int32[1].  Pass 0 / NULLs to skip. */
                      int index_depth, uint64_t* idx_bitmap, uint8_t* idx_flags,
                      int32_t* idx_prefix, int32_t* idx_rank_to_row, int32_t* idx_n_unique,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Device data step in front of the voxelizer: DataProcessor.mask_points_and_boxes_outside_range +
 * shuffle_points (pcdet/datasets/processor/data_processor.py:78-105, common_utils.py:60-63) on a capacity-sized
 * stacked buffer, shape-static and free of library calls (capturable).  point_batch: non-decreasing frame ids, B =
 * padding.  A point is kept when its frame id is valid and x / y lie in the CLOSED range range_xy = {x0, y0, x1, y1}
 * (HOST array).  Output: the kept points of frame 0, then frame 1, ... each frame in a pseudo-random order (shuffle
 * != 0: a keyed bijection of the frame's kept points, new for every call -- seed_and_calls = DEVICE uint64[2] {seed,
 * number of calls so far}, the call increments the second; shuffle == 0: the original order), then rows with frame id
 * B and zeros.  order (P, optional): source row of every output row, -1 for padding.  Not in place. */
size_t glx_mask_shuffle_workspace_bytes(int P, int B);
int glx_mask_shuffle(const float* points, const int32_t* point_batch, int P, int C, int B, const float* range_xy,
                     int shuffle, uint64_t* seed_and_calls, float* out_points, int32_t* out_batch, int32_t* order,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Dynamic voxelization + per-voxel mean, semantics of DynamicMeanVFE.forward
 * (pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:53-72): voxels enumerate in ascending
 * key b*XYZ + x*YZ + y*Z + z; features = mean over ALL points of the cell (columns 0..C-1).
 * Outputs capacity P rows: features (P, C), coords (P,4) [b,z,y,x], n_voxels device int32[1]. */
size_t glx_voxelize_dynamic_workspace_bytes(int P, int B, int gx, int gy, int gz);
int glx_voxelize_dynamic_mean(const float* points, const int32_t* point_batch, int P, int C,
                              int B, const float* vrange, const float* vsize, int gx, int gy,
                              int gz, float* features, int32_t* coords, int32_t* n_voxels,
                              void* workspace, size_t workspace_bytes, void* stream);

/* MeanVFE.forward (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31):
 * out[v,:] = sum_p voxels[v,p,:] / max(num_points[v], 1). */
int glx_mean_vfe(const float* voxels, const int32_t* num_points, int Nv, int max_points, int C,
                 float* out, const int32_t* n_live, void* stream);

/* ------------------------------------------------------------------------------------
 * Rotated boxes: BEV overlap / IoU, NMS, GLENet variance-voting NMS.
 * ------------------------------------------------------------------------------------ */

/* (N,7) x (M,7) -> (N,M) BEV overlap area (iou == 0) or IoU (iou != 0).
 * Replaces: iou3d_nms_cuda.boxes_overlap_bev_gpu / boxes_iou_bev_gpu
 * (pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:49-88, iou3d_nms_kernel.cu:236-260) and
 * boxes_iou_bev_cpu (iou3d_cpu.cpp:232-252). */
int glx_boxes_overlap_bev(const float* boxes_a, int N, const float* boxes_b, int M, int iou,
                          float* out, void* stream);
/* Same for the older `iou3d` library: boxes (N,5) [x1,y1,x2,y2,ry]
 * (pcdet/ops/iou3d/src/iou3d.cpp:31-96, iou3d_kernel.cu:270-307). */
int glx_iou3d_boxes_overlap_bev(const float* boxes_a, int N, const float* boxes_b, int M, int iou,
                                float* out, void* stream);
/* out[i] = overlap(a[i], b[i]); replaces iou3d_cuda.boxes_aligned_overlap_bev_gpu
 * (iou3d.cpp:55-73, iou3d_kernel.cu:284-293). */
int glx_iou3d_boxes_aligned_overlap_bev(const float* boxes_a, const float* boxes_b, int N,
                                        float* out, void* stream);

/* Rotated (normal == 0) or axis-aligned (normal != 0) NMS on boxes ALREADY sorted by score.
 * keep (N) int64 and num_out int32[1] are DEVICE buffers: keep[0..num_out) = indices of the
 * kept boxes in ascending order.  No host round trip: the suppression matrix (upper triangle
 * only) stays in `workspace` and is swept on the device.
 * Replaces: iou3d_nms_cuda.nms_gpu / nms_normal_gpu (iou3d_nms.cpp:90-186, which copy the
 * matrix to the host and sweep it serially there). */
size_t glx_nms_workspace_bytes(int N);
int glx_nms(const float* boxes_sorted, int N, float thresh, int normal, int64_t* keep,
            int32_t* num_out, void* workspace, size_t workspace_bytes, void* stream);
/* `frames` independent lists of N boxes in one launch sequence: boxes_sorted (frames,N,7), keep
 * (frames,N), num_out (frames); workspace >= frames * glx_nms_workspace_bytes(N).  max_keep > 0
 * ends a frame's sweep once that many boxes are kept (post-NMS truncation, e.g. NMS_POST_MAXSIZE of
 * class_agnostic_nms, pcdet/models/model_utils/model_nms_utils.py:6-24): num_out may then be any
 * value >= max_keep and keep[f, 0..max_keep) is the prefix the full sweep would produce. */
int glx_nms_batch(const float* boxes_sorted, int frames, int N, float thresh, int normal,
                  int max_keep, int64_t* keep, int32_t* num_out, void* workspace,
                  size_t workspace_bytes, void* stream);

/* GLENet's variance-voting NMS loop (nms_func, pcdet/ops/iou3d_nms/iou3d_nms_utils.py:227-273)
 * on the device.  boxes (N,7) and scores (N) are updated in place; variance (N, var_stride>=7)
 * may be NULL; ious_t (N,N) = TRANSPOSED BEV IoU matrix of the ORIGINAL boxes (ious_t[j][i] =
 * IoU(box i, box j): a round reads the chosen box's column of the reference's matrix, i.e. one
 * contiguous row here); scratch: N*8 floats (required with variance); N <= 4096. */
int glx_nms_vote(float* boxes, float* scores, const float* variance, int var_stride,
                 const float* ious_t, int N, float iou_thr, float score_thr, float* scratch,
                 void* stream);
/* Test / bench helper, no reference counterpart: rois (B,R,ld >= 7) and roi_labels (B,R) int64 -- slot g < G of frame b
 * takes gt_boxes[b][g][0:7] + offset7 and the row's class wherever gt_boxes[b][g][7] > 0 (gt_boxes (B,G,gld >= 8)): what a
 * trained first stage delivers (a random-init one proposes nothing near the ground truth). */
int glx_seed_rois(float* rois, int64_t* roi_labels, const float* gt_boxes, const float* offset7, int B, int R,
                  int roi_ld, int G, int gt_ld, void* stream);
/* `frames` independent lists in one launch sequence (the per-frame loop of Detector3DTemplate.post_processing,
 * pcdet/models/detectors/detector3d_template.py:196-309): boxes (frames,N,7), scores (frames,N), variance
 * (frames,N,var_stride), ious_t (frames,N,N), scratch (frames,N*8); counts (frames) int32 DEVICE or NULL = live boxes
 * per frame (<= N, the rest of a frame's rows is padding and is neither read nor written). */
int glx_nms_vote_batch(float* boxes, float* scores, const float* variance, int var_stride,
                       const float* ious_t, int frames, int N, const int32_t* counts, float iou_thr,
                       float score_thr, float* scratch, void* stream);
/* boxes (frames,N,7) -> out (frames,N,N) BEV IoU of every frame's boxes with themselves; transposed != 0:
 * out[f][j][i] = IoU(box i, box j) (the layout glx_nms_vote reads), else out[f][i][j]; counts as above.
 * Replaces: boxes_bev_iou_cpu(boxes, boxes) of nms_func (iou3d_nms_utils.py:235) per frame. */
int glx_boxes_iou_bev_self_batch(const float* boxes, int frames, int N, const int32_t* counts, int transposed,
                                 float* out, void* stream);
/* Inference post-processing around the top-k and the voting NMS (detector3d_template.py:179-317,
 * model_nms_utils.py:6-62).  glx_det_candidates: order (F,K) int64 = top-k indices into a frame's R boxes, counts (F)
 * = boxes that passed SCORE_THRESH; writes cand (F,K,7) with the heading wrapped to [-pi,pi) (iou3d_nms_utils.py:
 * 212-214) and var (F,K,7) = exp(std_preds) (model_nms_utils.py:18; var/std_preds may be NULL), padding slots zero / one.
 * glx_det_gather: new_scores (F,K) after glx_nms_vote_batch, top (F,K) the scores before it, cand the voted boxes;
 * survivors (new score > 0) in slot order, first P, original score > post_thr when use_post -> out_boxes (F,P,7),
 * out_scores (F,P), out_labels (F,P) = labels[f][source] (labels (F,R) int64 or NULL -> 1), out_index (F,P) source box
 * or -1, out_num (F) int32. */
int glx_det_candidates(const float* box_preds, const float* std_preds, int ld_box, int ld_std,
                       const int64_t* order, const int32_t* counts, int F, int R, int K, float* cand,
                       float* var, void* stream);
int glx_det_gather(const float* new_scores, const float* top, const float* cand, const int64_t* order,
                   const int64_t* labels, const int32_t* counts, int F, int R, int K, int P, float post_thr,
                   int use_post, float* out_boxes, float* out_scores, int64_t* out_labels,
                   int64_t* out_index, int32_t* out_num, void* stream);

/* ------------------------------------------------------------------------------------
 * Point / box operators.
 * ------------------------------------------------------------------------------------ */

/* boxes (B,T,7), pts (B,P,3) -> (B,P) index of the first box containing the point or -1.
 * Replaces: roiaware_pool3d_cuda.points_in_boxes_gpu (roiaware_pool3d.cpp:100-118). */
int glx_points_in_boxes(const float* boxes, const float* pts, int B, int T, int P,
                        int32_t* box_idx_of_points, void* stream);

/* boxes (N,7), pts (P,3) -> out (N,P) 0/1 inside flags with the given xy MARGIN (1e-2 for the
 * reference's CPU variant).  Replaces: roiaware_pool3d_cuda.points_in_boxes_cpu
 * (roiaware_pool3d.cpp:143-168). */
int glx_points_in_boxes_mask(const float* boxes, int N, const float* pts, int P, float margin,
                             int32_t* out, void* stream);

/* RoI-aware pooling.  pts_idx_of_voxels (N,ox,oy,oz,max_pts) i32 must arrive zero-filled (the reference's
 * buffer contents: count, then point indices, then zeros).  pooled (N,ox,oy,oz,C) is written in full (0 for
 * an empty voxel); argmax (N,ox,oy,oz,C) i32 is written in full by max pooling (-1 for an empty voxel)
 * and left untouched by avg pooling (the reference leaves its zeros).  pool_method 0 = max, 1 = avg.
 * Replaces: roiaware_pool3d_cuda.forward / backward (roiaware_pool3d.cpp:29-98). */
int glx_roiaware_pool3d_forward(const float* rois, int N, const float* pts, int P,
                                const float* pts_feature, int C, int ox, int oy, int oz,
                                int max_pts, int pool_method, int32_t* argmax,
                                int32_t* pts_idx_of_voxels, float* pooled, void* stream);
int glx_roiaware_pool3d_backward(const int32_t* pts_idx_of_voxels, const int32_t* argmax,
                                 const float* grad_out, int N, int ox, int oy, int oz, int C,
                                 int max_pts, int pool_method, float* grad_in, void* stream);

/* RoI point pooling: xyz (B,Np,3), boxes3d (B,M,7) ALREADY enlarged, pts_feature (B,Np,C) ->
 * pooled (B,M,S,3+C) and empty_flag (B,M), both zero-filled by the caller.
 * Replaces: roipoint_pool3d_cuda.forward (roipoint_pool3d.cpp:25-55). */
int glx_roipoint_pool3d(const float* xyz, const float* boxes3d, const float* pts_feature, int B,
                        int Np, int M, int C, int S, float* pooled, int32_t* empty_flag,
                        void* stream);

/* Voxel query over the dense (B,Z,Y,X) voxel->point map: the first nsample cells of the
 * (2r+1)^3 window (z, y, x ascending) whose centre is within `radius`; unused slots repeat the
 * first hit; idx[m,0] == -1 marks an empty ball (the other slots of such a row are left as they
 * arrived -- the reference passes a zero-filled idx).
 * Replaces: pointnet2_stack_cuda.voxel_query_wrapper (voxel_query.cpp:28-44). */
int glx_voxel_query(int M, int Z, int Y, int X, int nsample, float radius, int z_range,
                    int y_range, int x_range, const float* new_xyz, const float* xyz,
                    const int32_t* new_coords, const int32_t* point_indices, int32_t* idx,
                    void* stream);
/* Same query against the sparse tensor's cell index instead of the dense map (saves building
 * and clearing the map of pcdet/utils/common_utils.py:226-243 every step). */
int glx_voxel_query_index(int M, int Z, int Y, int X, int nsample, float radius, int z_range,
                          int y_range, int x_range, const float* new_xyz, const float* xyz,
                          const int32_t* new_coords, const uint64_t* bitmap, const int32_t* prefix,
                          const int32_t* rank_to_row, int32_t* idx, void* stream);

/* Replaces: pointnet2_stack_cuda.ball_query_wrapper (ball_query.cpp:24-43). */
int glx_ball_query(int B, int M, float radius, int nsample, const float* new_xyz,
                   const int32_t* new_xyz_batch_cnt, const float* xyz,
                   const int32_t* xyz_batch_cnt, int32_t* idx, void* stream);
/* Replaces: pointnet2_stack_cuda.group_points_wrapper / group_points_grad_wrapper
 * (group_points.cpp:31-69); grad_features (N,C) arrives zero-filled. */
int glx_group_points(int B, int M, int C, int nsample, const float* features,
                     const int32_t* features_batch_cnt, const int32_t* idx,
                     const int32_t* idx_batch_cnt, float* out, void* stream);
int glx_group_points_grad(int B, int M, int C, int N, int nsample, const float* grad_out,
                          const int32_t* idx, const int32_t* idx_batch_cnt,
                          const int32_t* features_batch_cnt, float* grad_features, void* stream);
/* Same gradient in gather form (every feature row sums its own references; no float atomics):
 * the right choice when many grid points share few rows (RoI-grid pooling).  grad_features (N,C) is
 * written completely (rows without references get 0); workspace >= the query below. */
size_t glx_group_points_grad_workspace_bytes(int M, int N, int nsample);
/* Row-major grouping on the voxel query's own output: idx (M,nsample) GLOBAL rows, idx[m,0] < 0 =
 * empty ball; out (M, nsample, C) = features[idx] (zeros for an empty ball).  glx_group_rows_grad:
 * grad_out (M, nsample, C) -> grad_features (N, C), gather form, same workspace query. */
int glx_group_rows(const float* features, const int32_t* idx, int M, int nsample, int C, float* out,
                   void* stream);
int glx_group_rows_grad(const float* grad_out, const int32_t* idx, int M, int nsample, int C, int N,
                        float* grad_features, void* workspace, size_t workspace_bytes, void* stream);
int glx_group_points_grad_gather(int B, int M, int C, int N, int nsample, const float* grad_out,
                                 const int32_t* idx, const int32_t* idx_batch_cnt,
                                 const int32_t* features_batch_cnt, float* grad_features,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* RoI-grid pooling aggregation of one scale after its voxel query, inference: out[m,:] =
 * relu(Wout . max_s relu(feats[idx[m,s],:] + Wpos . (xyz[idx[m,s]] - new_xyz[m]) + bpos) + bout);
 * empty[m] != 0 (or, with empty == NULL, idx[m,0] < 0 as glx_voxel_query leaves it): the ball is
 * empty (features and offsets count as 0).  idx (M,nsample) indexes the
 * stacked feats (N,Cm) / xyz (N,3) GLOBALLY; Wpos (Cm,3), Wout (Co,Cm) with the eval-mode
 * BatchNorms folded; Cm, Co <= 64.
 * Replaces: group_points x2 + mask + pos-MLP + ReLU + max-pool + out-MLP of
 * NeighborVoxelSAModuleMSG.forward (pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:88-108). */
int glx_voxel_pool_agg(const float* feats, const float* xyz, const float* new_xyz, const int32_t* idx,
                       const uint8_t* empty, int M, int nsample, int Cm, int Co, const float* Wpos,
                       const float* bpos, const float* Wout, const float* bout, float* out,
                       void* stream);

/* RoI-grid pooling of a whole stage, inference (VoxelRCNNHead.roi_grid_pool,
 * pcdet/models/roi_heads/voxelrcnn_head.py:106-191), without the centres / coordinate / count
 * tensors the reference builds per scale.  range_min, voxel_size: HOST float[3] (x,y,z).
 *
 * glx_roi_grid_points: rois (n_rois, cols >= 7) [x,y,z,dx,dy,dz,heading,..] -> grid_xyz
 *   (n_rois*G^3, 3) and coords (n_rois*G^3, 4) int32 [frame, z, y, x], the STRIDE-1 voxel
 *   coordinates `(p - range_min) // voxel_size` (float floor division as ATen evaluates it);
 *   frame = roi / rois_per_frame.  Replaces get_global_grid_points_of_roi + lines 128-141.
 * glx_roi_grid_query: voxel query of one scale: coords are floor-divided by `stride`, voxel
 *   centres are rebuilt from the sparse tensor's indices (N,4) [b,z,y,x] as get_voxel_centers
 *   does ((i + 0.5) * voxel*stride + range_min).  Output as glx_voxel_query_index.
 * glx_roi_grid_agg: glx_voxel_pool_agg with the same on-the-fly centres, writing its Co columns
 *   into a wider row-major output (row pitch out_stride floats; pass out + column offset).
 * glx_voxel_centers: get_voxel_centers (pcdet/utils/common_utils.py:66-82) of a sparse tensor's indices
 *   (N,4) [b,z,y,x] at downsample factor `stride` -> xyz (N,3), rounded step by step as the tensor
 *   expression does; one launch (the training path of the RoI-grid pooling keeps the centres tensor). */
int glx_voxel_centers(const int32_t* indices, int N, int stride, const float* range_min,
                      const float* voxel_size, float* xyz, void* stream);
int glx_roi_grid_points(const float* rois, int n_rois, int cols, int rois_per_frame, int grid_size,
                        const float* range_min, const float* voxel_size, float* grid_xyz,
                        int32_t* coords, void* stream);
int glx_roi_grid_query(int M, int Z, int Y, int X, int nsample, float radius, int z_range,
                       int y_range, int x_range, const float* grid_xyz, const int32_t* coords,
                       int stride, const int32_t* indices, const float* range_min,
                       const float* voxel_size, const uint64_t* bitmap, const int32_t* prefix,
                       const int32_t* rank_to_row, int32_t* idx, void* stream);
/* The same query for several feature scales in one launch (same grid points / coords; Z, Y, X, ranges, stride, tensors and
 * output per scale; x_range <= 15): RoIGridPool's three scales (voxelrcnn_head.py:106-191) were a serial chain. */
typedef struct glx_roi_query {
  int Z, Y, X, nsample, z_range, y_range, x_range, stride;
  float radius;
  const int32_t* indices; const uint64_t* bitmap; const int32_t* prefix; const int32_t* rank_to_row;
  int32_t* idx;
} glx_roi_query;
int glx_roi_grid_query_multi(int n_scales, const glx_roi_query* scales, int M, const float* grid_xyz, const int32_t* coords,
                             const float* range_min, const float* voxel_size, void* stream);
int glx_roi_grid_agg(const float* feats, const int32_t* indices, int stride, const float* range_min,
                     const float* voxel_size, const float* grid_xyz, const int32_t* idx, int M,
                     int nsample, int Cm, int Co, const float* Wpos, const float* bpos,
                     const float* Wout, const float* bout, float* out, int out_stride, void* stream);

/* GLENet's KL regression loss of the RoI head and its gradients in one launch.
 * rcnn_reg, rcnn_reg_std, rois, gt_of_rois (RoI frame), gt_uncertainty (label variances): (R,7);
 * fg_mask (R) float, > 0 = foreground; code_weights: HOST float[7] or NULL (= 1).
 * out5 (device float[5]) = { loss, src part, square part, log part, #foreground };
 * grad_reg / grad_std (R,7) or NULL = d loss / d rcnn_reg, d loss / d rcnn_reg_std.
 * Replaces: VoxelRCNNKLLabelIoUHead.get_box_reg_layer_loss lines 96-138
 * (pcdet/models/roi_heads/voxelrcnn_kl_label_iou_head.py) incl. ResidualCoder.encode_torch
 * (box_coder_utils.py:13-43) and WeightedSmoothL1Loss (loss_utils.py:100-130). */
int glx_kl_reg_loss(const float* rcnn_reg, const float* rcnn_reg_std, const float* rois,
                    const float* gt_of_rois, const float* gt_uncertainty, const float* fg_mask, int R,
                    const float* code_weights, float beta, float weight, float* out5,
                    float* grad_reg, float* grad_std, void* stream);

/* Corner-loss regularisation of the same head (lines 148-172): foreground RoIs only; rois (R,7) in
 * the LiDAR frame, gt_of_rois_src (R,7) the matched ground-truth boxes in the LiDAR frame.
 * out2 (device float[2]) = { mean corner loss over the foreground RoIs * weight, #foreground };
 * grad_reg (R,7) or NULL = d loss / d rcnn_reg (zero rows for background).
 * Replaces: decode_torch + rotate_points_along_z + loss_utils.get_corner_loss_lidar
 * (pcdet/utils/loss_utils.py:210-233, box_utils.boxes_to_corners_3d) and their autograd. */
int glx_corner_loss(const float* rcnn_reg, const float* rois, const float* gt_of_rois_src,
                    const float* fg_mask, int R, float weight, float* out2, float* grad_reg,
                    void* stream);

/* Canonical transformation of the matched ground truth into the RoI frame: rois (R, roi_cols >= 7),
 * gt_of_rois (R, gt_cols >= 7) -> out (R, gt_cols): centre difference rotated by -heading(roi),
 * heading difference folded into [-pi/2, pi/2] (turned by pi when it points the other way), other
 * channels copied.  Replaces: RoIHeadTemplate.assign_targets lines 140-159
 * (pcdet/models/roi_heads/roi_head_template.py). */
int glx_roi_canonical_gt(const float* rois, int roi_cols, const float* gt_of_rois, int gt_cols, int R,
                         float* out, void* stream);

/* Anchor target assignment of the dense head for ONE anchor class and all B frames: anchors (N,7)
 * shared by the frames, gt_boxes (B, M <= 128, gt_cols >= 8) zero-padded, last column = class id
 * (1-based).  Outputs per frame: box_cls_labels (B,N) int32 {class_id, 0, -1}, box_reg_targets
 * (B,N,7), reg_weights (B,N).  No sampling (POS_FRACTION < 0), nearest-BEV IoU (match_height off).
 * Replaces: AxisAlignedTargetAssigner.assign_targets_single + the per-frame trimming / class mask of
 * assign_targets (pcdet/models/dense_heads/target_assigner/axis_aligned_target_assigner.py:36-213),
 * box_utils.boxes3d_nearest_bev_iou and ResidualCoder.encode_torch. */
size_t glx_assign_targets_workspace_bytes(int B, int N);
int glx_assign_targets(const float* anchors, int N, const float* gt_boxes, int B, int M, int gt_cols,
                       int class_id, float matched_threshold, float unmatched_threshold,
                       int norm_by_num_examples, int32_t* box_cls_labels, float* box_reg_targets,
                       float* reg_weights, void* workspace, size_t workspace_bytes, void* stream);

/* Loss of the dense (anchor) head and its gradients: cls_preds (B,A,num_class) logits, box_preds
 * (B,A,7), dir_preds (B,A,2) or NULL, box_cls_labels (B,A) int32 {class, 0, -1}, box_reg_targets
 * (B,A,7), anchors (A,7).  out4 (device) = { total, cls, loc, dir } already weighted; grad_* (same
 * shapes as the predictions, may be NULL) = d total / d prediction.  code_weights: HOST float[7]/NULL.
 * Replaces: AnchorHeadTemplate.get_cls_layer_loss / get_box_reg_layer_loss / get_loss
 * (pcdet/models/dense_heads/anchor_head_template.py:108-232) with SigmoidFocalClassificationLoss,
 * WeightedSmoothL1Loss and WeightedCrossEntropyLoss (pcdet/utils/loss_utils.py:7-207). */
/* Same with uncertainty_gt_index (B, N) int32 (or NULL): for every positive anchor the row of the frame's ground truth
 * whose label uncertainty it carries -- WeightedAxisAlignedTargetAssigner (GLENet-S / -C,
 * weighted_axis_aligned_target_assigner.py:145-173): the ground truth that forced the match, overwritten by the
 * arg-max ground truth when the overlap is over the matched threshold; -1 for the other anchors. */
int glx_assign_targets_ex(const float* anchors, int N, const float* gt_boxes, int B, int M, int gt_cols,
                          int class_id, float matched_threshold, float unmatched_threshold,
                          int norm_by_num_examples, int32_t* box_cls_labels, float* box_reg_targets,
                          float* reg_weights, int32_t* uncertainty_gt_index, void* workspace,
                          size_t workspace_bytes, void* stream);

size_t glx_rpn_loss_workspace_bytes(int B, int A);
int glx_rpn_loss(const float* cls_preds, const float* box_preds, const float* dir_preds,
                 const int32_t* box_cls_labels, const float* box_reg_targets, const float* anchors,
                 int B, int A, int num_class, int class_agnostic, float alpha, float beta,
                 const float* code_weights, float dir_offset, float cls_weight, float loc_weight,
                 float dir_weight, float* out4, float* grad_cls, float* grad_box, float* grad_dir,
                 void* workspace, size_t workspace_bytes, void* stream);

/* RoI classification loss (BinaryCrossEntropy on soft IoU labels, label < 0 = ignored): rcnn_cls (R)
 * logits, rcnn_cls_labels (R) float.  out2 (device) = { loss * weight, #valid }; grad (R) or NULL.
 * Replaces: RoIHeadTemplate.get_box_cls_layer_loss (pcdet/models/roi_heads/roi_head_template.py:246-272). */
int glx_rcnn_cls_loss(const float* rcnn_cls, const float* rcnn_cls_labels, int R, float weight,
                      float* out2, float* grad, void* stream);

/* GLENet's score rescaling fused with that loss: rcnn_cls = log((p + 1e-6) / (1 - p + 1e-6)),
 * p = sigmoid(ori_cls) * sigmoid(std_logit) (R each), then glx_rcnn_cls_loss on rcnn_cls and the chain rule back to
 * the two logits: rcnn_cls (R) out, out2 = { loss * weight, #valid }, grad_ori / grad_std (R) or NULL.
 * rcnn_cls_labels NULL: the rescaling only (inference), out2 and the gradients untouched.
 * Replaces: voxelrcnn_kl_label_iou_head.py:70-76 + roi_head_template.py:246-272 (23 elementwise launches). */
int glx_cls_rescale_loss(const float* ori_cls, const float* std_logit, const float* rcnn_cls_labels,
                         int R, float weight, float* rcnn_cls, float* out2, float* grad_ori,
                         float* grad_std, void* stream);

/* The three RoI-head loss terms of a training step in ONE launch: GLENet's score rescaling + classification term
 * (glx_cls_rescale_loss), the KL regression term (glx_kl_reg_loss) and the corner term (glx_corner_loss) -- the same
 * arithmetic, one block.  Ground-truth rows through a row stride (the (R, 8) target rows as they are), the int64
 * reg_valid_mask as it is, d loss / d rcnn_reg = KL + corner gradient.
 * out (device float[10]) = { cls + KL + corner, cls, #valid, KL, src, square, log, #foreground, corner, #foreground };
 * gradients (d total / d input) or NULL.
 * Replaces: RoIHeadTemplate.get_loss (pcdet/models/roi_heads/roi_head_template.py:246-272 + 274-291) over
 * voxelrcnn_kl_label_iou_head.py:70-76, 93-172. */
typedef struct glx_roi_head_losses_args {
  const float* ori_cls; const float* std_logit;      /* (R) logits of the two towers */
  const float* cls_labels;                           /* (R) soft IoU labels, < 0 = ignored */
  const float* rcnn_reg; const float* rcnn_reg_std;  /* (R, 7) */
  const float* rois;                                 /* (R, 7) LiDAR frame */
  const float* gt_ct; int gt_ct_ld;                  /* matched ground truth in the RoI frame, row stride >= 7 floats */
  const float* gt_src; int gt_src_ld;                /* ... in the LiDAR frame */
  const float* label_var;                            /* (R, 7) label variances */
  const int64_t* reg_valid;                          /* (R) > 0 = foreground */
  int R;
  float code_weights[7];
  float beta, w_cls, w_reg, w_corner;
  float* rcnn_cls;                                   /* (R) out: the rescaled logit */
  float* out;                                        /* float[10] */
  float* grad_ori; float* grad_std_logit;            /* (R) or NULL */
  float* grad_reg; float* grad_reg_std;              /* (R, 7) or NULL */
} glx_roi_head_losses_args;
int glx_roi_head_losses(const glx_roi_head_losses_args* args, void* stream);

/* Predicted boxes of the anchor head: box_preds (B, A, 7) residuals, dir_preds (B, A, num_dir_bins) or NULL,
 * anchors (A, 7) -> boxes (B, A, 7): ResidualCoder.decode_torch + the heading moved into the predicted
 * direction bin, with the rounding of the tensor expression.
 * Replaces: AnchorHeadTemplate.generate_predicted_boxes (pcdet/models/dense_heads/anchor_head_template.py:222-271),
 * box_coder_utils.py:44-69, common_utils.limit_period (pcdet/utils/common_utils.py:35-38). */
int glx_predicted_boxes(const float* box_preds, const float* dir_preds, const float* anchors, int B, int A,
                        int num_dir_bins, float dir_offset, float dir_limit_offset, float* boxes,
                        void* stream);

/* ------------------------------------------------------------------------------------
 * Training-mode BatchNorm1d (+ ReLU) over sparse-tensor features x (N, C), C a multiple of 4 that
 * divides 1024 (16 ... 512) -- or any C <= 512 when N <= 4096 (one block per channel).  Semantics of nn.BatchNorm1d(eps, momentum) followed by nn.ReLU as
 * the backbone applies them after every sparse conv (spconv_backbone.py:21-25,73): batch
 * statistics (biased variance) normalise, the running estimates take the unbiased variance.
 * relu = 0 gives plain BatchNorm.  fp64 column sums.  n_live (device int32, or NULL): statistics and transform
 * cover the first min(N, *n_live) rows of a capacity-sized matrix; the rows past them are written as ZEROS in y / dx.
 * state: NULL, or a device buffer of glx_bn_state_bytes() that the caller zero-fills ONCE and then only hands to
 *   these two functions, one buffer per stream.  NULL: statistics, a one-block finalize and the transform are three
 *   launches and the reduction order is fixed (bitwise reproducible).  With a state buffer the statistics kernel
 *   accumulates with fp64 atomics and its last block finalizes (two launches; the buffer is left zeroed again);
 *   the order of the fp64 additions then varies between runs (differences of one fp64 ulp before the rounding to
 *   float).  Matrices of at most 4096 rows always take a single launch with a fixed order.
 * y_stride / dy_stride: floats between consecutive rows of y (forward) / dy (backward); 0 = C.  A wider stride
 *   writes the result into -- reads the gradient from -- a column block of a wider row-major matrix: the
 *   concatenation of several normalised maps along the channels (BaseBEVBackbone's `torch.cat(ups, dim=1)`,
 *   base_bev_backbone.py:100-104, on channels-last maps) then costs no copy in either direction.
 * ------------------------------------------------------------------------------------ */
/* Channel-major form (round 5): training-mode BatchNorm of a tensor in the reference's STACKED convention -- batch dimension 1,
 * x = (C, L) row-major, L = rows x nsample -- i.e. the (1, C, M) / (1, C, M, nsample) inputs of the BatchNorm1d / BatchNorm2d
 * layers in pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:70-130 (torch.nn.BatchNorm semantics: biased variance to
 * normalise, running_var updated with the unbiased one and `momentum`; gamma / beta may be NULL = 1 / 0).  Nothing is
 * prepared per L: the vendor library builds a kernel per problem size, 0.45 s per new voxel count.  workspace =
 * glx_bn_cm_workspace_bytes(C).  Backward: dx, and dgamma / dbeta where not NULL. */
size_t glx_bn_cm_workspace_bytes(int C);
int glx_bn_cm_train_forward(const float* x, int C, long long L, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, float* y, float* save_mean, float* save_invstd,
                            void* workspace, size_t workspace_bytes, void* stream);
int glx_bn_cm_backward(const float* x, const float* dy, int C, long long L, const float* gamma, const float* save_mean,
                       const float* save_invstd, float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                       void* stream);
size_t glx_bn_workspace_bytes(int C);
size_t glx_bn_state_bytes(void);
int glx_bn_relu_train_forward(const float* x, int N, int C, const float* gamma, const float* beta,
                              float eps, float momentum, int relu, float* running_mean,
                              float* running_var, float* y, float* save_mean, float* save_invstd,
                              const int32_t* n_live, void* workspace, size_t workspace_bytes,
                              void* state, int y_stride, void* stream);
/* The transform of glx_bn_relu_train_forward alone, for statistics taken in a sparse conv's epilogue
 * (glx_sconv_opts.bn): y = relu?(x * coef[c] + coef[C + c]) on the live rows, zeros on the rows past them. */
int glx_bn_apply_forward(const float* x, const float* coef, int relu, int N, int C, const int32_t* n_live,
                         float* y, int y_stride, void* stream);
/* y = relu?(x * scale + shift + res): glx_bn_apply_forward with the identity branch of a residual block added (SparseBasicBlock,
 * pcdet/models/backbones_3d/spconv_backbone.py:30-64: relu(bn2(conv2(.)) + identity)); res (N, C) dense.  One launch for three. */
int glx_bn_apply_add_forward(const float* x, const float* coef, const float* res, int relu, int N, int C, const int32_t* n_live,
                             float* y, void* stream);
/* dx (N,C), dgamma (C), dbeta (C) from dy and the forward's x, mean, invstd.  The ReLU mask: from the forward's
 * output y, or -- y = NULL -- re-derived from x with gamma / beta (the forward's own rounding, so it is the same
 * mask; a third less traffic).  beta is only read in that case. */
int glx_bn_relu_backward(const float* x, const float* dy, const float* y, int N, int C,
                         const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                         int relu, float* dx, float* dgamma, float* dbeta, const int32_t* n_live,
                         void* workspace, size_t workspace_bytes, void* state, int dy_stride, void* stream);
/* The transform half of glx_bn_relu_backward for sums that were taken elsewhere (glx_bn_bwd_stats):
 * dx = a * (dz - b - xhat * cc) with coef = (a, b, cc) (3 C floats), xhat = (x - mean) * invstd; dz already carries the
 * ReLU mask.  x, dz, dx (N, C) row-major; rows >= *n_live (device, may be NULL) get zeros. */
int glx_bn_backward_apply(const float* x, const float* dz, const float* coef, const float* mean, const float* invstd,
                          int N, int C, const int32_t* n_live, float* dx, void* stream);
/* The statistics half of glx_bn_relu_backward alone (one launch): dgamma, dbeta and coef3 = (a, b, cc) for a consumer that applies
 * dx = a * (dy [y > 0] - b - xhat * cc) itself on load (glx_rows_linear_bn_backward).  x = the BatchNorm's input, dy = the gradient of
 * its (ReLU'd) output, both (N, C) row-major; coef3: 3 C floats; state: glx_bn_state_bytes(), zero-filled once (required). */
int glx_bn_backward_sums(const float* x, const float* dy, int N, int C, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, int relu, float* dgamma, float* dbeta,
                         const int32_t* n_live, float* coef3, void* state, void* stream);

/* ------------------------------------------------------------------------------------
 * Conv(kernel 1, bias = False) + training-mode BatchNorm (+ ReLU) on (rows, C) matrices: the input / output MLPs of the RoI-grid
 * pool (pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:70-130, mlps_in / mlps_out: nn.Conv1d / nn.Conv2d with 1 x 1
 * kernels + BatchNorm in the reference's (1, C, M) layout) -- csrc/glx_rows.hip.  Cin, Cout in {16, 32, 64}
 * (glx_rows_linear_supported), fp32 MFMA (exact products, fp32 accumulation).
 *
 * forward: z (rows, Cout) = x (rows, Cin) @ w^T, w (Cout, Cin) row-major, on the first min(rows, *n_live) rows (n_live: device
 * int32, may be NULL; the rest of z is left untouched and is not read by the backward).  bn_state != NULL: the BatchNorm statistics
 * of z are taken in the same launch: coef (scale[Cout], shift[Cout]: what glx_bn_apply_forward takes), save_mean, save_invstd, the
 * running statistics (NULL: not tracked); gamma / beta NULL = 1 / 0.  bn_state NULL: the plain product.
 *
 * backward: dy (rows, Cout) = the gradient of the transformed output when coef3 != NULL (from glx_bn_backward_sums on (z, dy)): the
 * BatchNorm (+ ReLU: mask re-derived from z and coef_fwd) backward is applied to dy on load; coef3 == NULL: dy is the gradient of z.
 * gx (rows, Cin) = dz @ w, rows past the live count zeroed (NULL: not wanted); gw (Cout, Cin) = dz^T @ x (NULL: not wanted), per-block
 * partial sums in `workspace` (glx_rows_linear_workspace_bytes) added in a fixed order by a second launch. */
int glx_rows_linear_supported(int Cin, int Cout);
size_t glx_rows_linear_workspace_bytes(int Cin, int Cout);
int glx_rows_linear_bn_forward(const float* x, int rows, int Cin, const float* w, int Cout, const int32_t* n_live, float* z,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* coef, float* save_mean, float* save_invstd, void* bn_state, void* stream);
int glx_rows_linear_bn_backward(const float* x, const float* z, const float* dy, int rows, int Cin, const float* w, int Cout,
                                const int32_t* n_live, const float* coef_fwd, int relu, const float* coef3, const float* mean,
                                const float* invstd, float* gx, float* gw, void* workspace, size_t workspace_bytes, void* stream);
/* out_j (cout, cin) = gy_j (rows, cout)^T @ x_j (rows, cin) for njobs <= 8 problems of one shape in ONE launch (host arrays of
 * device pointers; cout % 64 == 0, cin % 64 == 0, dense row-major, rows >= 1; fixed summation order): the weight gradients of the
 * RoI head's 256 x 256 nn.Linear layers (pcdet/models/roi_heads/voxelrcnn_head.py:40-66), which autograd leaves as one library
 * GEMM + reduction each. */
int glx_linear_wgrad_multi(int njobs, const float* const* x, const float* const* gy, float* const* out, int rows, int cin, int cout,
                           void* stream);
/* The wide first Linear of the RoI head (pcdet/models/roi_heads/voxelrcnn_head.py:40-52: nn.Linear(GRID_SIZE^3 * C, 256, bias=False)
 * on the pooled (rows, 20 736) features), exact fp32 products (v_mfma_f32_16x16x4_f32), fixed summation order:
 *   glx_linear_wide_forward      y (rows, N) = x (rows, K) @ w (N, K)^T      N % 64 == 0, K % 16 == 0; split along K into slabs whose
 *                                partial tiles a second launch sums (workspace: glx_linear_wide_workspace_bytes)
 *   glx_linear_wide_input_grad   gx (rows, K) = dz (rows, N) @ w (N, K)      K % 64 == 0, N % 32 == 0
 * The weight gradient dz^T x is glx_linear_wgrad_multi with one job.  Replaces autograd's three library GEMMs for that layer. */
size_t glx_linear_wide_workspace_bytes(int rows, int N, int K);
int glx_linear_wide_forward(const float* x, const float* w, float* y, int rows, int N, int K, void* workspace, size_t workspace_bytes,
                            void* stream);
int glx_linear_wide_input_grad(const float* dz, const float* w, float* gx, int rows, int N, int K, void* stream);

/* ------------------------------------------------------------------------------------
 * VectorPool family of PV-RCNN++.  Output slots are laid out in ascending new-point order
 * (the reference's atomicAdd hands them out in a run-dependent order); everything else follows the sequential
 * scan over a frame's support points of the CUDA kernels.  workspace: glx_vector_pool_workspace_bytes(M).
 *   glx_query_stacked_local_neighbor_idxs: per new point the first min(nsample, 1000) support points inside the
 *     ball (neighbor_type 1) or cube of half-width max_dist, GLOBAL indices; start_len (M,2) [offset, length],
 *     cumsum (device int) = total; segments beyond avg_length * M are dropped / truncated (the caller retries).
 *     Replaces query_stacked_local_neighbor_idxs_wrapper_stack (vector_pool.cpp:34-72, vector_pool_gpu.cu:122-233).
 *   glx_query_three_nn_by_stacked_local_idxs: three nearest list members per grid centre -> idx / squared distance
 *     (M, G, 3); idx -1 and 1e40 (inf in float) for an empty list.  Replaces
 *     query_three_nn_by_stacked_local_idxs_wrapper_stack (vector_pool.cpp:75-112, vector_pool_gpu.cu:19-119).
 *   glx_vector_pool: sums of the support features (channel i folded onto i % (C_out / G)) and of the offsets per
 *     sub-voxel of the local cube, point counts, and one row [support idx, new idx, sub-voxel] per pooled point;
 *     pooling_type 0 = every point (avg after the caller's division), 1 = the first point per sub-voxel.  Outputs
 *     arrive zero-filled.  cum_sum (device int) = rows needed; when it exceeds num_max_sum_points NOTHING is
 *     written (the caller's retry loop re-allocates; this call synchronises the stream once to know).
 *     Replaces vector_pool_wrapper (vector_pool.cpp:115-170, vector_pool_gpu.cu:243-430).
 *   glx_vector_pool_grad: grad_support[s, c] += grad_new[p, g * cg + c % cg] / max(count[p, g], 1) per row
 *     (float atomics); grad_support arrives zero-filled.  Replaces vector_pool_grad_wrapper
 *     (vector_pool.cpp:173-200, vector_pool_gpu.cu:433-480).
 * ------------------------------------------------------------------------------------ */
size_t glx_vector_pool_workspace_bytes(int M);
int glx_query_stacked_local_neighbor_idxs(const float* support_xyz, const int32_t* xyz_batch_cnt,
                                          const float* new_xyz, const int32_t* new_xyz_batch_cnt, int B, int M,
                                          int32_t* stack_neighbor_idxs, int32_t* start_len, int32_t* cumsum,
                                          int avg_length, float max_dist, int nsample, int neighbor_type,
                                          void* workspace, size_t workspace_bytes, void* stream);
int glx_query_three_nn_by_stacked_local_idxs(const float* support_xyz, const float* new_xyz_grid_centers,
                                             int32_t* new_xyz_grid_idxs, float* new_xyz_grid_dist2,
                                             const int32_t* stack_neighbor_idxs, const int32_t* start_len, int M,
                                             int num_total_grids, void* stream);
int glx_vector_pool(const float* support_xyz, const float* support_features, const int32_t* xyz_batch_cnt,
                    const float* new_xyz, const int32_t* new_xyz_batch_cnt, int B, int M, int num_c_in,
                    int num_c_out, int num_grid_x, int num_grid_y, int num_grid_z, float max_dist, int use_xyz,
                    int num_max_sum_points, int nsample, int neighbor_type, int pooling_type, float* new_features,
                    float* new_local_xyz, int32_t* point_cnt_of_grid, int32_t* grouped_idxs, int32_t* cum_sum,
                    void* workspace, size_t workspace_bytes, void* stream);
int glx_vector_pool_grad(const float* grad_new_features, const int32_t* point_cnt_of_grid,
                         const int32_t* grouped_idxs, int num_idxs, int num_c_in, int num_c_out,
                         int num_total_grids, float* grad_support_features, void* stream);

/* ------------------------------------------------------------------------------------
 * RoI-grid pooling, training path: position MLP (Conv2d(3, C, 1, bias=False) + BatchNorm2d(C), training-mode
 * statistics or running ones) + add + ReLU + max over the neighbours, fused -- no (M, nsample, C) tensor exists.
 *   feats (N, C) row-major (the mlps_in output), xyz (N, 3) voxel centres, new_xyz (M, 3) grid points,
 *   idx (M, nsample) GLOBAL rows as glx_voxel_query_index / glx_roi_grid_query leave them (idx[m,0] < 0: empty
 *   ball = nsample rows of feature 0 and offset 0), w_pos (C, 3), gamma / beta (C) or NULL, C in {16, 32, 64}.
 *   pooled (M, C) = max_s relu(feats[idx[m,s]] + BN(w_pos . (xyz[idx[m,s]] - new_xyz[m]))), arg (M, C) uint8 = the
 *   winning slot (first on ties).  save (glx_pos_pool_save_floats(C) floats) and moments (9 doubles) carry the
 *   statistics to the backward call.  training != 0: batch statistics over all M * nsample rows (computed from
 *   the moments of the offsets, fp64), running_mean / running_var updated with `momentum` (unbiased variance);
 *   training == 0: the running estimates normalise.
 *   backward: dfeats (N, C) (zero-filled by the call, atomic adds at the winning slots), dW (C, 3), dgamma,
 *   dbeta (C) -- the exact gradients of the unfused formulation.
 * Replaces: NeighborVoxelSAModuleMSG.forward lines 91-104 (pcdet/ops/pointnet2/pointnet2_stack/
 * voxel_pool_modules.py) in training: grouping_operation x 2, mlps_pos, add, ReLU, max_pool2d and their autograd.
 * ------------------------------------------------------------------------------------ */
size_t glx_pos_pool_workspace_bytes(int C);
int glx_pos_pool_save_floats(int C);
int glx_pos_pool_forward(const float* feats, int N, int C, const float* xyz, const float* new_xyz,
                         const int32_t* idx, int M, int nsample, const float* w_pos, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         int training, float* pooled, uint8_t* arg, float* save, double* moments,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Form of glx_pos_pool_backward's feature gradient: 1 (default) = summed per block of 256 grid points in an LDS hash table, one global
 * atomic per (occupied slot, channel); 0 = one global atomic per (point, channel).  env GLX_RP_BWD_AGG at load; returns the previous. */
int glx_pos_pool_set_backward_form(int aggregate);
/* The same with the layer's output MLP (voxel_pool_modules.py:105-108, mlps_out = Conv1d(C, C, 1, bias=False) + BatchNorm1d +
 * ReLU): y_out (M, C) = pooled @ w_out^T (w_out (C, C) row-major) formed in the pooling launch, with the training-mode
 * BatchNorm statistics of y_out taken on the way (bn_out: glx_bn_stats as in glx_conv_opts.bn; the transform is
 * glx_bn_apply_forward).  C <= 32, M > 0.  w_out = y_out = bn_out = NULL: glx_pos_pool_forward. */
int glx_pos_pool_forward_out(const float* feats, int N, int C, const float* xyz, const float* new_xyz, const int32_t* idx,
                             int M, int nsample, const float* w_pos, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, int training,
                             float* pooled, uint8_t* arg, float* save, double* moments, const float* w_out, float* y_out,
                             const glx_bn_stats* bn_out, void* workspace, size_t workspace_bytes, void* stream);
int glx_pos_pool_backward(const float* dpooled, const float* pooled, const uint8_t* arg, const int32_t* idx,
                          const float* xyz, const float* new_xyz, int M, int nsample, int C, int N,
                          const float* w_pos, const float* gamma, const float* save, const double* moments,
                          int training, float* dfeats, float* dW, float* dgamma, float* dbeta, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Update of the training step on flat fp32 buffers: gradient-norm clipping + AdamW in two launches.
 * params / grads / exp_avg / exp_avg_sq: n floats each, 16-byte aligned; hyper: DEVICE float[2] =
 * { learning rate, beta1 } (written by the one-cycle schedule, read by a recorded HIP graph); step: DEVICE
 * int32 counter, incremented by the call; max_norm <= 0 disables clipping; norm_out (device float, may be
 * NULL) receives the total gradient norm before clipping.  Arithmetic of torch.optim.AdamW (decoupled
 * weight decay, bias corrections from `step`) after torch.nn.utils.clip_grad_norm_.
 * Replaces: tools/train_utils/train_utils.py:38-39 (clip_grad_norm_ + optimizer.step()) with the
 * adam_onecycle optimiser of tools/train_utils/optimization/__init__.py:29-53.
 * ------------------------------------------------------------------------------------ */
size_t glx_adamw_workspace_bytes(void);
int glx_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                        const float* hyper, float beta2, float eps, float weight_decay, float max_norm,
                        int32_t* step, float* norm_out, void* workspace, size_t workspace_bytes,
                        void* stream);
/* Same with the gradient taken as grads * grad_scale (each element rounded to float, then norm / clip / update):
 * grad_scale = 1 / world_size turns the SUM all-reduce of the data-parallel step into DistributedDataParallel's
 * average (tools/train.py:144-145) without a separate pass over the 30 MB buffer. */
int glx_adamw_clip_step_scaled(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                               const float* hyper, float beta2, float eps, float weight_decay, float max_norm,
                               float grad_scale, int32_t* step, float* norm_out, void* workspace,
                               size_t workspace_bytes, void* stream);

/* The CVAE decoder's small point extractor in TRAINING mode without an intermediate tensor (cvae_uncertainty/point_net.py:31-49
 * SimPointNetfeat(x = 0.5): Conv1d(C, 8, 1) + BatchNorm1d + ReLU, Conv1d(8, 8, 1) + BatchNorm1d + ReLU, Conv1d(8, 8, 1) + BatchNorm1d, max over
 * the points; autograd's backward of it).  points (B, C <= 8, P); w1 (8, C), w2, w3 (8, 8) row-major; b*: the convolutions' biases or NULL
 * (they only move the batch means: running means, zero gradient); gamma / beta: the BatchNorms' affine parameters; rmean / rvar: running
 * statistics updated with `momentum` (unbiased variance), or NULL.
 *   forward: out (B, 8) = max_p bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) with BATCH statistics over all B P rows, arg (B, 8) = the
 *     lowest point it occurs at, xh_ext (B, 8) = the normalised value there, coef (3, 4, 8) = per layer gamma invstd | beta - that * mean |
 *     mean | invstd of W h (no bias) -- what backward needs again.  Ten launches (three of them one block), every pass reads the points only.
 *   backward: gout (B, 8) -> grads (4, 64) = dW1 (8 x 8, the first C columns), dW2, dW3, then dgamma1, dbeta1, dgamma2, dbeta2, dgamma3, dbeta3
 *     (8 each).  Sums: fp32 over a block's rows, double over the blocks, fixed order.
 * workspace: glx_narrowfeat_workspace_bytes(). */
size_t glx_narrowfeat_workspace_bytes(void);
int glx_narrowfeat_train_forward(const float* points, int B, int C, int P, const float* w1, const float* b1, const float* gamma1,
                                 const float* beta1, float* rmean1, float* rvar1, const float* w2, const float* b2, const float* gamma2,
                                 const float* beta2, float* rmean2, float* rvar2, const float* w3, const float* b3, const float* gamma3,
                                 const float* beta3, float* rmean3, float* rvar3, float eps, float momentum, float* out, int32_t* arg,
                                 float* xh_ext, float* coef, void* workspace, size_t workspace_bytes, void* stream);
int glx_narrowfeat_train_backward(const float* points, int B, int C, int P, const float* w1, const float* w2, const float* w3,
                                  const float* coef, const float* gout, const int32_t* arg, const float* xh_ext, float* grads,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* The wide extractors' FIRST layer in training mode, from the points (cvae_uncertainty/point_net.py:10-16 conv1 + bn1 + relu of PointNetfeat;
 * autograd's backward of it): batch statistics of W x from the mean and covariance of x, one pass that writes
 * h1 (B P, 64) = relu(bn(conv(x))) as rows (row b P + p = point p of object b), and backward one pass over the gradient of h1.
 * points (B, C <= 8, P); w (64, C); bias or NULL (moves the batch mean only: running mean, zero gradient); coef (4, 64) out:
 * gamma invstd | beta - that * mean | mean | invstd of W x; moments (44 doubles) out: sums of x (8) and of its products (upper triangle,
 * 8 wide) -- backward takes coef and moments again.  grads: dW (64, C) | dgamma (64) | dbeta (64). */
size_t glx_point_layer1_workspace_bytes(void);
int glx_point_layer1_train_forward(const float* points, int B, int C, int P, const float* w, const float* bias, const float* gamma,
                                   const float* beta, float* rmean, float* rvar, float eps, float momentum, float* h1, float* coef,
                                   double* moments, void* workspace, size_t workspace_bytes, void* stream);
int glx_point_layer1_train_backward(const float* points, int B, int C, int P, const float* w, const float* coef, const double* moments,
                                    const float* dh1, float* grads, void* workspace, size_t workspace_bytes, void* stream);

/* The CVAE's weight regulariser on a flat parameter buffer (cvae_uncertainty/model.py:20-28 l2_regularisation: the SUM over the
 * parameter tensors of their 2-norms; train_utils.py:57-59 adds 1e-4 of it to the loss): segs = nseg pairs (start, length) of the
 * tensors inside params (ascending, disjoint); norms (nseg) <- the tensors' 2-norms, total (1) <- scale * their sum; workspace:
 * glx_flat_l2_workspace_bytes(nseg) (eight blocks per tensor leave their partial sums there). */
size_t glx_flat_l2_workspace_bytes(int nseg);
int glx_flat_l2_norms(const float* params, const int64_t* segs, int nseg, float scale, float* norms, float* total, void* workspace,
                      size_t workspace_bytes, void* stream);
/* ... and its gradient added into the flat gradient buffer: grads[i] += coef[0] * scale * params[i] / norms[tensor of i] (nothing for
 * a zero tensor, as torch's norm backward; coef == NULL: 1) -- the gradient of scale * total, for the loss that added it with the
 * upstream gradient coef.  n = the length of both buffers. */
int glx_flat_l2_norm_grad_add(const float* params, const int64_t* segs, int nseg, const float* norms, const float* coef, float scale,
                              float* grads, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * PV-RCNN set-abstraction operators (same extension module in the reference).
 * ------------------------------------------------------------------------------------ */

/* Stacked farthest point sampling: frame b contributes num_sampled[b] indices (global, into the
 * stacked xyz), starting with its first point.  temp (N) arrives filled with 1e10; max_points = an
 * upper bound of the largest frame (<= 16384 keeps the distances in registers; 0 = unknown).
 * Replaces: pointnet2_stack_cuda.stack_farthest_point_sampling_wrapper (sampling.cpp:40-60) and,
 * with B equal-sized frames, farthest_point_sampling_wrapper (sampling.cpp:24-37). */
int glx_stack_fps(const float* xyz, const int32_t* xyz_batch_cnt, int B, int max_points,
                  const int32_t* num_sampled, float* temp, int32_t* idxs, void* stream);
/* Squared distances + global indices of the 3 nearest known points of the same frame.
 * max_queries_per_frame bounds unknown_batch_cnt[] (0 = N).
 * Replaces: pointnet2_stack_cuda.three_nn_wrapper (interpolate.cpp:35-63). */
int glx_three_nn(int B, int N, int max_queries_per_frame, const float* unknown,
                 const int32_t* unknown_batch_cnt, const float* known,
                 const int32_t* known_batch_cnt, float* dist2, int32_t* idx, void* stream);
/* out[p,:] = sum_j weight[p,j] * features[idx[p,j],:]; grad_features (M,C) arrives zero-filled.
 * Replaces: three_interpolate_wrapper / three_interpolate_grad_wrapper (interpolate.cpp:66-110). */
int glx_three_interpolate(int N, int C, const float* features, const int32_t* idx,
                          const float* weight, float* out, void* stream);
int glx_three_interpolate_grad(int N, int C, const float* grad_out, const int32_t* idx,
                               const float* weight, float* grad_features, void* stream);

/* Test hook for csrc/glx_libm.h (the device restatement of glibc's sinf / cosf / atanf / atan2f that the rotated
 * overlap and the inside tests use in place of iou3d_cpu.cpp's libm calls, :30,:83,:141-142): out[i] = fn(x[i])
 * with fn 0 = sinf, 1 = cosf, 2 = atanf, or 3 = atan2f(x[i], y[i]). */
int glx_libm_eval(int fn, const float* x, const float* y, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * pcdet.ops.pointnet2.pointnet2_batch.pointnet2_batch_cuda (pointnet2_batch/src/pointnet2_api.cpp:10-24):
 * the batch-layout PointNet++ operators.  B equal frames; xyz (B, N, 3); features CHANNEL-major (B, C, N);
 * every index is local to its frame.  `pcdet.models` imports this module unconditionally
 * (backbones_3d/pointnet2_backbone.py:4, roi_heads/pointrcnn_head.py:4).
 * ------------------------------------------------------------------------------------ */

/* idx (B, m, nsample): first nsample points with d2 < radius^2 in index order, unused slots = first hit; a
 * ball without hits leaves its row untouched (the caller zero-fills it, pointnet2_utils.py:236).
 * Replaces: ball_query_wrapper_fast (ball_query.cpp:28-38, ball_query_gpu.cu:15-51). */
int glx_batch_ball_query(int B, int n, int m, float radius, int nsample, const float* new_xyz,
                         const float* xyz, int32_t* idx, void* stream);
/* out (B, C, npoints, nsample) = points[b, c, idx[b, p, s]].
 * Replaces: group_points_wrapper_fast (group_points.cpp:33-44, group_points_gpu.cu:57-78). */
int glx_batch_group_points(int B, int C, int n, int npoints, int nsample, const float* points,
                           const int32_t* idx, float* out, void* stream);
/* grad_points (B, C, n) += scatter of grad_out (B, C, npoints, nsample); arrives zero-filled.
 * Replaces: group_points_grad_wrapper_fast (group_points.cpp:20-30, group_points_gpu.cu:14-32). */
int glx_batch_group_points_grad(int B, int C, int n, int npoints, int nsample, const float* grad_out,
                                const int32_t* idx, float* grad_points, void* stream);
/* out (B, C, npoints) = points[b, c, idx[b, p]] and its adjoint.
 * Replaces: gather_points_wrapper_fast / gather_points_grad_wrapper_fast (sampling.cpp:17-37,
 * sampling_gpu.cu:14-71). */
int glx_batch_gather_points(int B, int C, int n, int npoints, const float* points, const int32_t* idx,
                            float* out, void* stream);
int glx_batch_gather_points_grad(int B, int C, int n, int npoints, const float* grad_out,
                                 const int32_t* idx, float* grad_points, void* stream);
/* idx (B, m) LOCAL indices; temp (B, N) arrives filled with 1e10; the first sample of a frame is its point 0.
 * Ties follow the block size the reference launches (opt_n_threads(N), cuda_utils.h:9-13).
 * Replaces: farthest_point_sampling_wrapper (sampling.cpp:40-49, sampling_gpu.cu:97-257) of both the batch
 * module and pointnet2_stack_cuda (pointnet2_stack/src/sampling.cpp:24-37). */
int glx_batch_fps(int B, int N, int m, const float* xyz, float* temp, int32_t* idx, void* stream);
/* dist2 (B, n, 3) squared distances, idx (B, n, 3) LOCAL indices of the 3 nearest of known (B, m, 3).
 * Replaces: three_nn_wrapper_fast (interpolate.cpp:21-31, interpolate_gpu.cu:15-60). */
int glx_batch_three_nn(int B, int n, int m, const float* unknown, const float* known, float* dist2,
                       int32_t* idx, void* stream);
/* out (B, C, n) = sum_j weight[b,p,j] * points[b, c, idx[b,p,j]], points (B, C, m); grad_points (B, C, m)
 * arrives zero-filled.  Replaces: three_interpolate_wrapper_fast / three_interpolate_grad_wrapper_fast
 * (interpolate.cpp:34-61, interpolate_gpu.cu:84-153). */
int glx_batch_three_interpolate(int B, int C, int m, int n, const float* points, const int32_t* idx,
                                const float* weight, float* out, void* stream);
int glx_batch_three_interpolate_grad(int B, int C, int n, int m, const float* grad_out, const int32_t* idx,
                                     const float* weight, float* grad_points, void* stream);

/* ------------------------------------------------------------------------------------
 * CVAE label-uncertainty generator (BASELINE config 4): the PointNet feature extractor as ONE
 * kernel.  out[b,:] = max_p W3 relu(W2 relu(W1 x[b,:,p] + b1) + b2) + b3 with widths 64/128/512;
 * points (B, Cin<=8, P); (W, b) = Conv1d(k=1) + eval-mode BatchNorm1d folded; W2p / W3p in MFMA
 * fragment order [tile_out][tile_in][q][i][e] (see csrc/glx_pointnet.hip).
 * Replaces: PointNetfeat.forward (cvae_uncertainty/point_net.py:10-28) in eval mode.
 * ------------------------------------------------------------------------------------ */
size_t glx_pointnet_feat_lds_bytes(void);
int glx_pointnet_feat(const float* points, int B, int Cin, int P, const float* W1, const float* b1,
                      const float* W2p, const float* b2, const float* W3p, const float* b3,
                      float* out, void* stream);
/* Same function for narrow extractors (all widths <= 16, e.g. SimPointNetfeat 8/8/8,
 * point_net.py:31-49); W1 (C1,Cin), W2 (C2,C1), W3 (C3,C2) row-major, folded like above. */
/* The same extractor with layers 2 and 3 as f16x2 products (two scaled fp16 pieces per operand, three v_mfma_f32_16x16x32_f16 per
 * product tile, >= 20.4 bits per product, fp32 sums -- the arithmetic of glx_conv3x3_forward's default).  W2h / W3h: the folded
 * (128, 64) / (512, 128) weights as two fp16 planes of w 2^ew[row] in the kernel's operand order, [output tile][k-step s][plane]
 * [lane 16 q + m][slot 4 h + e] = W[16 tile + m][32 s + 16 h + 4 q + e]; ew2 (128) / ew3 (512): the rows' exponents (max |w| 2^ew in
 * [2^14, 2^15), 0 for a zero row).  cvae_uncertainty/point_net.py:10-28, eval mode. */
size_t glx_pointnet_feat_f16x2_lds_bytes(void);
int glx_pointnet_feat_f16x2(const float* points, int B, int Cin, int P, const float* W1, const float* b1, const void* W2h,
                            const int32_t* ew2, const float* b2, const void* W3h, const int32_t* ew3, const float* b3, float* out,
                            void* stream);
/* ... with the narrow extractor (widths 8, 8, 8; cvae_uncertainty/point_net.py:31-49) of the same points in the same launch:
 * narrow = 216 floats (W1 (8 x 8, input channels zero-padded), b1, W2, b2, W3, b3; eval-mode BatchNorm folded), narrow_out (B, 8);
 * narrow == NULL: glx_pointnet_feat_f16x2. */
int glx_pointnet_feat_f16x2_pair(const float* points, int B, int Cin, int P, const float* W1, const float* b1, const void* W2h,
                                 const int32_t* ew2, const float* b2, const void* W3h, const int32_t* ew3, const float* b3,
                                 float* out, const float* narrow, float* narrow_out, void* stream);
/* Which kernel the f16 x 2 extractor entries launch: 1 (default) keeps W3 in registers and passes the points through LDS (eight waves per
 * object), 0 streams W3 through an LDS ring (four waves per object).  Same arithmetic, same results bit for bit; returns the previous
 * setting.  For measurements (tools/sampler_time.py). */
int glx_pointnet_feat_set_form(int w_stationary);
/* The two data terms of the CVAE's training loss with their gradients in one launch (cvae_uncertainty/model.py:296-345 reg_loss: code-
 * weighted smooth-L1 with the sin-difference heading + TWICE the direction cross-entropy, see dense_path.cvae_reg_loss; model.py:205-212:
 * KL(posterior || prior) of diagonal Gaussians with scale = exp(logvar) + 3e-22, mean over the batch).  out (3) = loss_loc, loss_dir,
 * latent, weights applied; d_* = the gradients of those weighted terms. */
int glx_cvae_losses(const float* pred, const float* labels, const float* code_weights, int B, int bins, float beta, float loc_weight,
                    float dir_weight, float dir_offset, const float* mu1, const float* logvar1, const float* mu2, const float* logvar2,
                    int L, float latent_weight, float* out, float* d_pred, float* d_mu1, float* d_logvar1, float* d_mu2,
                    float* d_logvar2, void* stream);
/* The eval branch of Generator.forward behind the extractors (cvae_uncertainty/model.py:245-265) in one launch: prior (mu, logvar)
 * from f512 (B, 512), z = eps exp(logvar / 2) + mu, the decoder on cat(f8 (B, 8), z), heading decoded from its bin.
 * w: WL (16 x 512: fc1 | fc2 rows) | bL (16) | W1T (16 x 64, [input][output], BatchNorm folded) | b1 (64) | W2T (64 x 64) | b2 (64) |
 * WhT (4 x 64 x 64, heads centre / size / heading / bins, [input][output]) | bh (4 x 64) | Wo ((7 + bins) x 64).  out (B, 7 + bins). */
int glx_cvae_sample_tail(const float* f512, const float* f8, const float* eps, const float* w, int B, int bins, float dir_offset,
                         float dir_limit_offset, float* out, void* stream);
int glx_pointnet_feat_small(const float* points, int B, int Cin, int P, int C1, int C2, int C3,
                            const float* W1, const float* b1, const float* W2, const float* b2,
                            const float* W3, const float* b3, float* out, void* stream);

/* The anchor head's three 1x1 convolutions (pcdet/models/dense_heads/anchor_head_single.py:19-37, 52-75: conv_cls, conv_box,
 * conv_dir_cls on spatial_features_2d, each followed by permute(0, 2, 3, 1).contiguous()) in one pass per direction over the
 * channels-last map x (M = B * H * W pixels, C channels, C a multiple of 64 up to 512).  W[k] (n[k], C) row-major, bias[k]
 * (n[k]) or NULL, out[k] / grad[k] (M, n[k]) row-major = the permuted predictions; n[0] + n[1] + n[2] <= 32, n[2] may be 0.
 * W / bias / n / out / grad / gW / gb are HOST arrays of three device pointers (ints). */
int glx_head1x1_forward(const float* x, int64_t M, int C, const float* const* W, const float* const* bias,
                        const int32_t* n, float* const* out, void* stream);
int glx_head1x1_input_grad(const float* const* grad, int64_t M, int C, const float* const* W, const int32_t* n, float* gx,
                           void* stream);
/* The same two for a map given as two column parts x0 (M, c0) | x1 (M, C - c0) (c0 a multiple of 16) that are transformed on
 * load, x' = relu(x * scale + shift) with coef_p = scale | shift of part p (2 * its width floats; both NULL: no transform) --
 * BaseBEVBackbone's deblock outputs with their training-mode BatchNorm + ReLU (base_bev_backbone.py:51-66, 100-104) read
 * without the concatenated map having been written. */
int glx_head1x1_forward_parts(const float* x0, const float* x1, int c0, const float* coef0, const float* coef1, int64_t M, int C,
                              const float* const* W, const float* const* bias, const int32_t* n, float* const* out, void* stream);
int glx_head1x1_weight_grad_parts(const float* const* grad, const float* x0, const float* x1, int c0, const float* coef0,
                                  const float* coef1, int64_t M, int C, const int32_t* n, float* const* gW, float* const* gb,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* The input gradient of glx_head1x1_forward_parts WITH the two BatchNorm + ReLU transforms' backward sums: dz_p (M, c_p) = the
 * gradient of part p's transformed map masked with its ReLU; bn_p (glx_bn_bwd_stats: y unused, coef_fwd = the forward scale |
 * shift, mean, invstd, gamma; outputs coef (3 * c_p) for glx_bn_backward_apply, dgamma, dbeta; both parts name the same
 * `state`).  C = 256.  Replaces glx_head1x1_input_grad + the statistics launch of each part's BatchNorm backward. */
int glx_head1x1_input_grad_bn(const float* const* grad, int64_t M, int C, const float* const* W, const int32_t* n, const float* y0,
                              const float* y1, int c0, const glx_bn_bwd_stats* bn0, const glx_bn_bwd_stats* bn1, float* dz0,
                              float* dz1, void* stream);
/* The same with the kernel form chosen by the caller: 0 = a wave owns 16 pixels x all channels, 1 = a wave owns 64 channels of
 * the block's 64 pixels (needs c0 % 64 == 0, else form 0 runs): 74 us against 109 alone on the 144 MB map, no difference inside
 * the recorded training step (DESIGN 9.22 viii). */
int glx_head1x1_input_grad_bn_form(const float* const* grad, int64_t M, int C, const float* const* W, const int32_t* n,
                                   const float* y0, const float* y1, int c0, const glx_bn_bwd_stats* bn0,
                                   const glx_bn_bwd_stats* bn1, float* dz0, float* dz1, int form, void* stream);
size_t glx_head1x1_wgrad_workspace_bytes(int C);
int glx_head1x1_weight_grad(const float* const* grad, const float* x, int64_t M, int C, const int32_t* n, float* const* gW,
                            float* const* gb, void* workspace, size_t workspace_bytes, void* stream);

/* Training twin of the extractor's last layer (cvae_uncertainty/point_net.py:22-28: conv3 + bn3, max over the points;
 * model.py:200-243 trains through it): the (B x P x 512) tensor is never built.  h2 (B * P, 128) row-major = the second
 * layer's output, W3p = the (512, 128) weight in the fragment order of glx_pointnet_feat.
 * glx_pointmax_forward: per (object, channel) max / min of y = W3 h2 over the object's points, the point indices they occur
 * at (ties: the lower index) and the sums of y and y^2 (BatchNorm's batch statistics); all outputs (B, 512).
 * glx_pointmax_scatter: dh2 (B * P, 128) = rows of W3 (512, 128) row-major scattered to arg[b, c] with weights coef[b, c]
 * on top of the row `init` (128 floats, may be NULL = zeros); every row is written, deterministic.
 * glx_pointmax_wsum: T (512, 128) = sum_b g[b, c] * h2[b * P + arg[b, c], :]. */
int glx_pointmax_forward(const float* h2, int B, int P, const float* W3p, float* vmax, float* vmin, int32_t* amax,
                         int32_t* amin, float* s1, float* s2, void* stream);
/* glx_pointmax_forward's one-sided twin with f16 x 2 products (three fp16 MFMAs per product tile, >= 20.4 bits, fp32 sums):
 * vext[b, c] = max_p (h2 W^T)[b, p, c] and the lowest point it occurs at; no sums (the caller takes them from the moments of h2).
 * W3h / ew3: the weight as glx_pointnet_feat_f16x2 takes it; a channel's MINIMUM = that row handed in negated.
 * pre_coef (here and in glx_rows128_moments / glx_rows128_affine_f16x2 / glx_pointmax_wsum_pre; NULL: none): scale | shift, 128 floats each --
 * the (rows, 128) operand is the RAW output z of the layer in front and its rows are read as relu(z scale + shift): that layer's
 * BatchNorm + ReLU applied on load, the transformed matrix is never written. */
int glx_pointmax_forward_f16x2(const float* h2, int B, int P, const void* W3h, const int32_t* ew3, float* vext, int32_t* aext,
                               const float* pre_coef, void* stream);
int glx_pointmax_scatter(const int32_t* arg, const float* coef, const float* W3, const float* init, int B, int P, float* dh2,
                         void* stream);
/* glx_pointmax_scatter_add: the same sums ADDED to the rows of dh2 some channel points at (the others are left alone).
 * glx_rows128_affine_f16x2: y (rows, 128) = init (128 or NULL) + x (rows, 128) W^T with f16 x 2 products; Wh / ew: the (128, 128)
 * weight as two fp16 planes of w 2^ew[row] in MFMA operand order (what glx_pointnet_feat_f16x2 takes for its layers).  Together the
 * input gradient of the 128 -> 512 layer + BatchNorm + max: dh2 = -v - h2 M, then the extreme points' rows of W3. */
int glx_pointmax_scatter_add(const int32_t* arg, const float* coef, const float* W3, int B, int P, float* dh2, void* stream);
/* The f16 x 2 weight image the point kernels take (glx_pointnet_feat_f16x2, glx_pointmax_forward_f16x2, glx_rows128_affine_f16x2,
 * glx_rows_linear_bn_*_64_128_f16x2): w (Cout, Cin) with element strides (stride_row, stride_col), times row_scale[row] (NULL: 1) times
 * scale (row_scale_sign_only: +1 / -1 by the sign of row_scale[row]) -> two fp16 planes of w 2^ew[row] in MFMA operand order (img:
 * Cout x Cin x 2 halfs) + ew (Cout).  Cout % 16 == 0,
 * Cin in {32, 64, 96, 128}. */
int glx_f16x2_pack(const float* w, int cout, int cin, long long stride_row, long long stride_col, const float* row_scale,
                   int row_scale_sign_only, float scale, void* img, int32_t* ew, void* stream);
/* glx_rows_linear_bn_forward for the CVAE's 64 -> 128 point layer with f16 x 2 products (memory-bound where the fp32-MFMA form is
 * matrix-bound): Wh / ew = the (128, 64) weight as two fp16 planes of w 2^ew[row] in MFMA operand order; the BatchNorm arguments
 * as there (bn_state == NULL: the product alone). */
int glx_rows_linear_bn_forward_64_128_f16x2(const float* x, int rows, const void* Wh, const int32_t* ew, const int32_t* n_live, float* z,
                                            const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                            float* running_var, float* coef, float* save_mean, float* save_invstd, void* bn_state,
                                            void* stream);
/* ... and its backward in one pass over the rows (arguments as glx_rows_linear_bn_backward's): dz formed on load, dX = dz W with
 * f16 x 2 products (Wth / ewt = the image of W^T (64, 128)), dW = dz^T x with bf16 x 3 products (the contraction runs over the rows).
 * workspace: glx_rows_bwd_64_128_workspace_bytes(). */
size_t glx_rows_bwd_64_128_workspace_bytes(void);
int glx_rows_linear_bn_backward_64_128_f16x2(const float* x, const float* z, const float* dy, int rows, const void* Wth,
                                             const int32_t* ewt, const int32_t* n_live, const float* coef_fwd, int relu,
                                             const float* coef3, const float* mean, const float* invstd, float* gx, float* gw,
                                             void* workspace, size_t workspace_bytes, void* stream);
/* G (128 x 128, fp64, row-major) = x^T x and h (128 floats) = the column sums of x (rows, 128) in one pass: the moments the
 * 128 -> 512 layer's batch statistics and weight gradient are made of (dense_path.PointMaxBN).  bf16 x 3 products, fp32 sums per
 * block of rows, fp64 over the blocks in a fixed order.  workspace: glx_rows128_moments_workspace_bytes(). */
size_t glx_rows128_moments_workspace_bytes(void);
int glx_rows128_moments(const float* x, long long rows, double* G, float* h, const float* pre_coef, void* workspace,
                        size_t workspace_bytes, void* stream);
int glx_rows128_affine_f16x2(const float* x, long long rows, const void* Wh, const int32_t* ew, const float* init, float* y,
                             const float* pre_coef, void* stream);
/* The training-mode BatchNorm around the max of the 128 -> 512 layer (cvae_uncertainty/point_net.py:22-28: bn3 + torch.max), as launches
 * instead of ~40 tensor statements per direction (dense_path.PointMaxBN):
 * glx_pointmax_bn_forward: batch statistics of y = h2 W3^T from the moments G = h2^T h2 (fp64), h = sum h2 (glx_rows128_moments) in
 *   fp64, running statistics (the conv bias folded into the running mean), ext = sign(gamma) vext in place, out = (ext - mean) scale +
 *   beta; mean / invstd / scale (512 each) are what the backward takes.
 * glx_pointmax_bn_backward_sums: dgamma, dbeta, the vectors bvec / cvec of dy = a g^ - bvec - cvec (y - mean), M = W3^T diag(cvec) W3
 *   and nv = -(bvec - cvec mean) W3 (the dense part of the input gradient is nv - h2 M: glx_rows128_affine_f16x2).
 * glx_pointmax_scatter_add_scaled: glx_pointmax_scatter_add with coef[b, c] chan_scale[c].
 * glx_pointmax_bn_backward_weight: dW3 = scale T - bvec (x) h - diag(cvec) (W3 G - mean (x) h), T from glx_pointmax_wsum on the
 *   unscaled gradient. */
int glx_pointmax_bn_forward(const float* W3, const double* G, const float* h, long long R, float* vext, int B, const float* gamma,
                            const float* beta, const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                            float* mean, float* invstd, float* scale, float* out, void* stream);
int glx_pointmax_bn_backward_sums(const float* g, const float* ext, int B, long long R, const float* W3, const float* mean,
                                  const float* invstd, const float* scale, float* dgamma, float* dbeta, float* bvec, float* cvec,
                                  float* M, float* nv, void* stream);
int glx_pointmax_bn_backward_weight(const float* W3, const double* G, const float* h, const float* T, const float* scale,
                                    const float* bvec, const float* cvec, const float* mean, float* dW, void* stream);
int glx_pointmax_scatter_add_scaled(const int32_t* arg, const float* coef, const float* chan_scale, const float* W3, int B, int P,
                                    float* dh2, void* stream);
/* The BatchNorm-backward sums of the layer in FRONT of the 128 -> 512 layer taken by the two launches that write its output gradient
 * (cvae_uncertainty/point_net.py:17,24 bn2 behind conv2; autograd would read dh2 and the layer's raw output z once more, 2 GB at
 * configs[3]).  dh2 is the gradient with respect to relu(z scale + shift) (pre_coef = scale | shift), dz = [z scale + shift > 0] dh2:
 *   glx_rows128_affine_f16x2_sums: glx_rows128_affine_f16x2 + bsum (glx_rows128_affine_blocks(rows), 2, 128) <- per block the sums of
 *     dz and dz z over its rows, of the values it writes (x is z here);
 *   glx_pointmax_scatter_add_scaled_sums: glx_pointmax_scatter_add_scaled + bsum (B, 2, 128) <- per object the same sums of what it ADDS;
 *   glx_bn_backward_from_partials: pa (na, 2, C) + pb (nb, 2, C) -> what glx_bn_backward_sums hands back (coef3 (3 C) = gamma invstd,
 *     sum dz / rows, sum dz xhat / rows; dgamma, dbeta), with sum dz xhat = invstd (sum dz z - mean sum dz); fixed summation order. */
int glx_rows128_affine_blocks(long long rows);
int glx_rows128_affine_f16x2_sums(const float* x, long long rows, const void* Wh, const int32_t* ew, const float* init, float* y,
                                  const float* pre_coef, float* bsum, void* stream);
int glx_pointmax_scatter_add_scaled_sums(const int32_t* arg, const float* coef, const float* chan_scale, const float* W3, int B, int P,
                                         float* dh2, const float* z, const float* pre_coef, float* bsum, void* stream);
int glx_bn_backward_from_partials(const float* pa, int na, const float* pb, int nb, int C, long long rows, const float* gamma,
                                  const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef3, void* stream);
size_t glx_pointmax_wsum_workspace_bytes(void);
int glx_pointmax_wsum(const float* g, const int32_t* arg, const float* h2, int B, int P, float* T, void* workspace,
                      size_t workspace_bytes, void* stream);
int glx_pointmax_wsum_pre(const float* g, const int32_t* arg, const float* h2, const float* pre_coef, int B, int P, float* T,
                          void* workspace, size_t workspace_bytes, void* stream);

/* out[m,c] = max_s relu(a[m,s,c] + b[m,s,c]) on row-major (M, nsample, C) tensors, arg = winning slot
 * (first on ties) -- add + ReLU + max_pool of the RoI-grid pooling MLP in training
 * (pcdet/ops/pointnet2/pointnet2_stack/voxel_pool_modules.py:96-104) in one pass.  _grad: grad_in
 * (M, nsample, C) = grad_out at the winning slot where out > 0, else 0 (the gradient of both a and b). */
int glx_relu_add_max(const float* a, const float* b, int M, int nsample, int C, float* out, int32_t* arg,
                     void* stream);
int glx_relu_add_max_grad(const float* grad_out, const float* out, const int32_t* arg, int M, int nsample,
                          int C, float* grad_in, void* stream);

/* RoI targets of the second stage for a whole batch, two launches, no host round trip
 * (replaces ProposalTargetLayer.sample_rois_for_rcnn + subsample_rois + get_max_iou_with_same_class,
 * pcdet/models/roi_heads/target_assigner/proposal_target_layer.py:65-239).
 * rois (B,R,roi_ld>=7), roi_labels (B,R) int64, gt_boxes (B,G,gt_ld>=8) zero-padded with the class id in
 * the last column.  key (B,R) and pick (B,P) are uniform [0,1) numbers supplied by the caller (foreground
 * order / draws with replacement).  Outputs: max_overlaps (B,R), assignment (B,R) ground-truth index,
 * n_gt (B) trimmed count, sampled (B,P) RoI index, sampled_gt (B,P) ground-truth index or -1. */
int glx_roi_targets(const float* rois, const int64_t* roi_labels, int B, int R, int roi_ld,
                    const float* gt_boxes, int G, int gt_ld, int same_class, const float* key,
                    const float* pick, int P, int fg_per_image, float fg_thresh, float bg_lo, float reg_fg,
                    double hard_ratio, float* max_overlaps, int32_t* assignment, int32_t* n_gt,
                    int32_t* sampled, int32_t* sampled_gt, void* stream);

/* The rest of ProposalTargetLayer.forward (proposal_target_layer.py:13-63) in one launch: rows of the sampled RoIs
 * (B,P,roi_ld), their ground truths (B,P,gt_ld) and label variances gt_unc (B,G,unc_ld) or NULL -- zero rows for
 * a frame without ground truth (sampled_gt < 0) --, IoU / score / label of each, reg_valid (int64) = IoU >
 * reg_fg, classification labels: score_type 0 ("cls") int64 {1, 0, -1 between the thresholds}, 1 ("roi_iou")
 * float {1 above cls_fg, 0 below cls_bg, (iou - bg) * cls_inv_span between; cls_inv_span = 1 / (fg - bg) as the
 * caller's float}.  roi_scores may be NULL. */
int glx_roi_target_gather(const float* rois, const int64_t* roi_labels, const float* roi_scores, int B, int R,
                          int roi_ld, const float* gt_boxes, int G, int gt_ld, const float* gt_unc,
                          int unc_ld, const float* max_overlaps, const int32_t* sampled,
                          const int32_t* sampled_gt, int P, float reg_fg, float cls_fg, float cls_bg,
                          float cls_inv_span, int score_type, float* out_rois, float* out_gt, float* out_iou, float* out_scores,
                          int64_t* out_labels, float* out_unc, int64_t* out_reg_valid, void* out_cls_labels,
                          void* stream);

/* n copy-then-fill regions of 32-bit words in one launch (arrays of length n on the HOST): dst[i][0 .. copy_words[i])
 * = src[i][...] (device to device), dst[i][copy_words[i] .. total_words[i]) = fill[i].  Loads the next batch into the
 * static input buffers of a recorded step (points, frame ids + padding value, zero-padded ground-truth blocks). */
int glx_copy_fill_multi(int n, void* const* dst, const void* const* src, const uint32_t* copy_words,
                        const uint32_t* total_words, const uint32_t* fill, void* stream);

/* The K largest of each frame's A scores in descending order (scores (frames, A) -> top (frames, K), order
 * (frames, K) int64 indices into the frame), equal scores by ascending index; one launch, one block per frame.
 * K <= min(A, glx_topk_max_k()).
 * Replaces: torch.topk(scores, k=NMS_PRE_MAXSIZE) of RoIHeadTemplate.proposal_layer
 * (pcdet/models/roi_heads/roi_head_template.py:81-88). */
int glx_topk_max_k(void);
int glx_topk_desc(const float* scores, int frames, int A, int K, float* top, int64_t* order, void* stream);
/* The same result from TKM_NB = 32 cooperating blocks per frame + a per-frame sort (two launches): `workspace` =
 * glx_topk_workspace_bytes(frames, K) bytes that were ZERO before the first call and are touched by these calls only (they
 * leave it zero); one stream at a time.  A <= 131 072, frames <= 16. */
size_t glx_topk_workspace_bytes(int frames, int K);
int glx_topk_desc_ws(const float* scores, int frames, int A, int K, float* top, int64_t* order, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Arithmetic of glx_conv3x3_forward[_ex] (and of the packs it reads), per process: 1 = f16x2 (default: two fp16 pieces per operand,
 * scaled per output channel / per staged chunk by powers of two, three MFMAs per product tile), 0 = bf16x3 (three bf16 pieces, six
 * MFMAs, products exact to 2^-22 at any magnitude).  env GLX_CONV3X3_ARITH=bf16x3 selects 0 at load.  set returns the previous value;
 * packs written under the other setting must be rewritten (glx_conv3x3_pack) before the next convolution reads them.
 * base_bev_backbone.py:30-49's nn.Conv2d(3 x 3, stride 1) layers. */
int glx_conv3x3_set_arith(int f16x2);
int glx_conv3x3_get_arith(void);
/* Kernel form of glx_conv3x3_wgrad[_ex]: 2 (default) = both operands through LDS, f16x2 with a running exponent over the block's
 * pixel tiles; 1 = the first form (bf16x3, gy from global memory per k-step).  env GLX_WGRAD_FORM=1 at load; returns the previous. */
int glx_conv3x3_set_wgrad_form(int form);
/* glx_conv3x3_pack / glx_conv3x3_pack_multi with the layout named per call / per weight: f16x2 = 1 / 0, -1 = the process setting
 * (f16x2 array NULL = the process setting for all).  glx_conv3x3s2_forward* reads the bf16x3 forward image whatever the setting. */
int glx_conv3x3_pack_arith(const float* W, long long s_co, long long s_ci, long long s_kh, long long s_kw, int Cin, int Cout,
                           void* fwd, void* bwd, int f16x2, void* stream);
int glx_conv3x3_pack_multi_arith(int n, const float* const* W, const long long* strides, const int32_t* Cin, const int32_t* Cout,
                                 void* const* fwd, void* const* bwd, const int32_t* f16x2, void* stream);
/* ---- dense 3x3 convolutions of the BEV backbone (stride 1, zero padding 1, channels-last fp32 maps) ----------------
 * Replaces, for the 3x3 / stride-1 layers: nn.Conv2d(c, c, 3, padding=1, bias=False) and ZeroPad2d(1) + Conv2d(.., 3)
 * of BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py:30-49), which the reference runs through cuDNN.
 * fp32 in, fp32 out; the products run on the bf16 matrix pipe with every operand split into three bf16 pieces
 * (six MFMAs per fp32-equivalent product tile, fp32 accumulation): results agree with an fp32 convolution to
 * fp32 rounding (csrc/glx_conv2d.hip).
 * glx_conv3x3_pack: W (Cout, Cin, 3, 3) with ELEMENT strides (s_co, s_ci, s_kh, s_kw) -> `fwd` (the pieces laid out
 * for the convolution Cin -> Cout; needs Cout % 64 == 0) and / or `bwd` (for its input gradient, the convolution
 * Cout -> Cin of the output gradient with flipped taps; needs Cin % 64 == 0); either may be NULL; both channel counts
 * multiples of 32; each glx_conv3x3_packed_bytes(Cin, Cout) bytes.  Once per weight update.
 * glx_conv3x3_forward: x (B, H, W, Cin) -> y (B, H, W, Cout), `packed` = the pack whose input channels are Cin. */
size_t glx_conv3x3_packed_bytes(int Cin, int Cout);
int glx_conv3x3_pack(const float* W, long long s_co, long long s_ci, long long s_kh, long long s_kw, int Cin, int Cout,
                     void* fwd, void* bwd, void* stream);
/* The same for n weights in one launch: arrays of length n on the HOST; strides[4 i ..] = s_co, s_ci, s_kh, s_kw. */
int glx_conv3x3_pack_multi(int n, const float* const* W, const long long* strides, const int32_t* Cin,
                           const int32_t* Cout, void* const* fwd, void* const* bwd, void* stream);
int glx_conv3x3_forward(const float* x, int B, int H, int W, int Cin, const void* packed, int Cout, float* y,
                        void* stream);
/* With per-call options (glx_conv_opts): opts->bn = the training-mode BatchNorm2d behind the convolution
 * (base_bev_backbone.py:37-49), statistics of y in the epilogue; opts->epilogue = the eval-mode BatchNorm2d (+ ReLU) folded
 * into it, y = relu?(conv * scale[c] + shift[c]) (ldc / coff must be 0 here).  At most one of the two. */
int glx_conv3x3_forward_ex(const float* x, int B, int H, int W, int Cin, const void* packed, int Cout, float* y,
                           const glx_conv_opts* opts, void* stream);
/* The weight gradient of the same convolution: dW (Cout, Cin, 3, 3), written through ELEMENT strides (s_co, s_ci,
 * s_kh, s_kw) (torch keeps the BEV filters in channels-last memory), = sum over pixels of gy (B, H, W, Cout) times
 * the shifted x (B, H, W, Cin); same split-bf16 arithmetic.  Two launches (block partial sums into the workspace,
 * then their sum); dW is overwritten, not accumulated into.  Replaces cuDNN's backward-filter call for these layers. */
size_t glx_conv3x3_wgrad_workspace_bytes(int Cin, int Cout);
int glx_conv3x3_wgrad(const float* x, const float* gy, int B, int H, int W, int Cin, int Cout, float* dW,
                      long long s_co, long long s_ci, long long s_kh, long long s_kw, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same for an input that is transformed on load (`pre` as glx_conv_opts.prologue; NULL = glx_conv3x3_wgrad). */
int glx_conv3x3_wgrad_ex(const float* x, const float* gy, int B, int H, int W, int Cin, int Cout, float* dW,
                         long long s_co, long long s_ci, long long s_kh, long long s_kw, const glx_epilogue* pre,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The two launches apart (as glx_sconv_wgrad_pairs_reduce): glx_conv3x3_wgrad_ex with dW == NULL leaves the blocks' partial
 * sums in `workspace`, glx_conv3x3_wgrad_reduce adds them into dW (strides as above). */
int glx_conv3x3_wgrad_reduce(int Cin, int Cout, float* dW, long long s_co, long long s_ci, long long s_kh, long long s_kw,
                             const void* workspace, size_t workspace_bytes, void* stream);
/* The partial sums of n layers in ONE launch (arrays of n; strides = 4 per layer: s_co, s_ci, s_kh, s_kw): the same sums in the
 * same order as glx_conv3x3_wgrad_reduce.  nn.Conv2d's weight gradient (base_bev_backbone.py:30-49's 3 x 3 layers). */
int glx_conv3x3_wgrad_reduce_multi(int n, const int32_t* Cin, const int32_t* Cout, float* const* dW, const long long* strides,
                                   const void* const* workspace, const size_t* workspace_bytes, void* stream);
/* ---- transposed convolutions with kernel = stride = u in {1, 2}, no padding (BaseBEVBackbone's deblocks,
 * base_bev_backbone.py:51-66: ConvTranspose2d(c, cu, u, stride=u, bias=False)), channels-last fp32 maps, same split-bf16
 * arithmetic (csrc/glx_deconv2d.hip).  W (Cin, Cout, u, u) with ELEMENT strides (s_ci, s_co, s_kh, s_kw); channel counts
 * multiples of 64.  glx_deconv_pack writes the piece images for the forward (`fwd`) and / or the input gradient (`bwd`),
 * glx_deconv_packed_bytes(Cin, Cout, u) bytes each.  x (B, H, W, Cin) -> y (B, u H, u W, Cout); gy -> gx likewise;
 * glx_deconv_wgrad overwrites dW (needs W % 8 == 0).  Replaces cuDNN's three calls for these layers. */
size_t glx_deconv_packed_bytes(int Cin, int Cout, int u);
int glx_deconv_pack(const float* W, long long s_ci, long long s_co, long long s_kh, long long s_kw, int Cin, int Cout,
                    int u, void* fwd, void* bwd, void* stream);
int glx_deconv_forward(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout, int u, float* y,
                       void* stream);
int glx_deconv_input_grad(const float* gy, int B, int H, int W, int Cin, const void* packed_bwd, int Cout, int u,
                          float* gx, void* stream);
/* Forward of the strided layer (ZeroPad2d(1) + Conv2d(c, 2c, 3, stride 2), base_bev_backbone.py:33-36) on the same kernel:
 * x (B, H, W, Cin) with even H, W -> y (B, H/2, W/2, Cout); `packed_fwd` = the forward image of glx_conv3x3_pack.
 * Bit-reproducible (the vendor's forward kernel for this layer sums split-K slices with atomics). */
int glx_conv3x3s2_forward(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout, float* y,
                          void* stream);
/* Inference: the two calls above with an epilogue and a placement (glx_epilogue): y = relu?(acc * scale[c] + shift[c])
 * (scale = shift = NULL: none), written with `ldc` floats between output pixels from channel `coff` of a pixel on
 * (ldc = 0: dense) -- a deblock's map straight into its slice of the concatenated map (base_bev_backbone.py:100-104:
 * torch.cat(ups, dim=1) on channels-last memory costs no copy). */
int glx_deconv_forward_ex(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout, int u, float* y,
                          const glx_epilogue* epilogue, void* stream);
/* The same with the training-mode BatchNorm statistics of y taken in the kernel's epilogue (glx_bn_stats as in
 * glx_conv_opts.bn: coef / saved mean / invstd for glx_bn_apply_forward and glx_bn_relu_backward, running statistics
 * updated) -- the BatchNorm2d behind a deblock's ConvTranspose2d (base_bev_backbone.py:51-66). */
int glx_deconv_forward_bn(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout, int u, float* y,
                          const glx_bn_stats* bn, void* stream);
/* ... with the training-mode BatchNorm statistics of y in the epilogue (glx_bn_stats as for glx_conv3x3_forward_ex). */
int glx_conv3x3s2_forward_bn(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout, float* y,
                             const glx_bn_stats* bn, void* stream);
int glx_conv3x3s2_forward_ex(const float* x, int B, int H, int W, int Cin, const void* packed_fwd, int Cout, float* y,
                             const glx_epilogue* epilogue, void* stream);
/* out (B, 2H, 2W, C) = gy (B, H, W, C) at the even pixels, zero elsewhere (channels-last, C % 4 == 0): the output gradient of
 * ZeroPad2d(1) + Conv2d(c, 2c, 3, stride 2) (base_bev_backbone.py:33-38) spread over the stride-1 map, on which the layer's two
 * gradients are glx_conv3x3_forward (flipped pack) and glx_conv3x3_wgrad -- autograd's two library calls for that layer. */
int glx_spread_stride2(const float* gy, int B, int H, int W, int C, float* out, void* stream);
size_t glx_deconv_wgrad_workspace_bytes(int Cin, int Cout, int u);
int glx_deconv_wgrad(const float* x, const float* gy, int B, int H, int W, int Cin, int Cout, int u, float* dW,
                     long long s_ci, long long s_co, long long s_kh, long long s_kw, void* workspace,
                     size_t workspace_bytes, void* stream);
/* Experiments only (tools/conv_side_load.py): blocks per launch of glx_conv3x3_forward (0 = as many as are resident) and rows
 * per tile (6 / 7 / 8; 0 = chosen per launch by the cost model). */
int glx_conv3x3_set_grid(int blocks, int rows_per_tile);

/* The RoI head's FC towers behind the first Linear, training mode, one launch per direction
 * (pcdet/models/roi_heads/voxelrcnn_kl_label_iou_head.py:38-92: shared_fc_layer[second Linear on] -> cls_fc_layers ->
 * cls_pred_layer and reg_fc_layers -> reg_pred_layer / reg_std_layer -> reg_std_bn -> reg_std_fc1 -> reg_std_bn1 ->
 * reg_std_fc2; the towers are voxelrcnn_head.py:40-66: Linear(bias=False) + BatchNorm1d + ReLU (+ Dropout behind the first
 * layer of a tower)).  Layers are numbered 0 (the 20 736 -> 256 Linear, computed by the caller: z0), 1 (second shared
 * layer), 2-3 (cls tower), 4-5 (reg tower); all are 256 wide.  Every BatchNorm runs on batch statistics over the R rows
 * (biased variance to normalise, running statistics updated with the unbiased one and `momentum`), dropout keeps an element
 * of layers 0 / 2 / 4 when its uniform draw is >= drop_p and scales it by 1 / (1 - drop_p) (nn.Dropout's arithmetic on
 * caller-supplied draws).  All tensors fp32, row-major, contiguous. */
typedef struct glx_fc_bn {
  const float* gamma; const float* beta;       /* (C) */
  float* running_mean; float* running_var;     /* (C), or both NULL */
  float* save_mean; float* save_invstd;        /* (C): written by the forward, read by the backward */
  float eps, momentum;
} glx_fc_bn;

typedef struct glx_fc_tower {
  int R;                                       /* rows (RoIs): a multiple of 16 in 16..1024 */
  float drop_p;                                /* dropout probability, 0 = none (then drop_u is NULL) */
  const float* drop_u;                         /* (3, R, 256) uniform draws for layers 0, 2, 4 */
  const float* z0;                             /* (R, 256) = pooled features x W0^T */
  const float* w[6];                           /* w[1..5] (256 out, 256 in); w[0] unused */
  glx_fc_bn bn[6];
  float* z[6];                                 /* z[1..5] (R, 256): Linear outputs, written by the forward; z[0] unused */
  float* h[6];                                 /* h[0..5] (R, 256): layer outputs, written by the forward */
  const float* w_cls; const float* b_cls;      /* cls_pred_layer (1, 256), (1) */
  const float* w_reg; const float* b_reg;      /* reg_pred_layer (7, 256), (7) */
  const float* w_std; const float* b_std;      /* reg_std_layer (7, 256), (7) */
  glx_fc_bn bn_s7;                             /* reg_std_bn (7) */
  const float* w_fc1; const float* b_fc1;      /* reg_std_fc1 (64, 7), (64) */
  glx_fc_bn bn_s64;                            /* reg_std_bn1 (64) */
  const float* w_fc2; const float* b_fc2;      /* reg_std_fc2 (1, 64), (1) */
  float* ori_cls; float* rcnn_reg; float* rcnn_reg_std; float* std_logit;   /* (R), (R, 7), (R, 7), (R): forward outputs */
  float* scratch;                              /* glx_fc_tower_scratch_bytes(R) */
  unsigned* barrier;                           /* 128 bytes, zero before the first launch; every launch leaves them zero.  Owned
                                                  by ONE tower descriptor / stream at a time; word 24 = error word (below) */
  int cooperative;                             /* 1: ONE launch of 32 co-resident blocks with grid barriers WHEN the runtime
                                                  confirms that they are resident together on the current device (else, and
                                                  with 0, a launch per phase: no residency assumption) */
} glx_fc_tower;

typedef struct glx_fc_tower_grads {
  const float* g_cls; const float* g_logit; const float* g_reg; const float* g_std;   /* dL/d(ori_cls, std_logit, rcnn_reg,
                                                                                         rcnn_reg_std); NULL = zero */
  float* dz[6];                                /* (R, 256): dL/d(Linear output of layer l); dz[0] = dL/dz0 */
  float* dgamma[6]; float* dbeta[6];           /* (256) */
  float* dw_cls; float* db_cls; float* dw_reg; float* db_reg; float* dw_std; float* db_std;
  float* dgamma7; float* dbeta7; float* dw_fc1; float* db_fc1; float* dgamma64; float* dbeta64; float* dw_fc2; float* db_fc2;
  float* scratch;                              /* glx_fc_tower_scratch_bytes(R) */
} glx_fc_tower_grads;

size_t glx_fc_tower_scratch_bytes(int R);
/* 1 when a tower of R rows can run on the current device at all (the phases' LDS fits a block), else 0 -- the caller then
 * keeps the module-by-module path (voxelrcnn_kl_label_iou_head.py:38-92 on library GEMMs).  *cooperative (may be NULL):
 * whether the one-launch forms will be used (occupancy x CUs >= 2 x 32 blocks for both directions). */
int glx_fc_tower_supported(int R, int* cooperative);
/* *gave_up != 0: a one-launch form stopped waiting at a grid barrier (its partner blocks never arrived within ~4 M polls) --
 * the results of that launch are invalid.  Host-synchronising; clears the word. */
int glx_fc_tower_barrier_status(void* barrier, int* gave_up);
int glx_fc_tower_forward(const glx_fc_tower* t, void* stream);
/* `t` as the forward left it (z, h, save_mean / save_invstd, rcnn_reg_std, drop_u).  The Linear weight gradients
 * dW_l = dz[l]^T h[l-1] (dW_0 = dz[0]^T pooled) and the pooled features' gradient dz[0] W0 are GEMMs left to the caller. */
int glx_fc_tower_backward(const glx_fc_tower* t, const glx_fc_tower_grads* g, void* stream);

/* Tail of the proposal layer (pcdet/models/roi_heads/roi_head_template.py:63-126: selected = keep[:NMS_POST_MAXSIZE],
 * rois / roi_scores / roi_labels rows, zero rows behind the survivors) in one launch.  cand (F, K, C) top-k candidates,
 * top (F, K) their scores, lab (F, A) class index per anchor, order (F, K) anchor index of every candidate,
 * keep (F, keep_stride) / num (F): the NMS keep list and its length.  Outputs rois (F, P, C), scores (F, P),
 * labels (F, P) = class + 1 (1 in empty slots, as the reference's `roi_labels + 1` on its zero-initialised rows). */
int glx_gather_proposals(const float* cand, const float* top, const int64_t* lab, const int64_t* order,
                         const int64_t* keep, const int* num, int F, int A, int K, int keep_stride, int P, int C,
                         float* rois, float* scores, int64_t* labels, void* stream);

/* Stage stamps: a one-thread launch on `stream` that stores the device's constant 100 MHz wall clock into
 * stamps[slot] (10 ns per tick).  Recorded into a captured training step at its stage boundaries
 * (StaticTrainPipeline.mark) it times the stages inside graph replays; the reference has no counterpart (it times
 * stages with host clocks around the calls, tools/train_utils/train_utils.py:18-90). */
int glx_stamp(unsigned long long* stamps, int slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GLENET_HIP_H_ */
