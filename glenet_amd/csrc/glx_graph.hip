// Graph surgery for a ROCm 7.2 defect: a hipMemsetAsync recorded into a HIP graph fills its buffer with the requested value on
// the FIRST launch of the graph and with a stale 16-byte pattern on every later launch (tools/graph_memset_repro.py; the
// buffer of a 64-byte memset of zeros holds "00 00 a0 14 e0 77 00 00 ..." afterwards -- host addresses by the look of them).
// Library code inside a recorded step does use memsets: torch's multi-block reductions zero their semaphores with one
// (aten/src/ATen/native/cuda/Reduce.cuh), so from the second replay on no block finds itself the last one and most outputs
// are never written (tools/graph_reduce_repro.py) -- the cause of the NaN gradients of the recorded CVAE training step, and a
// memset node sits in the recorded GLENet-VR training step as well.
// glx_graph_replace_memsets walks a captured hipGraph_t BEFORE it is instantiated and replaces every memset node by a
// kernel node that does the same fill (same dependencies, same dependents), which replays correctly.
#include <vector>

#include "glx_common.h"

__global__ void k_graph_fill(unsigned char* __restrict__ dst, unsigned value, unsigned elem, unsigned long long width,
                             unsigned long long height, unsigned long long pitch) {
  const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long nthreads = (unsigned long long)gridDim.x * blockDim.x;
  if (height <= 1) {
    // one row of width * elem bytes: 16-byte stores over the aligned middle (library memsets are tens of MB -- the
    // vendor convolutions zero their outputs with them), single elements at the ragged ends
    const unsigned word = elem == 4 ? value : elem == 2 ? (value & 0xFFFFu) * 0x10001u : (value & 0xFFu) * 0x01010101u;
    const unsigned long long bytes = width * elem;
    const unsigned long long head = ((16 - ((unsigned long long)dst & 15)) & 15) < bytes ? ((16 - ((unsigned long long)dst & 15)) & 15) : bytes;
    const unsigned long long nvec = (bytes - head) / 16, tail0 = head + nvec * 16;
    uint4* v = reinterpret_cast<uint4*>(dst + head);
    for (unsigned long long i = tid; i < nvec; i += nthreads) v[i] = make_uint4(word, word, word, word);
    // head and tail: whole elements (dst is elem-aligned, so head and tail0 are multiples of elem)
    for (unsigned long long b = tid * elem; b < head; b += nthreads * elem) {
      if (elem == 4) *reinterpret_cast<unsigned*>(dst + b) = value;
      else if (elem == 2) *reinterpret_cast<unsigned short*>(dst + b) = (unsigned short)value;
      else dst[b] = (unsigned char)value;
    }
    for (unsigned long long b = tail0 + tid * elem; b < bytes; b += nthreads * elem) {
      if (elem == 4) *reinterpret_cast<unsigned*>(dst + b) = value;
      else if (elem == 2) *reinterpret_cast<unsigned short*>(dst + b) = (unsigned short)value;
      else dst[b] = (unsigned char)value;
    }
    return;
  }
  const unsigned long long n = width * height;
  for (unsigned long long i = tid; i < n; i += nthreads) {
    const unsigned long long row = i / width, col = i - row * width;
    unsigned char* p = dst + row * pitch + col * elem;
    if (elem == 4) *reinterpret_cast<unsigned*>(p) = value;
    else if (elem == 2) *reinterpret_cast<unsigned short*>(p) = (unsigned short)value;
    else *p = (unsigned char)value;
  }
}

// Replaces the memset nodes of ONE graph level; child-graph nodes (hipGraphNodeTypeGraph: a library that records a
// sub-graph of its own) are visited recursively -- their embedded graphs replay through the same executor path.
static int replace_memsets_in(hipGraph_t graph, int depth, int* done) {
  GLX_REQUIRE(depth < 8, "glx_graph_replace_memsets: child graphs nested deeper than 8 levels");
  size_t n = 0;
  GLX_HIP(hipGraphGetNodes(graph, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  if (n) GLX_HIP(hipGraphGetNodes(graph, nodes.data(), &n));
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType type;
    GLX_HIP(hipGraphNodeGetType(nodes[i], &type));
    if (type == hipGraphNodeTypeGraph) {
      hipGraph_t child = nullptr;
      GLX_HIP(hipGraphChildGraphNodeGetGraph(nodes[i], &child));
      if (child) {
        const int rc = replace_memsets_in(child, depth + 1, done);
        if (rc != GLX_OK) return rc;
      }
      continue;
    }
    if (type != hipGraphNodeTypeMemset) continue;
    hipMemsetParams mp;
    GLX_HIP(hipGraphMemsetNodeGetParams(nodes[i], &mp));
    size_t nd = 0, nt = 0;
    GLX_HIP(hipGraphNodeGetDependencies(nodes[i], nullptr, &nd));
    std::vector<hipGraphNode_t> deps(nd);
    if (nd) GLX_HIP(hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd));
    GLX_HIP(hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nt));
    std::vector<hipGraphNode_t> outs(nt);
    if (nt) GLX_HIP(hipGraphNodeGetDependentNodes(nodes[i], outs.data(), &nt));
    unsigned char* dst = (unsigned char*)mp.dst;
    unsigned value = mp.value, elem = mp.elementSize ? mp.elementSize : 1;
    unsigned long long width = mp.width, height = mp.height ? mp.height : 1, pitch = height > 1 ? mp.pitch : width * elem;
    void* args[] = {&dst, &value, &elem, &width, &height, &pitch};
    hipKernelNodeParams kp = {};
    kp.func = (void*)k_graph_fill;
    const unsigned long long total = height > 1 ? width * height : (width * elem + 15) / 16 + 32;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    kp.gridDim = dim3(blocks);
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipGraphNode_t fill;
    GLX_HIP(hipGraphAddKernelNode(&fill, graph, nd ? deps.data() : nullptr, nd, &kp));
    for (size_t k = 0; k < nt; ++k) GLX_HIP(hipGraphAddDependencies(graph, &fill, &outs[k], 1));
    GLX_HIP(hipGraphDestroyNode(nodes[i]));
    ++*done;
  }
  return GLX_OK;
}

// graph: a hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()) that has not been instantiated yet.
// n_replaced (host int, may be NULL) receives the number of memset nodes that were replaced, child graphs included.
extern "C" int glx_graph_replace_memsets(void* graph_, int* n_replaced) {
  hipGraph_t graph = (hipGraph_t)graph_;
  GLX_REQUIRE(graph, "glx_graph_replace_memsets: null graph");
  int done = 0;
  const int rc = replace_memsets_in(graph, 0, &done);
  if (n_replaced) *n_replaced = done;
  return rc;
}
