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
  const unsigned long long n = width * height;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const unsigned long long row = i / width, col = i - row * width;
    unsigned char* p = dst + row * pitch + col * elem;
    if (elem == 4) *reinterpret_cast<unsigned*>(p) = value;
    else if (elem == 2) *reinterpret_cast<unsigned short*>(p) = (unsigned short)value;
    else *p = (unsigned char)value;
  }
}

// graph: a hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()) that has not been instantiated yet.
// n_replaced (host int, may be NULL) receives the number of memset nodes that were replaced.
extern "C" int glx_graph_replace_memsets(void* graph_, int* n_replaced) {
  hipGraph_t graph = (hipGraph_t)graph_;
  GLX_REQUIRE(graph, "glx_graph_replace_memsets: null graph");
  size_t n = 0;
  GLX_HIP(hipGraphGetNodes(graph, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  if (n) GLX_HIP(hipGraphGetNodes(graph, nodes.data(), &n));
  int done = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType type;
    GLX_HIP(hipGraphNodeGetType(nodes[i], &type));
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
    const unsigned long long total = width * height;
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
    ++done;
  }
  if (n_replaced) *n_replaced = done;
  return GLX_OK;
}
