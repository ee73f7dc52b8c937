// pybind glue for the reference's own pcdet/ops/iou3d/src/iou3d_cpu.cpp (compiled from
// /root/reference where it lies; nothing of it is copied here).  The reference keeps its
// PYBIND11_MODULE in iou3d.cpp next to the CUDA entry points, which cannot build here
// (needs cuda.h), so the three CPU functions are bound from this TU instead.
#include <torch/extension.h>

int boxes_overlap_bev_cpu(at::Tensor boxes_a, at::Tensor boxes_b, at::Tensor ans_overlap);
int boxes_iou_bev_cpu(at::Tensor boxes_a, at::Tensor boxes_b, at::Tensor ans_iou);
int boxes_iou3d_cpu(at::Tensor boxes_a, at::Tensor boxes_b, at::Tensor ans_iou);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("boxes_overlap_bev_cpu", &boxes_overlap_bev_cpu);
  m.def("boxes_iou_bev_cpu", &boxes_iou_bev_cpu);
  m.def("boxes_iou3d_cpu", &boxes_iou3d_cpu);
}
