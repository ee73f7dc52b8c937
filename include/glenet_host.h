/* glenet_host.h -- C ABI of libglenet_host.so: the entry points of the hot path whose CONTRACT is "host memory,
 * host arithmetic, callable from forked DataLoader worker processes" (SURVEY.md 8b).  Plain C++ compiled with g++;
 * the library does not link or load the HIP runtime, keeps no global state and is re-entrant, so a worker that
 * was forked from a process owning a GPU context can call it (tests/test_host_cpu.py does exactly that).
 *
 * Conventions: caller-owned host buffers, row-major float32 / int32, return 0 or a negative errno-style code.
 */
#ifndef GLENET_HOST_H_
#define GLENET_HOST_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int glxh_abi_version(void);

/* Rotated BEV IoU of boxes (N,7) x (M,7) [x,y,z,dx,dy,dz,heading] -> out (N,M).
 * Replaces: iou3d_nms_cuda.boxes_iou_bev_cpu (pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232-252; callers
 * iou3d_nms_utils.py:52-68 boxes_bev_iou_cpu <- database_sampler.py:246-247, nms_func iou3d_nms_utils.py:211). */
int glxh_boxes_iou_bev(const float* boxes_a, int N, const float* boxes_b, int M, float* out);

/* The older `iou3d` library's CPU twins on 5-float BEV boxes [x1,y1,x2,y2,ry] (corner construction, rotation
 * sign and the 1e-5 containment margin of that file): overlap area / IoU, out (N,M).
 * Replaces: iou3d_cuda.boxes_overlap_bev_cpu / boxes_iou_bev_cpu (pcdet/ops/iou3d/src/iou3d_cpu.cpp:232-282). */
int glxh_iou3d_boxes_overlap_bev(const float* boxes_a, int N, const float* boxes_b, int M, float* out);
int glxh_iou3d_boxes_iou_bev(const float* boxes_a, int N, const float* boxes_b, int M, float* out);

/* out (N,P) int32: 1 where point j lies in box i (z test |z - cz| <= dz/2, xy test with the 1e-2 margin).
 * Replaces: roiaware_pool3d_cuda.points_in_boxes_cpu (pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:143-168;
 * callers roiaware_pool3d_utils.py:9-28 <- box_utils.py:86, augmentor_utils.py, database_sampler.py). */
int glxh_points_in_boxes(const float* boxes, int N, const float* points, int P, int32_t* out);

/* Hard voxelization of one frame: points (P,C) -> voxels (max_voxels,max_points,C) zero-padded, coords
 * (max_voxels,3) [z,y,x], num_points (max_voxels); *num_voxels = voxels produced (first-seen order, max_voxels
 * then max_points truncation).  range = [xmin,ymin,zmin,xmax,ymax,zmax], grid = [gx,gy,gz].
 * Replaces: spconv.utils.VoxelGeneratorV2.generate / Point2VoxelCPU3d.point_to_voxel as called from
 * pcdet/datasets/processor/data_processor.py:15-60 (third-party arithmetic). */
int glxh_voxelize_hard(const float* points, int P, int C, const float* range, const float* voxel_size,
                       const int* grid, int max_points, int max_voxels, float* voxels, int32_t* coords,
                       int32_t* num_points, int* num_voxels);

#ifdef __cplusplus
}
#endif
#endif /* GLENET_HOST_H_ */
