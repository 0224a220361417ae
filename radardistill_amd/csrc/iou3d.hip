// Rotated-rectangle BEV overlap of aligned box pairs (IoU-head loss target).  Replaces the reference's
// boxes_aligned_overlap_kernel (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:266-277 over box_overlap :104-225; host
// wrapper iou3d_nms.cpp:50-72 boxes_aligned_overlap_bev_gpu).  Same algorithm and constants (EPS 1e-8, MARGIN 1e-2):
// clip polygon = 16 edge/edge intersections + corners inside the other box, sorted by angle, shoelace area.
// One lane per pair (N <= B*500 positives); launched on the caller's stream (the reference uses the legacy default
// stream, an implicit device sync).
#include "iou3d_dev.hpp"

using namespace rd;

__global__ void k_aligned_overlap(int n, const float *__restrict__ a, const float *__restrict__ b, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = overlap_area(a + (int64_t)i * 7, b + (int64_t)i * 7);
}

// (na, nb) overlap matrix: boxes_overlap_kernel (iou3d_nms_kernel.cu:236-249), host boxes_overlap_bev_gpu (iou3d_nms.cpp:29-48)
__global__ void k_pair_overlap(int na, const float *__restrict__ a, int nb, const float *__restrict__ b, float *out) {
    const int ib = blockIdx.x * blockDim.x + threadIdx.x, ia = blockIdx.y;
    if (ia < na && ib < nb) out[(int64_t)ia * nb + ib] = overlap_area(a + (int64_t)ia * 7, b + (int64_t)ib * 7);
}

extern "C" int rd_boxes_overlap_bev(int na, const float *boxes_a, int nb, const float *boxes_b, float *ans_overlap, void *stream) {
    RD_REQUIRE(na >= 0 && nb >= 0 && na <= 65535, "rd_boxes_overlap_bev: bad counts (na <= 65535)");
    if (na == 0 || nb == 0) return RD_OK;
    k_pair_overlap<<<dim3((unsigned)cdiv(nb, 64), (unsigned)na), 64, 0, S(stream)>>>(na, boxes_a, nb, boxes_b, ans_overlap);
    return check_launch("rd_boxes_overlap_bev");
}

extern "C" int rd_boxes_aligned_overlap_bev(int n, const float *boxes_a, const float *boxes_b, float *ans_overlap, void *stream) {
    RD_REQUIRE(n >= 0, "rd_boxes_aligned_overlap_bev: negative count");
    if (n == 0) return RD_OK;
    k_aligned_overlap<<<cdiv(n, 64), 64, 0, S(stream)>>>(n, boxes_a, boxes_b, ans_overlap);
    return check_launch("rd_boxes_aligned_overlap_bev");
}
