// Rotated-BEV non-maximum suppression, entirely on the device.  See include/rdamd.h section M.
// Replaces iou3d_nms_cuda.nms_gpu (pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:137-183 + nms_kernel, iou3d_nms_kernel.cu:295-340):
// the reference builds the 64-wide suppression bit matrix on the GPU, copies it to the host and runs the greedy pass on the CPU
// (a device sync + N*N/8 bytes over PCIe per call); here the greedy pass is a second kernel of ONE wavefront (64 lanes = the
// 64 words of a matrix row chunk), and the keep list and its length stay in device memory.
// Boxes must already be sorted by descending score (the Python wrapper sorts, like iou3d_nms_utils.nms_gpu:127-137).
#include "iou3d_dev.hpp"

using namespace rd;

constexpr int NMS_W = 64;   // boxes per bit-matrix word = wavefront width

// mask[i][cb] bit j = iou_bev(box i, box 64*cb + j) > thresh, for j > i only (upper triangle)
__global__ __launch_bounds__(NMS_W) void k_nms_mask(int n, float thresh, const float *__restrict__ boxes, unsigned long long *mask) {
    const int row_blk = blockIdx.y, col_blk = blockIdx.x;
    const int col_blocks = (n + NMS_W - 1) / NMS_W;
    const int row_size = min(n - row_blk * NMS_W, NMS_W), col_size = min(n - col_blk * NMS_W, NMS_W);
    __shared__ float blk[NMS_W * 7];
    if (threadIdx.x < col_size)
        for (int k = 0; k < 7; ++k) blk[threadIdx.x * 7 + k] = boxes[(int64_t)(NMS_W * col_blk + threadIdx.x) * 7 + k];
    __syncthreads();
    if (threadIdx.x < row_size) {
        const int i = NMS_W * row_blk + threadIdx.x;
        unsigned long long t = 0;
        if (col_blk >= row_blk) {          // lower-triangle words are never read by the greedy pass
            const float *cur = boxes + (int64_t)i * 7;
            const int start = (row_blk == col_blk) ? threadIdx.x + 1 : 0;
            for (int j = start; j < col_size; ++j)
                if (iou_bev(cur, blk + j * 7) > thresh) t |= 1ULL << j;
        }
        mask[(int64_t)i * col_blocks + col_blk] = t;
    }
}

// Greedy pass by one wavefront: lane l owns the removed-bits words l, l+64, ...; rows are fetched 16 at a time so that the
// dependent chain sees one memory latency per 16 boxes.
__global__ __launch_bounds__(NMS_W) void k_nms_reduce(int n, const unsigned long long *__restrict__ mask, long long *keep, int *num_keep) {
    constexpr int MAXW = 16;                       // words per lane -> up to 64*16*64 = 65536 boxes
    const int lane = threadIdx.x;
    const int col_blocks = (n + NMS_W - 1) / NMS_W;
    unsigned long long remv[MAXW];
#pragma unroll
    for (int k = 0; k < MAXW; ++k) remv[k] = 0;
    int count = 0;
    for (int i0 = 0; i0 < n; i0 += 16) {
        // this lane's first word (covers col_blocks <= 64, i.e. n <= 4096) of the next 16 rows, fetched together
        unsigned long long pre[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pre[r] = (i0 + r < n && lane < col_blocks) ? mask[(int64_t)(i0 + r) * col_blocks + lane] : 0ULL;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + r;
            if (i < n) {
                const int w = i >> 6;             // word w lives in slot w >> 6 of lane w & 63 (wave-uniform read)
                unsigned long long word = 0;
#pragma unroll
                for (int k = 0; k < MAXW; ++k)
                    if ((w >> 6) == k) word = __shfl(remv[k], w & 63, NMS_W);
                if (!((word >> (i & 63)) & 1ULL)) {
                    if (lane == 0) keep[count] = i;
                    ++count;
                    remv[0] |= pre[r];
#pragma unroll
                    for (int k = 1; k < MAXW; ++k) {   // n > 4096 only
                        const int cw = lane + 64 * k;
                        if (cw < col_blocks) remv[k] |= mask[(int64_t)i * col_blocks + cw];
                    }
                }
            }
        }
    }
    if (lane == 0) *num_keep = count;
}

extern "C" int64_t rd_nms_ws_bytes(int n) { return (int64_t)n * cdiv(n, NMS_W) * 8; }

extern "C" int rd_nms_bev(int n, const float *boxes_sorted, float thresh, void *mask_ws, int64_t ws_bytes, int64_t *keep, int32_t *num_keep,
                          void *stream) {
    RD_REQUIRE(n >= 0 && n <= 65535, "rd_nms_bev: n=%d outside 0..65535", n);
    RD_REQUIRE(num_keep != nullptr, "rd_nms_bev: num_keep is NULL");
    hipStream_t st = S(stream);
    if (n == 0) {
        RD_HIP(hipMemsetAsync(num_keep, 0, 4, st));
        return RD_OK;
    }
    RD_REQUIRE(ws_bytes >= rd_nms_ws_bytes(n), "rd_nms_bev: workspace too small");
    const int cb = (int)cdiv(n, NMS_W);
    k_nms_mask<<<dim3(cb, cb), NMS_W, 0, st>>>(n, thresh, boxes_sorted, reinterpret_cast<unsigned long long *>(mask_ws));
    k_nms_reduce<<<1, NMS_W, 0, st>>>(n, reinterpret_cast<const unsigned long long *>(mask_ws), reinterpret_cast<long long *>(keep), num_keep);
    return check_launch("rd_nms_bev");
}
