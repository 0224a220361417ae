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
            // bounding-circle reject: boxes whose centres are further apart than the sum of their half-diagonals cannot overlap, so
            // iou_bev would return exactly 0 (never > thresh for thresh >= 0); only the few close pairs pay for the polygon clip
            const float cx = cur[0], cy = cur[1], cr = 0.5f * sqrtf(cur[3] * cur[3] + cur[4] * cur[4]);
            for (int j = start; j < col_size; ++j) {
                const float *o = blk + j * 7;
                const float dx = o[0] - cx, dy = o[1] - cy, rr = cr + 0.5f * sqrtf(o[3] * o[3] + o[4] * o[4]) + 2e-2f;   // 2e-2 > the 1e-2 inside-box margin of the clip
                if (thresh >= 0.f && dx * dx + dy * dy > rr * rr) continue;
                if (iou_bev(cur, o) > thresh) t |= 1ULL << j;
            }
        }
        mask[(int64_t)i * col_blocks + col_blk] = t;
    }
}

// Greedy pass by one wavefront, one 64-box block of the matrix at a time.  Lane l owns the removed-bits words l, l+64, ...
// Inside a block the decisions are sequential (box i is kept unless a kept box before it suppressed it) but only need the 64
// DIAGONAL words of the block's rows, which are loaded once (lane l <- row 64*cb + l) and walked with register shuffles; the kept
// rows' other words are then OR-ed into the removed-bits words of the later blocks by independent, back-to-back loads.
__global__ __launch_bounds__(NMS_W) void k_nms_reduce(int n, const unsigned long long *__restrict__ mask, long long *keep, int *num_keep) {
    constexpr int MAXW = 16;                       // words per lane -> up to 64*16*64 = 65536 boxes
    const int lane = threadIdx.x;
    const int col_blocks = (n + NMS_W - 1) / NMS_W;
    unsigned long long remv[MAXW];
#pragma unroll
    for (int k = 0; k < MAXW; ++k) remv[k] = 0;
    int count = 0;
    for (int cb = 0; cb < col_blocks; ++cb) {
        const int row = cb * NMS_W + lane;
        const unsigned long long diag = row < n ? mask[(int64_t)row * col_blocks + cb] : 0ULL;
        unsigned long long removed = 0;              // removed-bits word of this block (wave-uniform)
#pragma unroll
        for (int k = 0; k < MAXW; ++k)
            if ((cb >> 6) == k) removed = __shfl(remv[k], cb & 63, NMS_W);
        const int in_block = min(NMS_W, n - cb * NMS_W);
        unsigned long long kept = 0;
        for (int i = 0; i < in_block; ++i) {
            const unsigned long long d = __shfl(diag, i, NMS_W);
            if (!((removed >> i) & 1ULL)) {
                kept |= 1ULL << i;
                removed |= d;
            }
        }
        // keep list (ascending) and the propagation to the later blocks
        const int rank = __popcll(kept & ((1ULL << lane) - 1ULL));
        if ((kept >> lane) & 1ULL) keep[count + rank] = row;
        count += __popcll(kept);
        unsigned long long k2 = kept;
        while (k2) {
            const int i = __ffsll((long long)k2) - 1;
            k2 &= k2 - 1;
            const int64_t r = (int64_t)(cb * NMS_W + i) * col_blocks;
#pragma unroll
            for (int k = 0; k < MAXW; ++k) {
                const int cw = lane + 64 * k;
                if (cw > cb && cw < col_blocks) remv[k] |= mask[r + cw];
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
