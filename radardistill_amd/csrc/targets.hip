// CenterHead target assignment on the GPU.  Replaces the CPU Python loops of
//   pcdet/models/dense_heads/radar_center_head.py:128-252 (assign_target_of_single_head / assign_targets) and
//   pcdet/models/model_utils/centernet_utils.py:9-69 (gaussian_radius, gaussian2D, draw_gaussian_to_heatmap),
// which the reference runs per head x per sample on the host, after copying gt_boxes back from the device.
// One workgroup per (sample, box).  All fp32 arithmetic uses the non-contracting __f*_rn intrinsics in the reference's
// operation order, so centre cells, radii, inds and masks are bit-exact; the gaussian is evaluated in double like
// numpy does and cast to fp32; overlapping gaussians combine with atomicMax on the fp32 bit pattern (values > 0 order
// like unsigned integers), which is order-independent, hence deterministic.
#include "common.hpp"

using namespace rd;

__device__ __forceinline__ float gaussian_radius_f32(float height, float width, float ov) {
    // centernet_utils.py:9-35, python scalars folded exactly as python does before they meet the fp32 tensor
    const float one_m = (float)(1.0 - (double)ov), one_p = (float)(1.0 + (double)ov);
    const float b1 = __fadd_rn(height, width);
    const float c1 = __fdiv_rn(__fmul_rn(__fmul_rn(width, height), one_m), one_p);
    const float sq1 = __fsqrt_rn(__fsub_rn(__fmul_rn(b1, b1), __fmul_rn(4.0f, c1)));
    const float r1 = __fdiv_rn(__fadd_rn(b1, sq1), 2.0f);
    const float b2 = __fmul_rn(2.0f, __fadd_rn(height, width));
    const float c2 = __fmul_rn(__fmul_rn(one_m, width), height);
    const float sq2 = __fsqrt_rn(__fsub_rn(__fmul_rn(b2, b2), __fmul_rn(16.0f, c2)));
    const float r2 = __fdiv_rn(__fadd_rn(b2, sq2), 2.0f);
    const float a3x4 = (float)(4.0 * (4.0 * (double)ov));                 // 4 * a3, a3 = 4 * min_overlap (python floats)
    const float b3 = __fmul_rn((float)(-2.0 * (double)ov), __fadd_rn(height, width));
    const float c3 = __fmul_rn(__fmul_rn((float)((double)ov - 1.0), width), height);
    const float sq3 = __fsqrt_rn(__fsub_rn(__fmul_rn(b3, b3), __fmul_rn(a3x4, c3)));
    const float r3 = __fdiv_rn(__fadd_rn(b3, sq3), 2.0f);
    return fminf(fminf(r1, r2), r3);
}

__global__ __launch_bounds__(128) void k_center_targets(const float *__restrict__ gt, int B, int M, int box_dim, rd_target_cfg cfg,
                                                        float *hm, float *tb, int64_t *inds, int64_t *masks, float *gtbox) {
    const int b = blockIdx.x / M, i = blockIdx.x % M;
    const float *g = gt + ((int64_t)b * M + i) * box_dim;
    const int cls = (int)g[box_dim - 1];
    if (cls <= 0 || cls > cfg.n_classes) return;              // 0 = padding ('bg')
    const int h = cfg.head_of_class[cls], local = cfg.local_of_class[cls];
    // slot = rank of this box among the boxes of the same head in this sample (original order, radar_center_head.py:215-223)
    int k = 0;
    for (int j = 0; j < i; ++j) {
        const int cj = (int)gt[((int64_t)b * M + j) * box_dim + box_dim - 1];
        if (cj > 0 && cj <= cfg.n_classes && cfg.head_of_class[cj] == h) ++k;
    }
    if (k >= cfg.max_objs) return;
    const float stride = (float)cfg.stride;
    float cx = __fdiv_rn(__fdiv_rn(__fsub_rn(g[0], cfg.pcr0), cfg.vs0), stride);
    float cy = __fdiv_rn(__fdiv_rn(__fsub_rn(g[1], cfg.pcr1), cfg.vs1), stride);
    cx = fminf(fmaxf(cx, 0.f), (float)((double)cfg.fx - 0.5));
    cy = fminf(fmaxf(cy, 0.f), (float)((double)cfg.fy - 0.5));
    const int cxi = (int)cx, cyi = (int)cy;
    const float dx = __fdiv_rn(__fdiv_rn(g[3], cfg.vs0), stride), dy = __fdiv_rn(__fdiv_rn(g[4], cfg.vs1), stride);
    int radius = (int)gaussian_radius_f32(dx, dy, cfg.overlap);
    radius = max(radius, cfg.min_radius);
    if (!(dx > 0.f && dy > 0.f)) return;
    if (!(cxi >= 0 && cxi <= cfg.fx && cyi >= 0 && cyi <= cfg.fy)) return;
    const int64_t slot = ((int64_t)h * B + b) * cfg.max_objs + k;
    if (threadIdx.x == 0) {
        inds[slot] = (int64_t)cyi * cfg.fx + cxi;
        masks[slot] = 1;
        float *t = tb + slot * box_dim;
        t[0] = __fsub_rn(cx, (float)cxi);
        t[1] = __fsub_rn(cy, (float)cyi);
        t[2] = g[2];
        t[3] = logf(g[3]); t[4] = logf(g[4]); t[5] = logf(g[5]);
        t[6] = cosf(g[6]); t[7] = sinf(g[6]);
        for (int q = 8; q < box_dim; ++q) t[q] = g[q - 1];      // velocities: gt_boxes[k, 7:-1]
        float *gb = gtbox + slot * 7;
        for (int q = 0; q < 7; ++q) gb[q] = g[q];
    }
    // draw_gaussian_to_heatmap: clipped (2r+1)^2 window, sigma = (2r+1)/6, elementwise max
    const int left = min(cxi, radius), right = min(cfg.fx - cxi, radius + 1);
    const int top = min(cyi, radius), bottom = min(cfg.fy - cyi, radius + 1);
    const int ww = left + right, wh = top + bottom;
    if (ww <= 0 || wh <= 0) return;
    const double sigma = (double)(2 * radius + 1) / 6.0;
    const double inv = 1.0 / (2.0 * sigma * sigma);
    float *plane = hm + (((int64_t)b * cfg.n_channels + cfg.chan_off[h] + local) * cfg.fy) * cfg.fx;
    for (int e = threadIdx.x; e < ww * wh; e += blockDim.x) {
        const int oy = e / ww - top, ox = e % ww - left;
        double v = exp(-((double)(ox * ox) + (double)(oy * oy)) * inv);
        if (v < 2.220446049250313e-16) v = 0.0;               // h[h < eps * h.max()] = 0 with h.max() == 1
        const float f = (float)v;
        if (f > 0.f) atomicMax(reinterpret_cast<unsigned int *>(plane + (int64_t)(cyi + oy) * cfg.fx + (cxi + ox)), __float_as_uint(f));
    }
}

extern "C" int rd_center_targets(const float *gt_boxes, int B, int M, int box_dim, const rd_target_cfg *cfg, float *heatmaps,
                                 float *target_boxes, int64_t *inds, int64_t *masks, float *gt_box, void *stream) {
    RD_REQUIRE(cfg != nullptr && B > 0 && M >= 0 && box_dim >= 8 && box_dim <= 16, "rd_center_targets: bad sizes");
    RD_REQUIRE(cfg->n_heads >= 1 && cfg->n_heads <= 8 && cfg->n_classes >= 1 && cfg->n_classes <= 15 && cfg->n_channels >= 1,
               "rd_center_targets: bad class configuration");
    RD_REQUIRE(cfg->fx > 0 && cfg->fy > 0 && cfg->max_objs > 0 && cfg->stride > 0, "rd_center_targets: bad geometry");
    hipStream_t st = S(stream);
    const int64_t slots = (int64_t)cfg->n_heads * B * cfg->max_objs;
    RD_HIP(hipMemsetAsync(heatmaps, 0, (size_t)B * cfg->n_channels * cfg->fy * cfg->fx * 4, st));
    {   // the four per-slot outputs: one fill when the caller carved them out of one allocation in this order (the Python host does)
        char *tb = reinterpret_cast<char *>(target_boxes), *in = reinterpret_cast<char *>(inds), *mk = reinterpret_cast<char *>(masks),
             *gb = reinterpret_cast<char *>(gt_box);
        const size_t n_tb = (size_t)slots * box_dim * 4, n_in = (size_t)slots * 8, n_gb = (size_t)slots * 7 * 4;
        if (in == tb + n_tb && mk == in + n_in && gb == mk + n_in) {
            RD_HIP(hipMemsetAsync(tb, 0, n_tb + 2 * n_in + n_gb, st));
        } else {
            RD_HIP(hipMemsetAsync(tb, 0, n_tb, st));
            RD_HIP(hipMemsetAsync(in, 0, n_in, st));
            RD_HIP(hipMemsetAsync(mk, 0, n_in, st));
            RD_HIP(hipMemsetAsync(gb, 0, n_gb, st));
        }
    }
    if (M == 0) return RD_OK;
    k_center_targets<<<B * M, 128, 0, st>>>(gt_boxes, B, M, box_dim, *cfg, heatmaps, target_boxes, inds, masks, gt_box);
    return check_launch("rd_center_targets");
}
