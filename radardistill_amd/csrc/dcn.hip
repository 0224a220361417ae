// DCNv2 (modulated deformable convolution) for channels-last maps.  Replaces the reference's in-tree CUDA op
//   pcdet/ops/basicblock/src/cuda/modulated_deform_im2col_cuda.cuh:127-194 (im2col), :196-254 (col2im, atomics),
//   :256-328 (col2im_coord) and modulated_deform_conv_cuda.cu:19-280 (im2col + cuBLAS GEMMs).
// MI355X design: no im2col buffer in the forward.  rd_dcn_prep turns the offset/mask maps into a sampling table
// (4 corner rows + 4 weights per (output pixel, tap), mask folded in); the implicit-GEMM kernel of conv.hip gathers
// and blends the corner rows while it stages its A tile (index mode DEFORM), so the 2304 x BHW column matrix never
// exists.  Backward: column gradient = one linear layer on the same kernel; rd_dcn_bwd_data turns it into the input
// gradient (row-shaped fp32 atomics, 1 KiB contiguous per instruction) and the offset / mask gradients (wave reductions).
#include "common.hpp"

using namespace rd;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DcnGeom {
    int B, H, W, Ho, Wo, KH, KW, stride, pad, dil;
};

struct Sample {
    int idx[4];
    float w[4];      // bilinear corner weights (no mask)
    float lh, lw;    // fractional parts
    bool inside;
};

__device__ __forceinline__ Sample dcn_sample(const DcnGeom &g, int j, int t, float off_h, float off_w) {
    Sample s;
    const int wo = j % g.Wo, ho = (j / g.Wo) % g.Ho, b = j / (g.Wo * g.Ho);
    const int ky = t / g.KW, kx = t % g.KW;
    // modulated_deform_im2col_cuda.cuh:151-178: h_im = h_in + i*dilation + offset_h (float arithmetic)
    const float h = (float)(ho * g.stride - g.pad + ky * g.dil) + off_h;
    const float w = (float)(wo * g.stride - g.pad + kx * g.dil) + off_w;
    s.inside = (h > -1.f) && (w > -1.f) && (h < (float)g.H) && (w < (float)g.W);
    const float hl = floorf(h), wl = floorf(w);
    const int h_low = (int)hl, w_low = (int)wl, h_high = h_low + 1, w_high = w_low + 1;
    s.lh = h - hl;
    s.lw = w - wl;
    const float hh = 1.f - s.lh, hw = 1.f - s.lw;
    s.w[0] = hh * hw; s.w[1] = hh * s.lw; s.w[2] = s.lh * hw; s.w[3] = s.lh * s.lw;
    const bool ok0 = s.inside && h_low >= 0 && w_low >= 0;
    const bool ok1 = s.inside && h_low >= 0 && w_high <= g.W - 1;
    const bool ok2 = s.inside && h_high <= g.H - 1 && w_low >= 0;
    const bool ok3 = s.inside && h_high <= g.H - 1 && w_high <= g.W - 1;
    const int base = b * g.H;
    s.idx[0] = ok0 ? (base + h_low) * g.W + w_low : -1;
    s.idx[1] = ok1 ? (base + h_low) * g.W + w_high : -1;
    s.idx[2] = ok2 ? (base + h_high) * g.W + w_low : -1;
    s.idx[3] = ok3 ? (base + h_high) * g.W + w_high : -1;
    return s;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ void k_dcn_prep(const float *__restrict__ offset, int off_stride, const float *__restrict__ mask, int mask_stride, int sig,
                           DcnGeom g, int n_rows, int taps, int32_t *samp_idx, float *samp_w) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * taps) return;
    const int j = i / taps, t = i % taps;
    const float oh = offset[(int64_t)j * off_stride + 2 * t], ow = offset[(int64_t)j * off_stride + 2 * t + 1];
    float m = mask[(int64_t)j * mask_stride + t];
    if (sig) m = sigmoidf_(m);
    Sample s = dcn_sample(g, j, t, oh, ow);
    int4 q = make_int4(s.idx[0], s.idx[1], s.idx[2], s.idx[3]);
    f32x4 w = {m * s.w[0], m * s.w[1], m * s.w[2], m * s.w[3]};
    *reinterpret_cast<int4 *>(samp_idx + (int64_t)i * 4) = q;
    *reinterpret_cast<f32x4 *>(samp_w + (int64_t)i * 4) = w;
}

static int check_geom(const DcnGeom &g, const char *who) {
    RD_REQUIRE(g.B > 0 && g.H > 0 && g.W > 0 && g.KH > 0 && g.KW > 0 && g.KH * g.KW <= 16 && g.stride > 0 && g.dil > 0 && g.pad >= 0,
               "%s: bad geometry", who);
    int Ho = (g.H + 2 * g.pad - (g.dil * (g.KH - 1) + 1)) / g.stride + 1;
    int Wo = (g.W + 2 * g.pad - (g.dil * (g.KW - 1) + 1)) / g.stride + 1;
    RD_REQUIRE(Ho == g.Ho && Wo == g.Wo, "%s: output size (%d,%d) does not match geometry (%d,%d)", who, g.Ho, g.Wo, Ho, Wo);
    return RD_OK;
}

extern "C" int rd_dcn_prep(const float *offset, int off_stride, const float *mask, int mask_stride, int apply_sigmoid, int B, int H, int W,
                           int Ho, int Wo, int KH, int KW, int stride, int pad, int dil, int32_t *samp_idx, float *samp_w, void *stream) {
    DcnGeom g{B, H, W, Ho, Wo, KH, KW, stride, pad, dil};
    int rc = check_geom(g, "rd_dcn_prep");
    if (rc) return rc;
    const int taps = KH * KW;
    RD_REQUIRE(off_stride >= 2 * taps && mask_stride >= taps, "rd_dcn_prep: row strides too small");
    const int n_rows = B * Ho * Wo;
    k_dcn_prep<<<cdiv((int64_t)n_rows * taps, 256), 256, 0, S(stream)>>>(offset, off_stride, mask, mask_stride, apply_sigmoid, g, n_rows, taps,
                                                                         samp_idx, samp_w);
    return check_launch("rd_dcn_prep");
}

// Deformed columns col[j][t][c] = sum over the 4 bilinear corners of samp_w[j][t][q] * x[samp_idx[j][t][q]][c] (the sampling table already
// carries the modulation mask): the DCNv2 convolution is then a plain GEMM of (rows x taps*C) columns with the [Cout][taps][C] weights,
// and its weight gradient a plain GEMM of the same columns with grad_out.  One wave per (pixel, tap) pair and 256-channel slab: every
// corner read and the write are contiguous 1 KiB rows (float4 per lane).  At the CMA shapes (8192 pixels, 256 channels) the columns are
// 75 MB -- written once, read by two GEMMs -- against a gathered kernel that re-blended four corner rows per (tile, tap, chunk) step and
// ran the same 9.7 GFLOP 3-5x slower than a plain convolution.
__global__ __launch_bounds__(256) void k_dcn_columns(const float *__restrict__ x, int C, const int32_t *__restrict__ samp_idx,
                                                     const float *__restrict__ samp_w, int64_t n_pairs, float *__restrict__ col) {
    const int lane = threadIdx.x & 63;
    const int64_t pair = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= n_pairs) return;
    const int4 q = *reinterpret_cast<const int4 *>(samp_idx + pair * 4);
    const f32x4 w = *reinterpret_cast<const f32x4 *>(samp_w + pair * 4);
    const int idx[4] = {q.x, q.y, q.z, q.w};
    for (int c = lane * 4; c < C; c += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (idx[k] >= 0) acc += w[k] * *reinterpret_cast<const f32x4 *>(x + (int64_t)idx[k] * C + c);
        *reinterpret_cast<f32x4 *>(col + pair * C + c) = acc;
    }
}

extern "C" int rd_dcn_columns(const float *x, int64_t in_rows, int C, const int32_t *samp_idx, const float *samp_w, int64_t out_rows, int taps,
                              float *col, void *stream) {
    RD_REQUIRE(x && samp_idx && samp_w && col, "rd_dcn_columns: null pointer");
    RD_REQUIRE(C > 0 && C % 4 == 0, "rd_dcn_columns: C=%d must be a positive multiple of 4", C);
    RD_REQUIRE(in_rows > 0 && out_rows > 0 && taps > 0 && taps <= 16, "rd_dcn_columns: bad sizes");
    const int64_t n_pairs = out_rows * taps;
    k_dcn_columns<<<(unsigned)cdiv(n_pairs, 4), 256, 0, S(stream)>>>(x, C, samp_idx, samp_w, n_pairs, col);
    return check_launch("rd_dcn_columns");
}

// One wave per (output pixel j, tap t); lanes sweep the C channels, lane l taking l, l + 64, ...
// ORDERED (rd_set_deterministic(1)): ONE workgroup walks all (pixel, tap) pairs in order; wave w owns the channels w*64 + lane
// (+ 64 * waves ...), so every grad_x element is updated by one lane in pair order with plain adds, and the per-pair sums over
// channels are combined across the waves in wave order through LDS.
template <bool ORDERED>
__global__ __launch_bounds__(256) void k_dcn_bwd_data(const float *__restrict__ x, int C, const float *__restrict__ colgrad,
                                                      const float *__restrict__ offset, int off_stride, const float *__restrict__ mask,
                                                      int mask_stride, int sig, DcnGeom g, int n_rows, int taps, float *grad_x,
                                                      float *grad_offset, int goff_stride, float *grad_mask, int gmask_stride) {
    __shared__ float s_part[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int64_t n_pairs = (int64_t)n_rows * taps;
    const int64_t first = ORDERED ? 0 : (int64_t)blockIdx.x * n_waves + wave;
    const int64_t last = ORDERED ? n_pairs : min(n_pairs, first + 1);
    for (int64_t pair = first; pair < last; ++pair) {
        const int j = (int)(pair / taps), t = (int)(pair % taps);
        const float oh = offset[(int64_t)j * off_stride + 2 * t], ow = offset[(int64_t)j * off_stride + 2 * t + 1];
        float mraw = mask[(int64_t)j * mask_stride + t];
        const float m = sig ? sigmoidf_(mraw) : mraw;
        const Sample s = dcn_sample(g, j, t, oh, ow);
        float dmask = 0.f, dh = 0.f, dw = 0.f;
        if (s.inside) {
            const float hh = 1.f - s.lh, hw = 1.f - s.lw;
            const float *gc = colgrad + ((int64_t)j * taps + t) * C;
            // lane l handles channels l, l + 64, ...: every atomic instruction of the wave covers 256 CONTIGUOUS bytes (the memory-side
            // float atomics run ~4x slower when a wave's addresses are strided, which the float4-per-lane layout of round 1 made them)
            for (int c = ORDERED ? wave * 64 + lane : lane; c < C; c += ORDERED ? 64 * n_waves : 64) {
                const float gv = gc[c];
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = s.idx[k] >= 0 ? x[(int64_t)s.idx[k] * C + c] : 0.f;
                const float val = s.w[0] * v[0] + s.w[1] * v[1] + s.w[2] * v[2] + s.w[3] * v[3];
                dmask += gv * val;
                // mdmcn_get_coordinate_weight (modulated_deform_im2col_cuda.cuh:84-125)
                dh += gv * (-hw * v[0] - s.lw * v[1] + hw * v[2] + s.lw * v[3]);
                dw += gv * (-hh * v[0] + hh * v[1] - s.lh * v[2] + s.lh * v[3]);
                // input gradient: grad_x[corner] += mask * corner weight * column gradient  (col2im, :196-254)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (s.idx[k] >= 0) {
                        if (ORDERED) grad_x[(int64_t)s.idx[k] * C + c] += m * s.w[k] * gv;
                        else atomicAdd(grad_x + (int64_t)s.idx[k] * C + c, m * s.w[k] * gv);
                    }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            dmask += __shfl_xor(dmask, d, 64);
            dh += __shfl_xor(dh, d, 64);
            dw += __shfl_xor(dw, d, 64);
        }
        if (ORDERED) {
            if (lane == 0) { s_part[wave][0] = dmask; s_part[wave][1] = dh; s_part[wave][2] = dw; }
            __syncthreads();
            dmask = dh = dw = 0.f;
            for (int w = 0; w < n_waves; ++w) { dmask += s_part[w][0]; dh += s_part[w][1]; dw += s_part[w][2]; }
            __syncthreads();
        }
        if (lane == 0 && (!ORDERED || wave == 0)) {
            grad_offset[(int64_t)j * goff_stride + 2 * t] = dh * m;
            grad_offset[(int64_t)j * goff_stride + 2 * t + 1] = dw * m;
            grad_mask[(int64_t)j * gmask_stride + t] = sig ? dmask * m * (1.f - m) : dmask;
        }
    }
}

extern "C" int rd_dcn_bwd_data(const float *x, int C, const float *colgrad, const float *offset, int off_stride, const float *mask,
                               int mask_stride, int apply_sigmoid, int B, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad,
                               int dil, float *grad_x, float *grad_offset, int goff_stride, float *grad_mask, int gmask_stride, void *stream) {
    DcnGeom g{B, H, W, Ho, Wo, KH, KW, stride, pad, dil};
    int rc = check_geom(g, "rd_dcn_bwd_data");
    if (rc) return rc;
    RD_REQUIRE(C > 0 && C % 4 == 0, "rd_dcn_bwd_data: C=%d must be a multiple of 4", C);
    const int taps = KH * KW;
    hipStream_t st = S(stream);
    RD_HIP(hipMemsetAsync(grad_x, 0, (size_t)B * H * W * C * 4, st));
    const int64_t pairs = (int64_t)B * Ho * Wo * taps;
    if (pairs == 0) return RD_OK;
    if (g_deterministic)
        k_dcn_bwd_data<true><<<1, 256, 0, st>>>(x, C, colgrad, offset, off_stride, mask, mask_stride, apply_sigmoid, g, B * Ho * Wo, taps,
                                                grad_x, grad_offset, goff_stride, grad_mask, gmask_stride);
    else
        k_dcn_bwd_data<false><<<cdiv(pairs, 4), 256, 0, st>>>(x, C, colgrad, offset, off_stride, mask, mask_stride, apply_sigmoid, g, B * Ho * Wo,
                                                              taps, grad_x, grad_offset, goff_stride, grad_mask, gmask_stride);
    return check_launch("rd_dcn_bwd_data");
}
