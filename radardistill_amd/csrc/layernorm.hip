// LayerNorm over the channel axis of channels-last rows (ConvNeXt block: pcdet/ops/basicblock/modules/Basicblock_convn.py:58-82,
// F.layer_norm(x, (C,), weight, bias, eps = 1e-6) on (B, H, W, C)), forward and backward.  See include/rdamd.h section R.
// One wavefront owns one row (C = 256: 64 lanes x float4 = one 1 KiB row), mean / variance / the two backward sums by wavefront
// shuffles; per-row (mean, rstd) are kept for the backward.  HBM-bound: forward reads x and writes y once; backward reads x and
// grad_y and writes grad_x once, the parameter gradients (column sums over all rows) are per-lane partials -> LDS -> one fp32
// atomic per channel and workgroup (rd_set_deterministic(1): one workgroup).  Replaces ATen's native_layer_norm (+ its two
// backward kernels and the casts around them): 1 + 1 launches per block and pass instead of ~5.
#include <algorithm>
#include "common.hpp"

using namespace rd;

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int LN_MAXV = 4;      // float4 per lane: C <= 1024

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_ln_fwd(const float *__restrict__ x, int64_t rows, int C, const float *__restrict__ gamma,
                                                const float *__restrict__ beta, float eps, float *__restrict__ y, float *__restrict__ mean_out,
                                                float *__restrict__ rstd_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = (C / 4 + 63) / 64;
    const float inv_c = 1.f / (float)C;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        f32x4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int c = (lane + 64 * k) * 4;
            v[k] = (k < nv && c < C) ? *reinterpret_cast<const f32x4 *>(x + r * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += v[k][0] + v[k][1] + v[k][2] + v[k][3];
        }
        const float mu = wsum(s) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (k < nv && c < C)
#pragma unroll
                for (int e = 0; e < 4; ++e) q += (v[k][e] - mu) * (v[k][e] - mu);
        }
        const float rstd = rsqrtf(wsum(q) * inv_c + eps);
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (k < nv && c < C) {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + c), b = *reinterpret_cast<const f32x4 *>(beta + c);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[k][e] - mu) * rstd * g[e] + b[e];
                *reinterpret_cast<f32x4 *>(y + r * C + c) = o;
            }
        }
        if (lane == 0) {
            mean_out[r] = mu;
            rstd_out[r] = rstd;
        }
    }
}

// dx = rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat));  dgamma_c = sum_r g*xhat,  dbeta_c = sum_r g
__global__ __launch_bounds__(256) void k_ln_bwd(const float *__restrict__ x, const float *__restrict__ gy, int64_t rows, int C,
                                                const float *__restrict__ gamma, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                float *__restrict__ gx, float *dgamma, float *dbeta) {
    extern __shared__ float red[];        // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = (C / 4 + 63) / 64;
    const float inv_c = 1.f / (float)C;
    f32x4 ag[LN_MAXV], ab[LN_MAXV], gm[LN_MAXV];
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = (lane + 64 * k) * 4;
        ag[k] = ab[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        gm[k] = (k < nv && c < C) ? *reinterpret_cast<const f32x4 *>(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        const float mu = mean[r], rs = rstd[r];
        f32x4 xh[LN_MAXV], g[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int c = (lane + 64 * k) * 4;
            const bool ok = k < nv && c < C;
            const f32x4 xv = ok ? *reinterpret_cast<const f32x4 *>(x + r * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            g[k] = ok ? *reinterpret_cast<const f32x4 *>(gy + r * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[k][e] = ok ? (xv[e] - mu) * rs : 0.f;
                const float gg = g[k][e] * gm[k][e];
                s1 += gg;
                s2 += gg * xh[k][e];
                ag[k][e] += g[k][e] * xh[k][e];
                ab[k][e] += g[k][e];
            }
        }
        const float m1 = wsum(s1) * inv_c, m2 = wsum(s2) * inv_c;
#pragma unroll
        for (int k = 0; k < LN_MAXV; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (k < nv && c < C) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[k][e] * gm[k][e] - m1 - xh[k][e] * m2);
                *reinterpret_cast<f32x4 *>(gx + r * C + c) = o;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
        const int c = (lane + 64 * k) * 4;
        if (k < nv && c < C) {
            *reinterpret_cast<f32x4 *>(red + (wave * 2 + 0) * C + c) = ag[k];
            *reinterpret_cast<f32x4 *>(red + (wave * 2 + 1) * C + c) = ab[k];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i % C;
        const float v = red[(0 * 2 + which) * C + c] + red[(1 * 2 + which) * C + c] + red[(2 * 2 + which) * C + c] + red[(3 * 2 + which) * C + c];
        atomicAdd(which ? &dbeta[c] : &dgamma[c], v);
    }
}

}  // namespace

extern "C" int rd_layernorm_fwd(const float *x, int64_t rows, int C, const float *gamma, const float *beta, float eps, float *y, float *mean,
                                float *rstd, void *stream) {
    RD_REQUIRE(C > 0 && C % 4 == 0 && C <= 256 * LN_MAXV, "rd_layernorm_fwd: C=%d must be a multiple of 4, <= %d", C, 256 * LN_MAXV);
    if (rows <= 0) return RD_OK;
    const int blocks = (int)std::min<int64_t>(cdiv(rows, 4), 4096);
    k_ln_fwd<<<blocks, 256, 0, S(stream)>>>(x, rows, C, gamma, beta, eps, y, mean, rstd);
    return check_launch("rd_layernorm_fwd");
}

// grad_gamma / grad_beta (C each) are zero-filled here and accumulated
extern "C" int rd_layernorm_bwd(const float *x, const float *grad_y, int64_t rows, int C, const float *gamma, const float *mean, const float *rstd,
                                float *grad_x, float *grad_gamma, float *grad_beta, void *stream) {
    RD_REQUIRE(C > 0 && C % 4 == 0 && C <= 256 * LN_MAXV, "rd_layernorm_bwd: C=%d must be a multiple of 4, <= %d", C, 256 * LN_MAXV);
    hipStream_t st = S(stream);
    if (grad_beta == grad_gamma + C) {
        RD_HIP(hipMemsetAsync(grad_gamma, 0, (size_t)2 * C * 4, st));          // one [dgamma | dbeta] buffer: one fill
    } else {
        RD_HIP(hipMemsetAsync(grad_gamma, 0, (size_t)C * 4, st));
        RD_HIP(hipMemsetAsync(grad_beta, 0, (size_t)C * 4, st));
    }
    if (rows <= 0) return RD_OK;
    const int blocks = g_deterministic ? 1 : (int)std::min<int64_t>(cdiv(rows, 4 * 8), 512);
    k_ln_bwd<<<blocks, 256, (size_t)8 * C * 4, st>>>(x, grad_y, rows, C, gamma, mean, rstd, grad_x, grad_gamma, grad_beta);
    return check_launch("rd_layernorm_bwd");
}
