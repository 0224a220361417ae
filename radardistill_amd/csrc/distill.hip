// AFD / PFD distillation losses over channels-last BEV feature maps (rows = BEV cells, 256 channels).
// Replaces ~45 ATen elementwise / reduce kernels of Radar_Distill.low_loss / high_loss
// (pcdet/models/backbones_2d/radar_distill_final.py:82-141).  HBM-bound: every map is read exactly once per pass;
// one wave owns one BEV cell (a 1 KiB row: 64 lanes x float4), row reductions by wave shuffles, per-block partials and a
// fixed-order final sum (deterministic).
#include "common.hpp"

using namespace rd;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

// ---------------------------------------------------------------------------------------------- AFD
// For each of the two radar maps (CMA output `8x_2` and intermediate `8x_1`) against the same lidar map:
//   lm = [sum_c lidar > 0], rs = sum_c radar, act = [rs > 0] + 0.5 lm
//   S_ar = sum over cells with act == 1.5 of sum_c (r - l)^2,  S_ir likewise for act == 1.0, N_ar / N_ir their counts,
//   M = sum |sigmoid(rs) - lm|.
// partial layout per block: [2 maps][5] = S_ar, S_ir, N_ar, N_ir, M.  rowinfo[2][rows] keeps (class, rs) for the backward.
constexpr int AFD_Q = 5;

// T = float, or __bf16 for maps stored in bf16 (BASELINE configs[2]: half the bytes of this HBM-bound pass; sums stay fp32)
template <typename T>
__device__ __forceinline__ f32x4 load4(const T *p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
template <>
__device__ __forceinline__ f32x4 load4<__bf16>(const __bf16 *p) {
    const uint2 u = *reinterpret_cast<const uint2 *>(p);
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}

template <typename T>
__global__ __launch_bounds__(256) void k_afd_fwd(const T *__restrict__ lidar, const T *__restrict__ ra, const T *__restrict__ rb,
                                                 int64_t rows, int C, float *partial, float *rowinfo) {
    __shared__ float red[4][2 * AFD_Q];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc[2 * AFD_Q];
#pragma unroll
    for (int i = 0; i < 2 * AFD_Q; ++i) acc[i] = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wid; r < rows; r += (int64_t)gridDim.x * 4) {
        float ls = 0.f, s[2] = {0.f, 0.f}, mse[2] = {0.f, 0.f};
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 l = load4<T>(lidar + r * C + c);
            const f32x4 a = load4<T>(ra + r * C + c);
            const f32x4 b = load4<T>(rb + r * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ls += l[e];
                s[0] += a[e];
                s[1] += b[e];
                mse[0] += (a[e] - l[e]) * (a[e] - l[e]);
                mse[1] += (b[e] - l[e]) * (b[e] - l[e]);
            }
        }
        ls = wave_sum(ls);
        const float lm = ls > 0.f ? 1.f : 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float rs = wave_sum(s[m]);
            const float ms = wave_sum(mse[m]);
            const bool ron = rs > 0.f;
            const int cls = ron ? (lm > 0.f ? 1 : 2) : 0;       // 1: active radar & active lidar (1.5), 2: active radar only (1.0)
            if (lane == 0) {
                if (cls == 1) { acc[m * AFD_Q + 0] += ms; acc[m * AFD_Q + 2] += 1.f; }
                if (cls == 2) { acc[m * AFD_Q + 1] += ms; acc[m * AFD_Q + 3] += 1.f; }
                acc[m * AFD_Q + 4] += fabsf(sigm(rs) - lm);
                rowinfo[((int64_t)m * rows + r) * 2 + 0] = (float)cls + 4.f * lm;   // class + lidar mask packed
                rowinfo[((int64_t)m * rows + r) * 2 + 1] = rs;
            }
        }
    }
    if (lane == 0)
        for (int i = 0; i < 2 * AFD_Q; ++i) red[wid][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < 2 * AFD_Q) {
        float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        partial[(int64_t)blockIdx.x * 2 * AFD_Q + threadIdx.x] = v;
    }
}

// sums[2][5] (double accumulate) then out[2][2] = (feature_loss, mask_loss) per map, and coef[2][3] for the backward:
//   feature = 3e-4 * S_ar / B + 5e-5 * (N_ar / N_ir) * S_ir / B   (NaN when N_ir == 0 and ... exactly as the reference: 0 * inf)
// One wavefront: lane l adds the partials of blocks l, l + 64, ... in double, a butterfly combines the lanes (a fixed order, so the
// result is deterministic); lanes 0 / 1 then finish map 0 / 1.  (Round 1 had two lanes walk all 1024 x 5 partials serially: 154 us.)
__global__ __launch_bounds__(64) void k_afd_final(const float *partial, int n_blocks, float inv_B, float inv_cells, float *out, float *coef) {
    double acc[2 * AFD_Q];
#pragma unroll
    for (int i = 0; i < 2 * AFD_Q; ++i) acc[i] = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += 64)
#pragma unroll
        for (int i = 0; i < 2 * AFD_Q; ++i) acc[i] += (double)partial[(int64_t)b * 2 * AFD_Q + i];
#pragma unroll
    for (int i = 0; i < 2 * AFD_Q; ++i)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc[i] += __shfl_xor(acc[i], d, 64);
    const int m = threadIdx.x;
    if (m >= 2) return;
    double s[AFD_Q];
#pragma unroll
    for (int q = 0; q < AFD_Q; ++q) s[q] = m == 0 ? acc[q] : acc[AFD_Q + q];
    const float S_ar = (float)s[0], S_ir = (float)s[1], N_ar = (float)s[2], N_ir = (float)s[3];
    // reference: mask_ir *= N_ar / N_ir (elementwise, so cells with mask 0 give 0 * ratio: NaN if ratio is inf or NaN);
    // loss_ir = sum(mse * mask_ir) / B
    const float ratio = N_ar / N_ir;
    float l_ir;
    if (isfinite(ratio)) l_ir = ratio * S_ir * inv_B;
    else l_ir = nanf("");                         // 0 * inf (or 0/0) appears in at least one cell -> the sum is NaN
    const float l_ar = S_ar * inv_B;
    out[m * 2 + 0] = 3e-4f * l_ar + 5e-5f * l_ir;
    out[m * 2 + 1] = (float)s[4] * inv_cells;
    coef[m * 3 + 0] = 3e-4f * inv_B;              // d feature / d mse_row on class-1 rows
    coef[m * 3 + 1] = 5e-5f * inv_B * ratio;      // ... on class-2 rows
    coef[m * 3 + 2] = inv_cells;                  // d mask_loss / d |.|
}

extern "C" int64_t rd_afd_ws_bytes(int64_t rows) { return (1024 * 2 * AFD_Q + 16) * 4 + rows * 2 * 2 * 4; }

extern "C" int rd_afd_fwd(const float *lidar, const float *radar_a, const float *radar_b, int64_t rows, int C, int batch,
                          float *out /*[4]*/, float *coef /*[6]*/, float *rowinfo /*[2][rows][2]*/, float *ws, int64_t ws_bytes, void *stream) {
    RD_REQUIRE(C % 4 == 0 && rows > 0 && batch > 0, "rd_afd_fwd: bad sizes");
    RD_REQUIRE(ws_bytes >= 1024 * 2 * AFD_Q * 4, "rd_afd_fwd: workspace too small");
    hipStream_t st = S(stream);
    int blocks = (int)std::min<int64_t>(1024, cdiv(rows, 4));
    k_afd_fwd<float><<<blocks, 256, 0, st>>>(lidar, radar_a, radar_b, rows, C, ws, rowinfo);
    k_afd_final<<<1, 64, 0, st>>>(ws, blocks, 1.0f / batch, 1.0f / (float)rows, out, coef);
    return check_launch("rd_afd_fwd");
}

// Same forward on maps STORED in bf16 (rd_lp_cast dtype 0 / the bf16 outputs of rd_lp_conv): BASELINE configs[2].
extern "C" int rd_afd_fwd_bf16(const void *lidar, const void *radar_a, const void *radar_b, int64_t rows, int C, int batch, float *out /*[4]*/,
                               float *coef /*[6]*/, float *rowinfo /*[2][rows][2]*/, float *ws, int64_t ws_bytes, void *stream) {
    RD_REQUIRE(C % 4 == 0 && rows > 0 && batch > 0, "rd_afd_fwd_bf16: bad sizes");
    RD_REQUIRE(ws_bytes >= 1024 * 2 * AFD_Q * 4, "rd_afd_fwd_bf16: workspace too small");
    hipStream_t st = S(stream);
    int blocks = (int)std::min<int64_t>(1024, cdiv(rows, 4));
    k_afd_fwd<__bf16><<<blocks, 256, 0, st>>>(reinterpret_cast<const __bf16 *>(lidar), reinterpret_cast<const __bf16 *>(radar_a),
                                             reinterpret_cast<const __bf16 *>(radar_b), rows, C, ws, rowinfo);
    k_afd_final<<<1, 64, 0, st>>>(ws, blocks, 1.0f / batch, 1.0f / (float)rows, out, coef);
    return check_launch("rd_afd_fwd_bf16");
}

// grad_radar_m[r][c] = g_feat[m] * coef_cls * 2 (r - l) + g_mask[m] * inv_cells * sign(sigmoid(rs) - lm) * sigmoid'(rs)
// gscale[4] = upstream gradients (feature_a, mask_a, feature_b, mask_b) on the device.
__global__ __launch_bounds__(256) void k_afd_bwd(const float *__restrict__ lidar, const float *__restrict__ ra, const float *__restrict__ rb,
                                                 int64_t rows, int C, const float *__restrict__ rowinfo, const float *__restrict__ coef,
                                                 const float *__restrict__ gscale, float *ga, float *gb) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wid; r < rows; r += (int64_t)gridDim.x * 4) {
        float k2[2], add[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float packed = rowinfo[((int64_t)m * rows + r) * 2 + 0], rs = rowinfo[((int64_t)m * rows + r) * 2 + 1];
            const int lmi = packed >= 4.f ? 1 : 0;
            const int cls = (int)(packed - 4.f * lmi);
            const float cf = cls == 1 ? coef[m * 3 + 0] : (cls == 2 ? coef[m * 3 + 1] : 0.f);
            k2[m] = 2.f * cf * gscale[m * 2 + 0];
            const float sg = sigm(rs), d = sg - (float)lmi;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            add[m] = gscale[m * 2 + 1] * coef[m * 3 + 2] * sgn * sg * (1.f - sg);
        }
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 l = *reinterpret_cast<const f32x4 *>(lidar + r * C + c);
            const f32x4 a = *reinterpret_cast<const f32x4 *>(ra + r * C + c);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(rb + r * C + c);
            f32x4 oa, ob;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                oa[e] = k2[0] * (a[e] - l[e]) + add[0];
                ob[e] = k2[1] * (b[e] - l[e]) + add[1];
            }
            *reinterpret_cast<f32x4 *>(ga + r * C + c) = oa;
            *reinterpret_cast<f32x4 *>(gb + r * C + c) = ob;
        }
    }
}

extern "C" int rd_afd_bwd(const float *lidar, const float *radar_a, const float *radar_b, int64_t rows, int C, const float *rowinfo,
                          const float *coef, const float *gscale, float *grad_a, float *grad_b, void *stream) {
    RD_REQUIRE(C % 4 == 0 && rows > 0, "rd_afd_bwd: bad sizes");
    int blocks = (int)std::min<int64_t>(2048, cdiv(rows, 4));
    k_afd_bwd<<<blocks, 256, 0, S(stream)>>>(lidar, radar_a, radar_b, rows, C, rowinfo, coef, gscale, grad_a, grad_b);
    return check_launch("rd_afd_bwd");
}

// ---------------------------------------------------------------------------------------------- PFD
// Cell weights from the ground-truth and predicted heat-maps (radar_distill_final.py:111-127):
//   gt = max_c gt_hm, pr = max_c clamp(sigmoid(logit), 1e-4, 1 - 1e-4)
//   class 1: gt > .1 (TP or FN: pr > .1 or pr < .1), class 2: gt < .1 and pr > .1 (FP), else 0
//   w = 5 / count(class 1) on class 1, 1 / count(class 2) on class 2.
// counts[2] int32 (zeroed inside).
__global__ void k_pfd_class(const float *__restrict__ gt_hm, const float *__restrict__ logits, int64_t rows, int nc, int8_t *cls, int *counts) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int c = 0;
    if (r < rows) {
        float g = -INFINITY, p = -INFINITY;
        for (int k = 0; k < nc; ++k) {
            g = fmaxf(g, gt_hm[r * nc + k]);
            float s = fminf(fmaxf(sigm(logits[r * nc + k]), 1e-4f), 1.f - 1e-4f);
            p = fmaxf(p, s);
        }
        const bool tp = g > 0.1f && p > 0.1f, fn = g > 0.1f && p < 0.1f, fp = g < 0.1f && p > 0.1f;
        c = (tp || fn) ? 1 : (fp ? 2 : 0);
        cls[r] = (int8_t)c;
    }
    // wave-aggregated counting
    unsigned long long b1 = __ballot(c == 1), b2 = __ballot(c == 2);
    if ((threadIdx.x & 63) == 0) {
        if (b1) atomicAdd(&counts[0], __popcll(b1));
        if (b2) atomicAdd(&counts[1], __popcll(b2));
    }
}

__device__ __forceinline__ float pfd_weight(int c, const int *counts) {
    return c == 1 ? 5.f / (float)counts[0] : (c == 2 ? 1.f / (float)counts[1] : 0.f);
}

// loss = 0.5 * sum_cells w * ( sum_c |softmax(r1) - softmax(l1)| + sum_c |softmax(r2) - softmax(l2)| )
// MODE 0: forward (per-block partial sums).  MODE 1: backward -> grad r1, grad r2 (times *gscale).
template <int MODE>
__global__ __launch_bounds__(256) void k_pfd(const float *__restrict__ r1, const float *__restrict__ l1, const float *__restrict__ r2,
                                             const float *__restrict__ l2, int64_t rows, int C, const int8_t *__restrict__ cls,
                                             const int *__restrict__ counts, float *partial, const float *gscale, float *g1, float *g2) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float acc = 0.f;
    const float gs = MODE == 1 ? 0.5f * gscale[0] : 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wid; r < rows; r += (int64_t)gridDim.x * 4) {
        const float w = pfd_weight(cls[r], counts);
        if (w == 0.f) {
            if (MODE == 1)
                for (int c = lane * 4; c < C; c += 256) {
                    *reinterpret_cast<f32x4 *>(g1 + r * C + c) = f32x4{0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4 *>(g2 + r * C + c) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            continue;
        }
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
            const float *rp = (pair == 0 ? r1 : r2) + r * C, *lp = (pair == 0 ? l1 : l2) + r * C;
            float *gp = MODE == 1 ? (pair == 0 ? g1 : g2) + r * C : nullptr;
            // C <= 1024: each lane holds up to 4 float4 of the row (static indexing keeps them in registers)
            f32x4 rv[4], lv[4];
            float rmax = -INFINITY, lmax = -INFINITY;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = lane * 4 + q * 256;
                if (c < C) {
                    rv[q] = *reinterpret_cast<const f32x4 *>(rp + c);
                    lv[q] = *reinterpret_cast<const f32x4 *>(lp + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rmax = fmaxf(rmax, rv[q][e]); lmax = fmaxf(lmax, lv[q][e]); }
                }
            }
            rmax = wave_max(rmax); lmax = wave_max(lmax);
            float rsum = 0.f, lsum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (lane * 4 + q * 256 < C) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        rv[q][e] = __expf(rv[q][e] - rmax); rsum += rv[q][e];
                        lv[q][e] = __expf(lv[q][e] - lmax); lsum += lv[q][e];
                    }
                }
            rsum = wave_sum(rsum); lsum = wave_sum(lsum);
            const float rinv = 1.f / rsum, linv = 1.f / lsum;
            float l1sum = 0.f, dot = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (lane * 4 + q * 256 < C) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float p = rv[q][e] * rinv, t = lv[q][e] * linv;
                        const float d = p - t;
                        l1sum += fabsf(d);
                        if (MODE == 1) {
                            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                            rv[q][e] = p;       // keep p
                            lv[q][e] = sgn;     // keep sign
                            dot += p * sgn;
                        }
                    }
                }
            if (MODE == 0) {
                acc += w * wave_sum(l1sum);
            } else {
                dot = wave_sum(dot);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = lane * 4 + q * 256;
                    if (c < C) {
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = gs * w * rv[q][e] * (lv[q][e] - dot);   // softmax backward of w*|p - t|
                        *reinterpret_cast<f32x4 *>(gp + c) = o;
                    }
                }
            }
        }
    }
    if (MODE == 0) {
        if (lane == 0) red[wid] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

__global__ __launch_bounds__(64) void k_sum_partials(const float *partial, int n, float scale, float *out) {
    double s = 0.0;                                        // one wavefront, strided adds + butterfly: fixed order
    for (int i = threadIdx.x; i < n; i += 64) s += (double)partial[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

extern "C" int rd_pfd_fwd(const float *r1, const float *l1, const float *r2, const float *l2, int64_t rows, int C, const float *gt_hm,
                          const float *hm_logits, int n_hm, int8_t *cls, int32_t *counts, float *out, float *ws, int64_t ws_bytes, void *stream) {
    RD_REQUIRE(C % 4 == 0 && C <= 1024 && rows > 0 && n_hm > 0, "rd_pfd_fwd: bad sizes");
    RD_REQUIRE(ws_bytes >= 1024 * 4, "rd_pfd_fwd: workspace too small");
    hipStream_t st = S(stream);
    RD_HIP(hipMemsetAsync(counts, 0, 8, st));
    k_pfd_class<<<cdiv(rows, 256), 256, 0, st>>>(gt_hm, hm_logits, rows, n_hm, cls, counts);
    int blocks = (int)std::min<int64_t>(1024, cdiv(rows, 4));
    k_pfd<0><<<blocks, 256, 0, st>>>(r1, l1, r2, l2, rows, C, cls, counts, ws, nullptr, nullptr, nullptr);
    k_sum_partials<<<1, 64, 0, st>>>(ws, blocks, 0.5f, out);
    return check_launch("rd_pfd_fwd");
}

extern "C" int rd_pfd_bwd(const float *r1, const float *l1, const float *r2, const float *l2, int64_t rows, int C, const int8_t *cls,
                          const int32_t *counts, const float *gscale, float *g1, float *g2, void *stream) {
    RD_REQUIRE(C % 4 == 0 && C <= 1024 && rows > 0, "rd_pfd_bwd: bad sizes");
    int blocks = (int)std::min<int64_t>(2048, cdiv(rows, 4));
    k_pfd<1><<<blocks, 256, 0, S(stream)>>>(r1, l1, r2, l2, rows, C, cls, counts, nullptr, gscale, g1, g2);
    return check_launch("rd_pfd_bwd");
}
