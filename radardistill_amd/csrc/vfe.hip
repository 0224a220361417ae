// Pillar feature encoder (PillarVFE) kernels: per-pillar mean, point-wise Linear(9+C -> 32), BatchNorm statistics,
// fused affine + ReLU + per-pillar max, and the student's backward.  See include/rdamd.h section B.
// Reference: pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:14-46 (PFNLayerV2), :214-241 (feature assembly).
//
// Thread mapping: one lane per (point, output channel); a wave covers 2 points x 32 channels, so the 7 point words
// are fetched once per half-wave (same-address loads broadcast) and the 32x(9+C) weight matrix sits in LDS.
// The per-pillar max uses one 64-bit atomicMax per (pillar, channel) on {value bits, ~point index}: post-ReLU values
// are >= 0 so their IEEE bit patterns order like unsigned integers, and the packed index makes the arg-max
// deterministic (smallest point index wins ties).  HBM-bound: points are read once per pass, pillars written once.
#include "common.hpp"

using namespace rd;

#define VFE_DISPATCH(NF, ...)                                   \
    switch (NF) {                                              \
        case 3: { constexpr int NFC = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int NFC = 4; __VA_ARGS__; } break; \
        case 5: { constexpr int NFC = 5; __VA_ARGS__; } break; \
        case 6: { constexpr int NFC = 6; __VA_ARGS__; } break; \
        case 7: { constexpr int NFC = 7; __VA_ARGS__; } break; \
        default: rd::set_error("unsupported n_feat %d (3..7)", NF); return RD_EINVAL; \
    }

constexpr int VFE_OUT = 32;
constexpr int VFE_MAX_IN = 16;  // 9 + n_feat <= 16  (LiDAR 14, radar 15)

__global__ void k_pillar_acc(const float *__restrict__ points, int n, int stride, const int32_t *__restrict__ point_row, float *acc) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int r = point_row[i];
    if (r < 0) return;
    const float *p = points + (int64_t)i * stride;
    atomicAdd(&acc[r * 4 + 0], p[1]);
    atomicAdd(&acc[r * 4 + 1], p[2]);
    atomicAdd(&acc[r * 4 + 2], p[3]);
    atomicAdd(&acc[r * 4 + 3], 1.0f);
}

// rd_set_deterministic(1): thread t owns the pillars with row % 256 == t and adds their points in point order (no atomics).
__global__ __launch_bounds__(256) void k_pillar_acc_ordered(const float *__restrict__ points, int n, int stride, const int32_t *__restrict__ point_row,
                                                            float *acc) {
    for (int i = 0; i < n; ++i) {
        const int r = point_row[i];
        if (r < 0 || (r & 255) != (int)threadIdx.x) continue;
        const float *p = points + (int64_t)i * stride;
        acc[r * 4 + 0] += p[1];
        acc[r * 4 + 1] += p[2];
        acc[r * 4 + 2] += p[3];
        acc[r * 4 + 3] += 1.0f;
    }
}

extern "C" int rd_vfe_pillar_mean(const float *points, int n_points, int n_feat, const int32_t *point_row, int n_pillars, float *pillar_acc, void *stream) {
    hipStream_t st = S(stream);
    if (n_pillars > 0) RD_HIP(hipMemsetAsync(pillar_acc, 0, (size_t)n_pillars * 16, st));
    if (n_points > 0 && g_deterministic) k_pillar_acc_ordered<<<1, 256, 0, st>>>(points, n_points, 1 + n_feat, point_row, pillar_acc);
    else if (n_points > 0) k_pillar_acc<<<cdiv(n_points, 256), 256, 0, st>>>(points, n_points, 1 + n_feat, point_row, pillar_acc);
    return check_launch("rd_vfe_pillar_mean");
}

// Assemble feature k of point i (dynamic_pillar_vfe.py:214-237):
//   [f_center(3), raw (x,y,z,feat...)(C), f_cluster(3), f_relative(3)]
struct Geom { float vx, vy, vz, xoff, yoff, zoff, x0, y0, z0; };

template <int n_feat>
__device__ __forceinline__ void point_feature(const float *p, const int32_t *coord /*(b,y,x)*/, const float *acc, const Geom &g, float *f) {
    float x = p[1], y = p[2], z = p[3];
    // f_center: x - (cx * vx + x_offset); pillar_coords store (b, y=cy, x=cx)
    f[0] = x - ((float)coord[2] * g.vx + g.xoff);
    f[1] = y - ((float)coord[1] * g.vy + g.yoff);
    f[2] = z - g.zoff;
#pragma unroll
    for (int k = 0; k < n_feat; ++k) f[3 + k] = p[1 + k];
    float cnt = fmaxf(acc[3], 1.0f);
    f[3 + n_feat + 0] = x - acc[0] / cnt;
    f[3 + n_feat + 1] = y - acc[1] / cnt;
    f[3 + n_feat + 2] = z - acc[2] / cnt;
    f[6 + n_feat + 0] = x - g.x0;
    f[6 + n_feat + 1] = y - g.y0;
    f[6 + n_feat + 2] = z - g.z0;
}

template <int n_feat>
__device__ __forceinline__ float linear_out(const float *p, const int32_t *coords, const float *pillar_acc, int row, const Geom &g,
                                            const float *w_lds /*[32][VFE_MAX_IN]*/, int c) {
    float f[VFE_MAX_IN];
    point_feature<n_feat>(p, coords + (int64_t)row * 3, pillar_acc + (int64_t)row * 4, g, f);
    constexpr int cin = 9 + n_feat;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < cin; ++k) s = fmaf(f[k], w_lds[c * VFE_MAX_IN + k], s);
    return s;
}

__device__ __forceinline__ void load_w(const float *weight, int cin, float *w_lds) {
    for (int i = threadIdx.x; i < VFE_OUT * VFE_MAX_IN; i += blockDim.x) {
        int c = i / VFE_MAX_IN, k = i % VFE_MAX_IN;
        w_lds[i] = k < cin ? weight[c * cin + k] : 0.f;
    }
}

__device__ __forceinline__ Geom load_geom(const float *g) { return Geom{g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8]}; }

// block = 256 threads = 8 points x 32 channels per pass, grid-stride over points; per-block partial sums -> atomics.
template <int n_feat>
__global__ __launch_bounds__(256) void k_vfe_stats(const float *__restrict__ points, int n, const int32_t *__restrict__ point_row,
                                                   const int32_t *__restrict__ coords, const float *__restrict__ pillar_acc,
                                                   const float *__restrict__ weight, const float *__restrict__ geom, float *stats) {
    __shared__ float w_lds[VFE_OUT * VFE_MAX_IN];
    __shared__ float red[2][8][VFE_OUT];
    load_w(weight, 9 + n_feat, w_lds);
    __syncthreads();
    const Geom g = load_geom(geom);
    const int c = threadIdx.x & 31, sub = threadIdx.x >> 5;
    float s = 0.f, ss = 0.f, cnt = 0.f;
    for (int i = blockIdx.x * 8 + sub; i < n; i += gridDim.x * 8) {
        int row = point_row[i];
        if (row < 0) continue;
        float v = linear_out<n_feat>(points + (int64_t)i * (1 + n_feat), coords, pillar_acc, row, g, w_lds, c);
        s += v;
        ss += v * v;
        cnt += 1.f;
    }
    red[0][sub][c] = s;
    red[1][sub][c] = ss;
    __syncthreads();
    if (sub == 0) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < 8; ++k) { a += red[0][k][c]; b += red[1][k][c]; }
        atomicAdd(&stats[c], a);
        atomicAdd(&stats[32 + c], b);
    }
    // valid-point count: only channel 0 lanes contribute
    if (c == 0) atomicAdd(&stats[64], cnt);
}

extern "C" int rd_vfe_linear_stats(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords,
                                   const float *pillar_acc, const float *weight, const float *geom, float *stats, void *stream) {
    RD_REQUIRE(9 + n_feat <= VFE_MAX_IN, "rd_vfe_linear_stats: 9 + n_feat = %d exceeds %d", 9 + n_feat, VFE_MAX_IN);
    hipStream_t st = S(stream);
    RD_HIP(hipMemsetAsync(stats, 0, 65 * 4, st));
    if (n_points > 0) {
        int blocks = g_deterministic ? 1 : (int)std::min<int64_t>(cdiv(n_points, 8), 1024);
        VFE_DISPATCH(n_feat, k_vfe_stats<NFC><<<blocks, 256, 0, st>>>(points, n_points, point_row, coords, pillar_acc, weight, geom, stats));
    }
    return check_launch("rd_vfe_linear_stats");
}

template <int n_feat>
__global__ __launch_bounds__(256) void k_vfe_max(const float *__restrict__ points, int n, const int32_t *__restrict__ point_row,
                                                 const int32_t *__restrict__ coords, const float *__restrict__ pillar_acc,
                                                 const float *__restrict__ weight, const float *__restrict__ geom,
                                                 const float *__restrict__ scale, const float *__restrict__ shift,
                                                 unsigned long long *packed) {
    __shared__ float w_lds[VFE_OUT * VFE_MAX_IN];
    load_w(weight, 9 + n_feat, w_lds);
    __syncthreads();
    const Geom g = load_geom(geom);
    const int c = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const float sc = scale[c], sh = shift[c];
    for (int i = blockIdx.x * 8 + sub; i < n; i += gridDim.x * 8) {
        int row = point_row[i];
        if (row < 0) continue;
        float v = linear_out<n_feat>(points + (int64_t)i * (1 + n_feat), coords, pillar_acc, row, g, w_lds, c);
        v = fmaxf(fmaf(v, sc, sh), 0.f);
        unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned int)(~(unsigned int)i);
        atomicMax(&packed[(int64_t)row * VFE_OUT + c], key);
    }
}

__global__ void k_vfe_unpack(const unsigned long long *packed, int64_t n, float *out, int32_t *argmax) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long k = packed[i];
    out[i] = __uint_as_float((unsigned int)(k >> 32));
    if (argmax) argmax[i] = (int32_t)(~(unsigned int)(k & 0xffffffffull));
}

extern "C" int rd_vfe_linear_bn_relu_max(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords,
                                         const float *pillar_acc, const float *weight, const float *geom, const float *scale,
                                         const float *shift, int n_pillars, float *out, int32_t *argmax, unsigned long long *ws_packed, void *stream) {
    RD_REQUIRE(9 + n_feat <= VFE_MAX_IN, "rd_vfe_linear_bn_relu_max: 9 + n_feat = %d exceeds %d", 9 + n_feat, VFE_MAX_IN);
    hipStream_t st = S(stream);
    if (n_pillars <= 0) return RD_OK;
    // key 0 == value +0.0 with index ~0: every pillar has >= 1 point so it is always overwritten
    RD_HIP(hipMemsetAsync(ws_packed, 0, (size_t)n_pillars * VFE_OUT * 8, st));
    if (n_points > 0) {
        int blocks = (int)std::min<int64_t>(cdiv(n_points, 8), 4096);
        VFE_DISPATCH(n_feat, k_vfe_max<NFC><<<blocks, 256, 0, st>>>(points, n_points, point_row, coords, pillar_acc, weight, geom, scale, shift, ws_packed));
    }
    int64_t tot = (int64_t)n_pillars * VFE_OUT;
    k_vfe_unpack<<<cdiv(tot, 256), 256, 0, st>>>(ws_packed, tot, out, argmax);
    return check_launch("rd_vfe_linear_bn_relu_max");
}

// ---------------------------------------------------------------------------------------------- backward (student radar VFE)
// x = W f, xh = (x - mean) * rstd, z = gamma*xh + beta, a = relu(z), out[p][c] = max_i a[i][c].
//   dz[i][c]  = grad_out[p][c] if i == argmax[p][c] and z > 0 else 0
//   dgamma[c] = sum_i dz*xh ; dbeta[c] = sum_i dz
//   dx[i][c]  = gamma*rstd * (dz - dbeta/n - xh * dgamma/n)           (train-mode BatchNorm over the n valid points)
//   dW[c][k]  = sum_i dx[i][c] * f[i][k]
// pass 1: scatter dz into ws_dz[n][32] (zeroed), accumulate dgamma/dbeta.  pass 2: dx and dW.
__global__ void k_vfe_bwd_scatter(const float *__restrict__ grad_out, const int32_t *__restrict__ argmax, const float *__restrict__ out_dummy,
                                  int64_t n_pc, float *dz) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pc) return;
    int c = (int)(i & 31);
    int pt = argmax[i];
    if (pt >= 0) dz[(int64_t)pt * VFE_OUT + c] = grad_out[i];   // one writer per (pillar, channel); a point belongs to one pillar
}

template <int n_feat>
__global__ __launch_bounds__(256) void k_vfe_bwd_reduce(const float *__restrict__ points, int n, const int32_t *__restrict__ point_row,
                                                        const int32_t *__restrict__ coords, const float *__restrict__ pillar_acc,
                                                        const float *__restrict__ weight, const float *__restrict__ geom,
                                                        const float *__restrict__ mean, const float *__restrict__ rstd,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float *dz, float *dgamma, float *dbeta) {
    __shared__ float w_lds[VFE_OUT * VFE_MAX_IN];
    __shared__ float red[2][8][VFE_OUT];
    load_w(weight, 9 + n_feat, w_lds);
    __syncthreads();
    const Geom g = load_geom(geom);
    const int c = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    float sg = 0.f, sb = 0.f;
    for (int i = blockIdx.x * 8 + sub; i < n; i += gridDim.x * 8) {
        int row = point_row[i];
        if (row < 0) continue;
        float d = dz[(int64_t)i * VFE_OUT + c];
        if (d == 0.f) continue;
        float x = linear_out<n_feat>(points + (int64_t)i * (1 + n_feat), coords, pillar_acc, row, g, w_lds, c);
        float xh = (x - mu) * rs;
        float z = fmaf(ga, xh, be);
        if (!(z > 0.f)) { d = 0.f; dz[(int64_t)i * VFE_OUT + c] = 0.f; }
        sg += d * xh;
        sb += d;
    }
    red[0][sub][c] = sg;
    red[1][sub][c] = sb;
    __syncthreads();
    if (sub == 0) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < 8; ++k) { a += red[0][k][c]; b += red[1][k][c]; }
        atomicAdd(&dgamma[c], a);
        atomicAdd(&dbeta[c], b);
    }
}

template <int n_feat>
__global__ __launch_bounds__(256) void k_vfe_bwd_weight(const float *__restrict__ points, int n, const int32_t *__restrict__ point_row,
                                                        const int32_t *__restrict__ coords, const float *__restrict__ pillar_acc,
                                                        const float *__restrict__ weight, const float *__restrict__ geom,
                                                        const float *__restrict__ mean, const float *__restrict__ rstd,
                                                        const float *__restrict__ gamma, const float *__restrict__ dz,
                                                        const float *__restrict__ dgamma, const float *__restrict__ dbeta, float inv_n_arg,
                                                        const float *__restrict__ count_dev, float *dW) {
    __shared__ float w_lds[VFE_OUT * VFE_MAX_IN];
    __shared__ float red[8][VFE_OUT][VFE_MAX_IN];
    load_w(weight, 9 + n_feat, w_lds);
    __syncthreads();
    const Geom g = load_geom(geom);
    constexpr int cin = 9 + n_feat;
    const int c = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const float mu = mean[c], rs = rstd[c], ga = gamma[c];
    const float inv_n = count_dev ? 1.0f / count_dev[0] : inv_n_arg;
    const float m1 = dbeta[c] * inv_n, m2 = dgamma[c] * inv_n;
    float acc[VFE_MAX_IN];
#pragma unroll
    for (int k = 0; k < VFE_MAX_IN; ++k) acc[k] = 0.f;
    for (int i = blockIdx.x * 8 + sub; i < n; i += gridDim.x * 8) {
        int row = point_row[i];
        if (row < 0) continue;
        float f[VFE_MAX_IN];
        point_feature<n_feat>(points + (int64_t)i * (1 + n_feat), coords + (int64_t)row * 3, pillar_acc + (int64_t)row * 4, g, f);
        float x = 0.f;
#pragma unroll
        for (int k = 0; k < cin; ++k) x = fmaf(f[k], w_lds[c * VFE_MAX_IN + k], x);
        float xh = (x - mu) * rs;
        float dx = ga * rs * (dz[(int64_t)i * VFE_OUT + c] - m1 - xh * m2);
#pragma unroll
        for (int k = 0; k < cin; ++k) acc[k] = fmaf(dx, f[k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < VFE_MAX_IN; ++k) red[sub][c][k] = acc[k];
    __syncthreads();
    for (int e = threadIdx.x; e < VFE_OUT * cin; e += blockDim.x) {
        int cc = e / cin, k = e % cin;
        float s = 0.f;
        for (int q = 0; q < 8; ++q) s += red[q][cc][k];
        atomicAdd(&dW[cc * cin + k], s);
    }
}

// scatter grad_out to the arg-max points (dz, in ws) + this rank's (grad_gamma, grad_beta)
static int vfe_bwd_reduce(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords, const float *pillar_acc,
                          const float *weight, const float *geom, const float *mean, const float *rstd, const float *gamma, const float *beta,
                          const float *grad_out, const int32_t *argmax, int n_pillars, float *grad_gamma, float *grad_beta, float *ws,
                          hipStream_t st, const char *who) {
    RD_REQUIRE(9 + n_feat <= VFE_MAX_IN, "%s: too many point features", who);
    RD_HIP(hipMemsetAsync(grad_gamma, 0, VFE_OUT * 4, st));
    RD_HIP(hipMemsetAsync(grad_beta, 0, VFE_OUT * 4, st));
    if (n_points <= 0 || n_pillars <= 0) return RD_OK;
    float *dz = ws;
    RD_HIP(hipMemsetAsync(dz, 0, (size_t)n_points * VFE_OUT * 4, st));
    int64_t n_pc = (int64_t)n_pillars * VFE_OUT;
    k_vfe_bwd_scatter<<<cdiv(n_pc, 256), 256, 0, st>>>(grad_out, argmax, nullptr, n_pc, dz);
    int blocks = g_deterministic ? 1 : (int)std::min<int64_t>(cdiv(n_points, 8), 1024);
    VFE_DISPATCH(n_feat, k_vfe_bwd_reduce<NFC><<<blocks, 256, 0, st>>>(points, n_points, point_row, coords, pillar_acc, weight, geom, mean, rstd,
                                                                       gamma, beta, dz, grad_gamma, grad_beta));
    return check_launch(who);
}

static int vfe_bwd_weight(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords, const float *pillar_acc,
                          const float *weight, const float *geom, const float *mean, const float *rstd, const float *gamma,
                          const float *sum_gamma, const float *sum_beta, int n_pillars, float inv_n, const float *count_dev, float *grad_weight,
                          const float *ws, hipStream_t st, const char *who) {
    const int cin = 9 + n_feat;
    RD_HIP(hipMemsetAsync(grad_weight, 0, (size_t)VFE_OUT * cin * 4, st));
    if (n_points <= 0 || n_pillars <= 0) return RD_OK;
    int blocks2 = g_deterministic ? 1 : (int)std::min<int64_t>(cdiv(n_points, 8), 256);
    VFE_DISPATCH(n_feat, k_vfe_bwd_weight<NFC><<<blocks2, 256, 0, st>>>(points, n_points, point_row, coords, pillar_acc, weight, geom, mean, rstd,
                                                                        gamma, ws, sum_gamma, sum_beta, inv_n, count_dev, grad_weight));
    return check_launch(who);
}

extern "C" int rd_vfe_backward(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords,
                               const float *pillar_acc, const float *weight, const float *geom, const float *mean, const float *rstd,
                               const float *gamma, const float *beta, const float *grad_out, const int32_t *argmax, int n_pillars,
                               int n_valid, float *grad_weight, float *grad_gamma, float *grad_beta, float *ws, void *stream) {
    hipStream_t st = S(stream);
    if (n_valid <= 0) n_pillars = 0;
    int rc = vfe_bwd_reduce(points, n_points, n_feat, point_row, coords, pillar_acc, weight, geom, mean, rstd, gamma, beta, grad_out, argmax, n_pillars,
                            grad_gamma, grad_beta, ws, st, "rd_vfe_backward");
    if (rc) return rc;
    return vfe_bwd_weight(points, n_points, n_feat, point_row, coords, pillar_acc, weight, geom, mean, rstd, gamma, grad_gamma, grad_beta, n_pillars,
                          n_valid > 0 ? 1.0f / (float)n_valid : 0.f, nullptr, grad_weight, ws, st, "rd_vfe_backward");
}

// SyncBatchNorm form of rd_vfe_backward: the caller all-reduces [grad_gamma | grad_beta] over the process group between the two
// calls and passes the group-wide sums and the group-wide valid-point count (device scalar) to the second; `ws` carries dz across.
extern "C" int rd_vfe_backward_reduce(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords,
                                      const float *pillar_acc, const float *weight, const float *geom, const float *mean, const float *rstd,
                                      const float *gamma, const float *beta, const float *grad_out, const int32_t *argmax, int n_pillars,
                                      float *grad_gamma, float *grad_beta, float *ws, void *stream) {
    return vfe_bwd_reduce(points, n_points, n_feat, point_row, coords, pillar_acc, weight, geom, mean, rstd, gamma, beta, grad_out, argmax, n_pillars,
                          grad_gamma, grad_beta, ws, S(stream), "rd_vfe_backward_reduce");
}

extern "C" int rd_vfe_backward_weight(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords,
                                      const float *pillar_acc, const float *weight, const float *geom, const float *mean, const float *rstd,
                                      const float *gamma, const float *sum_gamma, const float *sum_beta, const float *count_dev, int n_pillars,
                                      float *grad_weight, const float *ws, void *stream) {
    RD_REQUIRE(count_dev, "rd_vfe_backward_weight: the group-wide valid-point count (device scalar) is required");
    return vfe_bwd_weight(points, n_points, n_feat, point_row, coords, pillar_acc, weight, geom, mean, rstd, gamma, sum_gamma, sum_beta, n_pillars, 0.f,
                          count_dev, grad_weight, ws, S(stream), "rd_vfe_backward_weight");
}
