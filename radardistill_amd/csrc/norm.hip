// BatchNorm over rows (channels-last), fused affine + residual + activation, their backward, column sums, and the
// sparse <-> dense row scatter.  See include/rdamd.h sections D and E.  All kernels are HBM-bound streaming passes:
// float4 per lane, consecutive lanes on consecutive channels of a row (rows are contiguous, so a wave covers 1 KiB).
// Column reductions are one launch: per-block partial sums combined with fp32 atomics (see k_colreduce).
#include <stdlib.h>
#include "common.hpp"

using namespace rd;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 256 workgroups is the measured optimum for the BatchNorm backward reduction (rocprofv3 averages over the bench: 64 -> 61 us,
// 128 -> 42 us, 256 -> 29 us, 1024 -> 36 us, 2048 -> 34 us): fewer leave too few loads in flight, more pay for their partial sums'
// atomics (workgroups x 2C floats).  RD_RED_MAX_BLOCKS overrides.
constexpr int RED_MAX_BLOCKS = 256;

__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float z) {
    const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);
    return cdf + z * pdf;
}

// Generic column reduction: F(row, c) -> (v1, v2) per element, summed over rows into out1[C] / out2[C] (out2 may be null).
// Each block reduces its rows in registers and LDS, then adds its partial with fp32 atomics: one contiguous run of <= 512 columns
// per block, i.e. full-rate 256-byte atomic wave-instructions; 256 blocks x 2C floats is ~0.5 MB of atomic traffic per launch.
// The outputs therefore ACCUMULATE (callers zero-fill) and the summation order varies from run to run in the last bits.
// Columns are split into chunks of RED_CHUNK (grid.y), so any C % 4 == 0 works (the batched CenterHead BatchNorm has C = 2688).
constexpr int RED_CHUNK = 512;

template <class F>
__global__ __launch_bounds__(256) void k_colreduce(int64_t rows, int C, F f, float *out1, float *out2) {
    extern __shared__ float sm[];  // [groups][2][cw] staged reduction
    const int col0 = blockIdx.y * RED_CHUNK;
    const int cw = min(RED_CHUNK, C - col0);
    const int tpr = cw / 4;              // threads per row (<= 128)
    const int groups = 256 / tpr;
    const int tid = threadIdx.x;
    const int g = tid / tpr, c4 = (tid % tpr) * 4;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (g < groups) {
        const typename F::Ctx ctx = f.prepare(col0 + c4);
#pragma unroll 4
        for (int64_t r = (int64_t)blockIdx.x * groups + g; r < rows; r += (int64_t)gridDim.x * groups) {
            f32x4 a, b;
            f(ctx, r, col0 + c4, a, b);
            s1 += a;
            s2 += b;
        }
        float *dst = sm + (int64_t)g * 2 * cw;
        *reinterpret_cast<f32x4 *>(dst + c4) = s1;
        *reinterpret_cast<f32x4 *>(dst + cw + c4) = s2;
    }
    __syncthreads();
    for (int i = tid; i < (out2 ? 2 : 1) * cw; i += 256) {
        float s = 0.f;
        for (int q = 0; q < groups; ++q) s += sm[(int64_t)q * 2 * cw + i];
        if (i < cw) atomicAdd(&out1[col0 + i], s);
        else atomicAdd(&out2[col0 + i - cw], s);
    }
}

template <class F>
static int colreduce(int64_t rows, int C, F f, float *out1, float *out2, hipStream_t st, const char *who) {
    RD_REQUIRE(C % 4 == 0 && C >= 4, "%s: C=%d must be a positive multiple of 4", who, C);
    const int cw = std::min(C, RED_CHUNK), tpr = cw / 4, groups = 256 / tpr;
    const int chunks = (int)cdiv(C, RED_CHUNK);
    static const int red_max = getenv("RD_RED_MAX_BLOCKS") ? atoi(getenv("RD_RED_MAX_BLOCKS")) : RED_MAX_BLOCKS;    // tuning knob
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(red_max, cdiv(rows, (int64_t)groups * 8)));
    if (g_deterministic) blocks = 1;       // one contributor per output element: fixed summation order
    size_t shm = (size_t)groups * 2 * cw * 4;
    k_colreduce<F><<<dim3(blocks, chunks), 256, shm, st>>>(rows, C, f, out1, out2);
    return check_launch(who);
}

// Functors: prepare(c) loads whatever depends only on the thread's 4 channels (fixed over its row loop) into registers.
struct StatsF {
    const float *x;
    int C;
    struct Ctx {};
    __device__ Ctx prepare(int) const { return Ctx{}; }
    __device__ void operator()(const Ctx &, int64_t r, int c, f32x4 &a, f32x4 &b) const {
        a = *reinterpret_cast<const f32x4 *>(x + r * C + c);
        b = a * a;
    }
};

extern "C" int rd_bn_stats(const float *x, int64_t rows, int C, float *stats, void *stream) {
    if (rows <= 0) return RD_OK;
    return colreduce(rows, C, StatsF{x, C}, stats, stats + C, S(stream), "rd_bn_stats");
}

struct ColsumF {
    const float *x;
    int C;
    struct Ctx {};
    __device__ Ctx prepare(int) const { return Ctx{}; }
    __device__ void operator()(const Ctx &, int64_t r, int c, f32x4 &a, f32x4 &b) const {
        a = *reinterpret_cast<const f32x4 *>(x + r * C + c);
        b = f32x4{0.f, 0.f, 0.f, 0.f};
    }
};
extern "C" int rd_colsum(const float *x, int64_t rows, int C, float *out, void *stream) {
    if (rows <= 0) return RD_OK;
    return colreduce(rows, C, ColsumF{x, C}, out, nullptr, S(stream), "rd_colsum");
}

__global__ void k_bn_finalize(const float *stats, float n_arg, int C, const float *gamma, const float *beta, float eps, float momentum,
                              float *running_mean, float *running_var, float *mean_out, float *rstd_out, float *scale, float *shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float n = n_arg > 0.f ? n_arg : stats[2 * C];
    double mean = (double)stats[c] / n;
    double var = (double)stats[C + c] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    float rstd = (float)(1.0 / sqrt(var + (double)eps));
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    float sc = g * rstd;
    if (mean_out) mean_out[c] = (float)mean;
    if (rstd_out) rstd_out[c] = rstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        double unbiased = n > 1.f ? var * n / (n - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

extern "C" int rd_bn_finalize(const float *stats, int64_t rows, int C, const float *gamma, const float *beta, float eps, float momentum,
                              float *running_mean, float *running_var, float *mean, float *rstd, float *scale, float *shift, void *stream) {
    RD_REQUIRE(rows > 0, "rd_bn_finalize: BatchNorm over zero rows");
    k_bn_finalize<<<cdiv(C, 256), 256, 0, S(stream)>>>(stats, (float)rows, C, gamma, beta, eps, momentum, running_mean, running_var, mean,
                                                       rstd, scale, shift);
    return check_launch("rd_bn_finalize");
}

__global__ void k_affine_act(const float *__restrict__ x, int64_t n4, int C, const float *__restrict__ scale, const float *__restrict__ shift,
                             const float *__restrict__ residual, int act, float *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)((i * 4) % C);
        f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
        f32x4 sc = scale ? *reinterpret_cast<const f32x4 *>(scale + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        f32x4 sh = shift ? *reinterpret_cast<const f32x4 *>(shift + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        v = v * sc + sh;
        if (residual) v += reinterpret_cast<const f32x4 *>(residual)[i];
        if (act == 1) {
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        } else if (act == 2) {
            for (int k = 0; k < 4; ++k) v[k] = gelu_f(v[k]);
        }
        reinterpret_cast<f32x4 *>(y)[i] = v;
    }
}

extern "C" int rd_affine_act(const float *x, int64_t rows, int C, const float *scale, const float *shift, const float *residual, int act,
                             float *y, void *stream) {
    RD_REQUIRE(C % 4 == 0, "rd_affine_act: C=%d must be a multiple of 4", C);
    int64_t n4 = rows * C / 4;
    if (n4 <= 0) return RD_OK;
    int blocks = (int)std::min<int64_t>(cdiv(n4, 256), 4096);
    k_affine_act<<<blocks, 256, 0, S(stream)>>>(x, n4, C, scale, shift, residual, act, y);
    return check_launch("rd_affine_act");
}

// stats[2C + 1]: (sum, sum of squares, row count) summed over the process group (SyncBatchNorm)
extern "C" int rd_bn_finalize_sync(const float *stats, int C, const float *gamma, const float *beta, float eps, float momentum, float *running_mean,
                                   float *running_var, float *mean, float *rstd, float *scale, float *shift, void *stream) {
    k_bn_finalize<<<cdiv(C, 256), 256, 0, S(stream)>>>(stats, 0.f, C, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale,
                                                       shift);
    return check_launch("rd_bn_finalize_sync");
}

// Train-mode BatchNorm forward in ONE launch: every block derives scale/shift of all C channels from the batch sums into LDS
// (C / 256 channels per thread, a few double operations each), then streams its share of the rows; block 0 also writes
// mean / rstd / scale / shift for the backward pass and updates the running statistics.
__global__ __launch_bounds__(256) void k_bn_train_fwd(const float *__restrict__ x, int64_t n4, int C, const float *__restrict__ stats, float n_arg,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta, float eps, float momentum,
                                                      float *running_mean, float *running_var, const float *__restrict__ residual, int act,
                                                      float *__restrict__ y, float *mean_out, float *rstd_out, float *scale_out, float *shift_out) {
    extern __shared__ float sc_sh[];  // [C] scale, [C] shift
    const float n = n_arg > 0.f ? n_arg : stats[2 * C];      // synchronised statistics: the global row count rides behind the sums
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double mean = (double)stats[c] / n;
        double var = (double)stats[C + c] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const float sc = g * rstd, m = (float)mean;
        const float sh = b - m * sc;
        sc_sh[c] = sc;
        sc_sh[C + c] = sh;
        if (blockIdx.x == 0) {
            if (mean_out) mean_out[c] = m;
            if (rstd_out) rstd_out[c] = rstd;
            if (scale_out) scale_out[c] = sc;
            if (shift_out) shift_out[c] = sh;
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
            if (running_var) {
                const double unbiased = n > 1.f ? var * n / (n - 1.0) : var;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
    __syncthreads();
    // streaming part: 4 independent float4 per thread and iteration (all loads issued before the first use: the kernel is a pure
    // HBM stream, so bytes in flight per CU are what sets its rate), channel index advanced incrementally instead of a 64-bit
    // modulo per element
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int dc = (int)((stride * 4) % C);
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int c = (int)((i * 4) % C);
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x), *r4 = reinterpret_cast<const f32x4 *>(residual);
    f32x4 *y4 = reinterpret_cast<f32x4 *>(y);
    auto finish = [&](f32x4 xv, f32x4 rv, int cc) {
        const f32x4 sc4 = *reinterpret_cast<const f32x4 *>(sc_sh + cc), sh4 = *reinterpret_cast<const f32x4 *>(sc_sh + C + cc);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaf(xv[k], sc4[k], sh4[k]);   // the backward pass re-derives the ReLU mask with this exact expression
        if (residual) v += rv;
        if (act == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        } else if (act == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = gelu_f(v[k]);
        }
        return v;
    };
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 xv[4], rv[4];
        int cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xv[u] = x4[i + u * stride];
            rv[u] = residual ? r4[i + u * stride] : f32x4{0.f, 0.f, 0.f, 0.f};
            cc[u] = c;
            c += dc;
            if (c >= C) c -= C;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) y4[i + u * stride] = finish(xv[u], rv[u], cc[u]);
    }
    for (; i < n4; i += stride) {
        const f32x4 rv = residual ? r4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        y4[i] = finish(x4[i], rv, c);
        c += dc;
        if (c >= C) c -= C;
    }
}

static int bn_train_fwd_impl(const float *x, int64_t rows, float count, int C, const float *stats, const float *gamma, const float *beta, float eps,
                             float momentum, float *running_mean, float *running_var, const float *residual, int act, float *y, float *mean,
                             float *rstd, float *scale, float *shift, void *stream, const char *who) {
    RD_REQUIRE(rows > 0, "%s: BatchNorm over zero rows", who);
    RD_REQUIRE(C % 4 == 0 && C <= 8192 && act >= 0 && act <= 2, "%s: C=%d must be a multiple of 4 (<= 8192), act in 0..2", who, C);
    const int64_t n4 = rows * C / 4;
    // >= 8 float4 per thread amortise the per-block scale/shift table (C/256 channels per thread).  RD_BN_GRID_DIV / RD_BN_GRID_MAX
    // are tuning knobs: 4..16 float4 per thread and 512..4096 workgroups all measure 3.5-3.65 TB/s on the 57 MB launches (isolated) --
    // ~16 us per launch, of which ~4-5 us are launch ramp-up and tail, i.e. the stream itself runs at ~5 TB/s
    static const int gdiv = getenv("RD_BN_GRID_DIV") ? atoi(getenv("RD_BN_GRID_DIV")) : 8;
    static const int gmax = getenv("RD_BN_GRID_MAX") ? atoi(getenv("RD_BN_GRID_MAX")) : 2048;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n4, 256 * (int64_t)gdiv), gmax));
    k_bn_train_fwd<<<blocks, 256, (size_t)2 * C * 4, S(stream)>>>(x, n4, C, stats, count, gamma, beta, eps, momentum, running_mean,
                                                                   running_var, residual, act, y, mean, rstd, scale, shift);
    return check_launch(who);
}

extern "C" int rd_bn_train_fwd(const float *x, int64_t rows, int C, const float *stats, const float *gamma, const float *beta, float eps,
                               float momentum, float *running_mean, float *running_var, const float *residual, int act, float *y,
                               float *mean, float *rstd, float *scale, float *shift, void *stream) {
    return bn_train_fwd_impl(x, rows, (float)rows, C, stats, gamma, beta, eps, momentum, running_mean, running_var, residual, act, y, mean, rstd,
                             scale, shift, stream, "rd_bn_train_fwd");
}

// SyncBatchNorm (tools/train.py:144-145, torch.nn.SyncBatchNorm.convert_sync_batchnorm): stats[2C + 1] = (sum, sum of squares, row
// count) already summed over the process group; `rows` is this rank's share, which is what gets normalised here.
extern "C" int rd_bn_train_fwd_sync(const float *x, int64_t rows, int C, const float *stats, const float *gamma, const float *beta, float eps,
                                    float momentum, float *running_mean, float *running_var, const float *residual, int act, float *y,
                                    float *mean, float *rstd, float *scale, float *shift, void *stream) {
    return bn_train_fwd_impl(x, rows, 0.f, C, stats, gamma, beta, eps, momentum, running_mean, running_var, residual, act, y, mean, rstd, scale,
                             shift, stream, "rd_bn_train_fwd_sync");
}

// ---- backward of y = act(x*scale + shift [+ residual]) with train-mode batch statistics
//   g' = grad_y * act'(.)          (relu: y > 0; gelu: derivative at z = x*scale + shift; residual excluded for gelu)
//   dbeta = sum g', dgamma = sum g' * xhat, xhat = (x - mean) * rstd
//   dx = gamma*rstd * (g' - dbeta/n - xhat * dgamma/n);  grad_residual = g'
struct BnBwdF {
    const float *x, *y, *gy, *mean, *rstd, *scale, *shift;
    int C, act;
    struct Ctx {
        f32x4 mean, rstd, sc, sh;
    };
    __device__ Ctx prepare(int c) const {
        Ctx k;
        k.mean = *reinterpret_cast<const f32x4 *>(mean + c);
        k.rstd = *reinterpret_cast<const f32x4 *>(rstd + c);
        k.sc = *reinterpret_cast<const f32x4 *>(scale + c);
        k.sh = *reinterpret_cast<const f32x4 *>(shift + c);
        return k;
    }
    __device__ void operator()(const Ctx &p, int64_t r, int c, f32x4 &a, f32x4 &b) const {
        f32x4 g = *reinterpret_cast<const f32x4 *>(gy + r * C + c);
        f32x4 xv = *reinterpret_cast<const f32x4 *>(x + r * C + c);
        if (act == 1) {
            if (y) {
                f32x4 yv = *reinterpret_cast<const f32x4 *>(y + r * C + c);
                for (int k = 0; k < 4; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
            } else {  // no residual: the mask is (x*scale + shift > 0), the forward's own expression -- one tensor less to read
                for (int k = 0; k < 4; ++k) g[k] = fmaf(xv[k], p.sc[k], p.sh[k]) > 0.f ? g[k] : 0.f;
            }
        } else if (act == 2) {
            for (int k = 0; k < 4; ++k) g[k] *= gelu_grad(fmaf(xv[k], p.sc[k], p.sh[k]));
        }
        a = g;
        b = g * ((xv - p.mean) * p.rstd);
    }
};

// dx = gamma*rstd*(g' - sum_g/n - xhat*sum_gx/n) = A*g' + B*x + D per channel; the five per-channel coefficients (A, B, D and the
// forward's scale / shift for the activation mask) are built once per block in LDS.
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ gy,
                                                      int64_t n4, int C, const float *__restrict__ gamma, const float *__restrict__ mean,
                                                      const float *__restrict__ rstd, const float *__restrict__ scale,
                                                      const float *__restrict__ shift, int act, const float *__restrict__ sum_g,
                                                      const float *__restrict__ sum_gx, float inv_n_arg, const float *__restrict__ count_dev,
                                                      float *__restrict__ gx, float *__restrict__ gres) {
    extern __shared__ float tab[];  // [5][C]: A, B, D, scale, shift
    const float inv_n = count_dev ? 1.0f / count_dev[0] : inv_n_arg;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float r = rstd[c], A = (gamma ? gamma[c] : 1.f) * r;
        const float Bc = -A * r * (sum_gx[c] * inv_n);
        tab[c] = A;
        tab[C + c] = Bc;
        tab[2 * C + c] = -A * (sum_g[c] * inv_n) - Bc * mean[c];
        tab[3 * C + c] = scale[c];
        tab[4 * C + c] = shift[c];
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % C);
        f32x4 g = reinterpret_cast<const f32x4 *>(gy)[i];
        const f32x4 xv = reinterpret_cast<const f32x4 *>(x)[i];
        if (act == 1) {
            if (y) {
                const f32x4 yv = reinterpret_cast<const f32x4 *>(y)[i];
                for (int k = 0; k < 4; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
            } else {
                for (int k = 0; k < 4; ++k) g[k] = fmaf(xv[k], tab[3 * C + c + k], tab[4 * C + c + k]) > 0.f ? g[k] : 0.f;
            }
        } else if (act == 2) {
            for (int k = 0; k < 4; ++k) g[k] *= gelu_grad(fmaf(xv[k], tab[3 * C + c + k], tab[4 * C + c + k]));
        }
        if (gres) reinterpret_cast<f32x4 *>(gres)[i] = g;
        f32x4 o;
        for (int k = 0; k < 4; ++k) o[k] = fmaf(tab[c + k], g[k], fmaf(tab[C + c + k], xv[k], tab[2 * C + c + k]));
        reinterpret_cast<f32x4 *>(gx)[i] = o;
    }
}

static int bn_bwd_check(int64_t rows, int C, int act, int has_residual, const float *y, const float *grad_gamma, const float *grad_beta,
                        const char *who) {
    RD_REQUIRE(rows > 0, "%s: zero rows", who);
    RD_REQUIRE(act >= 0 && act <= 2, "%s: bad act", who);
    RD_REQUIRE(C <= 8192, "%s: C=%d > 8192", who, C);
    RD_REQUIRE(!(act == 2 && has_residual), "%s: gelu with residual is not supported", who);
    RD_REQUIRE(!(act == 1 && has_residual && y == nullptr), "%s: y is required for ReLU with a residual (the mask depends on the residual)", who);
    RD_REQUIRE(grad_gamma && grad_beta, "%s: grad_gamma / grad_beta are required (zero-filled by the caller; the apply kernel reads them)", who);
    return RD_OK;
}

static int bn_bwd_apply_launch(const float *x, const float *y, const float *grad_y, int64_t rows, int C, const float *gamma, const float *mean,
                               const float *rstd, const float *scale, const float *shift, int act, int has_residual, const float *sum_gamma,
                               const float *sum_beta, const float *count_dev, float *grad_x, float *grad_res, hipStream_t st, const char *who) {
    int64_t n4 = rows * C / 4;
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(n4, 256 * 8), 2048));   // >= 8 float4 per thread: the LDS table is amortised
    k_bn_bwd_apply<<<blocks, 256, (size_t)5 * C * 4, st>>>(x, y, grad_y, n4, C, gamma, mean, rstd, scale, shift, act, sum_beta, sum_gamma,
                                                           1.0f / (float)rows, count_dev, grad_x, has_residual ? grad_res : nullptr);
    return check_launch(who);
}

extern "C" int rd_bn_bwd(const float *x, const float *y, const float *grad_y, int64_t rows, int C, const float *gamma, const float *mean,
                         const float *rstd, const float *scale, const float *shift, int act, int has_residual, float *grad_x,
                         float *grad_res, float *grad_gamma, float *grad_beta, void *stream) {
    int rc = bn_bwd_check(rows, C, act, has_residual, y, grad_gamma, grad_beta, "rd_bn_bwd");
    if (rc) return rc;
    hipStream_t st = S(stream);
    rc = colreduce(rows, C, BnBwdF{x, y, grad_y, mean, rstd, scale, shift, C, act}, grad_beta, grad_gamma, st, "rd_bn_bwd");
    if (rc) return rc;
    return bn_bwd_apply_launch(x, y, grad_y, rows, C, gamma, mean, rstd, scale, shift, act, has_residual, grad_gamma, grad_beta, nullptr, grad_x,
                               grad_res, st, "rd_bn_bwd");
}

// SyncBatchNorm backward in two halves with the process group's all-reduce of [grad_gamma | grad_beta] between them: _reduce
// accumulates this rank's sums (= its parameter gradients, as torch's SyncBatchNorm returns them), _apply takes the group-wide
// sums and the group-wide row count (device scalar) for the input gradient.
extern "C" int rd_bn_bwd_reduce(const float *x, const float *y, const float *grad_y, int64_t rows, int C, const float *mean, const float *rstd,
                                const float *scale, const float *shift, int act, int has_residual, float *grad_gamma, float *grad_beta,
                                void *stream) {
    int rc = bn_bwd_check(rows, C, act, has_residual, y, grad_gamma, grad_beta, "rd_bn_bwd_reduce");
    if (rc) return rc;
    return colreduce(rows, C, BnBwdF{x, y, grad_y, mean, rstd, scale, shift, C, act}, grad_beta, grad_gamma, S(stream), "rd_bn_bwd_reduce");
}

extern "C" int rd_bn_bwd_apply(const float *x, const float *y, const float *grad_y, int64_t rows, int C, const float *gamma, const float *mean,
                               const float *rstd, const float *scale, const float *shift, int act, int has_residual, const float *sum_gamma,
                               const float *sum_beta, const float *count_dev, float *grad_x, float *grad_res, void *stream) {
    int rc = bn_bwd_check(rows, C, act, has_residual, y, sum_gamma, sum_beta, "rd_bn_bwd_apply");
    if (rc) return rc;
    RD_REQUIRE(count_dev, "rd_bn_bwd_apply: the group-wide row count (device scalar) is required");
    return bn_bwd_apply_launch(x, y, grad_y, rows, C, gamma, mean, rstd, scale, shift, act, has_residual, sum_gamma, sum_beta, count_dev, grad_x,
                               grad_res, S(stream), "rd_bn_bwd_apply");
}

// ---------------------------------------------------------------------------------------------- sparse <-> dense
__global__ void k_rows_to_dense(const float *__restrict__ feats, const int32_t *__restrict__ coords, int64_t n4, int C, int H, int W,
                                float *__restrict__ dense, int to_dense) {
    const int c4n = C / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i / c4n;
        int c = (int)(i % c4n) * 4;
        int b = coords[r * 3], y = coords[r * 3 + 1], x = coords[r * 3 + 2];
        int64_t cell = ((int64_t)b * H + y) * W + x;
        if (to_dense) *reinterpret_cast<f32x4 *>(dense + cell * C + c) = *reinterpret_cast<const f32x4 *>(feats + r * C + c);
        else *reinterpret_cast<f32x4 *>(const_cast<float *>(feats) + r * C + c) = *reinterpret_cast<const f32x4 *>(dense + cell * C + c);
    }
}

extern "C" int rd_rows_to_dense(const float *feats, const int32_t *coords, int n, int C, int batch, int H, int W, float *dense, void *stream) {
    RD_REQUIRE(C % 4 == 0, "rd_rows_to_dense: C %% 4 != 0");
    hipStream_t st = S(stream);
    RD_HIP(hipMemsetAsync(dense, 0, (size_t)batch * H * W * C * 4, st));
    int64_t n4 = (int64_t)n * C / 4;
    if (n4 <= 0) return RD_OK;
    k_rows_to_dense<<<(int)std::min<int64_t>(cdiv(n4, 256), 4096), 256, 0, st>>>(feats, coords, n4, C, H, W, dense, 1);
    return check_launch("rd_rows_to_dense");
}

extern "C" int rd_dense_to_rows(const float *dense, const int32_t *coords, int n, int C, int batch, int H, int W, float *feats, void *stream) {
    RD_REQUIRE(C % 4 == 0, "rd_dense_to_rows: C %% 4 != 0");
    int64_t n4 = (int64_t)n * C / 4;
    if (n4 <= 0) return RD_OK;
    k_rows_to_dense<<<(int)std::min<int64_t>(cdiv(n4, 256), 4096), 256, 0, S(stream)>>>(feats, coords, n4, C, H, W, const_cast<float *>(dense), 0);
    return check_launch("rd_dense_to_rows");
}


// ---------------------------------------------------------------------------------------------- channel concatenation of two row tensors
// torch.cat((a, b), dim=1) of two channels-last maps (base_bev_backbone.py:296, radar_distill_final.py:121-124) and its backward
// as ONE streaming launch each: out[r] = [a[r] | b[r]]; backward g -> (ga, gb) CONTIGUOUS.  ATen's cat backward hands out strided
// slices of g, which every consumer then copied (`contiguous`) and autograd's fan-out adds walked at 1.6 TB/s.
__global__ __launch_bounds__(256) void k_cat2_rows(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b, int64_t rows, int ca4, int cb4,
                                                   f32x4 *__restrict__ out, int to_out) {
    const int c4 = ca4 + cb4;
    const int64_t n = rows * c4;
    f32x4 *aw = const_cast<f32x4 *>(a), *bw = const_cast<f32x4 *>(b);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c4;
        const int c = (int)(i - r * c4);
        if (to_out) out[i] = c < ca4 ? a[r * ca4 + c] : b[r * cb4 + (c - ca4)];
        else if (c < ca4) aw[r * ca4 + c] = out[i];
        else bw[r * cb4 + (c - ca4)] = out[i];
    }
}

extern "C" int rd_cat2_rows(const float *a, int Ca, const float *b, int Cb, int64_t rows, float *out, void *stream) {
    RD_REQUIRE(Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0, "rd_cat2_rows: channel counts %d, %d must be positive multiples of 4", Ca, Cb);
    if (rows <= 0) return RD_OK;
    const int64_t n = rows * (Ca + Cb) / 4;
    k_cat2_rows<<<(unsigned)std::min<int64_t>(cdiv(n, 256 * 4), 4096), 256, 0, S(stream)>>>(reinterpret_cast<const f32x4 *>(a), reinterpret_cast<const f32x4 *>(b),
                                                                                           rows, Ca / 4, Cb / 4, reinterpret_cast<f32x4 *>(out), 1);
    return check_launch("rd_cat2_rows");
}

extern "C" int rd_split2_rows(const float *g, int64_t rows, int Ca, int Cb, float *ga, float *gb, void *stream) {
    RD_REQUIRE(Ca > 0 && Cb > 0 && Ca % 4 == 0 && Cb % 4 == 0, "rd_split2_rows: channel counts %d, %d must be positive multiples of 4", Ca, Cb);
    if (rows <= 0) return RD_OK;
    const int64_t n = rows * (Ca + Cb) / 4;
    k_cat2_rows<<<(unsigned)std::min<int64_t>(cdiv(n, 256 * 4), 4096), 256, 0, S(stream)>>>(reinterpret_cast<const f32x4 *>(ga), reinterpret_cast<const f32x4 *>(gb),
                                                                                           rows, Ca / 4, Cb / 4, reinterpret_cast<f32x4 *>(const_cast<float *>(g)), 0);
    return check_launch("rd_split2_rows");
}
