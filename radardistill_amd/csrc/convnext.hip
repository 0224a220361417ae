// GELU + Global Response Normalisation of the ConvNeXt-V2 MLP, fused.  See include/rdamd.h section O.
// Reference: pcdet/ops/basicblock/modules/Basicblock_convn.py:46-60 (act = nn.GELU(), GRN.forward):
//     a = gelu(z);  G[b][c] = ||a[b, :, :, c]||_2;  N = G / (mean_c G + 1e-6);  out = gamma * (a * N) + beta + a
// on a (B, H, W, 4*dim) tensor = rows (B*HW, C).  ATen runs this as ~9 forward and ~16 backward element-wise / reduction launches
// over a 33 MB tensor per block; here: forward = GELU + per-sample sum of squares (one pass) + apply (one pass), backward = one
// reduction pass + one apply pass that also applies gelu'(z).  HBM-bound.
//   backward:  S[b][c] = sum_p g a,  T[b] = sum_c gamma_c S_bc G_bc,  m = mean_c G,
//              da = g (1 + gamma N) + a/G * (gamma S / (m + eps) - T / (C (m + eps)^2)),  dz = da * gelu'(z),
//              dgamma_c = sum_b N_bc S_bc,  dbeta_c = sum g.
#include <algorithm>
#include "common.hpp"

using namespace rd;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gelu_e(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_d(float z) {
    const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * z * z);
    return cdf + z * pdf;
}

constexpr int GRN_COLS = 256;   // columns per block (64 lanes x float4)
constexpr int GRN_EPS_NUM = 0;  // (eps is 1e-6, below)

// MODE 0 (forward):  a = gelu(z) stored, acc1[b][c] += a^2
// MODE 1 (backward): acc1[b][c] += g * a,  acc2[c] += g
// grid = (row blocks per sample, column chunks of 256, B); 256 threads = 4 row groups x 64 lanes (float4 each)
template <int MODE>
__global__ __launch_bounds__(256) void k_grn_reduce(const float *__restrict__ x, const float *__restrict__ y, int64_t hw, int C, float *a_out,
                                                    float *acc1, float *acc2) {
    __shared__ float red[2][4][GRN_COLS];
    const int b = blockIdx.z, col = blockIdx.y * GRN_COLS + (threadIdx.x & 63) * 4, rg = threadIdx.x >> 6;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (col < C) {
        for (int64_t r = (int64_t)blockIdx.x * 4 + rg; r < hw; r += (int64_t)gridDim.x * 4) {
            const int64_t o = ((int64_t)b * hw + r) * C + col;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + o);
            if (MODE == 0) {
                f32x4 a;
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = gelu_e(v[k]);
                *reinterpret_cast<f32x4 *>(a_out + o) = a;
                s1 += a * a;
            } else {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(y + o);
                s1 += v * av;
                s2 += v;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[0][rg][(threadIdx.x & 63) * 4 + k] = s1[k];
        red[1][rg][(threadIdx.x & 63) * 4 + k] = s2[k];
    }
    __syncthreads();
    const int c = threadIdx.x;                       // 256 threads <-> 256 columns of the chunk
    const int gc = blockIdx.y * GRN_COLS + c;
    if (gc < C) {
        atomicAdd(&acc1[(int64_t)b * C + gc], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        if (MODE == 1 && acc2) atomicAdd(&acc2[gc], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}

// per-sample prologue shared by both apply kernels: G = sqrt(ssq), m = mean_c G -> LDS table of the per-(b,c) coefficients
// MODE 0 (forward):  out = a * (1 + gamma N) + beta
// MODE 1 (backward): dz = (g (1 + gamma N) + a * k2) * gelu'(z);  block (0, b) adds N_bc S_bc into dgamma
template <int MODE>
__global__ __launch_bounds__(256) void k_grn_apply(const float *__restrict__ a, const float *__restrict__ g, const float *__restrict__ z,
                                                   const float *__restrict__ ssq, const float *__restrict__ S, const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, int64_t hw, int C, float *out, float *dgamma) {
    extern __shared__ float tab[];   // [C] k1 = 1 + gamma N, [C] k2 (MODE 1) or beta (MODE 0)
    __shared__ float s_red[256];
    __shared__ float s_m, s_T;
    const int b = blockIdx.y, tid = threadIdx.x;
    float part = 0.f, partT = 0.f;
    for (int c = tid; c < C; c += 256) {
        const float G = sqrtf(ssq[(int64_t)b * C + c]);
        part += G;
        if (MODE == 1) partT += gamma[c] * S[(int64_t)b * C + c] * G;
    }
    s_red[tid] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) s_red[tid] += s_red[tid + s];
        __syncthreads();
    }
    if (tid == 0) s_m = s_red[0] / (float)C + 1e-6f;
    __syncthreads();
    if (MODE == 1) {
        s_red[tid] = partT;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) s_red[tid] += s_red[tid + s];
            __syncthreads();
        }
        if (tid == 0) s_T = s_red[0];
        __syncthreads();
    }
    const float m = s_m;
    for (int c = tid; c < C; c += 256) {
        const float G = sqrtf(ssq[(int64_t)b * C + c]);
        const float N = G / m;
        tab[c] = 1.f + gamma[c] * N;
        if (MODE == 0) {
            tab[C + c] = beta[c];
        } else {
            const float Sv = S[(int64_t)b * C + c];
            tab[C + c] = G > 0.f ? (gamma[c] * Sv / m - s_T / ((float)C * m * m)) / G : 0.f;
            if (blockIdx.x == 0 && dgamma) atomicAdd(&dgamma[c], N * Sv);
        }
    }
    __syncthreads();
    const int64_t n4 = hw * C / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 4) % C);
        const int64_t o = (int64_t)b * n4 + i;
        const f32x4 av = reinterpret_cast<const f32x4 *>(a)[o];
        f32x4 r;
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = fmaf(av[k], tab[c + k], tab[C + c + k]);
        } else {
            const f32x4 gv = reinterpret_cast<const f32x4 *>(g)[o], zv = reinterpret_cast<const f32x4 *>(z)[o];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = fmaf(gv[k], tab[c + k], av[k] * tab[C + c + k]) * gelu_d(zv[k]);
        }
        reinterpret_cast<f32x4 *>(out)[o] = r;
    }
}

// rd_set_deterministic(1): the two sums over SAMPLES (dbeta_c = sum g, dgamma_c = sum_b N_bc S_bc) in sample order, one block.
__global__ __launch_bounds__(256) void k_grn_param_grads_ordered(const float *__restrict__ g, const float *__restrict__ ssq, const float *__restrict__ S,
                                                                 int B, int64_t hw, int C, float *dgamma, float *dbeta) {
    __shared__ float s_red[256];
    __shared__ float s_m;
    const int tid = threadIdx.x;
    for (int b = 0; b < B; ++b) {
        float part = 0.f;
        for (int c = tid; c < C; c += 256) part += sqrtf(ssq[(int64_t)b * C + c]);
        s_red[tid] = part;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) s_red[tid] += s_red[tid + s];
            __syncthreads();
        }
        if (tid == 0) s_m = s_red[0] / (float)C + 1e-6f;
        __syncthreads();
        for (int c = tid; c < C; c += 256) dgamma[c] += sqrtf(ssq[(int64_t)b * C + c]) / s_m * S[(int64_t)b * C + c];
        __syncthreads();
    }
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int64_t r = 0; r < (int64_t)B * hw; ++r) s += g[r * C + c];
        dbeta[c] = s;
    }
}

static int grn_check(int B, int64_t hw, int C, const char *who) {
    RD_REQUIRE(B >= 1 && B <= 65535 && hw >= 1 && C >= 4 && C % 4 == 0 && C <= 8192, "%s: bad sizes (C multiple of 4, <= 8192)", who);
    return RD_OK;
}

extern "C" int rd_gelu_grn_fwd(const float *z, int B, int64_t hw, int C, const float *gamma, const float *beta, float *a, float *ssq, float *out,
                               void *stream) {
    int rc = grn_check(B, hw, C, "rd_gelu_grn_fwd");
    if (rc) return rc;
    hipStream_t st = S(stream);
    RD_HIP(hipMemsetAsync(ssq, 0, (size_t)B * C * 4, st));
    const int rb = g_deterministic ? 1 : (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(hw, 64), 64));
    k_grn_reduce<0><<<dim3(rb, (unsigned)cdiv(C, GRN_COLS), B), 256, 0, st>>>(z, nullptr, hw, C, a, ssq, nullptr);
    const int ab = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(hw * C / 4, 256 * 8), 256));
    k_grn_apply<0><<<dim3(ab, B), 256, (size_t)2 * C * 4, st>>>(a, nullptr, nullptr, ssq, nullptr, gamma, beta, hw, C, out, nullptr);
    return check_launch("rd_gelu_grn_fwd");
}

extern "C" int rd_gelu_grn_bwd(const float *grad_out, const float *a, const float *z, const float *ssq, int B, int64_t hw, int C, const float *gamma,
                               float *S_ws, float *grad_z, float *grad_gamma, float *grad_beta, void *stream) {
    int rc = grn_check(B, hw, C, "rd_gelu_grn_bwd");
    if (rc) return rc;
    hipStream_t st = S(stream);
    if (grad_gamma == S_ws + (size_t)B * C && grad_beta == grad_gamma + C) {          // one [S | grad_gamma | grad_beta] buffer: one fill
        RD_HIP(hipMemsetAsync(S_ws, 0, (size_t)(B + 2) * C * 4, st));
    } else {
        RD_HIP(hipMemsetAsync(S_ws, 0, (size_t)B * C * 4, st));
        RD_HIP(hipMemsetAsync(grad_gamma, 0, (size_t)C * 4, st));
        RD_HIP(hipMemsetAsync(grad_beta, 0, (size_t)C * 4, st));
    }
    const bool det = g_deterministic != 0;
    const int rb = det ? 1 : (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(hw, 64), 64));
    k_grn_reduce<1><<<dim3(rb, (unsigned)cdiv(C, GRN_COLS), B), 256, 0, st>>>(grad_out, a, hw, C, nullptr, S_ws, det ? nullptr : grad_beta);
    const int ab = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(hw * C / 4, 256 * 8), 256));
    k_grn_apply<1><<<dim3(ab, B), 256, (size_t)2 * C * 4, st>>>(a, grad_out, z, ssq, S_ws, gamma, nullptr, hw, C, grad_z, det ? nullptr : grad_gamma);
    if (det) k_grn_param_grads_ordered<<<1, 256, 0, st>>>(grad_out, ssq, S_ws, B, hw, C, grad_gamma, grad_beta);
    return check_launch("rd_gelu_grn_bwd");
}
