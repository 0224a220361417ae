// Narrow-output 3x3 convolutions of the CenterHead branches, all branches of all task heads in ONE launch.
// See include/rdamd.h section L.
//
// SeparateHead (pcdet/models/dense_heads/radar_center_head.py:28-63) ends every branch with Conv2d(64 -> n, k3, p1), n = 1..3
// (center 2, center_z 1, dim 3, rot 2, vel 2, iou 1, hm 1-2): 42 such convolutions per head module on nuScenes.  On the
// matrix cores a 1..3-wide output wastes > 90 % of every 32-wide tile, so these run on the vector ALUs instead, straight from
// the batched activation tensor y (rows, NB*64) that the batched first convolution + BatchNorm + ReLU of all branches produced:
//
//     out[p][col_b + n] = bias[col_b + n] + sum_t sum_c y[p + d_t][64 b + c] * w[col_b + n][c][t]
//
// Work split (all three kernels): grid = (B * ceil(H/8) row bands, NB branches), 256 threads; a block walks the 8x8 pixel
// tiles of its band.  A thread owns 4 consecutive channels (lane & 15) of the pixels p = (tid >> 4) + 16 i of the tile, so the 16
// lanes of a pixel read one contiguous 256-byte LDS row and weights / weight-gradient accumulators ([n][9 taps][4 channels])
// live in registers for the whole band.  The 10x10x64 input halo of a tile is staged through LDS once (1.56x re-read from L2
// instead of 9x).  HBM-bound: y is 352 MB at B = 8, 64x64 map, 42 branches.
#include <algorithm>
#include "common.hpp"

using namespace rd;

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NC_CB = 64;            // input channels per branch
constexpr int NC_T = 8;              // output tile edge
constexpr int NC_HALO = NC_T + 2;    // 10
constexpr int NC_LDW = NC_CB + 4;    // LDS floats per halo pixel
constexpr int NC_MAXB = 64;          // branches per launch
constexpr int NC_MAXN = 4;           // outputs per branch

struct NconvArgs {
    const float *y;        // (rows, ldy)
    int ldy;               // row stride of y / grad_y (floats)
    const float *w;        // torch Conv2d layout [NO][64][3][3]
    const float *bias;     // [NO] or null
    const float *go;       // (rows, NO)
    float *out;            // fwd: (rows, NO); dgrad: grad_y (rows, ldy); wgrad: grad_w [NO][64][9]
    int NO, NB, B, H, W;
    int cin_off[NC_MAXB];  // first channel of branch b in y
    int col_off[NC_MAXB];  // first output column of branch b
    int n_out[NC_MAXB];    // outputs of branch b (1..4)
    // data gradient fused with the backward of the train-mode BatchNorm + ReLU in front of these convolutions (k_nconv_dgrad_bn)
    const float *x = nullptr;                                  // BatchNorm input (rows, ldy): the first convolutions' raw output
    const float *mean = nullptr, *rstd = nullptr, *scale = nullptr, *shift = nullptr, *gamma = nullptr;     // per channel, [ldy]
    float *sum_g = nullptr, *sum_gx = nullptr;                 // pass 1 accumulates, pass 2 reads: [ldy] each
    float inv_n = 0.f;
};

__device__ __forceinline__ float reduce16(float v) {
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
    return v;
}

// The 10x10x64 halo of a tile is 1600 float4: 7 per thread (the last pass covers 64 of the 256 threads).  It is fetched into
// registers one tile AHEAD (the loads fly under the current tile's arithmetic) and written to LDS after the barrier.
constexpr int NC_HP = (NC_HALO * NC_HALO * (NC_CB / 4) + 255) / 256;   // 7

__device__ __forceinline__ void halo_fetch(const NconvArgs &a, int b, int y0, int x0, int cin0, f32x4 (&pre)[NC_HP]) {
#pragma unroll
    for (int p = 0; p < NC_HP; ++p) {
        const int i = threadIdx.x + 256 * p;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < NC_HALO * NC_HALO * (NC_CB / 4)) {
            const int hp = i >> 4, q = i & 15;
            const int gy = y0 - 1 + hp / NC_HALO, gx = x0 - 1 + hp % NC_HALO;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = *reinterpret_cast<const f32x4 *>(a.y + ((int64_t)(b * a.H + gy) * a.W + gx) * a.ldy + cin0 + 4 * q);
        }
        pre[p] = v;
    }
}

__device__ __forceinline__ void halo_store(const f32x4 (&pre)[NC_HP], float *lds) {
#pragma unroll
    for (int p = 0; p < NC_HP; ++p) {
        const int i = threadIdx.x + 256 * p;
        if (i < NC_HALO * NC_HALO * (NC_CB / 4)) *reinterpret_cast<f32x4 *>(lds + (i >> 4) * NC_LDW + 4 * (i & 15)) = pre[p];
    }
}

template <int N>
__device__ __forceinline__ void load_weights(const NconvArgs &a, int col0, int l16, float (&w)[N][9][4]) {
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 9; ++t) w[n][t][j] = a.w[((int64_t)(col0 + n) * NC_CB + 4 * l16 + j) * 9 + t];
}

template <int N>
__device__ __forceinline__ void nconv_fwd_body(const NconvArgs &a, float *lds) {
    const int br = blockIdx.y, col0 = a.col_off[br], cin0 = a.cin_off[br];
    const int bands = (a.H + NC_T - 1) / NC_T;
    const int b = blockIdx.x / bands, y0 = (blockIdx.x % bands) * NC_T;
    const int l16 = threadIdx.x & 15, pg = threadIdx.x >> 4;
    float w[N][9][4];
    load_weights<N>(a, col0, l16, w);
    float bias[N];
#pragma unroll
    for (int n = 0; n < N; ++n) bias[n] = a.bias ? a.bias[col0 + n] : 0.f;
    f32x4 pre[NC_HP];
    halo_fetch(a, b, y0, 0, cin0, pre);
    for (int x0 = 0; x0 < a.W; x0 += NC_T) {
        halo_store(pre, lds);
        __syncthreads();
        if (x0 + NC_T < a.W) halo_fetch(a, b, y0, x0 + NC_T, cin0, pre);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = pg + 16 * i, py = p >> 3, px = p & 7;
            float acc[N];
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f32x4 x = *reinterpret_cast<const f32x4 *>(lds + ((py + t / 3) * NC_HALO + px + t % 3) * NC_LDW + 4 * l16);
#pragma unroll
                for (int n = 0; n < N; ++n)
                    acc[n] += x[0] * w[n][t][0] + x[1] * w[n][t][1] + x[2] * w[n][t][2] + x[3] * w[n][t][3];
            }
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = reduce16(acc[n]);
            const int gy = y0 + py, gx = x0 + px;
            if (l16 == 0 && gy < a.H && gx < a.W) {
                float *o = a.out + ((int64_t)(b * a.H + gy) * a.W + gx) * a.NO + col0;
#pragma unroll
                for (int n = 0; n < N; ++n) o[n] = acc[n] + bias[n];
            }
        }
        __syncthreads();
    }
}

// NMAX = widest branch of the launch (3 on nuScenes): the <4> bodies are only compiled into the NMAX = 4 kernels, whose weight /
// accumulator arrays would otherwise set the register count (and halve the resident waves) for every launch.
template <int NMAX>
__global__ __launch_bounds__(256) void k_nconv_fwd(const NconvArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[NC_HALO * NC_HALO * NC_LDW];
    const int n = a.n_out[blockIdx.y];  // block-uniform
    if (n == 1) nconv_fwd_body<1>(a, lds);
    else if (n == 2) nconv_fwd_body<2>(a, lds);
    else if (n == 3 || NMAX == 3) nconv_fwd_body<3>(a, lds);
    else nconv_fwd_body<NMAX>(a, lds);
}

// grad_y[q][cin0 + c] = sum_t sum_n go[q - d_t][col0 + n] * w[col0 + n][c][t],  d_t = (t/3 - 1, t%3 - 1)
template <int N>
__device__ __forceinline__ void nconv_dgrad_body(const NconvArgs &a, float *lds) {
    const int br = blockIdx.y, col0 = a.col_off[br], cin0 = a.cin_off[br];
    const int bands = (a.H + NC_T - 1) / NC_T;
    const int b = blockIdx.x / bands, y0 = (blockIdx.x % bands) * NC_T;
    const int l16 = threadIdx.x & 15, pg = threadIdx.x >> 4;
    float w[N][9][4];
    load_weights<N>(a, col0, l16, w);
    for (int x0 = 0; x0 < a.W; x0 += NC_T) {
        // halo of grad_out: 10x10 pixels x N values
        for (int i = threadIdx.x; i < NC_HALO * NC_HALO * N; i += 256) {
            const int hp = i / N, n = i % N;
            const int gy = y0 - 1 + hp / NC_HALO, gx = x0 - 1 + hp % NC_HALO;
            float v = 0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = a.go[((int64_t)(b * a.H + gy) * a.W + gx) * a.NO + col0 + n];
            lds[hp * NC_MAXN + n] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = pg + 16 * i, py = p >> 3, px = p & 7;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float *g = lds + ((py + 2 - t / 3) * NC_HALO + px + 2 - t % 3) * NC_MAXN;
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    const float gv = g[n];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += gv * w[n][t][j];
                }
            }
            const int gy = y0 + py, gx = x0 + px;
            if (gy < a.H && gx < a.W)
                *reinterpret_cast<f32x4 *>(a.out + ((int64_t)(b * a.H + gy) * a.W + gx) * a.ldy + cin0 + 4 * l16) = acc;
        }
        __syncthreads();
    }
}

template <int NMAX>
__global__ __launch_bounds__(256) void k_nconv_dgrad(const NconvArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[NC_HALO * NC_HALO * NC_MAXN];
    const int n = a.n_out[blockIdx.y];
    if (n == 1) nconv_dgrad_body<1>(a, lds);
    else if (n == 2) nconv_dgrad_body<2>(a, lds);
    else if (n == 3 || NMAX == 3) nconv_dgrad_body<3>(a, lds);
    else nconv_dgrad_body<NMAX>(a, lds);
}

// Data gradient FUSED with the BatchNorm + ReLU backward in front of the narrow convolutions.  The plain sequence is k_nconv_dgrad
// (writes grad_y: 352 MB at B = 8, 42 branches) -> column reduction over (x, grad_y) -> apply pass over (x, grad_y) -> grad_x: 2.1 GB
// of traffic for a gradient that costs <= 27 fma per element to recompute from the 42-channel grad_out.  Here grad_y never exists:
//   PASS 1: recompute grad_y, mask it with the forward's ReLU decision (x * scale + shift > 0), accumulate sum(g) and
//           sum(g * xhat) per channel (registers over the band, LDS over the 16 pixel groups, one atomic per channel and workgroup);
//   PASS 2: recompute again and write grad_x = A g + B x + D straight away (the coefficients of k_bn_bwd_apply, norm.hip).
// Both read x once: 0.35 + 0.7 GB.  Same work split as k_nconv_dgrad.
template <int N, int PASS>
__device__ __forceinline__ void nconv_dgrad_bn_body(const NconvArgs &a, float *lds, float *red) {
    const int br = blockIdx.y, col0 = a.col_off[br], cin0 = a.cin_off[br];
    const int bands = (a.H + NC_T - 1) / NC_T;
    const int b = blockIdx.x / bands, y0 = (blockIdx.x % bands) * NC_T;
    const int l16 = threadIdx.x & 15, pg = threadIdx.x >> 4;
    const int c = cin0 + 4 * l16;
    float w[N][9][4];
    load_weights<N>(a, col0, l16, w);
    const f32x4 sc = *reinterpret_cast<const f32x4 *>(a.scale + c), sh = *reinterpret_cast<const f32x4 *>(a.shift + c);
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(a.mean + c), rs = *reinterpret_cast<const f32x4 *>(a.rstd + c);
    f32x4 cA, cB, cD, s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (PASS == 2) {
        const f32x4 sg = *reinterpret_cast<const f32x4 *>(a.sum_g + c), sgx = *reinterpret_cast<const f32x4 *>(a.sum_gx + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cA[j] = (a.gamma ? a.gamma[c + j] : 1.f) * rs[j];
            cB[j] = -cA[j] * rs[j] * (sgx[j] * a.inv_n);
            cD[j] = -cA[j] * (sg[j] * a.inv_n) - cB[j] * mu[j];
        }
    }
    for (int x0 = 0; x0 < a.W; x0 += NC_T) {
        for (int i = threadIdx.x; i < NC_HALO * NC_HALO * N; i += 256) {
            const int hp = i / N, n = i % N;
            const int gy = y0 - 1 + hp / NC_HALO, gx = x0 - 1 + hp % NC_HALO;
            float v = 0.f;
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = a.go[((int64_t)(b * a.H + gy) * a.W + gx) * a.NO + col0 + n];
            lds[hp * NC_MAXN + n] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = pg + 16 * i, py = p >> 3, px = p & 7;
            const int gy = y0 + py, gx = x0 + px;
            if (gy < a.H && gx < a.W) {
                const int64_t off = ((int64_t)(b * a.H + gy) * a.W + gx) * a.ldy + c;
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(a.x + off);
                f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float *gp = lds + ((py + 2 - t / 3) * NC_HALO + px + 2 - t % 3) * NC_MAXN;
#pragma unroll
                    for (int n = 0; n < N; ++n) {
                        const float gv = gp[n];
#pragma unroll
                        for (int j = 0; j < 4; ++j) g[j] += gv * w[n][t][j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = fmaf(xv[j], sc[j], sh[j]) > 0.f ? g[j] : 0.f;
                if (PASS == 1) {
                    s1 += g;
                    s2 += g * ((xv - mu) * rs);
                } else {
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(cA[j], g[j], fmaf(cB[j], xv[j], cD[j]));
                    *reinterpret_cast<f32x4 *>(a.out + off) = o;
                }
            }
        }
        __syncthreads();
    }
    if (PASS == 1) {
        // 16 pixel groups -> one value per channel: red[pg][2][64]
        *reinterpret_cast<f32x4 *>(red + (pg * 2 + 0) * NC_CB + 4 * l16) = s1;
        *reinterpret_cast<f32x4 *>(red + (pg * 2 + 1) * NC_CB + 4 * l16) = s2;
        __syncthreads();
        if (threadIdx.x < 2 * NC_CB) {
            const int which = threadIdx.x / NC_CB, cc = threadIdx.x % NC_CB;
            float v = 0.f;
            for (int q = 0; q < 16; ++q) v += red[(q * 2 + which) * NC_CB + cc];
            atomicAdd((which ? a.sum_gx : a.sum_g) + cin0 + cc, v);
        }
    }
}

template <int NMAX, int PASS>
__global__ __launch_bounds__(256) void k_nconv_dgrad_bn(const NconvArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[NC_HALO * NC_HALO * NC_MAXN];
    __shared__ __attribute__((aligned(16))) float red[16 * 2 * NC_CB];
    const int n = a.n_out[blockIdx.y];
    if (n == 1) nconv_dgrad_bn_body<1, PASS>(a, lds, red);
    else if (n == 2) nconv_dgrad_bn_body<2, PASS>(a, lds, red);
    else if (n == 3 || NMAX == 3) nconv_dgrad_bn_body<3, PASS>(a, lds, red);
    else nconv_dgrad_bn_body<NMAX, PASS>(a, lds, red);
}

// grad_w[col0 + n][c][t] += sum_p go[p][col0 + n] * y[p + d_t][cin0 + c]   (fp32 atomics combine the bands)
// ORDERED (rd_set_deterministic(1)): grid.x == 1, the workgroup walks every (sample, band) itself and the 16 pixel groups are
// combined in group order -- one contributor per output element, no atomics.
template <int N, bool ORDERED>
__device__ __forceinline__ void nconv_wgrad_body(const NconvArgs &a, float *lds, float *g_l, float *red) {
    const int br = blockIdx.y, col0 = a.col_off[br], cin0 = a.cin_off[br];
    const int bands = (a.H + NC_T - 1) / NC_T;
    const int l16 = threadIdx.x & 15, pg = threadIdx.x >> 4;
    float acc[N][9][4];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[n][t][j] = 0.f;
    const int vb0 = ORDERED ? 0 : (int)blockIdx.x, vb1 = ORDERED ? a.B * bands : (int)blockIdx.x + 1;
    for (int vb = vb0; vb < vb1; ++vb) {
        const int b = vb / bands, y0 = (vb % bands) * NC_T;
        f32x4 pre[NC_HP];
        halo_fetch(a, b, y0, 0, cin0, pre);
        for (int x0 = 0; x0 < a.W; x0 += NC_T) {
            halo_store(pre, lds);
            for (int i = threadIdx.x; i < NC_T * NC_T * N; i += 256) {
                const int p = i / N, n = i % N;
                const int gy = y0 + (p >> 3), gx = x0 + (p & 7);
                g_l[p * NC_MAXN + n] = (gy < a.H && gx < a.W) ? a.go[((int64_t)(b * a.H + gy) * a.W + gx) * a.NO + col0 + n] : 0.f;
            }
            __syncthreads();
            if (x0 + NC_T < a.W) halo_fetch(a, b, y0, x0 + NC_T, cin0, pre);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int p = pg + 16 * i, py = p >> 3, px = p & 7;
                float g[N];
#pragma unroll
                for (int n = 0; n < N; ++n) g[n] = g_l[p * NC_MAXN + n];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const f32x4 x = *reinterpret_cast<const f32x4 *>(lds + ((py + t / 3) * NC_HALO + px + t % 3) * NC_LDW + 4 * l16);
#pragma unroll
                    for (int n = 0; n < N; ++n)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[n][t][j] += g[n] * x[j];
                }
            }
            __syncthreads();
        }
    }
    if (ORDERED) {
        // stage one (n, tap) slice of the 16 pixel groups' accumulators at a time ([pg][64 channels], reusing the halo buffer) and
        // add them in group order
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) lds[pg * NC_CB + 4 * l16 + j] = acc[n][t][j];
                __syncthreads();
                if (threadIdx.x < NC_CB) {
                    float v = 0.f;
                    for (int q = 0; q < 16; ++q) v += lds[q * NC_CB + threadIdx.x];
                    a.out[(int64_t)col0 * NC_CB * 9 + (n * NC_CB + threadIdx.x) * 9 + t] = v;
                }
                __syncthreads();
            }
        return;
    }
    // Combine the 16 pixel groups, then one global atomic per element.  A wave holds 4 pixel groups (lanes 16 g + l16): they are summed
    // with two cross-lane shuffles per value, and only lanes 0..15 of each wave add to the LDS image ([n][c][t], the output layout) --
    // 4-way contention and 16x fewer LDS atomics than every lane adding its own value (PMC round 2: this kernel ran at 0.39 TB/s, 10x
    // its streaming time: 27.6 k same-address LDS atomics per workgroup, serialised 16 deep).
    for (int i = threadIdx.x; i < N * NC_CB * 9; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[n][t][j];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if ((threadIdx.x & 63) < 16) atomicAdd(&red[(n * NC_CB + 4 * l16 + j) * 9 + t], v);
            }
    __syncthreads();
    for (int i = threadIdx.x; i < N * NC_CB * 9; i += 256) {
        const float v = red[i];
        if (v != 0.f) atomicAdd(&a.out[(int64_t)col0 * NC_CB * 9 + i], v);
    }
}

template <int NMAX, bool ORDERED>
__global__ __launch_bounds__(256) void k_nconv_wgrad(const NconvArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[NC_HALO * NC_HALO * NC_LDW];
    __shared__ float g_l[NC_T * NC_T * NC_MAXN];
    __shared__ float red[NC_MAXN * NC_CB * 9];
    const int n = a.n_out[blockIdx.y];
    if (n == 1) nconv_wgrad_body<1, ORDERED>(a, lds, g_l, red);
    else if (n == 2) nconv_wgrad_body<2, ORDERED>(a, lds, g_l, red);
    else if (n == 3 || NMAX == 3) nconv_wgrad_body<3, ORDERED>(a, lds, g_l, red);
    else nconv_wgrad_body<NMAX, ORDERED>(a, lds, g_l, red);
}

static int fill_args(NconvArgs &a, const char *who, int B, int H, int W, int ldy, int NO, int NB, const int32_t *cin_off, const int32_t *col_off,
                     const int32_t *n_out) {
    RD_REQUIRE(B > 0 && H > 0 && W > 0, "%s: bad geometry", who);
    RD_REQUIRE(NB >= 1 && NB <= NC_MAXB, "%s: %d branches outside 1..%d", who, NB, NC_MAXB);
    RD_REQUIRE(cin_off && col_off && n_out, "%s: branch tables are NULL", who);
    RD_REQUIRE(ldy % 4 == 0, "%s: ldy=%d must be a multiple of 4", who, ldy);
    for (int b = 0; b < NB; ++b) {
        RD_REQUIRE(n_out[b] >= 1 && n_out[b] <= NC_MAXN, "%s: branch %d has %d outputs (1..%d)", who, b, n_out[b], NC_MAXN);
        RD_REQUIRE(col_off[b] >= 0 && col_off[b] + n_out[b] <= NO, "%s: branch %d columns outside [0,%d)", who, b, NO);
        RD_REQUIRE(cin_off[b] >= 0 && cin_off[b] % 4 == 0 && cin_off[b] + NC_CB <= ldy, "%s: branch %d channels outside the row", who, b);
        a.cin_off[b] = cin_off[b];
        a.col_off[b] = col_off[b];
        a.n_out[b] = n_out[b];
    }
    a.ldy = ldy; a.NO = NO; a.NB = NB; a.B = B; a.H = H; a.W = W;
    return RD_OK;
}

static int max_width(const NconvArgs &a) {
    int m = 1;
    for (int b = 0; b < a.NB; ++b) m = std::max(m, a.n_out[b]);
    return m;
}

extern "C" int rd_nconv_fwd(const float *y, int ldy, const float *weight, const float *bias, int B, int H, int W, int NO, int NB,
                            const int32_t *cin_off, const int32_t *col_off, const int32_t *n_out, float *out, void *stream) {
    NconvArgs a{};
    int rc = fill_args(a, "rd_nconv_fwd", B, H, W, ldy, NO, NB, cin_off, col_off, n_out);
    if (rc) return rc;
    a.y = y; a.w = weight; a.bias = bias; a.out = out;
    dim3 grid((unsigned)(B * cdiv(H, NC_T)), (unsigned)NB);
    if (max_width(a) <= 3) k_nconv_fwd<3><<<grid, 256, 0, S(stream)>>>(a);
    else k_nconv_fwd<4><<<grid, 256, 0, S(stream)>>>(a);
    return check_launch("rd_nconv_fwd");
}

extern "C" int rd_nconv_dgrad(const float *grad_out, const float *weight, int B, int H, int W, int NO, int NB, const int32_t *cin_off,
                              const int32_t *col_off, const int32_t *n_out, float *grad_y, int ldy, void *stream) {
    NconvArgs a{};
    int rc = fill_args(a, "rd_nconv_dgrad", B, H, W, ldy, NO, NB, cin_off, col_off, n_out);
    if (rc) return rc;
    a.go = grad_out; a.w = weight; a.out = grad_y;
    dim3 grid((unsigned)(B * cdiv(H, NC_T)), (unsigned)NB);
    if (max_width(a) <= 3) k_nconv_dgrad<3><<<grid, 256, 0, S(stream)>>>(a);
    else k_nconv_dgrad<4><<<grid, 256, 0, S(stream)>>>(a);
    return check_launch("rd_nconv_dgrad");
}

extern "C" int rd_nconv_dgrad_bn(const float *grad_out, const float *weight, const float *x, const float *gamma, const float *mean,
                                 const float *rstd, const float *scale, const float *shift, int B, int H, int W, int NO, int NB,
                                 const int32_t *cin_off, const int32_t *col_off, const int32_t *n_out, float *grad_x, int ldy,
                                 float *grad_gamma, float *grad_beta, void *stream) {
    NconvArgs a{};
    int rc = fill_args(a, "rd_nconv_dgrad_bn", B, H, W, ldy, NO, NB, cin_off, col_off, n_out);
    if (rc) return rc;
    RD_REQUIRE(grad_out && weight && x && mean && rstd && scale && shift && grad_x && grad_gamma && grad_beta, "rd_nconv_dgrad_bn: null pointer");
    RD_REQUIRE(!g_deterministic, "rd_nconv_dgrad_bn: the fused form combines its partial sums with atomics; use rd_nconv_dgrad + rd_bn_bwd");
    a.go = grad_out; a.w = weight; a.out = grad_x; a.x = x; a.gamma = gamma; a.mean = mean; a.rstd = rstd; a.scale = scale; a.shift = shift;
    a.sum_g = grad_beta; a.sum_gx = grad_gamma;          // = the BatchNorm's parameter gradients (zero-filled by the caller)
    a.inv_n = 1.0f / (float)((int64_t)B * H * W);
    dim3 grid((unsigned)(B * cdiv(H, NC_T)), (unsigned)NB);
    hipStream_t st = S(stream);
    if (max_width(a) <= 3) {
        k_nconv_dgrad_bn<3, 1><<<grid, 256, 0, st>>>(a);
        k_nconv_dgrad_bn<3, 2><<<grid, 256, 0, st>>>(a);
    } else {
        k_nconv_dgrad_bn<4, 1><<<grid, 256, 0, st>>>(a);
        k_nconv_dgrad_bn<4, 2><<<grid, 256, 0, st>>>(a);
    }
    return check_launch("rd_nconv_dgrad_bn");
}

extern "C" int rd_nconv_wgrad(const float *y, int ldy, const float *grad_out, int B, int H, int W, int NO, int NB, const int32_t *cin_off,
                              const int32_t *col_off, const int32_t *n_out, float *grad_w, void *stream) {
    NconvArgs a{};
    int rc = fill_args(a, "rd_nconv_wgrad", B, H, W, ldy, NO, NB, cin_off, col_off, n_out);
    if (rc) return rc;
    a.y = y; a.go = grad_out; a.out = grad_w;
    if (g_deterministic) {
        dim3 grid1(1, (unsigned)NB);
        if (max_width(a) <= 3) k_nconv_wgrad<3, true><<<grid1, 256, 0, S(stream)>>>(a);
        else k_nconv_wgrad<4, true><<<grid1, 256, 0, S(stream)>>>(a);
        return check_launch("rd_nconv_wgrad");
    }
    dim3 grid((unsigned)(B * cdiv(H, NC_T)), (unsigned)NB);
    if (max_width(a) <= 3) k_nconv_wgrad<3, false><<<grid, 256, 0, S(stream)>>>(a);
    else k_nconv_wgrad<4, false><<<grid, 256, 0, S(stream)>>>(a);
    return check_launch("rd_nconv_wgrad");
}
