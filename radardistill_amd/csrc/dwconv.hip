// Depthwise KxK convolution (ConvNeXt `dwconv`, 7x7, padding 3, groups = C) on channels-last maps: forward, data gradient
// and weight gradient.  Replaces cuDNN's depthwise conv in pcdet/ops/basicblock/modules/Basicblock_convn.py:13,47 (MIOpen
// falls back to naive / batched-GEMM kernels for this shape: 16 ms of a 116 ms step before this file existed).
// HBM/L2-bound: each lane owns 4 channels of one pixel (float4); the 49 neighbour rows are contiguous 1 KiB reads that hit
// L2, weights are pre-transposed to [tap][C] so a wave reads them coalesced.
#include <stdlib.h>
#include "common.hpp"

using namespace rd;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[p][c] = bias[c] + sum_t w[t][c] * in[p + off(t)][c];  FLIP reads tap (T-1-t) of w for tap t (data gradient).
__global__ __launch_bounds__(256) void k_dwconv(const float *__restrict__ in, const float *__restrict__ wt /*[T][C]*/, const float *__restrict__ bias,
                                                int B, int H, int W, int C, int K, int flip, float *__restrict__ out) {
    const int c4n = C / 4;
    const int64_t total = (int64_t)B * H * W * c4n;
    const int pad = K / 2, T = K * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const int64_t p = i / c4n;
        const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
        f32x4 acc = bias ? *reinterpret_cast<const f32x4 *>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < K; ++ky) {
            const int yy = y + ky - pad;
            if (yy < 0 || yy >= H) continue;
            for (int kx = 0; kx < K; ++kx) {
                const int xx = x + kx - pad;
                if (xx < 0 || xx >= W) continue;
                const int t = ky * K + kx;
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(wt + (int64_t)(flip ? T - 1 - t : t) * C + c);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (((int64_t)b * H + yy) * W + xx) * C + c);
                acc += wv * v;
            }
        }
        *reinterpret_cast<f32x4 *>(out + p * C + c) = acc;
    }
}

// 7x7, LDS-tiled: the plain kernel above fetches 49 neighbour rows and 49 weight rows per output float4 -- 49x the map through the
// L1/L2 path (411 MB for the 8192 x 256 ConvNeXt map: ~40 us per launch, six times the 16.8 MB of HBM traffic).  Here a workgroup
// stages the 22 x 22 halo of a 16 x 16 pixel tile for 32 channels (and the 49 x 32 weights) in LDS once -- 1.9x the map from L2 --
// and a thread computes a strip of 8 consecutive output pixels of one row for 4 channels: per kernel row it reads the strip's 14
// inputs from LDS once and uses each for up to 7 outputs (18 LDS reads per output float4 instead of 98 global ones).
constexpr int DW_K = 7, DW_T = 16, DW_HALO = DW_T + DW_K - 1, DW_CB = 32, DW_PX = DW_CB + 4;   // pixel stride 36 floats: strips 8 px apart sit 32 banks apart

__global__ __launch_bounds__(256, 2) void k_dwconv7_tiled(const float *__restrict__ in, const float *__restrict__ wt /*[49][C]*/,
                                                         const float *__restrict__ bias, int B, int H, int W, int C, int flip,
                                                         float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float s_in[DW_HALO * DW_HALO * DW_PX];
    __shared__ __attribute__((aligned(16))) float s_w[DW_K * DW_K * DW_CB];
    const int tiles_x = (W + DW_T - 1) / DW_T, tiles_y = (H + DW_T - 1) / DW_T;
    const int tile = blockIdx.x, c0 = blockIdx.y * DW_CB;
    const int b = tile / (tiles_x * tiles_y), y0 = ((tile / tiles_x) % tiles_y) * DW_T, x0 = (tile % tiles_x) * DW_T;
    const int tid = threadIdx.x;
    for (int i = tid; i < DW_K * DW_K * (DW_CB / 4); i += 256) {
        const int t = i / (DW_CB / 4), q = (i % (DW_CB / 4)) * 4;
        *reinterpret_cast<f32x4 *>(s_w + t * DW_CB + q) = *reinterpret_cast<const f32x4 *>(wt + (int64_t)(flip ? DW_K * DW_K - 1 - t : t) * C + c0 + q);
    }
    for (int i = tid; i < DW_HALO * DW_HALO * (DW_CB / 4); i += 256) {
        const int hp = i / (DW_CB / 4), q = (i % (DW_CB / 4)) * 4;
        const int gy = y0 - DW_K / 2 + hp / DW_HALO, gx = x0 - DW_K / 2 + hp % DW_HALO;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const f32x4 *>(in + (((int64_t)b * H + gy) * W + gx) * C + c0 + q);
        *reinterpret_cast<f32x4 *>(s_in + hp * DW_PX + q) = v;
    }
    __syncthreads();
    const int q = (tid & 7) * 4, strip = tid >> 3, sy = strip >> 1, sx = (strip & 1) * 8;
    f32x4 acc[8];
    const f32x4 bv = bias ? *reinterpret_cast<const f32x4 *>(bias + c0 + q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = bv;
#pragma unroll 1          // one kernel row at a time: unrolled, the compiler hoists all 98 row reads and spills
    for (int ky = 0; ky < DW_K; ++ky) {
        f32x4 row[8 + DW_K - 1];
        const float *src = s_in + ((sy + ky) * DW_HALO + sx) * DW_PX + q;
#pragma unroll
        for (int j = 0; j < 8 + DW_K - 1; ++j) row[j] = *reinterpret_cast<const f32x4 *>(src + j * DW_PX);
#pragma unroll
        for (int kx = 0; kx < DW_K; ++kx) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(s_w + (ky * DW_K + kx) * DW_CB + q);
#pragma unroll
            for (int o = 0; o < 8; ++o) acc[o] += wv * row[o + kx];
        }
    }
    const int gy = y0 + sy;
    if (gy < H) {
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int gx = x0 + sx + o;
            if (gx < W) *reinterpret_cast<f32x4 *>(out + (((int64_t)b * H + gy) * W + gx) * C + c0 + q) = acc[o];
        }
    }
}

// dw[t][c] = sum_p go[p][c] * in[p + off(t)][c].  grid = (pixel chunks, T); block = 256 = 4 pixel lanes x 64 channel quads
// (C <= 256 per pass; loops over channel groups for wider maps).  Per-block partial -> partial[chunk][t][c]; summed by
// k_dw_wsum in fixed order.
__global__ __launch_bounds__(256) void k_dwconv_wgrad(const float *__restrict__ in, const float *__restrict__ go, int B, int H, int W, int C, int K,
                                                      int pix_per_chunk, float *__restrict__ partial) {
    __shared__ float red[4][256];
    const int t = blockIdx.y, ky = t / K, kx = t % K, pad = K / 2;
    const int64_t n_pix = (int64_t)B * H * W;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_chunk, p1 = min(n_pix, p0 + pix_per_chunk);
    const int lane_c = (threadIdx.x & 63) * 4, sub = threadIdx.x >> 6;
    for (int cg = 0; cg < C; cg += 256) {
        const int c = cg + lane_c;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            for (int64_t p = p0 + sub; p < p1; p += 4) {
                const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
                const int yy = y + ky - pad, xx = x + kx - pad;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const f32x4 g = *reinterpret_cast<const f32x4 *>(go + p * C + c);
                const f32x4 v = *reinterpret_cast<const f32x4 *>(in + (((int64_t)b * H + yy) * W + xx) * C + c);
                acc += g * v;
            }
        }
        *reinterpret_cast<f32x4 *>(&red[sub][lane_c]) = acc;
        __syncthreads();
        if (threadIdx.x < 256 && cg + (int)threadIdx.x < C) {
            const int cc = threadIdx.x;
            // pixel chunks combine with fp32 atomics into the zero-filled [tap][C] result: a wave adds 64 contiguous floats (the
            // separate chunk-sum pass this replaces took 86 us for 12544 outputs, longer than the products themselves)
            atomicAdd(&partial[(int64_t)t * C + cg + cc], red[0][cc] + red[1][cc] + red[2][cc] + red[3][cc]);
        }
        __syncthreads();
    }
}

// 7x7 weight gradient, LDS-tiled.  The plain kernel above is one workgroup per (pixel chunk, tap): both maps travel through L2 once per
// tap, 49 times (823 MB for the 8192 x 256 map, 70 us).  Here a workgroup stages the 14 x 22 input halo and the 8 x 16 gradient tile of
// 32 channels once and a thread owns ONE KERNEL ROW (7 taps) of 4 channels for a quarter of the tile's rows: walking a row it keeps a
// sliding window of 7 inputs in registers, so a pixel costs one gradient read, one new input read and 7 fma per channel.  The four row
// quarters are combined through LDS; workgroups (tiles of the same channels) combine with fp32 atomics into the zero-filled result,
// or -- deterministic mode -- one workgroup per channel chunk walks every tile and stores.
constexpr int DWG_TY = 8, DWG_TX = 16, DWG_HY = DWG_TY + DW_K - 1, DWG_HX = DWG_TX + DW_K - 1;

__global__ __launch_bounds__(256, 2) void k_dwconv7_wgrad_tiled(const float *__restrict__ in, const float *__restrict__ go, int B, int H, int W, int C,
                                                               int atomic, float *__restrict__ dw /*[49][C]*/) {
    __shared__ __attribute__((aligned(16))) float s_in[DWG_HY * DWG_HX * DW_PX];
    __shared__ __attribute__((aligned(16))) float s_go[DWG_TY * DWG_TX * DW_PX];
    const int tiles_x = (W + DWG_TX - 1) / DWG_TX, tiles_y = (H + DWG_TY - 1) / DWG_TY, n_tiles = B * tiles_x * tiles_y;
    const int c0 = blockIdx.y * DW_CB, tid = threadIdx.x;
    const int q = (tid & 7) * 4, r = tid >> 3, ky = r % DW_K, sub = r / DW_K;          // r >= 28: stages only
    f32x4 acc[DW_K];
#pragma unroll
    for (int k = 0; k < DW_K; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / (tiles_x * tiles_y), y0 = ((tile / tiles_x) % tiles_y) * DWG_TY, x0 = (tile % tiles_x) * DWG_TX;
        for (int i = tid; i < DWG_HY * DWG_HX * (DW_CB / 4); i += 256) {
            const int hp = i / (DW_CB / 4), qq = (i % (DW_CB / 4)) * 4;
            const int gy = y0 - DW_K / 2 + hp / DWG_HX, gx = x0 - DW_K / 2 + hp % DWG_HX;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const f32x4 *>(in + (((int64_t)b * H + gy) * W + gx) * C + c0 + qq);
            *reinterpret_cast<f32x4 *>(s_in + hp * DW_PX + qq) = v;
        }
        for (int i = tid; i < DWG_TY * DWG_TX * (DW_CB / 4); i += 256) {
            const int p = i / (DW_CB / 4), qq = (i % (DW_CB / 4)) * 4;
            const int gy = y0 + p / DWG_TX, gx = x0 + p % DWG_TX;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy < H && gx < W) v = *reinterpret_cast<const f32x4 *>(go + (((int64_t)b * H + gy) * W + gx) * C + c0 + qq);
            *reinterpret_cast<f32x4 *>(s_go + p * DW_PX + qq) = v;
        }
        __syncthreads();
        if (sub < 4) {
#pragma unroll 1
            for (int yy = 2 * sub; yy < 2 * sub + 2; ++yy) {
                const float *irow = s_in + ((yy + ky) * DWG_HX) * DW_PX + q;
                const float *grow = s_go + (yy * DWG_TX) * DW_PX + q;
                f32x4 w[DW_K];
#pragma unroll
                for (int k = 1; k < DW_K; ++k) w[k] = *reinterpret_cast<const f32x4 *>(irow + (k - 1) * DW_PX);
#pragma unroll
                for (int x = 0; x < DWG_TX; ++x) {
#pragma unroll
                    for (int k = 0; k < DW_K - 1; ++k) w[k] = w[k + 1];
                    w[DW_K - 1] = *reinterpret_cast<const f32x4 *>(irow + (x + DW_K - 1) * DW_PX);
                    const f32x4 g = *reinterpret_cast<const f32x4 *>(grow + x * DW_PX);
#pragma unroll
                    for (int k = 0; k < DW_K; ++k) acc[k] += g * w[k];
                }
            }
        }
        __syncthreads();
    }
    // combine the four row quarters: red[sub][ky][kx][32 channels] in the (now free) halo buffer
    float *red = s_in;
    if (sub < 4) {
#pragma unroll
        for (int k = 0; k < DW_K; ++k) *reinterpret_cast<f32x4 *>(red + ((sub * DW_K + ky) * DW_K + k) * DW_CB + q) = acc[k];
    }
    __syncthreads();
    for (int i = tid; i < DW_K * DW_K * DW_CB; i += 256) {
        const int t = i / DW_CB, c = i % DW_CB;
        const float v = red[(0 * DW_K * DW_K + t) * DW_CB + c] + red[(1 * DW_K * DW_K + t) * DW_CB + c] + red[(2 * DW_K * DW_K + t) * DW_CB + c] +
                        red[(3 * DW_K * DW_K + t) * DW_CB + c];
        if (atomic) atomicAdd(&dw[(int64_t)t * C + c0 + c], v);
        else dw[(int64_t)t * C + c0 + c] = v;
    }
}

extern "C" int rd_dwconv_fwd(const float *in, const float *weight_tc, const float *bias, int B, int H, int W, int C, int K, int flip, float *out,
                             void *stream) {
    RD_REQUIRE(C % 4 == 0 && K % 2 == 1 && K <= 11 && B > 0 && H > 0 && W > 0, "rd_dwconv_fwd: bad sizes");
    int64_t total = (int64_t)B * H * W * (C / 4);
    static const bool tiled_off = getenv("RD_DWCONV_TILED") && getenv("RD_DWCONV_TILED")[0] == '0';
    if (K == DW_K && C % DW_CB == 0 && !tiled_off) {
        // (summation order per output: kernel rows outer, columns inner, as in the plain kernel; taps outside the map add 0 * w)
        dim3 grid((unsigned)(B * cdiv(H, DW_T) * cdiv(W, DW_T)), (unsigned)(C / DW_CB));
        k_dwconv7_tiled<<<grid, 256, 0, S(stream)>>>(in, weight_tc, bias, B, H, W, C, flip, out);
        return check_launch("rd_dwconv_fwd(tiled)");
    }
    k_dwconv<<<(int)std::min<int64_t>(cdiv(total, 256), 8192), 256, 0, S(stream)>>>(in, weight_tc, bias, B, H, W, C, K, flip, out);
    return check_launch("rd_dwconv_fwd");
}

extern "C" int64_t rd_dwconv_wgrad_ws_bytes(int B, int H, int W, int C, int K) {
    (void)B; (void)H; (void)W; (void)C; (void)K;
    return 0;   // kept for ABI stability: the chunk partials are combined with atomics, no workspace
}

extern "C" int rd_dwconv_wgrad(const float *in, const float *grad_out, int B, int H, int W, int C, int K, float *grad_w_tc, float *ws,
                               int64_t ws_bytes, void *stream) {
    (void)ws; (void)ws_bytes;
    RD_REQUIRE(C % 4 == 0 && K % 2 == 1 && K <= 11, "rd_dwconv_wgrad: bad sizes");
    int64_t n_pix = (int64_t)B * H * W;
    int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(64, cdiv(n_pix, 64)));
    if (g_deterministic) chunks = 1;
    int ppc = (int)cdiv(n_pix, chunks);
    hipStream_t st = S(stream);
    static const bool tiled_off = getenv("RD_DWCONV_TILED") && getenv("RD_DWCONV_TILED")[0] == '0';
    if (K == DW_K && C % DW_CB == 0 && !tiled_off) {
        const int64_t n_tiles = (int64_t)B * cdiv(H, DWG_TY) * cdiv(W, DWG_TX);
        const int n_cc = C / DW_CB;
        // >= 512 workgroups when the map has the tiles for it; one per channel chunk (plain stores, fixed order) in deterministic mode
        const int64_t per_cc = g_deterministic ? 1 : std::max<int64_t>(1, std::min<int64_t>(n_tiles, cdiv(512, n_cc)));
        if (!g_deterministic) RD_HIP(hipMemsetAsync(grad_w_tc, 0, (size_t)K * K * C * 4, st));
        k_dwconv7_wgrad_tiled<<<dim3((unsigned)per_cc, (unsigned)n_cc), 256, 0, st>>>(in, grad_out, B, H, W, C, g_deterministic ? 0 : 1, grad_w_tc);
        return check_launch("rd_dwconv_wgrad(tiled)");
    }
    RD_HIP(hipMemsetAsync(grad_w_tc, 0, (size_t)K * K * C * 4, st));
    dim3 grid((unsigned)chunks, (unsigned)(K * K));
    k_dwconv_wgrad<<<grid, 256, 0, st>>>(in, grad_out, B, H, W, C, K, ppc, grad_w_tc);
    return check_launch("rd_dwconv_wgrad");
}
