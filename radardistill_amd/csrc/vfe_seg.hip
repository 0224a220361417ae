// Pillar VFE, segmented form: points are first GROUPED by pillar (counting sort on the pillar row the voxeliser assigned: integer
// counts -> exclusive scan -> fill), then ONE wavefront per pillar does everything the reference's ~25 ATen kernels + torch_scatter
// atomics do (pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:214-241 feature assembly, :14-46 PFNLayerV2): per-pillar mean of
// xyz, the 9 + C point features, Linear(9+C -> 32), BatchNorm (folded scale / shift), ReLU and the per-pillar max -- with wavefront
// shuffles for the reductions and NO floating-point atomics: the per-pillar results are written once, coalesced (128 bytes of
// features per pillar).  See include/rdamd.h section B.
//
// Why (round-1 PMC, profiles/round1_pmc_hbm_traffic_f32.json): the first version reduced with one 64-bit atomicMax per (point,
// channel) into a packed [P][32] u64 buffer that was memset, hammered and unpacked again: 121 MB of HBM traffic per LiDAR launch
// against 33 MB algorithmic.  Here the point buffer is read once per pass (28-byte rows gathered through the order array; the 6.7 MB
// buffer is L2-resident), offsets / order add 8 bytes per point, outputs are written once.
//
// Lane map of the pillar kernel (round 3): a 16-lane group per pillar, two channels per lane, four pillars per wavefront -- see
// k_vfe_seg.  arg-max rule as before: smallest point index wins ties.  Within a pillar the points are visited in the order the fill
// pass stored them; rd_set_deterministic(1) sorts every segment by point index first, which makes the mean's summation order (and so
// every bit downstream) reproducible.
#include <algorithm>
#include "common.hpp"

using namespace rd;

namespace {

constexpr int VS_OUT = 32, VS_MAX_IN = 16;

__global__ void k_seg_count(const int32_t *__restrict__ point_row, int n, int32_t *cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && point_row[i] >= 0) atomicAdd(&cnt[point_row[i]], 1);          // integer: exact and order-independent
}

// exclusive scan of cnt[0..P) -> off[0..P], three launches of 1024-element tiles
__global__ __launch_bounds__(256) void k_seg_scan_local(const int32_t *__restrict__ cnt, int32_t *off, int32_t *tile_sum, int P) {
    __shared__ int32_t ws[4];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    int32_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = base + k < P ? cnt[base + k] : 0;
        s += v[k];
    }
    int32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t t = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63) >= d) incl += t;
    }
    if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
    __syncthreads();
    int32_t wbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wbase += ws[w];
    int32_t run = wbase + incl - s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < P) off[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255) tile_sum[blockIdx.x] = wbase + incl;
}
__global__ void k_seg_scan_tiles(int32_t *tile_sum, int n_tiles, int32_t *total_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {          // <= a few hundred tiles
        int32_t run = 0;
        for (int t = 0; t < n_tiles; ++t) {
            const int32_t v = tile_sum[t];
            tile_sum[t] = run;
            run += v;
        }
        *total_out = run;
    }
}
__global__ void k_seg_scan_add(int32_t *off, const int32_t *__restrict__ tile_sum, int P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) off[i] += tile_sum[i >> 10];
}
__global__ void k_seg_fill(const int32_t *__restrict__ point_row, int n, const int32_t *__restrict__ off, int32_t *cursor, int32_t *order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = point_row[i];
    if (r < 0) return;
    order[off[r] + atomicAdd(&cursor[r], 1)] = i;
}
// rd_set_deterministic(1): ascending point index inside every segment (one lane per pillar, insertion sort: segments are short)
__global__ void k_seg_sort(const int32_t *__restrict__ off, int P, int32_t *order) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int b = off[p], e = off[p + 1];
    for (int i = b + 1; i < e; ++i) {
        const int32_t v = order[i];
        int j = i - 1;
        while (j >= b && order[j] > v) {
            order[j + 1] = order[j];
            --j;
        }
        order[j + 1] = v;
    }
}

struct SegArgs {
    const float *points;
    int n_feat;
    const int32_t *order, *off, *coords;
    const float *weight, *geom, *scale, *shift;
    int P;
    float *out;
    int32_t *argmax;
    float *acc;        // [P][4] sum x, y, z, count (the backward kernels read it)
    float *stats;      // [65] sum, sum of squares per channel, count (STATS pass)
};

// MODE 0: statistics of the Linear outputs (train-mode BatchNorm, pass 1); MODE 1: affine + ReLU + per-pillar max (+ arg-max, + acc)
//
// Round 3 lane map: FOUR pillars per wavefront.  A 16-lane group owns one pillar; a lane owns TWO output channels (2 (lane & 15), +1)
// and keeps their two weight rows in registers; the group walks its pillar's points one after the other (every lane of the group
// reads the point's words at one address: a broadcast transaction), so there is no cross-lane reduction per pillar at all -- the
// per-pillar mean is summed redundantly by the 16 lanes in segment order, the max is a running register.  The previous map (one
// wavefront per pillar, two points x 32 channels per iteration, butterfly for the mean) spent ~3 us of dependent-load latency
// (offsets -> order -> point row) and 18 shuffles per pillar for the ~1.5 points a LiDAR pillar holds: 78.8 us per LiDAR launch
// (185 k pillars) against a 4 us streaming time.  Outputs are still written once, 128 bytes per pillar, coalesced.
// The statistics pass (MODE 0) runs on at most 256 workgroups: every workgroup ends with 65 atomics on the same 65 addresses, and
// 2048 of them queued ~130 k same-address atomics at the memory side (111 us for 16 k radar points).
template <int NF, int MODE>
__global__ __launch_bounds__(256) void k_vfe_seg(const SegArgs a) {
    __shared__ float red[2][16][VS_OUT];
    constexpr int CIN = 9 + NF, STR = 1 + NF;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 4, l16 = lane & 15, c0 = l16 * 2;
    float w0[CIN], w1[CIN];
#pragma unroll
    for (int k = 0; k < CIN; ++k) {
        w0[k] = a.weight[c0 * CIN + k];
        w1[k] = a.weight[(c0 + 1) * CIN + k];
    }
    const float vx = a.geom[0], vy = a.geom[1], xoff = a.geom[3], yoff = a.geom[4], zoff = a.geom[5], x0 = a.geom[6], y0 = a.geom[7], z0 = a.geom[8];
    const float sc0 = MODE == 1 ? a.scale[c0] : 1.f, sh0 = MODE == 1 ? a.shift[c0] : 0.f;
    const float sc1 = MODE == 1 ? a.scale[c0 + 1] : 1.f, sh1 = MODE == 1 ? a.shift[c0 + 1] : 0.f;
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f, n_pts = 0.f;
    for (int p = blockIdx.x * 16 + grp; p < a.P; p += gridDim.x * 16) {
        const int beg = a.off[p], k = a.off[p + 1] - beg;
        // ---- the first four points of the pillar (all of them for 97 % of LiDAR pillars) are fetched ONCE, their index and field
        // loads issued together: three dependent load levels per pillar (offsets -> order -> rows) instead of two per point and pass
        int idx[4];
        float fq[4][NF];
#pragma unroll
        for (int u = 0; u < 4; ++u) idx[u] = u < k ? a.order[beg + u] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *q = a.points + (int64_t)idx[u] * STR + 1;
#pragma unroll
            for (int t = 0; t < NF; ++t) fq[u][t] = u < k ? q[t] : 0.f;
        }
        // ---- per-pillar sum of xyz, in segment order
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < k) { sx += fq[u][0]; sy += fq[u][1]; sz += fq[u][2]; }
        if (k > 4) {          // longer pillars (3 % of them, up to a few dozen points near the sensor): the group's 16 lanes fetch 16 points at once
            float tx = 0.f, ty = 0.f, tz = 0.f;
            for (int j = 4 + l16; j < k; j += 16) {
                const float *q = a.points + (int64_t)a.order[beg + j] * STR;
                tx += q[1]; ty += q[2]; tz += q[3];
            }
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) {
                tx += __shfl_xor(tx, d, 16);
                ty += __shfl_xor(ty, d, 16);
                tz += __shfl_xor(tz, d, 16);
            }
            sx += tx; sy += ty; sz += tz;
        }
        const float cntf = fmaxf((float)k, 1.f);
        const float mx = sx / cntf, my = sy / cntf, mz = sz / cntf;
        const int32_t *cd = a.coords + (int64_t)p * 3;          // (b, y, x)
        const float cxc = (float)cd[2] * vx + xoff, cyc = (float)cd[1] * vy + yoff;
        float best0 = 0.f, best1 = 0.f;
        int bi0 = 0x7fffffff, bi1 = 0x7fffffff;
        auto point = [&](const int i, const float (&r)[NF]) {
            float f[CIN];
            const float x = r[0], y = r[1], z = r[2];
            f[0] = x - cxc; f[1] = y - cyc; f[2] = z - zoff;
#pragma unroll
            for (int t = 0; t < NF; ++t) f[3 + t] = r[t];
            f[3 + NF] = x - mx; f[4 + NF] = y - my; f[5 + NF] = z - mz;
            f[6 + NF] = x - x0; f[7 + NF] = y - y0; f[8 + NF] = z - z0;
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int t = 0; t < CIN; ++t) {
                v0 = fmaf(f[t], w0[t], v0);
                v1 = fmaf(f[t], w1[t], v1);
            }
            if (MODE == 0) {
                s1a += v0; s2a += v0 * v0;
                s1b += v1; s2b += v1 * v1;
                if (c0 == 0) n_pts += 1.f;
            } else {
                v0 = fmaxf(fmaf(v0, sc0, sh0), 0.f);
                v1 = fmaxf(fmaf(v1, sc1, sh1), 0.f);
                if (v0 > best0 || (v0 == best0 && i < bi0)) { best0 = v0; bi0 = i; }
                if (v1 > best1 || (v1 == best1 && i < bi1)) { best1 = v1; bi1 = i; }
            }
        };
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < k) point(idx[u], fq[u]);
        for (int base = 4; base < k; base += 16) {          // each lane fetches one point, then the group visits them in segment order
            const int j = base + l16;
            int ti = 0;
            float tf[NF];
#pragma unroll
            for (int t = 0; t < NF; ++t) tf[t] = 0.f;
            if (j < k) {
                ti = a.order[beg + j];
                const float *q = a.points + (int64_t)ti * STR + 1;
#pragma unroll
                for (int t = 0; t < NF; ++t) tf[t] = q[t];
            }
            const int cnt = min(16, k - base);
            for (int s2 = 0; s2 < cnt; ++s2) {
                const int i = __shfl(ti, s2, 16);
                float r[NF];
#pragma unroll
                for (int t = 0; t < NF; ++t) r[t] = __shfl(tf[t], s2, 16);
                point(i, r);
            }
        }
        if (MODE == 1) {
            *reinterpret_cast<float2 *>(a.out + (int64_t)p * VS_OUT + c0) = make_float2(best0, best1);
            if (a.argmax) *reinterpret_cast<int2 *>(a.argmax + (int64_t)p * VS_OUT + c0) = make_int2(bi0, bi1);
            if (c0 == 0 && a.acc) *reinterpret_cast<float4 *>(a.acc + (int64_t)p * 4) = make_float4(sx, sy, sz, (float)k);
        }
    }
    if (MODE == 0) {
        red[0][grp][c0] = s1a; red[0][grp][c0 + 1] = s1b;
        red[1][grp][c0] = s2a; red[1][grp][c0 + 1] = s2b;
        // the point count: one value per 16-lane group (its c0 == 0 lane), combined over the workgroup in LDS slot order below
        __shared__ float cnt[16];
        if (c0 == 0) cnt[grp] = n_pts;
        __syncthreads();
        if (threadIdx.x < VS_OUT) {
            const int cc = threadIdx.x;
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                t1 += red[0][g][cc];
                t2 += red[1][g][cc];
            }
            atomicAdd(&a.stats[cc], t1);
            atomicAdd(&a.stats[VS_OUT + cc], t2);
        }
        if (threadIdx.x == 0) {
            float n = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) n += cnt[g];
            if (n != 0.f) atomicAdd(&a.stats[2 * VS_OUT], n);
        }
    }
}

#define VS_DISPATCH(NF, ...)                                    \
    switch (NF) {                                              \
        case 3: { constexpr int NFC = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int NFC = 4; __VA_ARGS__; } break; \
        case 5: { constexpr int NFC = 5; __VA_ARGS__; } break; \
        case 6: { constexpr int NFC = 6; __VA_ARGS__; } break; \
        case 7: { constexpr int NFC = 7; __VA_ARGS__; } break; \
        default: rd::set_error("unsupported n_feat %d (3..7)", NF); return RD_EINVAL; \
    }

}  // namespace

extern "C" int64_t rd_vfe_group_ws_bytes(int n_pillars) { return ((int64_t)n_pillars + cdiv(std::max(n_pillars, 1), 1024) + 8) * 4; }

// point_row (n_points) -> offsets (n_pillars + 1) and order (n_valid point indices grouped by pillar).  ws: rd_vfe_group_ws_bytes.
extern "C" int rd_vfe_group(const int32_t *point_row, int n_points, int n_pillars, int32_t *offsets, int32_t *order, int32_t *ws, int64_t ws_bytes,
                            void *stream) {
    RD_REQUIRE(n_points >= 0 && n_pillars >= 0 && ws_bytes >= rd_vfe_group_ws_bytes(n_pillars), "rd_vfe_group: bad sizes / workspace too small");
    hipStream_t st = S(stream);
    if (n_pillars == 0) return RD_OK;
    int32_t *cursor = ws, *tile_sum = ws + n_pillars;
    const int n_tiles = (int)cdiv(n_pillars, 1024);
    RD_HIP(hipMemsetAsync(ws, 0, (size_t)n_pillars * 4, st));                     // counts, later the fill cursors
    if (n_points > 0) k_seg_count<<<cdiv(n_points, 256), 256, 0, st>>>(point_row, n_points, cursor);
    k_seg_scan_local<<<n_tiles, 256, 0, st>>>(cursor, offsets, tile_sum, n_pillars);
    k_seg_scan_tiles<<<1, 64, 0, st>>>(tile_sum, n_tiles, offsets + n_pillars);
    k_seg_scan_add<<<cdiv(n_pillars, 256), 256, 0, st>>>(offsets, tile_sum, n_pillars);
    RD_HIP(hipMemsetAsync(cursor, 0, (size_t)n_pillars * 4, st));
    if (n_points > 0) k_seg_fill<<<cdiv(n_points, 256), 256, 0, st>>>(point_row, n_points, offsets, cursor, order);
    if (g_deterministic) k_seg_sort<<<cdiv(n_pillars, 256), 256, 0, st>>>(offsets, n_pillars, order);
    return check_launch("rd_vfe_group");
}

// stats[65] = per-channel sum / sum of squares of the Linear outputs over all grouped points, and their count
extern "C" int rd_vfe_seg_stats(const float *points, int n_feat, const int32_t *order, const int32_t *offsets, const int32_t *coords,
                                const float *weight, const float *geom, int n_pillars, float *stats, void *stream) {
    RD_REQUIRE(9 + n_feat <= VS_MAX_IN, "rd_vfe_seg_stats: 9 + n_feat = %d exceeds %d", 9 + n_feat, VS_MAX_IN);
    hipStream_t st = S(stream);
    RD_HIP(hipMemsetAsync(stats, 0, 65 * 4, st));
    if (n_pillars <= 0) return RD_OK;
    SegArgs a{points, n_feat, order, offsets, coords, weight, geom, nullptr, nullptr, n_pillars, nullptr, nullptr, nullptr, stats};
    const int blocks = g_deterministic ? 1 : (int)std::min<int64_t>(cdiv(n_pillars, 16), 256);
    VS_DISPATCH(n_feat, k_vfe_seg<NFC, 0><<<blocks, 256, 0, st>>>(a));
    return check_launch("rd_vfe_seg_stats");
}

// out (P, 32) = max over the pillar's points of relu(Linear * scale + shift); argmax (P, 32) point indices or NULL; acc (P, 4) or NULL
extern "C" int rd_vfe_seg_max(const float *points, int n_feat, const int32_t *order, const int32_t *offsets, const int32_t *coords,
                              const float *weight, const float *geom, const float *scale, const float *shift, int n_pillars, float *out,
                              int32_t *argmax, float *pillar_acc, void *stream) {
    RD_REQUIRE(9 + n_feat <= VS_MAX_IN, "rd_vfe_seg_max: 9 + n_feat = %d exceeds %d", 9 + n_feat, VS_MAX_IN);
    if (n_pillars <= 0) return RD_OK;
    SegArgs a{points, n_feat, order, offsets, coords, weight, geom, scale, shift, n_pillars, out, argmax, pillar_acc, nullptr};
    const int blocks = (int)std::min<int64_t>(cdiv(n_pillars, 16), 8192);
    VS_DISPATCH(n_feat, k_vfe_seg<NFC, 1><<<blocks, 256, 0, S(stream)>>>(a));
    return check_launch("rd_vfe_seg_max");
}
