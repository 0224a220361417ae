// Hard (capacity-limited) voxeliser and the PillarVFE of the padded-voxel input format.  See include/rdamd.h section N.
//
// Reference: DataProcessor.transform_points_to_voxels -> VoxelGeneratorWrapper -> spconv's Point2VoxelCPU3d
// (pcdet/datasets/processor/data_processor.py:16-61,142-229; third-party, runs on the CPU per sample), PillarVFE / PFNLayer
// (pcdet/models/backbones_3d/vfe/pillar_vfe.py:8-123).  spconv's published algorithm, per sample, in point-stream order:
//   c = floor((p - range_min) / voxel_size) per axis (fp32); drop the point when any c is outside the grid;
//   a voxel is created by its first point -- unless max_voxels voxels already exist, then the point (and every later point of
//   that voxel) is dropped; a point is appended to its voxel while the voxel holds fewer than max_points points.
//   Outputs: voxels (M, max_points, C) zero padded, coordinates (M, 3) as (z, y, x), num_points_per_voxel (M).
// Voxel order = order of first appearance, point order inside a voxel = stream order: both are reproduced exactly.
//
// GPU formulation (no sort, no hash): one workgroup per sample finds, round by round, the r-th smallest point index of every
// cell with an atomicMin on a per-cell word (round 0 = the voxel's creator); the creators are numbered by a block-wide prefix
// count in index order; a second kernel writes the rows.  Rounds stop as soon as one assigns nothing.
#include <limits.h>
#include <algorithm>
#include "common.hpp"

using namespace rd;

struct HvGeom {
    float x0, y0, z0, vx, vy, vz;
    int gx, gy, gz;
};

constexpr int HV_BLOCK = 1024;

// start[b] = first point index whose batch id is >= b (points are sorted by batch id, as collate_batch concatenates them)
__global__ void k_hv_bounds(const float *__restrict__ points, int n, int stride, int batch, int32_t *start) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > batch) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)points[(int64_t)mid * stride] < b) lo = mid + 1;
        else hi = mid;
    }
    start[b] = lo;
}

__global__ __launch_bounds__(HV_BLOCK) void k_hv_sample(const float *__restrict__ points, int stride, const int32_t *__restrict__ start, HvGeom g,
                                                        int max_points, int max_voxels, int32_t *cell_of, int32_t *slot, int32_t *nth,
                                                        int32_t *vox_of_cell, int32_t *counts) {
    __shared__ int s_any;
    __shared__ int s_wave[HV_BLOCK / 64];
    __shared__ int s_carry;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int p0 = start[b], p1 = start[b + 1];
    const int64_t cells_per_sample = (int64_t)g.gz * g.gy * g.gx;
    for (int i = p0 + tid; i < p1; i += HV_BLOCK) {
        const float *p = points + (int64_t)i * stride;
        const int cx = (int)floorf((p[1] - g.x0) / g.vx), cy = (int)floorf((p[2] - g.y0) / g.vy), cz = (int)floorf((p[3] - g.z0) / g.vz);
        const bool ok = cx >= 0 && cx < g.gx && cy >= 0 && cy < g.gy && cz >= 0 && cz < g.gz;
        cell_of[i] = ok ? (int32_t)(b * cells_per_sample + ((int64_t)cz * g.gy + cy) * g.gx + cx) : -1;
        slot[i] = -1;
    }
    __syncthreads();
    for (int r = 0; r < max_points; ++r) {
        if (tid == 0) s_any = 0;
        for (int i = p0 + tid; i < p1; i += HV_BLOCK) {
            const int c = cell_of[i];
            if (c >= 0 && slot[i] < 0) atomicMin(&nth[c], i);
        }
        __syncthreads();
        int any = 0;
        for (int i = p0 + tid; i < p1; i += HV_BLOCK) {
            const int c = cell_of[i];
            // nth[] is updated by L2 atomics: read / re-arm it with agent-scope atomic accesses too (a plain load could be served
            // by a stale line of this CU's L1)
            if (c >= 0 && slot[i] < 0 && __hip_atomic_load(&nth[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == i) {   // unique winner
                slot[i] = r;
                __hip_atomic_store(&nth[c], INT_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next round
                any = 1;
            }
        }
        if (any) s_any = 1;
        __syncthreads();
        if (!s_any) break;                                 // block-uniform: every later round would be empty too
        __syncthreads();                                   // s_any is re-written at the top of the next round
    }
    // number the voxel creators (slot 0) in point order; creators past max_voxels lose their voxel
    if (tid == 0) s_carry = 0;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int base = p0; base < p1; base += HV_BLOCK) {
        const int i = base + tid;
        const bool first = i < p1 && cell_of[i] >= 0 && slot[i] == 0;
        const unsigned long long bal = __ballot(first);
        const int in_wave = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (first) {
            const int ord = before + in_wave;
            vox_of_cell[cell_of[i]] = ord < max_voxels ? ord : -1;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < HV_BLOCK / 64; ++w) tot += s_wave[w];
            s_carry += tot;
        }
        __syncthreads();
    }
    if (tid == 0) counts[b] = min(s_carry, max_voxels);
}

__global__ void k_hv_write(const float *__restrict__ points, int n, int stride, int n_feat, int batch, HvGeom g, int max_points,
                           const int32_t *__restrict__ cell_of, const int32_t *__restrict__ slot, const int32_t *__restrict__ vox_of_cell,
                           const int32_t *__restrict__ counts, int64_t max_rows, float *voxels, int32_t *coords, int32_t *num_points,
                           int32_t *n_voxels) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        int tot = 0;
        for (int b = 0; b < batch; ++b) tot += counts[b];
        *n_voxels = tot;
    }
    if (i >= n) return;
    const int c = cell_of[i], s = slot[i];
    if (c < 0 || s < 0) return;
    const int v = vox_of_cell[c];
    if (v < 0) return;
    const int64_t cells_per_sample = (int64_t)g.gz * g.gy * g.gx;
    const int b = (int)(c / cells_per_sample);
    int base = 0;
    for (int q = 0; q < b; ++q) base += counts[q];
    const int64_t row = (int64_t)base + v;
    if (row >= max_rows) return;
    const float *p = points + (int64_t)i * stride;
    float *dst = voxels + (row * max_points + s) * n_feat;
    for (int k = 0; k < n_feat; ++k) dst[k] = p[1 + k];
    atomicAdd(&num_points[row], 1);
    if (s == 0) {
        const int64_t r = c - b * cells_per_sample;
        coords[row * 4 + 0] = b;
        coords[row * 4 + 1] = (int)(r / ((int64_t)g.gy * g.gx));
        coords[row * 4 + 2] = (int)((r / g.gx) % g.gy);
        coords[row * 4 + 3] = (int)(r % g.gx);
    }
}

extern "C" int64_t rd_voxelize_hard_ws_bytes(int n_points, int batch, int gx, int gy, int gz) {
    const int64_t cells = (int64_t)batch * gz * gy * gx;
    return ((int64_t)2 * n_points + 2 * cells + 2 * (batch + 2)) * 4;
}

extern "C" int rd_voxelize_hard(const float *points, int n_points, int n_feat, int batch, int gx, int gy, int gz, float x0, float y0, float z0,
                                float vx, float vy, float vz, int max_points, int max_voxels, int64_t max_rows, float *voxels,
                                int32_t *coords, int32_t *num_points, int32_t *n_voxels, void *ws, int64_t ws_bytes, void *stream) {
    RD_REQUIRE(n_points >= 0 && n_feat >= 3 && batch >= 1 && batch <= 1024, "rd_voxelize_hard: bad sizes");
    RD_REQUIRE(gx > 0 && gy > 0 && gz > 0 && (int64_t)batch * gx * gy * gz < INT_MAX, "rd_voxelize_hard: cell space must fit 31 bits");
    RD_REQUIRE(max_points >= 1 && max_voxels >= 1 && max_rows >= 0, "rd_voxelize_hard: bad capacities");
    RD_REQUIRE(ws_bytes >= rd_voxelize_hard_ws_bytes(n_points, batch, gx, gy, gz), "rd_voxelize_hard: workspace too small");
    hipStream_t st = S(stream);
    const int64_t cells = (int64_t)batch * gz * gy * gx;
    int32_t *cell_of = reinterpret_cast<int32_t *>(ws), *slot = cell_of + n_points, *nth = slot + n_points, *vox = nth + cells;
    int32_t *start = vox + cells, *counts = start + batch + 2;
    RD_HIP(hipMemsetAsync(nth, 0x7f, (size_t)cells * 4, st));               // 0x7f7f7f7f > any point index
    RD_HIP(hipMemsetAsync(voxels, 0, (size_t)max_rows * max_points * n_feat * 4, st));
    RD_HIP(hipMemsetAsync(num_points, 0, (size_t)max_rows * 4, st));
    RD_HIP(hipMemsetAsync(coords, 0, (size_t)max_rows * 16, st));
    HvGeom g{x0, y0, z0, vx, vy, vz, gx, gy, gz};
    k_hv_bounds<<<cdiv(batch + 1, 64), 64, 0, st>>>(points, n_points, 1 + n_feat, batch, start);
    k_hv_sample<<<batch, HV_BLOCK, 0, st>>>(points, 1 + n_feat, start, g, max_points, max_voxels, cell_of, slot, nth, vox, counts);
    k_hv_write<<<(unsigned)std::max<int64_t>(1, cdiv(n_points, 256)), 256, 0, st>>>(points, n_points, 1 + n_feat, n_feat, batch, g, max_points,
                                                                               cell_of, slot, vox, counts, max_rows, voxels, coords, num_points,
                                                                               n_voxels);
    return check_launch("rd_voxelize_hard");
}

// ---------------------------------------------------------------------------------------------- PillarVFE (one PFN layer)
// features of slot p of voxel v (pillar_vfe.py:91-110): [raw (x,y,z,extra..) or extra only] ++ (xyz - mean_xyz of the voxel) ++
// (xyz - voxel centre) (++ |xyz|), zeroed for padded slots.  PFNLayer (last layer): Linear(no bias) -> BatchNorm1d over ALL M*P
// slots (padded slots contribute exact zeros to the sums but count in n) -> ReLU -> max over the P slots (a padded slot
// contributes relu(shift), pillar_vfe.py:40-44).
constexpr int PV_MAX_IN = 16, PV_COUT = 64, PV_VOX = 4, PV_MAXP = 64;

struct PvArgs {
    const float *voxels;      // (M, P, C)
    const int32_t *num;       // (M)
    const int32_t *coords;    // (M, 4) b, z, y, x
    const float *w;           // (Cout, Cin)
    int M, P, C, Cin, Cout, use_abs, with_dist;
    float vx, vy, vz, xoff, yoff, zoff;
};

__device__ __forceinline__ void pv_stage(const PvArgs &a, int v0, float (*feat)[PV_MAXP][PV_MAX_IN], float (*mean)[4], float *w_l) {
    const int tid = threadIdx.x;
    if (w_l)
        for (int i = tid; i < a.Cout * a.Cin; i += 256) w_l[i] = a.w[i];
    if (tid < PV_VOX * 3) {
        const int vv = tid / 3, k = tid % 3, v = v0 + vv;
        float s = 0.f;
        if (v < a.M) {
            for (int p = 0; p < a.P; ++p) s += a.voxels[((int64_t)v * a.P + p) * a.C + k];   // padded slots are zero, as in the reference sum
            s /= (float)a.num[v];
        }
        mean[vv][k] = s;
    }
    __syncthreads();
    for (int i = tid; i < PV_VOX * a.P; i += 256) {
        const int vv = i / a.P, p = i % a.P, v = v0 + vv;
        float *f = feat[vv][p];
        const bool real = v < a.M && p < a.num[v];
        if (!real) {
            for (int k = 0; k < a.Cin; ++k) f[k] = 0.f;
            continue;
        }
        const float *pt = a.voxels + ((int64_t)v * a.P + p) * a.C;
        int o = 0;
        for (int k = (a.use_abs ? 0 : 3); k < a.C; ++k) f[o++] = pt[k];
        for (int k = 0; k < 3; ++k) f[o++] = pt[k] - mean[vv][k];
        const int32_t *c = a.coords + (int64_t)v * 4;
        f[o++] = pt[0] - ((float)c[3] * a.vx + a.xoff);
        f[o++] = pt[1] - ((float)c[2] * a.vy + a.yoff);
        f[o++] = pt[2] - ((float)c[1] * a.vz + a.zoff);
        if (a.with_dist) f[o++] = sqrtf(pt[0] * pt[0] + pt[1] * pt[1] + pt[2] * pt[2]);
    }
    __syncthreads();
}

// MODE 0: accumulate sum / sumsq of the linear outputs into stats[2*Cout] (atomics, caller zero-fills)
// MODE 1: out[v][c] = max_p relu(lin * scale[c] + shift[c])
template <int MODE>
__global__ __launch_bounds__(256) void k_pvfe(const PvArgs a, const float *__restrict__ scale, const float *__restrict__ shift, float *stats,
                                              float *out) {
    __shared__ float feat[PV_VOX][PV_MAXP][PV_MAX_IN];
    __shared__ float mean[PV_VOX][4];
    __shared__ float w_l[PV_COUT * PV_MAX_IN];
    __shared__ float red[2][PV_COUT];
    const int v0 = blockIdx.x * PV_VOX;
    pv_stage(a, v0, feat, mean, w_l);
    const int c = threadIdx.x & 63, vv = threadIdx.x >> 6, v = v0 + vv;
    if (MODE == 0 && threadIdx.x < 2 * PV_COUT) red[threadIdx.x >> 6][threadIdx.x & 63] = 0.f;
    if (MODE == 0) __syncthreads();
    float s1 = 0.f, s2 = 0.f, best = -INFINITY;
    if (v < a.M && c < a.Cout) {
        const int np = a.num[v];
        const float sc = MODE == 1 ? scale[c] : 0.f, sh = MODE == 1 ? shift[c] : 0.f;
        for (int p = 0; p < np && p < a.P; ++p) {
            float lin = 0.f;
            for (int k = 0; k < a.Cin; ++k) lin = fmaf(feat[vv][p][k], w_l[c * a.Cin + k], lin);
            if (MODE == 0) {
                s1 += lin;
                s2 += lin * lin;
            } else {
                best = fmaxf(best, fmaxf(fmaf(lin, sc, sh), 0.f));
            }
        }
        if (MODE == 1) {
            if (np < a.P) best = fmaxf(best, fmaxf(sh, 0.f));   // padded slots: linear output 0 -> relu(shift)
            out[(int64_t)v * a.Cout + c] = best;
        }
    }
    if (MODE == 0) {
        if (c < a.Cout) {
            atomicAdd(&red[0][c], s1);
            atomicAdd(&red[1][c], s2);
        }
        __syncthreads();
        if (threadIdx.x < 2 * PV_COUT && (threadIdx.x & 63) < a.Cout) {
            const int which = threadIdx.x >> 6, cc = threadIdx.x & 63;
            atomicAdd(&stats[which * a.Cout + cc], red[which][cc]);
        }
    }
}

static int pv_check(const PvArgs &a, const char *who) {
    RD_REQUIRE(a.M >= 0 && a.P >= 1 && a.P <= PV_MAXP, "%s: max points per voxel %d outside 1..%d", who, a.P, PV_MAXP);
    RD_REQUIRE(a.C >= 3 && a.Cin >= 1 && a.Cin <= PV_MAX_IN, "%s: %d input features outside 1..%d", who, a.Cin, PV_MAX_IN);
    RD_REQUIRE(a.Cout >= 1 && a.Cout <= PV_COUT, "%s: %d output channels outside 1..%d", who, a.Cout, PV_COUT);
    const int expect = (a.use_abs ? a.C : a.C - 3) + 6 + (a.with_dist ? 1 : 0);
    RD_REQUIRE(a.Cin == expect, "%s: Cin=%d but the feature assembly yields %d", who, a.Cin, expect);
    return RD_OK;
}

extern "C" int rd_pillar_vfe_stats(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P, int C, const float *weight,
                                   int Cin, int Cout, int use_abs_xyz, int with_distance, float vx, float vy, float vz, float xoff, float yoff,
                                   float zoff, float *stats, void *stream) {
    PvArgs a{voxels, num_points, coords, weight, M, P, C, Cin, Cout, use_abs_xyz, with_distance, vx, vy, vz, xoff, yoff, zoff};
    int rc = pv_check(a, "rd_pillar_vfe_stats");
    if (rc) return rc;
    if (M == 0) return RD_OK;
    k_pvfe<0><<<(unsigned)cdiv(M, PV_VOX), 256, 0, S(stream)>>>(a, nullptr, nullptr, stats, nullptr);
    return check_launch("rd_pillar_vfe_stats");
}

extern "C" int rd_pillar_vfe_max(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P, int C, const float *weight,
                                 int Cin, int Cout, int use_abs_xyz, int with_distance, float vx, float vy, float vz, float xoff, float yoff,
                                 float zoff, const float *scale, const float *shift, float *out, void *stream) {
    PvArgs a{voxels, num_points, coords, weight, M, P, C, Cin, Cout, use_abs_xyz, with_distance, vx, vy, vz, xoff, yoff, zoff};
    int rc = pv_check(a, "rd_pillar_vfe_max");
    if (rc) return rc;
    RD_REQUIRE(scale && shift && out, "rd_pillar_vfe_max: scale / shift / out are required");
    if (M == 0) return RD_OK;
    k_pvfe<1><<<(unsigned)cdiv(M, PV_VOX), 256, 0, S(stream)>>>(a, scale, shift, nullptr, out);
    return check_launch("rd_pillar_vfe_max");
}

// ---------------------------------------------------------------------------------------------- PillarVFE, general path
// Training (gradients), several PFN layers, USE_NORM False: the slot features are materialised once as rows (M*P, ld) (zero for
// padded slots and for the columns >= Cin that pad the row to the implicit-GEMM's K step), every PFNLayer then is Linear (1-tap
// implicit GEMM) -> BatchNorm over all M*P rows (norm.hip) -> ReLU -> this pooling kernel: max over the P slots of a voxel, and for
// a non-last layer the concatenation [x | max repeated over the slots] (pillar_vfe.py:40-49).
__global__ __launch_bounds__(256) void k_pv_decorate(const PvArgs a, int ld, float *__restrict__ out) {
    __shared__ float feat[PV_VOX][PV_MAXP][PV_MAX_IN];
    __shared__ float mean[PV_VOX][4];
    const int v0 = blockIdx.x * PV_VOX;
    pv_stage(a, v0, feat, mean, nullptr);
    const int per_vox = a.P * ld;
    for (int i = threadIdx.x; i < PV_VOX * per_vox; i += 256) {
        const int vv = i / per_vox, r = i % per_vox, p = r / ld, k = r % ld, v = v0 + vv;
        if (v < a.M) out[((int64_t)v * a.P + p) * ld + k] = k < a.Cin ? feat[vv][p][k] : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_pfn_pool_fwd(const float *__restrict__ x, int M, int P, int C, int last, float *__restrict__ out,
                                                      int32_t *__restrict__ argmax) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * C) return;
    const int m = (int)(idx / C), c = (int)(idx % C);
    const float *col = x + (int64_t)m * P * C + c;
    float best = col[0];
    int bi = 0;
    for (int p = 1; p < P; ++p) {
        const float v = col[(int64_t)p * C];
        if (v > best) {          // first maximum on ties
            best = v;
            bi = p;
        }
    }
    if (argmax) argmax[idx] = bi;
    if (last) {
        out[idx] = best;
        return;
    }
    float *o = out + (int64_t)m * P * 2 * C + c;
    for (int p = 0; p < P; ++p) {
        o[(int64_t)p * 2 * C] = col[(int64_t)p * C];
        o[(int64_t)p * 2 * C + C] = best;
    }
}

__global__ __launch_bounds__(256) void k_pfn_pool_bwd(const float *__restrict__ g, const int32_t *__restrict__ argmax, int M, int P, int C, int last,
                                                      float *__restrict__ gx) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * C) return;
    const int m = (int)(idx / C), c = (int)(idx % C);
    const int bi = argmax[idx];
    float *o = gx + (int64_t)m * P * C + c;
    if (last) {
        const float gm = g[idx];
        for (int p = 0; p < P; ++p) o[(int64_t)p * C] = p == bi ? gm : 0.f;
        return;
    }
    const float *gi = g + (int64_t)m * P * 2 * C + c;
    float gm = 0.f;
    for (int p = 0; p < P; ++p) gm += gi[(int64_t)p * 2 * C + C];
    for (int p = 0; p < P; ++p) o[(int64_t)p * C] = gi[(int64_t)p * 2 * C] + (p == bi ? gm : 0.f);
}

extern "C" int rd_pillar_decorate(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P, int C, int Cin,
                                  int use_abs_xyz, int with_distance, float vx, float vy, float vz, float xoff, float yoff, float zoff, int ld,
                                  float *out, void *stream) {
    PvArgs a{voxels, num_points, coords, nullptr, M, P, C, Cin, 1, use_abs_xyz, with_distance, vx, vy, vz, xoff, yoff, zoff};
    int rc = pv_check(a, "rd_pillar_decorate");
    if (rc) return rc;
    RD_REQUIRE(ld >= Cin && out, "rd_pillar_decorate: row stride %d < %d features", ld, Cin);
    if (M == 0) return RD_OK;
    k_pv_decorate<<<(unsigned)cdiv(M, PV_VOX), 256, 0, S(stream)>>>(a, ld, out);
    return check_launch("rd_pillar_decorate");
}

extern "C" int rd_pfn_pool_fwd(const float *x, int M, int P, int C, int last, float *out, int32_t *argmax, void *stream) {
    RD_REQUIRE(M >= 0 && P >= 1 && C >= 1 && out, "rd_pfn_pool_fwd: bad sizes");
    if (M == 0) return RD_OK;
    k_pfn_pool_fwd<<<(unsigned)cdiv((int64_t)M * C, 256), 256, 0, S(stream)>>>(x, M, P, C, last, out, argmax);
    return check_launch("rd_pfn_pool_fwd");
}

extern "C" int rd_pfn_pool_bwd(const float *grad_out, const int32_t *argmax, int M, int P, int C, int last, float *grad_x, void *stream) {
    RD_REQUIRE(M >= 0 && P >= 1 && C >= 1 && argmax && grad_x, "rd_pfn_pool_bwd: bad sizes");
    if (M == 0) return RD_OK;
    k_pfn_pool_bwd<<<(unsigned)cdiv((int64_t)M * C, 256), 256, 0, S(stream)>>>(grad_out, argmax, M, P, C, last, grad_x);
    return check_launch("rd_pfn_pool_bwd");
}
