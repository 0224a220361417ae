// Composite entry points: one C call per layer and direction for the training step's most frequent layer,
//     convolution (+ bias) -> train-mode BatchNorm (statistics from the conv epilogue) -> (+ residual) -> activation.
// Each is exactly the sequence of library calls the host made one by one (same kernels, same arguments, same order of launches,
// same streams) -- what disappears is the host work between them: the step is 44 such layers, and per layer and direction the Python
// wrappers, argument checks, stream switches and autograd-helper calls around 2 (forward) / 4 (backward) launches cost 40-80 us of
// host time against 15.7 ms for a whole host-bound step (tools/diag/host_timers.py, round 3).
//   forward : rd_conv_fwd[_split] (statistics in the epilogue)  ->  rd_bn_train_fwd
//   backward: rd_bn_bwd (reduction + apply)  ->  data gradient (rd_conv_fwd_split on the [Cin][taps][Cout] operand, or rd_conv_dgrad)
//             ->  fork to the weight-gradient stream  ->  rd_conv_wgrad there
// Optional HIP events (bench.py's roofline hooks) are recorded around the convolution launches on the stream they go to.
#include "common.hpp"

using namespace rd;

static int record(void *ev, void *stream) {
    if (!ev) return RD_OK;
    RD_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ev), reinterpret_cast<hipStream_t>(stream)));
    return RD_OK;
}

extern "C" int rd_conv_bn_act_fwd(const float *in, int in_rows, int Cin, const void *weight, int w_format, int taps, const float *bias,
                                  float *raw, int out_rows, int Cout, const rd_conv_index *idx, float *stats, const float *gamma,
                                  const float *beta, float eps, float momentum, float *running_mean, float *running_var,
                                  const float *residual, int act, float *y, float *side, void *ev0, void *ev1, void *stream) {
    RD_REQUIRE(raw && y && side && stats, "rd_conv_bn_act_fwd: raw, y, side (4 x Cout) and stats (2 x Cout, zero-filled) are required");
    RD_REQUIRE(w_format >= 0 && w_format <= 2, "rd_conv_bn_act_fwd: weight format %d (0 fp32 kernel layout, 1 split, 2 fragment-major split)", w_format);
    int rc = record(ev0, stream);
    if (rc) return rc;
    if (w_format == 0)
        rc = rd_conv_fwd(in, in_rows, Cin, reinterpret_cast<const float *>(weight), taps, bias, raw, out_rows, Cout, idx, nullptr, nullptr, nullptr, 0,
                         stats, stream);
    else
        rc = rd_conv_fwd_split(in, 0, in_rows, Cin, weight, w_format, taps, bias, raw, out_rows, Cout, idx, nullptr, nullptr, nullptr, 0, stats,
                               stream);
    if (rc) return rc;
    rc = record(ev1, stream);
    if (rc) return rc;
    return rd_bn_train_fwd(raw, out_rows, Cout, stats, gamma, beta, eps, momentum, running_mean, running_var, residual, act, y, side,
                           side + Cout, side + 2 * Cout, side + 3 * Cout, stream);
}

extern "C" int rd_conv_bn_act_bwd(const float *raw, const float *y, const float *grad_y, int out_rows, int Cout, const float *gamma,
                                  const float *side, int act, int has_residual, float *grad_raw, float *grad_res, float *grad_gamma_beta,
                                  const void *w_dgrad, int w_format, int taps, float *grad_in, int in_rows, int Cin,
                                  const rd_conv_index *bwd_idx, const float *in, const rd_conv_index *fwd_idx, float *grad_wk,
                                  void *ev_d0, void *ev_d1, void *ev_w0, void *ev_w1, void *main_stream, void *side_stream) {
    RD_REQUIRE(raw && grad_y && side && grad_raw && grad_gamma_beta, "rd_conv_bn_act_bwd: raw, grad_y, side, grad_raw, grad_gamma_beta are required");
    RD_REQUIRE(w_format >= 0 && w_format <= 2, "rd_conv_bn_act_bwd: weight format %d", w_format);
    // BatchNorm (+ activation) backward: grad_gamma_beta = [grad_gamma | grad_beta], zero-filled by the caller
    int rc = rd_bn_bwd(raw, y, grad_y, out_rows, Cout, gamma, side, side + Cout, side + 2 * Cout, side + 3 * Cout, act, has_residual, grad_raw,
                       grad_res, grad_gamma_beta, grad_gamma_beta + Cout, main_stream);
    if (rc) return rc;
    if (grad_in) {          // data gradient: main stream (the next backward node consumes it)
        RD_REQUIRE(w_dgrad && bwd_idx, "rd_conv_bn_act_bwd: the data gradient needs its weight operand and the backward index");
        rc = record(ev_d0, main_stream);
        if (rc) return rc;
        if (w_format == 0)          // exact fp32: forward kernel layout [Cout][taps][Cin], read transposed
            rc = rd_conv_dgrad(grad_raw, out_rows, Cout, reinterpret_cast<const float *>(w_dgrad), taps, grad_in, in_rows, Cin, bwd_idx, main_stream);
        else                        // bf16x3: [Cin][taps][Cout] split operand, a forward convolution over the backward index
            rc = rd_conv_fwd_split(grad_raw, 0, out_rows, Cout, w_dgrad, w_format, taps, nullptr, grad_in, in_rows, Cin, bwd_idx, nullptr, nullptr,
                                   nullptr, 0, nullptr, main_stream);
        if (rc) return rc;
        rc = record(ev_d1, main_stream);
        if (rc) return rc;
    }
    if (grad_wk) {          // weight gradient: feeds nobody until the optimizer -> side stream, after it has seen grad_raw
        RD_REQUIRE(in && fwd_idx, "rd_conv_bn_act_bwd: the weight gradient needs the layer input and the forward index");
        void *ws = side_stream ? side_stream : main_stream;
        if (side_stream) {
            rc = rd_stream_fork(main_stream, side_stream);
            if (rc) return rc;
        }
        rc = record(ev_w0, ws);
        if (rc) return rc;
        rc = rd_conv_wgrad(in, in_rows, Cin, grad_raw, out_rows, Cout, taps, fwd_idx, grad_wk, ws);
        if (rc) return rc;
        rc = record(ev_w1, ws);
        if (rc) return rc;
    }
    return RD_OK;
}

// ---------------------------------------------------------------------------------------------- measurement probe
// What a bare v_mfma_f32_32x32x16_bf16 loop sustains on THIS chip with random operands (MI355X_MICROARCH.md, "DVFS give-back": the clock
// an MFMA-dense loop holds on random data is well under the 2.4 GHz behind the 2.5 PF spec figure, so no kernel issues MFMAs at the
// spec rate).  Operands live in registers, `waves_per_simd` waves per SIMD on every CU, 8 independent accumulators per wave, no
// memory traffic in the loop.  bench.py times it with HIP events and quotes the dominant kernel's issued rate against it.
namespace {
typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));
typedef float f32x16p __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_probe_mfma_bf16(int iters, unsigned seed, float *out) {
    // pseudo-random bf16 operands in [-1, 1): a hash of (thread, element)
    auto rnd = [&](unsigned k) {
        unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u ^ (k * 40503u + seed);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        return (__bf16)((float)(h & 0xffff) / 32768.f - 1.f);
    };
    bf16x8p a[2], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (i < 2) a[i][e] = rnd(i * 8 + e);
            b[i][e] = rnd(100 + i * 8 + e);
        }
    f32x16p acc[2][4];          // 8 independent accumulators (128 registers: two waves per SIMD fit)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123456.789f) out[0] = s;          // keeps the loop alive; practically never true
}
}  // namespace

extern "C" int rd_probe_mfma_bf16(int iters, int waves_per_simd, float *out_dev, double *flops_out, void *stream) {
    RD_REQUIRE(iters > 0 && waves_per_simd >= 1 && waves_per_simd <= 2 && out_dev, "rd_probe_mfma_bf16: iters > 0, 1 or 2 waves per SIMD, a device word");
    int dev = 0, cus = 0;
    RD_HIP(hipGetDevice(&dev));
    RD_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int blocks = cus * waves_per_simd;          // 256 threads = 4 waves = one per SIMD
    k_probe_mfma_bf16<<<blocks, 256, 0, S(stream)>>>(iters, 12345u, out_dev);
    if (flops_out) *flops_out = (double)blocks * 4.0 * iters * 8.0 * (2.0 * 32 * 32 * 16);
    return check_launch("rd_probe_mfma_bf16");
}

// ---------------------------------------------------------------------------------------------- geometry prelude
// The index work of one branch of a training step (pcdet's DynamicPillarVFE voxelisation + the active-site sets and rulebooks of the
// sparse PillarNet encoder: spconv's SubMConv2d / SparseConv2d(k3, s2, p1) index pairs -- pillar_backbone.py / spconv_backbone_2d of
// the reference build them lazily layer by layer) as TWO calls around the step's single device->host read:
//   rd_geometry_begin : points -> rank grid + point rows, then the rank grid of every stride-2 level below it (static launch shapes),
//                       and the sizes the host needs to allocate the rest -- pillars, in-range points, rows per level -- gathered
//                       into one small device array;
//   rd_geometry_finish: with those sizes known: coordinates and the SubM table of every level, the strided table and its transpose
//                       between consecutive levels.
// Same kernels, same order as the separate entry points (results are bit-identical by construction); what goes away is ~40 wrapper
// calls, ~25 small allocations and a dozen torch scalar ops per step at the one place where the GPU waits for the host
// (tools/diag/stream_timeline.py, round 3: main queue idle for the first 3-4 ms of a step).
namespace {
struct GeometryLevels {          // kernel argument block: device pointers + count-word offsets of up to 8 levels
    const uint32_t *rg[8];
    int64_t count_word[8];
};
__global__ void k_geometry_scalars(const int32_t *point_row, int n_points, int n_levels, GeometryLevels lv, int32_t *scalars) {
    // scalars[0] = pillars, [1] = in-range points (atomic, zeroed by the caller), [2 + l] = rows of level l + 1
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = 0;
    for (int p = i; p < n_points; p += gridDim.x * blockDim.x) c += point_row[p] >= 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(scalars + 1, c);
    if (i < n_levels) scalars[i == 0 ? 0 : 1 + i] = (int32_t)lv.rg[i][lv.count_word[i]];
}
inline int down(int v) { return (v + 2 - 3) / 2 + 1; }
}  // namespace

extern "C" int rd_geometry_begin(const float *points, int n_points, int n_feat, int batch, int gx, int gy, float x0, float y0, float vx,
                                 float vy, uint32_t *rankgrid, int32_t *point_row, int n_down, uint32_t *const *rankgrid_down,
                                 int32_t *scalars_dev, void *stream) {
    RD_REQUIRE(n_down >= 0 && n_down <= 7 && (n_down == 0 || rankgrid_down) && scalars_dev, "rd_geometry_begin: 0..7 levels below the pillar grid, a scalars array");
    int rc = rd_voxelize(points, n_points, n_feat, batch, gx, gy, x0, y0, vx, vy, rankgrid, point_row, stream);
    if (rc) return rc;
    GeometryLevels lv{};
    lv.rg[0] = rankgrid;
    // the pillar grid is x-major (rows in (b, cx, cy) key order) with H = gy, W = gx; levels below are y-major
    const uint32_t *src = rankgrid;
    int H = gy, W = gx, xmajor = 1;
    auto words = [](int64_t cells) { return (cells + 31) / 32; };
    lv.count_word[0] = 2 * words((int64_t)batch * H * W);
    for (int l = 0; l < n_down; ++l) {
        const int Ho = down(H), Wo = down(W);
        RD_REQUIRE(rankgrid_down[l], "rd_geometry_begin: rank grid of level %d is NULL", l + 1);
        rc = rd_rankgrid_downsample_grid(src, batch, H, W, xmajor, Ho, Wo, rankgrid_down[l], stream);
        if (rc) return rc;
        lv.rg[l + 1] = rankgrid_down[l];
        lv.count_word[l + 1] = 2 * words((int64_t)batch * Ho * Wo);
        src = rankgrid_down[l], H = Ho, W = Wo, xmajor = 0;
    }
    hipStream_t st = S(stream);
    RD_HIP(hipMemsetAsync(scalars_dev, 0, (2 + n_down) * 4, st));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv((int64_t)n_points, 256 * 8), 512));
    k_geometry_scalars<<<blocks, 256, 0, st>>>(point_row, n_points, 1 + n_down, lv, scalars_dev);
    return check_launch("rd_geometry_begin");
}

extern "C" int rd_geometry_finish(const uint32_t *const *rankgrid, int batch, int gx, int gy, int n_down, const int32_t *rows,
                                  int32_t *const *coords, int32_t *const *nbr_subm, int32_t *const *nbr_down, int32_t *const *nbr_up,
                                  void *stream) {
    RD_REQUIRE(n_down >= 0 && n_down <= 7 && rankgrid && rows && coords && nbr_subm, "rd_geometry_finish: level arrays are required");
    int Hs[8], Ws[8];
    Hs[0] = gy, Ws[0] = gx;
    for (int l = 1; l <= n_down; ++l) Hs[l] = down(Hs[l - 1]), Ws[l] = down(Ws[l - 1]);
    int rc;
    for (int l = 0; l <= n_down; ++l) {
        RD_REQUIRE(rows[l] >= 0 && (rows[l] == 0 || (coords[l] && nbr_subm[l])), "rd_geometry_finish: level %d has %d rows but no coords / table", l, rows[l]);
        rc = rd_rankgrid_coords(rankgrid[l], batch, Hs[l], Ws[l], l == 0, coords[l], rows[l], stream);
        if (rc) return rc;
        rc = rd_nbr_subm(coords[l], rows[l], rankgrid[l], batch, Hs[l], Ws[l], l == 0, nbr_subm[l], stream);
        if (rc) return rc;
    }
    for (int l = 0; l < n_down; ++l) {
        RD_REQUIRE(nbr_down && (rows[l + 1] == 0 || nbr_down[l]), "rd_geometry_finish: strided table of level %d is NULL", l);
        rc = rd_nbr_strided(coords[l + 1], rows[l + 1], rankgrid[l], batch, Hs[l], Ws[l], l == 0, nbr_down[l], stream);
        if (rc) return rc;
        if (nbr_up && nbr_up[l]) {
            rc = rd_nbr_strided_T(coords[l], rows[l], rankgrid[l + 1], batch, Hs[l + 1], Ws[l + 1], nbr_up[l], stream);
            if (rc) return rc;
        }
    }
    return RD_OK;
}
