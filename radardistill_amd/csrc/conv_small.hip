// Sparse 3x3 convolution of the first (32-channel) SparseEnc stage in bf16x3 arithmetic: one WAVEFRONT per 32 output rows, no LDS.
//
// Why: the student's first stage has ~15 k active pillars (2 k radar points per sample) and 32 channels -- 0.28 GFLOP and ~5 MB per
// layer.  On the tiled implicit-GEMM kernel (128-row x 32-channel tiles, K steps of one tap, double-buffered LDS) such a launch
// lasted 54 us: 121 workgroups, each walking nine dependent gather -> LDS -> barrier -> MFMA steps; the chain of latencies, not
// the work, set the time (PMC: 14 % MFMA busy, 41 % of wave time parked).  Here a row's whole neighbourhood is in flight at once:
// lane (m, h) of a wave owns output row m (0..31) and k-half h; it reads the nine neighbour indices of its row, then for every tap
// the 32-byte piece [16 ks + 8 h, +8) of the gathered input row and of weight row n = m -- which ARE the 32x32x16 MFMA operand
// fragments (A[row][k], B[k][n] with k contiguous per lane), so nothing goes through LDS -- splits both into bf16 hi / lo in
// registers (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate, as conv_b3.hip) and runs 6 MFMAs per tap.  Same epilogue
// contract as k_conv_igemm (bias, statistics, folded BatchNorm scale / shift, residual, ReLU).  The data gradient is the same
// launch with the mirrored neighbour table (ix.flip) and the [Cin][tap][Cout] weights.
#include "conv_common.hpp"

using namespace rd;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8 &hi, bf16x8 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h0 = (__bf16)a[e], h1 = (__bf16)b[e];
        hi[e] = h0;
        hi[4 + e] = h1;
        lo[e] = (__bf16)(a[e] - (float)h0);
        lo[4 + e] = (__bf16)(b[e] - (float)h1);
    }
}

constexpr int CS_C = 32, CS_TAPS = 9;

__global__ __launch_bounds__(256) void k_conv_small32_b3(const ConvArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (blockIdx.x * 4 + wave) * 32;
    if (r0 >= a.out_rows) return;          // wave-uniform; the kernel has no workgroup-level synchronisation
    const int m = lane & 31, kh = lane >> 5;
    const int j = r0 + m;
    const bool row_ok = j < a.out_rows;
    int src[CS_TAPS];
    {
        const int32_t *nb = a.ix.nbr + (int64_t)(row_ok ? j : a.out_rows - 1) * CS_TAPS;
#pragma unroll
        for (int t = 0; t < CS_TAPS; ++t) src[t] = row_ok ? nb[a.ix.flip ? CS_TAPS - 1 - t : t] : -1;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *wbase = a.w + (int64_t)m * CS_TAPS * CS_C + 8 * kh;          // weight row n = m, [n][tap][c]
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        f32x4 ra[3][4], rb[3][4];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = 3 * g + u;
            const float *arow = a.in + (int64_t)max(src[t], 0) * CS_C + 8 * kh;
            const float *wrow = wbase + t * CS_C;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ra[u][2 * ks] = *reinterpret_cast<const f32x4 *>(arow + 16 * ks);
                ra[u][2 * ks + 1] = *reinterpret_cast<const f32x4 *>(arow + 16 * ks + 4);
                rb[u][2 * ks] = *reinterpret_cast<const f32x4 *>(wrow + 16 * ks);
                rb[u][2 * ks + 1] = *reinterpret_cast<const f32x4 *>(wrow + 16 * ks + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int t = 3 * g + u;
            if (__ballot(src[t] >= 0) == 0) continue;          // no row of this tile has that neighbour
            const bool has = src[t] >= 0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f32x4 a0 = ra[u][2 * ks], a1 = ra[u][2 * ks + 1];
                if (!has) {
                    a0 = f32x4{0.f, 0.f, 0.f, 0.f};
                    a1 = a0;
                }
                bf16x8 ah, al, bh, bl;
                split8(a0, a1, ah, al);
                split8(rb[u][2 * ks], rb[u][2 * ks + 1], bh, bl);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
        }
    }
    // ---- epilogue: acc[r] = out[row r0 + 8 (r >> 2) + (r & 3) + 4 kh][column m]
    const float bias = a.bias ? a.bias[m] : 0.f;
    const float sc = a.scale ? a.scale[m] : 1.f, sh = a.shift ? a.shift[m] : 0.f;
    float csum = 0.f, csq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < a.out_rows) {
            float v = acc[r] + bias;
            csum += v;
            csq += v * v;
            v = fmaf(v, sc, sh);
            if (a.residual) v += a.residual[(int64_t)row * CS_C + m];
            if (a.relu) v = fmaxf(v, 0.f);
            a.out[(int64_t)row * CS_C + m] = v;
        }
    }
    if (a.stats) {
        csum += __shfl_xor(csum, 32, 64);
        csq += __shfl_xor(csq, 32, 64);
        if (kh == 0) {
            atomicAdd(&a.stats[m], csum);
            atomicAdd(&a.stats[CS_C + m], csq);
        }
    }
}

// Dense 3x3 convolution (any stride) with <= 32 output channels over many input channels -- the CMA blocks' 27-channel DCNv2
// offset / mask convolution (256 -> 27, stride 2): 4.8 GFLOP at most, but on a 128-row x 32-column tile of the implicit-GEMM kernel
// it is 256 workgroups that each walk 72 dependent (tap, 32-channel chunk) steps with nothing else resident to hide a step's
// latency: 267 us (18 TF/s) for the 8 x 128 x 128 map.  Same idea as above: one wavefront per 32 output pixels, operand fragments
// straight from global memory (a lane reads 64 contiguous bytes of its pixel's input row and of weight row n = m), and the
// contraction SPLIT over the channel chunks (slices), so 8 x as many waves are resident and every wave only has nine
// independent gathers in flight.  Slices combine with fp32 atomics into a zero-filled output (slice 0 adds the bias): the result
// is order-dependent in the last bits, so the deterministic mode keeps the tiled kernel.  Plain epilogue only (bias).
__global__ __launch_bounds__(256) void k_conv_narrow_b3(const ConvArgs a, const int n_tiles, const int n_slices, const int chunks_per_slice) {
    int tile, slice;
    if (!xcd_tile(n_tiles, n_slices, tile, slice)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (tile * 4 + wave) * 32;
    if (r0 >= a.out_rows) return;          // wave-uniform; no workgroup-level synchronisation below
    const int m = lane & 31, kh = lane >> 5;
    const int j = r0 + m;
    const bool row_ok = j < a.out_rows;
    const rd_conv_index &ix = a.ix;
    int src[CS_TAPS];
    {
        const int jj = row_ok ? j : a.out_rows - 1;
        const int ox = jj % ix.Wout, oy = (jj / ix.Wout) % ix.Hout, b = jj / (ix.Wout * ix.Hout);
#pragma unroll
        for (int t = 0; t < CS_TAPS; ++t) src[t] = row_ok ? src_row_dense_k(ix, b, oy, ox, t / 3, t % 3) : -1;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int n = min(m, a.Cout - 1);          // columns past Cout multiply a valid weight row and are never stored
    const int c_begin = slice * chunks_per_slice * CS_C, c_end = min(a.Cin, c_begin + chunks_per_slice * CS_C);
    for (int c = c_begin; c < c_end; c += CS_C) {
        const float *wbase = a.w + (int64_t)n * CS_TAPS * a.Cin + c + 8 * kh;          // [n][tap][Cin]
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            f32x4 ra[3][4], rb[3][4];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int t = 3 * g + u;
                const float *arow = a.in + (int64_t)max(src[t], 0) * a.Cin + c + 8 * kh;
                const float *wrow = wbase + (int64_t)t * a.Cin;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    ra[u][2 * ks] = *reinterpret_cast<const f32x4 *>(arow + 16 * ks);
                    ra[u][2 * ks + 1] = *reinterpret_cast<const f32x4 *>(arow + 16 * ks + 4);
                    rb[u][2 * ks] = *reinterpret_cast<const f32x4 *>(wrow + 16 * ks);
                    rb[u][2 * ks + 1] = *reinterpret_cast<const f32x4 *>(wrow + 16 * ks + 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int t = 3 * g + u;
                if (__ballot(src[t] >= 0) == 0) continue;          // no pixel of this tile has that tap inside the map
                const bool has = src[t] >= 0;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f32x4 a0 = ra[u][2 * ks], a1 = ra[u][2 * ks + 1];
                    if (!has) {
                        a0 = f32x4{0.f, 0.f, 0.f, 0.f};
                        a1 = a0;
                    }
                    bf16x8 ah, al, bh, bl;
                    split8(a0, a1, ah, al);
                    split8(rb[u][2 * ks], rb[u][2 * ks + 1], bh, bl);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue: acc[r] = out[row r0 + 8 (r >> 2) + (r & 3) + 4 kh][column m]
    if (m >= a.Cout) return;
    const float bias = (a.bias && slice == 0) ? a.bias[m] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < a.out_rows) atomicAdd(&a.out[(int64_t)row * a.Cout + m], acc[r] + bias);
    }
}

}  // namespace

// true when the split-K wavefront kernel applies: bf16x3 mode, dense forward geometry, 3x3, <= 32 output channels, >= 64 input
// channels (a multiple of 32), fp32 operands, bias-only epilogue, atomics allowed.  Zero-fills `out` (the slices accumulate into it).
bool launch_conv_narrow_b3(const ConvArgs &a, hipStream_t st) {
    if (a.ix.mode != 1 || a.taps != CS_TAPS || a.ix.KH != 3 || a.ix.KW != 3 || a.Cout > 32 || a.Cin < 64 || a.Cin % CS_C || a.in_split || a.w_split ||
        a.in_rows <= 0 || a.scale || a.shift || a.residual || a.relu || a.stats || g_deterministic)
        return false;
    static const bool off = getenv("RD_CONV_NARROW") && getenv("RD_CONV_NARROW")[0] == '0';
    if (off) return false;
    const int chunks = a.Cin / CS_C;
    // slices: enough waves to fill the chip (~4096) and no more -- measured alone, 256 -> 27 stride 2: 32768 output pixels 81 / 83 / 94 us
    // with 1 / 4 / 8 slices (97 us on the tiled kernel), 8192 pixels 75 / 31 / 32 us (90 us tiled)
    static const int force = getenv("RD_NARROW_SLICES") ? atoi(getenv("RD_NARROW_SLICES")) : 0;
    const int64_t want = force > 0 ? force : cdiv(4096, cdiv(a.out_rows, 32));
    const int slices = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(chunks, 8), want));
    const int cps = (int)cdiv(chunks, slices), n_slices = (int)cdiv(chunks, cps);
    const int n_tiles = (int)cdiv(a.out_rows, 128);
    if (hipMemsetAsync(a.out, 0, (size_t)a.out_rows * a.Cout * 4, st) != hipSuccess) return false;
    k_conv_narrow_b3<<<xcd_grid(n_tiles, n_slices), 256, 0, st>>>(a, n_tiles, n_slices, cps);
    return true;
}

// true when the wavefront-per-tile kernel applies (bf16x3 mode, neighbour-table geometry, 9 taps, 32 -> 32 channels, fp32 operands)
bool launch_conv_small_b3(const ConvArgs &a, hipStream_t st) {
    if (a.ix.mode != 0 || a.taps != CS_TAPS || a.Cin != CS_C || a.Cout != CS_C || a.in_split || a.w_split || a.in_rows <= 0) return false;
    static const bool off = getenv("RD_CONV_SMALL") && getenv("RD_CONV_SMALL")[0] == '0';
    if (off) return false;
    k_conv_small32_b3<<<(unsigned)cdiv(a.out_rows, 128), 256, 0, st>>>(a);
    return true;
}
