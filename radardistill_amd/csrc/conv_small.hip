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

}  // namespace

// true when the wavefront-per-tile kernel applies (bf16x3 mode, neighbour-table geometry, 9 taps, 32 -> 32 channels, fp32 operands)
bool launch_conv_small_b3(const ConvArgs &a, hipStream_t st) {
    if (a.ix.mode != 0 || a.taps != CS_TAPS || a.Cin != CS_C || a.Cout != CS_C || a.in_split || a.w_split || a.in_rows <= 0) return false;
    static const bool off = getenv("RD_CONV_SMALL") && getenv("RD_CONV_SMALL")[0] == '0';
    if (off) return false;
    k_conv_small32_b3<<<(unsigned)cdiv(a.out_rows, 128), 256, 0, st>>>(a);
    return true;
}
