// Implicit-GEMM convolution on the bf16 matrix cores with fp32-class accuracy ("bf16x3").
//
// Same contract, index modes and epilogue as k_conv_igemm in conv.hip, but each fp32 operand is split while it is staged
// into LDS, x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and every product is formed as
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (three v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// The dropped a_lo*b_lo term and the 16-bit operand representation bound the relative error of a dot product at ~4e-6
// (measured: 4.1e-6 at K = 2304 vs 4.5e-7 for exact fp32; 2.4e-3 for plain bf16; the reference's own GPU path runs cuDNN /
// spconv in TF32, ~1e-3).  One 32x32x16 bf16 MFMA issues in 32 cycles vs 8 x 64 cycles for the same K with
// v_mfma_f32_32x32x2_f32: 3 passes still cost 5.3x fewer matrix-pipe cycles than exact fp32.
//
// Tile: BM x BN output, K step 32 input channels of one tap (two k16 MFMA steps), 4 waves (2 x 2).  LDS image per operand and
// part: [rows][40] bf16 (80-byte rows: a 16-lane ds_read_b128 group covers all 64 banks once).  Lane l reads A[row l&31]
// [k = 8*(l>>5) + 0..7] as one 16-byte fragment (the 32x32x16 bf16 operand map).
#include "conv_common.hpp"

using namespace rd;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int KB3 = 32;
constexpr int LDB = 40;  // bf16 elements per LDS row (32 data + 8 pad)

__device__ __forceinline__ void split4(const f32x4 v, bf16x4 &hi, bf16x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)v[e];
        hi[e] = h;
        lo[e] = (__bf16)(v[e] - (float)h);
    }
}

template <int BM, int BN, bool DEFORM>
__global__ __launch_bounds__(256, 2) void k_conv_igemm_b3(const ConvArgs a) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int AP = BM / 32, BP = BN / 32;
    static_assert(MI >= 1 && NI >= 1, "wave tile at least 32x32");
    // per buffer: [A hi BM rows][A lo][B hi BN rows][B lo], 80 bytes per row
    constexpr int BUF = 2 * (BM + BN) * LDB;  // bf16 elements per buffer
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int row_tile, col_tile;
    if (!xcd_tile((a.out_rows + BM - 1) / BM, (a.Cout + BN - 1) / BN, row_tile, col_tile)) return;
    const int m0 = row_tile * BM, n0 = col_tile * BN;
    const int ld_r = tid >> 3, ld_c = (tid & 7) * 4;

    // ---- taps with a source row in this tile (the LDS array is free before the main loop: word 0 is the mask)
    int *s_mask = reinterpret_cast<int *>(lds);
    if (tid == 0) *s_mask = 0;
    __syncthreads();
    {
        int mask = 0;
        for (int p = 0; p < AP; ++p) {
            const int j = m0 + ld_r + 32 * p;
            if ((tid & 7) == 0)
                for (int t = 0; t < a.taps; ++t)
                    if (src_row(a, j, t) >= 0) mask |= 1 << t;
        }
        if (mask) atomicOr(s_mask, mask);
    }
    __syncthreads();
    const int tapmask = *s_mask;
    __syncthreads();
    const int kchunks = a.Cin / KB3;
    const int steps = __popc(tapmask) * kchunks;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[AP], rb[BP];
    int rows[AP];
    int4 sidx[DEFORM ? AP : 1];
    f32x4 sw[DEFORM ? AP : 1];
    int cur_tap = -1, tap_iter_mask = tapmask;

    auto load_tile = [&](int s) {
        const int kc = (s % kchunks) * KB3;
        if (s % kchunks == 0) {
            cur_tap = __ffs(tap_iter_mask) - 1;
            tap_iter_mask &= tap_iter_mask - 1;
            if constexpr (DEFORM) {
#pragma unroll
                for (int p = 0; p < AP; ++p) {
                    const int j = m0 + ld_r + 32 * p;
                    if (j < a.out_rows) {
                        const int64_t o = ((int64_t)j * a.taps + cur_tap) * 4;
                        sidx[p] = *reinterpret_cast<const int4 *>(a.ix.samp_idx + o);
                        sw[p] = *reinterpret_cast<const f32x4 *>(a.ix.samp_w + o);
                    } else {
                        sidx[p] = make_int4(-1, -1, -1, -1);
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < AP; ++p) rows[p] = src_row(a, m0 + ld_r + 32 * p, cur_tap);
            }
        }
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (DEFORM) {
                const float *base = a.in + kc + ld_c;
                if (sidx[p].x >= 0) v += sw[p][0] * *reinterpret_cast<const f32x4 *>(base + (int64_t)sidx[p].x * a.Cin);
                if (sidx[p].y >= 0) v += sw[p][1] * *reinterpret_cast<const f32x4 *>(base + (int64_t)sidx[p].y * a.Cin);
                if (sidx[p].z >= 0) v += sw[p][2] * *reinterpret_cast<const f32x4 *>(base + (int64_t)sidx[p].z * a.Cin);
                if (sidx[p].w >= 0) v += sw[p][3] * *reinterpret_cast<const f32x4 *>(base + (int64_t)sidx[p].w * a.Cin);
            } else {
                if (rows[p] >= 0) v = *reinterpret_cast<const f32x4 *>(a.in + (int64_t)rows[p] * a.Cin + kc + ld_c);
            }
            ra[p] = v;
        }
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            const int n = n0 + ld_r + 32 * p;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < a.Cout) v = *reinterpret_cast<const f32x4 *>(a.w + ((int64_t)n * a.taps + cur_tap) * a.Cin + kc + ld_c);
            rb[p] = v;
        }
    };
    auto store_tile = [&](int buf) {
        __bf16 *Ah = lds + buf * BUF, *Al = Ah + BM * LDB, *Bh = Al + BM * LDB, *Bl = Bh + BN * LDB;
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            bf16x4 hi, lo;
            split4(ra[p], hi, lo);
            *reinterpret_cast<bf16x4 *>(Ah + (ld_r + 32 * p) * LDB + ld_c) = hi;
            *reinterpret_cast<bf16x4 *>(Al + (ld_r + 32 * p) * LDB + ld_c) = lo;
        }
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            bf16x4 hi, lo;
            split4(rb[p], hi, lo);
            *reinterpret_cast<bf16x4 *>(Bh + (ld_r + 32 * p) * LDB + ld_c) = hi;
            *reinterpret_cast<bf16x4 *>(Bl + (ld_r + 32 * p) * LDB + ld_c) = lo;
        }
    };

    if (steps > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) load_tile(s + 1);
        const __bf16 *Ah = lds + buf * BUF + (wm * WM + fr) * LDB + 8 * fh;
        const __bf16 *Al = Ah + BM * LDB;
        const __bf16 *Bh = lds + buf * BUF + 2 * BM * LDB + (wn * WN + fr) * LDB + 8 * fh;
        const __bf16 *Bl = Bh + BN * LDB;
#pragma unroll
        for (int ks = 0; ks < KB3 / 16; ++ks) {
            bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8 *>(Ah + i * 32 * LDB + ks * 16);
                al[i] = *reinterpret_cast<const bf16x8 *>(Al + i * 32 * LDB + ks * 16);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8 *>(Bh + j * 32 * LDB + ks * 16);
                bl[j] = *reinterpret_cast<const bf16x8 *>(Bl + j * 32 * LDB + ks * 16);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    // small terms first, then the dominant one
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (s + 1 < steps) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue (identical to the fp32 kernel)
    float *red = reinterpret_cast<float *>(lds);
    if (a.stats) {
        for (int i = tid; i < 2 * BN; i += 256) red[i] = 0.f;
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * WN + j * 32 + fr;
        const bool col_ok = col < a.Cout;
        const float bias = (a.bias && col_ok) ? a.bias[col] : 0.f;
        const float sc = (a.scale && col_ok) ? a.scale[col] : 1.f;
        const float sh = (a.shift && col_ok) ? a.shift[col] : 0.f;
        float csum = 0.f, csq = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row < a.out_rows && col_ok) {
                    float v = acc[i][j][r] + bias;
                    csum += v;
                    csq += v * v;
                    v = fmaf(v, sc, sh);
                    if (a.residual) v += a.residual[(int64_t)row * a.Cout + col];
                    if (a.relu) v = fmaxf(v, 0.f);
                    a.out[(int64_t)row * a.Cout + col] = v;
                }
            }
        }
        if (a.stats && col_ok) {
            atomicAdd(&red[wn * WN + j * 32 + fr], csum);
            atomicAdd(&red[BN + wn * WN + j * 32 + fr], csq);
        }
    }
    if (a.stats) {
        __syncthreads();
        for (int i = tid; i < BN; i += 256) {
            const int col = n0 + i;
            if (col < a.Cout) {
                atomicAdd(&a.stats[col], red[i]);
                atomicAdd(&a.stats[a.Cout + col], red[BN + i]);
            }
        }
    }
}

// Launch for Cout > 32 (narrower outputs stay on the exact-fp32 kernel: they are bandwidth-bound level-1 sparse convs).
int launch_conv_b3(const ConvArgs &a, int mode, hipStream_t st) {
    const int64_t big_blocks = cdiv(a.out_rows, 128) * cdiv(a.Cout, 128);
    dim3 block(256);
    if (mode == 3) {
        if (big_blocks >= 384) k_conv_igemm_b3<128, 128, true><<<dim3(xcd_grid(cdiv(a.out_rows, 128), cdiv(a.Cout, 128))), block, 0, st>>>(a);
        else k_conv_igemm_b3<64, 64, true><<<dim3(xcd_grid(cdiv(a.out_rows, 64), cdiv(a.Cout, 64))), block, 0, st>>>(a);
    } else {
        if (big_blocks >= 384) k_conv_igemm_b3<128, 128, false><<<dim3(xcd_grid(cdiv(a.out_rows, 128), cdiv(a.Cout, 128))), block, 0, st>>>(a);
        else k_conv_igemm_b3<64, 64, false><<<dim3(xcd_grid(cdiv(a.out_rows, 64), cdiv(a.Cout, 64))), block, 0, st>>>(a);
    }
    return RD_OK;
}
