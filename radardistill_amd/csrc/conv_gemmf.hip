// 1-tap convolutions (nn.Linear, 1x1 Conv2d, the DCNv2 column GEMM and their data gradients) as a plain GEMM, bf16x3 arithmetic,
// WEIGHT FRAGMENTS STRAIGHT FROM L2 (fragment-major split format, RD_LAYOUT_FRAG) -- the GEMM twin of conv_d3f.hip.
//
// out[M][N] = in[M][K] * W^T (+ epilogue), M = rows, K = Cin (a multiple of 64), N = Cout (a multiple of 32).
// The gathered implicit-GEMM kernel (k_conv_igemm_b3<64,64,..,dense>) ran these shapes at 90-100 TF/s (PMC round 2: MFMA busy 13 %,
// 14.5 VALU instructions per MFMA, 55 % of the wave time parked): K steps of 32 channels with a barrier each, index arithmetic of a
// general convolution per row, both operands staged through LDS by every column tile.  Here:
//   * activations: a BM x 64-channel tile per K chunk is fetched as fp32 rows, split to bf16 hi + lo ONCE and parked in LDS (144-byte
//     rows: 16-byte fragment reads of 16 consecutive rows are conflict-free), double-buffered -> one barrier per 64 channels;
//   * weights: B fragments are 16-byte-per-lane coalesced global loads from the fragment-major image, a ring of four register sets,
//     requested three k16 half-steps ahead (4 half-steps per chunk: the set of a half-step is static in the unrolled loop);
//   * schedule pinned with sched_barrier (see conv_d3f.hip).
// Same epilogue contract as k_conv_igemm_b3 (bias, BatchNorm statistics, scale / shift, residual, ReLU); products and their order per
// accumulator are those of the other bf16x3 kernels (lo*hi, hi*lo, hi*hi; K ascending).
#include <stdlib.h>
#include "conv_common.hpp"

using namespace rd;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int GK = 64;       // channels per K chunk (four k16 half-steps)
constexpr int GROW = 72;     // bf16 elements per LDS row: 64 data + 8 pad = 144 bytes

__device__ __forceinline__ void split4(const f32x4 v, bf16x4 &hi, bf16x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)v[e];
        hi[e] = h;
        lo[e] = (__bf16)(v[e] - (float)h);
    }
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void k_gemm_b3f(const ConvArgs a) {
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
    constexpr int PART = BM * GROW;                    // bf16 elements of one (hi or lo) tile image
    constexpr int AL = BM * (GK / 4) / 256;            // float4 activation loads per thread and chunk (BM rows x 16 pieces)
    static_assert(MI >= 1 && NI >= 1 && AL >= 1, "wave tile at least 32x32");
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * 2 * PART];          // [buffer][hi | lo][row][72]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int row_tile, col_tile;
    if (!xcd_tile((a.out_rows + BM - 1) / BM, (a.Cout + BN - 1) / BN, row_tile, col_tile)) return;
    const int m0 = row_tile * BM, n0 = col_tile * BN;
    const int fr = lane & 31, fh = lane >> 5;
    const int kchunks = a.Cin / GK, c16n = a.Cin >> 4;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int aoff[MI];          // A fragment base of this lane: tile row, channels 8 fh .. +7 (buffer 0, hi image)
#pragma unroll
    for (int i = 0; i < MI; ++i) aoff[i] = (wm * WM + i * 32 + fr) * GROW + fh * 8;
    const uint4 *wf = reinterpret_cast<const uint4 *>(a.w);
    int wbase[NI];         // 16-byte unit index of (32-column block, k16 block 0, hi) + lane; blocks past Cout read the last one (never stored)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int nb = min((n0 + wn * WN + j * 32) >> 5, (a.Cout >> 5) - 1);
        wbase[j] = nb * c16n * 128 + lane;
    }

    struct BSet {
        uint4 v[NI][2];
    };
    struct ASet {
        bf16x8 h[MI], l[MI];
    };
    BSet Bq[4];
    ASet Aq[2];
    auto load_B = [&](BSet &S, int k16) {          // absolute k16 block
        const int o = k16 * 128;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            S.v[j][0] = wf[wbase[j] + o];
            S.v[j][1] = wf[wbase[j] + o + 64];
        }
    };
    auto read_A = [&](ASet &A, const __bf16 *Abuf, const int h) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            A.h[i] = *reinterpret_cast<const bf16x8 *>(Abuf + aoff[i] + h * 16);
            A.l[i] = *reinterpret_cast<const bf16x8 *>(Abuf + aoff[i] + PART + h * 16);
        }
    };
    auto mfmas = [&](const ASet &A, const BSet &S) {
        bf16x8 bh[NI], bl[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            bh[j] = __builtin_bit_cast(bf16x8, S.v[j][0]);
            bl[j] = __builtin_bit_cast(bf16x8, S.v[j][1]);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h[i], bh[j], acc[i][j], 0, 0, 0);
    };

    // activation staging: piece e = tid + 256 q of the tile = (row e >> 4, channels 4 (e & 15) .. +3); rows past the end read the last row
    // and are zeroed when split (never stored either)
    f32x4 ra[AL];
    int64_t goff[AL];
    int okmask = 0;
#pragma unroll
    for (int q = 0; q < AL; ++q) {
        const int e = tid + 256 * q;
        const int r = m0 + (e >> 4);
        if (r < a.out_rows) okmask |= 1 << q;
        goff[q] = (int64_t)min(r, a.out_rows - 1) * a.Cin + 4 * (e & 15);
    }
    auto load_A = [&](int kc) {
#pragma unroll
        for (int q = 0; q < AL; ++q) ra[q] = *reinterpret_cast<const f32x4 *>(a.in + goff[q] + kc);
    };
    auto store_A = [&](int buf) {
        __bf16 *Ah = lds + buf * 2 * PART, *Al = Ah + PART;
#pragma unroll
        for (int q = 0; q < AL; ++q) {
            const int e = tid + 256 * q;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bf16x4 hi, lo;
            split4((okmask >> q) & 1 ? ra[q] : z, hi, lo);
            *reinterpret_cast<bf16x4 *>(Ah + (e >> 4) * GROW + 4 * (e & 15)) = hi;
            *reinterpret_cast<bf16x4 *>(Al + (e >> 4) * GROW + 4 * (e & 15)) = lo;
        }
    };

    const int n_half = 4 * kchunks;          // k16 half-steps in total
    load_A(0);
    load_B(Bq[0], 0);
    load_B(Bq[1], 1);
    load_B(Bq[2], 2);
    store_A(0);
    __syncthreads();
    for (int kq = 0; kq < kchunks; ++kq) {
        const bool more = kq + 1 < kchunks;
        const __bf16 *Abuf = lds + (kq & 1) * 2 * PART;
        read_A(Aq[0], Abuf, 0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int hn = 4 * kq + h + 3;          // weight fragments of half-step hn: requested three half-steps ahead
            if (hn < n_half) load_B(Bq[(h + 3) & 3], hn);
            if (h == 0 && more) load_A((kq + 1) * GK);          // after this half-step's weight request
            if (h + 1 < 4) read_A(Aq[(h + 1) & 1], Abuf, h + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(Aq[h & 1], Bq[h & 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (h == 2 && more) {
                store_A((kq + 1) & 1);          // the other buffer: last read before the previous chunk's closing barrier
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue (as k_conv_igemm_b3)
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

}  // namespace

// 1-tap dense geometry whose source row is the output row (nn.Linear and 1x1 stride-1 convolutions, forward or data gradient),
// fp32 activations, Cin % 64 == 0, Cout % 32 == 0.  Mirrored by kernels.py::wants_frag_weights.
bool gemm_b3f_applies(const ConvArgs &a) {
    const rd_conv_index &ix = a.ix;
    if (a.taps != 1 || !(ix.mode == 1 || ix.mode == 2) || ix.KH != 1 || ix.KW != 1 || ix.stride != 1 || ix.pad != 0) return false;
    if (ix.Hin != ix.Hout || ix.Win != ix.Wout || a.in_rows != a.out_rows) return false;
    return !a.in_split && a.Cin % GK == 0 && a.Cout % 32 == 0 && a.out_rows > 0;
}

int launch_gemm_b3f(const ConvArgs &a, hipStream_t st) {
    dim3 block(256);
    static const int tile_env = getenv("RD_GEMMF_TILE") ? atoi(getenv("RD_GEMMF_TILE")) : 0;          // diagnostic: 128 / 64 forces the row tile
    const int64_t big = cdiv(a.out_rows, 128) * cdiv(a.Cout, 128);
    const bool bm128 = tile_env ? tile_env == 128 : big >= 256;
    if (a.Cout >= 128) {
        if (bm128) k_gemm_b3f<128, 128><<<dim3(xcd_grid(cdiv(a.out_rows, 128), cdiv(a.Cout, 128))), block, 0, st>>>(a);
        else k_gemm_b3f<64, 128><<<dim3(xcd_grid(cdiv(a.out_rows, 64), cdiv(a.Cout, 128))), block, 0, st>>>(a);
    } else {
        if (bm128) k_gemm_b3f<128, 64><<<dim3(xcd_grid(cdiv(a.out_rows, 128), cdiv(a.Cout, 64))), block, 0, st>>>(a);
        else k_gemm_b3f<64, 64><<<dim3(xcd_grid(cdiv(a.out_rows, 64), cdiv(a.Cout, 64))), block, 0, st>>>(a);
    }
    return RD_OK;
}
