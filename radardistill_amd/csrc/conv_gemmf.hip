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
//
// TABLE = true: the same kernel as a SPARSE convolution over a neighbour table (index mode 0: SubMConv2d, SparseConv2d and their data
// gradients).  K runs over (tap, 64-channel block); the A tile of a chunk is GATHERED -- row j of the tile comes from input row
// nbr[j][tap] (or is zero).  The tile's slice of the table is read into LDS once (coalesced), taps without any source row in the tile
// are skipped, and a tap costs Cin / 64 barriers instead of the Cin / 32 x (gather -> LDS -> barrier) steps of k_conv_igemm_b3.
#include <stdlib.h>
#include "conv_common.hpp"

using namespace rd;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int GK = 64;       // channels per K chunk (four k16 half-steps)
constexpr int GROW = 72;     // bf16 elements per LDS row: 64 data + 8 pad = 144 bytes
constexpr int TABLE_TAPS = 9;     // neighbour tables are 3x3 (a 128-row tile's slice + two 36-KiB tile buffers still let two workgroups share a CU)

__device__ __forceinline__ void split4(const f32x4 v, bf16x4 &hi, bf16x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)v[e];
        hi[e] = h;
        lo[e] = (__bf16)(v[e] - (float)h);
    }
}

template <int BM, int BN, bool TABLE, bool X1>
__global__ __launch_bounds__(256, 2) void k_gemm_b3f(const ConvArgs a) {
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
    constexpr int PART = BM * GROW;                    // bf16 elements of one (hi or lo) tile image
    constexpr int AL = BM * (GK / 4) / 256;            // float4 activation loads per thread and chunk (BM rows x 16 pieces)
    static_assert(MI >= 1 && NI >= 1 && AL >= 1, "wave tile at least 32x32");
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * 2 * PART];          // [buffer][hi | lo][row][72]
    __shared__ int s_nbr[TABLE ? BM * TABLE_TAPS : 1];                           // TABLE: this tile's rows of the neighbour table
    __shared__ int s_mask;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int row_tile, col_tile;
    if (!xcd_tile((a.out_rows + BM - 1) / BM, (a.Cout + BN - 1) / BN, row_tile, col_tile)) return;
    const int m0 = row_tile * BM, n0 = col_tile * BN;
    const int fr = lane & 31, fh = lane >> 5;
    const int cblocks = a.Cin / GK, c16n = a.Cin >> 4;          // 64-channel blocks per tap; k16 blocks per tap

    // ---- TABLE: stage the table slice, find the taps that have a source row in this tile
    int tapmask = 1;
    if constexpr (TABLE) {
        if (tid == 0) s_mask = 0;
        const int n_ent = min(BM, a.out_rows - m0) * a.taps;
        const int *tab = a.ix.nbr + (int64_t)m0 * a.taps;
        int mine = 0;
        for (int e = tid; e < BM * a.taps; e += 256) {
            const int v = e < n_ent ? tab[e] : -1;
            s_nbr[e] = v;
            // bit = WEIGHT tap of the table column (data gradient: columns are walked mirrored, so that K runs over ascending weight
            // taps exactly as in k_conv_igemm_b3)
            if (v >= 0) mine |= 1 << (a.ix.flip ? a.taps - 1 - e % a.taps : e % a.taps);
        }
        __syncthreads();
        if (mine) atomicOr(&s_mask, mine);
        __syncthreads();
        tapmask = __builtin_amdgcn_readfirstlane(s_mask);          // bit t = weight tap t has a source row in this tile; uniform -> scalar registers
    }
    const int ntaps_on = TABLE ? __popc(tapmask) : 1;
    const int kchunks = ntaps_on * cblocks;

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
        wbase[j] = nb * (TABLE ? a.taps : 1) * c16n * 128 + lane;
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
        if constexpr (!X1) {          // the two correction terms of bf16x3 (X1: rd_set_mfma_terms(1) keeps only hi * hi)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h[i], bl[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h[i], bh[j], acc[i][j], 0, 0, 0);
    };

    // activation staging: piece e = tid + 256 q of the tile = (row e >> 4, channels 4 (e & 15) .. +3).  Rows without a source (past the
    // end; TABLE: no neighbour at this tap) read a clamped, readable row and are zeroed when split (a select right after the load would
    // wait for it there and then).
    f32x4 ra[AL];
    int goff[AL];          // element offsets (the launcher checks rows * Cin < 2^31)
    int okmask = 0, ok_next = 0;
    // chunk walk: chunk c = (c / cblocks)-th set tap, 64-channel block c % cblocks; w16 = k16 block of the chunk's first half-step in the
    // fragment-major weight image (tap * c16n + 4 * block)
    int walk_mask = tapmask, walk_cb = 0, walk_tap = 0;
    auto next_chunk = [&](int &w16, int &kc) {          // -> true when the chunk starts a new tap (source rows change)
        const bool new_tap = walk_cb == 0;
        if (new_tap) {
            walk_tap = TABLE ? __ffs(walk_mask) - 1 : 0;
            walk_mask &= walk_mask - 1;
        }
        w16 = walk_tap * c16n + 4 * walk_cb;          // walk_tap = weight tap
        kc = walk_cb * GK;
        if (++walk_cb == cblocks) walk_cb = 0;
        return new_tap;
    };
    auto load_A = [&](bool new_tap, int kc) {
        if (new_tap) {
            ok_next = 0;
#pragma unroll
            for (int q = 0; q < AL; ++q) {
                const int e = tid + 256 * q;
                int src;
                if constexpr (TABLE) src = s_nbr[(e >> 4) * a.taps + (a.ix.flip ? a.taps - 1 - walk_tap : walk_tap)];
                else src = (m0 + (e >> 4) < a.out_rows) ? m0 + (e >> 4) : -1;
                if (src >= 0) ok_next |= 1 << q;
                goff[q] = max(src, 0) * a.Cin + 4 * (e & 15);
            }
        }
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

    // ---- pipeline.  Weight fragments: ring of four register sets, half-step H (chunk H / 4, k16 slice H % 4) is requested three half-
    // steps ahead; w16_cur / w16_nxt = first k16 block of the current / next chunk.  Activations: the next chunk's tile is requested at
    // half-step 0 (after that half-step's weight request) and split into the other LDS buffer after half-step 2.
    int w16_cur = 0, w16_nxt = 0, kc_nxt = 0;
    if (kchunks > 0) {
        const bool nt = next_chunk(w16_cur, kc_nxt);
        load_A(nt, kc_nxt);
        okmask = ok_next;
        load_B(Bq[0], w16_cur);
        load_B(Bq[1], w16_cur + 1);
        load_B(Bq[2], w16_cur + 2);
        store_A(0);
    }
    __syncthreads();
    for (int kq = 0; kq < kchunks; ++kq) {
        // The next chunk's requests (weights: half-steps 1..3, activations: half-step 0, split and stored after half-step 2) are issued
        // UNCONDITIONALLY: in the last chunk they re-read this chunk (valid memory, results unused).  Under `if (more)` the two paths
        // reach the following s_waitcnt with different numbers of loads in flight and the compiler waits for the smaller number -- on the
        // taken path that is the load just issued (conv_d3f.hip, round 3: `s_waitcnt vmcnt(0)` at every chunk start).
        const bool more = kq + 1 < kchunks;
        const __bf16 *Abuf = lds + (kq & 1) * 2 * PART;
        bool nt = false;
        if (more) nt = next_chunk(w16_nxt, kc_nxt);          // (index arithmetic only)
        else w16_nxt = w16_cur;
        read_A(Aq[0], Abuf, 0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            // half-step h + 3: slice 3 of this chunk (h == 0) or slice h - 1 of the next one
            if (h == 0) load_B(Bq[3], w16_cur + 3);
            else load_B(Bq[(h + 3) & 3], w16_nxt + h - 1);
            if (h == 0) load_A(nt, kc_nxt);          // after this half-step's weight request
            if (h + 1 < 4) read_A(Aq[(h + 1) & 1], Abuf, h + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(Aq[h & 1], Bq[h & 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (h == 2) {
                okmask = ok_next;
                store_A((kq + 1) & 1);          // the other buffer: last read before the previous chunk's closing barrier
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        w16_cur = w16_nxt;
        // LDS-only barrier: __syncthreads() carries a workgroup fence, and loads share vmcnt with stores on gfx9 -- it would drain the
        // weight fragments and the activation tile requested for the next chunk
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
            // residual values of the 16 rows first, all in flight together: a load placed between the stores waits -- loads and stores
            // share vmcnt -- for the store in front of it (ISA, round 3: `s_waitcnt vmcnt(0)` per element, one HBM round trip each)
            float resv[16];
            if (a.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    resv[r] = a.residual[(int64_t)min(row, a.out_rows - 1) * a.Cout + min(col, a.Cout - 1)];          // (clamped: unused outside)
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row < a.out_rows && col_ok) {
                    float v = acc[i][j][r] + bias;
                    csum += v;
                    csq += v * v;
                    v = fmaf(v, sc, sh);
                    if (a.residual) v += resv[r];
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

// Shapes the kernel takes (mirrored by kernels.py::wants_frag_weights): fp32 activations, Cin % 64 == 0, Cout % 32 == 0 and
//   * 1-tap dense geometry whose source row is the output row (nn.Linear and 1x1 stride-1 convolutions, forward or data gradient), or
//   * a neighbour table (index mode 0), up to 9 taps.
bool gemm_b3f_applies(const ConvArgs &a) {
    const rd_conv_index &ix = a.ix;
    if (a.in_split || a.Cin % GK != 0 || a.Cout % 32 != 0 || a.out_rows <= 0) return false;
    if ((int64_t)max(a.in_rows, 1) * a.Cin >= (int64_t)1 << 31) return false;          // 32-bit element offsets in the kernel
    if (ix.mode == 0) return a.taps >= 1 && a.taps <= TABLE_TAPS && ix.nbr != nullptr;
    if (a.taps != 1 || !(ix.mode == 1 || ix.mode == 2) || ix.KH != 1 || ix.KW != 1 || ix.stride != 1 || ix.pad != 0) return false;
    return ix.Hin == ix.Hout && ix.Win == ix.Wout && a.in_rows == a.out_rows;
}

template <bool TABLE>
static void launch_gemm_tiles(const ConvArgs &a, hipStream_t st) {
    dim3 block(256);
    static const int tile_env = getenv("RD_GEMMF_TILE") ? atoi(getenv("RD_GEMMF_TILE")) : 0;          // diagnostic: 128 / 64 forces the row tile
    const int64_t big = cdiv(a.out_rows, 128) * cdiv(a.Cout, 128);
    const bool bm128 = tile_env ? tile_env == 128 : big >= (TABLE ? 384 : 256);
#define RD_GEMMF_LAUNCH(BM_, BN_)                                                                                                \
    do {                                                                                                                         \
        const dim3 grid(xcd_grid(cdiv(a.out_rows, BM_), cdiv(a.Cout, BN_)));                                                     \
        if (a.x1) k_gemm_b3f<BM_, BN_, TABLE, true><<<grid, block, 0, st>>>(a);                                                  \
        else k_gemm_b3f<BM_, BN_, TABLE, false><<<grid, block, 0, st>>>(a);                                                      \
    } while (0)
    if (a.Cout >= 128) {
        if (bm128) RD_GEMMF_LAUNCH(128, 128);
        else RD_GEMMF_LAUNCH(64, 128);
    } else {
        if (bm128) RD_GEMMF_LAUNCH(128, 64);
        else RD_GEMMF_LAUNCH(64, 64);
    }
#undef RD_GEMMF_LAUNCH
}

int launch_gemm_b3f(const ConvArgs &a_in, hipStream_t st) {
    ConvArgs a = a_in;
    if (a.in_rows == 0) a.in = a.w;          // "no source" pieces read row 0 and are discarded: keep that address readable
    if (a.ix.mode == 0) launch_gemm_tiles<true>(a, st);
    else launch_gemm_tiles<false>(a, st);
    return RD_OK;
}
