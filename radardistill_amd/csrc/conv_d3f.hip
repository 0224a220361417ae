// Dense stride-1 3x3 convolution, bf16x3 arithmetic, third generation: halo in LDS, WEIGHT FRAGMENTS STRAIGHT FROM L2.
//
// k_conv_d3_b3 (conv_b3.hip) stages one (tap, 32-channel chunk) weight tile per step through LDS: per step a workgroup writes 16 KB
// of weights, reads them back as fragments and ends with a barrier -- 72 barrier-separated steps of 24 MFMAs per wave for a 256-channel
// layer.  PMC (round 2): MFMA busy 41 %, half of the wave cycles in issue stalls, a third of all LDS reads are weight fragments.
// The weights of a layer are <= 2.4 MB, L2-resident, and every workgroup reads ALL of them.  So here they are laid out once per
// optimizer step in FRAGMENT-MAJOR split format (rd_weight_layout_split, kind | RD_LAYOUT_FRAG: per 32 out-channels x 16 in-channels
// of one tap the hi and the lo operand image of v_mfma_f32_32x32x16_bf16, 1 KiB each) and a wave loads its B operands with coalesced
// 16-byte-per-lane global loads, one tap ahead in registers.  What remains in LDS is the activation halo, double-buffered, so a
// workgroup needs ONE barrier per 32-channel chunk (9 taps, 216 MFMAs per wave between barriers) and half the LDS reads.
// Same contract as k_conv_d3_b3: index mode 1 (forward) or mode 2 with stride 1 (data gradient: taps mirrored, weights in the
// [Cin][tap][Cout] orientation), fp32 activations split while staged, identical products and epilogue -- results are bit-identical
// to the LDS-staged kernel in deterministic order of accumulation (same K order: chunk-major, tap, k16).
#include <stdlib.h>
#include <type_traits>
#include "conv_common.hpp"

using namespace rd;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;      // channels per K chunk (two k16 MFMA steps)
constexpr int AROW = 40;    // bf16 elements per halo row in LDS: 80-byte rows, 16-byte fragment reads of 16 consecutive rows are conflict-free

__device__ __forceinline__ void split4(const f32x4 v, bf16x4 &hi, bf16x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)v[e];
        hi[e] = h;
        lo[e] = (__bf16)(v[e] - (float)h);
    }
}

template <int TY, int TX, int BN, int D, bool X1>
__global__ __launch_bounds__(256, 2) void k_conv_d3f_b3(const ConvArgs a, const int flip) {
    constexpr int BM = TY * TX;
    constexpr int HX = TX + 2, HR = (TY + 2) * HX;
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
    constexpr int PART = HR * AROW;                   // bf16 elements of one (hi or lo) halo image
    static_assert(MI >= 1 && NI >= 1 && BM % 64 == 0, "wave tile at least 32x32");
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * 2 * PART];          // [buffer][hi | lo][halo row][40]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int H = a.ix.Hout, W = a.ix.Wout;
    const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + TY - 1) / TY;
    const int n_row_tiles = (a.out_rows / (H * W)) * tiles_y * tiles_x;
    int row_tile, col_tile;
    if (!xcd_tile(n_row_tiles, (a.Cout + BN - 1) / BN, row_tile, col_tile)) return;
    const int b = row_tile / (tiles_y * tiles_x), y0 = ((row_tile / tiles_x) % tiles_y) * TY, x0 = (row_tile % tiles_x) * TX;
    const int n0 = col_tile * BN;
    const int fr = lane & 31, fh = lane >> 5;
    const int kchunks = a.Cin / KC, c16n = a.Cin >> 4;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A fragment bases: halo row of this lane's pixel at tap (0, 0), channels 8 fh .. +7 of the chunk (buffer 0)
    int aoff[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int p = wm * WM + i * 32 + fr;
        aoff[i] = ((p / TX) * HX + (p % TX)) * AROW + fh * 8;
    }
    // B fragments: 16-byte unit index of (32-column block, tap 0, k16 block 0, hi) + lane; column blocks past Cout read the last block
    // (their accumulators are never stored)
    const uint4 *wf = reinterpret_cast<const uint4 *>(a.w);
    int wbase[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int nb = min((n0 + wn * WN + j * 32) >> 5, (a.Cout >> 5) - 1);
        wbase[j] = nb * 9 * c16n * 128 + lane;
    }

    // One HALF-STEP = one k16 slice of one tap: NI x (hi, lo) weight fragments (4 coalesced 16-byte-per-lane loads at NI = 2), MI x (hi, lo)
    // halo fragments from LDS, 3 * MI * NI MFMAs.  18 half-steps per chunk; weight fragments are requested D half-steps ahead of their
    // MFMAs into a ring of D + 1 register sets (18 % (D + 1) == 0, so a half-step's set is static in the unrolled loop), halo fragments
    // go through TWO sets (read one half-step ahead).  The compiler's scheduler is pinned with sched_barrier: left alone (252 VGPRs with
    // whole-tap sets) it sank every load next to its first use to shorten live ranges -- `global_load; s_waitcnt vmcnt(0)` pairs in the ISA.
    struct BSet {
        uint4 v[NI][2];          // [column block][hi | lo]
    };
    struct ASet {
        bf16x8 h[MI], l[MI];
    };
    constexpr int RING = D + 1;          // register sets of weight fragments (18 % RING == 0: a half-step's set is static in the unrolled loop)
    static_assert(18 % RING == 0, "ring position must repeat every chunk");
    BSet Bq[RING];
    ASet Aq[2];
    auto load_B = [&](BSet &S, int h, int kq) {          // half-step h of chunk kq
        const int tap = h >> 1, ks = h & 1;
        const int tp = flip ? 8 - tap : tap;
        const int o = (tp * c16n + 2 * kq + ks) * 128;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            S.v[j][0] = wf[wbase[j] + o];
            S.v[j][1] = wf[wbase[j] + o + 64];
        }
    };
    auto read_A = [&](ASet &A, const __bf16 *Abuf, const int h) {
        const int g = h >> 1, ks = h & 1;
        const int shift = ((g / 3) * HX + g % 3) * AROW + ks * 16;          // compile-time per unrolled half-step
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            A.h[i] = *reinterpret_cast<const bf16x8 *>(Abuf + aoff[i] + shift);
            A.l[i] = *reinterpret_cast<const bf16x8 *>(Abuf + aoff[i] + PART + shift);
        }
    };
    auto mfmas = [&](const ASet &A, const BSet &S) {
        // per accumulator the order of k_conv_d3_b3 (lo*hi, hi*lo, then the dominant hi*hi); the three terms are issued term-major so
        // that consecutive MFMAs never depend on each other
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

    constexpr int HL = (HR * 8 + 255) / 256;          // float4 halo loads per thread and chunk
    f32x4 ra[HL];
    // this thread's halo pieces: element offset of (pixel, 4 channels) in the map -- clamped to a readable pixel when the halo position
    // lies outside the map (or past the last piece) -- and a bit saying whether the piece is real; both are the same for every chunk.
    // The loads are unconditional and the zero for outside positions is selected when the piece is SPLIT (a select right after the
    // load would wait for it there and then)
    int hoff[HL], okmask = 0;
#pragma unroll
    for (int q = 0; q < HL; ++q) {
        const int e = tid + 256 * q;
        const int hr = e >> 3, c4 = (e & 7) * 4;
        const int gy = y0 - 1 + hr / HX, gx = x0 - 1 + hr % HX;
        if (e < HR * 8 && gy >= 0 && gy < H && gx >= 0 && gx < W) okmask |= 1 << q;
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
        hoff[q] = ((b * H + cy) * W + cx) * a.Cin + c4;
    }
    auto load_halo = [&](int kc) {
#pragma unroll
        for (int q = 0; q < HL; ++q) ra[q] = *reinterpret_cast<const f32x4 *>(a.in + (int64_t)hoff[q] + kc);
    };
    auto store_halo = [&](int buf) {
        __bf16 *Ah = lds + buf * 2 * PART, *Al = Ah + PART;
#pragma unroll
        for (int q = 0; q < HL; ++q) {
            const int e = tid + 256 * q;
            if (e < HR * 8) {
                const int hr = e >> 3, c4 = (e & 7) * 4;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                bf16x4 hi, lo;
                split4((okmask >> q) & 1 ? ra[q] : z, hi, lo);
                *reinterpret_cast<bf16x4 *>(Ah + hr * AROW + c4) = hi;
                *reinterpret_cast<bf16x4 *>(Al + hr * AROW + c4) = lo;
            }
        }
    };

    load_halo(0);
#pragma unroll
    for (int h = 0; h < D; ++h) load_B(Bq[h % RING], h, 0);
    store_halo(0);
    __syncthreads();
    // Two forms of the chunk loop, chosen per tile by measurement (tools/diag/lib_ab.sh, both builds inside one GPU call):
    //   UNIFORM (64-column tile: the 8192-row layers, few workgroups per CU-second, latency-bound): the chunk body exists twice -- the
    //     loop's (every chunk but the last: requests the next chunk's weights in its last D half-steps and the next halo at h = 0, splits
    //     and stores it at h = 9, unconditionally) and the peeled last chunk (no requests) -- and the chunk ends with an LDS-only barrier.
    //     Inside either body there is no run-time condition around a load: a request under `if (more)` makes the two paths reach the
    //     following s_waitcnt with different numbers of loads in flight, the compiler waits for the smaller number, and on the taken path
    //     that is the load it has just issued (ISA: `s_waitcnt vmcnt(0)` at every chunk start); __syncthreads() drains vmcnt as well
    //     (its fence; loads and stores share the counter on gfx9).  +2 ... +4 % on those layers.
    //   otherwise (128-column tile, the dominant kernel): one loop with `if (more)` and __syncthreads().  The same change costs this
    //     tile 3 % (0.1239 -> 0.1283 ms at 8 x 64 x 64, 256 -> 256): with two workgroups per CU the drained loads are covered by the
    //     other workgroup's MFMAs, and the drain keeps the two workgroups of a CU out of phase.
    constexpr bool UNIFORM = BN == 64;
    auto chunk = [&](const int kq, auto last_tag, const bool more) {
        constexpr bool LAST = decltype(last_tag)::value;
        const bool nxt = UNIFORM ? !LAST : more;          // (a compile-time constant in the UNIFORM form)
        const __bf16 *Abuf = lds + (kq & 1) * 2 * PART;
        read_A(Aq[0], Abuf, 0);
#pragma unroll
        for (int h = 0; h < 18; ++h) {
            // weight fragments are requested D half-steps ahead of their MFMAs (the last D requests of a chunk are the next chunk's first)
            if (h + D < 18) load_B(Bq[(h + D) % RING], h + D, kq);
            else if (nxt) load_B(Bq[(h + D) % RING], h + D - 18, kq + 1);
            if (h == 0 && nxt) load_halo((kq + 1) * KC);          // after this half-step's weight request: no weight wait before h = D includes it
            if (h + 1 < 18) read_A(Aq[(h + 1) & 1], Abuf, h + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(Aq[h & 1], Bq[h % RING]);
            __builtin_amdgcn_sched_barrier(0);
            if (h == 9 && nxt) {
                store_halo((kq + 1) & 1);          // the other buffer: last read before the previous chunk's closing barrier
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // everyone finished reading this chunk's halo and writing the next one
        if constexpr (UNIFORM) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
    };
    if constexpr (UNIFORM) {
        for (int kq = 0; kq + 1 < kchunks; ++kq) chunk(kq, std::false_type{}, true);
        chunk(kchunks - 1, std::true_type{}, false);
    } else {
        for (int kq = 0; kq < kchunks; ++kq) chunk(kq, std::false_type{}, kq + 1 < kchunks);
    }

    // ---- epilogue (as k_conv_d3_b3; tile rows are pixels of the (TY, TX) patch)
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
            // residual values of the 16 rows first, all in flight together (a load between the stores waits for the store in front of it:
            // loads and stores share vmcnt)
            float resv[16];
            if (a.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const int gy = y0 + p / TX, gx = x0 + p % TX;
                    resv[r] = a.residual[((int64_t)(b * H + min(gy, H - 1)) * W + min(gx, W - 1)) * a.Cout + min(col, a.Cout - 1)];          // (clamped: unused outside)
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int gy = y0 + p / TX, gx = x0 + p % TX;
                if (gy < H && gx < W && col_ok) {
                    const int64_t row = (int64_t)(b * H + gy) * W + gx;
                    float v = acc[i][j][r] + bias;
                    csum += v;
                    csq += v * v;
                    v = fmaf(v, sc, sh);
                    if (a.residual) v += resv[r];
                    if (a.relu) v = fmaxf(v, 0.f);
                    a.out[row * a.Cout + col] = v;
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

// The shapes this kernel takes (shared with the host side's choice of weight format, kernels.py::wants_frag_weights): dense 3x3, stride 1,
// pad 1, same-size maps, Cin % 32 == 0, Cout % 32 == 0, fp32 activations.
bool conv_d3f_applies(const ConvArgs &a) {
    const rd_conv_index &ix = a.ix;
    if (!((ix.mode == 1 || ix.mode == 2) && ix.KH == 3 && ix.KW == 3 && ix.stride == 1 && ix.pad == 1 && ix.Hin == ix.Hout && ix.Win == ix.Wout))
        return false;
    if (a.in_split || a.taps != 9 || a.Cin % KC || a.Cout % 32) return false;
    const int64_t hw = (int64_t)ix.Hout * ix.Wout;
    return hw > 0 && a.out_rows % hw == 0 && a.in_rows == a.out_rows;
}

// Launch for weights in fragment-major split format (a.w_split == 2).  Tile choice as launch_conv_d3_b3 (conv_b3.hip).
int launch_conv_d3f_b3(const ConvArgs &a, hipStream_t st) {
    const rd_conv_index &ix = a.ix;
    const int flip = ix.mode == 2;
    const int64_t nb = a.out_rows / ((int64_t)ix.Hout * ix.Wout);
    const int64_t big_rows = nb * cdiv(ix.Hout, 8) * cdiv(ix.Wout, 16);
    static const int tile_env = getenv("RD_D3F_TILE") ? atoi(getenv("RD_D3F_TILE")) : 0;          // diagnostic: 128 / 64 forces the column tile
    const bool wide = tile_env ? tile_env == 128 : (a.Cout >= 128 && big_rows * cdiv(a.Cout, 128) >= 384);
    dim3 block(256);
    // weight fragments two half-steps ahead: three and four were measured equal or slower on every shape of the step (tools/diag/d3_ring.sh,
    // round 3) -- the L2 round trip is covered at two, and the 128-column tile has no registers left for more
    const dim3 gw(xcd_grid(big_rows, cdiv(a.Cout, 128))), gn(xcd_grid(big_rows, cdiv(a.Cout, 64)));
#define RD_D3F_LAUNCH(BN_, G_)                                                              \
    do {                                                                                    \
        if (a.x1) k_conv_d3f_b3<8, 16, BN_, 2, true><<<G_, block, 0, st>>>(a, flip);        \
        else k_conv_d3f_b3<8, 16, BN_, 2, false><<<G_, block, 0, st>>>(a, flip);            \
    } while (0)
    if (wide) RD_D3F_LAUNCH(128, gw);
    else RD_D3F_LAUNCH(64, gn);
#undef RD_D3F_LAUNCH
    return RD_OK;
}
