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
// part: [rows][32] bf16, 16-byte chunks XOR-swizzled by the row (a 16-lane ds_read_b128 group covers all 64 banks once).  Lane l reads A[row l&31]
// [k = 8*(l>>5) + 0..7] as one 16-byte fragment (the 32x32x16 bf16 operand map).
#include <algorithm>
#include <stdlib.h>
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

// 16-byte split-format group -> its hi and lo halves
__device__ __forceinline__ void unpack4(const f32x4 v, bf16x4 &hi, bf16x4 &lo) {
    union {
        f32x4 f;
        bf16x4 h[2];
    } u;
    u.f = v;
    hi = u.h[0];
    lo = u.h[1];
}

// fp32 -> split format, elementwise (n4 groups of 4): the producer-side half of the pre-split operand path
__global__ void k_split_bf16(const f32x4 *__restrict__ x, int64_t n4, f32x4 *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        bf16x4 hi, lo;
        split4(x[i], hi, lo);
        union {
            f32x4 f;
            bf16x4 h[2];
        } u;
        u.h[0] = hi;
        u.h[1] = lo;
        out[i] = u.f;
    }
}

extern "C" int rd_split_bf16(const float *x, int64_t n, void *out, void *stream) {
    RD_REQUIRE(n >= 0 && n % 4 == 0, "rd_split_bf16: element count %lld must be a multiple of 4", (long long)n);
    if (n == 0) return RD_OK;
    const int64_t n4 = n / 4;
    k_split_bf16<<<(unsigned)std::min<int64_t>(cdiv(n4, 256), 4096), 256, 0, S(stream)>>>(reinterpret_cast<const f32x4 *>(x), n4,
                                                                                         reinterpret_cast<f32x4 *>(out));
    return check_launch("rd_split_bf16");
}

// BT = true: data gradient on the forward weights, B[k][n] = w[k][tap][n] (see k_conv_igemm in conv.hip).  The weight tile then
// arrives n-contiguous; every thread takes a 4 (k) x 4 (n) block, transposes it in registers and writes k-contiguous pieces.
// neighbour-table geometry (index mode 0) without the other modes' code
__device__ __forceinline__ int table_row(const ConvArgs &a, int j, int t) {
    if (j >= a.out_rows) return -1;
    const int tt = a.ix.flip ? (a.taps - 1 - t) : t;
    return a.ix.nbr[(int64_t)j * a.taps + tt];
}

// SPEC >= 0 fixes three block-uniform run-time switches at compile time (bit 0: dense geometry = index modes 1 / 2 instead of the
// neighbour table, bit 1: weights pre-split, bit 2: input pre-split): as run-time flags each of them put a branch around every
// operand piece in the K loop (ISA: the loop body was ~60 basic blocks), which kept the compiler from scheduling the loads and the
// splits of a step together.  SPEC = -1 keeps all three as run-time values (the rarely used entry points).
template <int BM, int BN, bool DEFORM, bool BT = false, int SPEC = -1>
__global__ __launch_bounds__(256, 2) void k_conv_igemm_b3(const ConvArgs a) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int AP = BM / 32, BP = BN / 32;
    static_assert(MI >= 1 && NI >= 1, "wave tile at least 32x32");
    // per buffer: [A hi BM rows][A lo][B hi BN rows][B lo]; UNPADDED 64-byte rows (32 bf16) with the 16-byte chunk index XOR-ed
    // by (row >> 2) & 3: a 16-lane ds_read_b128 group (16 rows, one chunk) still covers all 64 banks once, and the 128x128 tile
    // needs 64 KiB instead of 80 KiB of LDS, so two workgroups fit one CU
    constexpr int LDS_ROW = KB3;                  // bf16 elements per row
    constexpr int BUF = 2 * (BM + BN) * LDS_ROW;  // bf16 elements per buffer
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * BUF];

    // SPEC bit 3: SUB-PIXEL form of stride-2 TRANSPOSED geometry (ConvTranspose2d k 4 s 2 p 1; the data gradient of a stride-2
    // conv).  Output pixel (2y + py, 2x + px) only meets the taps with (py + pad - ky) and (px + pad - kx) even, at the input pixels
    // (y + (py + pad - ky) / 2, x + (px + pad - kx) / 2): a tile is BM pixels (y, x) of ONE parity class (py, px) on the half-resolution
    // grid, its tap mask holds that class's taps only (4 of 16 for the 4x4 kernel, 1 / 2 / 2 / 4 of 9 for 3x3 -- the gathered form
    // walks all of them for every row) and its rows are scattered to the class's output pixels in the epilogue.  The four classes of
    // a tile position are adjacent in launch order (they read the same input rows).
    constexpr bool SUBPIX = SPEC >= 0 && (SPEC & 8) != 0;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    int row_tile, col_tile;
    const int sp_h = (a.ix.Hout + 1) >> 1, sp_w = (a.ix.Wout + 1) >> 1;          // half-resolution class grid
    const int tile_rows = SUBPIX ? a.ix.B * sp_h * sp_w : a.out_rows;          // rows the tiles enumerate
    if (!xcd_tile((SUBPIX ? 4 : 1) * ((tile_rows + BM - 1) / BM), (a.Cout + BN - 1) / BN, row_tile, col_tile)) return;
    const int sp_py = SUBPIX ? ((row_tile >> 1) & 1) : 0, sp_px = SUBPIX ? (row_tile & 1) : 0;
    const int m0 = (SUBPIX ? (row_tile >> 2) : row_tile) * BM, n0 = col_tile * BN;
    const int ld_r = tid >> 3, ld_c = (tid & 7) * 4;

    // ---- dense geometry (modes 1 / 2): (b, oy, ox) of every tile row is worked out ONCE (three integer divisions per row, by one
    // thread per row) and kept in LDS; a tap's source row is then a handful of integer operations.  The first version evaluated
    // src_row() -- divisions by the map size included -- for every (row, tap) in the pre-pass below and again per thread at every
    // tap change: 80-105 VALU instructions each (ISA), a fixed cost that dominated short-K launches and was a quarter of the
    // VALU stream of the 9-tap ones.
    __shared__ int4 s_pix[BM];
    const bool dense = SPEC >= 0 ? (!DEFORM && (SPEC & 1)) : (!DEFORM && (a.ix.mode == 1 || a.ix.mode == 2));
    const bool w_presplit = SPEC >= 0 ? ((SPEC & 2) != 0) : (a.w_split != 0);
    const bool in_presplit = SPEC >= 0 ? ((SPEC & 4) != 0) : (a.in_split != 0);
    if (dense && tid < BM) {
        const int j = m0 + tid;
        int4 q = make_int4(0, 0, 0, 0);
        const int gw = SUBPIX ? sp_w : a.ix.Wout, gh = SUBPIX ? sp_h : a.ix.Hout;
        if (j < tile_rows) {
            const int t1 = j / gw;
            q = make_int4(t1 / gh, t1 % gh, j - t1 * gw, 1);
            if (SUBPIX && (2 * q.y + sp_py >= a.ix.Hout || 2 * q.z + sp_px >= a.ix.Wout)) q.w = 0;          // odd map sizes: no such output pixel
        }
        s_pix[tid] = q;
    }
    // ---- taps with a source row in this tile (the LDS array is free before the main loop: word 0 is the mask)
    int *s_mask = reinterpret_cast<int *>(lds);
    if (tid == 0) *s_mask = 0;
    __syncthreads();
    {
        int mask = 0;
        if (SUBPIX) {          // block-uniform: the taps of this parity class
            for (int t = 0; t < a.taps; ++t) {
                const int ky = t / a.ix.KW, kx = t - ky * a.ix.KW;
                if (((sp_py + a.ix.pad - ky) & 1) == 0 && ((sp_px + a.ix.pad - kx) & 1) == 0) mask |= 1 << t;
            }
        } else if (dense) {
            if (tid < BM) {
                const int4 q = s_pix[tid];
                if (q.w)
                    for (int t = 0; t < a.taps; ++t)
                        if (src_row_dense(a.ix, q.x, q.y, q.z, t) >= 0) mask |= 1 << t;
            }
        } else {
            for (int p = 0; p < AP; ++p) {
                const int j = m0 + ld_r + 32 * p;
                if ((tid & 7) == 0)
                    for (int t = 0; t < a.taps; ++t)
                        if ((SPEC >= 0 && !DEFORM ? table_row(a, j, t) : src_row(a, j, t)) >= 0) mask |= 1 << t;
            }
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

    // two operand register sets: tile s+1 waits in one while tile s+2 is still in flight into the other (loads are issued two
    // K steps before their LDS store; the halo kernel below measured +11 % from the same change)
    struct Regs {
        f32x4 ra[AP], rb[BT ? 1 : BP], rbt[BT ? 4 : 1];
    };
    Regs rx, ry;
    const int bt_g = tid & 7, bt_q = tid >> 3;   // BT loader: k rows 4 bt_g .. +3, weight columns 4 bt_q .. +3
    int rows[AP];
    int4 sidx[DEFORM ? AP : 1];
    f32x4 sw[DEFORM ? AP : 1];
    int cur_tap = -1, tap_iter_mask = tapmask;
    int w_tap = 0;                               // weight tap of the current step (= cur_tap except in the sub-pixel form)

    int next_kq = 0;                             // K chunk of the next tile to fetch (tiles are fetched strictly in order:
    auto load_tile = [&](int, Regs &R) {         // a counter instead of s % kchunks, which costs ~50 VALU instructions per step)
        const int kc = next_kq * KB3;
        const bool new_tap = next_kq == 0;
        if (++next_kq == kchunks) next_kq = 0;
        if (new_tap) {
            cur_tap = __ffs(tap_iter_mask) - 1;
            tap_iter_mask &= tap_iter_mask - 1;
            w_tap = cur_tap;
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
            } else if (SUBPIX) {
                const int ky = cur_tap / a.ix.KW, kx = cur_tap - ky * a.ix.KW;          // block-uniform; parities match by the mask
                const int oy_off = (sp_py + a.ix.pad - ky) >> 1, ox_off = (sp_px + a.ix.pad - kx) >> 1;          // arithmetic shift: exact, may be negative
#pragma unroll
                for (int p = 0; p < AP; ++p) {
                    const int4 q = s_pix[ld_r + 32 * p];
                    const int iy = q.y + oy_off, ixx = q.z + ox_off;
                    rows[p] = (q.w && iy >= 0 && iy < a.ix.Hin && ixx >= 0 && ixx < a.ix.Win) ? (q.x * a.ix.Hin + iy) * a.ix.Win + ixx : -1;
                }
            } else if (dense) {
                const int ky = cur_tap / a.ix.KW, kx = cur_tap - ky * a.ix.KW;          // block-uniform
#pragma unroll
                for (int p = 0; p < AP; ++p) {
                    const int4 q = s_pix[ld_r + 32 * p];
                    rows[p] = q.w ? src_row_dense_k(a.ix, q.x, q.y, q.z, ky, kx) : -1;
                }
            } else {
#pragma unroll
                for (int p = 0; p < AP; ++p)
                    rows[p] = SPEC >= 0 ? table_row(a, m0 + ld_r + 32 * p, cur_tap) : src_row(a, m0 + ld_r + 32 * p, cur_tap);
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
                // unconditional load (row 0 stands in for "no source"; the host guarantees one readable row) + select: no exec-mask
                // branch between the loads of a tile, so they issue back to back
                const f32x4 g = *reinterpret_cast<const f32x4 *>(a.in + (int64_t)max(rows[p], 0) * a.Cin + kc + ld_c);
                if (rows[p] >= 0) v = g;
            }
            R.ra[p] = v;
        }
        if constexpr (BT) {
            const int n = n0 + 4 * bt_q;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (bt_q < BN / 4 && n < a.Cout)
                    v = *reinterpret_cast<const f32x4 *>(a.w + ((int64_t)(kc + 4 * bt_g + e) * a.taps + w_tap) * a.Cout + n);
                R.rbt[e] = v;
            }
        } else {
#pragma unroll
            for (int p = 0; p < BP; ++p) {          // weight rows past Cout read the last row: those columns are never stored
                const int n = min(n0 + ld_r + 32 * p, a.Cout - 1);
                R.rb[p] = *reinterpret_cast<const f32x4 *>(a.w + ((int64_t)n * a.taps + w_tap) * a.Cin + kc + ld_c);
            }
        }
    };
    auto store_tile = [&](int buf, const Regs &R) {
        __bf16 *Ah = lds + buf * BUF, *Al = Ah + BM * LDS_ROW, *Bh = Al + BM * LDS_ROW, *Bl = Bh + BN * LDS_ROW;
        // element offset of this thread's 4 k-values in a row: chunk (ld_c >> 3) swizzled by the row, half-chunk ld_c & 4
        auto off = [&](int row) { return row * LDS_ROW + ((((ld_c >> 3) ^ (row >> 2)) & 3) << 3) + (ld_c & 4); };
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            bf16x4 hi, lo;
            if (!DEFORM && in_presplit) unpack4(R.ra[p], hi, lo);     // block-uniform
            else split4(R.ra[p], hi, lo);
            *reinterpret_cast<bf16x4 *>(Ah + off(ld_r + 32 * p)) = hi;
            *reinterpret_cast<bf16x4 *>(Al + off(ld_r + 32 * p)) = lo;
        }
        if constexpr (BT) {
            if (bt_q < BN / 4) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {          // column c of the 4x4 block: 4 consecutive k of weight row n = 4 bt_q + c
                    const f32x4 col = {R.rbt[0][c], R.rbt[1][c], R.rbt[2][c], R.rbt[3][c]};
                    bf16x4 hi, lo;
                    split4(col, hi, lo);
                    const int row = 4 * bt_q + c;
                    const int o = row * LDS_ROW + ((((bt_g >> 1) ^ (row >> 2)) & 3) << 3) + ((bt_g & 1) << 2);
                    *reinterpret_cast<bf16x4 *>(Bh + o) = hi;
                    *reinterpret_cast<bf16x4 *>(Bl + o) = lo;
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                bf16x4 hi, lo;
                if (w_presplit) unpack4(R.rb[p], hi, lo);
                else split4(R.rb[p], hi, lo);
                *reinterpret_cast<bf16x4 *>(Bh + off(ld_r + 32 * p)) = hi;
                *reinterpret_cast<bf16x4 *>(Bl + off(ld_r + 32 * p)) = lo;
            }
        }
    };

    if (steps > 0) {
        load_tile(0, rx);
        store_tile(0, rx);
    }
    if (steps > 1) load_tile(1, rx);          // odd tiles travel through rx, even tiles through ry
    if (steps > 2) load_tile(2, ry);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    auto k_step = [&](int s, Regs &R) {
        const int buf = s & 1;
        // fragment of lane (row fr, k = 16 ks + 8 fh .. +7) = chunk 2 ks + fh of its row, swizzled; tile rows are multiples of 32
        // apart, so (row >> 2) & 3 only depends on fr
        const __bf16 *Ah = lds + buf * BUF + (wm * WM + fr) * LDS_ROW;
        const __bf16 *Al = Ah + BM * LDS_ROW;
        const __bf16 *Bh = lds + buf * BUF + 2 * BM * LDS_ROW + (wn * WN + fr) * LDS_ROW;
        const __bf16 *Bl = Bh + BN * LDS_ROW;
        const int swz = (fr >> 2) & 3;
#pragma unroll
        for (int ks = 0; ks < KB3 / 16; ++ks) {
            const int ch = (((2 * ks + fh) ^ swz) & 3) << 3;
            bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8 *>(Ah + i * 32 * LDS_ROW + ch);
                al[i] = *reinterpret_cast<const bf16x8 *>(Al + i * 32 * LDS_ROW + ch);
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                bh[j] = *reinterpret_cast<const bf16x8 *>(Bh + j * 32 * LDS_ROW + ch);
                bl[j] = *reinterpret_cast<const bf16x8 *>(Bl + j * 32 * LDS_ROW + ch);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    // small terms first, then the dominant one
                    if (!a.x1) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (s + 1 < steps) store_tile(buf ^ 1, R);          // R holds tile s+1
        if (s + 3 < steps) load_tile(s + 3, R);
        __syncthreads();
    };
    for (int s = 0; s < steps; s += 2) {
        k_step(s, rx);
        if (s + 1 < steps) k_step(s + 1, ry);
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
            float resv[16];          // residual values first, all in flight together (a load between the stores waits for the store before it)
            if (a.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    bool row_ok = row < tile_rows;
                    if (SUBPIX) {
                        const int4 q = s_pix[row - m0];
                        row_ok = q.w != 0;
                        row = (q.x * a.ix.Hout + 2 * q.y + sp_py) * a.ix.Wout + 2 * q.z + sp_px;
                    }
                    resv[r] = a.residual[(int64_t)(row_ok ? row : 0) * a.Cout + min(col, a.Cout - 1)];          // (readable address: unused outside)
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                bool row_ok = row < tile_rows;
                if (SUBPIX) {          // this class's output pixel of input pixel (b, y, x)
                    const int4 q = s_pix[row - m0];
                    row_ok = q.w != 0;
                    row = (q.x * a.ix.Hout + 2 * q.y + sp_py) * a.ix.Wout + 2 * q.z + sp_px;
                }
                if (row_ok && col_ok) {
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

// Data gradient (BT) launch: GEMM rows = a.out_rows, K = a.Cin (forward Cout), N = a.Cout (forward Cin) > 32.
int launch_dgrad_b3(const ConvArgs &a, hipStream_t st) {
    const int64_t big_blocks = cdiv(a.out_rows, 128) * cdiv(a.Cout, 128);
    dim3 block(256);
    if (big_blocks >= 384) k_conv_igemm_b3<128, 128, false, true><<<dim3(xcd_grid(cdiv(a.out_rows, 128), cdiv(a.Cout, 128))), block, 0, st>>>(a);
    else k_conv_igemm_b3<64, 64, false, true><<<dim3(xcd_grid(cdiv(a.out_rows, 64), cdiv(a.Cout, 64))), block, 0, st>>>(a);
    return RD_OK;
}

// Launch for Cout > 32 (narrower outputs stay on the exact-fp32 kernel: they are bandwidth-bound level-1 sparse convs).
bool launch_conv_d3_b3(const ConvArgs &a, hipStream_t st);

int launch_conv_b3(const ConvArgs &a_in, int mode, hipStream_t st) {
    if (launch_conv_d3_b3(a_in, st)) return RD_OK;      // dense 3x3 stride 1: halo-staged kernel
    ConvArgs a = a_in;
    if (a.in_rows == 0) a.in = a.w;          // the gathered kernel loads row 0 for "no source" and discards it: keep that address readable
    const int64_t big_blocks = cdiv(a.out_rows, 128) * cdiv(a.Cout, 128);
    dim3 block(256);
    const dim3 g128(xcd_grid(cdiv(a.out_rows, 128), cdiv(a.Cout, 128))), g64(xcd_grid(cdiv(a.out_rows, 64), cdiv(a.Cout, 64)));
    // 1-tap layers (the ConvNeXt projections, 1x1 aggregations: K = Cin only) take 64x64 tiles whatever their size: alone the two
    // tiles time the same (35-38 us for 8192x256->1024), inside the step the 32 KB workgroups find room beside the other streams'
    // 64 KB ones sooner (step -0.5 %).  RD_GEMM_TILE64=0 restores the size rule.
    static const bool gemm64 = !(getenv("RD_GEMM_TILE64") && getenv("RD_GEMM_TILE64")[0] == '0');
    static const int big_min = getenv("RD_BIG_TILES") ? atoi(getenv("RD_BIG_TILES")) : 384;
    const bool big = big_blocks >= big_min && !(gemm64 && a.taps == 1);
    if (mode == 3) {
        if (big) k_conv_igemm_b3<128, 128, true><<<g128, block, 0, st>>>(a);
        else k_conv_igemm_b3<64, 64, true><<<g64, block, 0, st>>>(a);
        return RD_OK;
    }
    // (a 128x64 tile for the 8192-row layers -- 256 workgroups, one per CU -- was measured at 86 vs 116 TF/s for 64x64)
    static const bool spec_off = getenv("RD_CONV_SPEC") && getenv("RD_CONV_SPEC")[0] == '0';
    const int spec = spec_off ? -1 : ((mode == 1 || mode == 2) ? 1 : 0) | (a.w_split ? 2 : 0) | (a.in_split ? 4 : 0);
    // mid-size layers (fewer than 384 128x128 tiles): 64x64 tiles, except the SPARSE ones with Cout >= 128, which take 64 (rows) x 128
    // (channels) -- the neighbour gather of a row tile is then done once per 128 output channels (measured, 17 k rows 256->256:
    // 123.5 -> 110.5 us; the dense 1x1 projections lose 15 % with the same tile and stay on 64x64)
    static const bool wide_off = getenv("RD_TILE_MID") && getenv("RD_TILE_MID")[0] == '0';
    // ... and every sparse layer with exactly 128 output channels, big or not (75 k rows 128->128: 134 -> 116 us; twice the
    // workgroups of the 128x128 tile, the same single column tile)
    // ... and stride-2 TRANSPOSED geometry (ConvTranspose2d 4x4 s2 of the CMA decoders): an output row only has the taps of its own
    // (oy, ox) parity class; a 64-row tile that is one output line (Wout a multiple of 64) has a single oy parity, so the tap mask
    // drops half the taps, where a 128-row tile (two lines) keeps all sixteen
    static const bool tline_off = getenv("RD_TILE_TLINE") && getenv("RD_TILE_TLINE")[0] == '0';
    const bool tline = mode == 2 && a.ix.stride == 2 && a.ix.Wout % 64 == 0 && a.Cout >= 128 && !tline_off;
    const bool wide_mid = ((mode == 0 && (!big || a.Cout == 128)) || tline) && a.Cout >= 128 && !wide_off;
    // stride-2 transposed geometry with pre-split weights: the sub-pixel form (only the taps of an output pixel's parity class)
    static const bool subpix_off = getenv("RD_SUBPIX") && getenv("RD_SUBPIX")[0] == '0';
    if (!subpix_off && mode == 2 && a.ix.stride == 2 && !a.in_split && a.in_rows > 0 && a.taps <= 16) {
        const int64_t class_rows = (int64_t)a.ix.B * ((a.ix.Hout + 1) / 2) * ((a.ix.Wout + 1) / 2);
        const dim3 gw(xcd_grid(4 * cdiv(class_rows, 64), cdiv(a.Cout, 128))), gn(xcd_grid(4 * cdiv(class_rows, 64), cdiv(a.Cout, 64)));
        if (a.w_split) {
            if (a.Cout >= 128) k_conv_igemm_b3<64, 128, false, false, 11><<<gw, block, 0, st>>>(a);
            else k_conv_igemm_b3<64, 64, false, false, 11><<<gn, block, 0, st>>>(a);
        } else {          // fp32 weights split in the kernel (the zero-padded 27 -> 32 channel DCN offset convolution's data gradient)
            if (a.Cout >= 128) k_conv_igemm_b3<64, 128, false, false, 9><<<gw, block, 0, st>>>(a);
            else k_conv_igemm_b3<64, 64, false, false, 9><<<gn, block, 0, st>>>(a);
        }
        return RD_OK;
    }
    const dim3 g64128(xcd_grid(cdiv(a.out_rows, 64), cdiv(a.Cout, 128)));
    // big layers with <= 64 output channels (the LiDAR branch's 64-channel sparse stage, 167 k rows): a 128-column tile would compute
    // 64 columns of nothing; 128 rows x 64 columns instead
    static const int narrow_mode = getenv("RD_TILE_NARROW") ? atoi(getenv("RD_TILE_NARROW")) : 1;
    const bool narrow = narrow_mode == 1 && a.Cout <= 64;
    const dim3 g12864(xcd_grid(cdiv(a.out_rows, 128), cdiv(a.Cout, 64)));
#define RD_LAUNCH_SPEC(S)                                                                      \
    case S:                                                                                    \
        if (wide_mid) k_conv_igemm_b3<64, 128, false, false, S><<<g64128, block, 0, st>>>(a);  \
        else if (big && narrow) k_conv_igemm_b3<128, 64, false, false, S><<<g12864, block, 0, st>>>(a); \
        else if (big) k_conv_igemm_b3<128, 128, false, false, S><<<g128, block, 0, st>>>(a);   \
        else k_conv_igemm_b3<64, 64, false, false, S><<<g64, block, 0, st>>>(a);               \
        break;
    switch (spec) {
        RD_LAUNCH_SPEC(2)          // table geometry, weights pre-split (sparse layers: the default configuration)
        RD_LAUNCH_SPEC(3)          // dense geometry, weights pre-split (1x1, strided, transposed layers)
        RD_LAUNCH_SPEC(6)
        RD_LAUNCH_SPEC(7)
    default:
        if (big) k_conv_igemm_b3<128, 128, false><<<g128, block, 0, st>>>(a);
        else k_conv_igemm_b3<64, 64, false><<<g64, block, 0, st>>>(a);
    }
#undef RD_LAUNCH_SPEC
    return RD_OK;
}

// ---------------------------------------------------------------------------------------------- weight gradient, bf16x3
// Same contract as k_conv_wgrad (conv.hip): grad_w[n][t][c] += sum_j grad_out[j][n] * in[src(j,t)][c], 128 (Cout) x 128 (Cin)
// tile of one tap per workgroup, K = rows in steps of 32, row chunks combined with fp32 atomics.
// Both operands arrive k-major (rows x channels) but the 32x32x16 bf16 MFMA wants 8 consecutive k per lane, so every thread
// loads a 4 (rows) x 4 (channels) block, splits it into bf16 hi/lo, transposes it in registers and writes 4 k-contiguous
// 8-byte pieces into [channel][k] LDS images (80-byte rows, read back as 16-byte fragments exactly like the forward kernel).
// Loader map: k-group g = tid & 7 (rows 4g..4g+3), channel quad q = tid >> 3: a wave's load of one row offset covers 8 rows x
// 128 contiguous bytes; its LDS writes are 2-way bank conflicted at worst.
struct WgradArgsB3 {
    const float *in;
    int in_rows, Cin;
    const float *go;
    int out_rows, Cout, taps;
    rd_conv_index ix;
    float *gw;
    int rows_per_block;
    int in_split, go_split;   // operands already in split format (rd_split_bf16)
    int x1;                   // 1: hi * hi term only (rd_set_mfma_terms)
};

// TN = Cin tile (128, or 64 for Cin <= 64 layers such as the batched CenterHead first conv); the Cout tile is always 128.
template <bool DEFORM, int TN>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_b3(const WgradArgsB3 a) {
    constexpr int T = 128;                       // Cout tile
    constexpr int NJ = TN / 64;                  // 32-wide MFMA column tiles per wave (wave tile 64 couts x TN/2 cins)
    // ONE LDS buffer, ONE operand register set: 168 VGPRs and 40 KiB let THREE workgroups share a CU, which is worth more than either
    // refinement tried on top (PMC: waves parked in s_waitcnt / s_barrier 51 % of the time, matrix pipe busy 12.6 %): a second LDS
    // buffer (80 KiB dynamic, one barrier per K step) leaves one workgroup per CU and ran 1.7x slower; a second register set (operands
    // fetched two K steps ahead, 214 VGPRs) leaves two per CU and cost 4 % of the step.
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * (T + TN) * LDB];   // [G hi][G lo][X hi][X lo], rows of 40 bf16
    __shared__ int s_any;
    __bf16 *Gh = lds, *Gl = Gh + T * LDB, *Xh = Gl + T * LDB, *Xl = Xh + TN * LDB;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tile = blockIdx.y;
    const int n_nt = (a.Cin + TN - 1) / TN, n_mt = (a.Cout + T - 1) / T;
    const int t = tile / (n_mt * n_nt);
    const int mt = (tile / n_nt) % n_mt, nt = tile % n_nt;
    const int co0 = mt * T, ci0 = nt * TN;
    const int r_begin = blockIdx.x * a.rows_per_block;
    const int r_end = min(a.out_rows, r_begin + a.rows_per_block);
    const int n_steps = (r_end - r_begin + KB3 - 1) / KB3;
    const int g = tid & 7, q = tid >> 3;         // rows 4g..4g+3 of the step, channels 4q..4q+3 of the tile

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // dense geometry: a (b, oy, ox) cursor at this thread's first row of the step, advanced without divisions
    const bool dense = a.ix.mode == 1 || a.ix.mode == 2;
    const int t_ky = dense ? t / max(a.ix.KW, 1) : 0, t_kx = dense ? t - t_ky * a.ix.KW : 0;
    int cb = 0, cy = 0, cx = 0;
    if (dense) {
        const int j = r_begin + 4 * g;
        cx = j % a.ix.Wout;
        cy = (j / a.ix.Wout) % a.ix.Hout;
        cb = j / (a.ix.Wout * a.ix.Hout);
    }
    auto advance = [&](int &b, int &y, int &x, int n) {
        x += n;
        while (x >= a.ix.Wout) {
            x -= a.ix.Wout;
            if (++y == a.ix.Hout) {
                y = 0;
                ++b;
            }
        }
    };

    f32x4 rg[4], rx[4];
    int any_next = 0;
    auto load_tile = [&](int s) {
        const int r0 = r_begin + s * KB3 + 4 * g;
        any_next = 0;
        int b = cb, y = cy, x = cx;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = r0 + e;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (DEFORM) {
                if (j < r_end) {
                    const int64_t o = ((int64_t)j * a.taps + t) * 4;
                    const int4 c4 = *reinterpret_cast<const int4 *>(a.ix.samp_idx + o);
                    if (max(max(c4.x, c4.y), max(c4.z, c4.w)) >= 0) {
                        any_next = 1;
                        if (4 * q < TN && ci0 + 4 * q < a.Cin) {
                            const f32x4 w = *reinterpret_cast<const f32x4 *>(a.ix.samp_w + o);
                            const float *base = a.in + ci0 + 4 * q;
                            if (c4.x >= 0) v += w[0] * *reinterpret_cast<const f32x4 *>(base + (int64_t)c4.x * a.Cin);
                            if (c4.y >= 0) v += w[1] * *reinterpret_cast<const f32x4 *>(base + (int64_t)c4.y * a.Cin);
                            if (c4.z >= 0) v += w[2] * *reinterpret_cast<const f32x4 *>(base + (int64_t)c4.z * a.Cin);
                            if (c4.w >= 0) v += w[3] * *reinterpret_cast<const f32x4 *>(base + (int64_t)c4.w * a.Cin);
                        }
                    }
                }
            } else {
                int src = -1;
                if (j < r_end) {
                    if (dense) {
                        src = src_row_dense_k(a.ix, b, y, x, t_ky, t_kx);
                    } else {
                        ConvArgs c;
                        c.out_rows = a.out_rows;
                        c.taps = a.taps;
                        c.ix = a.ix;
                        src = src_row(c, j, t);
                    }
                }
                if (dense) advance(b, y, x, 1);
                if (src >= 0) {
                    any_next = 1;
                    if (4 * q < TN && ci0 + 4 * q < a.Cin) v = *reinterpret_cast<const f32x4 *>(a.in + (int64_t)src * a.Cin + ci0 + 4 * q);
                }
            }
            rx[e] = v;
            f32x4 u = {0.f, 0.f, 0.f, 0.f};
            if (j < r_end) {
                const int co = co0 + 4 * q;
                const float *src = a.go + (int64_t)j * a.Cout + co;
                if (co + 3 < a.Cout && (a.Cout & 3) == 0) u = *reinterpret_cast<const f32x4 *>(src);
                else {
                    if (co + 0 < a.Cout) u[0] = src[0];
                    if (co + 1 < a.Cout) u[1] = src[1];
                    if (co + 2 < a.Cout) u[2] = src[2];
                    if (co + 3 < a.Cout) u[3] = src[3];
                }
            }
            rg[e] = u;
        }
        if (dense) advance(cb, cy, cx, KB3);
    };
    // 4x4 register transpose + hi/lo split: element (row e, channel c) -> piece of channel c holding rows 0..3
    auto store_block = [&](const f32x4 (&blk)[4], __bf16 *hi_img, __bf16 *lo_img, int presplit) {
        bf16x4 ph[4], pl[4];
        if (presplit) {
#pragma unroll
            for (int e = 0; e < 4; ++e) unpack4(blk[e], ph[e], pl[e]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x4 hi, lo;
            if (presplit) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hi[e] = ph[e][c];
                    lo[e] = pl[e][c];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = blk[e][c];
                    const __bf16 h = (__bf16)v;
                    hi[e] = h;
                    lo[e] = (__bf16)(v - (float)h);
                }
            }
            *reinterpret_cast<bf16x4 *>(hi_img + (4 * q + c) * LDB + 4 * g) = hi;
            *reinterpret_cast<bf16x4 *>(lo_img + (4 * q + c) * LDB + 4 * g) = lo;
        }
    };

    if (tid == 0) s_any = 0;
    if (n_steps > 0) load_tile(0);
    const int fr = lane & 31, fh = lane >> 5;
    for (int s = 0; s < n_steps; ++s) {
        __syncthreads();                          // previous step's fragment reads are done
        store_block(rg, Gh, Gl, a.go_split);
        if (4 * q < TN) store_block(rx, Xh, Xl, DEFORM ? 0 : a.in_split);
        if (any_next) s_any = s + 1;              // tag = step index + 1: no reset pass needed
        __syncthreads();
        const bool any = s_any == s + 1;
        if (s + 1 < n_steps) load_tile(s + 1);    // global loads of the next step fly under this step's MFMAs
        if (any) {
            const __bf16 *Ah = Gh + (wm * 64 + fr) * LDB + 8 * fh;
            const __bf16 *Bh = Xh + (wn * (TN / 2) + fr) * LDB + 8 * fh;
#pragma unroll
            for (int ks = 0; ks < KB3 / 16; ++ks) {
                bf16x8 ah[2], al[2], bh[NJ], bl[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8 *>(Ah + i * 32 * LDB + ks * 16);
                    al[i] = *reinterpret_cast<const bf16x8 *>(Ah + T * LDB + i * 32 * LDB + ks * 16);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8 *>(Bh + j * 32 * LDB + ks * 16);
                    bl[j] = *reinterpret_cast<const bf16x8 *>(Bh + TN * LDB + j * 32 * LDB + ks * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (!a.x1) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ci = ci0 + wn * (TN / 2) + j * 32 + fr;
        if (ci < a.Cin) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const float v = acc[i][j][r];
                    if (co < a.Cout && v != 0.f) atomicAdd(&a.gw[((int64_t)co * a.taps + t) * a.Cin + ci], v);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------- weight gradient, bf16x3, transposing reads
// Second generation of k_conv_wgrad_b3 for the non-deformable cases (sparse, 1x1, strided and transposed layers; the dense
// stride-1 3x3 ones have the halo kernel).  PMC on the first one: 18 VALU instructions per MFMA and 10 % MFMA busy -- every
// element of both operands went through a 4x4 register transpose after its split, and every thread looked up FOUR source rows per
// K step.  Here the LDS images stay row-major, [32-channel chunk][row of the K step][32 channels] in 64-byte rows, exactly as
// the operands lie in memory: a thread owns ONE row of the step (one neighbour lookup) and four 16-byte pieces of it, splits each
// and stores it with a plain 8-byte write (a wavefront covers 8 consecutive 64-byte rows: conflict-free); the k-major fragments
// the 32x32x16 MFMA wants come out of ds_read_b64_tr_b16 (gfx950's transposing LDS read: 4 rows x 16 channels per 16-lane group),
// as in conv_wgrad_d3.hip.  Same tiles, chunking, `any` skip and atomic epilogue as before.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_frag32(const __bf16 *p) {
    typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + 4 * 32));
    const s16x8_t v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// SPEC >= 0: bit 0 = dense geometry (index modes 1 / 2) else the neighbour table; both operands plain fp32 and Cout % 4 == 0 (the
// training step's configuration) -- block-uniform run-time flags otherwise put a branch around every staged piece.  SPEC = -1: all
// of them stay run-time values.
template <int TN, int SPEC = -1>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad_tr_b3(const WgradArgsB3 a) {
    constexpr int T = 128;                       // Cout tile
    constexpr int NJ = TN / 64;                  // 32-wide ci chunks per wave (wave tile 64 couts x TN/2 cins)
    constexpr int GC = T / 32, XC = TN / 32;     // 32-channel chunks per operand
    constexpr int PART_G = GC * KB3 * 32, PART_X = XC * KB3 * 32;          // bf16 elements per (hi or lo) image
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * (PART_G + PART_X)];
    __shared__ int s_any;
    __bf16 *Gh = lds, *Gl = Gh + PART_G, *Xh = Gl + PART_G, *Xl = Xh + PART_X;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tile = blockIdx.y;
    const int n_nt = (a.Cin + TN - 1) / TN, n_mt = (a.Cout + T - 1) / T;
    const int t = tile / (n_mt * n_nt);
    const int mt = (tile / n_nt) % n_mt, nt = tile % n_nt;
    const int co0 = mt * T, ci0 = nt * TN;
    const int r_begin = blockIdx.x * a.rows_per_block;
    const int r_end = min(a.out_rows, r_begin + a.rows_per_block);
    const int n_steps = (r_end - r_begin + KB3 - 1) / KB3;
    const int lr = tid >> 3, lc = (tid & 7) * 4;     // this thread's row of the K step and its 4 channels inside every 32-channel chunk

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // dense geometry: a (b, oy, ox) cursor at this thread's row, advanced by 32 rows per step without divisions
    const bool dense = SPEC >= 0 ? ((SPEC & 1) != 0) : (a.ix.mode == 1 || a.ix.mode == 2);
    const bool go_presplit = SPEC >= 0 ? false : (a.go_split != 0), in_presplit = SPEC >= 0 ? false : (a.in_split != 0);
    const int t_ky = dense ? t / max(a.ix.KW, 1) : 0, t_kx = dense ? t - t_ky * a.ix.KW : 0;
    int cb = 0, cy = 0, cx = 0;
    if (dense) {
        const int j = r_begin + lr;
        cx = j % a.ix.Wout;
        cy = (j / a.ix.Wout) % a.ix.Hout;
        cb = j / (a.ix.Wout * a.ix.Hout);
    }
    const bool co_vec = SPEC >= 0 ? true : ((a.Cout & 3) == 0);

    f32x4 rg[GC], rx[XC];
    int any_next = 0;
    auto load_tile = [&](int s) {
        const int j = r_begin + s * KB3 + lr;
        int src = -1;
        if (j < r_end) {
            if (dense) {
                src = src_row_dense_k(a.ix, cb, cy, cx, t_ky, t_kx);
            } else {
                const int tt = a.ix.flip ? (a.taps - 1 - t) : t;
                src = a.ix.nbr[(int64_t)j * a.taps + tt];
            }
        }
        if (dense) {
            cx += KB3;
            while (cx >= a.ix.Wout) {
                cx -= a.ix.Wout;
                if (++cy == a.ix.Hout) {
                    cy = 0;
                    ++cb;
                }
            }
        }
        any_next = src >= 0;
        const float *xrow = a.in + (int64_t)max(src, 0) * a.Cin + ci0 + lc;
#pragma unroll
        for (int p = 0; p < XC; ++p) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (src >= 0 && ci0 + 32 * p + lc < a.Cin) v = *reinterpret_cast<const f32x4 *>(xrow + 32 * p);
            rx[p] = v;
        }
        const float *grow = a.go + (int64_t)min(j, a.out_rows - 1) * a.Cout + co0 + lc;
#pragma unroll
        for (int p = 0; p < GC; ++p) {
            f32x4 u = {0.f, 0.f, 0.f, 0.f};
            const int co = co0 + 32 * p + lc;
            if (j < r_end) {
                if (co_vec) {
                    if (co < a.Cout) u = *reinterpret_cast<const f32x4 *>(grow + 32 * p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < a.Cout) u[e] = grow[32 * p + e];
                }
            }
            rg[p] = u;
        }
    };
    const int st_off = lr * 32 + lc;             // element offset of this thread's piece inside a chunk image
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < GC; ++p) {
            bf16x4 hi, lo;
            if (go_presplit) unpack4(rg[p], hi, lo);
            else split4(rg[p], hi, lo);
            *reinterpret_cast<bf16x4 *>(Gh + p * KB3 * 32 + st_off) = hi;
            *reinterpret_cast<bf16x4 *>(Gl + p * KB3 * 32 + st_off) = lo;
        }
#pragma unroll
        for (int p = 0; p < XC; ++p) {
            bf16x4 hi, lo;
            if (in_presplit) unpack4(rx[p], hi, lo);
            else split4(rx[p], hi, lo);
            *reinterpret_cast<bf16x4 *>(Xh + p * KB3 * 32 + st_off) = hi;
            *reinterpret_cast<bf16x4 *>(Xl + p * KB3 * 32 + st_off) = lo;
        }
    };

    // fragment addressing (ds_read_b64_tr_b16): 16-lane group grp reads 4 rows x 16 channels; lane 4q + p of the group supplies the
    // address of row q, channels 4p..4p+3 and receives channel (lane & 15) of the 4 rows.  Groups 0 / 1: channels 0-15 / 16-31 at
    // k = 0..7, groups 2 / 3: the same channels at k = 8..15.
    const int grp = lane >> 4, li = lane & 15, fq = li >> 2, fp = li & 3, fhh = grp >> 1, cbb = (grp & 1) * 16;
    const int f_base = (8 * fhh + fq) * 32 + cbb + 4 * fp;          // + chunk * KB3 * 32 + ks * 16 * 32

    if (tid == 0) s_any = 0;
    if (n_steps > 0) load_tile(0);
    for (int s = 0; s < n_steps; ++s) {
        __syncthreads();                          // previous step's fragment reads are done
        store_tile();
        if (any_next) s_any = s + 1;              // tag = step index + 1: no reset pass needed
        __syncthreads();
        const bool any = s_any == s + 1;
        if (s + 1 < n_steps) load_tile(s + 1);    // global loads of the next step fly under this step's MFMAs
        if (any) {
#pragma unroll
            for (int ks = 0; ks < KB3 / 16; ++ks) {
                bf16x8 ah[2], al[2], bh[NJ], bl[NJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int o = (wm * 2 + i) * KB3 * 32 + ks * 16 * 32 + f_base;
                    ah[i] = tr_frag32(Gh + o);
                    al[i] = tr_frag32(Gl + o);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int o = (wn * NJ + j) * KB3 * 32 + ks * 16 * 32 + f_base;
                    bh[j] = tr_frag32(Xh + o);
                    bl[j] = tr_frag32(Xl + o);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (!a.x1) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    }
    // ---- combine: acc[i][j][r] is (co = 8 (r >> 2) + (r & 3) + 4 (lane >> 5), ci = lane & 31) of the (i, j) 32 x 32 block
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ci = ci0 + (wn * NJ + j) * 32 + fr;
        if (ci < a.Cin) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const float v = acc[i][j][r];
                    if (co < a.Cout && v != 0.f) atomicAdd(&a.gw[((int64_t)co * a.taps + t) * a.Cin + ci], v);
                }
        }
    }
}

int launch_wgrad_b3(const float *in, int in_rows, int Cin, const float *go, int out_rows, int Cout, int taps, const rd_conv_index *idx, float *gw,
                    int rows_per_block, int64_t chunks, int tiles, int cin_tile, int in_split, int go_split, hipStream_t st) {
    WgradArgsB3 a{in, in_rows, Cin, go, out_rows, Cout, taps, *idx, gw, rows_per_block, in_split, go_split, g_mfma_single};
    dim3 grid((unsigned)chunks, (unsigned)tiles);
    static const bool tr_off = getenv("RD_WGRAD_TR") && getenv("RD_WGRAD_TR")[0] == '0';          // A/B switch: the first-generation kernel
    if (idx->mode != 3 && !tr_off) {
        if (a.in_rows == 0) a.in = go;          // "no source" rows read row 0 and discard it: keep that address readable
        const bool plain = !in_split && !go_split && (Cout & 3) == 0;
        const bool dense = idx->mode == 1 || idx->mode == 2;
        if (cin_tile == 128) {
            if (plain && dense) k_conv_wgrad_tr_b3<128, 1><<<grid, 256, 0, st>>>(a);
            else if (plain) k_conv_wgrad_tr_b3<128, 0><<<grid, 256, 0, st>>>(a);
            else k_conv_wgrad_tr_b3<128><<<grid, 256, 0, st>>>(a);
        } else {
            if (plain && dense) k_conv_wgrad_tr_b3<64, 1><<<grid, 256, 0, st>>>(a);
            else if (plain) k_conv_wgrad_tr_b3<64, 0><<<grid, 256, 0, st>>>(a);
            else k_conv_wgrad_tr_b3<64><<<grid, 256, 0, st>>>(a);
        }
        return RD_OK;
    }
    if (cin_tile == 128) {
        if (idx->mode == 3) k_conv_wgrad_b3<true, 128><<<grid, 256, 0, st>>>(a);
        else k_conv_wgrad_b3<false, 128><<<grid, 256, 0, st>>>(a);
    } else {
        if (idx->mode == 3) k_conv_wgrad_b3<true, 64><<<grid, 256, 0, st>>>(a);
        else k_conv_wgrad_b3<false, 64><<<grid, 256, 0, st>>>(a);
    }
    return RD_OK;
}

// ---------------------------------------------------------------------------------------------- dense 3x3, stride 1: halo-staged bf16x3
// The gathered kernel above fetches (and splits) every input row once per tap: 9 times for a 3x3 convolution.  For DENSE stride-1
// 3x3 convolutions (DenseEnc, CMA, head first stages: most of the step's flops) the 9 taps of a TY x TX pixel tile read the same
// (TY+2) x (TX+2) halo, so this kernel stages and splits the halo ONCE per 32-channel K chunk and walks the 9 taps with shifted
// LDS fragment addresses: 5.7x (8x16 tile) / 5.8x (8x8) fewer activation loads, splits and LDS writes per MFMA; weights stream per
// (tap, chunk) as before.  PMC (round 1, 8x64x64 256->256): the first version issued 6.7 VALU instructions per MFMA (weight split,
// swizzle and address arithmetic), i.e. as many VALU as MFMA cycles; hence
//   * WS: weights arrive pre-split (rd_weight_layout_split writes them in the per-step re-layout launch that exists anyway),
//   * halo rows are PADDED (80-byte rows) instead of XOR-swizzled, so a tap is a constant LDS offset, and the 9 taps are unrolled:
//     fragment addresses are immediates.
// Same contract as k_conv_igemm_b3 for index mode 1 (and mode 2 = data gradient: taps mirrored, flip = 1, weights in the
// [Cin][tap][Cout] layout), same epilogue.
template <int TY, int TX, int BN, bool WS>
__global__ __launch_bounds__(256, 2) void k_conv_d3_b3(const ConvArgs a, const int flip) {
    constexpr int BM = TY * TX;
    constexpr int HX = TX + 2, HR = (TY + 2) * HX;
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32, BP = BN / 32;
    constexpr int ROW = KB3;    // weight tile: 64-byte rows, chunks XOR-swizzled by the row (as the gathered kernel)
    constexpr int AROW = 40;    // halo: 80-byte rows (16-byte fragment reads of 16 consecutive rows are conflict-free)
    static_assert(MI >= 1 && NI >= 1 && BM % 64 == 0, "wave tile at least 32x32");
    constexpr int A_EL = 2 * HR * AROW, B_EL = 2 * BN * ROW;
    __shared__ __attribute__((aligned(16))) __bf16 lds[A_EL + 2 * B_EL];
    __bf16 *Ah = lds, *Al = Ah + HR * AROW;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int H = a.ix.Hout, W = a.ix.Wout;
    const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + TY - 1) / TY;
    const int n_row_tiles = (a.out_rows / (H * W)) * tiles_y * tiles_x;
    int row_tile, col_tile;
    if (!xcd_tile(n_row_tiles, (a.Cout + BN - 1) / BN, row_tile, col_tile)) return;
    const int b = row_tile / (tiles_y * tiles_x), y0 = ((row_tile / tiles_x) % tiles_y) * TY, x0 = (row_tile % tiles_x) * TX;
    const int n0 = col_tile * BN;
    const int ld_r = tid >> 3, ld_c = (tid & 7) * 4;
    const int fr = lane & 31, fh = lane >> 5;
    const int kchunks = a.Cin / KB3;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment bases: A = halo row of this lane's pixel at tap (0,0), chunk fh; B = weight row of this lane
    const __bf16 *afrag[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int p = wm * WM + i * 32 + fr;
        afrag[i] = Ah + ((p / TX) * HX + (p % TX)) * AROW + fh * 8;
    }
    const int bswz = (fr >> 2) & 3;
    const int bch0 = ((fh ^ bswz) & 3) << 3, bch1 = (((2 + fh) ^ bswz) & 3) << 3;

    // weight loader: row n0 + ld_r + 32 p, channels kc + ld_c .. +3 of tap wt; rows past Cout read row 0 and are never stored to `out`
    const float *wrow[BP];
    int bst[BP];
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        const int n = n0 + ld_r + 32 * p;
        wrow[p] = a.w + (int64_t)(n < a.Cout ? n : 0) * 9 * a.Cin + ld_c;
        const int row = ld_r + 32 * p;
        bst[p] = row * ROW + ((((ld_c >> 3) ^ (row >> 2)) & 3) << 3) + (ld_c & 4);
    }
    // ---- pipeline.  Weights: tile s = (chunk, tap) is fetched THREE steps before its MFMAs (two register sets in flight, one LDS
    // buffer being filled while the other is read).  Halo: the next chunk's rows are fetched at tap 0 and written to LDS after
    // tap 8.  PMC on the first version: a third of all wave cycles sat in s_waitcnt vmcnt with a one-step prefetch.
    constexpr int HL = (HR * 8 + 255) / 256;     // float4 halo loads per thread
    f32x4 rbA[BP], rbB[BP], ra[HL];
    auto load_B = [&](f32x4 (&r)[BP], int g, int kc) {
        const int woff = (flip ? 8 - g : g) * a.Cin + kc;
#pragma unroll
        for (int p = 0; p < BP; ++p) r[p] = *reinterpret_cast<const f32x4 *>(wrow[p] + woff);
    };
    auto store_B = [&](__bf16 *Bh, const f32x4 (&r)[BP]) {
        __bf16 *Bl = Bh + BN * ROW;
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            bf16x4 hi, lo;
            if (WS) unpack4(r[p], hi, lo);
            else split4(r[p], hi, lo);
            *reinterpret_cast<bf16x4 *>(Bh + bst[p]) = hi;
            *reinterpret_cast<bf16x4 *>(Bl + bst[p]) = lo;
        }
    };
    auto load_halo = [&](int kc) {
#pragma unroll
        for (int q = 0; q < HL; ++q) {
            const int e = tid + 256 * q;
            const int hr = e >> 3, c4 = (e & 7) * 4;
            const int gy = y0 - 1 + hr / HX, gx = x0 - 1 + hr % HX;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (e < HR * 8 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const f32x4 *>(a.in + ((int64_t)(b * H + gy) * W + gx) * a.Cin + kc + c4);
            ra[q] = v;
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int q = 0; q < HL; ++q) {
            const int e = tid + 256 * q;
            if (e < HR * 8) {
                const int hr = e >> 3, c4 = (e & 7) * 4;
                bf16x4 hi, lo;
                split4(ra[q], hi, lo);
                *reinterpret_cast<bf16x4 *>(Ah + hr * AROW + c4) = hi;
                *reinterpret_cast<bf16x4 *>(Al + hr * AROW + c4) = lo;
            }
        }
    };

    __bf16 *bcur = lds + A_EL, *bnext = bcur + B_EL;
    load_halo(0);
    load_B(rbA, 0, 0);
    store_halo();
    store_B(bcur, rbA);
    load_B(rbA, 1, 0);            // tile s+1
    load_B(rbB, 2, 0);            // tile s+2
    __syncthreads();

    for (int kq = 0; kq < kchunks; ++kq) {
        const int kc = kq * KB3;
        const bool more = kq + 1 < kchunks;
        if (more) load_halo(kc + KB3);
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            const int shift = ((g / 3) * HX + g % 3) * AROW;        // compile-time per unrolled tap
            const __bf16 *Bh = bcur + (wn * WN + fr) * ROW, *Bl = Bh + BN * ROW;
#pragma unroll
            for (int ks = 0; ks < KB3 / 16; ++ks) {
                bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8 *>(afrag[i] + shift + ks * 16);
                    al[i] = *reinterpret_cast<const bf16x8 *>(afrag[i] + HR * AROW + shift + ks * 16);
                }
                const int bch = ks ? bch1 : bch0;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8 *>(Bh + j * 32 * ROW + bch);
                    bl[j] = *reinterpret_cast<const bf16x8 *>(Bl + j * 32 * ROW + bch);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        if (!a.x1) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
            // here rbA = tile s+1, rbB = tile s+2 (s = kq * 9 + g)
            if (g + 1 < 9 || more) store_B(bnext, rbA);        // bnext was last read in step s-1, which ended with a barrier
            if (g + 3 < 9) load_B(rbA, g + 3, kc);
            else if (more) load_B(rbA, g + 3 - 9, kc + KB3);
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                const f32x4 t = rbA[p];
                rbA[p] = rbB[p];
                rbB[p] = t;
            }
            __bf16 *tb = bcur;
            bcur = bnext;
            bnext = tb;
            __syncthreads();
        }
        if (more) {
            store_halo();          // every wave passed the barrier of tap 8: nobody reads the old halo any more
            __syncthreads();
        }
    }
    __syncthreads();

    // ---- epilogue (as k_conv_igemm_b3; tile rows are pixels of the (TY, TX) patch)
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
                const int p = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int gy = y0 + p / TX, gx = x0 + p % TX;
                if (gy < H && gx < W && col_ok) {
                    const int64_t row = (int64_t)(b * H + gy) * W + gx;
                    float v = acc[i][j][r] + bias;
                    csum += v;
                    csq += v * v;
                    v = fmaf(v, sc, sh);
                    if (a.residual) v += a.residual[row * a.Cout + col];
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

// true when the halo kernel applies; launches it (mode 1 forward or mode 2 = stride-1 data gradient, 3x3, pad 1, fp32 activations,
// weights fp32 or pre-split)
bool launch_conv_d3_b3(const ConvArgs &a, hipStream_t st) {
    const rd_conv_index &ix = a.ix;
    if (!((ix.mode == 1 || ix.mode == 2) && ix.KH == 3 && ix.KW == 3 && ix.stride == 1 && ix.pad == 1 && ix.Hin == ix.Hout && ix.Win == ix.Wout))
        return false;
    if (a.in_split || a.taps != 9 || a.Cin % KB3) return false;
    static const bool off = getenv("RD_D3") && getenv("RD_D3")[0] == '0';
    if (off) return false;
    const int flip = ix.mode == 2;
    const int64_t nb = a.out_rows / ((int64_t)ix.Hout * ix.Wout);
    if (nb * ix.Hout * ix.Wout != a.out_rows || a.in_rows != a.out_rows) return false;
    const int64_t big_rows = nb * cdiv(ix.Hout, 8) * cdiv(ix.Wout, 16);
    dim3 block(256);
    if (big_rows * cdiv(a.Cout, 128) >= 384) {
        const dim3 grid(xcd_grid(big_rows, cdiv(a.Cout, 128)));
        if (a.w_split) k_conv_d3_b3<8, 16, 128, true><<<grid, block, 0, st>>>(a, flip);
        else k_conv_d3_b3<8, 16, 128, false><<<grid, block, 0, st>>>(a, flip);
    } else {
        // 8 x 16 pixels x 64 channels (wave tile 64 x 32: 6 LDS fragment reads per 6 MFMAs instead of 4 per 3) when that still gives every
        // CU a workgroup -- the 8192-row 256 -> 256 layers, 30 launches per step: 20.09 -> 19.69 ms per step; smaller maps keep 8 x 8 x 64
        // (twice the workgroups).  RD_D3_SMALL: 0 = always 8x8x64, 1 = 8x16x64 whenever the weights are pre-split, 2 = 8x8x128.
        static const int small_env = getenv("RD_D3_SMALL") ? atoi(getenv("RD_D3_SMALL")) : -1;
        const int64_t rows64 = nb * cdiv(ix.Hout, 8) * cdiv(ix.Wout, 8);
        const int small = small_env >= 0 ? small_env : (big_rows * cdiv(a.Cout, 64) >= 256 ? 1 : 0);
        if (small == 1 && a.w_split) {
            k_conv_d3_b3<8, 16, 64, true><<<dim3(xcd_grid(big_rows, cdiv(a.Cout, 64))), block, 0, st>>>(a, flip);
        } else if (small == 2 && a.w_split) {
            k_conv_d3_b3<8, 8, 128, true><<<dim3(xcd_grid(rows64, cdiv(a.Cout, 128))), block, 0, st>>>(a, flip);
        } else {
            const dim3 grid(xcd_grid(rows64, cdiv(a.Cout, 64)));
            if (a.w_split) k_conv_d3_b3<8, 8, 64, true><<<grid, block, 0, st>>>(a, flip);
            else k_conv_d3_b3<8, 8, 64, false><<<grid, block, 0, st>>>(a, flip);
        }
    }
    return true;
}
