// Convolution as gathered implicit GEMM on the gfx950 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
// See include/rdamd.h section C.  One kernel family for sparse (rulebook/neighbour-table) and dense channels-last
// convolutions, transposed convolutions and linear layers:
//
//     out[j][n] = epilogue( sum_t sum_c in[src(j,t)][c] * W[n][t][c] )
//
// Block tile 128 (rows) x BN (out channels), K step 32 input channels of one tap; 4 waves (one per SIMD).
// A rows are gathered straight from HBM/L2 by `src(j,t)` (a neighbour-table entry or dense geometry), 128-byte
// segments per row, staged through registers into a padded LDS image (row stride 36 floats: ds_read_b128 of
// 16 rows x 4 floats is bank-conflict free); weights W[n][t][c..c+31] are staged the same way.  Double-buffered LDS,
// next tile's global loads are in flight while the current tile runs on the MFMA pipe.  Taps for which no row of the
// tile has a source are skipped (sparse rulebooks are far from full at the shallow stages).
// MFMA operand maps (guide section 3): A: lane l holds A[i = l&31][k = l>>5]; B: B[k = l>>5][j = l&31];
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#include "conv_common.hpp"

using namespace rd;

// BT = true: the weight operand is read TRANSPOSED, B[k][n] = w[k][tap][n] with w in the forward kernel layout [K][taps][N].
// That is the data-gradient GEMM (K = forward Cout, N = forward Cin) on the forward weights as they are: no re-layout launch.
// KBT = K step (input channels of one tap per LDS tile).  64 was tried for the 64x64 tile (longer steps against gather latency):
// 5-10 % slower -- the doubled LDS footprint halves the resident workgroups -- so every launch uses 32.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool DEFORM, bool BT = false, int KBT = 32>
__global__ __launch_bounds__(256, 2) void k_conv_igemm(const ConvArgs a) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int LDKT = KBT + 4;        // padded LDS row (floats)
    constexpr int TPRK = KBT / 4;        // threads per tile row (one float4 each)
    constexpr int RP = 256 / TPRK;       // tile rows covered per pass of the 256 threads
    constexpr int BP = BN * KBT / 1024;  // B float4 loads per thread
    constexpr int AP = BM * KBT / 1024;  // A float4 loads per thread
    static_assert(WAVES_M * WAVES_N == 4 && MI >= 1 && NI >= 1, "4 waves, each at least one 32x32 MFMA tile");
    constexpr int LDBT = BN + 4;                                     // BT: B tile stored [k][n], row stride BN + 4 floats
    constexpr int BSZ = BT ? KBT * LDBT : BN * LDKT;                 // floats of one B tile
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM * LDKT + BSZ)];
    __shared__ int s_tapmask;
    constexpr int BUF = BM * LDKT + BSZ;  // floats per buffer: [A tile BM x (KBT + 4)][B tile]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    int row_tile, col_tile;
    if (!xcd_tile((a.out_rows + BM - 1) / BM, (a.Cout + BN - 1) / BN, row_tile, col_tile)) return;
    const int m0 = row_tile * BM, n0 = col_tile * BN;
    const int ld_r = tid / TPRK, ld_c = (tid % TPRK) * 4;

    // ---- which taps have any source row in this tile
    if (tid == 0) s_tapmask = 0;
    __syncthreads();
    {
        int mask = 0;
        for (int p = 0; p < AP; ++p) {
            int j = m0 + ld_r + RP * p;
            if ((tid % TPRK) == 0)
                for (int t = 0; t < a.taps; ++t)
                    if (src_row(a, j, t) >= 0) mask |= 1 << t;
        }
        if (mask) atomicOr(&s_tapmask, mask);
    }
    __syncthreads();
    const int tapmask = s_tapmask;
    const int kchunks = a.Cin / KBT;
    const int n_active = __popc(tapmask);
    const int steps = n_active * kchunks;

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
        int kc = (s % kchunks) * KBT;
        if (s % kchunks == 0) {  // next active tap
            cur_tap = __ffs(tap_iter_mask) - 1;
            tap_iter_mask &= tap_iter_mask - 1;
            if constexpr (DEFORM) {
#pragma unroll
                for (int p = 0; p < AP; ++p) {
                    const int j = m0 + ld_r + RP * p;
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
                for (int p = 0; p < AP; ++p) rows[p] = src_row(a, m0 + ld_r + RP * p, cur_tap);
            }
        }
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (DEFORM) {
                // A element = mask * bilinear(x): up to four weighted rows (weights already hold mask * corner weight)
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
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (BT) {
                const int idx = tid + 256 * p, kr = idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
                if (n < a.Cout) v = *reinterpret_cast<const f32x4 *>(a.w + ((int64_t)(kc + kr) * a.taps + cur_tap) * a.Cout + n);
            } else {
                const int n = n0 + ld_r + RP * p;
                if (n < a.Cout) v = *reinterpret_cast<const f32x4 *>(a.w + ((int64_t)n * a.taps + cur_tap) * a.Cin + kc + ld_c);
            }
            rb[p] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int p = 0; p < AP; ++p) *reinterpret_cast<f32x4 *>(lds + buf * BUF + (ld_r + RP * p) * LDKT + ld_c) = ra[p];
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            if constexpr (BT) {
                const int idx = tid + 256 * p;
                *reinterpret_cast<f32x4 *>(lds + buf * BUF + BM * LDKT + (idx / (BN / 4)) * LDBT + (idx % (BN / 4)) * 4) = rb[p];
            } else {
                *reinterpret_cast<f32x4 *>(lds + buf * BUF + BM * LDKT + (ld_r + RP * p) * LDKT + ld_c) = rb[p];
            }
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
        const float *Ab = lds + buf * BUF + (wm * WM + fr) * LDKT + 4 * fh;
        const float *Bb = BT ? lds + buf * BUF + BM * LDKT + (4 * fh) * LDBT + wn * WN + fr
                             : lds + buf * BUF + BM * LDKT + (wn * WN + fr) * LDKT + 4 * fh;
#pragma unroll
        for (int kk = 0; kk < KBT / 8; ++kk) {
            f32x4 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f32x4 *>(Ab + i * 32 * LDKT + kk * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if constexpr (BT) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) bf[j][q] = Bb[(kk * 8 + q) * LDBT + j * 32];
                } else {
                    bf[j] = *reinterpret_cast<const f32x4 *>(Bb + j * 32 * LDKT + kk * 8);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], bf[j][q], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < steps) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue
    float *red = lds;  // [2][BN] column sums for BatchNorm statistics (LDS is free now: last loop iteration ended with a barrier)
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
            int col = n0 + i;
            if (col < a.Cout) {
                atomicAdd(&a.stats[col], red[i]);
                atomicAdd(&a.stats[a.Cout + col], red[BN + i]);
            }
        }
    }
}

static int validate_index(const rd_conv_index *ix, int taps, int in_rows, int out_rows, const char *who) {
    RD_REQUIRE(ix != nullptr, "%s: index spec is NULL", who);
    RD_REQUIRE(taps >= 1 && taps <= MAX_TAPS, "%s: taps=%d outside 1..%d", who, taps, MAX_TAPS);
    if (ix->mode == 0) {
        RD_REQUIRE(ix->nbr != nullptr || out_rows == 0, "%s: TABLE mode needs nbr", who);
    } else if (ix->mode == 3) {
        RD_REQUIRE((ix->samp_idx != nullptr && ix->samp_w != nullptr) || out_rows == 0, "%s: DEFORM mode needs samp_idx / samp_w", who);
    } else {
        RD_REQUIRE(ix->mode == 1 || ix->mode == 2, "%s: bad index mode %d", who, ix->mode);
        RD_REQUIRE(ix->KH * ix->KW == taps, "%s: KH*KW=%d != taps=%d", who, ix->KH * ix->KW, taps);
        RD_REQUIRE(ix->stride >= 1 && ix->pad >= 0, "%s: bad stride/pad", who);
        RD_REQUIRE((int64_t)ix->B * ix->Hin * ix->Win == in_rows, "%s: in_rows=%d != B*Hin*Win=%lld", who, in_rows,
                   (long long)ix->B * ix->Hin * ix->Win);
        RD_REQUIRE((int64_t)ix->B * ix->Hout * ix->Wout == out_rows, "%s: out_rows=%d != B*Hout*Wout=%lld", who, out_rows,
                   (long long)ix->B * ix->Hout * ix->Wout);
    }
    return RD_OK;
}

// 0 = exact fp32 MFMA (default), 1 = bf16x3 split MFMA (conv_b3.hip)
static int g_conv_math = 0;
int launch_conv_b3(const ConvArgs &a, int mode, hipStream_t st);
bool launch_conv_small_b3(const ConvArgs &a, hipStream_t st);
bool launch_conv_narrow_b3(const ConvArgs &a, hipStream_t st);
int launch_dgrad_b3(const ConvArgs &a, hipStream_t st);
int launch_wgrad_b3(const float *in, int in_rows, int Cin, const float *go, int out_rows, int Cout, int taps, const rd_conv_index *idx, float *gw,
                    int rows_per_block, int64_t chunks, int tiles, int cin_tile, int in_split, int go_split, hipStream_t st);
extern "C" int rd_set_conv_math(int mode) {
    RD_REQUIRE(mode == 0 || mode == 1, "rd_set_conv_math: mode must be 0 (f32) or 1 (bf16x3)");
    g_conv_math = mode;
    return RD_OK;
}
extern "C" int rd_get_conv_math(void) { return g_conv_math; }

static int conv_fwd_impl(const float *in, int in_rows, int Cin, const float *weight_k, int taps, const float *bias, float *out,
                         int out_rows, int Cout, const rd_conv_index *idx, const float *scale, const float *shift,
                         const float *residual, int relu, float *stats, int in_split, int w_split, void *stream);
bool conv_d3f_applies(const ConvArgs &a);                      // conv_d3f.hip
int launch_conv_d3f_b3(const ConvArgs &a, hipStream_t st);
bool gemm_b3f_applies(const ConvArgs &a);                      // conv_gemmf.hip
int launch_gemm_b3f(const ConvArgs &a, hipStream_t st);

extern "C" int rd_conv_fwd(const float *in, int in_rows, int Cin, const float *weight_k, int taps, const float *bias, float *out,
                           int out_rows, int Cout, const rd_conv_index *idx, const float *scale, const float *shift,
                           const float *residual, int relu, float *stats, void *stream) {
    return conv_fwd_impl(in, in_rows, Cin, weight_k, taps, bias, out, out_rows, Cout, idx, scale, shift, residual, relu, stats, 0, 0, stream);
}

// Same convolution with operands already in split format (rd_split_bf16): bf16x3 mode, Cout > 32, no deformable sampling of a
// split input (its bilinear blend needs fp32 rows).
extern "C" int rd_conv_fwd_split(const void *in, int in_is_split, int in_rows, int Cin, const void *weight_k, int w_is_split, int taps,
                                 const float *bias, float *out, int out_rows, int Cout, const rd_conv_index *idx, const float *scale,
                                 const float *shift, const float *residual, int relu, float *stats, void *stream) {
    RD_REQUIRE(g_conv_math == 1 && Cout > 32, "rd_conv_fwd_split: needs bf16x3 mode (rd_set_conv_math(1)) and Cout > 32");
    RD_REQUIRE(!(idx && idx->mode == 3 && in_is_split), "rd_conv_fwd_split: a deformable conv blends fp32 input rows (pass the input unsplit)");
    return conv_fwd_impl(reinterpret_cast<const float *>(in), in_rows, Cin, reinterpret_cast<const float *>(weight_k), taps, bias, out, out_rows,
                         Cout, idx, scale, shift, residual, relu, stats, in_is_split ? 1 : 0, w_is_split == 2 ? 2 : (w_is_split ? 1 : 0), stream);
}

static int conv_fwd_impl(const float *in, int in_rows, int Cin, const float *weight_k, int taps, const float *bias, float *out,
                         int out_rows, int Cout, const rd_conv_index *idx, const float *scale, const float *shift,
                         const float *residual, int relu, float *stats, int in_split, int w_split, void *stream) {
    RD_REQUIRE(Cin > 0 && Cin % KB == 0, "rd_conv_fwd: Cin=%d must be a multiple of %d", Cin, KB);
    RD_REQUIRE(Cout > 0 && in_rows >= 0 && out_rows >= 0, "rd_conv_fwd: bad sizes");
    int rc = validate_index(idx, taps, in_rows, out_rows, "rd_conv_fwd");
    if (rc) return rc;
    if (out_rows == 0) return RD_OK;
    if (g_deterministic && stats) {
        // fixed summation order: the column sums come from their own single-block pass over the finished output instead of the
        // epilogue's per-tile partials + atomics
        rc = conv_fwd_impl(in, in_rows, Cin, weight_k, taps, bias, out, out_rows, Cout, idx, scale, shift, residual, relu, nullptr, in_split,
                           w_split, stream);
        if (rc) return rc;
        RD_REQUIRE(!scale && !shift && !residual && !relu, "rd_conv_fwd: statistics are taken from the raw convolution output");
        return rd_bn_stats(out, out_rows, Cout, stats, stream);
    }
    ConvArgs a{in, in_rows, Cin, weight_k, taps, bias, out, out_rows, Cout, *idx, scale, shift, residual, relu, stats};
    a.in_split = in_split;
    a.w_split = w_split;
    a.x1 = g_mfma_single;
    hipStream_t st = S(stream);
    dim3 block(256);
    if (w_split == 2) {          // weights in fragment-major split format (RD_LAYOUT_FRAG): only the kernels that read fragments from L2
        RD_REQUIRE(g_conv_math == 1 && (conv_d3f_applies(a) || gemm_b3f_applies(a)),
                   "rd_conv_fwd_split: fragment-major weights (w_is_split = 2) need bf16x3 mode, fp32 activations and either a dense stride-1 "
                   "3x3 convolution (Cin %% 32 == 0, Cout %% 32 == 0) or a 1-tap GEMM / neighbour-table convolution (Cin %% 64 == 0, Cout %% 32 == 0); got Cin %d, Cout %d, "
                   "taps %d, mode %d", Cin, Cout, taps, idx->mode);
        if (idx->mode == 0 || taps == 1) launch_gemm_b3f(a, st);
        else launch_conv_d3f_b3(a, st);
        return check_launch("rd_conv_fwd(bf16x3, fragment-major weights)");
    }
    if (g_conv_math == 1 && Cout > 32) {
        launch_conv_b3(a, idx->mode, st);
        return check_launch("rd_conv_fwd(bf16x3)");
    }
    if (g_conv_math == 1 && launch_conv_small_b3(a, st)) return check_launch("rd_conv_fwd(bf16x3, 32-channel sparse)");
    if (g_conv_math == 1 && launch_conv_narrow_b3(a, st)) return check_launch("rd_conv_fwd(bf16x3, narrow output, split K)");
    // Tile choice: 128x128 when that already gives every CU two workgroups (512 resident blocks), otherwise 64x64 tiles
    // (4x the workgroups; operands are L2-resident at these sizes, so the extra re-reads stay on chip).
    const int64_t big_blocks = cdiv(out_rows, 128) * cdiv(Cout, 128);
    if (idx->mode == 3) {
        if (big_blocks >= 384) {
            dim3 grid(xcd_grid(cdiv(out_rows, 128), cdiv(Cout, 128)));
            k_conv_igemm<128, 128, 2, 2, true><<<grid, block, 0, st>>>(a);
        } else {
            dim3 grid(xcd_grid(cdiv(out_rows, 64), cdiv(Cout, 64)));
            k_conv_igemm<64, 64, 2, 2, true><<<grid, block, 0, st>>>(a);
        }
    } else if (Cout > 64) {
        if (big_blocks >= 384) {
            dim3 grid(xcd_grid(cdiv(out_rows, 128), cdiv(Cout, 128)));
            k_conv_igemm<128, 128, 2, 2, false><<<grid, block, 0, st>>>(a);
        } else {
            dim3 grid(xcd_grid(cdiv(out_rows, 64), cdiv(Cout, 64)));
            k_conv_igemm<64, 64, 2, 2, false><<<grid, block, 0, st>>>(a);
        }
    } else if (Cout > 32) {
        if (cdiv(out_rows, 128) >= 384) {
            dim3 grid(xcd_grid(cdiv(out_rows, 128), 1));
            k_conv_igemm<128, 64, 2, 2, false><<<grid, block, 0, st>>>(a);
        } else {
            dim3 grid(xcd_grid(cdiv(out_rows, 64), 1));
            k_conv_igemm<64, 64, 2, 2, false><<<grid, block, 0, st>>>(a);
        }
    } else {
        dim3 grid(xcd_grid(cdiv(out_rows, 128), 1));
        k_conv_igemm<128, 32, 4, 1, false><<<grid, block, 0, st>>>(a);
    }
    return check_launch("rd_conv_fwd");
}

// Data gradient on the FORWARD weights (no transposed copy): grad_in[i][c] = sum_t sum_n grad_out[src_bwd(i,t)][n] * w[n][t][c].
// idx is the backward index (transposed table / flip for sub-manifold, transposed geometry for dense convolutions).
extern "C" int rd_conv_dgrad(const float *grad_out, int out_rows, int Cout, const float *weight_k, int taps, float *grad_in, int in_rows,
                             int Cin, const rd_conv_index *idx, void *stream) {
    RD_REQUIRE(Cout > 0 && Cout % KB == 0, "rd_conv_dgrad: Cout=%d must be a multiple of %d (zero-pad narrow outputs)", Cout, KB);
    RD_REQUIRE(Cin > 0 && Cin % 4 == 0 && in_rows >= 0 && out_rows >= 0, "rd_conv_dgrad: bad sizes");
    RD_REQUIRE(idx && idx->mode != 3, "rd_conv_dgrad: deformable sampling has its own data gradient (rd_dcn_bwd_data)");
    int rc = validate_index(idx, taps, out_rows, in_rows, "rd_conv_dgrad");
    if (rc) return rc;
    if (in_rows == 0) return RD_OK;
    // GEMM view: rows = in_rows, K = Cout (forward), N = Cin (forward)
    ConvArgs a{grad_out, out_rows, Cout, weight_k, taps, nullptr, grad_in, in_rows, Cin, *idx, nullptr, nullptr, nullptr, 0, nullptr};
    hipStream_t st = S(stream);
    dim3 block(256);
    if (g_conv_math == 1 && Cin > 32) {
        launch_dgrad_b3(a, st);
        return check_launch("rd_conv_dgrad(bf16x3)");
    }
    const int64_t big_blocks = cdiv(in_rows, 128) * cdiv(Cin, 128);
    if (Cin > 64) {
        if (big_blocks >= 384) k_conv_igemm<128, 128, 2, 2, false, true><<<dim3(xcd_grid(cdiv(in_rows, 128), cdiv(Cin, 128))), block, 0, st>>>(a);
        else k_conv_igemm<64, 64, 2, 2, false, true><<<dim3(xcd_grid(cdiv(in_rows, 64), cdiv(Cin, 64))), block, 0, st>>>(a);
    } else if (Cin > 32) {
        if (cdiv(in_rows, 128) >= 384) k_conv_igemm<128, 64, 2, 2, false, true><<<dim3(xcd_grid(cdiv(in_rows, 128), 1)), block, 0, st>>>(a);
        else k_conv_igemm<64, 64, 2, 2, false, true><<<dim3(xcd_grid(cdiv(in_rows, 64), 1)), block, 0, st>>>(a);
    } else {
        k_conv_igemm<128, 32, 4, 1, false, true><<<dim3(xcd_grid(cdiv(in_rows, 128), 1)), block, 0, st>>>(a);
    }
    return check_launch("rd_conv_dgrad");
}

// ---------------------------------------------------------------------------------------------- weight gradient
// grad_w[n][t][c] += sum_j grad_out[j][n] * in[src(j,t)][c].  GEMM with M = Cout tile (128), N = Cin tile (BNW = 64 or 128), K = rows.
// Both operands are k-major in memory (rows x channels), so LDS tiles are stored [k][m] and read with ds_read_b32
// (consecutive lanes -> consecutive channels: conflict free).  Split over row chunks; fp32 atomics combine the chunks: every
// atomic wave-instruction adds two contiguous 128-byte row segments of the kernel-layout gradient (the full-rate shape).
// BNW = 128 halves the operand bytes staged per flop (the 64-wide tile is operand-delivery bound) and is used when Cin >= 128.
constexpr int WG_KB = 32;   // rows per K step
constexpr int WG_BM = 128;  // Cout tile

struct WgradArgs {
    const float *in;
    int in_rows, Cin;
    const float *go;
    int out_rows, Cout, taps;
    rd_conv_index ix;
    float *gw;
    int rows_per_block;
};

__device__ __forceinline__ int src_row_w(const WgradArgs &a, int j, int t) {
    ConvArgs c;
    c.out_rows = a.out_rows;
    c.taps = a.taps;
    c.ix = a.ix;
    return src_row(c, j, t);
}

template <bool DEFORM, int BNW>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad(const WgradArgs a) {
    constexpr int NJ = BNW / 64;   // 32-wide MFMA column tiles per wave (wave tile 64 couts x BNW/2 cins)
    constexpr int XP = BNW / 32;   // float4 loads of the input tile per thread
    __shared__ __attribute__((aligned(16))) float G_l[2][WG_KB][WG_BM + 4];  // grad_out tile [k][cout]
    __shared__ __attribute__((aligned(16))) float X_l[2][WG_KB][BNW + 4];    // gathered input tile [k][cin]
    __shared__ int s_any[2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;  // 2x2 waves
    const int tile = blockIdx.y;            // (tap, mt, nt)
    const int n_nt = (a.Cin + BNW - 1) / BNW;
    const int n_mt = (a.Cout + WG_BM - 1) / WG_BM;
    const int t = tile / (n_mt * n_nt);
    const int mt = (tile / n_nt) % n_mt, nt = tile % n_nt;
    const int co0 = mt * WG_BM, ci0 = nt * BNW;
    const int r_begin = blockIdx.x * a.rows_per_block;
    const int r_end = min(a.out_rows, r_begin + a.rows_per_block);

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // loaders: G tile 32 rows x 128 couts = 1024 float4 -> 4 / thread; X tile 32 rows x BNW cins -> XP / thread
    const int g_r = tid >> 5, g_c = (tid & 31) * 4;                              // rows g_r + 8*p, p<4
    constexpr int X_TPR = BNW / 4;                                               // threads per X row
    const int x_r = tid / X_TPR, x_c = (tid % X_TPR) * 4;                        // rows x_r + (256/X_TPR)*p, p<XP
    constexpr int X_RSTEP = 256 / X_TPR;
    f32x4 rg[4], rx[XP];
    const int n_steps = (r_end - r_begin + WG_KB - 1) / WG_KB;
    int any_next = 0, cur_store_step = 0;
    // dense geometry: every thread walks its XP rows with (b, oy, ox) cursors advanced by WG_KB rows per step -- the per-row
    // divisions by the map size would otherwise sit in front of every gather of every K step
    const bool dense = a.ix.mode == 1 || a.ix.mode == 2;
    const int t_ky = dense ? t / max(a.ix.KW, 1) : 0, t_kx = dense ? t - t_ky * a.ix.KW : 0;
    int cb[XP], cy[XP], cx[XP];
    if (dense) {
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            const int j = r_begin + x_r + X_RSTEP * p;
            cx[p] = j % a.ix.Wout;
            cy[p] = (j / a.ix.Wout) % a.ix.Hout;
            cb[p] = j / (a.ix.Wout * a.ix.Hout);
        }
    }

    auto load_tile = [&](int s) {
        const int r0 = r_begin + s * WG_KB;
        any_next = 0;
#pragma unroll
        for (int p = 0; p < XP; ++p) {
            int j = r0 + x_r + X_RSTEP * p;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (DEFORM) {
                if (j < r_end) {
                    const int64_t o = ((int64_t)j * a.taps + t) * 4;
                    const int4 q = *reinterpret_cast<const int4 *>(a.ix.samp_idx + o);
                    if (max(max(q.x, q.y), max(q.z, q.w)) >= 0) {
                        any_next = 1;
                        if (ci0 + x_c < a.Cin) {
                            const f32x4 w = *reinterpret_cast<const f32x4 *>(a.ix.samp_w + o);
                            const float *base = a.in + ci0 + x_c;
                            if (q.x >= 0) v += w[0] * *reinterpret_cast<const f32x4 *>(base + (int64_t)q.x * a.Cin);
                            if (q.y >= 0) v += w[1] * *reinterpret_cast<const f32x4 *>(base + (int64_t)q.y * a.Cin);
                            if (q.z >= 0) v += w[2] * *reinterpret_cast<const f32x4 *>(base + (int64_t)q.z * a.Cin);
                            if (q.w >= 0) v += w[3] * *reinterpret_cast<const f32x4 *>(base + (int64_t)q.w * a.Cin);
                        }
                    }
                }
            } else {
                int src = -1;
                if (j < r_end) src = dense ? src_row_dense_k(a.ix, cb[p], cy[p], cx[p], t_ky, t_kx) : src_row_w(a, j, t);
                if (dense) {  // advance the cursor to this thread's row of the next K step
                    cx[p] += WG_KB;
                    while (cx[p] >= a.ix.Wout) {
                        cx[p] -= a.ix.Wout;
                        if (++cy[p] == a.ix.Hout) {
                            cy[p] = 0;
                            ++cb[p];
                        }
                    }
                }
                if (src >= 0) {
                    any_next = 1;
                    if (ci0 + x_c < a.Cin) v = *reinterpret_cast<const f32x4 *>(a.in + (int64_t)src * a.Cin + ci0 + x_c);
                }
            }
            rx[p] = v;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int j = r0 + g_r + 8 * p;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j < r_end) {
                int co = co0 + g_c;
                const float *src = a.go + (int64_t)j * a.Cout + co;
                if (co + 3 < a.Cout && (a.Cout & 3) == 0) v = *reinterpret_cast<const f32x4 *>(src);
                else {
                    if (co + 0 < a.Cout) v[0] = src[0];
                    if (co + 1 < a.Cout) v[1] = src[1];
                    if (co + 2 < a.Cout) v[2] = src[2];
                    if (co + 3 < a.Cout) v[3] = src[3];
                }
            }
            rg[p] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int p = 0; p < XP; ++p) *reinterpret_cast<f32x4 *>(&X_l[buf][x_r + X_RSTEP * p][x_c]) = rx[p];
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(&G_l[buf][g_r + 8 * p][g_c]) = rg[p];
        if (any_next) s_any[buf] = cur_store_step;   // tag = 1 + index of the K step this tile belongs to (no reset pass needed)
    };

    if (tid < 2) s_any[tid] = 0;
    __syncthreads();
    if (n_steps > 0) {
        load_tile(0);
        cur_store_step = 1;
        store_tile(0);
    }
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int s = 0; s < n_steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < n_steps) load_tile(s + 1);
        if (s_any[buf] == s + 1) {  // block-uniform: skip K steps whose 32 rows have no source for this tap
            // operands of k-pair kk+1 are fetched from LDS while the MFMAs of k-pair kk run (explicit register double buffer:
            // the compiler otherwise re-uses one register set and waits for LDS in front of every group of MFMAs)
            float av[2][2], bv[2][NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[0][i] = G_l[buf][fh][wm * 64 + i * 32 + fr];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[0][j] = X_l[buf][fh][wn * (BNW / 2) + j * 32 + fr];
#pragma unroll
            for (int kk = 0; kk < WG_KB / 2; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < WG_KB / 2) {
                    const int k = (kk + 1) * 2 + fh;
#pragma unroll
                    for (int i = 0; i < 2; ++i) av[nxt][i] = G_l[buf][k][wm * 64 + i * 32 + fr];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bv[nxt][j] = X_l[buf][k][wn * (BNW / 2) + j * 32 + fr];
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads above in front of this group of MFMAs
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
            }
        }
        if (s + 1 < n_steps) {      // the other buffer was last read in step s-1, which ended with the barrier below
            cur_store_step = s + 2;
            store_tile(buf ^ 1);
        }
        __syncthreads();
    }
    // accumulate into grad_w[co][t][ci]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ci = ci0 + wn * (BNW / 2) + j * 32 + fr;
        if (ci < a.Cin) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    float v = acc[i][j][r];
                    if (co < a.Cout && v != 0.f) atomicAdd(&a.gw[((int64_t)co * a.taps + t) * a.Cin + ci], v);
                }
        }
    }
}

bool launch_wgrad_d3_b3(const float *in, int in_rows, int Cin, const float *go, int out_rows, int Cout, int taps, const rd_conv_index *ix, float *gw,
                        hipStream_t st);

static int conv_wgrad_impl(const float *in, int in_rows, int Cin, const float *grad_out, int out_rows, int Cout, int taps,
                           const rd_conv_index *idx, float *grad_wk, int in_split, int go_split, void *stream);

extern "C" int rd_conv_wgrad(const float *in, int in_rows, int Cin, const float *grad_out, int out_rows, int Cout, int taps,
                             const rd_conv_index *idx, float *grad_wk, void *stream) {
    return conv_wgrad_impl(in, in_rows, Cin, grad_out, out_rows, Cout, taps, idx, grad_wk, 0, 0, stream);
}

// Weight gradient with operands already in split format (bf16x3 mode, Cout >= 64, Cin >= 64, Cout % 4 == 0).
extern "C" int rd_conv_wgrad_split(const void *in, int in_is_split, int in_rows, int Cin, const void *grad_out, int go_is_split, int out_rows,
                                   int Cout, int taps, const rd_conv_index *idx, float *grad_wk, void *stream) {
    RD_REQUIRE(g_conv_math == 1 && Cout >= 64 && Cin >= 64 && Cout % 4 == 0, "rd_conv_wgrad_split: needs bf16x3 mode, Cout >= 64 (multiple of 4), Cin >= 64");
    RD_REQUIRE(!(idx && idx->mode == 3 && in_is_split), "rd_conv_wgrad_split: a deformable conv blends fp32 input rows (pass the input unsplit)");
    return conv_wgrad_impl(reinterpret_cast<const float *>(in), in_rows, Cin, reinterpret_cast<const float *>(grad_out), out_rows, Cout, taps, idx,
                           grad_wk, in_is_split ? 1 : 0, go_is_split ? 1 : 0, stream);
}

static int conv_wgrad_impl(const float *in, int in_rows, int Cin, const float *grad_out, int out_rows, int Cout, int taps,
                           const rd_conv_index *idx, float *grad_wk, int in_split, int go_split, void *stream) {
    RD_REQUIRE(Cin > 0 && Cin % 32 == 0, "rd_conv_wgrad: Cin=%d must be a multiple of 32", Cin);
    RD_REQUIRE(Cout > 0, "rd_conv_wgrad: bad Cout");
    int rc = validate_index(idx, taps, in_rows, out_rows, "rd_conv_wgrad");
    if (rc) return rc;
    if (out_rows == 0) return RD_OK;
    // bf16x3 mode: every shape with Cout >= 64 and Cin >= 64 runs the split-bf16 kernel (Cin tile 128, or 64 when Cin == 64).
    // Exact fp32: 128-wide Cin tiles when that still leaves >= 32 (tap, tile) pairs to spread over the chip (3x3 convs); 1x1 keep 64.
    const bool b3 = g_conv_math == 1 && Cout >= 64 && Cin >= 64;
    // dense stride-1 3x3 layers: the halo-staged kernel (conv_wgrad_d3.hip) runs all nine taps on one staged pixel tile
    if (g_conv_math == 1 && !in_split && !go_split && launch_wgrad_d3_b3(in, in_rows, Cin, grad_out, out_rows, Cout, taps, idx, grad_wk, S(stream)))
        return check_launch("rd_conv_wgrad(bf16x3, halo)");
    const bool wide = b3 ? Cin >= 128 : (Cin >= 128 && Cout >= 64 && (int64_t)taps * cdiv(Cout, WG_BM) * cdiv(Cin, 128) >= 32);
    const int bn = wide ? 128 : 64;
    const int n_mt = (int)cdiv(Cout, WG_BM), n_nt = (int)cdiv(Cin, bn);
    const int tiles = taps * n_mt * n_nt;
    // Row chunks.  512 workgroups are resident at once (two per CU); a launch runs in ceil(chunks*tiles/512) rounds of
    // rows/chunks rows each, and every block ends with an atomic pass over its tile (memory-side atomics, ~1.3 TB/s chip-wide:
    // ~50k cycles per round for 128x128 tiles).  Pick the chunk count with the smallest modelled time; >= 256 rows per block.
    // (bf16x3 spends about a third of the fp32 cycles per row.)
    int64_t chunks = 1;
    if (taps == 1 && !b3) {  // 1x1 convs: little work per (tile, chunk), latency bound -> more, smaller blocks (measured 75 us vs 102 us)
        chunks = std::max<int64_t>(1, std::min<int64_t>(cdiv(out_rows, 256), cdiv(2048, tiles)));
    } else {
        const int64_t max_chunks = std::max<int64_t>(1, std::min<int64_t>(cdiv(out_rows, 256), 128));
        const double row_cycles = (b3 ? 0.7 : 2.0) * bn;
        double best = 1e30;
        for (int64_t c = 1; c <= max_chunks; ++c) {
            const double rounds = (double)cdiv(c * tiles, 512);
            const double cost = rounds * ((double)cdiv(out_rows, c) * row_cycles + 50000.0 * bn / 128.0);
            if (cost < best) { best = cost; chunks = c; }
        }
    }
    if (g_deterministic) chunks = 1;     // every (tap, tile) accumulates all rows in one block: no atomics between row chunks
    int rows_per_block = (int)(cdiv(cdiv(out_rows, chunks), WG_KB) * WG_KB);
    chunks = cdiv(out_rows, rows_per_block);
    hipStream_t st = S(stream);
    if (b3) {
        launch_wgrad_b3(in, in_rows, Cin, grad_out, out_rows, Cout, taps, idx, grad_wk, rows_per_block, chunks, tiles, bn, in_split, go_split, st);
        return check_launch("rd_conv_wgrad(bf16x3)");
    }
    WgradArgs a{in, in_rows, Cin, grad_out, out_rows, Cout, taps, *idx, grad_wk, rows_per_block};
    dim3 grid((unsigned)chunks, (unsigned)tiles);
    if (idx->mode == 3) {
        if (wide) k_conv_wgrad<true, 128><<<grid, 256, 0, st>>>(a);
        else k_conv_wgrad<true, 64><<<grid, 256, 0, st>>>(a);
    } else {
        if (wide) k_conv_wgrad<false, 128><<<grid, 256, 0, st>>>(a);
        else k_conv_wgrad<false, 64><<<grid, 256, 0, st>>>(a);
    }
    return check_launch("rd_conv_wgrad");
}

// source index of destination element i (i enumerates the DESTINATION linearly) for every layout kind
__device__ __forceinline__ int64_t layout_src(int64_t i, int Cout, int Cin, int taps, int kind, int flip) {
    if (kind == 0) {  // [Cout][taps][Cin] -> same, optional tap flip
        int c = (int)(i % Cin), t = (int)((i / Cin) % taps), n = (int)(i / ((int64_t)Cin * taps));
        int ts = flip ? taps - 1 - t : t;
        return ((int64_t)n * taps + ts) * Cin + c;
    } else if (kind == 1) {  // torch conv [Cout][Cin][taps] -> [Cout][taps][Cin]
        int c = (int)(i % Cin), t = (int)((i / Cin) % taps), n = (int)(i / ((int64_t)Cin * taps));
        int ts = flip ? taps - 1 - t : t;
        return ((int64_t)n * Cin + c) * taps + ts;
    } else if (kind == 2) {  // kernel layout [Cout][taps][Cin] -> [Cin][taps][Cout]
        int n = (int)(i % Cout), t = (int)((i / Cout) % taps), c = (int)(i / ((int64_t)Cout * taps));
        int ts = flip ? taps - 1 - t : t;
        return ((int64_t)n * taps + ts) * Cin + c;
    } else if (kind == 3) {  // torch ConvTranspose2d [Cin][Cout][taps] -> [Cout][taps][Cin]
        int c = (int)(i % Cin), t = (int)((i / Cin) % taps), n = (int)(i / ((int64_t)Cin * taps));
        int ts = flip ? taps - 1 - t : t;
        return ((int64_t)c * Cout + n) * taps + ts;
    } else if (kind == 4) {  // [Cout][taps][Cin] -> torch conv [Cout][Cin][taps]
        int t = (int)(i % taps), c = (int)((i / taps) % Cin), n = (int)(i / ((int64_t)Cin * taps));
        int ts = flip ? taps - 1 - t : t;
        return ((int64_t)n * taps + ts) * Cin + c;
    } else if (kind == 7 || kind == 8) {  // torch conv [Cout][Cin][taps] (7) / ConvTranspose2d [Cin][Cout][taps] (8) -> [Cin][taps][Cout]
        int n = (int)(i % Cout), t = (int)((i / Cout) % taps), c = (int)(i / ((int64_t)Cout * taps));
        int ts = flip ? taps - 1 - t : t;
        return kind == 7 ? ((int64_t)n * Cin + c) * taps + ts : ((int64_t)c * Cout + n) * taps + ts;
    } else if (kind == 6) {  // kernel layout [Cout][taps][Cin] -> [taps][Cin][Cout]  (DCN column-gradient operand)
        int n = (int)(i % Cout), c = (int)((i / Cout) % Cin), t = (int)(i / ((int64_t)Cout * Cin));
        return ((int64_t)n * taps + t) * Cin + c;
    } else {  // kind 5: [Cout][taps][Cin] -> torch ConvTranspose2d [Cin][Cout][taps]
        int t = (int)(i % taps), n = (int)((i / taps) % Cout), c = (int)(i / ((int64_t)Cout * taps));
        int ts = flip ? taps - 1 - t : t;
        return ((int64_t)n * taps + ts) * Cin + c;
    }
}

__global__ void k_weight_layout(const float *src, float *dst, int Cout, int Cin, int taps, int kind, int flip) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Cout * Cin * taps) return;
    dst[i] = src[layout_src(i, Cout, Cin, taps, kind, flip)];
}

// same transform, destination written in split format (rd_split_bf16): 4 consecutive destination elements per thread
__global__ void k_weight_layout_split(const float *src, uint2 *dst_hi_lo, int Cout, int Cin, int taps, int kind, int flip) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g * 4 >= (int64_t)Cout * Cin * taps) return;
    unsigned short hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float v = src[layout_src(g * 4 + e, Cout, Cin, taps, kind, flip)];
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        hi[e] = __builtin_bit_cast(unsigned short, h);
        lo[e] = __builtin_bit_cast(unsigned short, l);
    }
    dst_hi_lo[2 * g] = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
    dst_hi_lo[2 * g + 1] = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
}

// FRAGMENT-MAJOR split format (kind | RD_LAYOUT_FRAG): the destination [A][taps][B] (B = the GEMM's K axis) is cut into blocks of
// 32 (A) x 16 (B) of one tap, and a block is stored as the two 1-KiB images a wavefront feeds to v_mfma_f32_32x32x16_bf16 as its B
// operand -- first the hi parts, then the lo parts, each as 64 x 16 bytes: lane l = 32 * kh + r holds A-row r, B-elements 8 kh .. 8 kh + 7.
// A wave therefore fetches an operand fragment with ONE fully coalesced 16-byte-per-lane load straight from L2, no LDS staging:
//   unit16(a, t, b, part) = (((a >> 5) * taps + t) * (B >> 4) + (b >> 4)) * 128 + part * 64 + ((b >> 3) & 1) * 32 + (a & 31)
// Same size as the fp32 tensor.  A % 32 == 0, B % 16 == 0.
__device__ __forceinline__ int64_t frag_unit(int a, int t, int b, int taps, int B) {
    return ((((int64_t)(a >> 5) * taps + t) * (B >> 4) + (b >> 4)) << 7) + (((b >> 3) & 1) << 5) + (a & 31);
}

__global__ void k_weight_layout_frag(const float *src, uint4 *dst, int Cout, int Cin, int taps, int kind, int flip) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // one thread per 8 consecutive destination elements
    if (g * 8 >= (int64_t)Cout * Cin * taps) return;
    const bool b_is_cin = kind == 0 || kind == 1 || kind == 3;
    const int A = b_is_cin ? Cout : Cin, B = b_is_cin ? Cin : Cout;
    const int64_t i0 = g * 8;
    const int b = (int)(i0 % B), t = (int)((i0 / B) % taps), a = (int)(i0 / ((int64_t)B * taps));
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = src[layout_src(i0 + e, Cout, Cin, taps, kind, flip)];
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        hi[e] = __builtin_bit_cast(unsigned short, h);
        lo[e] = __builtin_bit_cast(unsigned short, l);
    }
    const int64_t u = frag_unit(a, t, b, taps, B);
    (void)A;
    dst[u] = make_uint4(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16), hi[4] | ((unsigned)hi[5] << 16), hi[6] | ((unsigned)hi[7] << 16));
    dst[u + 64] = make_uint4(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16), lo[4] | ((unsigned)lo[5] << 16), lo[6] | ((unsigned)lo[7] << 16));
}

// destination in split format: kinds whose destination's fastest axis has a multiple-of-4 extent (the K axis of the consuming GEMM)
extern "C" int rd_weight_layout_split(const float *src, void *dst, int Cout, int Cin, int taps, int kind, int flip, void *stream) {
    if (kind & RD_LAYOUT_FRAG) {
        const int k = kind & ~RD_LAYOUT_FRAG;
        RD_REQUIRE((k >= 0 && k <= 3) || k == 7 || k == 8, "rd_weight_layout_split: kinds 0..3, 7, 8 (operand layouts) only, got %d", k);
        const bool b_is_cin = k == 0 || k == 1 || k == 3;
        RD_REQUIRE((b_is_cin ? Cout : Cin) % 32 == 0 && (b_is_cin ? Cin : Cout) % 16 == 0,
                   "rd_weight_layout_split: fragment-major needs the slow axis %% 32 == 0 and the fast (K) axis %% 16 == 0 (Cout %d, Cin %d, kind %d)", Cout, Cin, k);
        const int64_t total = (int64_t)Cout * Cin * taps;
        if (total <= 0) return RD_OK;
        k_weight_layout_frag<<<cdiv(total / 8, 256), 256, 0, S(stream)>>>(src, reinterpret_cast<uint4 *>(dst), Cout, Cin, taps, k, flip);
        return check_launch("rd_weight_layout_split(fragment-major)");
    }
    RD_REQUIRE((kind >= 0 && kind <= 3) || kind == 7 || kind == 8, "rd_weight_layout_split: kinds 0..3, 7, 8 (operand layouts) only, got %d", kind);
    RD_REQUIRE((kind == 2 || kind >= 7 ? Cout : Cin) % 4 == 0, "rd_weight_layout_split: the destination's fastest axis must be a multiple of 4");
    int64_t total = (int64_t)Cout * Cin * taps;
    if (total <= 0) return RD_OK;
    k_weight_layout_split<<<cdiv(total / 4, 256), 256, 0, S(stream)>>>(src, reinterpret_cast<uint2 *>(dst), Cout, Cin, taps, kind, flip);
    return check_launch("rd_weight_layout_split");
}

// Many weights in ONE launch (the per-step refresh of every trainable conv weight's GEMM operands), LDS-tiled.
// Every operand layout is dst[A][taps][B] (B fastest, split format in groups of 4 along B) read from a source whose fastest axis is
// the tap (torch layouts), B (kind 0) or A (kind 2): element (a, t, b) sits at src[a*sa + t*st + b*sb].  The first version walked the
// DESTINATION linearly and gathered single floats at stride `taps` or `Cin*taps` -- PMC: 1.12 GB of fetch traffic for 0.30 GB of
// algorithmic bytes (3.7x).  Here work item `chunk_group[c]` of job `chunk_job[c]` is a 16 (A) x 64 (B) x <= 9 (taps) tile: it is
// read in SOURCE order (consecutive threads on consecutive source addresses: runs of 64 x taps or 16 x taps floats), parked in LDS,
// and written in destination order as 16-byte split groups (8 threads = one 128-byte segment, 16 segments per row of the tile).
constexpr int WL_TA = 16, WL_TB = 64, WL_TT = 9, WL_LD = WL_TB + 4;

__device__ __forceinline__ void layout_strides(int kind, int Cout, int Cin, int taps, int &A, int &B, int64_t &sa, int64_t &st, int64_t &sb) {
    switch (kind) {
    case 0: A = Cout; B = Cin; sa = (int64_t)taps * Cin; st = Cin; sb = 1; break;                  // [Cout][taps][Cin] -> same
    case 1: A = Cout; B = Cin; sa = (int64_t)Cin * taps; st = 1; sb = taps; break;                 // torch conv [Cout][Cin][taps]
    case 2: A = Cin; B = Cout; sa = 1; st = Cin; sb = (int64_t)taps * Cin; break;                  // [Cout][taps][Cin] -> [Cin][taps][Cout]
    case 3: A = Cout; B = Cin; sa = taps; st = 1; sb = (int64_t)Cout * taps; break;                // ConvTranspose2d [Cin][Cout][taps]
    case 7: A = Cin; B = Cout; sa = taps; st = 1; sb = (int64_t)Cin * taps; break;                 // torch conv -> [Cin][taps][Cout]
    default: A = Cin; B = Cout; sa = (int64_t)Cout * taps; st = 1; sb = taps; break;               // 8: ConvTranspose2d -> [Cin][taps][Cout]
    }
}

__global__ __launch_bounds__(256) void k_weight_layout_split_multi(const rd_layout_job *__restrict__ jobs, const int *__restrict__ chunk_job,
                                                                   const int *__restrict__ chunk_group) {
    __shared__ __attribute__((aligned(16))) float tile[WL_TA * WL_TT * WL_LD];
    rd_layout_job j = jobs[chunk_job[blockIdx.x]];
    const bool frag = (j.kind & RD_LAYOUT_FRAG) != 0;          // fragment-major destination (see k_weight_layout_frag)
    j.kind &= ~RD_LAYOUT_FRAG;
    int A, B;
    int64_t sa, st, sb;
    layout_strides(j.kind, j.Cout, j.Cin, j.taps, A, B, sa, st, sb);
    const int n_b = (B + WL_TB - 1) / WL_TB, n_tc = (j.taps + WL_TT - 1) / WL_TT;
    int item = chunk_group[blockIdx.x];
    const int tc = item % n_tc;
    item /= n_tc;
    const int a0 = (item / n_b) * WL_TA, b0 = (item % n_b) * WL_TB, t0 = tc * WL_TT;
    const int ta = min(WL_TA, A - a0), tb = min(WL_TB, B - b0), tt = min(WL_TT, j.taps - t0);
    if (ta <= 0 || tb <= 0) return;
    // ---- read in source order: innermost = the axis with the smallest source stride
    const int n_el = ta * tt * tb;
    if (st == 1) {                       // taps innermost; then whichever of a / b has stride `taps`
        const bool b_mid = sb == j.taps;
        const int mid = b_mid ? tb : ta;
        for (int e = threadIdx.x; e < n_el; e += 256) {
            const int t = e % tt, r = e / tt, m = r % mid, o = r / mid;
            const int a = b_mid ? o : m, b = b_mid ? m : o;
            tile[(a * WL_TT + t) * WL_LD + b] = j.src[(a0 + a) * sa + (t0 + t) * st + (b0 + b) * sb];
        }
    } else if (sb == 1) {                // kind 0: b innermost, then t, then a
        for (int e = threadIdx.x; e < n_el; e += 256) {
            const int b = e % tb, r = e / tb, t = r % tt, a = r / tt;
            tile[(a * WL_TT + t) * WL_LD + b] = j.src[(a0 + a) * sa + (t0 + t) * st + (b0 + b) * sb];
        }
    } else {                             // kind 2: a innermost, then t, then b
        for (int e = threadIdx.x; e < n_el; e += 256) {
            const int a = e % ta, r = e / ta, t = r % tt, b = r / tt;
            tile[(a * WL_TT + t) * WL_LD + b] = j.src[(a0 + a) * sa + (t0 + t) * st + (b0 + b) * sb];
        }
    }
    __syncthreads();
    if (frag) {          // groups of 8 along b -> one 16-byte hi unit and one lo unit (consecutive a = consecutive units: 16 x 16 B runs)
        uint4 *dstf = reinterpret_cast<uint4 *>(j.dst);
        const int gb8 = tb / 8, n_g8 = ta * tt * gb8;
        for (int g = threadIdx.x; g < n_g8; g += 256) {
            const int a = g % ta, r = g / ta, b8 = r % gb8, t = r / gb8;
            const float *v = &tile[(a * WL_TT + t) * WL_LD + 8 * b8];
            unsigned short hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 h = (__bf16)v[e];
                const __bf16 l = (__bf16)(v[e] - (float)h);
                hi[e] = __builtin_bit_cast(unsigned short, h);
                lo[e] = __builtin_bit_cast(unsigned short, l);
            }
            const int64_t u = frag_unit(a0 + a, t0 + t, b0 + 8 * b8, j.taps, B);
            dstf[u] = make_uint4(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16), hi[4] | ((unsigned)hi[5] << 16), hi[6] | ((unsigned)hi[7] << 16));
            dstf[u + 64] = make_uint4(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16), lo[4] | ((unsigned)lo[5] << 16), lo[6] | ((unsigned)lo[7] << 16));
        }
        return;
    }
    // ---- write in destination order: groups of 4 along b
    uint2 *dst = reinterpret_cast<uint2 *>(j.dst);
    const int gb = tb / 4, n_g = ta * tt * gb;
    for (int g = threadIdx.x; g < n_g; g += 256) {
        const int b4 = g % gb, r = g / gb, t = r % tt, a = r / tt;
        const f32x4 v = *reinterpret_cast<const f32x4 *>(&tile[(a * WL_TT + t) * WL_LD + 4 * b4]);
        unsigned short hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 h = (__bf16)v[e];
            const __bf16 l = (__bf16)(v[e] - (float)h);
            hi[e] = __builtin_bit_cast(unsigned short, h);
            lo[e] = __builtin_bit_cast(unsigned short, l);
        }
        const int64_t G = (((int64_t)(a0 + a) * j.taps + t0 + t) * B + b0) / 4 + b4;
        dst[2 * G] = make_uint2(hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16));
        dst[2 * G + 1] = make_uint2(lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16));
    }
}

// number of work items (16 x 64 x <= 9 tiles) of one job: the host builds chunk_job / chunk_group from it
extern "C" int rd_weight_layout_split_items(int Cout, int Cin, int taps, int kind) {
    kind &= ~RD_LAYOUT_FRAG;
    const int A = (kind == 0 || kind == 1 || kind == 3) ? Cout : Cin, B = (kind == 0 || kind == 1 || kind == 3) ? Cin : Cout;
    return ((A + WL_TA - 1) / WL_TA) * ((B + WL_TB - 1) / WL_TB) * ((taps + WL_TT - 1) / WL_TT);
}

extern "C" int rd_weight_layout_split_multi(const rd_layout_job *jobs_dev, const int *chunk_job_dev, const int *chunk_group_dev, int n_chunks,
                                            void *stream) {
    RD_REQUIRE(n_chunks >= 0, "rd_weight_layout_split_multi: bad chunk count");
    if (n_chunks == 0) return RD_OK;
    RD_REQUIRE(jobs_dev && chunk_job_dev && chunk_group_dev, "rd_weight_layout_split_multi: null table");
    k_weight_layout_split_multi<<<n_chunks, 256, 0, S(stream)>>>(jobs_dev, chunk_job_dev, chunk_group_dev);
    return check_launch("rd_weight_layout_split_multi");
}

// Up to RD_LAYOUT_MULTI_MAX re-layouts in ONE launch, the job table passed by value in the kernel arguments (no device copy of a
// descriptor table): the ~48 weight gradients of a backward pass leave the GEMMs in kernel layout [Cout][taps][Cin] and all go to
// their parameters' torch layouts together at the end of the pass.  Block b serves the job whose element range contains b * 1024.
struct LayoutMulti {
    rd_layout_job jobs[RD_LAYOUT_MULTI_MAX];
    int first_block[RD_LAYOUT_MULTI_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void k_weight_layout_multi(const LayoutMulti t) {
    int j = 0;
    while (j + 1 < t.n && (int)blockIdx.x >= t.first_block[j + 1]) ++j;
    const rd_layout_job job = t.jobs[j];
    const int64_t total = (int64_t)job.Cout * job.Cin * job.taps;
    const float *src = reinterpret_cast<const float *>(job.src);
    float *dst = reinterpret_cast<float *>(job.dst);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = ((int64_t)(blockIdx.x - t.first_block[j]) * 4 + u) * 256 + threadIdx.x;
        if (i < total) dst[i] = src[layout_src(i, job.Cout, job.Cin, job.taps, job.kind, 0)];
    }
}

extern "C" int rd_weight_layout_multi(const rd_layout_job *jobs_host, int n_jobs, void *stream) {
    RD_REQUIRE(n_jobs >= 0 && n_jobs <= RD_LAYOUT_MULTI_MAX, "rd_weight_layout_multi: at most %d jobs per call", RD_LAYOUT_MULTI_MAX);
    if (n_jobs == 0) return RD_OK;
    LayoutMulti t;
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const rd_layout_job &job = jobs_host[j];
        RD_REQUIRE(job.kind >= 0 && job.kind <= 8 && job.src && job.dst && job.Cout > 0 && job.Cin > 0 && job.taps > 0, "rd_weight_layout_multi: bad job %d", j);
        t.jobs[j] = job;
        t.first_block[j] = blocks;
        blocks += (int)cdiv((int64_t)job.Cout * job.Cin * job.taps, 1024);
    }
    t.first_block[n_jobs] = blocks;
    t.n = n_jobs;
    k_weight_layout_multi<<<blocks, 256, 0, S(stream)>>>(t);
    return check_launch("rd_weight_layout_multi");
}

extern "C" int rd_weight_layout(const float *src, float *dst, int Cout, int Cin, int taps, int kind, int flip, void *stream) {
    RD_REQUIRE(kind >= 0 && kind <= 8, "rd_weight_layout: bad kind %d", kind);
    int64_t total = (int64_t)Cout * Cin * taps;
    if (total <= 0) return RD_OK;
    k_weight_layout<<<cdiv(total, 256), 256, 0, S(stream)>>>(src, dst, Cout, Cin, taps, kind, flip);
    return check_launch("rd_weight_layout");
}
