// Low-precision DenseEnc path (BASELINE configs[2] and [4]): the frozen teacher's dense BEV convolutions with bf16 or OCP fp8
// (e4m3fn) STORAGE and fp32 accumulation on the matrix cores.  See include/rdamd.h section Q.
//
// Reference graph: BaseBEVBackboneV2 (pcdet/models/backbones_2d/base_bev_backbone.py:206-308) in eval mode -- ZeroPad+Conv3x3 /
// Conv3x3(p1) + BatchNorm(eps 1e-3) + ReLU stacks and one ConvTranspose2d(k2, s2) + BatchNorm + ReLU, all 256 channels: every
// layer is conv -> per-channel affine -> ReLU, so BatchNorm folds into the epilogue and activations can stay in the narrow format
// from the first layer to the last (one HBM round trip per layer at 2 or 1 bytes per element instead of 4).
//
//   k_lp_conv<DT, KS, ...>   dense stride-1 KS x KS convolution (KS = 3: pad 1; KS = 1 also serves the 2x2 stride-2 transposed
//                            convolution as four 1x1 products scattered to the (2y+dy, 2x+dx) output pixels), channels-last rows.
//                            A (TY x TX) pixel tile's halo is staged ONCE per 64-byte channel chunk and the 9 taps are walked by
//                            constant LDS offsets (as k_conv_d3_b3, but the operands already ARE MFMA operands: no split, no VALU
//                            work per element); weights stream per (chunk, tap) through two LDS buffers, prefetched one step ahead.
//     DT = bf16: v_mfma_f32_32x32x16_bf16, 32 channels per chunk (2 MFMAs per chunk and 32x32 tile)
//     DT = fp8 : v_mfma_f32_32x32x64_f8f6f4 (e4m3 x e4m3, unit block scales), 64 channels per chunk, ONE MFMA of twice the cycles:
//                2x the bf16 rate per clock (MI355X_MICROARCH.md, Matrix cores: the non-scaled K=16 fp8 form only runs at bf16 rate)
//   epilogue: out = act(acc * alpha[co] + beta[co]) written as bf16 / fp8 / fp32.  alpha folds the activation scale, the
//   per-output-channel weight scale, the BatchNorm scale and the next layer's activation scale; beta the BatchNorm shift.
//   k_lp_cast / k_lp_uncast / k_lp_quant_weights / k_lp_amax: conversions and the calibration reduction.
// Bounds: MFMA (dense bf16 2.5 PF / fp8 5 PF); bytes per layer = rows * (Cin + Cout) * elt + 9 * Cin * Cout * elt.
#include <algorithm>
#include "conv_common.hpp"

using namespace rd;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int LP_BF16 = 0, LP_FP8 = 1, LP_F32 = 2;
constexpr float FP8_MAX = 448.f;   // largest finite e4m3fn value

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -FP8_MAX), FP8_MAX); b = fminf(fmaxf(b, -FP8_MAX), FP8_MAX);
    c = fminf(fmaxf(c, -FP8_MAX), FP8_MAX); d = fminf(fmaxf(d, -FP8_MAX), FP8_MAX);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
    return (unsigned)r;
}
__device__ __forceinline__ unsigned char one_fp8(float a) { return (unsigned char)(pack4_fp8(a, 0.f, 0.f, 0.f) & 0xffu); }
__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

struct LpConvArgs {
    const unsigned char *in;    // rows x in_ld elements of the narrow type; the convolution reads channels [0, Cin)
    int B, H, W, Cin, in_ld;
    const unsigned char *w;     // [Cout][taps][Cin], narrow type
    int taps;
    const float *alpha, *beta;  // [Cout]
    void *out;
    int Cout, out_ld, out_col0, relu;
    int deconv;                 // 1: tap = blockIdx.y of a 2x2 stride-2 transposed convolution, output map (2H, 2W)
};

// DT: operand type; KS: 3 (pad 1) or 1; ODT: output type.  256 threads = 2 x 2 waves over a (TY*TX pixels) x BN tile.
template <int DT, int KS, int TY, int TX, int BN, int ODT>
__global__ __launch_bounds__(256, 2) void k_lp_conv(const LpConvArgs a) {
    constexpr int ESZ = DT == LP_BF16 ? 2 : 1;          // bytes per element
    constexpr int BM = TY * TX, HX = TX + KS - 1, HY = TY + KS - 1, HR = HX * HY;
    constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
    constexpr int AROW = 80;                            // halo row: 64 data bytes + 16 pad (conflict-free 16-byte fragment reads)
    constexpr int A_BYTES = HR * AROW, B_BYTES = BN * 64;
    constexpr int HL = (HR * 4 + 255) / 256, BL = (BN * 4 + 255) / 256;
    constexpr int TAPS = KS * KS;
    static_assert(MI >= 1 && NI >= 1, "wave tile at least 32 x 32");
    __shared__ __attribute__((aligned(16))) unsigned char lds[A_BYTES + 2 * B_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, fr = lane & 31, fh = lane >> 5;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + TY - 1) / TY;
    const int n_row_tiles = a.B * tiles_y * tiles_x;
    int row_tile, col_tile;
    if (!xcd_tile(n_row_tiles, (a.Cout + BN - 1) / BN, row_tile, col_tile)) return;
    const int b = row_tile / (tiles_y * tiles_x), y0 = ((row_tile / tiles_x) % tiles_y) * TY, x0 = (row_tile % tiles_x) * TX;
    const int n0 = col_tile * BN;
    const int dtap = a.deconv ? (int)blockIdx.y : 0;
    const int chunks = a.Cin * ESZ / 64;
    const int64_t w_row_bytes = (int64_t)a.taps * a.Cin * ESZ;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int aoff[MI];                                       // byte offset of this lane's pixel row at tap (0, 0)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int p = wm * WM + i * 32 + fr;
        aoff[i] = ((p / TX) * HX + (p % TX)) * AROW;
    }
    const int bswz = (fr >> 2) & 3;

    u32x4 ra[HL], rb[BL];
    auto load_halo = [&](int kq) {
#pragma unroll
        for (int q = 0; q < HL; ++q) {
            const int e = tid + 256 * q, hr = e >> 2, piece = e & 3;
            const int gy = y0 - KS / 2 + hr / HX, gx = x0 - KS / 2 + hr % HX;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (e < HR * 4 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const u32x4 *>(a.in + ((int64_t)(b * H + gy) * W + gx) * a.in_ld * ESZ + kq * 64 + piece * 16);
            ra[q] = v;
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int q = 0; q < HL; ++q) {
            const int e = tid + 256 * q;
            if (e < HR * 4) *reinterpret_cast<u32x4 *>(lds + (e >> 2) * AROW + (e & 3) * 16) = ra[q];
        }
    };
    auto load_B = [&](int kq, int tap) {
#pragma unroll
        for (int q = 0; q < BL; ++q) {
            const int e = tid + 256 * q, row = e >> 2, piece = e & 3;
            const int n = n0 + row;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (e < BN * 4 && n < a.Cout)
                v = *reinterpret_cast<const u32x4 *>(a.w + n * w_row_bytes + ((int64_t)tap * a.Cin) * ESZ + kq * 64 + piece * 16);
            rb[q] = v;
        }
    };
    auto store_B = [&](unsigned char *buf) {
#pragma unroll
        for (int q = 0; q < BL; ++q) {
            const int e = tid + 256 * q, row = e >> 2, piece = e & 3;
            if (e < BN * 4) *reinterpret_cast<u32x4 *>(buf + row * 64 + ((piece ^ ((row >> 2) & 3)) << 4)) = rb[q];
        }
    };

    unsigned char *bcur = lds + A_BYTES, *bnext = bcur + B_BYTES;
    const int S = chunks * TAPS;
    auto tap_of = [&](int g) { return a.deconv ? dtap : g; };
    load_halo(0);
    load_B(0, tap_of(0));
    store_halo();
    store_B(bcur);
    if (S > 1) load_B(TAPS > 1 ? 0 : 1, tap_of(TAPS > 1 ? 1 : 0));
    __syncthreads();

    for (int kq = 0; kq < chunks; ++kq) {
        const bool more = kq + 1 < chunks;
        if (more) load_halo(kq + 1);
#pragma unroll
        for (int g = 0; g < TAPS; ++g) {
            const int s = kq * TAPS + g;
            const int shift = ((g / KS) * HX + g % KS) * AROW;     // compile-time per unrolled tap
            const unsigned char *Bw = bcur + (wn * WN + fr) * 64;
            if constexpr (DT == LP_BF16) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 af[MI], bf[NI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(lds + aoff[i] + shift + ks * 32 + fh * 16);
#pragma unroll
                    for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const bf16x8 *>(Bw + j * 32 * 64 + ((((2 * ks + fh) ^ bswz) & 3) << 4));
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                }
            } else {
                i32x8 af[MI], bf[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const u32x4 lo = *reinterpret_cast<const u32x4 *>(lds + aoff[i] + shift + fh * 32);
                    const u32x4 hi = *reinterpret_cast<const u32x4 *>(lds + aoff[i] + shift + fh * 32 + 16);
                    af[i] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const u32x4 lo = *reinterpret_cast<const u32x4 *>(Bw + j * 32 * 64 + ((((2 * fh) ^ bswz) & 3) << 4));
                    const u32x4 hi = *reinterpret_cast<const u32x4 *>(Bw + j * 32 * 64 + ((((2 * fh + 1) ^ bswz) & 3) << 4));
                    bf[j] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bf[j], acc[i][j], 0, 0, 0, 0, 0, 0);
            }
            // registers hold tile s + 1; the other buffer was last read in step s - 1, which ended with a barrier
            if (s + 1 < S) store_B(bnext);
            if (s + 2 < S) {
                const int s2 = s + 2;
                load_B(s2 / TAPS, tap_of(s2 % TAPS));
            }
            unsigned char *tb = bcur;
            bcur = bnext;
            bnext = tb;
            __syncthreads();
        }
        if (more) {
            store_halo();          // every wave passed the last tap's barrier: nobody reads the old halo any more
            __syncthreads();
        }
    }

    // ---- epilogue: per-channel affine (+ ReLU), narrow or fp32 store
    const int OH = a.deconv ? 2 * H : H, OW = a.deconv ? 2 * W : W;
    const int ody = a.deconv ? dtap >> 1 : 0, odx = a.deconv ? dtap & 1 : 0, omul = a.deconv ? 2 : 1;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * WN + j * 32 + fr;
        const bool col_ok = col < a.Cout;
        const float al = col_ok ? a.alpha[col] : 0.f, be = col_ok ? a.beta[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int gy = y0 + p / TX, gx = x0 + p % TX;
                if (gy < H && gx < W && col_ok) {
                    float v = fmaf(acc[i][j][r], al, be);
                    if (a.relu) v = fmaxf(v, 0.f);
                    const int64_t o = ((int64_t)(b * OH + gy * omul + ody) * OW + gx * omul + odx) * a.out_ld + a.out_col0 + col;
                    if constexpr (ODT == LP_F32) reinterpret_cast<float *>(a.out)[o] = v;
                    else if constexpr (ODT == LP_BF16) reinterpret_cast<unsigned short *>(a.out)[o] = bf16_bits(v);
                    else reinterpret_cast<unsigned char *>(a.out)[o] = one_fp8(v);
                }
            }
    }
}

template <int DT, int KS, int ODT>
void launch_lp(const LpConvArgs &a, hipStream_t st) {
    const int64_t big = (int64_t)a.B * cdiv(a.H, 8) * cdiv(a.W, 16);
    const unsigned gy = a.deconv ? 4u : 1u;
    if (big * cdiv(a.Cout, 128) >= 384) {
        k_lp_conv<DT, KS, 8, 16, 128, ODT><<<dim3(xcd_grid(big, cdiv(a.Cout, 128)), gy), 256, 0, st>>>(a);
    } else {
        const int64_t small = (int64_t)a.B * cdiv(a.H, 8) * cdiv(a.W, 8);
        k_lp_conv<DT, KS, 8, 8, 64, ODT><<<dim3(xcd_grid(small, cdiv(a.Cout, 64)), gy), 256, 0, st>>>(a);
    }
}

template <int DT, int KS>
void launch_lp_o(const LpConvArgs &a, int odt, hipStream_t st) {
    if (odt == LP_F32) launch_lp<DT, KS, LP_F32>(a, st);
    else if (odt == LP_BF16) launch_lp<DT, KS, LP_BF16>(a, st);
    else launch_lp<DT, KS, LP_FP8>(a, st);
}

// ---- conversions.  x (rows, C) fp32 contiguous -> narrow rows of pitch out_ld starting at column out_col0, value * mul
template <int DT>
__global__ void k_lp_cast(const float *__restrict__ x, int64_t rows, int C, float mul, unsigned char *out, int out_ld, int out_col0) {
    const int64_t n4 = rows * C / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i] * mul;
        const int64_t r = (i * 4) / C;
        const int c = (int)((i * 4) % C);
        const int64_t o = r * out_ld + out_col0 + c;
        if (DT == LP_BF16) {
            const unsigned lo = bf16_bits(v[0]) | ((unsigned)bf16_bits(v[1]) << 16), hi = bf16_bits(v[2]) | ((unsigned)bf16_bits(v[3]) << 16);
            *reinterpret_cast<uint2 *>(out + o * 2) = make_uint2(lo, hi);
        } else {
            *reinterpret_cast<unsigned *>(out + o) = pack4_fp8(v[0], v[1], v[2], v[3]);
        }
    }
}

template <int DT>
__global__ void k_lp_uncast(const unsigned char *__restrict__ x, int64_t n, float mul, float *out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (DT == LP_BF16) v = (float)__builtin_bit_cast(__bf16, reinterpret_cast<const unsigned short *>(x)[i]);
        else v = __builtin_amdgcn_cvt_f32_fp8((int)x[i], 0);
        out[i] = v * mul;
    }
}

// max |x| (bit pattern of a non-negative float orders like an unsigned integer: atomicMax is exact and order-independent)
__global__ void k_lp_amax(const float *__restrict__ x, int64_t n, unsigned *out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

// weights [Cout][K] fp32 (K = taps * Cin, kernel layout) -> narrow, one workgroup per output channel.
// fp8: per-output-channel scale s = max|w| / 448 (so the row uses the whole e4m3 range), w_q = w / s; bf16: s = 1.
template <int DT>
__global__ __launch_bounds__(256) void k_lp_quant_weights(const float *__restrict__ w, int K, unsigned char *wq, float *scale) {
    __shared__ float red[256];
    const int co = blockIdx.x;
    const float *row = w + (int64_t)co * K;
    float s = 1.f;
    if (DT == LP_FP8) {
        float m = 0.f;
        for (int k = threadIdx.x; k < K; k += 256) m = fmaxf(m, fabsf(row[k]));
        red[threadIdx.x] = m;
        __syncthreads();
        for (int d = 128; d >= 1; d >>= 1) {
            if ((int)threadIdx.x < d) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + d]);
            __syncthreads();
        }
        s = red[0] > 0.f ? red[0] / FP8_MAX : 1.f;
    }
    if (threadIdx.x == 0) scale[co] = s;
    const float inv = 1.f / s;
    for (int k = threadIdx.x; k < K; k += 256) {
        if (DT == LP_BF16) reinterpret_cast<unsigned short *>(wq)[(int64_t)co * K + k] = bf16_bits(row[k]);
        else wq[(int64_t)co * K + k] = one_fp8(row[k] * inv);
    }
}

}  // namespace

extern "C" int rd_lp_cast(const float *x, int64_t rows, int C, int dtype, float mul, void *out, int out_ld, int out_col0, void *stream) {
    RD_REQUIRE(dtype == LP_BF16 || dtype == LP_FP8, "rd_lp_cast: dtype must be 0 (bf16) or 1 (fp8 e4m3fn)");
    RD_REQUIRE(C > 0 && C % 4 == 0 && out_ld >= out_col0 + C && out_ld % 4 == 0 && out_col0 % 4 == 0 && rows >= 0, "rd_lp_cast: bad sizes");
    if (rows == 0) return RD_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv(rows * C / 4, 256), 8192);
    if (dtype == LP_BF16) k_lp_cast<LP_BF16><<<grid, 256, 0, S(stream)>>>(x, rows, C, mul, reinterpret_cast<unsigned char *>(out), out_ld, out_col0);
    else k_lp_cast<LP_FP8><<<grid, 256, 0, S(stream)>>>(x, rows, C, mul, reinterpret_cast<unsigned char *>(out), out_ld, out_col0);
    return check_launch("rd_lp_cast");
}

extern "C" int rd_lp_uncast(const void *x, int64_t n, int dtype, float mul, float *out, void *stream) {
    RD_REQUIRE(dtype == LP_BF16 || dtype == LP_FP8, "rd_lp_uncast: dtype must be 0 (bf16) or 1 (fp8 e4m3fn)");
    if (n <= 0) return RD_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n, 256), 8192);
    if (dtype == LP_BF16) k_lp_uncast<LP_BF16><<<grid, 256, 0, S(stream)>>>(reinterpret_cast<const unsigned char *>(x), n, mul, out);
    else k_lp_uncast<LP_FP8><<<grid, 256, 0, S(stream)>>>(reinterpret_cast<const unsigned char *>(x), n, mul, out);
    return check_launch("rd_lp_uncast");
}

extern "C" int rd_lp_amax(const float *x, int64_t n, float *out1, void *stream) {
    hipStream_t st = S(stream);
    RD_HIP(hipMemsetAsync(out1, 0, 4, st));
    if (n <= 0) return RD_OK;
    k_lp_amax<<<(unsigned)std::min<int64_t>(cdiv(n, 256 * 8), 2048), 256, 0, st>>>(x, n, reinterpret_cast<unsigned *>(out1));
    return check_launch("rd_lp_amax");
}

extern "C" int rd_lp_quant_weights(const float *w_k, int Cout, int K, int dtype, void *w_q, float *w_scale, void *stream) {
    RD_REQUIRE(dtype == LP_BF16 || dtype == LP_FP8, "rd_lp_quant_weights: dtype must be 0 (bf16) or 1 (fp8 e4m3fn)");
    RD_REQUIRE(Cout > 0 && K > 0, "rd_lp_quant_weights: bad sizes");
    if (dtype == LP_BF16) k_lp_quant_weights<LP_BF16><<<Cout, 256, 0, S(stream)>>>(w_k, K, reinterpret_cast<unsigned char *>(w_q), w_scale);
    else k_lp_quant_weights<LP_FP8><<<Cout, 256, 0, S(stream)>>>(w_k, K, reinterpret_cast<unsigned char *>(w_q), w_scale);
    return check_launch("rd_lp_quant_weights");
}

extern "C" int rd_lp_conv(const void *in, int dtype, int B, int H, int W, int Cin, int in_ld, const void *w_q, int ksize, int deconv,
                          const float *alpha, const float *beta, int relu, void *out, int out_dtype, int Cout, int out_ld, int out_col0,
                          void *stream) {
    RD_REQUIRE(dtype == LP_BF16 || dtype == LP_FP8, "rd_lp_conv: dtype must be 0 (bf16) or 1 (fp8 e4m3fn)");
    RD_REQUIRE(out_dtype >= 0 && out_dtype <= 2, "rd_lp_conv: out_dtype must be 0 (bf16), 1 (fp8) or 2 (fp32)");
    RD_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0, "rd_lp_conv: bad geometry");
    const int ce = dtype == LP_BF16 ? 32 : 64;
    RD_REQUIRE(Cin > 0 && Cin % ce == 0 && in_ld >= Cin && (in_ld * (dtype == LP_BF16 ? 2 : 1)) % 16 == 0,
               "rd_lp_conv: Cin=%d must be a multiple of %d and rows 16-byte aligned", Cin, ce);
    RD_REQUIRE((deconv == 0 && (ksize == 3 || ksize == 1)) || (deconv == 1 && ksize == 2), "rd_lp_conv: 3x3 / 1x1 convolution or 2x2 stride-2 transposed");
    RD_REQUIRE(out_ld >= out_col0 + Cout && out_col0 >= 0, "rd_lp_conv: output window outside the row");
    RD_REQUIRE(alpha && beta, "rd_lp_conv: alpha / beta are required");
    LpConvArgs a{reinterpret_cast<const unsigned char *>(in), B, H, W, Cin, in_ld, reinterpret_cast<const unsigned char *>(w_q),
                 deconv ? 4 : ksize * ksize, alpha, beta, out, Cout, out_ld, out_col0, relu, deconv};
    hipStream_t st = S(stream);
    if (dtype == LP_BF16) {
        if (ksize == 3) launch_lp_o<LP_BF16, 3>(a, out_dtype, st);
        else launch_lp_o<LP_BF16, 1>(a, out_dtype, st);
    } else {
        if (ksize == 3) launch_lp_o<LP_FP8, 3>(a, out_dtype, st);
        else launch_lp_o<LP_FP8, 1>(a, out_dtype, st);
    }
    return check_launch("rd_lp_conv");
}
