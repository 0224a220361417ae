// Weight gradient of DENSE stride-1 3x3 convolutions (DenseEnc, CMA, conv5 blocks, CenterHead first stages) in bf16x3 arithmetic,
// halo-staged: the counterpart of k_conv_d3_b3 for the backward pass.  Same contract as k_conv_wgrad_b3 (conv_b3.hip):
//     grad_w[co][t][ci] += sum_p grad_out[p][co] * in[p + off(t)][ci]        (t = ky*3 + kx, off = (ky - 1, kx - 1), zero outside the map)
//
// Why a second kernel: the gathered kernel stages a (128 co + 128 ci) x 32-row fp32 slab (32 KB) per 96 MFMAs of ONE tap -- at the
// bf16 MFMA rate that asks the memory system for ~26 TB/s of operand traffic, so it sat at 12.6 % MFMA busy with its waves parked
// on loads (profiles/round1_pmc_mfma_busy_bf16x3.json), splitting and transposing every element 18 times over.  Here a workgroup
// stages ONE 8x8-pixel tile of grad_out (64 px x 128 co) and the tile's 10x10 input halo (100 px x 64 ci) -- 57 KB -- and runs all
// NINE taps on it: 864 MFMAs per staged tile, 5x the arithmetic intensity, every element split once per (co, ci) tile pair.
//   * both operands are k-major in memory (pixels x channels) while the 32x32x16 MFMA wants 8 consecutive k per lane: the LDS
//     images stay row-major [pixel][32 channels] (64-byte rows, written with plain 8-byte stores) and the fragments are fetched
//     with ds_read_b64_tr_b16, gfx950's transposing LDS read (4 pixels x 16 channels per 16-lane group, conflict-free on 64-byte
//     rows) -- no register transposes; a tap is a constant byte offset into the halo image.
//   * 8 waves: wave w owns output channels 32 (w & 3) .. +31 x input channels 32 (w >> 2) .. +31 x 9 taps = 9 accumulator tiles
//     (144 VGPRs).  Two LDS stages (2 x 57 KB): the next tile is split and written while this one is multiplied; the four waves that
//     share a SIMD with another four run the two halves of an iteration in opposite order, so one wave's VALU / LDS-write phase
//     sits under its partner's MFMA phase.
//   * row (pixel-tile) chunks are combined with fp32 atomics on 128-byte row segments of the kernel-layout gradient, the chunk
//     count chosen by a cost model (atomic traffic = workgroups x 295 KB); rd_set_deterministic(1): one chunk, no atomics race.
// Bound: MFMA (bf16 2.5 PF, 3 issued products per algorithmic one).  Algorithmic bytes: rows x (Cin + Cout) x 4 + 9 Cin Cout x 4.
#include <algorithm>
#include <stdlib.h>
#include "conv_common.hpp"

using namespace rd;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int WD_T = 8, WD_PX = 64, WD_HX = 10, WD_HP = 100;     // 8x8 pixel tile, 10x10 halo
constexpr int WD_CO = 128, WD_CI = 64;                           // workgroup tile
constexpr int WD_G_EL = 4 * WD_PX * 32, WD_X_EL = 2 * WD_HP * 32; // bf16 elements per part (hi or lo) of one stage
constexpr int WD_STAGE_EL = 2 * WD_G_EL + 2 * WD_X_EL;           // [G hi][G lo][X hi][X lo]
constexpr int WD_LDS_BYTES = 2 * WD_STAGE_EL * 2;                // two stages: 116 736 bytes

struct WgradD3Args {
    const float *in, *go;
    float *gw;
    int B, H, W, Cin, Cout;
    int tiles_per_block, n_ci_tiles;
    int x1;          // 1: hi * hi term only (rd_set_mfma_terms)
};

__device__ __forceinline__ void split4(const f32x4 v, bf16x4 &hi, bf16x4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 h = (__bf16)v[e];
        hi[e] = h;
        lo[e] = (__bf16)(v[e] - (float)h);
    }
}

// one 32x32x16 operand fragment = two transposing reads of 4 pixels x 16 channels each (k = 8h + 0..3 and 8h + 4..7)
__device__ __forceinline__ bf16x8 tr_frag(const __bf16 *p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(p + 4 * 32));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512, 1) void k_conv_wgrad_d3_b3(const WgradD3Args a) {
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wc = w & 3, wj = w >> 2;
    const int H = a.H, W = a.W;
    const int tiles_x = (W + WD_T - 1) / WD_T, tiles_y = (H + WD_T - 1) / WD_T;
    const int total_tiles = a.B * tiles_y * tiles_x;
    const int t_begin = blockIdx.x * a.tiles_per_block, t_end = min(total_tiles, t_begin + a.tiles_per_block);
    const int co0 = (blockIdx.y / a.n_ci_tiles) * WD_CO, ci0 = (blockIdx.y % a.n_ci_tiles) * WD_CI;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- loaders.  grad_out tile: 64 px x 128 co = 2048 float4, 4 per thread; halo: 100 px x 64 ci = 1600 float4, <= 4 per thread
    f32x4 rg[4], rx[4];
    auto load_tile = [&](int tile) {
        const int b = tile / (tiles_y * tiles_x), y0 = ((tile / tiles_x) % tiles_y) * WD_T, x0 = (tile % tiles_x) * WD_T;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 512 * q, px = idx >> 5, c4 = idx & 31;
            const int gy = y0 + (px >> 3), gx = x0 + (px & 7), co = co0 + 4 * c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy < H && gx < W && co < a.Cout) v = *reinterpret_cast<const f32x4 *>(a.go + ((int64_t)(b * H + gy) * W + gx) * a.Cout + co);
            rg[q] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 512 * q, hp = idx >> 4, c4 = idx & 15;
            const int gy = y0 - 1 + hp / WD_HX, gx = x0 - 1 + hp % WD_HX, ci = ci0 + 4 * c4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < WD_HP * 16 && gy >= 0 && gy < H && gx >= 0 && gx < W && ci < a.Cin)
                v = *reinterpret_cast<const f32x4 *>(a.in + ((int64_t)(b * H + gy) * W + gx) * a.Cin + ci);
            rx[q] = v;
        }
    };
    auto store_tile = [&](__bf16 *st) {
        __bf16 *Gh = st, *Gl = Gh + WD_G_EL, *Xh = Gl + WD_G_EL, *Xl = Xh + WD_X_EL;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 512 * q, px = idx >> 5, c4 = idx & 31;
            const int o = ((c4 >> 3) * WD_PX + px) * 32 + (c4 & 7) * 4;
            bf16x4 hi, lo;
            split4(rg[q], hi, lo);
            *reinterpret_cast<bf16x4 *>(Gh + o) = hi;
            *reinterpret_cast<bf16x4 *>(Gl + o) = lo;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 512 * q, hp = idx >> 4, c4 = idx & 15;
            if (idx < WD_HP * 16) {
                const int o = ((c4 >> 3) * WD_HP + hp) * 32 + (c4 & 7) * 4;
                bf16x4 hi, lo;
                split4(rx[q], hi, lo);
                *reinterpret_cast<bf16x4 *>(Xh + o) = hi;
                *reinterpret_cast<bf16x4 *>(Xl + o) = lo;
            }
        }
    };

    // ---- fragment addressing (ds_read_b64_tr_b16): 16-lane group g = lane >> 4 reads 4 pixels x 16 channels; lane 4q + p of the group
    // supplies the address of pixel row q, channels 4p .. 4p+3 and receives channel (lane & 15) of the 4 pixels.  Groups 0 / 1 are
    // channels 0-15 / 16-31 at k = 0..7 (h = 0), groups 2 / 3 the same channels at k = 8..15 (h = 1).
    const int grp = lane >> 4, li = lane & 15, fq = li >> 2, fp = li & 3, fh = grp >> 1, cb = (grp & 1) * 16;
    const int a_base = (wc * WD_PX + 8 * fh + fq) * 32 + cb + 4 * fp;            // + (16 kc) * 32 per k chunk
    const int b_base = (wj * WD_HP + fh * WD_HX + fq) * 32 + cb + 4 * fp;        // + ((2 kc + dy) * 10 + dx) * 32 per (k chunk, tap)

    auto compute = [&](const __bf16 *st) {
        const __bf16 *Gh = st + a_base, *Gl = Gh + WD_G_EL, *Xh = st + 2 * WD_G_EL + b_base, *Xl = Xh + WD_X_EL;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            // k = 16 kc .. 16 kc + 15 = pixel rows 2 kc, 2 kc + 1 of the tile; within a lane half: row 2 kc + h, columns 0..7;
            // the two reads of a fragment are columns 0..3 and 4..7 (4 consecutive halo pixels = 4 consecutive LDS rows)
            const bf16x8 ah = tr_frag(Gh + kc * 16 * 32), al = tr_frag(Gl + kc * 16 * 32);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int off = ((2 * kc + t / 3) * WD_HX + t % 3) * 32;
                const bf16x8 bh = tr_frag(Xh + off), bl = tr_frag(Xl + off);
                if (!a.x1) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            }
        }
    };

    if (t_begin < t_end) {
        load_tile(t_begin);
        store_tile(lds);
        if (t_begin + 1 < t_end) load_tile(t_begin + 1);
    }
    __syncthreads();
    const bool stage_first = w < 4;          // waves w and w + 4 share a SIMD: opposite phase order inside an iteration
    for (int t = t_begin; t < t_end; ++t) {
        __bf16 *cur = lds + ((t - t_begin) & 1) * WD_STAGE_EL, *nxt = lds + (((t - t_begin) & 1) ^ 1) * WD_STAGE_EL;
        if (stage_first) {
            if (t + 1 < t_end) store_tile(nxt);          // `nxt` was last read in iteration t - 1, which ended with a barrier
            if (t + 2 < t_end) load_tile(t + 2);
            compute(cur);
        } else {
            compute(cur);
            if (t + 1 < t_end) store_tile(nxt);
            if (t + 2 < t_end) load_tile(t + 2);
        }
        __syncthreads();
    }

    // ---- combine: acc[t][r] is (co = 8 (r >> 2) + (r & 3) + 4 (lane >> 5), ci = lane & 31) of this wave's 32 x 32 tile
    const int ci = ci0 + wj * 32 + (lane & 31);
    if (ci < a.Cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = acc[t][r];
                if (co < a.Cout && v != 0.f) atomicAdd(&a.gw[((int64_t)co * 9 + t) * a.Cin + ci], v);
            }
    }
}

}  // namespace

// true when the halo kernel applies (bf16x3 mode, dense 3x3 stride 1 pad 1 forward geometry, fp32 operands); launches it.
bool launch_wgrad_d3_b3(const float *in, int in_rows, int Cin, const float *go, int out_rows, int Cout, int taps, const rd_conv_index *ix, float *gw,
                        hipStream_t st) {
    if (!(ix->mode == 1 && taps == 9 && ix->KH == 3 && ix->KW == 3 && ix->stride == 1 && ix->pad == 1 && ix->Hin == ix->Hout && ix->Win == ix->Wout))
        return false;
    if (Cin % 4 || Cout % 4 || Cin < 32 || Cout < 32 || in_rows != out_rows) return false;
    static const bool off = getenv("RD_WGRAD_D3") && getenv("RD_WGRAD_D3")[0] == '0';
    if (off) return false;
    const int64_t B = out_rows / ((int64_t)ix->Hout * ix->Wout);
    if (B * ix->Hout * ix->Wout != out_rows) return false;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_wgrad_d3_b3), hipFuncAttributeMaxDynamicSharedMemorySize, WD_LDS_BYTES) != hipSuccess)
            return false;
        attr_set = true;
    }
    const int64_t total_tiles = B * cdiv(ix->Hout, WD_T) * cdiv(ix->Wout, WD_T);
    const int n_co = (int)cdiv(Cout, WD_CO), n_ci = (int)cdiv(Cin, WD_CI), n_cc = n_co * n_ci;
    // chunk count: rounds of <= 256 resident workgroups x tiles per workgroup x ~1.9 us per tile, plus the atomic pass
    // (every workgroup adds its 128 x 64 x 9 tile: ~0.23 us at the ~1.3 TB/s the memory-side fp32 atomics sustain chip-wide)
    int64_t chunks = 1;
    if (!g_deterministic) {
        double best = 1e30;
        // ONE round on at most `res` CUs.  A workgroup of this kernel owns its CU's whole register file (8 waves x 256 VGPRs): with
        // 256 of them resident nothing of the main stream -- the BatchNorm backward passes and data gradients this launch is meant
        // to overlap with -- could start anywhere for the launch's duration (kernel trace: 91-99 us for a 33.6 MB BatchNorm pass
        // that takes 30 us alone).  The weight-gradient stream has slack, the main stream is the critical path: measured step
        // 19.41 ms at 256, 19.14 / 19.07 / 18.93 / 18.74 / 18.5 / 18.65 ms at 224 / 192 / 128 / 96 / 80 / 64, 19.9 ms at 48, 22.1 ms at 32.
        static const int res = getenv("RD_WGRAD_D3_RES") ? std::max(1, atoi(getenv("RD_WGRAD_D3_RES"))) : 80;
        for (int64_t c = 1; c <= std::min<int64_t>(total_tiles, 256); ++c) {
            if (res < 256 && c * n_cc > res && c > 1) break;
            const double cost = (double)cdiv(c * n_cc, res) * (double)cdiv(total_tiles, c) * 1.9 + (double)c * n_cc * 0.23;
            if (cost < best) { best = cost; chunks = c; }
        }
    }
    const int tpb = (int)cdiv(total_tiles, chunks);
    chunks = cdiv(total_tiles, tpb);
    WgradD3Args a{in, go, gw, (int)B, ix->Hout, ix->Wout, Cin, Cout, tpb, n_ci, g_mfma_single};
    k_conv_wgrad_d3_b3<<<dim3((unsigned)chunks, (unsigned)n_cc), 512, WD_LDS_BYTES, st>>>(a);
    return true;
}
