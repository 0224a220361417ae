// Shared pieces of the implicit-GEMM convolution kernels (conv.hip: exact fp32 MFMA; conv_b3.hip: bf16x3 split MFMA).
#pragma once
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KB = 32;
constexpr int LDK = KB + 4;  // padded LDS row (floats)
constexpr int MAX_TAPS = 16;

struct ConvArgs {
    const float *in;
    int in_rows, Cin;
    const float *w;
    int taps;
    const float *bias;
    float *out;
    int out_rows, Cout;
    rd_conv_index ix;
    const float *scale, *shift, *residual;
    int relu;
    float *stats;
    // bf16x3 kernels only: operand already in "split format" (rd_split_bf16): every 16-byte group of 4 channels holds
    // [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] as bf16 instead of 4 floats -- same size, same addressing, no in-loop split
    int in_split = 0, w_split = 0;
    // bf16x3 kernels: 1 = keep only the hi*hi term (plain bf16 products, fp32 accumulate: the mixed-precision mode of rd_set_mfma_terms)
    int x1 = 0;
};

__device__ __forceinline__ int src_row(const ConvArgs &a, int j, int t) {
    if (j >= a.out_rows) return -1;
    const rd_conv_index &ix = a.ix;
    if (ix.mode == 0) {
        int tt = ix.flip ? (a.taps - 1 - t) : t;
        return ix.nbr[(int64_t)j * a.taps + tt];
    }
    if (ix.mode == 3) {  // deformable sampling: "has a source" == any of the 4 bilinear corners is inside the map
        const int4 q = *reinterpret_cast<const int4 *>(ix.samp_idx + ((int64_t)j * a.taps + t) * 4);
        return max(max(q.x, q.y), max(q.z, q.w));
    }
    int ox = j % ix.Wout;
    int oy = (j / ix.Wout) % ix.Hout;
    int b = j / (ix.Wout * ix.Hout);
    int ky = t / ix.KW, kx = t % ix.KW;
    int iy, ixx;
    if (ix.mode == 1) {
        iy = oy * ix.stride - ix.pad + ky;
        ixx = ox * ix.stride - ix.pad + kx;
    } else {  // transposed: oy = iy*stride - pad + ky
        int ny = oy + ix.pad - ky, nx = ox + ix.pad - kx;
        if (ny < 0 || nx < 0 || (ny % ix.stride) || (nx % ix.stride)) return -1;
        iy = ny / ix.stride;
        ixx = nx / ix.stride;
    }
    if (iy < 0 || iy >= ix.Hin || ixx < 0 || ixx >= ix.Win) return -1;
    return (b * ix.Hin + iy) * ix.Win + ixx;
}


// Dense-geometry source row from output coordinates (no division by the map size): same result as src_row() for modes 1 / 2.
// (ky, kx) of a tap are block constants in the weight-gradient kernels: callers hoist the division out of their row loops
// (PMC: k_conv_wgrad_b3 issued 15.8 VALU instructions per MFMA, a quarter of them this division repeated for every row)
__device__ __forceinline__ int src_row_dense_k(const rd_conv_index &ix, int b, int oy, int ox, int ky, int kx) {
    int iy, ixx;
    if (ix.mode == 1) {
        iy = oy * ix.stride - ix.pad + ky;
        ixx = ox * ix.stride - ix.pad + kx;
    } else {
        const int ny = oy + ix.pad - ky, nx = ox + ix.pad - kx;
        if (ny < 0 || nx < 0) return -1;
        if (ix.stride == 1) {
            iy = ny;
            ixx = nx;
        } else if (ix.stride == 2) {
            if ((ny | nx) & 1) return -1;
            iy = ny >> 1;
            ixx = nx >> 1;
        } else {
            if ((ny % ix.stride) || (nx % ix.stride)) return -1;
            iy = ny / ix.stride;
            ixx = nx / ix.stride;
        }
    }
    if (iy < 0 || iy >= ix.Hin || ixx < 0 || ixx >= ix.Win) return -1;
    return (b * ix.Hin + iy) * ix.Win + ixx;
}

__device__ __forceinline__ int src_row_dense(const rd_conv_index &ix, int b, int oy, int ox, int t) {
    const int ky = t / ix.KW, kx = t - ky * ix.KW;
    return src_row_dense_k(ix, b, oy, ox, ky, kx);
}

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2), so give every XCD a
// contiguous slab of row tiles and walk (row tile, column tile) pairs in that slab consecutively: the column tiles of one row
// tile re-use its gathered A rows from the XCD's L2, and neighbouring row tiles share their halo rows.  Placement only affects
// speed, never correctness (blocks past the last tile exit).
__device__ __forceinline__ bool xcd_tile(int n_row_tiles, int n_col_tiles, int &row_tile, int &col_tile) {
    const int id = blockIdx.x;
    const int per_xcd = (n_row_tiles + 7) / 8;
    const int xcd = id & 7, local = id >> 3;
    row_tile = xcd * per_xcd + local / n_col_tiles;
    col_tile = local % n_col_tiles;
    return row_tile < n_row_tiles && (local / n_col_tiles) < per_xcd;
}
inline unsigned xcd_grid(int64_t n_row_tiles, int64_t n_col_tiles) { return (unsigned)(8 * ((n_row_tiles + 7) / 8) * n_col_tiles); }
