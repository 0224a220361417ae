// Diagnostic (GPU box): how fast can a workgroup stage gathered operand rows from L2 into LDS on gfx950, and what limits it?
// The convolution kernels of this repo all stage 128-byte row segments (8 lanes x 16 B) through registers: global_load_dwordx4 ->
// (split) -> ds_write.  Variants timed here, all reading a 32 MB table (L2 / Infinity-Cache resident) of 1 KB rows by a random
// index list, 2 workgroups of 256 threads per CU, same bytes:
//   0  register-staged, 128-byte segments per row and K step (the pattern of k_conv_igemm_b3), one tile in flight
//   1  the same, two tiles in flight (loads of tile s+1 issued before the LDS stores of tile s)
//   2  register-staged, WHOLE 1 KB rows per wavefront instruction (64 lanes x 16 B), two in flight
//   3  LDS-DMA (global_load_lds_dwordx4), 128-byte segments, lane-linear LDS image
//   4  LDS-DMA, whole rows
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_micro tools/diag/micro/fetch_micro.hip && /tmp/fetch_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ROW_F = 256;            // floats per row (1 KB)
constexpr int TILE_ROWS = 128;        // rows per tile
constexpr int KSTEP = 32;             // floats of a row per K step (128 B)

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_fetch(const float *__restrict__ table, const int *__restrict__ idx, int tiles_per_wg, float *sink) {
    __shared__ __attribute__((aligned(16))) float lds[2][TILE_ROWS * KSTEP];          // 2 x 16 KB
    const int tid = threadIdx.x;
    float acc = 0.f;
    const int ld_r = tid >> 3, ld_c = (tid & 7) * 4;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int *rows = idx + ((size_t)blockIdx.x * tiles_per_wg + t) * TILE_ROWS;
        if (MODE == 0 || MODE == 1) {
            int r[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) r[p] = rows[ld_r + 32 * p];
            f32x4 cur[4], nxt[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) cur[p] = *reinterpret_cast<const f32x4 *>(table + (size_t)r[p] * ROW_F + ld_c);
            for (int k = 0; k < ROW_F / KSTEP; ++k) {
                if (MODE == 1 && k + 1 < ROW_F / KSTEP)
#pragma unroll
                    for (int p = 0; p < 4; ++p) nxt[p] = *reinterpret_cast<const f32x4 *>(table + (size_t)r[p] * ROW_F + (k + 1) * KSTEP + ld_c);
#pragma unroll
                for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4 *>(&lds[k & 1][(ld_r + 32 * p) * KSTEP + ld_c]) = cur[p];
                __syncthreads();
                acc += lds[k & 1][(tid * 7) & (TILE_ROWS * KSTEP - 1)];
                if (MODE == 0 && k + 1 < ROW_F / KSTEP)
#pragma unroll
                    for (int p = 0; p < 4; ++p) nxt[p] = *reinterpret_cast<const f32x4 *>(table + (size_t)r[p] * ROW_F + (k + 1) * KSTEP + ld_c);
#pragma unroll
                for (int p = 0; p < 4; ++p) cur[p] = nxt[p];
            }
        } else if (MODE == 2) {
            // whole rows: wave w handles rows w, w+4, ...; a lane reads 16 B of the row; 32 rows of 1 KB = one 32 KB LDS fill (both buffers)
            const int wave = tid >> 6, lane = tid & 63;
            float *flat = &lds[0][0];
            for (int g = 0; g < TILE_ROWS / 32; ++g) {
                f32x4 v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4 *>(table + (size_t)rows[g * 32 + wave + 4 * q] * ROW_F + lane * 4);
#pragma unroll
                for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4 *>(flat + (size_t)(wave + 4 * q) * ROW_F + lane * 4) = v[q];
                __syncthreads();
                acc += flat[(tid * 7) & (32 * ROW_F - 1)];
                __syncthreads();
            }
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef __attribute__((address_space(3))) void lds_void;
            typedef __attribute__((address_space(1))) const void glb_void;
            const int wave = tid >> 6, lane = tid & 63;
            float *flat = &lds[0][0];
            if (MODE == 3) {
                int r[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) r[p] = rows[ld_r + 32 * p];
                for (int k = 0; k < ROW_F / KSTEP; ++k) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {          // wave covers 8 rows x 128 B = 1 KB, lane-linear in LDS
                        const float *src = table + (size_t)r[p] * ROW_F + k * KSTEP + ld_c;
                        float *dst = &lds[k & 1][(32 * p + wave * 8) * KSTEP];          // wave-uniform base; lane offset added by the hardware
                        __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dst, 16, 0, 0);
                    }
                    __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0)
                    __syncthreads();
                    acc += lds[k & 1][(tid * 7) & (TILE_ROWS * KSTEP - 1)];
                }
            } else {
                for (int g = 0; g < TILE_ROWS / 32; ++g) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float *src = table + (size_t)rows[g * 32 + wave + 4 * q] * ROW_F + lane * 4;
                        float *dst = flat + (size_t)(wave + 4 * q) * ROW_F;
                        __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dst, 16, 0, 0);
                    }
                    __builtin_amdgcn_s_waitcnt(0);
                    __syncthreads();
                    acc += flat[(tid * 7) & (32 * ROW_F - 1)];
                    __syncthreads();
                }
            }
#endif
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
void run(const char *name, const float *table, const int *idx, int n_wg, int tiles_per_wg, float *sink) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) k_fetch<MODE><<<n_wg, 256>>>(table, idx, tiles_per_wg, sink);
    CK(hipDeviceSynchronize());
    const int iters = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) k_fetch<MODE><<<n_wg, 256>>>(table, idx, tiles_per_wg, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double bytes = (double)n_wg * tiles_per_wg * TILE_ROWS * ROW_F * 4;
    printf("mode %d  %-58s %8.1f us  %7.2f TB/s\n", MODE, name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const int table_rows = 32768;                          // 32 MB
    const int n_wg = 512, tiles_per_wg = 16;
    std::vector<int> h((size_t)n_wg * tiles_per_wg * TILE_ROWS);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) % table_rows; }
    float *table, *sink; int *idx;
    CK(hipMalloc(&table, (size_t)table_rows * ROW_F * 4)); CK(hipMemset(table, 0, (size_t)table_rows * ROW_F * 4));
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&idx, h.size() * 4));
    CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<0>("register-staged 128-B segments, 1 tile in flight", table, idx, n_wg, tiles_per_wg, sink);
    run<1>("register-staged 128-B segments, 2 tiles in flight", table, idx, n_wg, tiles_per_wg, sink);
    run<2>("register-staged whole 1-KB rows", table, idx, n_wg, tiles_per_wg, sink);
    run<3>("LDS-DMA (global_load_lds_dwordx4) 128-B segments", table, idx, n_wg, tiles_per_wg, sink);
    run<4>("LDS-DMA whole 1-KB rows", table, idx, n_wg, tiles_per_wg, sink);
    // the same five with a table that fits the XCDs' L2 (2 MB): the staging pipeline against the L2 instead of the Infinity Cache
    {
        std::vector<int> h2(h.size());
        for (size_t i = 0; i < h.size(); ++i) h2[i] = h[i] % 2048;
        CK(hipMemcpy(idx, h2.data(), h2.size() * 4, hipMemcpyHostToDevice));
        printf("-- 2 MB table (L2-resident)\n");
        run<0>("register-staged 128-B segments, 1 tile in flight", table, idx, n_wg, tiles_per_wg, sink);
        run<1>("register-staged 128-B segments, 2 tiles in flight", table, idx, n_wg, tiles_per_wg, sink);
        run<2>("register-staged whole 1-KB rows", table, idx, n_wg, tiles_per_wg, sink);
        run<3>("LDS-DMA 128-B segments", table, idx, n_wg, tiles_per_wg, sink);
        run<4>("LDS-DMA whole 1-KB rows", table, idx, n_wg, tiles_per_wg, sink);
    }
    return 0;
}
