// Shared helpers for librdamd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/rdamd.h"

namespace rd {

void set_error(const char *fmt, ...);

// rd_set_deterministic (test switch): every floating-point reduction runs in one fixed order -- reductions that normally combine
// per-block partial sums with fp32 atomics (BatchNorm statistics, weight-gradient row chunks, column sums, ...) are launched so that
// each output element has exactly ONE contributing block, or run their own ordered variant.  Results then do not depend on how
// kernels of different streams interleave; speed is not a goal of this mode.
extern int g_deterministic;
extern int g_mfma_single;          // rd_set_mfma_terms(1): the bf16x3 kernels keep only the hi * hi product

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return RD_EHIP;
    }
    return RD_OK;
}

#define RD_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            rd::set_error(__VA_ARGS__);  \
            return RD_EINVAL;            \
        }                                \
    } while (0)

#define RD_HIP(call)                                                   \
    do {                                                               \
        hipError_t e__ = (call);                                       \
        if (e__ != hipSuccess) {                                       \
            rd::set_error("%s: %s", #call, hipGetErrorString(e__));    \
            return RD_EHIP;                                            \
        }                                                              \
    } while (0)

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline hipStream_t S(void *s) { return reinterpret_cast<hipStream_t>(s); }

// rank grid view: [n_words bitmap][n_words prefix][1 count]
struct RankGrid {
    const uint32_t *bits;
    const uint32_t *prefix;
    int64_t n_words;
};
__host__ __device__ inline int64_t rg_words(int64_t n_cells) { return (n_cells + 31) / 32; }

__device__ __forceinline__ int rg_lookup(const uint32_t *bits, const uint32_t *prefix, int64_t cell) {
    uint32_t w = bits[cell >> 5];
    uint32_t b = 1u << (cell & 31);
    if (!(w & b)) return -1;
    return (int)(prefix[cell >> 5] + __popc(w & (b - 1)));
}

}  // namespace rd
