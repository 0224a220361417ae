// Active-site index structures: rank grids (bitmap + popcount prefix) and neighbour tables.
// Replaces spconv's GPU hash-table indice-pair generation and torch.unique (see include/rdamd.h section A).
// All integer work; HBM-bound but tiny (the bitmap of an 8 x 512 x 512 cell space is 256 KiB and lives in L2).
#include "common.hpp"
#include <string.h>

namespace rd {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace rd

using namespace rd;

namespace rd { int g_deterministic = 0; int g_mfma_single = 0; }
extern "C" int rd_set_deterministic(int on) { rd::g_deterministic = on ? 1 : 0; return RD_OK; }
extern "C" int rd_get_deterministic(void) { return rd::g_deterministic; }
// Mixed-precision arithmetic of the bf16x3 kernels (rd_set_conv_math(1)): terms = 3 (default) forms a product from the three bf16 MFMA
// terms hi*hi + hi*lo + lo*hi (fp32-class, ~4e-6); terms = 1 keeps hi*hi only -- operands rounded to bf16, fp32 accumulation and fp32
// storage, ~2.4e-3 per product: the arithmetic torch autocast gives the reference's convolutions under --use_amp
// (tools/train_utils/train_utils.py:57-58), at a third of the matrix-core work.
extern "C" int rd_set_mfma_terms(int terms) {
    RD_REQUIRE(terms == 1 || terms == 3, "rd_set_mfma_terms: 1 (plain bf16 products) or 3 (bf16x3), got %d", terms);
    rd::g_mfma_single = terms == 1 ? 1 : 0;
    return RD_OK;
}
extern "C" int rd_get_mfma_terms(void) { return rd::g_mfma_single ? 1 : 3; }
extern "C" const char *rd_last_error(void) { return rd::g_err; }
extern "C" int rd_abi_version(void) { return 3; }
// Fork: `to` waits for everything enqueued on `from` so far.  One library-owned event per waiting stream, re-recorded on every call (a
// wait captures the event's state when it is enqueued, so re-recording afterwards is safe).  Exists because the weight-gradient side
// stream forks ~90 times per backward pass: one C call instead of torch's Event.record + Stream.wait_event (~3 us of Python each time).
extern "C" int rd_stream_fork(void *from_stream, void *to_stream) {
    static hipStream_t keys[16];
    static hipEvent_t events[16];
    static int n = 0;
    hipStream_t to = reinterpret_cast<hipStream_t>(to_stream), from = reinterpret_cast<hipStream_t>(from_stream);
    int i = 0;
    while (i < n && keys[i] != to) ++i;
    if (i == n) {
        RD_REQUIRE(n < 16, "rd_stream_fork: more than 16 waiting streams");
        RD_HIP(hipEventCreateWithFlags(&events[n], hipEventDisableTiming));
        keys[n++] = to;
    }
    RD_HIP(hipEventRecord(events[i], from));
    RD_HIP(hipStreamWaitEvent(to, events[i], 0));
    return RD_OK;
}

extern "C" int rd_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

extern "C" int64_t rd_rankgrid_bytes(int64_t n_cells) {
    int64_t w = rg_words(n_cells);
    return (2 * w + 1 + cdiv(w, 1024)) * 4;   // bitmap, prefix, count, scan block sums
}

// ---------------------------------------------------------------------------------------------- scan
// Exclusive prefix of popcounts over n_words words.  Three small kernels: per-block sums (1024 words / block),
// scan of the block sums by one block, add.  The last pass also writes the total to the count word.
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 4;                       // words per thread
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;  // 1024 words

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t *total) {
    // wave scan via shuffles, then scan of the 4 wave sums
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        uint32_t s = lds[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_local(const uint32_t *bits, uint32_t *prefix, uint32_t *block_sums, int64_t n_words) {
    __shared__ uint32_t lds[SCAN_BLOCK / 64];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t c[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        c[i] = (base + i < n_words) ? __popc(bits[base + i]) : 0;
        sum += c[i];
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(sum, lds, &tot);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n_words) prefix[base + i] = ex;
        ex += c[i];
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// one block scans all block sums in place (exclusive), writes the grand total
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_sums(uint32_t *block_sums, int n_blocks, uint32_t *count) {
    __shared__ uint32_t lds[SCAN_BLOCK / 64];
    uint32_t carry = 0;
    for (int start = 0; start < n_blocks; start += SCAN_BLOCK) {
        int i = start + threadIdx.x;
        uint32_t v = (i < n_blocks) ? block_sums[i] : 0;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan(v, lds, &tot);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *count = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_add(uint32_t *prefix, const uint32_t *block_sums, int64_t n_words) {
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t off = block_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n_words) prefix[base + i] += off;
}

static int rankgrid_scan(uint32_t *rankgrid, int64_t n_cells, hipStream_t st) {
    int64_t n_words = rg_words(n_cells);
    int n_tiles = (int)cdiv(n_words, SCAN_TILE);
    RD_REQUIRE(n_tiles <= 65536, "rank grid too large: %lld cells", (long long)n_cells);
    // block sums live in the tail of the rank-grid buffer itself (no allocation on this path)
    uint32_t *bits = rankgrid, *prefix = rankgrid + n_words, *count = rankgrid + 2 * n_words, *sums = count + 1;
    k_scan_local<<<n_tiles, SCAN_BLOCK, 0, st>>>(bits, prefix, sums, n_words);
    k_scan_sums<<<1, SCAN_BLOCK, 0, st>>>(sums, n_tiles, count);
    k_scan_add<<<n_tiles, SCAN_BLOCK, 0, st>>>(prefix, sums, n_words);
    return check_launch("rankgrid_scan");
}

// ---------------------------------------------------------------------------------------------- voxelise
__global__ void k_vox_mark(const float *__restrict__ points, int n, int stride, int batch, int gx, int gy,
                           float x0, float y0, float vx, float vy, uint32_t *bits, int32_t *point_key) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = points + (int64_t)i * stride;
    // reference: floor((xy - range_min) / voxel).int() with a true division (dynamic_pillar_vfe.py:201-202)
    float fx = floorf(__fdiv_rn(p[1] - x0, vx));
    float fy = floorf(__fdiv_rn(p[2] - y0, vy));
    int b = (int)p[0];
    int key = -1;
    if (fx >= 0.f && fx < (float)gx && fy >= 0.f && fy < (float)gy && b >= 0 && b < batch) {
        key = (b * gx + (int)fx) * gy + (int)fy;
        atomicOr(&bits[key >> 5], 1u << (key & 31));
    }
    point_key[i] = key;
}

__global__ void k_vox_rank(int32_t *point_key, int n, const uint32_t *bits, const uint32_t *prefix) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int key = point_key[i];
    if (key >= 0) point_key[i] = rg_lookup(bits, prefix, key);
}

extern "C" int rd_voxelize(const float *points, int n_points, int n_feat, int batch, int gx, int gy, float x0, float y0,
                           float vx, float vy, uint32_t *rankgrid, int32_t *point_row, void *stream) {
    RD_REQUIRE(n_points >= 0 && n_feat >= 3 && batch > 0 && gx > 0 && gy > 0, "rd_voxelize: bad sizes");
    RD_REQUIRE((int64_t)batch * gx * gy < (1ll << 31), "rd_voxelize: cell space exceeds int32 keys");
    hipStream_t st = S(stream);
    int64_t n_cells = (int64_t)batch * gx * gy, n_words = rg_words(n_cells);
    RD_HIP(hipMemsetAsync(rankgrid, 0, (2 * n_words + 1) * 4, st));
    if (n_points > 0) k_vox_mark<<<cdiv(n_points, 256), 256, 0, st>>>(points, n_points, 1 + n_feat, batch, gx, gy, x0, y0, vx, vy, rankgrid, point_row);
    int rc = rankgrid_scan(rankgrid, n_cells, st);
    if (rc) return rc;
    if (n_points > 0) k_vox_rank<<<cdiv(n_points, 256), 256, 0, st>>>(point_row, n_points, rankgrid, rankgrid + n_words);
    return check_launch("rd_voxelize");
}

// ---------------------------------------------------------------------------------------------- coords of active cells
__global__ void k_rg_coords(const uint32_t *bits, const uint32_t *prefix, int64_t n_words, int H, int W, int xmajor,
                            int32_t *coords, int max_rows) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t word = bits[w];
    int row = (int)prefix[w];
    while (word) {
        int bit = __ffs(word) - 1;
        word &= word - 1;
        int64_t cell = w * 32 + bit;
        if (row < max_rows) {
            int b, y, x;
            if (xmajor) {  // cell = (b*W + x)*H + y   (voxeliser key order: gx = W, gy = H)
                y = (int)(cell % H);
                x = (int)((cell / H) % W);
                b = (int)(cell / ((int64_t)H * W));
            } else {
                x = (int)(cell % W);
                y = (int)((cell / W) % H);
                b = (int)(cell / ((int64_t)H * W));
            }
            coords[(int64_t)row * 3 + 0] = b;
            coords[(int64_t)row * 3 + 1] = y;
            coords[(int64_t)row * 3 + 2] = x;
        }
        ++row;
    }
}

extern "C" int rd_rankgrid_coords(const uint32_t *rankgrid, int batch, int H, int W, int xmajor, int32_t *coords, int max_rows, void *stream) {
    int64_t n_words = rg_words((int64_t)batch * H * W);
    if (max_rows <= 0) return RD_OK;
    k_rg_coords<<<cdiv(n_words, 256), 256, 0, S(stream)>>>(rankgrid, rankgrid + n_words, n_words, H, W, xmajor, coords, max_rows);
    return check_launch("rd_rankgrid_coords");
}

__device__ __forceinline__ int64_t cell_of(int b, int y, int x, int H, int W, int xmajor) {
    return xmajor ? ((int64_t)b * W + x) * H + y : ((int64_t)b * H + y) * W + x;
}

__global__ void k_mark_coords(const int32_t *coords, int n, int batch, int H, int W, int xmajor, uint32_t *bits) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = coords[i * 3], y = coords[i * 3 + 1], x = coords[i * 3 + 2];
    if (b < 0 || b >= batch || y < 0 || y >= H || x < 0 || x >= W) return;
    int64_t c = cell_of(b, y, x, H, W, xmajor);
    atomicOr(&bits[c >> 5], 1u << (c & 31));
}

extern "C" int rd_rankgrid_from_coords(const int32_t *coords, int n, int batch, int H, int W, int xmajor, uint32_t *rankgrid, void *stream) {
    hipStream_t st = S(stream);
    int64_t n_cells = (int64_t)batch * H * W, n_words = rg_words(n_cells);
    RD_HIP(hipMemsetAsync(rankgrid, 0, (2 * n_words + 1) * 4, st));
    if (n > 0) k_mark_coords<<<cdiv(n, 256), 256, 0, st>>>(coords, n, batch, H, W, xmajor, rankgrid);
    return rankgrid_scan(rankgrid, n_cells, st);
}

// SparseConv2d(k3, s2, p1): input (y,x) feeds outputs oy = (y + 1 - ky) / 2 for ky with even numerator.
__global__ void k_mark_down(const int32_t *coords, int n, int Ho, int Wo, uint32_t *bits) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int b = coords[i * 3], y = coords[i * 3 + 1], x = coords[i * 3 + 2];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        int ny = y + 1 - ky;
        if (ny < 0 || (ny & 1)) continue;
        int oy = ny >> 1;
        if (oy >= Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            int nx = x + 1 - kx;
            if (nx < 0 || (nx & 1)) continue;
            int ox = nx >> 1;
            if (ox >= Wo) continue;
            int64_t c = ((int64_t)b * Ho + oy) * Wo + ox;
            atomicOr(&bits[c >> 5], 1u << (c & 31));
        }
    }
}

extern "C" int rd_rankgrid_downsample(const int32_t *in_coords, int n_in, int batch, int Ho, int Wo, uint32_t *out_rankgrid, void *stream) {
    hipStream_t st = S(stream);
    int64_t n_cells = (int64_t)batch * Ho * Wo, n_words = rg_words(n_cells);
    RD_HIP(hipMemsetAsync(out_rankgrid, 0, (2 * n_words + 1) * 4, st));
    if (n_in > 0) k_mark_down<<<cdiv(n_in, 256), 256, 0, st>>>(in_coords, n_in, Ho, Wo, out_rankgrid);
    return rankgrid_scan(out_rankgrid, n_cells, st);
}

// The same output set straight from the INPUT RANK GRID (no coordinate list, hence no row count from the host): a whole pyramid of
// rank grids can be built with static launch shapes and all its level sizes read back in ONE device->host copy.
__global__ void k_mark_down_bits(const uint32_t *__restrict__ in_bits, int64_t n_words_in, int H, int W, int xmajor, int Ho, int Wo,
                                 uint32_t *out_bits) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words_in) return;
    uint32_t word = in_bits[w];
    while (word) {
        const int bit = __ffs(word) - 1;
        word &= word - 1;
        const int64_t cell = w * 32 + bit;
        int b, y, x;
        if (xmajor) {
            y = (int)(cell % H);
            x = (int)((cell / H) % W);
        } else {
            x = (int)(cell % W);
            y = (int)((cell / W) % H);
        }
        b = (int)(cell / ((int64_t)H * W));
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ny = y + 1 - ky;
            if (ny < 0 || (ny & 1)) continue;
            const int oy = ny >> 1;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int nx = x + 1 - kx;
                if (nx < 0 || (nx & 1)) continue;
                const int ox = nx >> 1;
                if (ox >= Wo) continue;
                const int64_t c = ((int64_t)b * Ho + oy) * Wo + ox;
                atomicOr(&out_bits[c >> 5], 1u << (c & 31));
            }
        }
    }
}

extern "C" int rd_rankgrid_downsample_grid(const uint32_t *in_rankgrid, int batch, int H, int W, int in_xmajor, int Ho, int Wo,
                                           uint32_t *out_rankgrid, void *stream) {
    RD_REQUIRE(batch > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "rd_rankgrid_downsample_grid: bad sizes");
    hipStream_t st = S(stream);
    const int64_t n_in = rg_words((int64_t)batch * H * W);
    const int64_t n_cells = (int64_t)batch * Ho * Wo, n_words = rg_words(n_cells);
    RD_HIP(hipMemsetAsync(out_rankgrid, 0, (2 * n_words + 1) * 4, st));
    k_mark_down_bits<<<cdiv(n_in, 256), 256, 0, st>>>(in_rankgrid, n_in, H, W, in_xmajor, Ho, Wo, out_rankgrid);
    return rankgrid_scan(out_rankgrid, n_cells, st);
}

// ---------------------------------------------------------------------------------------------- neighbour tables
__global__ void k_nbr_subm(const int32_t *coords, int n, const uint32_t *bits, const uint32_t *prefix, int H, int W, int xmajor, int32_t *nbr) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * 9) return;
    int j = g / 9, t = g % 9;
    int b = coords[j * 3], y = coords[j * 3 + 1] + t / 3 - 1, x = coords[j * 3 + 2] + t % 3 - 1;
    int r = -1;
    if (y >= 0 && y < H && x >= 0 && x < W) r = rg_lookup(bits, prefix, cell_of(b, y, x, H, W, xmajor));
    nbr[g] = r;
}

extern "C" int rd_nbr_subm(const int32_t *coords, int n, const uint32_t *rankgrid, int batch, int H, int W, int xmajor, int32_t *nbr, void *stream) {
    if (n <= 0) return RD_OK;
    int64_t n_words = rg_words((int64_t)batch * H * W);
    k_nbr_subm<<<cdiv((int64_t)n * 9, 256), 256, 0, S(stream)>>>(coords, n, rankgrid, rankgrid + n_words, H, W, xmajor, nbr);
    return check_launch("rd_nbr_subm");
}

__global__ void k_nbr_strided(const int32_t *oc, int n_out, const uint32_t *bits, const uint32_t *prefix, int H, int W, int xmajor, int32_t *nbr) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_out * 9) return;
    int j = g / 9, t = g % 9;
    int b = oc[j * 3], y = oc[j * 3 + 1] * 2 - 1 + t / 3, x = oc[j * 3 + 2] * 2 - 1 + t % 3;
    int r = -1;
    if (y >= 0 && y < H && x >= 0 && x < W) r = rg_lookup(bits, prefix, cell_of(b, y, x, H, W, xmajor));
    nbr[g] = r;
}

extern "C" int rd_nbr_strided(const int32_t *out_coords, int n_out, const uint32_t *in_rankgrid, int batch, int H, int W, int in_xmajor, int32_t *nbr, void *stream) {
    if (n_out <= 0) return RD_OK;
    int64_t n_words = rg_words((int64_t)batch * H * W);
    k_nbr_strided<<<cdiv((int64_t)n_out * 9, 256), 256, 0, S(stream)>>>(out_coords, n_out, in_rankgrid, in_rankgrid + n_words, H, W, in_xmajor, nbr);
    return check_launch("rd_nbr_strided");
}

__global__ void k_nbr_strided_T(const int32_t *ic, int n_in, const uint32_t *bits, const uint32_t *prefix, int Ho, int Wo, int32_t *nbrT) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_in * 9) return;
    int i = g / 9, t = g % 9;
    int b = ic[i * 3], ny = ic[i * 3 + 1] + 1 - t / 3, nx = ic[i * 3 + 2] + 1 - t % 3;
    int r = -1;
    if (ny >= 0 && nx >= 0 && !(ny & 1) && !(nx & 1)) {
        int oy = ny >> 1, ox = nx >> 1;
        if (oy < Ho && ox < Wo) r = rg_lookup(bits, prefix, ((int64_t)b * Ho + oy) * Wo + ox);
    }
    nbrT[g] = r;
}

extern "C" int rd_nbr_strided_T(const int32_t *in_coords, int n_in, const uint32_t *out_rankgrid, int batch, int Ho, int Wo, int32_t *nbrT, void *stream) {
    if (n_in <= 0) return RD_OK;
    int64_t n_words = rg_words((int64_t)batch * Ho * Wo);
    k_nbr_strided_T<<<cdiv((int64_t)n_in * 9, 256), 256, 0, S(stream)>>>(in_coords, n_in, out_rankgrid, out_rankgrid + n_words, Ho, Wo, nbrT);
    return check_launch("rd_nbr_strided_T");
}
