/* rdamd.h -- C ABI of librdamd.so: the MI355X (gfx950) kernels of the RadarDistill training hot path.
 *
 * Drop-in boundary.  Plain pointers and sizes only; no torch / ATen types.
 * All pointers are DEVICE pointers unless a parameter is named *_host.  All tensors are contiguous,
 * feature maps are "rows x channels" (channels-last: a dense BEV map (B,C,H,W) is stored as
 * rows (b,y,x) x C), float32 unless noted, indices int32.  `stream` is a hipStream_t passed as void*.
 * Functions only enqueue work on `stream`; they never allocate or synchronise, except
 * rd_index_* which documents its one device->host readback.  Scratch memory comes from the caller
 * (`ws`, size from the matching *_ws_bytes function).
 * Return value: 0 on success, negative RD_E* on error; rd_last_error() gives the message
 * (the reference's ops raise c10::Error / exit(-1): modulated_deform_conv_cuda.cu:39-73, iou3d_nms.cpp:14-26;
 * the Python host turns a non-zero code into RuntimeError).
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef RDAMD_H
#define RDAMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RD_OK 0
#define RD_EINVAL (-1)   /* bad shape / argument */
#define RD_EHIP (-2)     /* HIP runtime error at launch */
#define RD_ENOSPC (-3)   /* workspace too small */

const char *rd_last_error(void);
int rd_abi_version(void);
/* Test switch (process-wide, default 0).  1 = every floating-point reduction of the library runs in one fixed order: sums that are
 * normally combined from per-block partials with fp32 atomics (BatchNorm statistics, weight-gradient row chunks, column sums, GRN /
 * VFE / DCN scatter sums) get exactly one contributing block per output element or an ordered variant.  Results are then
 * bit-identical from run to run and independent of stream placement; launches are much slower.  The reference has no such switch
 * (cuDNN / spconv / torch_scatter atomics are non-deterministic there too); it exists so that tests can separate scheduling
 * defects from summation-order noise. */
int rd_set_deterministic(int on);
int rd_get_deterministic(void);
/* 1 if a gfx950 device is usable, 0 otherwise (never throws). */
int rd_device_ok(void);
/* Stream fork in one call: `to_stream` waits for everything enqueued on `from_stream` so far (hipEventRecord on a library-owned event of
 * the waiting stream + hipStreamWaitEvent).  Host plumbing of the weight-gradient side stream (radardistill_amd/autograd.py). */
int rd_stream_fork(void *from_stream, void *to_stream);

/* ------------------------------------------------------------------------------------------------
 * A. Active-site index structures ("rulebooks").  Replaces spconv's hash-table indice-pair generation
 *    (third-party spconv 2.x, call sites pcdet/models/backbones_3d/spconv_backbone_2d.py:9-28,264-269)
 *    and torch.unique in pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:212.
 *
 *    A "rank grid" over a dense cell space of n_cells cells is a bitmap (1 bit / cell) plus an exclusive
 *    popcount prefix per 32-bit word; the row of an active cell is prefix[word] + popc(bits below it), so rows
 *    are sorted by cell number.  Layout of a rank-grid buffer (uint32): [n_words bitmap][n_words prefix][1 count]
 *    [ceil(n_words/1024) scan scratch], n_words = ceil(n_cells / 32).
 * ---------------------------------------------------------------------------------------------- */
int64_t rd_rankgrid_bytes(int64_t n_cells);

/* Dynamic pillar voxelisation (dynamic_pillar_vfe.py:200-212,243-248).
 * points (n_points, 1 + n_feat): col 0 = batch idx, cols 1..3 = x,y,z.  cell key = b*gx*gy + cx*gy + cy with
 * cx = floor((x - x0)/vx), cy = floor((y - y0)/vy) (true fp32 division), points outside [0,g) dropped.
 * Outputs: rankgrid over batch*gx*gy cells (zero-initialised inside), point_row[n_points] = pillar row or -1.
 * After this call *count_dev (= rankgrid count word) holds the number of pillars P. */
int rd_voxelize(const float *points, int n_points, int n_feat, int batch, int gx, int gy,
                float x0, float y0, float vx, float vy, uint32_t *rankgrid, int32_t *point_row, void *stream);

/* Pillar coordinates in row order: coords[P,3] = (b, y=cy, x=cx) int32 (dynamic_pillar_vfe.py:243-248).
 * `xmajor` = 1 for the voxeliser's key order (b, cx, cy); 0 for cell = (b*H + y)*W + x grids. */
int rd_rankgrid_coords(const uint32_t *rankgrid, int batch, int H, int W, int xmajor, int32_t *coords, int max_rows, void *stream);

/* Generic: mark cells given by coords (n,3)=(b,y,x) in a (zeroed here) rank grid, then scan.  Rows of `coords`
 * must be unique cells.  Used when a SparseConvTensor is built from caller-supplied, sorted indices. */
int rd_rankgrid_from_coords(const int32_t *coords, int n, int batch, int H, int W, int xmajor, uint32_t *rankgrid, void *stream);

/* SparseConv2d(k3,s2,p1) output set (spconv semantics): out cell (b,oy,ox) active iff an active input lies in its
 * 3x3/stride-2 window.  in_coords (n_in,3).  Builds the output rank grid (Ho = (H-1)/2+1 ...). */
int rd_rankgrid_downsample(const int32_t *in_coords, int n_in, int batch, int Ho, int Wo, uint32_t *out_rankgrid, void *stream);
/* Same output set computed from the input RANK GRID instead of a coordinate list (in_xmajor: cell order of the input grid as in
 * rd_rankgrid_coords): needs no row count from the host, so a whole pyramid can be marked and all level sizes read back at once. */
int rd_rankgrid_downsample_grid(const uint32_t *in_rankgrid, int batch, int H, int W, int in_xmajor, int Ho, int Wo, uint32_t *out_rankgrid,
                                void *stream);

/* Neighbour tables.  nbr[n_out][9] int32: row of the input feeding tap t = ky*3+kx of output row j, or -1.
 *   subm   : input == output set; tap reads (y+ky-1, x+kx-1).
 *   strided: output (oy,ox) tap reads input (2*oy-1+ky, 2*ox-1+kx).
 *   strided_T (for the data gradient): nbrT[n_in][9] = output row o that uses input i as its tap t, or -1. */
int rd_nbr_subm(const int32_t *coords, int n, const uint32_t *rankgrid, int batch, int H, int W, int xmajor, int32_t *nbr, void *stream);
int rd_nbr_strided(const int32_t *out_coords, int n_out, const uint32_t *in_rankgrid, int batch, int H, int W, int in_xmajor, int32_t *nbr, void *stream);
int rd_nbr_strided_T(const int32_t *in_coords, int n_in, const uint32_t *out_rankgrid, int batch, int Ho, int Wo, int32_t *nbrT, void *stream);

/* ------------------------------------------------------------------------------------------------
 * B. Pillar feature encoder.  Replaces (Radar_)DynamicPillarVFESimple2D.forward + PFNLayerV2
 *    (dynamic_pillar_vfe.py:14-46,195-313) incl. torch_scatter.scatter_mean / scatter_max.
 * ---------------------------------------------------------------------------------------------- */
/* Per-pillar xyz sums and counts: pillar_acc[P][4] = (sum x, sum y, sum z, count), zero-initialised inside. */
int rd_vfe_pillar_mean(const float *points, int n_points, int n_feat, const int32_t *point_row, int n_pillars,
                       float *pillar_acc, void *stream);
/* Linear(Cin->32, no bias) of the 9+n_feat point features, and per-channel sum / sum of squares over the valid
 * points (train-mode BatchNorm1d statistics).  stats[64] zero-initialised inside; stats[64] = n_valid as float at [64]. */
int rd_vfe_linear_stats(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords,
                        const float *pillar_acc, const float *weight /*[32][9+n_feat]*/, const float *geom /*[9]: vx,vy,vz,xoff,yoff,zoff,x0,y0,z0*/,
                        float *stats /*[65]*/, void *stream);
/* Linear -> affine (scale/shift = folded BatchNorm) -> ReLU -> per-pillar max.  out[P][32]; argmax[P][32]
 * (point index of the maximum, smallest index on ties; may be NULL). */
int rd_vfe_linear_bn_relu_max(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords,
                              const float *pillar_acc, const float *weight, const float *geom,
                              const float *scale, const float *shift, int n_pillars,
                              float *out, int32_t *argmax, unsigned long long *ws_packed /*[P*32]*/, void *stream);
/* Backward of the student VFE: grad_out[P][32] -> grad_weight[32][Cin], grad_gamma[32], grad_beta[32].
 * mean/rstd = batch statistics saved by the forward; gamma = BN weight.  ws: n_points*32 floats + 128 floats. */
int rd_vfe_backward(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords,
                    const float *pillar_acc, const float *weight, const float *geom,
                    const float *mean, const float *rstd, const float *gamma, const float *beta,
                    const float *grad_out, const int32_t *argmax, int n_pillars, int n_valid,
                    float *grad_weight, float *grad_gamma, float *grad_beta, float *ws, void *stream);
/* The same in two halves for SyncBatchNorm (see section D): _reduce scatters grad_out to the arg-max points (kept in ws) and
 * accumulates this rank's (grad_gamma, grad_beta); the host all-reduces a copy over its process group; _weight takes the group-wide
 * sums and count_dev[0] = group-wide number of in-range points. */
int rd_vfe_backward_reduce(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords,
                           const float *pillar_acc, const float *weight, const float *geom, const float *mean, const float *rstd,
                           const float *gamma, const float *beta, const float *grad_out, const int32_t *argmax, int n_pillars,
                           float *grad_gamma, float *grad_beta, float *ws, void *stream);
int rd_vfe_backward_weight(const float *points, int n_points, int n_feat, const int32_t *point_row, const int32_t *coords,
                           const float *pillar_acc, const float *weight, const float *geom, const float *mean, const float *rstd,
                           const float *gamma, const float *sum_gamma, const float *sum_beta, const float *count_dev, int n_pillars,
                           float *grad_weight, const float *ws, void *stream);

/* Segmented form of the VFE (vfe_seg.hip; the default path): rd_vfe_group groups the in-range points by pillar (integer counting sort on
 * point_row: offsets (n_pillars + 1), order (point indices, n_valid of them used); ws: rd_vfe_group_ws_bytes(n_pillars) bytes), then
 * ONE wavefront per pillar computes the pillar mean, the 9 + C features, Linear, folded BatchNorm, ReLU and the per-pillar max with
 * wavefront shuffles -- no floating-point atomics, every output written once.  rd_vfe_seg_stats: stats[65] as rd_vfe_linear_stats
 * (train-mode pass 1); rd_vfe_seg_max: out (P, 32), argmax (P, 32) or NULL (smallest point index wins ties), pillar_acc (P, 4) =
 * sum x, y, z, count or NULL (rd_vfe_backward reads it).  Replace torch_scatter.scatter_mean / scatter_max and the ~25 ATen kernels of
 * dynamic_pillar_vfe.py:214-241 like the atomic entries above. */
int64_t rd_vfe_group_ws_bytes(int n_pillars);
int rd_vfe_group(const int32_t *point_row, int n_points, int n_pillars, int32_t *offsets, int32_t *order, int32_t *ws, int64_t ws_bytes,
                 void *stream);
int rd_vfe_seg_stats(const float *points, int n_feat, const int32_t *order, const int32_t *offsets, const int32_t *coords, const float *weight,
                     const float *geom, int n_pillars, float *stats, void *stream);
int rd_vfe_seg_max(const float *points, int n_feat, const int32_t *order, const int32_t *offsets, const int32_t *coords, const float *weight,
                   const float *geom, const float *scale, const float *shift, int n_pillars, float *out, int32_t *argmax, float *pillar_acc,
                   void *stream);

/* ------------------------------------------------------------------------------------------------
 * C. Convolution as gathered implicit GEMM on the matrix cores (fp32 MFMA).
 *    One kernel family serves
 *      - spconv SubMConv2d / SparseConv2d forward and data gradient (index mode TABLE),
 *      - nn.Conv2d / nn.ConvTranspose2d on channels-last dense BEV maps (index modes DENSE / DENSE_T),
 *      - nn.Linear / 1x1 conv (1 tap).
 *    out[j][:] = act( (sum_t in[src(j,t)][:] @ W[:, t, :]^T + bias) * scale + shift + residual[j] )
 *    Replaces: spconv gather-GEMM-scatter (spconv_backbone_2d.py:13-15,49-56), ATen/cuDNN conv2d
 *    (base_bev_backbone.py:222-262, radar_distill_final.py:38-77, radar_center_head.py:40-45).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int mode;            /* 0 TABLE (nbr[n_out][taps]), 1 DENSE conv, 2 DENSE_T (transposed conv / data-grad of strided conv),
                            3 DEFORM (DCNv2 sampling table from rd_dcn_prep) */
    const int32_t *nbr;  /* TABLE mode */
    int B, Hin, Win, Hout, Wout;   /* dense modes: input / output map sizes */
    int KH, KW, stride, pad;       /* dense modes */
    int flip;            /* TABLE mode: 1 = read table column (taps-1-t) for weight tap t (SubM data gradient) */
    const int32_t *samp_idx;       /* DEFORM mode: [n_out][taps][4] input rows of the 4 bilinear corners, -1 = outside */
    const float *samp_w;           /* DEFORM mode: [n_out][taps][4] mask * corner weight */
} rd_conv_index;

/* weight_k: kernel layout [Cout][taps][Cin] (Cin contiguous).  bias/scale/shift/residual may be NULL.
 * stats (may be NULL): [2*Cout] per-channel sum and sum of squares of the value BEFORE scale/shift/residual/act
 * but after bias (accumulated with atomics; caller zeroes).  relu: 0/1.  Cin % 32 == 0 required. */
int rd_conv_fwd(const float *in, int in_rows, int Cin, const float *weight_k, int taps, const float *bias,
                float *out, int out_rows, int Cout, const rd_conv_index *idx,
                const float *scale, const float *shift, const float *residual, int relu, float *stats, void *stream);

/* Arithmetic of rd_conv_fwd for Cout > 32: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32, default); 1 = "bf16x3": operands split
 * into bf16 hi + lo while staged, three bf16 MFMAs per product, fp32 accumulate (~4e-6 relative error, 5x fewer matrix cycles). */
int rd_set_conv_math(int mode);
int rd_get_conv_math(void);
/* Mixed precision inside mode 1: terms = 3 (default) is bf16x3; terms = 1 keeps only the hi * hi MFMA term -- operands rounded to bf16,
 * fp32 accumulation, fp32 storage (~2.4e-3 per product): the arithmetic of torch.cuda.amp.autocast convolutions, which the reference's
 * `--use_amp` training loop runs under (tools/train_utils/train_utils.py:23,57-64), at a third of the matrix-core work.  The 32-channel
 * wavefront kernels (conv_small.hip) keep three terms. */
int rd_set_mfma_terms(int terms);
int rd_get_mfma_terms(void);

/* Pre-split operands for the bf16x3 kernels.  "Split format": every 16-byte group of 4 consecutive fp32 elements is replaced by
 * [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] as bf16 (hi = bf16(x), lo = bf16(x - hi)) -- same size, same addressing, so a tensor is
 * split ONCE (rd_split_bf16, or rd_weight_layout_split for weights) instead of once per tap, column tile and GEMM that reads it.
 * rd_conv_fwd_split / rd_conv_wgrad_split are rd_conv_fwd / rd_conv_wgrad with a flag per operand saying which format it is in
 * (bf16x3 mode only; an input sampled by a deformable index must stay fp32). */
int rd_split_bf16(const float *x, int64_t n, void *out, void *stream);
/* kind | RD_LAYOUT_FRAG: FRAGMENT-MAJOR split format for weights -- the destination [A][taps][B] (B = the consuming GEMM's K axis) is
 * stored in blocks of 32 (A) x 16 (B) of one tap, each block as the two 1-KiB operand images of v_mfma_f32_32x32x16_bf16 (hi parts,
 * then lo parts; lane 32 * kh + r: A-row r, B-elements 8 kh .. +7), so a wavefront loads a B fragment with one coalesced 16-byte-per-
 * lane read from L2 and the kernel needs no LDS staging for the weights.  A % 32 == 0, B % 16 == 0; same size as the fp32 tensor.
 * Consumed by rd_conv_fwd_split with w_is_split = 2 (dense stride-1 3x3 convolutions and 1-tap GEMMs in bf16x3 mode). */
#define RD_LAYOUT_FRAG 16
int rd_weight_layout_split(const float *src, void *dst, int Cout, int Cin, int taps, int kind, int flip, void *stream);
/* The same conversion for MANY weights in one launch (all trainable conv weights once per optimizer step).  jobs_dev: device array of
 * jobs (kinds as rd_weight_layout_split, no tap flip); chunk c (one 256-thread workgroup) converts work item chunk_group[c] of job
 * chunk_job[c] -- a 16 x 64 x <= 9 (slow axis x fast axis x taps) tile of the destination, staged through LDS so that both the
 * source and the destination are touched in address order; a job has rd_weight_layout_split_items() work items, numbered from 0,
 * and the caller builds the chunk tables so that every item of every job appears once. */
typedef struct {
    const float *src;
    void *dst;
    int Cout, Cin, taps, kind;
} rd_layout_job;
/* jobs_host: HOST array of up to RD_LAYOUT_MULTI_MAX plain (non-split) re-layouts, all done by one launch (the table travels in the
 * kernel arguments).  Used for the weight gradients of a backward pass (kernel layout -> nn.Conv2d / nn.ConvTranspose2d layout). */
#define RD_LAYOUT_MULTI_MAX 96
int rd_weight_layout_multi(const rd_layout_job *jobs_host, int n_jobs, void *stream);
int rd_weight_layout_split_items(int Cout, int Cin, int taps, int kind);
int rd_weight_layout_split_multi(const rd_layout_job *jobs_dev, const int *chunk_job_dev, const int *chunk_group_dev, int n_chunks, void *stream);
int rd_conv_fwd_split(const void *in, int in_is_split, int in_rows, int Cin, const void *weight_k, int w_is_split, int taps,
                      const float *bias, float *out, int out_rows, int Cout, const rd_conv_index *idx, const float *scale,
                      const float *shift, const float *residual, int relu, float *stats, void *stream);
int rd_conv_wgrad_split(const void *in, int in_is_split, int in_rows, int Cin, const void *grad_out, int go_is_split, int out_rows,
                        int Cout, int taps, const rd_conv_index *idx, float *grad_wk, void *stream);

/* COMPOSITE calls: convolution (+ bias) -> train-mode BatchNorm (batch statistics from the convolution's epilogue) -> (+ residual) ->
 * activation, one call per layer and direction -- exactly the launches of rd_conv_fwd[_split] + rd_bn_train_fwd (forward) and
 * rd_bn_bwd + data gradient + rd_conv_wgrad (backward), in that order, without the host work between them.
 * w_format: 0 = fp32 kernel layout (rd_conv_fwd / rd_conv_dgrad: the data gradient reads the FORWARD weights transposed),
 *           1 = split format, 2 = fragment-major split format (rd_conv_fwd_split; backward: the [Cin][taps][Cout] operand).
 * stats (2 * Cout) and grad_gamma_beta (2 * Cout: [grad_gamma | grad_beta]) and grad_wk (Cout * taps * Cin) accumulate: zero-filled by
 * the caller.  side (4 * Cout) = [mean | rstd | scale | shift] written by the forward, read by the backward.  grad_in = NULL skips
 * the data gradient, grad_wk = NULL the weight gradient; side_stream = NULL keeps the weight gradient on main_stream, otherwise it
 * is launched on side_stream after that stream has been made to wait for main_stream (rd_stream_fork).  ev_*: optional hipEvent_t
 * handles recorded before / after the convolution launches on the stream they go to (NULL = none).
 * Replaces one PillarRes18 / DenseEnc / CMA layer of the reference: spconv or nn.Conv2d + nn.BatchNorm + ReLU / GELU
 * (pcdet/models/backbones_3d/spconv_backbone_2d.py:9-28,41-77, backbones_2d/base_bev_backbone.py:232-262) and their autograd. */
int rd_conv_bn_act_fwd(const float *in, int in_rows, int Cin, const void *weight, int w_format, int taps, const float *bias, float *raw,
                       int out_rows, int Cout, const rd_conv_index *idx, float *stats, const float *gamma, const float *beta, float eps,
                       float momentum, float *running_mean, float *running_var, const float *residual, int act, float *y, float *side,
                       void *ev0, void *ev1, void *stream);
int rd_conv_bn_act_bwd(const float *raw, const float *y, const float *grad_y, int out_rows, int Cout, const float *gamma, const float *side,
                       int act, int has_residual, float *grad_raw, float *grad_res, float *grad_gamma_beta, const void *w_dgrad, int w_format,
                       int taps, float *grad_in, int in_rows, int Cin, const rd_conv_index *bwd_idx, const float *in,
                       const rd_conv_index *fwd_idx, float *grad_wk, void *ev_d0, void *ev_d1, void *ev_w0, void *ev_w1, void *main_stream,
                       void *side_stream);

/* Geometry prelude of one branch as two calls around the step's single device->host read (composite.hip).  They issue exactly the
 * launches of rd_voxelize / rd_rankgrid_downsample_grid / rd_rankgrid_coords / rd_nbr_subm / rd_nbr_strided / rd_nbr_strided_T, in that
 * order; replaces the per-layer index-pair construction spconv does inside pillar_backbone's SubMConv2d / SparseConv2d(k3, s2, p1)
 * layers (pcdet/models/backbones_3d/spconv_backbone_2d.py:60-140, dynamic_pillar_vfe.py:193-240 for the pillar grid).
 *   rd_geometry_begin : rankgrid / point_row as rd_voxelize; rankgrid_down[l] (host array of n_down device pointers, each
 *       rd_rankgrid_bytes(batch * H_l * W_l) bytes, H_{l+1} = (H_l - 1) / 2 + 1) receives level l + 1; scalars_dev (int32, 2 + n_down)
 *       = {pillars, in-range points, rows of level 1, ..., rows of level n_down}.
 *   rd_geometry_finish: rankgrid[0 .. n_down], rows[0 .. n_down] (host) from the read; coords[l] (rows[l] x 3), nbr_subm[l] (rows[l] x 9),
 *       nbr_down[l] (rows[l+1] x 9: inputs of each output row) and, when nbr_up != NULL, nbr_up[l] (rows[l] x 9: the transposed table). */
int rd_geometry_begin(const float *points, int n_points, int n_feat, int batch, int gx, int gy, float x0, float y0, float vx, float vy,
                      uint32_t *rankgrid, int32_t *point_row, int n_down, uint32_t *const *rankgrid_down, int32_t *scalars_dev, void *stream);
int rd_geometry_finish(const uint32_t *const *rankgrid, int batch, int gx, int gy, int n_down, const int32_t *rows, int32_t *const *coords,
                       int32_t *const *nbr_subm, int32_t *const *nbr_down, int32_t *const *nbr_up, void *stream);

/* Measurement probe (bench.py): a register-only v_mfma_f32_32x32x16_bf16 loop on random operands, `waves_per_simd` (1 or 2) waves per SIMD
 * on every CU, `iters` x 8 independent MFMAs per wave.  *flops_out (host) = the flops the launch issues.  The rate it sustains is what
 * the chip's clock under MFMA load allows (MI355X_MICROARCH.md, DVFS give-back) -- the practical ceiling next to the 2.5 PF spec peak. */
int rd_probe_mfma_bf16(int iters, int waves_per_simd, float *out_dev, double *flops_out, void *stream);

/* Data gradient on the forward weights: grad_in[i][c] = sum_t sum_n grad_out[src_bwd(i,t)][n] * weight_k[n][t][c], with weight_k the
 * FORWARD kernel layout [Cout][taps][Cin] (the kernel reads it transposed; no re-laid-out copy) and idx the backward index
 * (transposed neighbour table / flip = 1 for sub-manifold, the transposed geometry for dense convolutions).  Cout % 32 == 0
 * (zero-pad narrower outputs and use rd_conv_fwd), Cin % 4 == 0; follows rd_set_conv_math like rd_conv_fwd.
 * Replaces spconv's backward-data implicit GEMM and cuDNN's conv backward-data (autograd of spconv_backbone_2d.py / Conv2d). */
int rd_conv_dgrad(const float *grad_out, int out_rows, int Cout, const float *weight_k, int taps, float *grad_in, int in_rows, int Cin,
                  const rd_conv_index *idx, void *stream);

/* Weight gradient: grad_wk[Cout][taps][Cin] += sum_j grad_out[j][:]^T (x) in[src(j,t)][:]  (atomic accumulation, caller zeroes).
 * Kernel layout only: each atomic wave-instruction then adds two contiguous 128-byte row segments, the shape global float atomics
 * run at full rate in (a tap-strided parameter layout would be ~17x slower; rd_weight_layout kinds 4/5 convert afterwards).
 * Cout % 32 == 0 or Cout < 32 handled by masking; Cin % 32 == 0. */
int rd_conv_wgrad(const float *in, int in_rows, int Cin, const float *grad_out, int out_rows, int Cout, int taps,
                  const rd_conv_index *idx, float *grad_wk, void *stream);

/* Weight layout transforms between parameter layouts and the kernel layout [Cout][taps][Cin].
 *   kind 0: spconv [Cout][kh][kw][Cin]      -> same memory (copy), flip=1 reverses taps
 *   kind 1: torch conv [Cout][Cin][kh][kw]  -> [Cout][taps][Cin]
 *   kind 2: data-gradient operand: from kernel layout [Cout][taps][Cin] -> [Cin][taps][Cout], flip reverses taps
 *   kind 3: torch ConvTranspose2d [Cin][Cout][kh][kw] -> kernel layout of the equivalent DENSE_T conv [Cout][taps][Cin]
 *   kind 4: inverse of kind 1 (kernel layout grad -> torch conv layout); kind 5: inverse of kind 3
 *   kind 6: kernel layout [Cout][taps][Cin] -> [taps][Cin][Cout] (DCN column-gradient operand)
 *   kind 7 / 8: data-gradient operand [Cin][taps][Cout] straight from the torch conv (7) / ConvTranspose2d (8) parameter layout */
int rd_weight_layout(const float *src, float *dst, int Cout, int Cin, int taps, int kind, int flip, void *stream);

/* column sums: out[C] += sum_j x[j][:] (bias gradients).  ACCUMULATES with fp32 atomics: the caller zero-fills out. */
int rd_colsum(const float *x, int64_t rows, int C, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * D. BatchNorm over rows (BatchNorm1d on sparse rows == BatchNorm2d on channels-last maps), fused with
 *    residual add and ReLU.  Replaces nn.BatchNorm1d/2d + replace_feature round trips
 *    (spconv_backbone_2d.py:61-77,80-112).
 * ---------------------------------------------------------------------------------------------- */
/* stats[2C] (sum, sumsq; e.g. from rd_conv_fwd or rd_bn_stats) -> mean/rstd (saved for backward), scale/shift,
 * running stats update (momentum, unbiased var), all on device.  count = number of rows. */
/* rd_bn_stats ACCUMULATES (sum, sumsq) into stats[2C] with fp32 atomics (one launch): the caller zero-fills stats. */
int rd_bn_stats(const float *x, int64_t rows, int C, float *stats /*[2C]*/, void *stream);
/* Train-mode forward in one launch: rd_bn_finalize + rd_affine_act fused (mean/rstd/scale/shift outputs may be NULL). */
int rd_bn_train_fwd(const float *x, int64_t rows, int C, const float *stats, const float *gamma, const float *beta, float eps,
                    float momentum, float *running_mean, float *running_var, const float *residual, int act, float *y,
                    float *mean, float *rstd, float *scale, float *shift, void *stream);
int rd_bn_finalize(const float *stats, int64_t rows, int C, const float *gamma, const float *beta, float eps, float momentum,
                   float *running_mean, float *running_var, float *mean, float *rstd, float *scale, float *shift, void *stream);
/* y = x*scale + shift (+ residual) ; act: 0 none, 1 relu, 2 gelu(erf). */
int rd_affine_act(const float *x, int64_t rows, int C, const float *scale, const float *shift, const float *residual,
                  int act, float *y, void *stream);
/* Backward of y = act(bn(x) + residual) in train mode.  Inputs: x (pre-BN), y (output, for the ReLU mask; may be NULL when
 * there is no residual: the mask is then re-derived as x*scale + shift > 0, the forward's own expression), the
 * pre-activation recomputed for GELU, grad_y.  Outputs grad_x, grad_gamma, grad_beta, grad_residual (= masked grad_y,
 * may be NULL).  grad_gamma / grad_beta are ACCUMULATED with fp32 atomics by the reduction pass and then read by the apply
 * pass: the caller zero-fills them. */
int rd_bn_bwd(const float *x, const float *y, const float *grad_y, int64_t rows, int C, const float *gamma,
              const float *mean, const float *rstd, const float *scale, const float *shift, int act, int has_residual,
              float *grad_x, float *grad_res, float *grad_gamma, float *grad_beta, void *stream);
/* SyncBatchNorm (tools/train.py:34,144-145: --sync_bn -> torch.nn.SyncBatchNorm.convert_sync_batchnorm).  The library never
 * communicates; the host all-reduces two small buffers per layer over its process group and these entries take the group-wide
 * values from DEVICE memory (no host round trip per layer):
 *   forward : stats[2C + 1] = (sum, sum of squares, row count) summed over the group -> rd_bn_train_fwd_sync / rd_bn_finalize_sync
 *             (`rows` = this rank's rows; mean / variance / running statistics use the group-wide count stats[2C]);
 *   backward: rd_bn_bwd_reduce accumulates THIS rank's (grad_gamma, grad_beta) (zero-filled by the caller) -- these are the
 *             parameter gradients, as torch's SyncBatchNorm returns them; the host all-reduces a copy; rd_bn_bwd_apply computes
 *             grad_x (and grad_res) from the group-wide sums and count_dev[0] = group-wide row count. */
int rd_bn_train_fwd_sync(const float *x, int64_t rows, int C, const float *stats /*[2C+1]*/, const float *gamma, const float *beta, float eps,
                         float momentum, float *running_mean, float *running_var, const float *residual, int act, float *y,
                         float *mean, float *rstd, float *scale, float *shift, void *stream);
int rd_bn_finalize_sync(const float *stats /*[2C+1]*/, int C, const float *gamma, const float *beta, float eps, float momentum,
                        float *running_mean, float *running_var, float *mean, float *rstd, float *scale, float *shift, void *stream);
int rd_bn_bwd_reduce(const float *x, const float *y, const float *grad_y, int64_t rows, int C, const float *mean, const float *rstd,
                     const float *scale, const float *shift, int act, int has_residual, float *grad_gamma, float *grad_beta, void *stream);
int rd_bn_bwd_apply(const float *x, const float *y, const float *grad_y, int64_t rows, int C, const float *gamma, const float *mean,
                    const float *rstd, const float *scale, const float *shift, int act, int has_residual, const float *sum_gamma,
                    const float *sum_beta, const float *count_dev, float *grad_x, float *grad_res, void *stream);

/* Channel concatenation of two channels-last row tensors, torch.cat((a, b), dim=1) of base_bev_backbone.py:296 and
 * radar_distill_final.py:121-124: out (rows, Ca + Cb) = [a | b]; rd_split2_rows is its backward, grad (rows, Ca + Cb) -> two
 * CONTIGUOUS gradients (ATen's cat backward returns strided slices that every consumer copies).  Ca, Cb multiples of 4. */
int rd_cat2_rows(const float *a, int Ca, const float *b, int Cb, int64_t rows, float *out, void *stream);
int rd_split2_rows(const float *grad, int64_t rows, int Ca, int Cb, float *grad_a, float *grad_b, void *stream);

/* ------------------------------------------------------------------------------------------------
 * E. Sparse -> dense BEV (SparseConvTensor.dense(), spconv_backbone_2d.py:299) in channels-last, and back.
 * ---------------------------------------------------------------------------------------------- */
int rd_rows_to_dense(const float *feats, const int32_t *coords, int n, int C, int batch, int H, int W, float *dense, void *stream);
int rd_dense_to_rows(const float *dense, const int32_t *coords, int n, int C, int batch, int H, int W, float *feats, void *stream);

/* ------------------------------------------------------------------------------------------------
 * F. DCNv2 (modulated deformable convolution), channels-last.  Replaces the `DCN` extension:
 *    modulated_deform_conv_forward / _backward (pcdet/ops/basicblock/src/vision.cpp:9-10,
 *    src/modulated_deform_conv.h:10-86, src/cuda/modulated_deform_conv_cuda.cu:19-280), deformable_groups = groups = 1.
 *    offset rows: channel 2t = dh, 2t+1 = dw of tap t; mask rows: channel t; both given as (pointer, row stride) so
 *    they can alias the fused 27-channel output of conv_offset_mask1; apply_sigmoid = 1 applies sigmoid to the mask.
 *    Forward  = rd_dcn_prep + rd_conv_fwd(index mode 3)            (bias is ALWAYS added, as in the reference)
 *    Backward = rd_conv_fwd (column gradient, weight kind 6) + rd_dcn_bwd_data + rd_conv_wgrad(index mode 3).
 *    (The training path uses the column form below; index mode 3 -- sampling fused into the GEMM's operand staging -- stays available.)
 * ---------------------------------------------------------------------------------------------- */
int rd_dcn_prep(const float *offset, int off_stride, const float *mask, int mask_stride, int apply_sigmoid, int B, int H, int W,
                int Ho, int Wo, int KH, int KW, int stride, int pad, int dil, int32_t *samp_idx, float *samp_w, void *stream);
/* Column form of the same convolution: col (out_rows, taps, C) = the bilinear-sampled, mask-modulated input rows of the sampling table
 * of rd_dcn_prep.  Forward = rd_dcn_prep + rd_dcn_columns + rd_conv_fwd(col as (out_rows, taps*C) rows, 1 tap, linear index);
 * weight gradient = rd_conv_wgrad over the same columns.  Replaces modulated_deformable_im2col_cuda
 * (pcdet/ops/basicblock/src/cuda/modulated_deform_im2col_cuda.cuh:127-194) for all B samples in one launch. */
int rd_dcn_columns(const float *x, int64_t in_rows, int C, const int32_t *samp_idx, const float *samp_w, int64_t out_rows, int taps,
                   float *col, void *stream);
/* x (B*H*W, C); colgrad (B*Ho*Wo, taps, C) = grad_out @ W per tap; outputs: grad_x (zeroed inside, atomics), grad of the
 * offset / mask rows (written with the given row strides; mask gradient is w.r.t. the pre-sigmoid value when apply_sigmoid). */
int rd_dcn_bwd_data(const float *x, int C, const float *colgrad, const float *offset, int off_stride, const float *mask, int mask_stride,
                    int apply_sigmoid, int B, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad, int dil,
                    float *grad_x, float *grad_offset, int goff_stride, float *grad_mask, int gmask_stride, void *stream);

/* ------------------------------------------------------------------------------------------------
 * G. Distillation losses over channels-last BEV maps.  Replaces Radar_Distill.low_loss (AFD) and high_loss (PFD),
 *    pcdet/models/backbones_2d/radar_distill_final.py:82-141.
 * ---------------------------------------------------------------------------------------------- */
int64_t rd_afd_ws_bytes(int64_t rows);
/* AFD of two radar maps against one lidar map in one pass.  out[4] = (feature_a, mask_a, feature_b, mask_b);
 * coef[6], rowinfo[2*rows*2] are saved for the backward. */
int rd_afd_fwd(const float *lidar, const float *radar_a, const float *radar_b, int64_t rows, int C, int batch, float *out, float *coef,
               float *rowinfo, float *ws, int64_t ws_bytes, void *stream);
/* gscale[4] (device) = upstream gradients of out[4]. */
/* rd_afd_fwd on maps stored as bf16 (section Q, BASELINE configs[2]); same outputs, sums in fp32. */
int rd_afd_fwd_bf16(const void *lidar, const void *radar_a, const void *radar_b, int64_t rows, int C, int batch, float *out, float *coef,
                    float *rowinfo, float *ws, int64_t ws_bytes, void *stream);
int rd_afd_bwd(const float *lidar, const float *radar_a, const float *radar_b, int64_t rows, int C, const float *rowinfo, const float *coef,
               const float *gscale, float *grad_a, float *grad_b, void *stream);
/* PFD: gt_hm / hm_logits (rows, n_hm) concatenated heat-map channels; cls[rows] int8, counts[2] int32 saved for backward;
 * out[1] = 0.5 * sum_cells w * (|softmax r1 - softmax l1|_1 + |softmax r2 - softmax l2|_1). */
int rd_pfd_fwd(const float *r1, const float *l1, const float *r2, const float *l2, int64_t rows, int C, const float *gt_hm,
               const float *hm_logits, int n_hm, int8_t *cls, int32_t *counts, float *out, float *ws, int64_t ws_bytes, void *stream);
int rd_pfd_bwd(const float *r1, const float *l1, const float *r2, const float *l2, int64_t rows, int C, const int8_t *cls,
               const int32_t *counts, const float *gscale, float *g1, float *g2, void *stream);

/* ------------------------------------------------------------------------------------------------
 * H. Rotated BEV overlap of aligned pairs.  Replaces iou3d_nms_cuda.boxes_aligned_overlap_bev_gpu(boxes_a (N,7),
 *    boxes_b (N,7), ans (N,1)) (pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:12, iou3d_nms.cpp:50-72).
 * ---------------------------------------------------------------------------------------------- */
int rd_boxes_aligned_overlap_bev(int n, const float *boxes_a, const float *boxes_b, float *ans_overlap, void *stream);

/* ------------------------------------------------------------------------------------------------
 * I. Optimizer: clip_grad_norm_ + OptimWrapper.step (decoupled decay + Adam) over all trainable tensors in two
 *    launches (tools/train_utils/train_utils.py:60-64, optimization/fastai_optim.py:135-152).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t numel;
} rd_opt_tensor;
int rd_opt_chunk_elems(void);   /* elements per chunk of the chunk table */
/* tensors_dev: device array of rd_opt_tensor; chunks_dev: device array of (tensor id, element offset) int32 pairs.
 * out2[0] = total 2-norm, out2[1] = clip coefficient min(1, max_norm / (norm + 1e-6)).
 * Data parallelism: rd_pack_grads gathers every t.grad into ONE flat buffer laid out like the moment buffers (tensor t at element
 * offset t.exp_avg - tensors[0].exp_avg); after the all-reduce (SUM) of that buffer, pass it as flat_grad with grad_scale =
 * 1 / world size to the two entries below (flat_grad = NULL, grad_scale = 1: gradients are read from t.grad). */
/* One bucket of the overlapped exchange: up to RD_PACK_LIST_MAX (src, dst, numel) copies in one launch, the list (HOST memory) travels
 * in the kernel arguments; src NULL packs zeros (no gradient on this rank). */
#define RD_PACK_LIST_MAX 128
typedef struct {
    const float *src;
    float *dst;
    int64_t numel;
} rd_pack_job;
int rd_pack_grads_list(const rd_pack_job *jobs_host, int n_jobs, void *stream);
int rd_pack_grads(const rd_opt_tensor *tensors_dev, const int32_t *chunks_dev, int n_chunks, float *flat, void *stream);
/* inv_loss_scale_dev (NULL = 1): device scalar 1 / S of mixed-precision loss scaling (torch.cuda.amp.GradScaler in
 * tools/train_utils/train_utils.py:23,57-64): gradients in memory are S times too large, norm and update use g / S.  max_norm <= 0:
 * no clipping (coefficient 1), the norm is still written (overflow check).  overflow_count_dev (NULL or one int32): incremented when
 * the norm is not finite; rd_adam_step subtracts it from `step` (a step GradScaler skips is not an Adam step). */
int rd_grad_norm(const rd_opt_tensor *tensors_dev, const int32_t *chunks_dev, int n_chunks, float max_norm, float *out2, float *ws,
                 int64_t ws_bytes, const float *flat_grad, float grad_scale, const float *inv_loss_scale_dev, int32_t *overflow_count_dev,
                 const float *present_dev, void *stream);
/* Data parallelism: which tensors take part in this step must be the same answer on every rank (torch's DistributedDataParallel
 * raises when a parameter is unused on one rank only; deciding from the rank-local `grad` pointer would give the ranks different
 * norms, clip coefficients and Adam step counts).  rd_grad_presence writes present[t] = 1.0 / 0.0 from the table's grad pointers; the
 * caller all-reduces it (MAX) over the ranks and passes it as present_dev (with flat_grad) to rd_grad_norm / rd_adam_step, which
 * then take a tensor's gradient from the flat buffer whenever ANY rank had one (absent ranks packed zeros).  present_dev = NULL: the
 * decision is the local pointer (single process). */
int rd_grad_presence(const rd_opt_tensor *tensors_dev, int n_tensors, float *present_dev, void *stream);
/* step = 1-based count of optimizer steps (bias correction).  clip_dev may be NULL (no clipping) or out2 of rd_grad_norm.
 * A tensor whose `grad` is NULL sits the step out as in torch.optim.Adam (only the decoupled decay p *= 1 - wd*lr of
 * OptimWrapper.step touches it; rd_grad_norm ignores it, rd_pack_grads packs zeros); skipped_dev (NULL = all zero, nothing counted)
 * holds per tensor how many steps it sat out so far, so that its own bias-correction count is step - skipped[t]; the launch itself
 * adds 1 for every tensor that sits THIS step out (ABI 3: the counters are device-owned).  Hyper-parameters are doubles: the
 * reference forms 1 - wd*lr, lr / (1 - beta1^t), sqrt(1 - beta2^t) in Python floats before they meet fp32 tensors.
 * skip_nonfinite = 1 (GradScaler.step): when clip_dev[0] (the gradient norm) is not finite the launch changes nothing. */
int rd_adam_step(const rd_opt_tensor *tensors_dev, const int32_t *chunks_dev, int n_chunks, double lr, double beta1, double beta2, double eps,
                 double weight_decay, int step, int32_t *skipped_dev, const float *clip_dev, const float *flat_grad, float grad_scale,
                 const float *inv_loss_scale_dev, int skip_nonfinite, const int32_t *overflow_count_dev, const float *present_dev,
                 void *stream);

/* ------------------------------------------------------------------------------------------------
 * J. Depthwise KxK convolution on channels-last maps (ConvNeXt dwconv 7x7, groups = C, padding K/2).  Replaces cuDNN's
 *    depthwise conv2d in pcdet/ops/basicblock/modules/Basicblock_convn.py:13,47.  weight_tc is [K*K][C] (tap-major);
 *    flip = 1 reads taps reversed (data gradient).  rd_dwconv_wgrad -> grad in the same [K*K][C] layout (pixel chunks are
 *    combined with atomics; ws / ws_bytes are unused since ABI 1 and may be NULL / 0, rd_dwconv_wgrad_ws_bytes returns 0).
 * ---------------------------------------------------------------------------------------------- */
int rd_dwconv_fwd(const float *in, const float *weight_tc, const float *bias, int B, int H, int W, int C, int K, int flip, float *out,
                  void *stream);
int64_t rd_dwconv_wgrad_ws_bytes(int B, int H, int W, int C, int K);
int rd_dwconv_wgrad(const float *in, const float *grad_out, int B, int H, int W, int C, int K, float *grad_w_tc, float *ws, int64_t ws_bytes,
                    void *stream);

/* ------------------------------------------------------------------------------------------------
 * K. CenterHead target assignment.  Replaces the host loops of Radar_CenterHead.assign_targets /
 *    assign_target_of_single_head (pcdet/models/dense_heads/radar_center_head.py:128-252) and draw_gaussian_to_heatmap
 *    (pcdet/models/model_utils/centernet_utils.py:38-69).  gt_boxes (B, M, box_dim) with the 1-based global class id in the
 *    last column (0 = padding).  Outputs (zeroed inside): heatmaps (B, n_channels, fy, fx) with the heads' classes
 *    concatenated on the channel axis; target_boxes (n_heads, B, max_objs, box_dim); inds / masks (n_heads, B, max_objs) int64;
 *    gt_box (n_heads, B, max_objs, 7).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int n_classes, n_heads, n_channels;
    int head_of_class[16];   /* global class id (1-based) -> head */
    int local_of_class[16];  /* global class id -> 0-based class index inside its head */
    int chan_off[8];         /* first heat-map channel of each head */
    float pcr0, pcr1, vs0, vs1;
    int stride, fx, fy, max_objs, min_radius;
    float overlap;
} rd_target_cfg;
int rd_center_targets(const float *gt_boxes, int B, int M, int box_dim, const rd_target_cfg *cfg, float *heatmaps, float *target_boxes,
                      int64_t *inds, int64_t *masks, float *gt_box, void *stream);

/* ------------------------------------------------------------------------------------------------
 * L. Narrow-output 3x3 convolutions of all CenterHead branches in one launch (vector ALUs; stride 1, zero padding 1).
 *    Replaces the per-branch final nn.Conv2d(64, n, 3, padding=1) of SeparateHead
 *    (pcdet/models/dense_heads/radar_center_head.py:39-40,57-58; center_head.py:23-24,43-44), n = 1..4, for NB <= 64 branches.
 *    y (B*H*W, ldy) channels-last rows holding every branch's 64 input channels at cin_off[b]; weight is the torch Conv2d layout
 *    [NO][64][3][3] of the branches' weights concatenated on dim 0 (branch b owns rows col_off[b] .. col_off[b]+n_out[b]);
 *    out / grad_out (B*H*W, NO).  cin_off / col_off / n_out are HOST arrays of NB ints (copied into the launch).
 *    rd_nconv_dgrad writes the 64 channels of every branch of grad_y (rows, ldy) (other columns untouched);
 *    rd_nconv_wgrad ACCUMULATES into grad_w [NO][64][3][3] (caller zero-fills).
 * ---------------------------------------------------------------------------------------------- */
int rd_nconv_fwd(const float *y, int ldy, const float *weight, const float *bias, int B, int H, int W, int NO, int NB,
                 const int32_t *cin_off, const int32_t *col_off, const int32_t *n_out, float *out, void *stream);
int rd_nconv_dgrad(const float *grad_out, const float *weight, int B, int H, int W, int NO, int NB, const int32_t *cin_off,
                   const int32_t *col_off, const int32_t *n_out, float *grad_y, int ldy, void *stream);
/* rd_nconv_dgrad FUSED with the backward of the train-mode BatchNorm + ReLU whose output y the narrow convolutions read (every
 * column of the (rows, ldy) tensors must belong to a branch): x = BatchNorm input, (mean, rstd, scale, shift) of its forward
 * (rd_bn_train_fwd), gamma may be NULL (= 1).  Writes grad_x (rows, ldy) and ACCUMULATES grad_gamma / grad_beta [ldy] (caller
 * zero-fills); grad_y is never materialised (two launches that recompute it: 1.05 GB instead of 2.1 GB of traffic for the 352 MB
 * tensor of 42 branches at B = 8).  Not available in deterministic mode (atomics). */
int rd_nconv_dgrad_bn(const float *grad_out, const float *weight, const float *x, const float *gamma, const float *mean, const float *rstd,
                      const float *scale, const float *shift, int B, int H, int W, int NO, int NB, const int32_t *cin_off,
                      const int32_t *col_off, const int32_t *n_out, float *grad_x, int ldy, float *grad_gamma, float *grad_beta, void *stream);
int rd_nconv_wgrad(const float *y, int ldy, const float *grad_out, int B, int H, int W, int NO, int NB, const int32_t *cin_off,
                   const int32_t *col_off, const int32_t *n_out, float *grad_w, void *stream);

/* ------------------------------------------------------------------------------------------------
 * M. Inference post-processing: rotated-BEV NMS and pairwise BEV overlap.
 *    rd_nms_bev replaces iou3d_nms_cuda.nms_gpu (pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:137-183, nms_kernel
 *    iou3d_nms_kernel.cu:295-340; Python wrapper iou3d_nms_utils.nms_gpu:119-137): boxes (n,7) [x,y,z,dx,dy,dz,heading]
 *    sorted by descending score; box j is suppressed by a kept box i < j when iou_bev(i, j) > thresh.  keep[0..*num_keep) =
 *    indices of the kept boxes in ascending order; both stay in DEVICE memory (the reference copies the n*n/8-byte bit matrix
 *    to the host and finishes on the CPU).  mask_ws: rd_nms_ws_bytes(n) bytes of scratch.
 *    rd_boxes_overlap_bev replaces boxes_overlap_bev_gpu (iou3d_nms.cpp:29-48): ans (na, nb) overlap areas (recall records).
 * ---------------------------------------------------------------------------------------------- */
int64_t rd_nms_ws_bytes(int n);
int rd_nms_bev(int n, const float *boxes_sorted, float thresh, void *mask_ws, int64_t ws_bytes, int64_t *keep, int32_t *num_keep,
               void *stream);
int rd_boxes_overlap_bev(int na, const float *boxes_a, int nb, const float *boxes_b, float *ans_overlap, void *stream);

/* ------------------------------------------------------------------------------------------------
 * N. Padded-voxel input format: hard voxeliser, PillarVFE, (PointPillarScatter = rd_rows_to_dense).
 *    rd_voxelize_hard replaces DataProcessor.transform_points_to_voxels -> VoxelGeneratorWrapper -> spconv Point2VoxelCPU3d
 *    (pcdet/datasets/processor/data_processor.py:16-61,142-229), batched: points (N, 1+C) [batch id, x, y, z, ...] sorted by
 *    batch id.  Per sample, in point order: voxels are created by their first in-range point until max_voxels exist, every
 *    voxel keeps its first max_points points.  Outputs (zero-filled inside, max_rows rows allocated by the caller,
 *    max_rows >= batch * max_voxels is always enough): voxels (M, max_points, C), coords (M, 4) = (b, z, y, x),
 *    num_points (M); *n_voxels = M (device).  Voxel order = sample, then first appearance; bit-exact vs the CPU algorithm.
 *    rd_pillar_vfe_{stats,max} replace PillarVFE.forward with one PFNLayer (pcdet/models/backbones_3d/vfe/pillar_vfe.py:8-123):
 *    stats[2*Cout] += (sum, sum of squares) of the Linear outputs over all real slots (BatchNorm1d batch statistics: divide by
 *    M*P, padded slots are exact zeros); max: out (M, Cout) = max over the P slots of relu(lin*scale + shift).
 *    weight (Cout, Cin) row-major (nn.Linear), Cin = (use_abs_xyz ? C : C-3) + 6 + with_distance, Cout <= 64, P <= 64.
 * ---------------------------------------------------------------------------------------------- */
int64_t rd_voxelize_hard_ws_bytes(int n_points, int batch, int gx, int gy, int gz);
int rd_voxelize_hard(const float *points, int n_points, int n_feat, int batch, int gx, int gy, int gz, float x0, float y0, float z0,
                     float vx, float vy, float vz, int max_points, int max_voxels, int64_t max_rows, float *voxels, int32_t *coords,
                     int32_t *num_points, int32_t *n_voxels, void *ws, int64_t ws_bytes, void *stream);
int rd_pillar_vfe_stats(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P, int C, const float *weight,
                        int Cin, int Cout, int use_abs_xyz, int with_distance, float vx, float vy, float vz, float xoff, float yoff,
                        float zoff, float *stats, void *stream);
int rd_pillar_vfe_max(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P, int C, const float *weight,
                      int Cin, int Cout, int use_abs_xyz, int with_distance, float vx, float vy, float vz, float xoff, float yoff,
                      float zoff, const float *scale, const float *shift, float *out, void *stream);
/* General PillarVFE path (training with gradients, several PFN layers, USE_NORM False; pillar_vfe.py:29-49,94-123):
 * rd_pillar_decorate materialises the masked slot features as rows: out (M*P, ld), columns [0, Cin) as above, [Cin, ld) zero
 * (ld = Cin rounded up to the implicit GEMM's K step).  Each PFNLayer is then rd_conv_fwd (1 tap) -> rd_bn_* -> rd_pfn_pool_fwd:
 * x (M*P, C) post-ReLU rows -> last != 0: out (M, C) = max over the P slots; last == 0: out (M*P, 2C) = [x | max repeated]
 * (torch.cat([x, x_max.repeat(1, P, 1)], dim=2)).  argmax (M, C) int32 = slot of the maximum (first on ties), for the backward:
 * grad_x (M*P, C) fully written = (last ? 0 : grad_out[:, :C]) + (slot == argmax ? sum over slots of the max's gradient : 0). */
int rd_pillar_decorate(const float *voxels, const int32_t *num_points, const int32_t *coords, int M, int P, int C, int Cin, int use_abs_xyz,
                       int with_distance, float vx, float vy, float vz, float xoff, float yoff, float zoff, int ld, float *out, void *stream);
int rd_pfn_pool_fwd(const float *x, int M, int P, int C, int last, float *out, int32_t *argmax, void *stream);
int rd_pfn_pool_bwd(const float *grad_out, const int32_t *argmax, int M, int P, int C, int last, float *grad_x, void *stream);

/* ------------------------------------------------------------------------------------------------
 * O. GELU + Global Response Normalisation of the ConvNeXt-V2 MLP (pcdet/ops/basicblock/modules/Basicblock_convn.py:46-60,83-85):
 *    z (B*hw, C) rows of B samples -> a = gelu(z) (saved), ssq (B, C) = per-sample column sums of a^2 (saved),
 *    out = gamma * (a * N) + beta + a with N = G / (mean_c G + 1e-6), G = sqrt(ssq).  Backward: grad_z, grad_gamma, grad_beta
 *    (S_ws: B*C floats of scratch).  All buffers are zero-filled inside where they accumulate.
 * ---------------------------------------------------------------------------------------------- */
int rd_gelu_grn_fwd(const float *z, int B, int64_t hw, int C, const float *gamma, const float *beta, float *a, float *ssq, float *out,
                    void *stream);
int rd_gelu_grn_bwd(const float *grad_out, const float *a, const float *z, const float *ssq, int B, int64_t hw, int C, const float *gamma,
                    float *S_ws, float *grad_z, float *grad_gamma, float *grad_beta, void *stream);

/* ---- P. CenterHead training loss, all task heads at once (replaces radar_center_head.py:258-330 get_loss: FocalLossCenterNet
 * loss_utils.py:169-200, RegLossCenterNet :203-250, per-cell decode :300-314, IouLoss :618-640, IouRegLoss :643-662 with
 * centernet_utils.bbox3d_overlaps_diou :462-497 and boxes_aligned_iou3d_gpu iou3d_nms_utils.py:83-117).
 * maps: (B, H, W, NO) fp32 channels-last, column groups [hm | center | center_z | dim | rot | vel | iou], heads inner
 * (column = base + head * width + j); heatmaps (B, n_ch, H, W); inds / masks (n_heads, B, K) int64; target_boxes (n_heads, B, K, target_dim >= 10);
 * gt_box (n_heads, B, K, gt_dim >= 7).  fwd: out[4*h + {0,1,2,3}] = hm / loc / iou / iou_reg loss of head h (weights applied as in
 * the reference: cls_w, loc_w * code_w, 1, 1), out[4*n_heads] = sum_h (hm + loc + iou + loc_w * iou_reg); scale (4*n_heads) and ws
 * (rd_center_loss_ws_floats floats) carry the normalisers and per-object gradients to bwd, which writes d out[4*n_heads] / d maps
 * times grad_loss[0] into grad_maps (every element written). */
typedef struct {
    int B, H, W, NO, n_heads, n_ch, K;
    int hm_c0, c0_center, c0_z, c0_dim, c0_rot, c0_vel, c0_iou;
    int head_of_ch[16];
    float code_w[10];
    float cls_w, loc_w;
    float stride, vs_x, vs_y, org_x, org_y; /* x = ((cell_x + center0) * stride) * vs_x + org_x */
} rd_center_loss_cfg;
int64_t rd_center_loss_ws_floats(const rd_center_loss_cfg *cfg);
int rd_center_loss_fwd(const rd_center_loss_cfg *cfg, const float *maps, const float *heatmaps, const int64_t *inds, const int64_t *masks,
                       const float *target_boxes, int target_dim, const float *gt_box, int gt_dim, float *out, float *scale, float *ws,
                       void *stream);
int rd_center_loss_bwd(const rd_center_loss_cfg *cfg, const float *maps, const float *heatmaps, const int64_t *inds, const int64_t *masks,
                       const float *scale, const float *ws, const float *grad_loss, float *grad_maps, void *stream);

/* ---- Q. Low-precision DenseEnc path (BASELINE configs[2] bf16 and configs[4] fp8): the frozen teacher's dense BEV convolutions
 * (BaseBEVBackboneV2, pcdet/models/backbones_2d/base_bev_backbone.py:206-308, eval mode: every layer is conv -> folded BatchNorm
 * -> ReLU) with activations and weights STORED as bf16 (dtype 0) or OCP fp8 e4m3fn (dtype 1) and accumulated in fp32 on the
 * matrix cores (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x64_f8f6f4).  Replaces cuDNN's fp32 / TF32 conv2d + ATen
 * batch_norm + relu for those modules.  All maps are channels-last rows (B*H*W, ld) of the narrow type.
 *   rd_lp_cast          x (rows, C) fp32 -> narrow, value * mul, written at column out_col0 of rows of pitch out_ld (concat by placement)
 *   rd_lp_uncast        n narrow elements -> fp32, value * mul
 *   rd_lp_amax          out1[0] = max |x| (calibration of the per-tensor activation scale)
 *   rd_lp_quant_weights w_k (Cout, K = taps * Cin) fp32 kernel layout -> narrow; fp8: per-output-channel scale w_scale[co] =
 *                       max|w[co]| / 448 and w_q = w / w_scale; bf16: w_scale = 1
 *   rd_lp_conv          ksize 3 (pad 1, stride 1), 1, or deconv = 1 with ksize 2 (ConvTranspose2d k2 s2: output map (2H, 2W),
 *                       weights [Cout][dy*2+dx][Cin]): out[p][out_col0 + co] = act(acc * alpha[co] + beta[co]), out_dtype 0 bf16,
 *                       1 fp8, 2 fp32.  alpha folds the input scale, the weight scale, the BatchNorm scale and 1 / output scale;
 *                       beta the BatchNorm shift / output scale.  Cin must be a multiple of 32 (bf16) / 64 (fp8). */
int rd_lp_cast(const float *x, int64_t rows, int C, int dtype, float mul, void *out, int out_ld, int out_col0, void *stream);
int rd_lp_uncast(const void *x, int64_t n, int dtype, float mul, float *out, void *stream);
int rd_lp_amax(const float *x, int64_t n, float *out1, void *stream);
int rd_lp_quant_weights(const float *w_k, int Cout, int K, int dtype, void *w_q, float *w_scale, void *stream);
int rd_lp_conv(const void *in, int dtype, int B, int H, int W, int Cin, int in_ld, const void *w_q, int ksize, int deconv, const float *alpha,
               const float *beta, int relu, void *out, int out_dtype, int Cout, int out_ld, int out_col0, void *stream);

/* ---- R. LayerNorm over the channel axis of channels-last rows (ConvNeXt block, pcdet/ops/basicblock/modules/Basicblock_convn.py:
 * 58-82: F.layer_norm(x, (C,), weight, bias, 1e-6) on (B, H, W, C)).  Replaces ATen native_layer_norm and its backward kernels.
 * fwd: y = (x - mean_c) * rstd_c * gamma + beta per row (biased variance), mean / rstd (rows each) kept for bwd.
 * bwd: grad_x (rows, C); grad_gamma / grad_beta (C each) are zero-filled and accumulated here.  C % 4 == 0, C <= 1024. */
int rd_layernorm_fwd(const float *x, int64_t rows, int C, const float *gamma, const float *beta, float eps, float *y, float *mean, float *rstd,
                     void *stream);
int rd_layernorm_bwd(const float *x, const float *grad_y, int64_t rows, int C, const float *gamma, const float *mean, const float *rstd,
                     float *grad_x, float *grad_gamma, float *grad_beta, void *stream);

#ifdef __cplusplus
}
#endif
#endif
