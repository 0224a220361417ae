// CenterHead training loss, all task heads, forward and backward in six launches.  Every sum runs in a fixed
// order (no floating-point atomics): loss values and gradients are bit-identical from run to run.
// Replaces the reference's per-head Python loop (pcdet/models/dense_heads/radar_center_head.py:258-330: FocalLossCenterNet
// utils/loss_utils.py:169-200 neg_loss_cornernet, RegLossCenterNet :203-250 _reg_loss, the decode of every cell :300-314, IouLoss
// :618-640 on boxes_aligned_iou3d_gpu, IouRegLoss :643-662 on centernet_utils.bbox3d_overlaps_diou :462-497) -- ~100 small ATen
// launches forward and ~250 backward, i.e. ~4 ms of HOST time per 30 ms step -- by
//   k_cl_focal      heat maps: clamp(sigmoid) focal terms, per-channel sums (one workgroup per channel, ordered tree)
//   k_cl_slots      one lane per (head, sample, object slot): gather the 11 regression channels at the object's cell, L1 terms,
//                   box decode, rotated aligned IoU target, axis-aligned DIoU, and the per-slot GRADIENTS of all three terms
//   k_cl_head_sums  the per-slot loss terms of each head added in slot order
//   k_cl_finalize   per-head normalisation, the four per-head losses, the total, the backward scale factors
//   k_cl_bwd_dense  d loss / d maps: focal gradient in the heat-map columns, zero elsewhere
//   k_cl_bwd_slots  per-slot gradients added into their cells (several objects may share a cell: its first slot adds them all)
// Maps are the batched-branch output: ONE channels-last (B, H, W, NO) tensor, columns [hm | center | center_z | dim | rot | vel | iou],
// heads inner (column = base + head * width + j).  Arithmetic follows the torch expressions term by term (fp32).
#include <algorithm>
#include "iou3d_dev.hpp"

using namespace rd;

namespace {

constexpr int CL_ACC = 13;   // per head: 10 L1 code sums, IoU-loss sum, DIoU-loss sum, positives

// One workgroup per heat-map channel: thread-private sums over its strided pixels, then a fixed-order tree -- every run adds the same
// terms in the same order (the first version combined all threads through LDS atomics: non-deterministic last bits in the loss).
__global__ __launch_bounds__(256) void k_cl_focal(const rd_center_loss_cfg c, const float *__restrict__ maps, const float *__restrict__ gt,
                                                  float *__restrict__ acc_ch) {
    __shared__ float s_sum[256], s_pos[256];
    const int ch = blockIdx.x;
    const int64_t n_pix = (int64_t)c.B * c.H * c.W;
    const int HW = c.H * c.W;
    float sum = 0.f, pos = 0.f;
    for (int64_t pix = threadIdx.x; pix < n_pix; pix += 256) {
        const int b = (int)(pix / HW), cell = (int)(pix - (int64_t)b * HW);
        const float x = maps[pix * c.NO + c.hm_c0 + ch];
        const float g = gt[((int64_t)b * c.n_ch + ch) * HW + cell];
        const float s = 1.f / (1.f + expf(-x));
        const float hm = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
        if (g == 1.f) {
            sum += logf(hm) * ((1.f - hm) * (1.f - hm));
            pos += 1.f;
        } else if (g < 1.f) {
            const float q = 1.f - g, q2 = q * q;
            sum += logf(1.f - hm) * (hm * hm) * (q2 * q2);
        }
    }
    s_sum[threadIdx.x] = sum;
    s_pos[threadIdx.x] = pos;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + d];
            s_pos[threadIdx.x] += s_pos[threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        acc_ch[ch] = s_sum[0];
        acc_ch[16 + ch] = s_pos[0];
    }
}

// min / max with torch's tie rule for the gradient: the selected operand gets 1, a tie 0.5 each
__device__ __forceinline__ float pick_lt(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

struct Diou {
    float val;
    float g[6];   // d val / d (x, y, z, dx, dy, dz) of the predicted box
};

__device__ inline Diou diou_with_grad(const float *p, const float *q) {
    Diou r;
    const float qminx = p[0] - 0.5f * p[3], qmaxx = p[0] + 0.5f * p[3], qminy = p[1] - 0.5f * p[4], qmaxy = p[1] + 0.5f * p[4];
    const float gminx = q[0] - 0.5f * q[3], gmaxx = q[0] + 0.5f * q[3], gminy = q[1] - 0.5f * q[4], gmaxy = q[1] + 0.5f * q[4];
    const float imaxx = fminf(qmaxx, gmaxx), iminx = fmaxf(qminx, gminx), imaxy = fminf(qmaxy, gmaxy), iminy = fmaxf(qminy, gminy);
    const float omaxx = fmaxf(qmaxx, gmaxx), ominx = fminf(qminx, gminx), omaxy = fmaxf(qmaxy, gmaxy), ominy = fminf(qminy, gminy);
    const float vp = p[3] * p[4] * p[5], vg = q[3] * q[4] * q[5];
    const float zt = p[2] + 0.5f * p[5], zb = p[2] - 0.5f * p[5], gzt = q[2] + 0.5f * q[5], gzb = q[2] - 0.5f * q[5];
    const float ihr = fminf(zt, gzt) - fmaxf(zb, gzb), ih = fmaxf(ihr, 0.f);
    const float ixr = imaxx - iminx, iyr = imaxy - iminy, ix = fmaxf(ixr, 0.f), iy = fmaxf(iyr, 0.f);
    const float vi = ix * iy * ih, vu = vg + vp - vi;
    const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
    const float idiag = dx * dx + dy * dy + dz * dz;
    const float ohr = fmaxf(gzt, zt) - fminf(gzb, zb), oh = fmaxf(ohr, 0.f);
    const float oxr = omaxx - ominx, oyr = omaxy - ominy, ox = fmaxf(oxr, 0.f), oy = fmaxf(oyr, 0.f);
    const float odiag = ox * ox + oy * oy + oh * oh;
    const float raw = vi / vu - idiag / odiag;
    r.val = fminf(fmaxf(raw, -1.f), 1.f);
    const float g_raw = (raw >= -1.f && raw <= 1.f) ? 1.f : 0.f;
    const float g_vu = -g_raw * vi / (vu * vu);
    const float g_vi = g_raw / vu - g_vu;                 // vu = vg + vp - vi
    const float g_vp = g_vu;
    const float g_idiag = -g_raw / odiag, g_odiag = g_raw * idiag / (odiag * odiag);
    const float g_ix = (ixr >= 0.f) ? g_vi * iy * ih : 0.f, g_iy = (iyr >= 0.f) ? g_vi * ix * ih : 0.f, g_ih = (ihr >= 0.f) ? g_vi * ix * iy : 0.f;
    const float g_ox = (oxr >= 0.f) ? g_odiag * 2.f * ox : 0.f, g_oy = (oyr >= 0.f) ? g_odiag * 2.f * oy : 0.f, g_oh = (ohr >= 0.f) ? g_odiag * 2.f * oh : 0.f;
    // x axis: qmax feeds imax (min) and omax (max); qmin feeds imin (max) and omin (min)
    const float g_qmaxx = g_ix * pick_lt(qmaxx, gmaxx) + g_ox * pick_lt(gmaxx, qmaxx);
    const float g_qminx = -g_ix * pick_lt(gminx, qminx) - g_ox * pick_lt(qminx, gminx);
    const float g_qmaxy = g_iy * pick_lt(qmaxy, gmaxy) + g_oy * pick_lt(gmaxy, qmaxy);
    const float g_qminy = -g_iy * pick_lt(gminy, qminy) - g_oy * pick_lt(qminy, gminy);
    const float g_zt = g_ih * pick_lt(zt, gzt) + g_oh * pick_lt(gzt, zt);
    const float g_zb = -g_ih * pick_lt(gzb, zb) - g_oh * pick_lt(zb, gzb);
    r.g[0] = g_qmaxx + g_qminx - 2.f * dx * g_idiag;
    r.g[1] = g_qmaxy + g_qminy - 2.f * dy * g_idiag;
    r.g[2] = g_zt + g_zb - 2.f * dz * g_idiag;
    r.g[3] = 0.5f * (g_qmaxx - g_qminx) + g_vp * p[4] * p[5];
    r.g[4] = 0.5f * (g_qmaxy - g_qminy) + g_vp * p[3] * p[5];
    r.g[5] = 0.5f * (g_zt - g_zb) + g_vp * p[3] * p[4];
    return r;
}

// slot_grad[s][17]: [0..9] d L1 / d code (code weight applied), [10..15] d (1 - diou) / d (center0, center1, center_z, dim0..2), [16] d iou term / d iou
__global__ __launch_bounds__(256) void k_cl_slots(const rd_center_loss_cfg c, const float *__restrict__ maps, const int64_t *__restrict__ inds,
                                                  const int64_t *__restrict__ masks, const float *__restrict__ tgt_boxes, int tgt_dim,
                                                  const float *__restrict__ gt_box, int gt_dim, float *__restrict__ slot_terms,
                                                  float *__restrict__ slot_grad) {
    const int n_slots = c.n_heads * c.B * c.K;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_slots) {
        const int h = s / (c.B * c.K), b = (s / c.K) % c.B;
        const int HW = c.H * c.W;
        const int64_t cell = inds[s];
        const bool valid = masks[s] != 0;
        const float mf = valid ? 1.f : 0.f;
        float *sg = slot_grad + (int64_t)s * 17;
        float pred[10];
        float ioup = 0.f;
        const bool cell_ok = cell >= 0 && cell < HW;           // targets.hip writes 0 for empty slots
        const float *px = maps + ((int64_t)b * HW + (cell_ok ? cell : 0)) * c.NO;
        pred[0] = px[c.c0_center + h * 2]; pred[1] = px[c.c0_center + h * 2 + 1];
        pred[2] = px[c.c0_z + h];
        pred[3] = px[c.c0_dim + h * 3]; pred[4] = px[c.c0_dim + h * 3 + 1]; pred[5] = px[c.c0_dim + h * 3 + 2];
        pred[6] = px[c.c0_rot + h * 2]; pred[7] = px[c.c0_rot + h * 2 + 1];
        pred[8] = px[c.c0_vel + h * 2]; pred[9] = px[c.c0_vel + h * 2 + 1];
        ioup = px[c.c0_iou + h];
        float *acc = slot_terms + (int64_t)s * CL_ACC;      // this slot's 13 loss terms; k_cl_finalize adds the slots in slot order
#pragma unroll
        for (int j = 0; j < CL_ACC; ++j) acc[j] = 0.f;
        // ---- L1 on the 10 regression codes: |pred*m - tgt*m|, m = mask * !isnan(tgt)
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const float t = tgt_boxes[(int64_t)s * tgt_dim + j];
            const float m = mf * (isnan(t) ? 0.f : 1.f);
            const float d = pred[j] * m - t * m;
            acc[j] = fabsf(d);
            sg[j] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * m * c.code_w[j];
        }
        float gd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gi = 0.f;
        if (valid) {
            acc[12] = 1.f;
            // ---- decode this cell (radar_center_head.py:300-314; the int() truncation of the range origin is in c.org_*)
            const int cy = (int)(cell / c.W), cx = (int)(cell - (int64_t)cy * c.W);
            float pb[7], gb[7];
            pb[0] = (((float)cx + pred[0]) * c.stride) * c.vs_x + c.org_x;
            pb[1] = (((float)cy + pred[1]) * c.stride) * c.vs_y + c.org_y;
            pb[2] = pred[2];
            float dclamp[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                dclamp[j] = (pred[3 + j] >= -5.f && pred[3 + j] <= 5.f) ? 1.f : 0.f;
                pb[3 + j] = expf(fminf(fmaxf(pred[3 + j], -5.f), 5.f));
            }
            pb[6] = atan2f(pred[7], pred[6]);
#pragma unroll
            for (int j = 0; j < 7; ++j) gb[j] = gt_box[(int64_t)s * gt_dim + j];
            // ---- IoU head: L1 to 2 * IoU3D(pred.detach(), gt) - 1  (boxes_aligned_iou3d_gpu)
            const float bev = overlap_area(pb, gb);
            const float oh = fmaxf(fminf(pb[2] + pb[5] / 2.f, gb[2] + gb[5] / 2.f) - fmaxf(pb[2] - pb[5] / 2.f, gb[2] - gb[5] / 2.f), 0.f);
            const float o3 = bev * oh;
            const float iou = o3 / fmaxf(pb[3] * pb[4] * pb[5] + gb[3] * gb[4] * gb[5] - o3, 1e-6f);
            const float di = ioup - (2.f * iou - 1.f);
            acc[10] = fabsf(di);
            gi = di > 0.f ? 1.f : (di < 0.f ? -1.f : 0.f);
            // ---- DIoU regression loss: 1 - diou
            const Diou dd = diou_with_grad(pb, gb);
            acc[11] = 1.f - dd.val;
            gd[0] = -dd.g[0] * (c.stride * c.vs_x);
            gd[1] = -dd.g[1] * (c.stride * c.vs_y);
            gd[2] = -dd.g[2];
#pragma unroll
            for (int j = 0; j < 3; ++j) gd[3 + j] = -dd.g[3 + j] * pb[3 + j] * dclamp[j];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) sg[10 + j] = gd[j];
        sg[16] = gi;
    }
}

// acc_head[h][13] = sum over the (sample, slot) pairs of head h of the per-slot terms: one workgroup per head, thread-private sums
// over strided slots + a fixed-order tree.
__global__ __launch_bounds__(256) void k_cl_head_sums(const rd_center_loss_cfg c, const float *__restrict__ slot_terms, float *__restrict__ acc_head) {
    __shared__ float s_red[256];
    const int h = blockIdx.x, per_head = c.B * c.K;
    const float *base = slot_terms + (int64_t)h * per_head * CL_ACC;
    for (int q = 0; q < CL_ACC; ++q) {
        float v = 0.f;
        for (int i = threadIdx.x; i < per_head; i += 256) v += base[(int64_t)i * CL_ACC + q];
        s_red[threadIdx.x] = v;
        __syncthreads();
        for (int d = 128; d >= 1; d >>= 1) {
            if ((int)threadIdx.x < d) s_red[threadIdx.x] += s_red[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) acc_head[h * CL_ACC + q] = s_red[0];
        __syncthreads();
    }
}

// out[h*4 + {0,1,2,3}] = hm / loc / iou / iou_reg loss of head h, out[4*nh] = total; scale[h*4 + {0..3}] = backward factors
__global__ void k_cl_finalize(const rd_center_loss_cfg c, const float *__restrict__ acc_ch, const float *__restrict__ acc_head,
                              float *__restrict__ out, float *__restrict__ scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = 0.f;
    for (int h = 0; h < c.n_heads; ++h) {
        float ssum = 0.f, npos = 0.f;
        for (int ch = 0; ch < c.n_ch; ++ch)
            if (c.head_of_ch[ch] == h) {
                ssum += acc_ch[ch];
                npos += acc_ch[16 + ch];
            }
        const float *a = acc_head + h * CL_ACC;
        const float n = a[12];
        const float hm = -ssum / fmaxf(npos, 1.f) * c.cls_w;
        float loc = 0.f;
        for (int j = 0; j < 10; ++j) loc += a[j] / fmaxf(n, 1.f) * c.code_w[j];
        loc *= c.loc_w;
        const float iou = a[10] / (n + 1e-4f), reg = a[11] / (n + 1e-4f);
        out[h * 4] = hm; out[h * 4 + 1] = loc; out[h * 4 + 2] = iou; out[h * 4 + 3] = reg;
        total += hm + loc + iou + c.loc_w * reg;
        scale[h * 4] = -c.cls_w / fmaxf(npos, 1.f);
        scale[h * 4 + 1] = c.loc_w / fmaxf(n, 1.f);
        scale[h * 4 + 2] = 1.f / (n + 1e-4f);
        scale[h * 4 + 3] = c.loc_w / (n + 1e-4f);
    }
    out[4 * c.n_heads] = total;
}

__global__ __launch_bounds__(256) void k_cl_bwd_dense(const rd_center_loss_cfg c, const float *__restrict__ maps, const float *__restrict__ gt,
                                                      const float *__restrict__ scale, const float *__restrict__ g_up, float *__restrict__ grad) {
    const int64_t total = (int64_t)c.B * c.H * c.W * c.NO;
    const int HW = c.H * c.W;
    const float up = g_up[0];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(e % c.NO);
        const int ch = col - c.hm_c0;
        float g = 0.f;
        if (ch >= 0 && ch < c.n_ch) {
            const int64_t pix = e / c.NO;
            const int b = (int)(pix / HW), cell = (int)(pix - (int64_t)b * HW);
            const float t = gt[((int64_t)b * c.n_ch + ch) * HW + cell];
            const float s = 1.f / (1.f + expf(-maps[e]));
            if (s >= 1e-4f && s <= 1.f - 1e-4f) {        // clamp passes the gradient inside [min, max]
                const float hm = s;
                float d = 0.f;
                if (t == 1.f) {
                    d = (1.f - hm) * (1.f - hm) / hm - 2.f * (1.f - hm) * logf(hm);
                } else if (t < 1.f) {
                    const float q = 1.f - t, q2 = q * q;
                    d = (2.f * hm * logf(1.f - hm) - hm * hm / (1.f - hm)) * (q2 * q2);
                }
                g = up * scale[c.head_of_ch[ch] * 4] * d * (s * (1.f - s));
            }
        }
        grad[e] = g;
    }
}

// Several objects of one head and sample may share a cell.  The FIRST valid slot of a cell owns it: it adds the gradients of all
// later slots of the same cell in slot order and writes once (plain adds onto k_cl_bwd_dense's output) -- no atomics, one fixed order.
__global__ __launch_bounds__(256) void k_cl_bwd_slots(const rd_center_loss_cfg c, const int64_t *__restrict__ inds, const int64_t *__restrict__ masks,
                                                      const float *__restrict__ slot_grad, const float *__restrict__ scale,
                                                      const float *__restrict__ g_up, float *__restrict__ grad) {
    const int n_slots = c.n_heads * c.B * c.K;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots || masks[s] == 0) return;
    const int h = s / (c.B * c.K), b = (s / c.K) % c.B, k = s % c.K;
    const int64_t cell = inds[s];
    if (cell < 0 || cell >= (int64_t)c.H * c.W) return;
    const int s0 = s - k;                                   // first slot of this (head, sample)
    for (int q = 0; q < k; ++q)
        if (masks[s0 + q] != 0 && inds[s0 + q] == cell) return;          // an earlier slot owns the cell
    const float up = g_up[0];
    const float s1 = up * scale[h * 4 + 1], si = up * scale[h * 4 + 2], sd = up * scale[h * 4 + 3];
    float add[11];
#pragma unroll
    for (int j = 0; j < 11; ++j) add[j] = 0.f;
    for (int q = k; q < c.K; ++q) {
        if (masks[s0 + q] == 0 || inds[s0 + q] != cell) continue;
        const float *sg = slot_grad + (int64_t)(s0 + q) * 17;
#pragma unroll
        for (int j = 0; j < 6; ++j) add[j] += s1 * sg[j] + sd * sg[10 + j];
#pragma unroll
        for (int j = 6; j < 10; ++j) add[j] += s1 * sg[j];
        add[10] += si * sg[16];
    }
    float *px = grad + ((int64_t)b * c.H * c.W + cell) * c.NO;
    px[c.c0_center + h * 2] += add[0];
    px[c.c0_center + h * 2 + 1] += add[1];
    px[c.c0_z + h] += add[2];
    px[c.c0_dim + h * 3] += add[3];
    px[c.c0_dim + h * 3 + 1] += add[4];
    px[c.c0_dim + h * 3 + 2] += add[5];
    px[c.c0_rot + h * 2] += add[6];
    px[c.c0_rot + h * 2 + 1] += add[7];
    px[c.c0_vel + h * 2] += add[8];
    px[c.c0_vel + h * 2 + 1] += add[9];
    px[c.c0_iou + h] += add[10];
}

int check_cfg(const rd_center_loss_cfg *c, const char *who) {
    RD_REQUIRE(c, "%s: null cfg", who);
    RD_REQUIRE(c->B > 0 && c->H > 0 && c->W > 0 && c->K > 0, "%s: bad sizes", who);
    RD_REQUIRE(c->n_heads >= 1 && c->n_heads <= 8 && c->n_ch >= 1 && c->n_ch <= 16, "%s: at most 8 heads / 16 heat-map channels", who);
    const int need = c->n_ch + 11 * c->n_heads;
    RD_REQUIRE(c->NO >= need, "%s: NO=%d columns < %d", who, c->NO, need);
    const int bases[7] = {c->hm_c0, c->c0_center, c->c0_z, c->c0_dim, c->c0_rot, c->c0_vel, c->c0_iou};
    const int widths[7] = {c->n_ch, 2 * c->n_heads, c->n_heads, 3 * c->n_heads, 2 * c->n_heads, 2 * c->n_heads, c->n_heads};
    for (int i = 0; i < 7; ++i) RD_REQUIRE(bases[i] >= 0 && bases[i] + widths[i] <= c->NO, "%s: column group %d outside the map", who, i);
    for (int i = 0; i < c->n_ch; ++i) RD_REQUIRE(c->head_of_ch[i] >= 0 && c->head_of_ch[i] < c->n_heads, "%s: bad head_of_ch", who);
    return RD_OK;
}

}  // namespace

extern "C" int64_t rd_center_loss_ws_floats(const rd_center_loss_cfg *cfg) {
    if (!cfg) return 0;
    return 32 + 8 * CL_ACC + (int64_t)cfg->n_heads * cfg->B * cfg->K * (17 + CL_ACC);
}

extern "C" int rd_center_loss_fwd(const rd_center_loss_cfg *cfg, const float *maps, const float *heatmaps, const int64_t *inds,
                                  const int64_t *masks, const float *target_boxes, int target_dim, const float *gt_box, int gt_dim,
                                  float *out, float *scale, float *ws, void *stream) {
    int rc = check_cfg(cfg, "rd_center_loss_fwd");
    if (rc) return rc;
    RD_REQUIRE(target_dim >= 10 && gt_dim >= 7, "rd_center_loss_fwd: target boxes need >= 10 codes, gt boxes >= 7 values");
    hipStream_t st = S(stream);
    const int n_slots = cfg->n_heads * cfg->B * cfg->K;
    float *acc_ch = ws, *acc_head = ws + 32, *slot_grad = ws + 32 + 8 * CL_ACC, *slot_terms = slot_grad + (int64_t)n_slots * 17;
    k_cl_focal<<<(unsigned)cfg->n_ch, 256, 0, st>>>(*cfg, maps, heatmaps, acc_ch);
    k_cl_slots<<<(unsigned)cdiv(n_slots, 256), 256, 0, st>>>(*cfg, maps, inds, masks, target_boxes, target_dim, gt_box, gt_dim, slot_terms, slot_grad);
    k_cl_head_sums<<<(unsigned)cfg->n_heads, 256, 0, st>>>(*cfg, slot_terms, acc_head);
    k_cl_finalize<<<1, 64, 0, st>>>(*cfg, acc_ch, acc_head, out, scale);
    return check_launch("rd_center_loss_fwd");
}

extern "C" int rd_center_loss_bwd(const rd_center_loss_cfg *cfg, const float *maps, const float *heatmaps, const int64_t *inds,
                                  const int64_t *masks, const float *scale, const float *ws, const float *grad_loss, float *grad_maps,
                                  void *stream) {
    int rc = check_cfg(cfg, "rd_center_loss_bwd");
    if (rc) return rc;
    hipStream_t st = S(stream);
    const float *slot_grad = ws + 32 + 8 * CL_ACC;
    const int64_t total = (int64_t)cfg->B * cfg->H * cfg->W * cfg->NO;
    k_cl_bwd_dense<<<(unsigned)std::min<int64_t>(cdiv(total, 256), 4096), 256, 0, st>>>(*cfg, maps, heatmaps, scale, grad_loss, grad_maps);
    const int n_slots = cfg->n_heads * cfg->B * cfg->K;
    k_cl_bwd_slots<<<(unsigned)cdiv(n_slots, 256), 256, 0, st>>>(*cfg, inds, masks, slot_grad, scale, grad_loss, grad_maps);
    return check_launch("rd_center_loss_bwd");
}
