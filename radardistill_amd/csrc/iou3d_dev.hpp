// Rotated-rectangle BEV overlap (device functions shared by iou3d.hip and nms.hip).  Algorithm and constants of the reference's
// box_overlap (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-225; EPS 1e-8, MARGIN 1e-2): clip polygon = 16 edge/edge
// intersections + corners inside the other box, sorted by angle, shoelace area.
#pragma once
#include "common.hpp"

namespace rd {

struct P2 { float x, y; };

__device__ __forceinline__ float crs3(P2 p1, P2 p2, P2 p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

__device__ __forceinline__ bool bbox_cross(P2 p1, P2 p2, P2 q1, P2 q2) {
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

__device__ __forceinline__ bool inside_box(const float *box, P2 p) {
    const float MARGIN = 1e-2f;
    const float c = cosf(-box[6]), s = sinf(-box[6]);
    const float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
    const float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
    return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}

__device__ __forceinline__ bool seg_x(P2 p1, P2 p0, P2 q1, P2 q0, P2 &ans) {
    if (!bbox_cross(p0, p1, q0, q1)) return false;
    const float s1 = crs3(q0, p1, p0), s2 = crs3(p1, q1, p0), s3 = crs3(p0, q1, q0), s4 = crs3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = crs3(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

__device__ __forceinline__ void spin(P2 c, float ca, float sa, P2 &p) {
    const float nx = (p.x - c.x) * ca + (p.y - c.y) * (-sa) + c.x;
    const float ny = (p.x - c.x) * sa + (p.y - c.y) * ca + c.y;
    p.x = nx; p.y = ny;
}

__device__ inline float overlap_area(const float *a, const float *b) {
    const float adx = a[3] / 2, ady = a[4] / 2, bdx = b[3] / 2, bdy = b[4] / 2;
    const P2 ca{a[0], a[1]}, cb{b[0], b[1]};
    P2 A[5] = {{a[0] - adx, a[1] - ady}, {a[0] + adx, a[1] - ady}, {a[0] + adx, a[1] + ady}, {a[0] - adx, a[1] + ady}, {0, 0}};
    P2 Bq[5] = {{b[0] - bdx, b[1] - bdy}, {b[0] + bdx, b[1] - bdy}, {b[0] + bdx, b[1] + bdy}, {b[0] - bdx, b[1] + bdy}, {0, 0}};
    const float aco = cosf(a[6]), asi = sinf(a[6]), bco = cosf(b[6]), bsi = sinf(b[6]);
    for (int k = 0; k < 4; ++k) { spin(ca, aco, asi, A[k]); spin(cb, bco, bsi, Bq[k]); }
    A[4] = A[0]; Bq[4] = Bq[0];
    P2 cp[16];
    P2 ctr{0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            P2 x;
            if (seg_x(A[i + 1], A[i], Bq[j + 1], Bq[j], x)) { cp[cnt++] = x; ctr.x += x.x; ctr.y += x.y; }
        }
    for (int k = 0; k < 4; ++k) {
        if (inside_box(a, Bq[k])) { ctr.x += Bq[k].x; ctr.y += Bq[k].y; cp[cnt++] = Bq[k]; }
        if (inside_box(b, A[k])) { ctr.x += A[k].x; ctr.y += A[k].y; cp[cnt++] = A[k]; }
    }
    ctr.x /= cnt; ctr.y /= cnt;
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (atan2f(cp[i].y - ctr.y, cp[i].x - ctr.x) > atan2f(cp[i + 1].y - ctr.y, cp[i + 1].x - ctr.x)) {
                P2 t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k)
        area += (cp[k].x - cp[0].x) * (cp[k + 1].y - cp[0].y) - (cp[k].y - cp[0].y) * (cp[k + 1].x - cp[0].x);
    return fabsf(area) / 2.0f;
}


// iou_bev (iou3d_nms_kernel.cu:227-234)
__device__ inline float iou_bev(const float *a, const float *b) {
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    const float so = overlap_area(a, b);
    return so / fmaxf(sa + sb - so, 1e-8f);
}

}  // namespace rd
