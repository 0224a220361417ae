/* Oracle: rotated-rectangle BEV overlap of aligned box pairs.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C restatement of the algorithm in the reference's
 *   pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:35-225  (cross, check_rect_cross, check_in_box2d,
 *       intersection, rotate_around_center, point_cmp, box_overlap)
 *   pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:266-277 (boxes_aligned_overlap_kernel)
 * in single precision, same operation order, EPS 1e-8, MARGIN 1e-2.
 * PARITY UNPINNED by reference tests (none exist); pinned by analytic known answers in
 * tests/test_oracle_kat.py.  Built by oracle/Makefile into oracle/_build/liboracle_iou3d.so.
 */
#include <math.h>
#include <stdlib.h>

typedef struct { float x, y; } pt;

static const float EPSF = 1e-8f;

static float cross3(pt p1, pt p2, pt p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static float cross2(pt a, pt b) { return a.x * b.y - a.y * b.x; }

static int rect_cross(pt p1, pt p2, pt q1, pt q2) {
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

static int in_box2d(const float *box, pt p) {
    const float MARGIN = 1e-2f;
    float cx = box[0], cy = box[1];
    float c = cosf(-box[6]), s = sinf(-box[6]);
    float rx = (p.x - cx) * c + (p.y - cy) * (-s);
    float ry = (p.x - cx) * s + (p.y - cy) * c;
    return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}

static int seg_intersection(pt p1, pt p0, pt q1, pt q0, pt *ans) {
    if (!rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = cross3(q0, p1, p0);
    float s2 = cross3(p1, q1, p0);
    float s3 = cross3(p0, q1, q0);
    float s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > EPSF) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

static void rot(pt c, float ca, float sa, pt *p) {
    float nx = (p->x - c.x) * ca + (p->y - c.y) * (-sa) + c.x;
    float ny = (p->x - c.x) * sa + (p->y - c.y) * ca + c.y;
    p->x = nx; p->y = ny;
}

static float box_overlap(const float *a, const float *b) {
    float a_ang = a[6], b_ang = b[6];
    float adx = a[3] / 2, bdx = b[3] / 2, ady = a[4] / 2, bdy = b[4] / 2;
    pt ca = {a[0], a[1]}, cb = {b[0], b[1]};
    pt A[5] = {{a[0] - adx, a[1] - ady}, {a[0] + adx, a[1] - ady}, {a[0] + adx, a[1] + ady}, {a[0] - adx, a[1] + ady}};
    pt B[5] = {{b[0] - bdx, b[1] - bdy}, {b[0] + bdx, b[1] - bdy}, {b[0] + bdx, b[1] + bdy}, {b[0] - bdx, b[1] + bdy}};
    float aco = cosf(a_ang), asi = sinf(a_ang), bco = cosf(b_ang), bsi = sinf(b_ang);
    for (int k = 0; k < 4; k++) { rot(ca, aco, asi, &A[k]); rot(cb, bco, bsi, &B[k]); }
    A[4] = A[0]; B[4] = B[0];
    pt cp[16], center = {0, 0};
    int cnt = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (seg_intersection(A[i + 1], A[i], B[j + 1], B[j], &cp[cnt])) {
                center.x += cp[cnt].x; center.y += cp[cnt].y; cnt++;
            }
    for (int k = 0; k < 4; k++) {
        if (in_box2d(a, B[k])) { center.x += B[k].x; center.y += B[k].y; cp[cnt++] = B[k]; }
        if (in_box2d(b, A[k])) { center.x += A[k].x; center.y += A[k].y; cp[cnt++] = A[k]; }
    }
    center.x /= cnt; center.y /= cnt;
    for (int j = 0; j < cnt - 1; j++)
        for (int i = 0; i < cnt - j - 1; i++)
            if (atan2f(cp[i].y - center.y, cp[i].x - center.x) > atan2f(cp[i + 1].y - center.y, cp[i + 1].x - center.x)) {
                pt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
            }
    float area = 0;
    for (int k = 0; k < cnt - 1; k++) {
        pt u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y}, v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
        area += cross2(u, v);
    }
    return fabsf(area) / 2.0f;
}

/* boxes_a, boxes_b: (n, 7) [x, y, z, dx, dy, dz, heading]; out: (n,) BEV overlap area of pair i. */
void oracle_boxes_aligned_overlap_bev(int n, const float *boxes_a, const float *boxes_b, float *out) {
    for (int i = 0; i < n; i++) out[i] = box_overlap(boxes_a + 7 * i, boxes_b + 7 * i);
}

/* (na, nb) overlap matrix: boxes_overlap_kernel (iou3d_nms_kernel.cu:236-249). */
void oracle_boxes_overlap_bev(int na, const float *boxes_a, int nb, const float *boxes_b, float *out) {
    for (int i = 0; i < na; i++)
        for (int j = 0; j < nb; j++) out[(long)i * nb + j] = box_overlap(boxes_a + 7 * i, boxes_b + 7 * j);
}

/* iou_bev (iou3d_nms_kernel.cu:227-234). */
static float iou_bev(const float *a, const float *b) {
    float sa = a[3] * a[4], sb = b[3] * b[4];
    float so = box_overlap(a, b);
    return so / fmaxf(sa + sb - so, 1e-8f);
}

/* nms_gpu (iou3d_nms.cpp:137-183 over nms_kernel, iou3d_nms_kernel.cu:295-340): boxes sorted by descending score; box j > i is
 * suppressed by a KEPT box i when iou_bev(i, j) > thresh.  The 64-wide bit matrix of the reference is only a storage format;
 * the greedy decision below is the same.  keep: indices of kept boxes (ascending); returns their number. */
int oracle_nms_bev(int n, const float *boxes, float thresh, long *keep) {
    unsigned char *removed = (unsigned char *)calloc(n > 0 ? n : 1, 1);
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        if (removed[i]) continue;
        keep[cnt++] = i;
        for (int j = i + 1; j < n; j++)
            if (!removed[j] && iou_bev(boxes + 7 * i, boxes + 7 * j) > thresh) removed[j] = 1;
    }
    free(removed);
    return cnt;
}
