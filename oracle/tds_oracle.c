/*
 * tds_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library.
 * The product (torchdrivesim_amd) never imports, links or executes it.
 *
 * Every function restates, in plain scalar C and in the reference's own operation order (one
 * IEEE-754 binary32 rounding per torch op; build with -ffp-contract=off), what the reference
 * computes with torch on the CPU.  Citations are file:line into the reference checkout
 * (torchdrivesim v0.2.3).  Pinning: tests/test_oracle_*.py check this file against the golden
 * vectors under tests/golden/ that tools/gen_golden.py captured from the imported reference.
 *
 * PARITY UNPINNED (no reference-derived vector exists in this environment):
 *   orc_fill_convex_poly / orc_line -- a restatement of OpenCV's cv2.fillConvexPoly
 *   (opencv-python, version unpinned by the reference, pyproject.toml:27; call site
 *   rendering/cv2.py:59).  OpenCV is not installed here and its source is not under
 *   /root/reference; the algorithm below follows modules/imgproc/src/drawing.cpp
 *   (FillConvexPoly, Line/LineIterator, clipLine) of OpenCV 4.x as published.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

static const float PI_F = 3.14159265358979323846f;

/* ------------------------------------------------------------------------------------------
 * R1  kinematics                                                       kinematic.py:400-523
 * ---------------------------------------------------------------------------------------- */

/* KinematicBicycle.step, kinematic.py:462-477.  state/out: n x [x,y,psi,v]; action n x [a,beta] normalised. */
ORC_API void orc_bicycle_step(const float *state, const float *action, const float *lr, float *out, int64_t n,
                              float dt, float max_acc, float max_steer, int left_handed) {
    for (int64_t i = 0; i < n; ++i) {
        float a = action[2 * i] * max_acc;          /* denormalize_action :459-460 */
        float beta = action[2 * i + 1] * max_steer;
        if (left_handed) beta = -beta;              /* :466-467 */
        float x = state[4 * i], y = state[4 * i + 1], psi = state[4 * i + 2], v = state[4 * i + 3];
        v = v + a * dt;                             /* :471 */
        float pb = psi + beta;
        x = x + (v * cosf(pb)) * dt;                /* :472 */
        y = y + (v * sinf(pb)) * dt;                /* :473 */
        psi = psi + ((v / lr[i]) * sinf(beta)) * dt; /* :474, no angle wrap */
        out[4 * i] = x; out[4 * i + 1] = y; out[4 * i + 2] = psi; out[4 * i + 3] = v;
    }
}

/* BicycleNoReversing.step, kinematic.py:513-523: acceleration clipped so speed never goes negative,
 * then re-normalised and fed to the plain bicycle step. */
ORC_API void orc_bicycle_norev_step(const float *state, const float *action, const float *lr, float *out, int64_t n,
                                    float dt, float max_acc, float max_steer, int left_handed) {
    for (int64_t i = 0; i < n; ++i) {
        float acc = action[2 * i] * max_acc, beta = action[2 * i + 1] * max_steer;
        float v = state[4 * i + 3];
        int reversing = (v + acc * dt) < 0.0f;
        float macc = reversing ? (-v) / dt : acc;
        float act[2] = { macc / max_acc, beta / max_steer };   /* normalize_action :456-457 */
        orc_bicycle_step(state + 4 * i, act, lr + i, out + 4 * i, 1, dt, max_acc, max_steer, left_handed);
    }
}

/* SimpleKinematicModel.step :362-367 and OrientedKinematicModel.step :384-389.
 * norm = [max_dx, max_dx, max_dpsi, max_dv]. */
ORC_API void orc_simple_step(const float *state, const float *action, float *out, int64_t n, float dt,
                             const float *norm, int oriented) {
    for (int64_t i = 0; i < n; ++i) {
        float a0 = action[4 * i], a1 = action[4 * i + 1];
        if (oriented) {                              /* utils.rotate :56-69: [[c,-s],[s,c]] @ v */
            float psi = state[4 * i + 2];
            float c = cosf(psi), s = sinf(psi);
            float r0 = c * a0 + (-s) * a1;
            float r1 = s * a0 + c * a1;
            a0 = r0; a1 = r1;
        }
        float act[4] = { a0, a1, action[4 * i + 2], action[4 * i + 3] };
        for (int k = 0; k < 4; ++k) out[4 * i + k] = state[4 * i + k] + (act[k] * norm[k]) * dt;
    }
}

static float signf_(float x) { return (x > 0.0f) - (x < 0.0f); }
static float remainder_py(float a, float b) {       /* torch.remainder: result has the sign of b */
    float r = fmodf(a, b);
    if (r != 0.0f && ((r < 0.0f) != (b < 0.0f))) r += b;
    return r;
}

/* KinematicBicycle.fit_action, kinematic.py:479-506. */
ORC_API void orc_bicycle_fit_action(const float *future, const float *current, float *action, int64_t n, float dt,
                                    float max_acc, float max_steer, int left_handed) {
    for (int64_t i = 0; i < n; ++i) {
        float vx = (future[4 * i] - current[4 * i]) / dt;
        float vy = (future[4 * i + 1] - current[4 * i + 1]) / dt;
        float v = sqrtf(vx * vx + vy * vy);
        float beta = atan2f(vy, vx) - current[4 * i + 2] * signf_(fabsf(v));
        beta = remainder_py(beta + PI_F, 2.0f * PI_F) - PI_F;
        int reversing = signf_(cosf(beta)) == -1.0f;
        v = sqrtf(vx * vx + vy * vy) * (reversing ? -1.0f : 1.0f);
        if (reversing) beta = beta - PI_F * signf_(beta);
        float a = (v - current[4 * i + 3]) / dt;
        if (left_handed) beta = -beta;
        action[2 * i] = a / max_acc;
        action[2 * i + 1] = beta / max_steer;
    }
}

/* ------------------------------------------------------------------------------------------
 * R2  box -> corners                                                _iou_utils.py:270-299
 * box = [x,y,len,wid,(psi)] with sin/cos supplied (sc = [sin,cos]) so that the caller decides
 * where the transcendental comes from (torch on the reference's device).
 * ---------------------------------------------------------------------------------------- */
static void box2corners(const float *box, float s, float c, float cor[4][2]) {
    static const float sx[4] = { 0.5f, -0.5f, -0.5f, 0.5f }, sy[4] = { 0.5f, 0.5f, -0.5f, -0.5f };
    for (int k = 0; k < 4; ++k) {
        float x4 = sx[k] * box[2], y4 = sy[k] * box[3];
        /* corners @ [[c,s],[-s,c]]  (bmm, one rounding per product and per sum) */
        float rx = x4 * c + y4 * (-s);
        float ry = x4 * s + y4 * c;
        cor[k][0] = rx + box[0];
        cor[k][1] = ry + box[1];
    }
}

ORC_API void orc_box2corners(const float *box5, const float *sc, float *corners, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        float cor[4][2];
        float s = sc ? sc[2 * i] : sinf(box5[5 * i + 4]), c = sc ? sc[2 * i + 1] : cosf(box5[5 * i + 4]);
        box2corners(box5 + 5 * i, s, c, cor);
        memcpy(corners + 8 * i, cor, sizeof(cor));
    }
}

/* ------------------------------------------------------------------------------------------
 * R3  Rotated-IoU                                                    _iou_utils.py:42-367
 * ---------------------------------------------------------------------------------------- */
static float precision_rounding(float x) {           /* :38-39, torch.round = half to even */
    return rintf(x * 1000000.0f) / 1000000.0f;
}

static void corners_in_box(float c1[4][2], float c2[4][2], int in[4]) {     /* box1_in_box2 :87-114 */
    float ax = c2[0][0], ay = c2[0][1];
    float abx = c2[1][0] - ax, aby = c2[1][1] - ay;
    float adx = c2[3][0] - ax, ady = c2[3][1] - ay;
    float norm_ab = abx * abx + aby * aby, norm_ad = adx * adx + ady * ady;
    const float lo = (float)(-1e-6), hi = (float)(1 + 1e-6);
    for (int k = 0; k < 4; ++k) {
        float amx = c1[k][0] - ax, amy = c1[k][1] - ay;
        float p_ab = abx * amx + aby * amy;
        float p_ad = adx * amx + ady * amy;
        float cond1 = precision_rounding(p_ab / norm_ab);
        float cond2 = precision_rounding(p_ad / norm_ad);
        in[k] = (cond1 > lo) && (cond1 < hi) && (cond2 > lo) && (cond2 < hi);
    }
}

typedef struct { float v[24][2]; int m[24]; int idx[9]; int nvalid; float area; } orc_iou_dbg;

static float intersection_area(float c1[4][2], float c2[4][2], orc_iou_dbg *dbg) {
    float vert[24][2];
    int mask[24];
    for (int k = 0; k < 4; ++k) { vert[k][0] = c1[k][0]; vert[k][1] = c1[k][1]; vert[4 + k][0] = c2[k][0]; vert[4 + k][1] = c2[k][1]; }
    /* box_intersection_th :42-84 */
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float x1 = c1[i][0], y1 = c1[i][1], x2 = c1[(i + 1) & 3][0], y2 = c1[(i + 1) & 3][1];
        float x3 = c2[j][0], y3 = c2[j][1], x4 = c2[(j + 1) & 3][0], y4 = c2[(j + 1) & 3][1];
        float num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
        float den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
        float t = den_t / num;
        if (fabsf(num) < (float)1e-4) t = -1.0f;
        int mask_t = (t > 0.0f) && (t < 1.0f);
        float den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3);
        float u = -den_u / num;
        if (fabsf(num) < (float)1e-4) u = -1.0f;
        int mask_u = (u > 0.0f) && (u < 1.0f);
        int mk = mask_t && mask_u;
        t = den_t / (num + (float)1e-8);
        float fm = mk ? 1.0f : 0.0f;
        vert[8 + i * 4 + j][0] = (x1 + t * (x2 - x1)) * fm;
        vert[8 + i * 4 + j][1] = (y1 + t * (y2 - y1)) * fm;
        mask[8 + i * 4 + j] = mk;
    }
    corners_in_box(c1, c2, mask);          /* c1_in_2 */
    corners_in_box(c2, c1, mask + 4);      /* c2_in_1 */

    /* sort_indices :160-227 */
    int nvalid = 0;
    for (int k = 0; k < 24; ++k) nvalid += mask[k];
    /* torch.sum over the strided dim of 24: four interleaved accumulators combined in order (probed) */
    float acc[4][2] = { { 0 } };
    for (int k = 0; k < 24; ++k) {
        float fm = mask[k] ? 1.0f : 0.0f;
        acc[k & 3][0] = acc[k & 3][0] + vert[k][0] * fm;
        acc[k & 3][1] = acc[k & 3][1] + vert[k][1] * fm;
    }
    float cx = (((acc[0][0] + acc[1][0]) + acc[2][0]) + acc[3][0]) / (float)nvalid;
    float cy = (((acc[0][1] + acc[1][1]) + acc[2][1]) + acc[3][1]) / (float)nvalid;
    float ang[24];
    for (int k = 0; k < 24; ++k) {
        float dx = vert[k][0] - cx, dy = vert[k][1] - cy;
        float r = sqrtf(dx * dx + dy * dy);
        float a = acosf(dx / r);
        ang[k] = (dy > 0.0f) ? a : (2.0f * PI_F - a);
    }
    int order[24];
    for (;;) {
        for (int k = 0; k < 24; ++k) order[k] = k;
        /* stable insertion sort, NaN and masked-out entries last (argsort ascending; ties are measure-zero) */
        for (int a = 1; a < 24; ++a) {
            int o = order[a];
            float key = mask[o] ? ang[o] : INFINITY;
            int b = a - 1;
            while (b >= 0) {
                float kb = mask[order[b]] ? ang[order[b]] : INFINITY;
                int gt = (isnan(kb) && !isnan(key)) || (kb > key);
                if (!gt) break;
                order[b + 1] = order[b];
                --b;
            }
            order[b + 1] = o;
        }
        if (nvalid <= 8) break;
        /* repair path :191-214: drop the first vertex of the closest consecutive pair */
        int best = 0;
        float bestd = INFINITY;
        for (int k = 0; k < nvalid - 1; ++k) {
            float dx = vert[order[k]][0] - vert[order[k + 1]][0], dy = vert[order[k]][1] - vert[order[k + 1]][1];
            float d = sqrtf(dx * dx + dy * dy);
            if (d < bestd) { bestd = d; best = k; }
        }
        mask[order[best]] = 0;
        --nvalid;
    }
    int pad = 8;                                                     /* :216 first invalid slot >= 8 */
    for (int k = 8; k < 24; ++k) if (!mask[k]) { pad = k; break; }
    int idx[9];
    for (int k = 0; k < 9; ++k) idx[k] = (nvalid < 3 || k >= nvalid) ? pad : order[k];
    idx[nvalid] = idx[0];                                            /* :224-225 (idx[0] is pad when nvalid<3) */
    /* calculate_area :230-247 */
    float total = 0.0f;
    for (int k = 0; k < 8; ++k) {
        float xa = vert[idx[k]][0], ya = vert[idx[k]][1], xb = vert[idx[k + 1]][0], yb = vert[idx[k + 1]][1];
        total = total + (xa * yb - ya * xb);
    }
    float area = fabsf(total) / 2.0f;
    if (dbg) {
        memcpy(dbg->v, vert, sizeof(vert)); memcpy(dbg->m, mask, sizeof(mask)); memcpy(dbg->idx, idx, sizeof(idx));
        dbg->nvalid = nvalid; dbg->area = area;
    }
    return area;
}

static float iou_pair(const float *b1, float s1, float c1s, const float *b2, float s2, float c2s, orc_iou_dbg *dbg) {
    float c1[4][2], c2[4][2];
    box2corners(b1, s1, c1s, c1);
    box2corners(b2, s2, c2s, c2);
    float inter = intersection_area(c1, c2, dbg);
    float area1 = b1[2] * b1[3], area2 = b2[2] * b2[3];
    float u = area1 + area2 - inter;
    return inter / u;                                                /* iou_differentiable_fast :344-367 */
}

/* iou_differentiable(box1, box2) elementwise over n pairs, boxes n x 5; sc1/sc2 n x [sin,cos] or NULL. */
ORC_API void orc_iou_pairs(const float *box1, const float *sc1, const float *box2, const float *sc2, float *iou,
                           int64_t n, int32_t *dbg_idx, int8_t *dbg_nvalid, float *dbg_area) {
    for (int64_t i = 0; i < n; ++i) {
        orc_iou_dbg d;
        float s1 = sc1 ? sc1[2 * i] : sinf(box1[5 * i + 4]), c1 = sc1 ? sc1[2 * i + 1] : cosf(box1[5 * i + 4]);
        float s2 = sc2 ? sc2[2 * i] : sinf(box2[5 * i + 4]), c2 = sc2 ? sc2[2 * i + 1] : cosf(box2[5 * i + 4]);
        iou[i] = iou_pair(box1 + 5 * i, s1, c1, box2 + 5 * i, s2, c2, &d);
        if (dbg_idx) for (int k = 0; k < 9; ++k) dbg_idx[9 * i + k] = d.idx[k];
        if (dbg_nvalid) dbg_nvalid[i] = (int8_t)d.nvalid;
        if (dbg_area) dbg_area[i] = d.area;
    }
}

/* R3d  discs                                                       infractions.py:378-426,503-545 */
static void box2discs(const float *b, float s, float c, float cen[5][2], float *r) {
    float len = b[2], wid = b[3];
    float rr = fminf(len, wid) / 2.0f;
    float half = fmaxf(len, wid) / 2.0f - rr;
    for (int i = -2; i <= 2; ++i) {
        float dx = ((float)i * half) / 2.0f;
        /* x = dx*cos - 0*sin ; y = dx*sin + 0*cos */
        cen[i + 2][0] = (dx * c - 0.0f * s) + b[0];
        cen[i + 2][1] = (dx * s + 0.0f * c) + b[1];
    }
    *r = rr;
}

static float discs_pair(const float *b1, float s1, float c1, const float *b2, float s2, float c2) {
    float ce1[5][2], ce2[5][2], r1, r2;
    box2discs(b1, s1, c1, ce1, &r1);
    box2discs(b2, s2, c2, ce2, &r2);
    float d = INFINITY;
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) {
        float dx = ce1[i][0] - ce2[j][0], dy = ce1[i][1] - ce2[j][1];
        /* torch.cdist (p=2, <=25 points: direct path) accumulates the squares with a fused multiply-add (probed) */
        float dd = sqrtf(fmaf(dy, dy, dx * dx));
        if (dd < d || isnan(dd)) d = dd;
    }
    float l = 1.0f - d / (r1 + r2);
    return l > 0.0f ? l : (isnan(l) ? l : 0.0f);                     /* relu keeps NaN */
}

/* discs use yaw + (pi/2)*(wid>len) (:404); the caller supplies sc of THAT angle or NULL to use libm. */
ORC_API void orc_discs_pairs(const float *box1, const float *sc1, const float *box2, const float *sc2, float *out, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        const float *a = box1 + 5 * i, *b = box2 + 5 * i;
        float ya = a[4] + (PI_F / 2.0f) * (a[3] > a[2] ? 1.0f : 0.0f), yb = b[4] + (PI_F / 2.0f) * (b[3] > b[2] ? 1.0f : 0.0f);
        float s1 = sc1 ? sc1[2 * i] : sinf(ya), c1 = sc1 ? sc1[2 * i + 1] : cosf(ya);
        float s2 = sc2 ? sc2[2 * i] : sinf(yb), c2 = sc2 ? sc2[2 * i + 1] : cosf(yb);
        out[i] = discs_pair(a, s1, c1, b, s2, c2);
    }
}

/* the same for num_discs = 2 * nps + 1 discs per box (bbox2discs :390-400: centres i * (max/2 - r) / nps), num_discs <= 25 */
ORC_API void orc_discs_pairs_n(const float *box1, const float *sc1, const float *box2, const float *sc2, float *out, int64_t n, int num_discs) {
    const int nps = (num_discs - 1) / 2;
    for (int64_t k = 0; k < n; ++k) {
        const float *a = box1 + 5 * k, *b = box2 + 5 * k;
        float ya = a[4] + (PI_F / 2.0f) * (a[3] > a[2] ? 1.0f : 0.0f), yb = b[4] + (PI_F / 2.0f) * (b[3] > b[2] ? 1.0f : 0.0f);
        float s1 = sc1 ? sc1[2 * k] : sinf(ya), c1 = sc1 ? sc1[2 * k + 1] : cosf(ya);
        float s2 = sc2 ? sc2[2 * k] : sinf(yb), c2 = sc2 ? sc2[2 * k + 1] : cosf(yb);
        float r1 = fminf(a[2], a[3]) / 2.0f, r2 = fminf(b[2], b[3]) / 2.0f;
        float h1 = fmaxf(a[2], a[3]) / 2.0f - r1, h2 = fmaxf(b[2], b[3]) / 2.0f - r2;
        float d = INFINITY;
        for (int i = -nps; i <= nps; ++i) {
            float da = ((float)i * h1) / (float)nps;
            float ax = (da * c1 - 0.0f * s1) + a[0], ay = (da * s1 + 0.0f * c1) + a[1];
            for (int j = -nps; j <= nps; ++j) {
                float db = ((float)j * h2) / (float)nps;
                float bx = (db * c2 - 0.0f * s2) + b[0], by = (db * s2 + 0.0f * c2) + b[1];
                float dx = ax - bx, dy = ay - by;
                float dd = sqrtf(fmaf(dy, dy, dx * dx));
                if (dd < d || isnan(dd)) d = dd;
            }
        }
        float l = 1.0f - d / (r1 + r2);
        out[k] = l > 0.0f ? l : (isnan(l) ? l : 0.0f);
    }
}

static float nan_to_num(float x) {
    if (isnan(x)) return 0.0f;
    if (isinf(x)) return x > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    return x;
}

/* Simulator.compute_collision, simulator.py:1064-1109,1161-1194.
 * boxes: B x N x 5 (all agents = exposed agents followed by NPCs), sc: B x N x 2 = [sin,cos] of the
 * angle the metric uses (psi for iou; psi + pi/2*(wid>len) for discs) or NULL, present: B x N (uint8).
 * out: B x A (the first A agents are the exposed ones).  metric: 0 = iou, 1 = discs.
 * collision_i = sum_j o_ij*present_j - max_j o_ij*present_j  (self overlap assumed to be the max). */
ORC_API void orc_collision(const float *boxes, const float *sc, const uint8_t *present, float *out,
                           int64_t B, int64_t A, int64_t N, int metric) {
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b) {
        float *bx = (float *)malloc(sizeof(float) * 5 * N);
        float *scl = (float *)malloc(sizeof(float) * 2 * N);
        for (int64_t j = 0; j < N; ++j) {
            for (int k = 0; k < 5; ++k) bx[5 * j + k] = nan_to_num(boxes[(b * N + j) * 5 + k]);
            float ang = bx[5 * j + 4];
            if (metric == 1) ang = ang + (PI_F / 2.0f) * (bx[5 * j + 3] > bx[5 * j + 2] ? 1.0f : 0.0f);
            /* a scrubbed NaN angle must see sin(0), cos(0), exactly as the reference after nan_to_num */
            int scrub = sc && isnan(boxes[(b * N + j) * 5 + 4]);
            scl[2 * j] = (sc && !scrub) ? sc[(b * N + j) * 2] : sinf(ang);
            scl[2 * j + 1] = (sc && !scrub) ? sc[(b * N + j) * 2 + 1] : cosf(ang);
        }
        for (int64_t i = 0; i < A; ++i) {
            float sum = 0.0f, mx = -INFINITY;
            for (int64_t j = 0; j < N; ++j) {
                float o = metric == 0
                    ? iou_pair(bx + 5 * i, scl[2 * i], scl[2 * i + 1], bx + 5 * j, scl[2 * j], scl[2 * j + 1], NULL)
                    : discs_pair(bx + 5 * i, scl[2 * i], scl[2 * i + 1], bx + 5 * j, scl[2 * j], scl[2 * j + 1]);
                o = nan_to_num(o) * (present[b * N + j] ? 1.0f : 0.0f);
                sum = sum + o;
                if (o > mx) mx = o;
            }
            out[b * A + i] = N > 0 ? sum - mx : 0.0f;
        }
        free(bx); free(scl);
    }
}

/* ------------------------------------------------------------------------------------------
 * R4  offroad: squared point -> mesh distance                        infractions.py:86-229
 * The reference lifts everything to 3-D with z = 0; in that plane t = 0 and p0 = p exactly, so
 * the restatement works on (x,y) with the same products/sums the 3-D dot products reduce to.
 * ---------------------------------------------------------------------------------------- */
static float dot2(float ax, float ay, float bx, float by) { return (ax * bx + ay * by) + 0.0f; }

static float point_segment_d2(float px, float py, float ax, float ay, float bx, float by) {   /* :147-159 */
    float ex = bx - ax, ey = by - ay;
    float l2 = dot2(ex, ey, ex, ey);
    float t = dot2(ex, ey, px - ax, py - ay) / (l2 + (float)1e-8);
    float tt = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
    float qx = ax + tt * ex, qy = ay + tt * ey;
    float d = dot2(px - qx, py - qy, px - qx, py - qy);
    if (l2 <= (float)1e-8) d = dot2(px - bx, py - by, px - bx, py - by);
    return d;
}

static float point_triangle_d2(float px, float py, const float *v0, const float *v1, const float *v2) {
    /* cross(v2-v0, v1-v0).z and its norm :102-103 */
    float cz = (v2[0] - v0[0]) * (v1[1] - v0[1]) - (v2[1] - v0[1]) * (v1[0] - v0[0]);
    float norm_normal = sqrtf(cz * cz);
    /* bary_centric_coords_3d :120-135 */
    float p0x = v1[0] - v0[0], p0y = v1[1] - v0[1], p1x = v2[0] - v0[0], p1y = v2[1] - v0[1];
    float p2x = px - v0[0], p2y = py - v0[1];
    float d00 = dot2(p0x, p0y, p0x, p0y), d01 = dot2(p0x, p0y, p1x, p1y), d11 = dot2(p1x, p1y, p1x, p1y);
    float d20 = dot2(p2x, p2y, p0x, p0y), d21 = dot2(p2x, p2y, p1x, p1y);
    float denom = d00 * d11 - d01 * d01 + (float)1e-8;
    float w1 = (d11 * d20 - d01 * d21) / denom;
    float w2 = (d00 * d21 - d01 * d20) / denom;
    float w0 = 1.0f - w1 - w2;
    int inside = (0.0f <= w0) && (w0 <= 1.0f) && (0.0f <= w1) && (w1 <= 1.0f) && (0.0f <= w2) && (w2 <= 1.0f);
    /* area_of_triangle :109-118 (hypot of a single non-zero component) */
    float area = fabsf(p0x * p1y - p0y * p1x) / 2.0f;
    if (area < (float)5e-3) inside = 0;
    float e01 = point_segment_d2(px, py, v0[0], v0[1], v1[0], v1[1]);
    float e02 = point_segment_d2(px, py, v0[0], v0[1], v2[0], v2[1]);
    float e12 = point_segment_d2(px, py, v1[0], v1[1], v2[0], v2[1]);
    float dist = fminf(fminf(e01, e02), e12);
    float cond = (inside && norm_normal > (float)1e-8) ? 1.0f : 0.0f;
    return (0.0f * 0.0f) * cond + dist * (1.0f - cond);               /* :170, t == 0 in the plane */
}

/* offroad_infraction_loss(use_pytorch3d=False) :176-229 then x present (simulator.py:1043-1044).
 * state B x A x 4, lenwid B x A x 2, sc B x A x 2 or NULL, verts Bm x V x 2, faces Bm x F x 3 (Bm = 1 or B),
 * present may be NULL. */
ORC_API void orc_offroad(const float *state, const float *lenwid, const float *sc, const uint8_t *present,
                         const float *verts, const int32_t *faces, int64_t B, int64_t A, int64_t V, int64_t F,
                         int64_t mesh_batch, float threshold, float *out) {
    #pragma omp parallel for schedule(dynamic, 4)
    for (int64_t ba = 0; ba < B * A; ++ba) {
        int64_t b = ba / A;
        const float *vb = verts + (mesh_batch > 1 ? b : 0) * V * 2;
        const int32_t *fb = faces + (mesh_batch > 1 ? b : 0) * F * 3;
        float box[5] = { state[4 * ba], state[4 * ba + 1], lenwid[2 * ba], lenwid[2 * ba + 1], state[4 * ba + 2] };
        float s = sc ? sc[2 * ba] : sinf(box[4]), c = sc ? sc[2 * ba + 1] : cosf(box[4]);
        float cor[4][2];
        box2corners(box, s, c, cor);
        float total = 0.0f;
        for (int k = 0; k < 4; ++k) {
            float best = INFINITY;
            int any_nan = 0;
            for (int64_t f = 0; f < F; ++f) {
                float d = point_triangle_d2(cor[k][0], cor[k][1], vb + 2 * fb[3 * f], vb + 2 * fb[3 * f + 1], vb + 2 * fb[3 * f + 2]);
                if (isnan(d)) any_nan = 1;                            /* torch.min propagates NaN */
                else if (d < best) best = d;
            }
            if (any_nan) best = NAN;
            best = isnan(best) ? 0.0f : nan_to_num(best);
            best = best > threshold ? best : 0.0f;                    /* F.threshold(d, thr, 0) :172 */
            total = total + best;
        }
        if (F == 0) total = 0.0f;                                     /* :197-198 */
        if (present) total = total * (present[ba] ? 1.0f : 0.0f);
        out[ba] = total;
    }
}

/* ------------------------------------------------------------------------------------------
 * R5  scene assembly + CV2 rendering                 mesh.py:911-1157, rendering/base.py, cv2.py
 * ---------------------------------------------------------------------------------------- */

/* Actor template, mesh.py:911-996 (render_agent_direction=True): per agent 7 verts
 * [ (l,w)/2, (l,-w)/2, (-l,-w)/2, (-l,w)/2, tip, base+, base- ], faces [0,1,3],[1,3,2],[4,5,6]. */
ORC_API void orc_actor_template(const float *lenwid, float *tmpl, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        float l = lenwid[2 * i], w = lenwid[2 * i + 1];
        float *t = tmpl + 14 * i;
        t[0] = l * 0.5f;  t[1] = w * 0.5f;
        t[2] = l * 0.5f;  t[3] = (-w) * 0.5f;
        t[4] = (-l) * 0.5f; t[5] = (-w) * 0.5f;
        t[6] = (-l) * 0.5f; t[7] = w * 0.5f;
        float off = l * (float)(0.5 - 0.3);                           /* :927-930 */
        t[8] = l * 0.3f + off;  t[9] = 0.0f + 0.0f;                   /* :922-926 after flip */
        t[10] = 0.0f + off;     t[11] = w * 0.5f + 0.0f;
        t[12] = 0.0f + off;     t[13] = (-w) * 0.5f + 0.0f;
    }
}

/* utils.transform :82-96: rotate(points, psi) + xy with rot = [[c,-s],[s,c]]. */
static void transform_pt(float px, float py, float s, float c, float x, float y, float *ox, float *oy) {
    *ox = (c * px + (-s) * py) + x;
    *oy = (s * px + c * py) + y;
}

/* ---- OpenCV restatement (PARITY UNPINNED, see header) ----
 * clip_line, orc_line and orc_fill_convex_poly below follow, step by step, cv::clipLine, cv::Line / LineIterator and
 * cv::FillConvexPoly of OpenCV's modules/imgproc/src/drawing.cpp.  OpenCV is Copyright (C) 2000-2024 the OpenCV team and its
 * contributors (Intel Corporation, Willow Garage Inc., NVIDIA Corporation, Advanced Micro Devices Inc., OpenCV Foundation, Itseez
 * Inc., Xperience AI, Shenzhen Institute of Artificial Intelligence and Robotics for Society) and is distributed under the Apache
 * License 2.0 (OpenCV 4.5.0 and later; the 3-clause BSD licence before that): https://opencv.org/license/.  This derived
 * restatement is used as a test oracle only and is offered under the same terms.
 * The live check against the real library is oracle/opencv_check.py (runs wherever `cv2` imports). */
typedef struct { float *img; int W, H; } orc_img;      /* raw OpenCV image: img[(y*W + x)*3 + ch] */

static void put_px(orc_img *im, int x, int y, const float *col) {
    float *p = im->img + ((int64_t)y * im->W + x) * 3;
    p[0] = col[0]; p[1] = col[1]; p[2] = col[2];
}

/* cv::clipLine(Size, Point2l&, Point2l&), drawing.cpp */
static int clip_line(int W, int H, int64_t *x1, int64_t *y1, int64_t *x2, int64_t *y2) {
    int c1, c2;
    int64_t right = W - 1, bottom = H - 1;
    if (W <= 0 || H <= 0) return 0;
    c1 = (*x1 < 0) + (*x1 > right) * 2 + (*y1 < 0) * 4 + (*y1 > bottom) * 8;
    c2 = (*x2 < 0) + (*x2 > right) * 2 + (*y2 < 0) * 4 + (*y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            *x1 += (int64_t)((double)(a - *y1) * (*x2 - *x1) / (*y2 - *y1));
            *y1 = a;
            c1 = (*x1 < 0) + (*x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            *x2 += (int64_t)((double)(a - *y2) * (*x2 - *x1) / (*y2 - *y1));
            *y2 = a;
            c2 = (*x2 < 0) + (*x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                *y1 += (int64_t)((double)(a - *x1) * (*y2 - *y1) / (*x2 - *x1));
                *x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                *y2 += (int64_t)((double)(a - *x2) * (*y2 - *y1) / (*x2 - *x1));
                *x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

/* cv::Line (8-connected) = LineIterator(img, pt1, pt2, 8, leftToRight=true) */
static void orc_line(orc_img *im, int ax, int ay, int bx, int by, const float *col) {
    int64_t x1 = ax, y1 = ay, x2 = bx, y2 = by;
    if ((uint64_t)x1 >= (uint64_t)im->W || (uint64_t)x2 >= (uint64_t)im->W ||
        (uint64_t)y1 >= (uint64_t)im->H || (uint64_t)y2 >= (uint64_t)im->H) {
        if (!clip_line(im->W, im->H, &x1, &y1, &x2, &y2)) return;
    }
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    int px = (int)x1, py = (int)y1;
    int step_x = 1, step_y = 1;
    if (dx < 0) { dx = -dx; dy = -dy; px = (int)x2; py = (int)y2; }        /* leftToRight */
    if (dy < 0) { dy = -dy; step_y = -1; }
    int vert = dy > dx;
    if (vert) { int t = dx; dx = dy; dy = t; }
    int err = dx - (dy + dy), plus_delta = dx + dx, minus_delta = -(dy + dy);
    int count = dx + 1;
    for (int i = 0; i < count; ++i) {
        put_px(im, px, py, col);
        int neg = err < 0;
        err += minus_delta + (neg ? plus_delta : 0);
        if (vert) { py += step_y; if (neg) px += step_x; }
        else { px += step_x; if (neg) py += step_y; }
    }
}

/* one directed cv::Line call on a raw image (tests/fill_rows_model.c draws single edges with it: clipLine depends on the order of the end points) */
ORC_API void orc_line_px(float *img, int W, int H, int ax, int ay, int bx, int by, const float *col) {
    orc_img im = { img, W, H };
    orc_line(&im, ax, ay, bx, by, col);
}

/* cv::FillConvexPoly, shift = 0, line_type = 8 (LINE_AA is downgraded for non-8-bit images). */
ORC_API void orc_fill_convex_poly(float *img, int W, int H, const int32_t *pts, int npts, const float *col) {
    enum { XY_SHIFT = 16 };
    const int64_t XY_ONE = 1 << XY_SHIFT;
    orc_img im = { img, W, H };
    struct { int idx, di; int64_t x, dx; int ye; } edge[2];
    int i, y, imin = 0;
    int edges = npts;
    int64_t xmin, xmax, ymin, ymax;
    const int delta1 = (int)(XY_ONE >> 1), delta2 = (int)(XY_ONE >> 1);
    if (npts <= 0) return;
    int64_t p0x = pts[2 * (npts - 1)], p0y = pts[2 * (npts - 1) + 1];
    xmin = xmax = pts[0];
    ymin = ymax = pts[1];
    for (i = 0; i < npts; ++i) {
        int64_t px = pts[2 * i], py = pts[2 * i + 1];
        if (py < ymin) { ymin = py; imin = i; }
        if (py > ymax) ymax = py;
        if (px > xmax) xmax = px;
        if (px < xmin) xmin = px;
        orc_line(&im, (int)p0x, (int)p0y, (int)px, (int)py, col);
        p0x = px; p0y = py;
    }
    if (npts < 3 || (int)xmax < 0 || (int)ymax < 0 || (int)xmin >= W || (int)ymin >= H) return;
    if (ymax > H - 1) ymax = H - 1;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y = (int)ymin;
    edge[0].di = 1;
    edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -XY_ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (i = 0; i < 2; ++i) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                int ty = 0;
                for (; edges-- > 0;) {
                    ty = pts[2 * idx + 1];
                    if (ty > y) {
                        int64_t xs = (int64_t)pts[2 * idx0] << XY_SHIFT;
                        int64_t xe = (int64_t)pts[2 * idx] << XY_SHIFT;
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + delta1) >> XY_SHIFT);
            int xx2 = (int)((edge[right].x + delta2) >> XY_SHIFT);
            if (xx2 >= 0 && xx1 < W) {
                if (xx1 < 0) xx1 = 0;
                if (xx2 >= W) xx2 = W - 1;
                for (int x = xx1; x <= xx2; ++x) put_px(&im, x, y, col);
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

/* utils.is_inside_polygon :99-122 for a 4-gon */
static int inside_quad(float px, float py, float poly[4][2]) {
    int all_right = 1, all_left = 1;
    for (int k = 0; k < 4; ++k) {
        float x0 = poly[k][0], y0 = poly[k][1], x1 = poly[(k + 1) & 3][0], y1 = poly[(k + 1) & 3][1];
        float a = y1 - y0, b = x0 - x1;
        float c = (-a) * x0 - b * y0;
        int right = ((a * px + b * py) + c) >= 0.0f;
        all_right &= right;
        all_left &= !right;
    }
    return all_right || all_left;
}

/* Cameras.reverse_transform_points_screen of the 4 image corners, then the 1.05x safety margin
 * (rendering/cv2.py:32-40, base.py:117-130); camera xy is already zero after the shift. */
static void viewing_polygon(float s, float c, float scale, int W, int H, float poly[4][2]) {
    const float cor[4][2] = { { 0, 0 }, { 0, (float)H }, { (float)W, (float)H }, { (float)W, 0 } };
    float mn = (float)(H < W ? H : W) / 2.0f;
    float sumx = 0.0f, sumy = 0.0f;
    for (int k = 0; k < 4; ++k) {
        float x = cor[k][0] - (float)W / 2.0f, y = cor[k][1] - (float)H / 2.0f;
        x = x / mn; y = y / mn;
        x = (-x) / scale; y = (-y) / scale;
        /* rot_mat^T = [[c,-s],[s,c]] */
        float rx = c * x + (-s) * y;
        float ry = s * x + c * y;
        poly[k][0] = rx + 0.0f; poly[k][1] = ry + 0.0f;
        sumx = sumx + poly[k][0]; sumy = sumy + poly[k][1];
    }
    float cx = sumx / 4.0f, cy = sumy / 4.0f;
    for (int k = 0; k < 4; ++k) {
        poly[k][0] = cx + (poly[k][0] - cx) * 1.05f;
        poly[k][1] = cy + (poly[k][1] - cy) * 1.05f;
    }
}

/* Cameras.transform_points_screen, base.py:102-115, camera at the origin; then .to(int32) cv2.py:48 */
static void project_px(float vx, float vy, float s, float c, float scale, int W, int H, int32_t *ox, int32_t *oy) {
    float x = vx - 0.0f, y = vy - 0.0f;
    float rx = c * x + s * y;
    float ry = (-s) * x + c * y;
    rx = (-rx) * scale; ry = (-ry) * scale;
    float mn = (float)(H < W ? H : W);
    rx = (rx * mn) / 2.0f; ry = (ry * mn) / 2.0f;
    rx = rx + (float)W / 2.0f; ry = ry + (float)H / 2.0f;
    *ox = (int32_t)rx; *oy = (int32_t)ry;
}

typedef struct { float z; uint32_t rgb; int32_t f; } zface;
/* Painter order: z descending (cv2.py:47).  torch.argsort is not stable, so the reference leaves the order of
 * equal-z faces unspecified; the documented tie-break of this project is: packed colour
 * 0x00RRGGBB ascending (the larger colour is drawn later and wins), then original face index. */
static int zcmp(const void *a, const void *b) {
    const zface *p = (const zface *)a, *q = (const zface *)b;
    if (p->z > q->z) return -1;
    if (p->z < q->z) return 1;
    if (p->rgb != q->rgb) return p->rgb < q->rgb ? -1 : 1;
    return (p->f > q->f) - (p->f < q->f);
}

static void quantise_color(const float *attr, float col[3], uint32_t *packed) {       /* cv2.py:50 */
    *packed = 0;
    for (int ch = 0; ch < 3; ++ch) {
        float q = floorf((attr[ch] * (float)(1.0 - 1e-3)) * 256.0f);
        col[ch] = (float)(uint8_t)q;
        *packed = (*packed << 8) | (uint8_t)q;
    }
}

/* CV2RendererConfig.trim_mesh_before_rendering (cv2.py:15): 1 = mesh.trim(viewing_polygon) before drawing (cv2.py:32-41, the default),
 * 0 = every face goes to fillConvexPoly.  A process-wide switch of the oracle (test infrastructure). */
static int g_trim_mesh = 1;
ORC_API void orc_set_trim_mesh(int on) { g_trim_mesh = on; }

/* CV2Renderer.render_rgb_mesh, rendering/cv2.py:27-70, for ONE image.
 * verts V x 3 (x,y,z), attrs V x 3 in [0,1], faces F x 3, camera (cx,cy,sin,cos).
 * image: H x W x 3 float, ALREADY transposed as the reference returns it: image[px][py][ch].
 * If rec_tris != NULL the ordered pre-raster call list is written there (cap rec_cap calls): 6 ints + 3 colour. */
/* scratch of one render call; render_scratch_* let a caller that renders many images of one size reuse it (large blocks come from mmap in
 * glibc: allocating and zero-filling 2.4 MB per image from hundreds of threads serialises on the process's address-space lock) */
typedef struct { float *sv; uint8_t *ins; zface *order; float *raw; int64_t V, F; int W, H; } render_scratch;
static void render_scratch_init(render_scratch *rs, int64_t V, int64_t F, int W, int H) {
    rs->V = V; rs->F = F; rs->W = W; rs->H = H;
    rs->sv = (float *)malloc(sizeof(float) * 2 * (V > 0 ? V : 1));
    rs->ins = (uint8_t *)malloc(V > 0 ? V : 1);
    rs->order = (zface *)malloc(sizeof(zface) * (F > 0 ? F : 1));
    rs->raw = (float *)malloc(sizeof(float) * (size_t)W * H * 3);
}
static void render_scratch_free(render_scratch *rs) { free(rs->sv); free(rs->ins); free(rs->order); free(rs->raw); }

static int64_t render_rgb_mesh_one(render_scratch *rs, const float *verts, const float *attrs, const int32_t *faces, int64_t V, int64_t F,
                                   float cx, float cy, float s, float c, float scale, int W, int H,
                                   float *image, int32_t *rec_tris, uint8_t *rec_cols, int64_t rec_cap) {
    float poly[4][2];
    viewing_polygon(s, c, scale, W, H, poly);
    float *sv = rs->sv;
    uint8_t *ins = rs->ins;
    for (int64_t v = 0; v < V; ++v) {
        sv[2 * v] = verts[3 * v] + (-cx);                             /* mesh.translate(-cameras.xy) cv2.py:29-31 */
        sv[2 * v + 1] = verts[3 * v + 1] + (-cy);
        ins[v] = (uint8_t)inside_quad(sv[2 * v], sv[2 * v + 1], poly);
    }
    zface *order = rs->order;
    int64_t nk = 0;
    for (int64_t f = 0; f < F; ++f) {                                 /* mesh.trim: keep iff >= 1 vertex inside */
        const int32_t *fv = faces + 3 * f;
        if (!g_trim_mesh || ins[fv[0]] || ins[fv[1]] || ins[fv[2]]) {
            float tmpc[3];
            order[nk].z = verts[3 * fv[0] + 2]; order[nk].f = (int32_t)f;
            quantise_color(attrs + 3 * fv[0], tmpc, &order[nk].rgb);
            ++nk;
        }
    }
    qsort(order, nk, sizeof(zface), zcmp);                            /* painter order, cv2.py:44-47 */
    float *raw = rs->raw;                                              /* OpenCV image: raw[y][x], zeroed (cv2.py:53) */
    memset(raw, 0, sizeof(float) * (size_t)W * H * 3);
    for (int64_t k = 0; k < nk; ++k) {
        const int32_t *fv = faces + 3 * order[k].f;
        int32_t pts[6];
        for (int j = 0; j < 3; ++j) project_px(sv[2 * fv[j]], sv[2 * fv[j] + 1], s, c, scale, W, H, &pts[2 * j], &pts[2 * j + 1]);
        float col[3];
        uint32_t packed;
        quantise_color(attrs + 3 * fv[0], col, &packed);
        if (rec_tris && k < rec_cap) {
            memcpy(rec_tris + 6 * k, pts, sizeof(pts));
            for (int ch = 0; ch < 3; ++ch) rec_cols[3 * k + ch] = (uint8_t)col[ch];
        }
        orc_fill_convex_poly(raw, W, H, pts, 3, col);
    }
    if (image)                                                        /* image.transpose(-2,-3) cv2.py:61; lh flips cancel :63-69 */
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x)
            memcpy(image + ((int64_t)x * H + y) * 3, raw + ((int64_t)y * W + x) * 3, 3 * sizeof(float));
    return nk;
}

ORC_API int64_t orc_render_rgb_mesh_one(const float *verts, const float *attrs, const int32_t *faces, int64_t V, int64_t F,
                                        float cx, float cy, float s, float c, float scale, int W, int H,
                                        float *image, int32_t *rec_tris, uint8_t *rec_cols, int64_t rec_cap) {
    render_scratch rs;
    render_scratch_init(&rs, V, F, W, H);
    int64_t nk = render_rgb_mesh_one(&rs, verts, attrs, faces, V, F, cx, cy, s, c, scale, W, H, image, rec_tris, rec_cols, rec_cap);
    render_scratch_free(&rs);
    return nk;
}

/* Simulator.render_egocentric, simulator.py:920-1033 + mesh.generate mesh.py:1053-1157 + render_frame,
 * restated with the reference's dataflow: for every camera the static map is concatenated with the
 * transformed actor mesh, then trimmed / ordered / projected / filled.
 *   state   B x N x 4, agent_sc B x N x 2 ([sin,cos] of psi, as torch computed them), tmpl B x N x 7 x 2
 *   actor_z / actor_rgb: B x N x 2 x {1,3}: (body, direction) z level and colour in [0,1]
 *   mask    B x Nc x N  uint8 (present & rendering mask)
 *   cam_xy, cam_sc  B x Nc x 2
 *   static mesh shared by all scenes: sverts Vs x 3 (x,y,z), sattrs Vs x 3, sfaces Fs x 3
 *   out     B x Nc x 3 x H x W  float32 (CHW as render_frame returns it, base.py:202-203)
 */
ORC_API void orc_render_scenes(const float *state, const float *agent_sc, const float *tmpl, const float *actor_z,
                               const float *actor_rgb, const uint8_t *mask, const float *cam_xy, const float *cam_sc,
                               const float *sverts, const float *sattrs, const int32_t *sfaces, int64_t Vs, int64_t Fs,
                               int64_t B, int64_t Nc, int64_t N, float scale, int W, int H, float *out,
                               int32_t *rec_tris, uint8_t *rec_cols, int64_t rec_cap, int64_t *rec_n) {
    const int64_t V = Vs + 7 * N, F = Fs + 3 * N;
    /* The reference concatenates the background with the actors per camera (background_mesh.expand(Nc) + concat, mesh.py:1147-1156).
     * Here every THREAD holds one copy of the concatenated arrays and rewrites only the actor part per image: same arithmetic per
     * image, without a malloc + 1.5 MB copy per image that made the multi-threaded timing of this port memory-bound. */
    #pragma omp parallel
    {
    float *verts = (float *)malloc(sizeof(float) * 3 * V);
    float *attrs = (float *)malloc(sizeof(float) * 3 * V);
    int32_t *faces = (int32_t *)malloc(sizeof(int32_t) * 3 * F);
    float *hwc = (float *)malloc(sizeof(float) * 3 * W * H);
    memcpy(verts, sverts, sizeof(float) * 3 * Vs);
    memcpy(attrs, sattrs, sizeof(float) * 3 * Vs);
    memcpy(faces, sfaces, sizeof(int32_t) * 3 * Fs);
    render_scratch rs;
    render_scratch_init(&rs, V, F, W, H);
    #pragma omp for schedule(dynamic, 1)
    for (int64_t img = 0; img < B * Nc; ++img) {
        int64_t b = img / Nc;
        for (int64_t a = 0; a < N; ++a) {
            const float *st = state + (b * N + a) * 4;
            float s = agent_sc[(b * N + a) * 2], c = agent_sc[(b * N + a) * 2 + 1];
            for (int k = 0; k < 7; ++k) {
                float *vo = verts + 3 * (Vs + 7 * a + k);
                const float *tp = tmpl + ((b * N + a) * 7 + k) * 2;
                transform_pt(tp[0], tp[1], s, c, st[0], st[1], &vo[0], &vo[1]);
                int part = k >= 4;
                vo[2] = actor_z[(b * N + a) * 2 + part];
                memcpy(attrs + 3 * (Vs + 7 * a + k), actor_rgb + ((b * N + a) * 2 + part) * 3, 3 * sizeof(float));
            }
            static const int tf[3][3] = { { 0, 1, 3 }, { 1, 3, 2 }, { 4, 5, 6 } };
            int on = mask[img * N + a] != 0;
            for (int f = 0; f < 3; ++f) for (int j = 0; j < 3; ++j)
                /* masked agents: faces * 0, then + Vs in concat -> alias the first actor vertex (mesh.py:1083-1089) */
                faces[3 * (Fs + 3 * a + f) + j] = (int32_t)(Vs + (on ? 7 * a + tf[f][j] : 0));
        }
        int64_t n = render_rgb_mesh_one(&rs, verts, attrs, faces, V, F, cam_xy[2 * img], cam_xy[2 * img + 1],
                                            cam_sc[2 * img], cam_sc[2 * img + 1], scale, W, H, out ? hwc : NULL,
                                            rec_tris ? rec_tris + img * rec_cap * 6 : NULL,
                                            rec_cols ? rec_cols + img * rec_cap * 3 : NULL, rec_cap);
        if (rec_n) rec_n[img] = n;
        if (out) {
            float *o = out + img * 3 * (int64_t)W * H;                /* permute(0,3,1,2) base.py:203 */
            for (int64_t p = 0; p < (int64_t)W * H; ++p) for (int ch = 0; ch < 3; ++ch) o[ch * (int64_t)W * H + p] = hwc[3 * p + ch];
        }
    }
    free(verts); free(attrs); free(faces); free(hwc);
    render_scratch_free(&rs);
    }
}

/* ---- observation model -------------------------------------------------------------------------------------------- */

/* utils.line_circle_intersection :139-187: does the segment p1-p2 touch the disc (centre c, radius r)?  One rounding per operation,
 * sums over the last dimension (2 elements) as x + y. */
static int line_circle(float p1x, float p1y, float p2x, float p2y, float cx, float cy, float r) {
    float dx = p2x - p1x, dy = p2y - p1y;
    float fx = p1x - cx, fy = p1y - cy;
    float a = dx * dx + dy * dy;
    float b = 2.0f * (fx * dx + fy * dy);
    float c = (fx * fx + fy * fy) - (r * r);
    float disc = b * b - (4.0f * a) * c;
    int has = disc >= 0.0f;
    float sq = sqrtf(disc < 0.0f ? 0.0f : disc);                 /* clamp(min=0) keeps NaN */
    if (disc != disc) sq = disc;
    float a_safe = (fabsf(a) < (float)1e-8) ? (float)1e-8 : a;
    float t1 = (-b - sq) / (2.0f * a_safe), t2 = (-b + sq) / (2.0f * a_safe);
    float tmin = t1 < t2 ? t1 : t2, tmax = t1 > t2 ? t1 : t2;
    if (t1 != t1 || t2 != t2) { tmin = tmax = t1 + t2; }        /* torch.min / max propagate NaN */
    return has && (tmin <= 1.0f) && (tmax >= 0.0f);
}

/* StandardSensingObservationNoise.get_noisy_present_mask, observation_noise.py:89-132.
 * state B x E x 4 (exposed agents first, then NPCs), size B x E x 2, present B x E; out B x A x E:
 * out[b,a,e] = present[b,e] and no entity o (o != e, o != a) whose disc of radius width/2 touches the sight line ego_a -> e. */
ORC_API void orc_occlusion_mask(const float *state, const float *size, const uint8_t *present, uint8_t *out, int64_t B, int64_t A, int64_t E) {
    for (int64_t b = 0; b < B; ++b)
        for (int64_t a = 0; a < A; ++a)
            for (int64_t e = 0; e < E; ++e) {
                const float *ego = state + (b * E + a) * 4, *tg = state + (b * E + e) * 4;
                int occluded = 0;
                for (int64_t o = 0; o < E && !occluded; ++o) {
                    if (o == e || o == a) continue;
                    const float *oc = state + (b * E + o) * 4;
                    float r = size[(b * E + o) * 2 + 1] / 2.0f;
                    occluded = line_circle(ego[0], ego[1], tg[0], tg[1], oc[0], oc[1], r);
                }
                out[(b * A + a) * E + e] = (uint8_t)(present[b * E + e] && !occluded);
            }
}

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORC_API void orc_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
