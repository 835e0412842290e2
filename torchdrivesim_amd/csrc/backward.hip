// Backward kernels of K2a (collision) and K2b (offroad) -- BASELINE config 5.
// They differentiate exactly what torch autograd differentiates in the reference:
//   collision  simulator.py:1064-1109 -> iou_differentiable_fast (_iou_utils.py:344-367; sort_indices is under no_grad, the
//              gradient flows through the gathered vertices :242-246) or collision_detection_with_discs (infractions.py:503-545)
//   offroad    infractions.py:86-229 (pure torch path): the arg-min face / edge, clamp, threshold
// The forward quantities are recomputed (float, same formulas as the forward kernels); results are compared with the
// reference's autograd gradients (fixture G7) in tests/test_gpu_backward.py.
#include "tds_common.h"

using tds::GridEntry;
using tds::MapView;

namespace {

constexpr int GBLOCK = 256;
constexpr int SCENE_BWD_MAX_PAIRS = 16384;          // pairs (exposed agents x boxes) of a scene that collision_scene_bwd_kernel takes: list entries are uint16

struct Box { float x, y, l, w, s, c; };
struct BoxGrad { float x, y, l, w, s, c; };

__device__ __forceinline__ float scrub(float v) {
    if (v != v) return 0.0f;
    if (__builtin_isinf(v)) return v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    return v;
}

__device__ __forceinline__ Box load_box(const float *boxes, const float *sc, int64_t idx) {
    Box b;
    const float *p = boxes + idx * 5;
    b.x = scrub(p[0]); b.y = scrub(p[1]); b.l = scrub(p[2]); b.w = scrub(p[3]);
    b.s = sc[idx * 2]; b.c = sc[idx * 2 + 1];
    if (p[4] != p[4]) { b.s = 0.0f; b.c = 1.0f; }
    return b;
}

__device__ __forceinline__ void corners_of(const Box &b, float *cx, float *cy) {
    const float sx[4] = {0.5f, -0.5f, -0.5f, 0.5f}, sy[4] = {0.5f, 0.5f, -0.5f, -0.5f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x4 = sx[k] * b.l, y4 = sy[k] * b.w;
        cx[k] = (x4 * b.c + y4 * (-b.s)) + b.x;
        cy[k] = (x4 * b.s + y4 * b.c) + b.y;
    }
}

// d corners -> d box   (corner_k = (sx l c - sy w s + x, sx l s + sy w c + y))
__device__ __forceinline__ void corners_bwd(const Box &b, const float *gx, const float *gy, BoxGrad &g) {
    const float sx[4] = {0.5f, -0.5f, -0.5f, 0.5f}, sy[4] = {0.5f, 0.5f, -0.5f, -0.5f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        g.x += gx[k]; g.y += gy[k];
        g.l += gx[k] * sx[k] * b.c + gy[k] * sx[k] * b.s;
        g.w += -gx[k] * sy[k] * b.s + gy[k] * sy[k] * b.c;
        g.c += gx[k] * sx[k] * b.l + gy[k] * sy[k] * b.w;
        g.s += -gx[k] * sy[k] * b.w + gy[k] * sx[k] * b.l;
    }
}

__device__ __forceinline__ unsigned corners_in(const float *px, const float *py, const float *qx, const float *qy) {
    float ax = qx[0], ay = qy[0];
    float abx = qx[1] - ax, aby = qy[1] - ay, adx = qx[3] - ax, ady = qy[3] - ay;
    float nab = abx * abx + aby * aby, nad = adx * adx + ady * ady;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float amx = px[k] - ax, amy = py[k] - ay;
        float r1 = rintf(((abx * amx + aby * amy) / nab) * 1000000.0f), r2 = rintf(((adx * amx + ady * amy) / nad) * 1000000.0f);
        bool in = (r1 >= 0.0f) && (r1 <= 1000000.0f) && (r2 >= 0.0f) && (r2 <= 1000000.0f);
        m |= (in ? 1u : 0u) << k;
    }
    return m;
}

// IoU forward + backward for one pair.  Returns iou; when `gout != 0` adds gout * d iou / d(box1), d(box2) to g1, g2.
// Candidates live in per-thread arrays (scratch): this kernel runs once per training step on B*A*N pairs, most of which
// exit on the bounding-circle test.
__device__ float iou_pair_bwd(const Box &b1, const Box &b2, float gout, BoxGrad &g1, BoxGrad &g2, bool want_grad) {
    float a1 = b1.l * b1.w, a2 = b2.l * b2.w;
    float ddx = b1.x - b2.x, ddy = b1.y - b2.y;
    float r1 = 0.5f * sqrtf(b1.l * b1.l + b1.w * b1.w), r2 = 0.5f * sqrtf(b2.l * b2.l + b2.w * b2.w);
    float reach = r1 + r2 + 0.05f + 1e-5f * fmaxf(fmaxf(fabsf(b1.x), fabsf(b1.y)), fmaxf(fabsf(b2.x), fabsf(b2.y)));
    float inter = 0.0f;
    float c1x[4], c1y[4], c2x[4], c2y[4];
    float vx[24], vy[24], tt[16];
    unsigned mask = 0;
    int order[9];
    int n = 0;
    float total = 0.0f;
    bool overlap = !(ddx * ddx + ddy * ddy > reach * reach);
    if (overlap) {
        corners_of(b1, c1x, c1y);
        corners_of(b2, c2x, c2y);
        mask = corners_in(c1x, c1y, c2x, c2y) | (corners_in(c2x, c2y, c1x, c1y) << 4);
        for (int k = 0; k < 4; ++k) { vx[k] = c1x[k]; vy[k] = c1y[k]; vx[4 + k] = c2x[k]; vy[4 + k] = c2y[k]; }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float x1 = c1x[i], y1 = c1y[i], x2 = c1x[(i + 1) & 3], y2 = c1y[(i + 1) & 3];
                float x3 = c2x[j], y3 = c2y[j], x4 = c2x[(j + 1) & 3], y4 = c2y[(j + 1) & 3];
                float num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
                float den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
                float nden_u = -((x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3));
                float an = fabsf(num);
                bool ok = !(an < (float)1e-4) && (an == an);
                bool mt = ok && ((den_t > 0.0f) == (num > 0.0f)) && (den_t != 0.0f) && (fabsf(den_t) < an);
                bool mu = ok && ((nden_u > 0.0f) == (num > 0.0f)) && (nden_u != 0.0f) && (fabsf(nden_u) < an);
                float t = den_t / (num + (float)1e-8);
                tt[i * 4 + j] = t;
                bool mk = mt && mu;
                vx[8 + i * 4 + j] = mk ? x1 + t * (x2 - x1) : 0.0f;
                vy[8 + i * 4 + j] = mk ? y1 + t * (y2 - y1) : 0.0f;
                if (mk) mask |= 1u << (8 + i * 4 + j);
            }
        n = __popc(mask);
        if (n >= 3 && n <= 8) {
            float ax4[4] = {0, 0, 0, 0}, ay4[4] = {0, 0, 0, 0};
            for (int k = 0; k < 24; ++k)
                if ((mask >> k) & 1) { ax4[k & 3] += vx[k]; ay4[k & 3] += vy[k]; }
            float cx = (((ax4[0] + ax4[1]) + ax4[2]) + ax4[3]) / (float)n, cy = (((ay4[0] + ay4[1]) + ay4[2]) + ay4[3]) / (float)n;
            float key[8];
            int m = 0;
            for (int k = 0; k < 24; ++k)
                if ((mask >> k) & 1) {
                    float dx = vx[k] - cx, dy = vy[k] - cy;
                    float q = dx / sqrtf(dx * dx + dy * dy);
                    float kk = (dy > 0.0f) ? -q : 2.0f + q;
                    kk = (kk == kk) ? kk : 3.5f;
                    int p = m++;
                    while (p > 0 && key[p - 1] > kk) { key[p] = key[p - 1]; order[p] = order[p - 1]; --p; }
                    key[p] = kk; order[p] = k;
                }
            order[n] = order[0];
            for (int k = 0; k < n; ++k) total = total + (vx[order[k]] * vy[order[k + 1]] - vy[order[k]] * vx[order[k + 1]]);
            inter = fabsf(total) / 2.0f;
        }
        // n > 8 (never observed): forward falls back to the general path; the gradient is dropped here
    }
    float u = a1 + a2 - inter;
    float iou = inter / u;
    if (!want_grad || !(gout != 0.0f) || !(iou == iou)) return iou;
    // ---- backward
    float g_inter = gout * (u + inter) / (u * u);          // d(inter/u)/d inter with u = a1 + a2 - inter
    float g_area = -gout * inter / (u * u);
    g1.l += g_area * b1.w; g1.w += g_area * b1.l;
    g2.l += g_area * b2.w; g2.w += g_area * b2.l;
    if (!(inter > 0.0f) || n < 3 || n > 8) return iou;
    float g_total = g_inter * 0.5f * (total >= 0.0f ? 1.0f : -1.0f);
    float g1x[4] = {0, 0, 0, 0}, g1y[4] = {0, 0, 0, 0}, g2x[4] = {0, 0, 0, 0}, g2y[4] = {0, 0, 0, 0};
    for (int k = 0; k < n; ++k) {
        int prev = order[(k + n - 1) % n], cur = order[k], next = order[k + 1];
        float gvx = g_total * (vy[next] - vy[prev]), gvy = g_total * (vx[prev] - vx[next]);
        if (cur < 4) { g1x[cur] += gvx; g1y[cur] += gvy; }
        else if (cur < 8) { g2x[cur - 4] += gvx; g2y[cur - 4] += gvy; }
        else {
            int i = (cur - 8) >> 2, j = (cur - 8) & 3, i2 = (i + 1) & 3, j2 = (j + 1) & 3;
            float x1 = c1x[i], y1 = c1y[i], x2 = c1x[i2], y2 = c1y[i2], x3 = c2x[j], y3 = c2y[j], x4 = c2x[j2], y4 = c2y[j2];
            float t = tt[cur - 8];
            float num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4) + (float)1e-8;
            float den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
            // P = A + t (B - A)
            g1x[i] += gvx * (1.0f - t); g1y[i] += gvy * (1.0f - t);
            g1x[i2] += gvx * t; g1y[i2] += gvy * t;
            float gt = gvx * (x2 - x1) + gvy * (y2 - y1);
            float gden = gt / num, gnum = -gt * den_t / (num * num);
            // num = (x1-x2)(y3-y4) - (y1-y2)(x3-x4)
            g1x[i] += gnum * (y3 - y4); g1x[i2] -= gnum * (y3 - y4);
            g1y[i] -= gnum * (x3 - x4); g1y[i2] += gnum * (x3 - x4);
            g2y[j] += gnum * (x1 - x2); g2y[j2] -= gnum * (x1 - x2);
            g2x[j] -= gnum * (y1 - y2); g2x[j2] += gnum * (y1 - y2);
            // den_t = (x1-x3)(y3-y4) - (y1-y3)(x3-x4)
            g1x[i] += gden * (y3 - y4);
            g1y[i] -= gden * (x3 - x4);
            g2x[j] += gden * (-(y3 - y4) - (y1 - y3));
            g2y[j] += gden * ((x1 - x3) + (x3 - x4));
            g2y[j2] -= gden * (x1 - x3);
            g2x[j2] += gden * (y1 - y3);
        }
    }
    corners_bwd(b1, g1x, g1y, g1);
    corners_bwd(b2, g2x, g2y, g2);
    return iou;
}

// discs forward + backward (infractions.py:378-426,503-545): d = min over the 25 centre distances, o = relu(1 - d / (r1 + r2))
__device__ float discs_pair_bwd(const Box &b1, const Box &b2, float gout, BoxGrad &g1, BoxGrad &g2, bool want_grad) {
    float ra = fminf(b1.l, b1.w) / 2.0f, rb = fminf(b2.l, b2.w) / 2.0f;
    float ha = fmaxf(b1.l, b1.w) / 2.0f - ra, hb = fmaxf(b2.l, b2.w) / 2.0f - rb;
    float d = __builtin_inff();
    int bi = 0, bj = 0;
    float bex = 0.0f, bey = 0.0f;
    for (int i = -2; i <= 2; ++i) {
        float da = ((float)i * ha) / 2.0f;
        float ax = da * b1.c + b1.x, ay = da * b1.s + b1.y;
        for (int j = -2; j <= 2; ++j) {
            float db = ((float)j * hb) / 2.0f;
            float bx = db * b2.c + b2.x, by = db * b2.s + b2.y;
            float ex = ax - bx, ey = ay - by;
            float dd = sqrtf(__fmaf_rn(ey, ey, ex * ex));
            if (dd < d) { d = dd; bi = i; bj = j; bex = ex; bey = ey; }
        }
    }
    float rs = ra + rb;
    float l = 1.0f - d / rs;
    float o = (l != l) ? l : fmaxf(l, 0.0f);
    if (!want_grad || !(gout != 0.0f) || !(l > 0.0f)) return o;
    float gd = -gout / rs, grs = gout * d / (rs * rs);
    // radii: r = min(l, w) / 2
    auto add_min = [](const Box &b, BoxGrad &g, float v) { if (b.l <= b.w) g.l += v; else g.w += v; };
    auto add_max = [](const Box &b, BoxGrad &g, float v) { if (b.l >= b.w) g.l += v; else g.w += v; };
    add_min(b1, g1, 0.5f * grs);
    add_min(b2, g2, 0.5f * grs);
    if (d > 0.0f) {
        float ux = bex / d, uy = bey / d;                            // d dist / d centre_a ; centre_b gets the negative
        float gax = gd * ux, gay = gd * uy;
        float da = ((float)bi * ha) / 2.0f, db = ((float)bj * hb) / 2.0f;
        g1.x += gax; g1.y += gay; g1.c += gax * da; g1.s += gay * da;
        g2.x -= gax; g2.y -= gay; g2.c -= gax * db; g2.s -= gay * db;
        // da = i * (max/2 - min/2) / 2
        float gda = gax * b1.c + gay * b1.s, gdb = -(gax * b2.c + gay * b2.s);
        add_max(b1, g1, gda * (float)bi * 0.25f); add_min(b1, g1, -gda * (float)bi * 0.25f);
        add_max(b2, g2, gdb * (float)bj * 0.25f); add_min(b2, g2, -gdb * (float)bj * 0.25f);
    }
    return o;
}

// pairs whose bounding circles are disjoint have overlap exactly 0 and no gradient under both metrics (the test of the forward's
// collision_scene_iou_kernel; the discs of a box lie inside its circumscribed circle)
__device__ __forceinline__ bool circles_touch(const Box &b1, const Box &b2) {
    const float dx = b1.x - b2.x, dy = b1.y - b2.y;
    const float r1 = 0.5f * sqrtf(b1.l * b1.l + b1.w * b1.w), r2 = 0.5f * sqrtf(b2.l * b2.l + b2.w * b2.w);
    const float reach = r1 + r2 + 0.05f + 1e-5f * fmaxf(fmaxf(fabsf(b1.x), fabsf(b1.y)), fmaxf(fabsf(b2.x), fabsf(b2.y)));
    return !(dx * dx + dy * dy > reach * reach);
}

// one WAVEFRONT per (scene, exposed agent i), lanes = partners j (the first version ran one thread per row over all N partners,
// twice: 1.25 ms at B = 256 x 64 x 64 against 0.02 ms for the forward).  Only lanes whose pair passes the bounding-circle test run the
// pair function.  grad_boxes (B,N,5) and grad_sc (B,N,2) are accumulated with atomics (they were zeroed by the host wrapper).
template <int METRIC>
__global__ void __launch_bounds__(GBLOCK) collision_bwd_kernel(const float *__restrict__ boxes, const float *__restrict__ sc,
                                                               const uint8_t *__restrict__ present, const float *__restrict__ gout,
                                                               float *__restrict__ gboxes, float *__restrict__ gsc, int64_t B, int A, int N) {
    const int lane = threadIdx.x & 63;
    int64_t t = ((int64_t)blockIdx.x * GBLOCK + threadIdx.x) >> 6;
    if (t >= B * A) return;
    int64_t b = t / A;
    int i = (int)(t - b * A);
    float go = gout[t];
    if (!(go != 0.0f)) return;
    Box bi = load_box(boxes, sc, b * N + i);
    // pass 1: arg-max of the masked overlaps (the reference subtracts overlap.max, simulator.py:1108); first index on ties
    BoxGrad dummy1 = {0, 0, 0, 0, 0, 0}, dummy2 = {0, 0, 0, 0, 0, 0};
    float mx = -__builtin_inff();
    int arg = 0x7fffffff;
    for (int j = lane; j < N; j += 64) {
        Box bj = load_box(boxes, sc, b * N + j);
        float o = 0.0f;
        if (circles_touch(bi, bj))
            o = METRIC == TDS_METRIC_IOU ? iou_pair_bwd(bi, bj, 0.0f, dummy1, dummy2, false) : discs_pair_bwd(bi, bj, 0.0f, dummy1, dummy2, false);
        o = scrub(o) * (present[b * N + j] ? 1.0f : 0.0f);
        if (o > mx) { mx = o; arg = j; }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        float om = __shfl_xor(mx, d);
        int oa = __shfl_xor(arg, d);
        if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    // pass 2: d(sum - max) / d o_ij = present_j * (1 - [j == arg])
    BoxGrad gi = {0, 0, 0, 0, 0, 0};
    for (int j = lane; j < N; j += 64) {
        if (j == arg || !present[b * N + j]) continue;
        Box bj = load_box(boxes, sc, b * N + j);
        if (!circles_touch(bi, bj)) continue;
        BoxGrad gj = {0, 0, 0, 0, 0, 0};
        BoxGrad gl = {0, 0, 0, 0, 0, 0};
        float o = METRIC == TDS_METRIC_IOU ? iou_pair_bwd(bi, bj, go, gl, gj, true) : discs_pair_bwd(bi, bj, go, gl, gj, true);
        if (!(o == o)) continue;                                     // nan_to_num: no gradient through a scrubbed NaN
        gi.x += gl.x; gi.y += gl.y; gi.l += gl.l; gi.w += gl.w; gi.s += gl.s; gi.c += gl.c;
        float *gb = gboxes + (b * N + j) * 5, *gs = gsc + (b * N + j) * 2;
        if (gj.x != 0.0f) atomicAdd(gb + 0, gj.x);
        if (gj.y != 0.0f) atomicAdd(gb + 1, gj.y);
        if (gj.l != 0.0f) atomicAdd(gb + 2, gj.l);
        if (gj.w != 0.0f) atomicAdd(gb + 3, gj.w);
        if (gj.s != 0.0f) atomicAdd(gs + 0, gj.s);
        if (gj.c != 0.0f) atomicAdd(gs + 1, gj.c);
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        gi.x += __shfl_xor(gi.x, d); gi.y += __shfl_xor(gi.y, d); gi.l += __shfl_xor(gi.l, d);
        gi.w += __shfl_xor(gi.w, d); gi.s += __shfl_xor(gi.s, d); gi.c += __shfl_xor(gi.c, d);
    }
    if (lane == 0) {
        float *gb = gboxes + (b * N + i) * 5, *gs = gsc + (b * N + i) * 2;
        atomicAdd(gb + 0, gi.x); atomicAdd(gb + 1, gi.y); atomicAdd(gb + 2, gi.l); atomicAdd(gb + 3, gi.w);
        atomicAdd(gs + 0, gi.s); atomicAdd(gs + 1, gi.c);
    }
}

// Whole scene per workgroup (the backward of collision_scene_iou_kernel): with one wavefront per row the pair function ran twice per
// row for the handful of lanes whose partner is near (0.22 ms at B = 256 x 64 x 64 against 0.026 ms for the forward).  Here the near
// pairs of the scene's rows that carry a gradient are gathered into an LDS list first and evaluated on full waves: overlaps ->
// arg-max per row (first index on ties) -> gradients of every near pair but the row's maximum, summed per box and stored once (the
// scene's boxes belong to this workgroup alone: no global atomics, no zeroing beforehand).  Same pair functions, same rules as
// collision_bwd_kernel.
// DETERMINISTIC since round 6 (it used ds_add_f32 in arrival order: an ulp of a gradient moved from run to run): the near list is built
// in pair order (ballot + prefix over the waves, not an atomic counter), every listed pair leaves its two box gradients in a slot of an
// LDS table, and one thread per box adds its box's contributions in a fixed order -- its row's pairs by partner, then the pairs of the
// other rows that have it as partner, by row -- found through a pair -> slot map.  Lists longer than the table are worked off in chunks,
// in list order.
// LDS: boxes N x 6, sums N x 6, gout / arg-max A, overlaps A x N (float), the pair list A x N (uint16), the slot map A x N (uint16),
// the gradient table `cap` x 12.
template <int METRIC>
__global__ void __launch_bounds__(GBLOCK, 2) collision_scene_bwd_kernel(const float *__restrict__ boxes, const float *__restrict__ sc,
                                                                     const uint8_t *__restrict__ present, const float *__restrict__ gout,
                                                                     float *__restrict__ gboxes, float *__restrict__ gsc, int A, int N, int cap) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int wave_cnt[2][GBLOCK / 64];               // (two sets, used in turn: one barrier per round of the list build)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    float *bx = smem;                                       // N x 6
    float *acc = bx + N * 6;                                // N x 6
    float *go = acc + N * 6;                                // A
    int *arg = (int *)(go + A);                             // A
    float *O = (float *)(arg + A);                          // A x N
    uint16_t *list = (uint16_t *)(O + A * N);               // A x N at most (A x N <= SCENE_BWD_MAX_PAIRS)
    uint16_t *slot = list + A * N;                          // A x N: pair -> place in the list (0xffff: not listed)
    float *pg = (float *)(slot + A * N);                    // cap x 12: the box gradients of the listed pairs of the current chunk (list + slot: 4 A N bytes)
    for (int j = tid; j < N; j += GBLOCK) {
        const Box q = load_box(boxes, sc, b * N + j);
        bx[6 * j] = q.x; bx[6 * j + 1] = q.y; bx[6 * j + 2] = q.l; bx[6 * j + 3] = q.w; bx[6 * j + 4] = q.s; bx[6 * j + 5] = q.c;
    }
    for (int k = tid; k < N * 6; k += GBLOCK) acc[k] = 0.0f;
    for (int i = tid; i < A; i += GBLOCK) go[i] = gout[b * A + i];
    __syncthreads();
    auto box_at = [&](int j) { Box q; q.x = bx[6 * j]; q.y = bx[6 * j + 1]; q.l = bx[6 * j + 2]; q.w = bx[6 * j + 3]; q.s = bx[6 * j + 4]; q.c = bx[6 * j + 5]; return q; };
    // the near pairs of the rows that carry a gradient, IN PAIR ORDER
    int total = 0;                                          // (the same in every thread)
    for (int p0 = 0, round = 0; p0 < A * N; p0 += GBLOCK, round ^= 1) {
        const int p = p0 + tid;
        bool near = false;
        if (p < A * N) {
            const int i = p / N, j = p - i * N;
            O[p] = 0.0f;
            near = go[i] != 0.0f && circles_touch(box_at(i), box_at(j));
        }
        const unsigned long long bm = __ballot(near);
        if (lane == 0) wave_cnt[round][wave] = __popcll(bm);
        __syncthreads();                                    // (the other set is not written again before every wave has passed the next barrier)
        int before = total, all = 0;
#pragma unroll
        for (int w = 0; w < GBLOCK / 64; ++w) { before += w < wave ? wave_cnt[round][w] : 0; all += wave_cnt[round][w]; }
        const int q = before + __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
        if (p < A * N) slot[p] = near ? (uint16_t)q : (uint16_t)0xffffu;
        if (near) list[q] = (uint16_t)p;
        total += all;
    }
    __syncthreads();
    BoxGrad dummy1 = {0, 0, 0, 0, 0, 0}, dummy2 = {0, 0, 0, 0, 0, 0};
    for (int q = tid; q < total; q += GBLOCK) {
        const int p = (int)list[q], i = p / N, j = p - i * N;
        const Box b1 = box_at(i), b2 = box_at(j);
        const float o = METRIC == TDS_METRIC_IOU ? iou_pair_bwd(b1, b2, 0.0f, dummy1, dummy2, false) : discs_pair_bwd(b1, b2, 0.0f, dummy1, dummy2, false);
        O[p] = scrub(o) * (present[b * N + j] ? 1.0f : 0.0f);
    }
    __syncthreads();
    // arg-max of the masked overlaps per row (the reference subtracts overlap.max, simulator.py:1108); first index on ties
    for (int i = wave; i < A; i += GBLOCK / 64) {
        float mx = -__builtin_inff();
        int am = 0x7fffffff;
        for (int j = lane; j < N; j += 64) {
            const float o = O[i * N + j];
            if (o > mx) { mx = o; am = j; }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float om = __shfl_xor(mx, d);
            const int oa = __shfl_xor(am, d);
            if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
        }
        if (lane == 0) arg[i] = am;
    }
    __syncthreads();
    // d(sum - max) / d o_ij = present_j * (1 - [j == arg]); chunk by chunk of the list: the pairs' gradients into the table, then every box
    // collects its own in a fixed order
    for (int q0 = 0; q0 < total; q0 += cap) {
        const int nq = min(cap, total - q0);
        for (int q = tid; q < nq; q += GBLOCK) {
            const int p = (int)list[q0 + q], i = p / N, j = p - i * N;
            BoxGrad gi = {0, 0, 0, 0, 0, 0}, gj = {0, 0, 0, 0, 0, 0};
            if (j != arg[i] && present[b * N + j]) {
                const Box b1 = box_at(i), b2 = box_at(j);
                const float o = METRIC == TDS_METRIC_IOU ? iou_pair_bwd(b1, b2, go[i], gi, gj, true) : discs_pair_bwd(b1, b2, go[i], gi, gj, true);
                if (!(o == o)) { gi = dummy1; gj = dummy1; }             // nan_to_num: no gradient through a scrubbed NaN
            }
            float *t = pg + 12 * q;
            t[0] = gi.x; t[1] = gi.y; t[2] = gi.l; t[3] = gi.w; t[4] = gi.s; t[5] = gi.c;
            t[6] = gj.x; t[7] = gj.y; t[8] = gj.l; t[9] = gj.w; t[10] = gj.s; t[11] = gj.c;
        }
        __syncthreads();
        // four threads per box, each a fixed quarter of the box's candidates (its row's pairs by partner: two halves; the other rows' pairs that
        // have it as partner, by row: two halves), combined in a fixed tree: (a + b) + (c + d), then onto the box's sum so far
        for (int kk = tid; kk < 4 * N; kk += GBLOCK) {
            const int k = kk >> 2, part = kk & 3;
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f, s4 = 0.0f, s5 = 0.0f;
            auto take = [&](int p, int off) {
                const int q = (int)slot[p] - q0;
                if ((unsigned)q < (unsigned)nq) {               // listed (0xffff is beyond every chunk: total <= 16 384) and in this chunk
                    const float *t = pg + 12 * q + off;
                    s0 += t[0]; s1 += t[1]; s2 += t[2]; s3 += t[3]; s4 += t[4]; s5 += t[5];
                }
            };
            if (part < 2) {
                if (k < A) { const int h = (N + 1) >> 1; for (int j = part * h; j < min(N, (part + 1) * h); ++j) take(k * N + j, 0); }      // the first box of the pair (k, j)
            } else {
                const int h = (A + 1) >> 1;
                for (int i = (part - 2) * h; i < min(A, (part - 1) * h); ++i) take(i * N + k, 6);                                         // the second box of the pair (i, k)
            }
            // lanes 4 k .. 4 k + 3 of one wave hold the box's four partial sums
            auto tree = [&](float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); return v; };
            s0 = tree(s0); s1 = tree(s1); s2 = tree(s2); s3 = tree(s3); s4 = tree(s4); s5 = tree(s5);
            if (part == 0) { acc[6 * k] += s0; acc[6 * k + 1] += s1; acc[6 * k + 2] += s2; acc[6 * k + 3] += s3; acc[6 * k + 4] += s4; acc[6 * k + 5] += s5; }
        }
        __syncthreads();
    }
    for (int j = tid; j < N; j += GBLOCK) {                  // every element of both outputs (the heading's gradient flows through grad_sc)
        float *gb = gboxes + (b * N + j) * 5, *gs = gsc + (b * N + j) * 2;
        gb[0] = acc[6 * j]; gb[1] = acc[6 * j + 1]; gb[2] = acc[6 * j + 2]; gb[3] = acc[6 * j + 3]; gb[4] = 0.0f;
        gs[0] = acc[6 * j + 4]; gs[1] = acc[6 * j + 5];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// offroad backward
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) { return (ax * bx + ay * by) + 0.0f; }

// squared distance to a segment and its gradient w.r.t. the point (autograd of infractions.py:147-159)
__device__ __forceinline__ float seg_d2_grad(float px, float py, float ax, float ay, float bx, float by, float &gx, float &gy) {
    float ex = bx - ax, ey = by - ay;
    float l2 = dot2(ex, ey, ex, ey);
    float t = dot2(ex, ey, px - ax, py - ay) / (l2 + (float)1e-8);
    float tc = fminf(fmaxf(t, 0.0f), 1.0f);
    float qx = ax + tc * ex, qy = ay + tc * ey;
    float rx = px - qx, ry = py - qy;
    float d = dot2(rx, ry, rx, ry);
    gx = 2.0f * rx; gy = 2.0f * ry;
    if (t >= 0.0f && t <= 1.0f) {                                    // clamp passes the gradient on [0, 1]
        float s = 2.0f * (rx * ex + ry * ey) / (l2 + (float)1e-8);
        gx -= s * ex; gy -= s * ey;
    }
    if (l2 <= (float)1e-8) { d = dot2(px - bx, py - by, px - bx, py - by); gx = 2.0f * (px - bx); gy = 2.0f * (py - by); }
    return d;
}

__device__ __forceinline__ float tri_d2_grad(float px, float py, const GridEntry &e, float &gx, float &gy) {
    float cz = (e.x2 - e.x0) * (e.y1 - e.y0) - (e.y2 - e.y0) * (e.x1 - e.x0);
    float norm_normal = sqrtf(cz * cz);
    float p0x = e.x1 - e.x0, p0y = e.y1 - e.y0, p1x = e.x2 - e.x0, p1y = e.y2 - e.y0, p2x = px - e.x0, p2y = py - e.y0;
    float d00 = dot2(p0x, p0y, p0x, p0y), d01 = dot2(p0x, p0y, p1x, p1y), d11 = dot2(p1x, p1y, p1x, p1y);
    float d20 = dot2(p2x, p2y, p0x, p0y), d21 = dot2(p2x, p2y, p1x, p1y);
    float denom = d00 * d11 - d01 * d01 + (float)1e-8;
    float w1 = (d11 * d20 - d01 * d21) / denom, w2 = (d00 * d21 - d01 * d20) / denom, w0 = 1.0f - w1 - w2;
    bool inside = (0.0f <= w0) && (w0 <= 1.0f) && (0.0f <= w1) && (w1 <= 1.0f) && (0.0f <= w2) && (w2 <= 1.0f);
    float area = fabsf(p0x * p1y - p0y * p1x) / 2.0f;
    inside = inside && !(area < (float)5e-3) && (norm_normal > (float)1e-8);
    float g1x, g1y, g2x, g2y, g3x, g3y;
    float e01 = seg_d2_grad(px, py, e.x0, e.y0, e.x1, e.y1, g1x, g1y);
    float e02 = seg_d2_grad(px, py, e.x0, e.y0, e.x2, e.y2, g2x, g2y);
    float e12 = seg_d2_grad(px, py, e.x1, e.y1, e.x2, e.y2, g3x, g3y);
    float dist = e01; gx = g1x; gy = g1y;
    if (e02 < dist) { dist = e02; gx = g2x; gy = g2y; }
    if (e12 < dist) { dist = e12; gx = g3x; gy = g3y; }
    if (inside) { dist = 0.0f; gx = 0.0f; gy = 0.0f; }
    return dist;
}

// `stop`: below it the caller's gradient is zero (F.threshold), so the walk may end early (see nearest_face_d2 in map.hip)
__device__ float nearest_face_d2_grad(const MapView &m, float px, float py, float &gx, float &gy, float stop) {
    float best = __builtin_inff();
    gx = gy = 0.0f;
    if (m.nx <= 0 || !(px == px) || !(py == py) || __builtin_isinf(px) || __builtin_isinf(py)) return best;
    float fx = fminf(fmaxf((px - m.ox) * m.inv_cell, -1.0e6f), 1.0e6f), fy = fminf(fmaxf((py - m.oy) * m.inv_cell, -1.0e6f), 1.0e6f);
    int cx = (int)floorf(fx), cy = (int)floorf(fy);
    auto visit = [&](int x, int y) {
        float bx0 = m.ox + (float)x * m.cell, by0 = m.oy + (float)y * m.cell;
        float ddx = fmaxf(fmaxf(bx0 - px, px - (bx0 + m.cell)), 0.0f), ddy = fmaxf(fmaxf(by0 - py, py - (by0 + m.cell)), 0.0f);
        if ((ddx * ddx + ddy * ddy) * 0.998f - 1e-3f >= best) return;
        int s = m.cell_start[y * m.nx + x], e = m.cell_start[y * m.nx + x + 1];
        for (int i = s; i < e && best > stop; ++i) {
            GridEntry ge = m.entries[i];
            float tgx, tgy;
            float d = tri_d2_grad(px, py, ge, tgx, tgy);
            if (d < best) { best = d; gx = tgx; gy = tgy; }
        }
    };
    int k = max(max(0, max(-cx, cx - (m.nx - 1))), max(-cy, cy - (m.ny - 1)));
    for (;; ++k) {
        int y0 = max(cy - k, 0), y1 = min(cy + k, m.ny - 1), x0 = max(cx - k, 0), x1 = min(cx + k, m.nx - 1);
        for (int y = y0; y <= y1; ++y) {
            if (y == cy - k || y == cy + k) {
                for (int x = x0; x <= x1; ++x) visit(x, y);
            } else {
                if (cx - k >= 0 && cx - k < m.nx) visit(cx - k, y);
                if (k > 0 && cx + k >= 0 && cx + k < m.nx) visit(cx + k, y);
            }
        }
        if (best <= stop) break;
        float bound = (float)k * m.cell * 0.999f;
        if (best <= bound * bound) break;
        if (cx - k <= 0 && cx + k >= m.nx - 1 && cy - k <= 0 && cy + k >= m.ny - 1) break;
    }
    return best;
}

// the same arg-min by the ordered descent of the hierarchy over the faces (see nearest_face_d2_bvh in map.hip): the group's BL lanes hold the
// same point, weigh a child each at an inner node and a face each at a leaf; boxes AT the running minimum are still opened (a face in them may
// tie it, and the lowest face index among ties gives the gradient); the gradient is taken once, from the winning face
constexpr int BL = 8;             // lanes per corner (a power of two, <= 16: a wavefront holds whole agents)
using tds::BVH_STACK;             // (tds_common.h; tds_map_create refuses to attach a hierarchy the stack cannot hold)
__device__ float nearest_face_d2_grad_bvh(const tds::NearView &nv, float px, float py, float &gx, float &gy, float stop, int sub) {
    static_assert(BL == 8, "nearest_face_d2_grad_bvh: a lane per child of a node");
    __shared__ int2 stacks[GBLOCK / BL][BVH_STACK];
    int2 *st = stacks[threadIdx.x / BL];
    const float inf = __builtin_inff();
    const int shift = (int)(threadIdx.x & 63 & ~(BL - 1));                              // the group's first lane within the wavefront
    float best = inf;
    int best_f = 0x7fffffff;                 // among faces at the same distance the one of lowest index gives the gradient (torch.min's choice)
    gx = gy = 0.0f;
    auto box_lb = [&](float x0, float y0, float x1, float y1) {
        const float ex = fmaxf(fmaxf(x0 - px, px - x1), 0.0f), ey = fmaxf(fmaxf(y0 - py, py - y1), 0.0f);
        return (ex * ex + ey * ey) * 0.998f - 1e-3f;
    };
    int sp = 0, cur = 0;
    for (;;) {
        if (cur >= 0) {
            const float4 bx = nv.bvh[cur].box[sub];
            const int ch = nv.bvh[cur].child[sub];
            const float lb = box_lb(bx.x, bx.y, bx.z, bx.w);
            const bool open = lb <= best && lb < inf;
            const float key = open ? lb : inf;
            int above = 0;                                                             // open children that will lie above this lane's on the stack
#pragma unroll
            for (int k = 1; k < BL; ++k) {
                const float o = __shfl_xor(key, k);
                above += (o < key || (o == key && (sub ^ k) < sub)) ? 1 : 0;
            }
            const int nopen = __popc((unsigned)(__ballot(open) >> shift) & 0xffu);
            const int slot = sp + nopen - 1 - above;
            if (open && slot < BVH_STACK) st[slot] = make_int2(ch, __float_as_int(lb));
            sp = min(sp + nopen, BVH_STACK);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            const int code = -1 - cur, first = code >> 4, cnt = code & 15;
            float d = inf;
            int f = 0x7fffffff;
            if (sub < cnt) {
                const int fi = nv.bvh_idx[first + sub];
                const GridEntry ge = nv.faces[fi];
                const float fx0 = fminf(ge.x0, fminf(ge.x1, ge.x2)), fx1 = fmaxf(ge.x0, fmaxf(ge.x1, ge.x2));
                const float fy0 = fminf(ge.y0, fminf(ge.y1, ge.y2)), fy1 = fmaxf(ge.y0, fmaxf(ge.y1, ge.y2));
                if (!(box_lb(fx0, fy0, fx1, fy1) > best)) {                             // (a box AT the minimum may hold a face that ties it)
                    float tgx, tgy;
                    const float t = tri_d2_grad(px, py, ge, tgx, tgy);                // (the distance alone: the gradient is taken below)
                    if (t < inf) { d = t; f = fi; }                                   // a NaN distance never becomes the minimum
                }
            }
#pragma unroll
            for (int k = 1; k < BL; k <<= 1) {
                const float od = __shfl_xor(d, k);
                const int of = __shfl_xor(f, k);
                if (od < d || (od == d && of < f)) { d = od; f = of; }
            }
            if (d < best || (d == best && f < best_f)) { best = d; best_f = f; }
        }
        // the next subtree that can still hold a face as near
        float lb = inf;
        bool done = false;
        do {
            if (sp == 0 || best <= stop) { done = true; break; }
            --sp;
            cur = st[sp].x; lb = __int_as_float(st[sp].y);
        } while (lb > best);
        if (done) break;
    }
    if (best_f != 0x7fffffff) best = tri_d2_grad(px, py, nv.faces[best_f], gx, gy);
    return best;
}

// the same arg-min from the candidate list of the point's cell (tds::NearView, see nearest_face_d2_lists in map.hip).  As in the forward
// kernel a point is served by a GROUP of BL consecutive lanes that all hold it; `sub` is the lane's place in its group: a round evaluates BL
// candidates' DISTANCES side by side (one lane per point walked the list four at a time, gradients and all: 0.12 ms at 16 384 agents against the
// forward's 0.04), the group agrees on (minimum, lowest face index), and the gradient is taken once, from the winning face.
__device__ float nearest_face_d2_grad_lists(const MapView &m, const tds::NearView &nv, float px, float py, float &gx, float &gy, float stop, int sub) {
    if (nv.cand == nullptr || m.nx <= 0 || !(px == px) || !(py == py) || __builtin_isinf(px) || __builtin_isinf(py))
        return nearest_face_d2_grad(m, px, py, gx, gy, stop);
    // beyond the lists' grid: the hierarchy over the faces where the map has one (walked by the group together), else the walk over grid rings
    // (every lane of the group for itself: the same point, the same answer)
    auto beyond = [&]() { return nv.bvh != nullptr ? nearest_face_d2_grad_bvh(nv, px, py, gx, gy, stop, sub) : nearest_face_d2_grad(m, px, py, gx, gy, stop); };
    const float fx = (px - nv.ox) * m.inv_cell, fy = (py - nv.oy) * m.inv_cell;
    if (!(fx >= 0.0f && fy >= 0.0f && fx < (float)nv.nx && fy < (float)nv.ny)) return beyond();
    const int cx = tds::cell_coord(px, nv.ox, m.inv_cell), cy = tds::cell_coord(py, nv.oy, m.inv_cell);
    if (cx < 0 || cy < 0 || cx >= nv.nx || cy >= nv.ny) return beyond();
    const int s = nv.cand_start[cy * nv.nx + cx], e = nv.cand_start[cy * nv.nx + cx + 1];
    const float inf = __builtin_inff();
    float best = inf;                        // the group's minimum so far, the same in all its lanes
    int best_f = 0x7fffffff;                 // among faces at the same distance the one of lowest index gives the gradient (torch.min's choice)
    gx = gy = 0.0f;
    for (int i = s; i < e && best > stop; i += BL) {
        const int j = i + sub;
        tds::NearCand c;
        c.face = 0; c.lb = inf;
        if (j < e) c = nv.cand[j];
        const float lb0 = __shfl(c.lb, (threadIdx.x & 63 & ~(BL - 1)));                // the round's first candidate
        if (lb0 > best) break;                                                        // sorted by lb: nothing further can be as near
        float d = inf;
        int f = 0x7fffffff;
        if (j < e && !(c.lb > best)) {
            const GridEntry ge = nv.faces[c.face];
            // a face whose bounding box is already farther than the running minimum is not evaluated
            const float fx0 = fminf(ge.x0, fminf(ge.x1, ge.x2)), fx1 = fmaxf(ge.x0, fmaxf(ge.x1, ge.x2));
            const float fy0 = fminf(ge.y0, fminf(ge.y1, ge.y2)), fy1 = fmaxf(ge.y0, fmaxf(ge.y1, ge.y2));
            const float ex = fmaxf(fmaxf(fx0 - px, px - fx1), 0.0f), ey = fmaxf(fmaxf(fy0 - py, py - fy1), 0.0f);
            if (!((ex * ex + ey * ey) * 0.998f - 1e-3f > best)) {
                float tgx, tgy;
                const float t = tri_d2_grad(px, py, ge, tgx, tgy);                    // (the distance alone: the gradient is taken below)
                if (t < inf) { d = t; f = c.face; }                                   // a NaN distance never becomes the minimum
            }
        }
#pragma unroll
        for (int k = 1; k < BL; k <<= 1) {
            const float od = __shfl_xor(d, k);
            const int of = __shfl_xor(f, k);
            if (od < d || (od == d && of < f)) { d = od; f = of; }
        }
        if (d < best || (d == best && f < best_f)) { best = d; best_f = f; }
    }
    if (best_f != 0x7fffffff) best = tri_d2_grad(px, py, nv.faces[best_f], gx, gy);
    return best;
}

// BL lanes per agent corner, 4 consecutive groups = one agent; the 4 corners of an agent are reduced with shuffles
__global__ void __launch_bounds__(GBLOCK) offroad_bwd_kernel(MapView m, tds::NearView nv, const float4 *__restrict__ state, const float2 *__restrict__ lenwid,
                                                             const float2 *__restrict__ sc, const uint8_t *__restrict__ present,
                                                             const float *__restrict__ gout, float4 *__restrict__ gstate,
                                                             float2 *__restrict__ glenwid, float2 *__restrict__ gsc, int64_t n, float threshold,
                                                             const MapView *__restrict__ views, const tds::NearView *__restrict__ nears,
                                                             const int32_t *__restrict__ scene_map, int agents_per_scene) {
    const int64_t t = (int64_t)blockIdx.x * GBLOCK + threadIdx.x;
    const int64_t a = t / (4 * BL);
    const int k = (int)((t / BL) & 3), sub = (int)(t & (BL - 1));
    float gx = 0, gy = 0, gl = 0, gw = 0, gs = 0, gc = 0;
    if (a < n && views != nullptr) {
        const int im = scene_map[a / agents_per_scene];
        m = views[im]; nv = nears[im];
    }
    if (a < n && m.n_faces > 0) {
        float go = gout[a] * ((present && !present[a]) ? 0.0f : 1.0f);
        if (go != 0.0f) {                                                            // (uniform within the agent's 4 groups)
            float4 s = state[a];
            float2 lw = lenwid[a];
            float2 scv = sc[a];
            const float sx = (k == 0 || k == 3) ? 0.5f : -0.5f, sy = (k < 2) ? 0.5f : -0.5f;
            float x4 = sx * lw.x, y4 = sy * lw.y;
            float px = (x4 * scv.y + y4 * (-scv.x)) + s.x, py = (x4 * scv.x + y4 * scv.y) + s.y;
            float dgx, dgy;
            float d = nearest_face_d2_grad_lists(m, nv, px, py, dgx, dgy, fmaxf(threshold, 0.0f), sub);
            if (d == d && !__builtin_isinf(d) && d > threshold) {
                float ggx = go * dgx, ggy = go * dgy;
                gx = ggx; gy = ggy;
                gl = ggx * sx * scv.y + ggy * sx * scv.x;
                gw = -ggx * sy * scv.x + ggy * sy * scv.y;
                gc = ggx * x4 + ggy * y4;
                gs = -ggx * y4 + ggy * x4;
            }
        }
    }
    // sum over the 4 corners (every lane of a group holds its corner's values)
#pragma unroll
    for (int d = BL; d < 4 * BL; d <<= 1) {
        gx += __shfl_xor(gx, d); gy += __shfl_xor(gy, d); gl += __shfl_xor(gl, d); gw += __shfl_xor(gw, d); gs += __shfl_xor(gs, d); gc += __shfl_xor(gc, d);
    }
    if (a < n && k == 0 && sub == 0) {
        if (gstate) gstate[a] = make_float4(gx, gy, 0.0f, 0.0f);
        if (glenwid) glenwid[a] = make_float2(gl, gw);
        if (gsc) gsc[a] = make_float2(gs, gc);
    }
}

}  // namespace

TDS_EXPORT int tds_collision_bwd_f32(const float *boxes, const float *sc, const uint8_t *present, const float *grad_out, float *grad_boxes,
                                     float *grad_sc, int64_t B, int64_t A, int64_t N, int metric, void *stream) {
    TDS_CHECK_ARG(B >= 0 && A >= 0 && N >= A, "tds_collision_bwd_f32: bad sizes");
    TDS_CHECK_ARG(metric == TDS_METRIC_IOU || metric == TDS_METRIC_DISCS, "tds_collision_bwd_f32: unknown metric %d", metric);
    TDS_CHECK_ARG(grad_boxes && grad_sc, "tds_collision_bwd_f32: null gradient outputs");
    if (B == 0 || N == 0) return TDS_OK;
    auto zero_outputs = [&]() -> int {
        TDS_HIP(tds::zero_async(grad_boxes, (size_t)B * N * 5 * sizeof(float), (hipStream_t)stream));
        TDS_HIP(tds::zero_async(grad_sc, (size_t)B * N * 2 * sizeof(float), (hipStream_t)stream));
        return TDS_OK;
    };
    if (A == 0) return zero_outputs();
    TDS_CHECK_ARG(boxes && sc && present && grad_out, "tds_collision_bwd_f32: null pointer");
    // whole scene per workgroup where its tables fit (see collision_scene_bwd_kernel); one wavefront per (scene, agent) otherwise
    const size_t lds_base = ((size_t)N * 12 + 2 * (size_t)A + (size_t)A * N) * sizeof(float) + 2 * (size_t)A * N * sizeof(uint16_t);
    // (up to 16 384 pairs -- 64 exposed agents among 256 boxes, 128 among 128 -- since the kernel is the deterministic one: exposed agents plus NPCs
    // easily exceed the forward kernel's 4 096; beyond, or where the tables do not fit, one wavefront per row with global atomics)
    if (A * N <= SCENE_BWD_MAX_PAIRS && lds_base + 256 * 48 <= 150 * 1024) {
        // the gradient table: 256 .. 1024 pairs, as many as keep the workgroup within 80 KiB of LDS (two workgroups per CU, which is also what its
        // registers allow) where the scene's own tables leave room for that, else within 150 KiB (one per CU)
        const int64_t budget = lds_base + 256 * 48 <= 80 * 1024 ? 80 * 1024 : 150 * 1024;
        int64_t cap = (budget - (int64_t)lds_base) / 48;
        cap = std::max<int64_t>(256, std::min<int64_t>(1024, cap & ~(int64_t)255));
        const size_t lds_scene = lds_base + (size_t)cap * 48;
        auto launch = [&](auto kern) {
            if (lds_scene + 64 > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_scene);
            hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(GBLOCK), lds_scene, (hipStream_t)stream, boxes, sc, present, grad_out, grad_boxes, grad_sc, (int)A, (int)N, (int)cap);
        };
        if (metric == TDS_METRIC_IOU) launch(collision_scene_bwd_kernel<TDS_METRIC_IOU>); else launch(collision_scene_bwd_kernel<TDS_METRIC_DISCS>);
        TDS_LAUNCH_CHECK("collision_scene_bwd_kernel");
        return TDS_OK;                                                         // (that kernel writes every element of both outputs)
    }
    if (int rc = zero_outputs()) return rc;                                    // the row kernel adds into them
    dim3 grid((unsigned)((B * A * 64 + GBLOCK - 1) / GBLOCK));
    if (metric == TDS_METRIC_IOU)
        hipLaunchKernelGGL(collision_bwd_kernel<TDS_METRIC_IOU>, grid, dim3(GBLOCK), 0, (hipStream_t)stream, boxes, sc, present, grad_out,
                           grad_boxes, grad_sc, B, (int)A, (int)N);
    else
        hipLaunchKernelGGL(collision_bwd_kernel<TDS_METRIC_DISCS>, grid, dim3(GBLOCK), 0, (hipStream_t)stream, boxes, sc, present, grad_out,
                           grad_boxes, grad_sc, B, (int)A, (int)N);
    TDS_LAUNCH_CHECK("collision_bwd_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_offroad_bwd_f32(const tds_map_t *map, const float *state, const float *lenwid, const float *sc, const uint8_t *present,
                                   const float *grad_out, float *grad_state, float *grad_lenwid, float *grad_sc, int64_t n_agents,
                                   float threshold, void *stream) {
    TDS_CHECK_ARG(map, "tds_offroad_bwd_f32: null map");
    TDS_CHECK_ARG(n_agents >= 0, "tds_offroad_bwd_f32: bad agent count");
    if (n_agents == 0) return TDS_OK;
    TDS_CHECK_ARG(state && lenwid && sc && grad_out, "tds_offroad_bwd_f32: null pointer");
    int64_t threads = n_agents * 4 * BL;
    hipLaunchKernelGGL(offroad_bwd_kernel, dim3((unsigned)((threads + GBLOCK - 1) / GBLOCK)), dim3(GBLOCK), 0, (hipStream_t)stream, map->view,
                       map->near, (const float4 *)state, (const float2 *)lenwid, (const float2 *)sc, present, grad_out, (float4 *)grad_state,
                       (float2 *)grad_lenwid, (float2 *)grad_sc, n_agents, threshold, (const MapView *)nullptr, (const tds::NearView *)nullptr,
                       (const int32_t *)nullptr, 1);
    TDS_LAUNCH_CHECK("offroad_bwd_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_offroad_multi_bwd_f32(const tds_mapset_t *set, const int32_t *scene_map, int64_t agents_per_scene, const float *state,
                                         const float *lenwid, const float *sc, const uint8_t *present, const float *grad_out, float *grad_state,
                                         float *grad_lenwid, float *grad_sc, int64_t n_agents, float threshold, void *stream) {
    TDS_CHECK_ARG(set && set->n > 0 && scene_map, "tds_offroad_multi_bwd_f32: null map set or scene index array");
    TDS_CHECK_ARG(agents_per_scene > 0 && agents_per_scene < (1 << 30), "tds_offroad_multi_bwd_f32: bad number of agents per scene");
    TDS_CHECK_ARG(n_agents >= 0, "tds_offroad_multi_bwd_f32: bad agent count");
    if (n_agents == 0) return TDS_OK;
    TDS_CHECK_ARG(state && lenwid && sc && grad_out, "tds_offroad_multi_bwd_f32: null pointer");
    int64_t threads = n_agents * 4 * BL;
    hipLaunchKernelGGL(offroad_bwd_kernel, dim3((unsigned)((threads + GBLOCK - 1) / GBLOCK)), dim3(GBLOCK), 0, (hipStream_t)stream, tds::MapView{},
                       tds::NearView{nullptr, nullptr, nullptr, 0.0f, 0.0f, 0, 0, nullptr, nullptr}, (const float4 *)state, (const float2 *)lenwid, (const float2 *)sc, present,
                       grad_out, (float4 *)grad_state, (float2 *)grad_lenwid, (float2 *)grad_sc, n_agents, threshold, (const MapView *)set->d_views,
                       (const tds::NearView *)set->d_near, scene_map, (int)agents_per_scene);
    TDS_LAUNCH_CHECK("offroad_bwd_kernel");
    return TDS_OK;
}
