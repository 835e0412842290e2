// K2a: pairwise oriented-box overlap (Rotated-IoU and 5-disc metric) and Simulator.compute_collision fused over all
// exposed agents of a scene.  One wavefront owns a row i of a scene (lanes = other agents j); 4 rows per wave.
// Reference: simulator.py:1064-1109,1161-1194; _iou_utils.py:42-367; infractions.py:378-426,503-545.
// Compute bound (~2k VALU ops per surviving pair, 16 IEEE divisions); bytes are negligible (20 B/agent).
#include "tds_common.h"
#include <algorithm>

namespace {

constexpr int CBLOCK = 256;            // 4 waves
constexpr int ROWS_PER_WAVE = 4;
constexpr float PI_F = 3.14159265358979323846f;

struct Box { float x, y, l, w; float s, c; };
struct Corners { float x[4], y[4]; };

__device__ __forceinline__ float scrub(float v) {          // torch.nan_to_num(nan=0) simulator.py:1095-1096
    if (v != v) return 0.0f;
    if (__builtin_isinf(v)) return v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    return v;
}

// box2corners_th, _iou_utils.py:270-299: corners @ [[c,s],[-s,c]] + centre, one rounding per op
__device__ __forceinline__ Corners corners_of(const Box &b) {
    const float sx[4] = {0.5f, -0.5f, -0.5f, 0.5f}, sy[4] = {0.5f, 0.5f, -0.5f, -0.5f};
    Corners c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x4 = sx[k] * b.l, y4 = sy[k] * b.w;
        float rx = x4 * b.c + y4 * (-b.s);
        float ry = x4 * b.s + y4 * b.c;
        c.x[k] = rx + b.x;
        c.y[k] = ry + b.y;
    }
    return c;
}

// corner k of `p` inside box `q` (box1_in_box2, _iou_utils.py:87-114).
// round(q*1e6)/1e6 in (-1e-6, 1+1e-6)  <=>  0 <= rint(q*1e6) <= 1e6  (both divisions by 1e6 are correctly rounded
// and monotone, fl(-1/1e6) == fl32(-1e-6), fl(1000001/1e6) == fl32(1+1e-6)), which saves two divisions per test.
__device__ __forceinline__ unsigned corners_in(const Corners &p, const Corners &q) {
    float ax = q.x[0], ay = q.y[0];
    float abx = q.x[1] - ax, aby = q.y[1] - ay;
    float adx = q.x[3] - ax, ady = q.y[3] - ay;
    float nab = abx * abx + aby * aby, nad = adx * adx + ady * ady;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float amx = p.x[k] - ax, amy = p.y[k] - ay;
        float r1 = rintf(((abx * amx + aby * amy) / nab) * 1000000.0f);
        float r2 = rintf(((adx * amx + ady * amy) / nad) * 1000000.0f);
        bool in = (r1 >= 0.0f) && (r1 <= 1000000.0f) && (r2 >= 0.0f) && (r2 <= 1000000.0f);
        m |= (in ? 1u : 0u) << k;
    }
    return m;
}

// Rare path (more than 8 valid candidates, _iou_utils.py:191-214) and the general definition: mirrors
// sort_indices/calculate_area on arrays (lives in scratch; only entered when the fast path cannot apply).
__device__ __noinline__ float area_general(const float *vx, const float *vy, unsigned mask, float cx, float cy) {
    float ang[24];
    int order[24];
    int n = __popc(mask);
    for (int k = 0; k < 24; ++k) {
        float dx = vx[k] - cx, dy = vy[k] - cy;
        float r = sqrtf(dx * dx + dy * dy);
        float a = acosf(dx / r);
        ang[k] = (dy > 0.0f) ? a : (2.0f * PI_F - a);
    }
    for (;;) {
        for (int k = 0; k < 24; ++k) order[k] = k;
        for (int a = 1; a < 24; ++a) {
            int o = order[a];
            float key = ((mask >> o) & 1) ? ang[o] : __builtin_inff();
            int b = a - 1;
            while (b >= 0) {
                float kb = ((mask >> order[b]) & 1) ? ang[order[b]] : __builtin_inff();
                bool gt = ((kb != kb) && (key == key)) || (kb > key);
                if (!gt) break;
                order[b + 1] = order[b];
                --b;
            }
            order[b + 1] = o;
        }
        if (n <= 8) break;
        int best = 0;
        float bestd = __builtin_inff();
        for (int k = 0; k < n - 1; ++k) {
            float dx = vx[order[k]] - vx[order[k + 1]], dy = vy[order[k]] - vy[order[k + 1]];
            float d = sqrtf(dx * dx + dy * dy);
            if (d < bestd) { bestd = d; best = k; }
        }
        mask &= ~(1u << order[best]);
        --n;
    }
    if (n < 3) return 0.0f;
    float total = 0.0f;
    for (int k = 0; k < n; ++k) {
        int a = order[k], b = order[(k + 1 == n) ? 0 : k + 1];
        total = total + (vx[a] * vy[b] - vy[a] * vx[b]);
    }
    return fabsf(total) / 2.0f;
}

// Intersection area of two rectangles (oriented_box_intersection_2d, _iou_utils.py:250-267).
// `scr` is this wave's LDS scratch: 16 planes of 64 floats (8 slots x {x,y}) addressed [plane*64 + lane].
__device__ float intersection_area(const Corners &c1, const Corners &c2, float *scr, int lane) {
    float acc_x[4] = {0.f, 0.f, 0.f, 0.f}, acc_y[4] = {0.f, 0.f, 0.f, 0.f};
    int n = 0;
    unsigned mask = 0;
    auto push = [&](int k, float x, float y) {
        acc_x[k & 3] = acc_x[k & 3] + x;           // torch.sum over the strided dim: 4 interleaved accumulators
        acc_y[k & 3] = acc_y[k & 3] + y;
        if (n < 8) { scr[(2 * n) * 64 + lane] = x; scr[(2 * n + 1) * 64 + lane] = y; }
        ++n;
        mask |= 1u << k;
    };
    unsigned in12 = corners_in(c1, c2), in21 = corners_in(c2, c1);
#pragma unroll
    for (int k = 0; k < 4; ++k) if ((in12 >> k) & 1) push(k, c1.x[k], c1.y[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) if ((in21 >> k) & 1) push(4 + k, c2.x[k], c2.y[k]);
    // box_intersection_th, _iou_utils.py:42-84.  0 < fl(a/b) < 1  <=>  a, b same sign and |a| < |b| for finite floats,
    // so the two mask divisions are replaced by comparisons; only the point itself needs den_t / (num + 1e-8).
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float x1 = c1.x[i], y1 = c1.y[i], x2 = c1.x[(i + 1) & 3], y2 = c1.y[(i + 1) & 3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x3 = c2.x[j], y3 = c2.y[j], x4 = c2.x[(j + 1) & 3], y4 = c2.y[(j + 1) & 3];
            float num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
            float den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
            float nden_u = -((x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3));
            float an = fabsf(num);
            bool ok = !(an < (float)1e-4) && (an == an);
            bool mt = ok && ((den_t > 0.0f) == (num > 0.0f)) && (den_t != 0.0f) && (fabsf(den_t) < an);
            bool mu = ok && ((nden_u > 0.0f) == (num > 0.0f)) && (nden_u != 0.0f) && (fabsf(nden_u) < an);
            if (mt && mu) {
                float t = den_t / (num + (float)1e-8);
                push(8 + i * 4 + j, x1 + t * (x2 - x1), y1 + t * (y2 - y1));
            }
        }
    }
    if (n < 3) return 0.0f;
    float fn = (float)n;
    float cx = (((acc_x[0] + acc_x[1]) + acc_x[2]) + acc_x[3]) / fn;
    float cy = (((acc_y[0] + acc_y[1]) + acc_y[2]) + acc_y[3]) / fn;
    if (n > 8) {
        // rebuild the 24-candidate arrays for the general path (never seen in 1M random pairs)
        float vx[24], vy[24];
        for (int k = 0; k < 4; ++k) { vx[k] = c1.x[k]; vy[k] = c1.y[k]; vx[4 + k] = c2.x[k]; vy[4 + k] = c2.y[k]; }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            float x1 = c1.x[i], y1 = c1.y[i], x2 = c1.x[(i + 1) & 3], y2 = c1.y[(i + 1) & 3];
            float x3 = c2.x[j], y3 = c2.y[j], x4 = c2.x[(j + 1) & 3], y4 = c2.y[(j + 1) & 3];
            float num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
            float den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
            float t = den_t / (num + (float)1e-8);
            float fm = ((mask >> (8 + i * 4 + j)) & 1) ? 1.0f : 0.0f;
            vx[8 + i * 4 + j] = (x1 + t * (x2 - x1)) * fm;
            vy[8 + i * 4 + j] = (y1 + t * (y2 - y1)) * fm;
        }
        return area_general(vx, vy, mask, cx, cy);
    }
    // angular sort of the <= 8 valid vertices.  acos is monotone, so ordering by the angle equals ordering by
    // key = (dy > 0) ? -q : 2 + q with q = dx / r  (_iou_utils.py:181-186); invalid slots sort last.
    float sxv[8], syv[8], key[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float x = scr[(2 * s) * 64 + lane], y = scr[(2 * s + 1) * 64 + lane];
        float dx = x - cx, dy = y - cy;
        float q = dx / sqrtf(dx * dx + dy * dy);
        float k = (dy > 0.0f) ? -q : 2.0f + q;
        bool live = s < n;
        sxv[s] = live ? x : 0.0f;
        syv[s] = live ? y : 0.0f;
        key[s] = live ? ((k == k) ? k : 3.5f) : 4.0f;      // NaN keys after valid ones, dead slots last
    }
#define TDS_CSWAP(a, b)                                                                      \
    {                                                                                        \
        bool sw = key[a] > key[b];                                                           \
        float tk = sw ? key[b] : key[a], uk = sw ? key[a] : key[b];                          \
        float tx = sw ? sxv[b] : sxv[a], ux = sw ? sxv[a] : sxv[b];                          \
        float ty = sw ? syv[b] : syv[a], uy = sw ? syv[a] : syv[b];                          \
        key[a] = tk; key[b] = uk; sxv[a] = tx; sxv[b] = ux; syv[a] = ty; syv[b] = uy;        \
    }
    TDS_CSWAP(0, 1) TDS_CSWAP(2, 3) TDS_CSWAP(4, 5) TDS_CSWAP(6, 7)
    TDS_CSWAP(0, 2) TDS_CSWAP(1, 3) TDS_CSWAP(4, 6) TDS_CSWAP(5, 7)
    TDS_CSWAP(1, 2) TDS_CSWAP(5, 6) TDS_CSWAP(0, 4) TDS_CSWAP(3, 7)
    TDS_CSWAP(1, 5) TDS_CSWAP(2, 6)
    TDS_CSWAP(1, 4) TDS_CSWAP(3, 6)
    TDS_CSWAP(2, 4) TDS_CSWAP(3, 5)
    TDS_CSWAP(3, 4)
#undef TDS_CSWAP
    // shoelace over [v0..v(n-1), v0] (calculate_area, _iou_utils.py:230-247), summed in order
    float total = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float nx = (k + 1 < 8) ? sxv[k + 1] : 0.0f, ny = (k + 1 < 8) ? syv[k + 1] : 0.0f;
        if (k + 1 >= n) { nx = sxv[0]; ny = syv[0]; }
        float term = sxv[k] * ny - syv[k] * nx;
        if (k < n) total = total + term;
    }
    return fabsf(total) / 2.0f;
}

__device__ float iou_pair(const Box &b1, const Corners &c1, const Box &b2, const Corners &c2, float *scr, int lane) {
    float a1 = b1.l * b1.w, a2 = b2.l * b2.w;
    // disjoint bounding circles (with margin) -> no valid candidate -> area 0 exactly, as in the reference
    float dx = b1.x - b2.x, dy = b1.y - b2.y;
    float r1 = 0.5f * sqrtf(b1.l * b1.l + b1.w * b1.w), r2 = 0.5f * sqrtf(b2.l * b2.l + b2.w * b2.w);
    float reach = r1 + r2 + 0.05f + 1e-5f * fmaxf(fmaxf(fabsf(b1.x), fabsf(b1.y)), fmaxf(fabsf(b2.x), fabsf(b2.y)));
    float inter = 0.0f;
    if (!(dx * dx + dy * dy > reach * reach)) inter = intersection_area(c1, c2, scr, lane);
    float u = a1 + a2 - inter;
    return inter / u;                          // iou_differentiable_fast :363-367
}

// bbox2discs + cdist + relu, infractions.py:378-426,503-545.  b.s/b.c are of yaw + pi/2*(wid>len).
__device__ float discs_pair(const Box &b1, const Box &b2) {
    float ra = fminf(b1.l, b1.w) / 2.0f, rb = fminf(b2.l, b2.w) / 2.0f;
    float ha = fmaxf(b1.l, b1.w) / 2.0f - ra, hb = fmaxf(b2.l, b2.w) / 2.0f - rb;
    float d = __builtin_inff();
    bool any_nan = false;
#pragma unroll
    for (int i = -2; i <= 2; ++i) {
        float da = ((float)i * ha) / 2.0f;
        float ax = (da * b1.c - 0.0f * b1.s) + b1.x, ay = (da * b1.s + 0.0f * b1.c) + b1.y;
#pragma unroll
        for (int j = -2; j <= 2; ++j) {
            float db = ((float)j * hb) / 2.0f;
            float bx = (db * b2.c - 0.0f * b2.s) + b2.x, by = (db * b2.s + 0.0f * b2.c) + b2.y;
            float ex = ax - bx, ey = ay - by;
            float dd = sqrtf(__fmaf_rn(ey, ey, ex * ex));    // torch.cdist accumulates with an FMA (probed)
            any_nan |= (dd != dd);
            d = fminf(d, dd);
        }
    }
    if (any_nan) d = __builtin_nanf("");
    float l = 1.0f - d / (ra + rb);
    return (l != l) ? l : fmaxf(l, 0.0f);
}

// the same with 2 * nps + 1 discs per box (`num_discs`, infractions.py:390-400): centres at i * (max/2 - r) / nps
__device__ float discs_pair_n(const Box &b1, const Box &b2, int nps) {
    float ra = fminf(b1.l, b1.w) / 2.0f, rb = fminf(b2.l, b2.w) / 2.0f;
    float ha = fmaxf(b1.l, b1.w) / 2.0f - ra, hb = fmaxf(b2.l, b2.w) / 2.0f - rb;
    float d = __builtin_inff();
    bool any_nan = false;
    const float fn = (float)nps;
    for (int i = -nps; i <= nps; ++i) {
        float da = ((float)i * ha) / fn;
        float ax = (da * b1.c - 0.0f * b1.s) + b1.x, ay = (da * b1.s + 0.0f * b1.c) + b1.y;
        for (int j = -nps; j <= nps; ++j) {
            float db = ((float)j * hb) / fn;
            float bx = (db * b2.c - 0.0f * b2.s) + b2.x, by = (db * b2.s + 0.0f * b2.c) + b2.y;
            float ex = ax - bx, ey = ay - by;
            float dd = sqrtf(__fmaf_rn(ey, ey, ex * ex));
            any_nan |= (dd != dd);
            d = fminf(d, dd);
        }
    }
    if (any_nan) d = __builtin_nanf("");
    float l = 1.0f - d / (ra + rb);
    return (l != l) ? l : fmaxf(l, 0.0f);
}

__device__ __forceinline__ Box load_box(const float *boxes, const float *sc, int64_t idx) {
    Box b;
    const float *p = boxes + idx * 5;
    b.x = scrub(p[0]); b.y = scrub(p[1]); b.l = scrub(p[2]); b.w = scrub(p[3]);
    float psi = p[4];
    b.s = sc[idx * 2]; b.c = sc[idx * 2 + 1];
    if (psi != psi) { b.s = 0.0f; b.c = 1.0f; }      // a NaN heading is scrubbed to 0 before sin/cos (simulator.py:1095)
    return b;
}

// Row reduction of K2a on a wavefront (north star: "wavefront ballot/reduce for overlap flags"): the lanes hold the overlaps o_ij of 64
// consecutive partners j.  collision_i = sum_j o_ij - max_j o_ij (simulator.py:1105-1108) keeps the reference's INDEX-ORDER sum bit for bit:
// the overlaps are >= 0 and all but a few are exactly 0, and x + 0 = x, so lane 0 adds only the non-zero ones (ballot), in index order.
// The maximum, the overlap bit mask (ballot of o > 0, j != i) and the arg-max partner (lowest j attaining the largest o > 0) are
// order-independent wave reductions.
struct RowAcc { float sum, mx, best; int arg; uint64_t bits; };
__device__ __forceinline__ RowAcc row_acc_init() { RowAcc a; a.sum = 0.0f; a.mx = -__builtin_inff(); a.best = 0.0f; a.arg = -1; a.bits = 0; return a; }
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
}
// o: this lane's overlap (already scrubbed and masked), valid: j = j0 + lane < N.  Every lane returns the same accumulator.
__device__ __forceinline__ void row_acc_chunk(RowAcc &a, float o, bool valid, int i, int j0, int lane) {
    const int j = j0 + lane;
    const unsigned long long nz = __ballot(valid && o != 0.0f);
    for (unsigned long long m = nz; m != 0; m &= m - 1) {                 // wave-uniform loop over the non-zero overlaps, ascending j
        const int l = (int)__ffsll((long long)m) - 1;
        a.sum = a.sum + __shfl(o, l);
    }
    a.mx = fmaxf(a.mx, wave_max_f32(valid ? o : -__builtin_inff()));
    const bool hit = valid && j != i && o > 0.0f;
    if (j0 == 0) a.bits = __ballot(hit);
    const float cm = wave_max_f32(hit ? o : 0.0f);
    if (cm > a.best) {                                                    // wave-uniform
        a.best = cm;
        a.arg = j0 + (int)__ffsll((long long)__ballot(hit && o == cm)) - 1;
    }
}

// grid = (B, ceil(A / (4 waves * ROWS_PER_WAVE))), dynamic LDS = 4 * (16*64 + Npad) floats
template <int METRIC>
__global__ void __launch_bounds__(CBLOCK) collision_kernel(const float *__restrict__ boxes, const float *__restrict__ sc,
                                                           const uint8_t *__restrict__ present, float *__restrict__ out,
                                                           uint64_t *__restrict__ overlap, int32_t *__restrict__ partner, int A, int N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int npad = (N + 63) & ~63;
    float *scr = smem + wave * (16 * 64 + npad);
    const int64_t b = blockIdx.x;
    const int row0 = (blockIdx.y * (CBLOCK / 64) + wave) * ROWS_PER_WAVE;
    for (int r = 0; r < ROWS_PER_WAVE; ++r) {
        const int i = row0 + r;
        if (i >= A) break;                                           // wave-uniform
        Box bi = load_box(boxes, sc, b * N + i);
        Corners ci = corners_of(bi);
        RowAcc acc = row_acc_init();
        for (int j0 = 0; j0 < N; j0 += 64) {
            int j = j0 + lane;
            float o = 0.0f;
            if (j < N) {
                Box bj = load_box(boxes, sc, b * N + j);
                if (METRIC == TDS_METRIC_IOU) {
                    Corners cj = corners_of(bj);
                    o = iou_pair(bi, ci, bj, cj, scr, lane);
                } else {
                    o = discs_pair(bi, bj);
                }
                o = scrub(o) * (present[b * N + j] ? 1.0f : 0.0f);   // simulator.py:1103-1104
            }
            row_acc_chunk(acc, o, j < N, i, j0, lane);
        }
        if (lane == 0) {
            out[b * A + i] = acc.sum - acc.mx;
            if (overlap) overlap[b * A + i] = acc.bits;
            if (partner) partner[b * A + i] = acc.arg;
        }
    }
}

// IoU metric, whole scene per workgroup: most of the A x N pairs are far apart (bounding circles disjoint -> overlap exactly 0, see
// iou_pair) and every row contains its own box, so instead of running the Rotated-IoU pipeline once per row for a handful of live
// lanes, the near pairs of the scene are first gathered into an LDS list and then evaluated on full waves.  Same arithmetic per pair
// and the same summation order per row as collision_kernel.  LDS: boxes N x 6, O = A x N overlaps, the pair list, 4 x scratch.
__global__ void __launch_bounds__(CBLOCK) collision_scene_iou_kernel(const float *__restrict__ boxes, const float *__restrict__ sc,
                                                                     const uint8_t *__restrict__ present, float *__restrict__ out,
                                                                     uint64_t *__restrict__ overlap, int32_t *__restrict__ partner, int A, int N) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int n_near;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    float *scr = smem + wave * (16 * 64);
    float *bx = smem + (CBLOCK / 64) * (16 * 64);          // N x 6
    float *O = bx + N * 6;                                  // A x N
    uint32_t *list = (uint32_t *)(O + A * N);               // A x N at most
    if (tid == 0) n_near = 0;
    for (int j = tid; j < N; j += CBLOCK) {
        const Box q = load_box(boxes, sc, b * N + j);
        bx[6 * j] = q.x; bx[6 * j + 1] = q.y; bx[6 * j + 2] = q.l; bx[6 * j + 3] = q.w; bx[6 * j + 4] = q.s; bx[6 * j + 5] = q.c;
    }
    __syncthreads();
    auto box_at = [&](int j) { Box q; q.x = bx[6 * j]; q.y = bx[6 * j + 1]; q.l = bx[6 * j + 2]; q.w = bx[6 * j + 3]; q.s = bx[6 * j + 4]; q.c = bx[6 * j + 5]; return q; };
    for (int p = tid; p < A * N; p += CBLOCK) {
        const int i = p / N, j = p - i * N;
        const Box b1 = box_at(i), b2 = box_at(j);
        // the test of iou_pair: pairs that fail it have overlap 0 (scrubbed) whatever else happens
        const float dx = b1.x - b2.x, dy = b1.y - b2.y;
        const float r1 = 0.5f * sqrtf(b1.l * b1.l + b1.w * b1.w), r2 = 0.5f * sqrtf(b2.l * b2.l + b2.w * b2.w);
        const float reach = r1 + r2 + 0.05f + 1e-5f * fmaxf(fmaxf(fabsf(b1.x), fabsf(b1.y)), fmaxf(fabsf(b2.x), fabsf(b2.y)));
        O[p] = 0.0f;
        if (!(dx * dx + dy * dy > reach * reach)) list[atomicAdd(&n_near, 1)] = (uint32_t)p;
    }
    __syncthreads();
    const int total = n_near;
    for (int q = tid; q < total; q += CBLOCK) {
        const int p = (int)list[q], i = p / N, j = p - i * N;
        const Box b1 = box_at(i), b2 = box_at(j);
        const Corners c1 = corners_of(b1), c2 = corners_of(b2);
        float o = iou_pair(b1, c1, b2, c2, scr, lane);
        O[p] = scrub(o) * (present[b * N + j] ? 1.0f : 0.0f);       // simulator.py:1103-1104
    }
    __syncthreads();
    // sum_j o_ij in index order and max_j (simulator.py:1105-1108): one wavefront per row, lanes = partners (row_acc_chunk)
    for (int i = wave; i < A; i += CBLOCK / 64) {
        RowAcc acc = row_acc_init();
        for (int j0 = 0; j0 < N; j0 += 64) {
            const int j = j0 + lane;
            row_acc_chunk(acc, j < N ? O[i * N + j] : 0.0f, j < N, i, j0, lane);
        }
        if (lane == 0) {
            out[b * A + i] = acc.sum - acc.mx;
            if (overlap) overlap[b * A + i] = acc.bits;
            if (partner) partner[b * A + i] = acc.arg;
        }
    }
}

template <int METRIC>
__global__ void __launch_bounds__(CBLOCK) pairwise_kernel(const float *__restrict__ box1, const float *__restrict__ sc1,
                                                          const float *__restrict__ box2, const float *__restrict__ sc2,
                                                          float *__restrict__ out, int64_t n) {
    __shared__ float smem[(CBLOCK / 64) * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t i = (int64_t)blockIdx.x * CBLOCK + threadIdx.x;
    if (i >= n) return;
    // element-wise API: no NaN scrubbing here (iou_differentiable itself does none, infractions.py:307-324)
    Box a, b;
    a.x = box1[5 * i]; a.y = box1[5 * i + 1]; a.l = box1[5 * i + 2]; a.w = box1[5 * i + 3]; a.s = sc1[2 * i]; a.c = sc1[2 * i + 1];
    b.x = box2[5 * i]; b.y = box2[5 * i + 1]; b.l = box2[5 * i + 2]; b.w = box2[5 * i + 3]; b.s = sc2[2 * i]; b.c = sc2[2 * i + 1];
    float o;
    if (METRIC == TDS_METRIC_IOU) {
        Corners ca = corners_of(a), cb = corners_of(b);
        o = iou_pair(a, ca, b, cb, smem + wave * 16 * 64, lane);
    } else {
        o = discs_pair(a, b);
    }
    out[i] = o;
}

__global__ void __launch_bounds__(CBLOCK) pairwise_discs_n_kernel(const float *__restrict__ box1, const float *__restrict__ sc1,
                                                                  const float *__restrict__ box2, const float *__restrict__ sc2,
                                                                  float *__restrict__ out, int64_t n, int nps) {
    int64_t i = (int64_t)blockIdx.x * CBLOCK + threadIdx.x;
    if (i >= n) return;
    Box a, b;
    a.x = box1[5 * i]; a.y = box1[5 * i + 1]; a.l = box1[5 * i + 2]; a.w = box1[5 * i + 3]; a.s = sc1[2 * i]; a.c = sc1[2 * i + 1];
    b.x = box2[5 * i]; b.y = box2[5 * i + 1]; b.l = box2[5 * i + 2]; b.w = box2[5 * i + 3]; b.s = sc2[2 * i]; b.c = sc2[2 * i + 1];
    out[i] = discs_pair_n(a, b, nps);
}

__global__ void __launch_bounds__(CBLOCK) box2corners_kernel(const float *__restrict__ box, const float *__restrict__ sc,
                                                             float *__restrict__ corners, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * CBLOCK + threadIdx.x;
    if (i >= n) return;
    Box b;
    b.x = box[5 * i]; b.y = box[5 * i + 1]; b.l = box[5 * i + 2]; b.w = box[5 * i + 3]; b.s = sc[2 * i]; b.c = sc[2 * i + 1];
    Corners c = corners_of(b);
    float4 *o = (float4 *)(corners + 8 * i);
    o[0] = make_float4(c.x[0], c.y[0], c.x[1], c.y[1]);
    o[1] = make_float4(c.x[2], c.y[2], c.x[3], c.y[3]);
}

}  // namespace

TDS_EXPORT int tds_collision_f32(const float *boxes, const float *sc, const uint8_t *present, float *out, uint64_t *overlap,
                                 int32_t *partner, int64_t B, int64_t A, int64_t N, int metric, void *stream) {
    TDS_CHECK_ARG(B >= 0 && A >= 0 && N >= A, "tds_collision_f32: bad sizes B=%lld A=%lld N=%lld", (long long)B, (long long)A, (long long)N);
    TDS_CHECK_ARG(metric == TDS_METRIC_IOU || metric == TDS_METRIC_DISCS, "tds_collision_f32: unknown metric %d", metric);
    if (B == 0 || A == 0) return TDS_OK;
    TDS_CHECK_ARG(boxes && sc && present && out, "tds_collision_f32: null pointer");
    TDS_CHECK_ARG(!overlap || N <= 64, "tds_collision_f32: overlap bit masks need N <= 64 (got %lld)", (long long)N);
    TDS_CHECK_ARG(N <= 8192 && B < 65536 * 32768ll, "tds_collision_f32: N=%lld too large", (long long)N);
    // whole scene per workgroup: near pairs gathered first (see collision_scene_iou_kernel).  Its LDS footprint grows with N, so scenes
    // with few exposed agents and many NPCs (A=4, N=1024: 73.7 KB) take the row kernel instead of exceeding the 64 KiB default.
    const size_t lds_scene = ((size_t)(CBLOCK / 64) * 16 * 64 + (size_t)N * 6 + 2 * (size_t)A * N) * sizeof(float);
    if (metric == TDS_METRIC_IOU && A * N <= 4096 && lds_scene <= 64 * 1024) {
        hipLaunchKernelGGL(collision_scene_iou_kernel, dim3((unsigned)B), dim3(CBLOCK), lds_scene, (hipStream_t)stream, boxes, sc, present, out,
                           overlap, partner, (int)A, (int)N);
        TDS_LAUNCH_CHECK("collision_scene_iou_kernel");
        return TDS_OK;
    }
    const int rows_per_block = (CBLOCK / 64) * ROWS_PER_WAVE;
    dim3 grid((unsigned)B, (unsigned)((A + rows_per_block - 1) / rows_per_block));
    size_t lds = (size_t)(CBLOCK / 64) * (16 * 64 + ((N + 63) & ~63)) * sizeof(float);
    if (metric == TDS_METRIC_IOU)
        hipLaunchKernelGGL(collision_kernel<TDS_METRIC_IOU>, grid, dim3(CBLOCK), lds, (hipStream_t)stream, boxes, sc, present, out,
                           overlap, partner, (int)A, (int)N);
    else
        hipLaunchKernelGGL(collision_kernel<TDS_METRIC_DISCS>, grid, dim3(CBLOCK), lds, (hipStream_t)stream, boxes, sc, present, out,
                           overlap, partner, (int)A, (int)N);
    TDS_LAUNCH_CHECK("collision_kernel");
    return TDS_OK;
}

namespace {

// ---------------------------------------------------------------------------------------------------------
// `nograd` metric: per agent the NUMBER of other present agents whose rectangle overlaps it with non-zero area
// (simulator.py:1111-1149 -> infractions.compute_agent_collisions_metric :352-375 -> get_all_intersections :429-474, where shapely
// answers `a.intersection(b).area != 0`).  The corners are built like infractions.rectangle_vertices (:476-500, float32, length along
// the heading); two convex quadrilaterals share area iff no edge line of either one has the whole other one on its outer side or on
// the line itself -- evaluated in float64 on the float32 corners (the orientation determinants are then exact up to one rounding of a
// 60-bit product, far below the fp32 noise that the IoU pipeline in absolute coordinates carries for centimetre-deep overlaps).
// One wavefront per agent i, lanes = partners j; the hits of 64 partners are one ballot + popcount.
// ---------------------------------------------------------------------------------------------------------
struct Quad { float x[4], y[4]; };

__device__ inline Quad rectangle_vertices(float cx, float cy, float w, float h, float s, float c) {
    const float dx = w / 2.0f, dy = h / 2.0f;
    const float dxcos = dx * c, dxsin = dx * s, dycos = dy * c, dysin = dy * s;
    Quad q;
    q.x[0] = cx + (-dxcos - -dysin); q.y[0] = cy + (-dxsin + -dycos);
    q.x[1] = cx + (dxcos - -dysin);  q.y[1] = cy + (dxsin + -dycos);
    q.x[2] = cx + (dxcos - dysin);   q.y[2] = cy + (dxsin + dycos);
    q.x[3] = cx + (-dxcos - dysin);  q.y[3] = cy + (-dxsin + dycos);
    return q;
}

// some edge line of p has all of q on its right or on it (p counter-clockwise: its inside is on the left)
__device__ inline bool separated_by_an_edge_of(const Quad &p, const Quad &q) {
    bool sep = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double ax = p.x[e], ay = p.y[e], ex = (double)p.x[(e + 1) & 3] - ax, ey = (double)p.y[(e + 1) & 3] - ay;
        bool all_out = true;
#pragma unroll
        for (int v = 0; v < 4; ++v) all_out = all_out && (ex * ((double)q.y[v] - ay) - ey * ((double)q.x[v] - ax) <= 0.0);
        sep = sep || all_out;
    }
    return sep;
}

__global__ void __launch_bounds__(CBLOCK) overlap_count_kernel(const float *__restrict__ boxes, const float *__restrict__ sc,
                                                               const uint8_t *__restrict__ present, double *__restrict__ out, int A) {
    extern __shared__ float lds[];                      // A x 8 corner coordinates of the scene
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *bx = boxes + (size_t)b * A * 5, *bs = sc + (size_t)b * A * 2;
    const uint8_t *pr = present + (size_t)b * A;
    for (int j = tid; j < A; j += CBLOCK) {
        const Quad q = rectangle_vertices(bx[5 * j], bx[5 * j + 1], bx[5 * j + 2], bx[5 * j + 3], bs[2 * j], bs[2 * j + 1]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { lds[8 * j + 2 * k] = q.x[k]; lds[8 * j + 2 * k + 1] = q.y[k]; }
    }
    __syncthreads();
    for (int i = blockIdx.y * (CBLOCK / 64) + wave; i < A; i += gridDim.y * (CBLOCK / 64)) {
        Quad p;
#pragma unroll
        for (int k = 0; k < 4; ++k) { p.x[k] = lds[8 * i + 2 * k]; p.y[k] = lds[8 * i + 2 * k + 1]; }
        int count = 0;
        for (int j0 = 0; j0 < A; j0 += 64) {
            const int j = j0 + lane;
            bool hit = false;
            if (j < A && j != i && pr[j]) {
                Quad q;
#pragma unroll
                for (int k = 0; k < 4; ++k) { q.x[k] = lds[8 * j + 2 * k]; q.y[k] = lds[8 * j + 2 * k + 1]; }
                hit = !(separated_by_an_edge_of(p, q) || separated_by_an_edge_of(q, p));
            }
            count += __popcll(__ballot(hit));
        }
        if (lane == 0) out[(size_t)b * A + i] = pr[i] ? (double)count : 0.0;
    }
}

}  // namespace

TDS_EXPORT int tds_overlap_count_f32(const float *boxes, const float *sc, const uint8_t *present, double *out, int64_t B, int64_t A,
                                     void *stream) {
    TDS_CHECK_ARG(B >= 0 && A >= 0, "tds_overlap_count_f32: bad sizes B=%lld A=%lld", (long long)B, (long long)A);
    if (B == 0 || A == 0) return TDS_OK;
    TDS_CHECK_ARG(boxes && sc && present && out, "tds_overlap_count_f32: null pointer");
    TDS_CHECK_ARG(A <= 4096 && B < 65536 * 32768ll, "tds_overlap_count_f32: A=%lld too large (the corners of a scene are staged in LDS)", (long long)A);
    const size_t lds = (size_t)A * 8 * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)overlap_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int rows_per_block = CBLOCK / 64;
    unsigned gy = (unsigned)std::min<int64_t>((A + rows_per_block - 1) / rows_per_block, 64);
    // the scene index rides on grid.x (2^31 - 1 blocks)
    hipLaunchKernelGGL(overlap_count_kernel, dim3((unsigned)B, gy), dim3(CBLOCK), lds, (hipStream_t)stream, boxes, sc, present, out, (int)A);
    TDS_LAUNCH_CHECK("overlap_count_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_pairwise_overlap_f32(const float *box1, const float *sc1, const float *box2, const float *sc2, float *out,
                                        int64_t n, int metric, void *stream) {
    TDS_CHECK_ARG(n >= 0, "tds_pairwise_overlap_f32: negative n");
    TDS_CHECK_ARG(metric == TDS_METRIC_IOU || metric == TDS_METRIC_DISCS, "tds_pairwise_overlap_f32: unknown metric %d", metric);
    if (n == 0) return TDS_OK;
    TDS_CHECK_ARG(box1 && sc1 && box2 && sc2 && out, "tds_pairwise_overlap_f32: null pointer");
    dim3 grid((unsigned)((n + CBLOCK - 1) / CBLOCK));
    if (metric == TDS_METRIC_IOU)
        hipLaunchKernelGGL(pairwise_kernel<TDS_METRIC_IOU>, grid, dim3(CBLOCK), 0, (hipStream_t)stream, box1, sc1, box2, sc2, out, n);
    else
        hipLaunchKernelGGL(pairwise_kernel<TDS_METRIC_DISCS>, grid, dim3(CBLOCK), 0, (hipStream_t)stream, box1, sc1, box2, sc2, out, n);
    TDS_LAUNCH_CHECK("pairwise_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_pairwise_discs_f32(const float *box1, const float *sc1, const float *box2, const float *sc2, float *out, int64_t n,
                                      int num_discs, void *stream) {
    TDS_CHECK_ARG(n >= 0, "tds_pairwise_discs_f32: negative n");
    // torch.cdist switches to a matrix-multiply formulation above 25 points per set: the direct form restated here stops there
    TDS_CHECK_ARG(num_discs > 1 && (num_discs & 1) && num_discs <= 25, "tds_pairwise_discs_f32: num_discs must be odd, 3 .. 25 (got %d)", num_discs);
    if (n == 0) return TDS_OK;
    TDS_CHECK_ARG(box1 && sc1 && box2 && sc2 && out, "tds_pairwise_discs_f32: null pointer");
    hipLaunchKernelGGL(pairwise_discs_n_kernel, dim3((unsigned)((n + CBLOCK - 1) / CBLOCK)), dim3(CBLOCK), 0, (hipStream_t)stream, box1, sc1,
                       box2, sc2, out, n, (num_discs - 1) / 2);
    TDS_LAUNCH_CHECK("pairwise_discs_n_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_box2corners_f32(const float *box, const float *sc, float *corners, int64_t n, void *stream) {
    TDS_CHECK_ARG(n >= 0, "tds_box2corners_f32: negative n");
    if (n == 0) return TDS_OK;
    TDS_CHECK_ARG(box && sc && corners, "tds_box2corners_f32: null pointer");
    hipLaunchKernelGGL(box2corners_kernel, dim3((unsigned)((n + CBLOCK - 1) / CBLOCK)), dim3(CBLOCK), 0, (hipStream_t)stream, box, sc,
                       corners, n);
    TDS_LAUNCH_CHECK("box2corners_kernel");
    return TDS_OK;
}

// -----------------------------------------------------------------------------------------------------------------------------
// Occlusion mask of the standard sensing model (observation_noise.py:89-132, utils.line_circle_intersection :139-187):
// entity e is hidden from ego a when the disc (radius = width / 2) of another entity o touches the sight line a -> e.
// One wave per (scene, ego); lanes = targets, loop over occluders.  Same operation order as the reference, one rounding each.
// -----------------------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ bool line_circle(float p1x, float p1y, float p2x, float p2y, float cx, float cy, float r) {
    float dx = p2x - p1x, dy = p2y - p1y;
    float fx = p1x - cx, fy = p1y - cy;
    float a = dx * dx + dy * dy;
    float b = 2.0f * (fx * dx + fy * dy);
    float c = (fx * fx + fy * fy) - (r * r);
    float disc = b * b - (4.0f * a) * c;
    bool has = disc >= 0.0f;
    float sq = sqrtf(disc < 0.0f ? 0.0f : disc);
    if (disc != disc) sq = disc;
    float a_safe = (fabsf(a) < (float)1e-8) ? (float)1e-8 : a;
    float t1 = (-b - sq) / (2.0f * a_safe), t2 = (-b + sq) / (2.0f * a_safe);
    float tmin = t1 < t2 ? t1 : t2, tmax = t1 > t2 ? t1 : t2;
    if (t1 != t1 || t2 != t2) { tmin = tmax = t1 + t2; }
    return has && (tmin <= 1.0f) && (tmax >= 0.0f);
}

__global__ void __launch_bounds__(256) occlusion_kernel(const float4 *__restrict__ state, const float2 *__restrict__ size, const uint8_t *__restrict__ present,
                                                        uint8_t *__restrict__ out, int64_t n_ego, int A, int E) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);          // (scene, ego)
    if (w >= n_ego) return;
    const int64_t b = w / A;
    const int a = (int)(w - b * A);
    const float4 ego = state[b * E + a];
    for (int e0 = 0; e0 < E; e0 += 64) {
        const int e = e0 + lane;
        bool occluded = false;
        float4 tg = e < E ? state[b * E + e] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int o = 0; o < E; ++o) {                                         // wave-uniform
            const float4 oc = state[b * E + o];
            const float r = size[b * E + o].y / 2.0f;
            if (o != e && o != a && !occluded) occluded = line_circle(ego.x, ego.y, tg.x, tg.y, oc.x, oc.y, r);
        }
        if (e < E) out[(b * A + a) * E + e] = (uint8_t)((present[b * E + e] != 0) && !occluded);
    }
}

}  // namespace

TDS_EXPORT int tds_occlusion_mask_f32(const float *state, const float *size, const uint8_t *present, uint8_t *out, int64_t B, int64_t A,
                                      int64_t E, void *stream) {
    TDS_CHECK_ARG(B >= 0 && A >= 0 && E >= A && E < (1 << 20), "tds_occlusion_mask_f32: bad sizes");
    if (B * A == 0 || E == 0) return TDS_OK;
    TDS_CHECK_ARG(state && size && present && out, "tds_occlusion_mask_f32: null pointer");
    const int64_t n = B * A;
    hipLaunchKernelGGL(occlusion_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float4 *)state,
                       (const float2 *)size, present, out, n, (int)A, (int)E);
    TDS_LAUNCH_CHECK("occlusion_kernel");
    return TDS_OK;
}
