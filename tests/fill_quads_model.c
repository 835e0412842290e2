/* Test infrastructure: the exact ROW RULE for painting a PAIR of same-key triangles that share an edge (a triangulated quad of the road or
 * lane-marking mesh) as one, checked against the oracle's restatement of cv::fillConvexPoly (oracle/tds_oracle.c) called once per triangle,
 * as the reference does (rendering/cv2.py:44-59: one cv2.fillConvexPoly per face; equal keys, so the union does not depend on the order).
 *
 * Status (round 5).  The rendering grid pairs such faces (torchdrivesim_amd/csrc/map.hip, tds_common.h: QuadEntry) and the scan kernel of
 * the split form fetches, projects and trims a pair once.  The RASTERISER still paints triangle by triangle: this rule was built into
 * process_batch_bits in two organisations ("hull of both triangles' intervals" and the leaner "master + third chain" of model_pair_x
 * below), was bit-exact against the oracle on every GPU parity test, and LOST on the clock -- the row items of a pair need twice the
 * cross-lane traffic and the cut rows, which costs more than the halved number of rows gives back (uint8 256 x 256, one box: 5.09 ms
 * triangle by triangle, 5.75 paired; DESIGN_HISTORY.md section 4).  The model stays as the proof of the rule for whoever takes it up again.
 *
 * What is being proved.  tests/fill_rows_model.c proves the per-triangle row rule: in every row the painted pixels of a triangle are ONE
 * interval, the hull of the row ends of its active 16.16 edge chains (outline edges inside the image merged into the rows).  For a pair,
 * the INTERIOR rows -- the rows strictly between two consecutive vertex rows of the four points -- can be painted once, as the hull of
 * both triangles' intervals (one ds_or per row instead of two, one work item instead of two).  That is the union of the two fills iff the
 * two intervals overlap or touch in every such row.  They do whenever both triangles contain the row's pixels of the shared edge
 * ("diagonal") d: its chain has the same end points, hence the same slope, class and offsets, in both triangles.  Both triangles are
 * active in an interior row outside d's rows only if both apexes lie strictly above d's top row or both strictly below its bottom row:
 * such a pair ("apexes on one side") has to be painted triangle by triangle.  What stays per triangle: the exact walk of the edges that
 * are not merged, the vertex rows, and a triangle's own interval in the row of the OTHER triangle's apex when that row is interior
 * to it (it is no vertex row of its own, and the pair's items skip all four vertex rows).  "kernel-" modes: the same pixels organised as
 * two lanes would compute them (model_pair_x).
 *
 * Build + run: see tests/test_fill_quads_model.py.  Exit status 0 = no differing pixel. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern void orc_fill_convex_poly(float *img, int W, int H, const int32_t *pts, int npts, const float *col);
extern void orc_line_px(float *img, int W, int H, int ax, int ay, int bx, int by, const float *col);

enum { DY_BIAS_MAX = 100, DY_NOBIAS_MAX = 147, BIAS = 160 };
static int g_always_pair = 0;      /* "always-..." modes: pairs with both apexes on one side of the diagonal are NOT excluded -- the check must fail */
static int g_skip_apex_row = 0;    /* "noapex-..." modes: a triangle's interval in the row of the other one's apex is left out -- the check must fail */

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int ctz(int v) { return __builtin_ctz((unsigned)v); }

static int edge_dx(int xs, int xe, int dy) {
    int64_t n = ((int64_t)(xe - xs) << 17) + dy;
    return (int)(n / (2 * (int64_t)dy));
}

typedef struct { int merge, bias, xmajor; } ecls;

static int g_W, g_H;
static ecls classify(int ax, int ay, int bx, int by) {
    ecls c = {0, 0, 0};
    const int W = g_W, H = g_H;
    const int inside = (unsigned)ax < (unsigned)W && (unsigned)bx < (unsigned)W && (unsigned)ay < (unsigned)H && (unsigned)by < (unsigned)H;
    if (!inside) return c;
    const int adx = abs(bx - ax), ady = abs(by - ay);
    c.xmajor = adx >= ady;
    if (ady <= DY_BIAS_MAX) { c.merge = 1; c.bias = BIAS; return c; }
    if (ady <= DY_NOBIAS_MAX) {
        const int tiefree = c.xmajor ? (ctz(adx) <= ctz(ady)) : (adx == 0 || ctz(adx) >= ctz(ady));
        if (tiefree) { c.merge = 1; c.bias = 0; }
    }
    return c;
}

static void offsets(ecls c, int s, int *offL, int *offR) {
    *offL = *offR = 32768;
    if (!c.merge) return;
    if (!c.xmajor) { *offL = 32768 - c.bias; return; }
    const int h = abs(s) >> 1;
    *offL = imin(32768, 65536 - h + c.bias);
    *offR = imax(32768, h + c.bias);
}

static void reach(ecls c, int x0, int d, int *L, int *R) {
    if (!c.merge) return;
    *L = imin(*L, x0); *R = imax(*R, x0);
    if (!c.xmajor) return;
    const int v = (int)((((int64_t)x0 << 16) + d + c.bias) >> 16);
    if (d >= 0) *R = imax(*R, v); else *L = imin(*L, v + 1);
}

static uint8_t *g_mask;
static void paint(int y, int L, int R) {
    if (y < 0 || y >= g_H) return;
    L = imax(L, 0); R = imin(R, g_W - 1);
    for (int x = L; x <= R; ++x) g_mask[y * g_W + x] = 1;
}

static float *g_line_img;
static void line_exact(int ax, int ay, int bx, int by) {
    const float one[3] = {1, 1, 1};
    orc_line_px(g_line_img, g_W, g_H, ax, ay, bx, by, one);
}

static int half_of(int s) { return s >= 0 ? (s >> 1) : -((-s) >> 1); }

/* ---- one triangle, as process_batch_bits sets it up (tests/fill_rows_model.c: model_fill, cut into its stages) ---- */
typedef struct {
    int hit;                         /* the bounding box meets the image */
    int xt, yt, xm, ym, xb, yb;      /* vertices by row */
    ecls cTM, cMB, cTB;
    int sTB, sTM, sMB;
    int oL_TB, oR_TB, oL_TM, oR_TM, oL_MB, oR_MB;
} tri;

static void tri_setup(tri *t, const int32_t *pts) {
    int x[3] = {pts[0], pts[2], pts[4]}, y[3] = {pts[1], pts[3], pts[5]};
    const int ea[3] = {2, 0, 1}, eb[3] = {0, 1, 2};
    for (int l = 0; l < 3; ++l) {                                       /* edges that are not merged: the exact walk (the kernel's edge ring) */
        ecls c = classify(x[ea[l]], y[ea[l]], x[eb[l]], y[eb[l]]);
        if (!c.merge) line_exact(x[ea[l]], y[ea[l]], x[eb[l]], y[eb[l]]);
    }
    const int xmin = imin(x[0], imin(x[1], x[2])), xmax = imax(x[0], imax(x[1], x[2]));
    const int ymin = imin(y[0], imin(y[1], y[2])), ymax = imax(y[0], imax(y[1], y[2]));
    t->hit = !(xmax < 0 || ymax < 0 || xmin >= g_W || ymin >= g_H);
    /* by (row, column), like the kernel's sort of the packed vertices */
    int o[3] = {0, 1, 2};
    for (int i = 1; i < 3; ++i)
        for (int j = i; j > 0 && (y[o[j - 1]] > y[o[j]] || (y[o[j - 1]] == y[o[j]] && x[o[j - 1]] > x[o[j]])); --j) { int s = o[j]; o[j] = o[j - 1]; o[j - 1] = s; }
    t->xt = x[o[0]]; t->yt = y[o[0]]; t->xm = x[o[1]]; t->ym = y[o[1]]; t->xb = x[o[2]]; t->yb = y[o[2]];
    t->cTM = classify(t->xt, t->yt, t->xm, t->ym); t->cMB = classify(t->xm, t->ym, t->xb, t->yb); t->cTB = classify(t->xt, t->yt, t->xb, t->yb);
    t->sTB = t->sTM = t->sMB = 0;
    if (t->yt < t->yb) {
        t->sTB = edge_dx(t->xt, t->xb, t->yb - t->yt);
        t->sTM = t->ym > t->yt ? edge_dx(t->xt, t->xm, t->ym - t->yt) : 0;
        t->sMB = t->yb > t->ym ? edge_dx(t->xm, t->xb, t->yb - t->ym) : 0;
    }
    offsets(t->cTB, t->sTB, &t->oL_TB, &t->oR_TB); offsets(t->cTM, t->sTM, &t->oL_TM, &t->oR_TM); offsets(t->cMB, t->sMB, &t->oL_MB, &t->oR_MB);
}

/* the rows of the triangle's own vertices */
static void tri_vertex_rows(const tri *t) {
    if (!t->hit) return;
    const int xt = t->xt, yt = t->yt, xm = t->xm, ym = t->ym, xb = t->xb, yb = t->yb;
    if (yt == yb) {
        int L = 0x7fffffff, R = -0x7fffffff;
        if (t->cTM.merge) { L = imin(L, imin(xt, xm)); R = imax(R, imax(xt, xm)); }
        if (t->cMB.merge) { L = imin(L, imin(xm, xb)); R = imax(R, imax(xm, xb)); }
        if (t->cTB.merge) { L = imin(L, imin(xt, xb)); R = imax(R, imax(xt, xb)); }
        if (L <= R) paint(yt, L, R);
        return;
    }
    {
        int L, R;
        if (ym > yt) { L = R = xt; reach(t->cTM, xt, half_of(t->sTM), &L, &R); }
        else { L = imin(xt, xm); R = imax(xt, xm); reach(t->cMB, xm, half_of(t->sMB), &L, &R); }
        reach(t->cTB, xt, half_of(t->sTB), &L, &R);
        paint(yt, L, R);
    }
    if (ym > yt && ym < yb) {
        const int64_t xc = ((int64_t)xt << 16) + (int64_t)(ym - yt) * t->sTB;
        int L = imin(xm, (int)((xc + t->oL_TB) >> 16)), R = imax(xm, (int)((xc + t->oR_TB) >> 16));
        reach(t->cTM, xm, -half_of(t->sTM), &L, &R);
        reach(t->cMB, xm, half_of(t->sMB), &L, &R);
        paint(ym, L, R);
    }
    {
        int L = 0x7fffffff, R = -0x7fffffff;
        reach(t->cTB, xb, -half_of(t->sTB), &L, &R);
        if (ym < yb) reach(t->cMB, xb, -half_of(t->sMB), &L, &R);
        else {
            reach(t->cTM, xm, -half_of(t->sTM), &L, &R);
            if (t->cMB.merge) { L = imin(L, imin(xm, xb)); R = imax(R, imax(xm, xb)); }
        }
        if (L <= R) paint(yb, L, R);
    }
}

/* an INTERIOR row of the triangle (strictly between its top and bottom rows, not the row of its middle vertex): -> 1 and the interval
 * [L, R] = the hull of the row ends of its two active chains (before the cut to the image); 0: the triangle has nothing of its own there */
static int tri_row(const tri *t, int y, int *L, int *R) {
    if (!t->hit || !(t->yt < y && y < t->yb) || y == t->ym) return 0;
    const int64_t xc = ((int64_t)t->xt << 16) + (int64_t)(y - t->yt) * t->sTB;
    int64_t xa;
    int oLa, oRa;
    if (y < t->ym) { xa = ((int64_t)t->xt << 16) + (int64_t)(y - t->yt) * t->sTM; oLa = t->oL_TM; oRa = t->oR_TM; }
    else { xa = ((int64_t)t->xm << 16) + (int64_t)(y - t->ym) * t->sMB; oLa = t->oL_MB; oRa = t->oR_MB; }
    *L = imin((int)((xa + oLa) >> 16), (int)((xc + t->oL_TB) >> 16));
    *R = imax((int)((xa + oRa) >> 16), (int)((xc + t->oR_TB) >> 16));
    return 1;
}

static void tri_interior_rows(const tri *t) {
    for (int y = imax(t->yt + 1, 0); y < t->yb && y < g_H; ++y) {
        int L, R;
        if (tri_row(t, y, &L, &R)) paint(y, L, R);
    }
}

static long g_pairs_merged, g_pairs_split;

/* quad = four points; T1 = (P0, P1, P2), T2 = (P[b0], P[b1], P[b2]) with P3 its apex; a1 = the vertex of T1 that T2 does not have */
static void model_pair(const int32_t *P, const int *b, int a1) {
    int32_t p1[6] = {P[0], P[1], P[2], P[3], P[4], P[5]}, p2[6];
    for (int i = 0; i < 3; ++i) { p2[2 * i] = P[2 * b[i]]; p2[2 * i + 1] = P[2 * b[i] + 1]; }
    tri t1, t2;
    tri_setup(&t1, p1); tri_setup(&t2, p2);
    tri_vertex_rows(&t1); tri_vertex_rows(&t2);
    /* the rows of the shared edge and of the two apexes */
    const int s0 = a1 == 0 ? 1 : 0, s1 = a1 == 2 ? 1 : 2;
    const int ydT = imin(P[2 * s0 + 1], P[2 * s1 + 1]), ydB = imax(P[2 * s0 + 1], P[2 * s1 + 1]);
    const int ya1 = P[2 * a1 + 1], ya2 = P[7];
    const int one_side = (ya1 < ydT && ya2 < ydT) || (ya1 > ydB && ya2 > ydB);
    if (one_side && !g_always_pair) {                   /* painted triangle by triangle */
        ++g_pairs_split;
        tri_interior_rows(&t1); tri_interior_rows(&t2);
        return;
    }
    ++g_pairs_merged;
    /* a triangle's own interval in the row of the other one's apex, when that row is interior to it */
    if (!g_skip_apex_row) {
        int L, R;
        if (tri_row(&t1, ya2, &L, &R)) paint(ya2, L, R);
        if (tri_row(&t2, ya1, &L, &R)) paint(ya1, L, R);
    }
    /* the four vertex rows by height; the interior rows of the pair lie strictly between two consecutive ones */
    const int lo = imin(ya1, ya2), hi = imax(ya1, ya2);
    const int m1 = imax(lo, ydT), m2 = imin(hi, ydB);
    const int lev[4] = {imin(lo, ydT), imin(m1, m2), imax(m1, m2), imax(hi, ydB)};
    for (int k = 0; k < 3; ++k)
        for (int y = imax(lev[k] + 1, 0); y < lev[k + 1] && y < g_H; ++y) {
            int L1, R1, L2, R2;
            const int a = tri_row(&t1, y, &L1, &R1), c = tri_row(&t2, y, &L2, &R2);
            if (a && c) paint(y, imin(L1, L2), imax(R1, R2));          /* ONE span: the hull */
            else if (a) paint(y, L1, R1);
            else if (c) paint(y, L2, R2);
        }
}

/* ---- the same pair as a KERNEL would organise it (round 5 built it into raster.hip's process_batch_bits): two lanes, one per triangle, each with its own set-up.
 * The lane of T1 is the pair's MASTER: it owns every row strictly inside T1 -- its own two chains plus, where T2 has rows of its own, T2's
 * chain that is not the shared edge (the "third chain"; the shared edge is one of the master's own two there) -- in up to three parts cut at
 * the rows of T1's middle vertex and of T2's apex.  A part that starts at such a row starts WITH it: at the row of T1's middle vertex its own
 * short chain is left out (that row is T1's vertex row, painted by its lane; what remains -- the long chain, i.e. the shared edge, and the
 * third chain -- is T2's interval there), at the row of T2's apex the third chain is left out (T2's vertex row, painted by its lane).  The lane
 * of T2 owns the rows of T2 beyond T1's (at most one run, on the side of its apex), as a lone triangle would. */
static int chain_iv(int xv, int yv, int s, int oL, int oR, int y, int *L, int *R) {
    const int64_t x = ((int64_t)xv << 16) + (int64_t)(y - yv) * s;
    *L = (int)((x + oL) >> 16); *R = (int)((x + oR) >> 16);
    return 1;
}
static void model_pair_x(const int32_t *P, const int *b, int a1) {
    int32_t p1[6] = {P[0], P[1], P[2], P[3], P[4], P[5]}, p2[6];
    for (int i = 0; i < 3; ++i) { p2[2 * i] = P[2 * b[i]]; p2[2 * i + 1] = P[2 * b[i] + 1]; }
    tri t1, t2;
    tri_setup(&t1, p1); tri_setup(&t2, p2);
    const int s0 = a1 == 0 ? 1 : 0, s1 = a1 == 2 ? 1 : 2;
    const int ydT = imin(P[2 * s0 + 1], P[2 * s1 + 1]), ydB = imax(P[2 * s0 + 1], P[2 * s1 + 1]);
    const int ya1 = P[2 * a1 + 1], ya2 = P[7];
    const int one_side = (ya1 < ydT && ya2 < ydT) || (ya1 > ydB && ya2 > ydB);
    if (one_side) { ++g_pairs_split; tri_vertex_rows(&t1); tri_vertex_rows(&t2); tri_interior_rows(&t1); tri_interior_rows(&t2); return; }
    ++g_pairs_merged;
    /* a pair is set up as a whole: both triangles when the bounding box of the four points meets the image */
    t1.hit = t2.hit = t1.hit || t2.hit;
    tri_vertex_rows(&t1); tri_vertex_rows(&t2);
    if (!t1.hit) return;
    /* where T2's apex sits among T2's vertices by (row, column): 0 top, 1 middle, 2 bottom (the kernel compares packed vertices) */
    const int xa2 = P[6];
    const int apos = (xa2 == t2.xt && ya2 == t2.yt) ? 0 : ((xa2 == t2.xm && ya2 == t2.ym) ? 1 : 2);
    const int yt = t1.yt, ym = t1.ym, yb = t1.yb, yt2 = t2.yt, ym2 = t2.ym, yb2 = t2.yb;
    /* master (the lane of T1): every row strictly inside T1 */
    for (int y = imax(yt + 1, 0); y < yb && y < g_H; ++y) {
        const int third = yt2 < y && y < yb2 && y != ym2;                       /* T2 has a row of its own here */
        const int own_short = y != ym, own_long = y != ym || third;             /* T1's middle vertex row: only what stands for T2's interval */
        int L = 0x7fffffff, R = -0x7fffffff, l, r;
        if (own_long) { chain_iv(t1.xt, t1.yt, t1.sTB, t1.oL_TB, t1.oR_TB, y, &l, &r); L = imin(L, l); R = imax(R, r); }
        if (own_short) {
            if (y < ym) chain_iv(t1.xt, t1.yt, t1.sTM, t1.oL_TM, t1.oR_TM, y, &l, &r);
            else chain_iv(t1.xm, t1.ym, t1.sMB, t1.oL_MB, t1.oR_MB, y, &l, &r);
            L = imin(L, l); R = imax(R, r);
        }
        if (third) {
            /* T2's chain that is not the shared edge: below the row of T2's apex the edge apex -> bottom end of the shared edge, above it
             * top end of the shared edge -> apex; in terms of T2's vertices by row: */
            const int dn = apos == 0 ? 1 : (apos == 1 ? y > ym2 : 0);
            if (dn) { if (apos == 0) chain_iv(t2.xt, t2.yt, t2.sTB, t2.oL_TB, t2.oR_TB, y, &l, &r); else chain_iv(t2.xm, t2.ym, t2.sMB, t2.oL_MB, t2.oR_MB, y, &l, &r); }
            else { if (apos == 2) chain_iv(t2.xt, t2.yt, t2.sTB, t2.oL_TB, t2.oR_TB, y, &l, &r); else chain_iv(t2.xt, t2.yt, t2.sTM, t2.oL_TM, t2.oR_TM, y, &l, &r); }
            L = imin(L, l); R = imax(R, r);
        }
        if (L <= R) paint(y, L, R);
    }
    /* the lane of T2: the rows of T2 strictly beyond T1's rows, as a lone triangle paints them */
    for (int y = imax(yt2 + 1, 0); y < yb2 && y < g_H; ++y) {
        int L, R;
        if ((y < yt || y > yb) && tri_row(&t2, y, &L, &R)) paint(y, L, R);
    }
}
static int g_kernel_form = 0;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }
static int rnd_range(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }

static long g_checked, g_bad;
static void check(const int32_t *P, const int *b, int a1, float *img) {
    const int W = g_W, H = g_H;
    const float one[3] = {1, 1, 1};
    int bx0 = W, bx1 = -1, by0 = H, by1 = -1;
    for (int i = 0; i < 4; ++i) { bx0 = imin(bx0, P[2 * i]); bx1 = imax(bx1, P[2 * i]); by0 = imin(by0, P[2 * i + 1]); by1 = imax(by1, P[2 * i + 1]); }
    bx0 = imax(bx0 - 2, 0); bx1 = imin(bx1 + 2, W - 1); by0 = imax(by0 - 2, 0); by1 = imin(by1 + 2, H - 1);
    if ((g_checked & 4095) == 0) { bx0 = by0 = 0; bx1 = W - 1; by1 = H - 1; }
    int32_t p2[6];
    for (int i = 0; i < 3; ++i) { p2[2 * i] = P[2 * b[i]]; p2[2 * i + 1] = P[2 * b[i] + 1]; }
    orc_fill_convex_poly(img, W, H, P, 3, one);                        /* the reference: one call per face */
    orc_fill_convex_poly(img, W, H, p2, 3, one);
    if (g_kernel_form) model_pair_x(P, b, a1); else model_pair(P, b, a1);
    ++g_checked;
    int bad = 0;
    for (int yy = by0; yy <= by1 && !bad; ++yy) for (int xx = bx0; xx <= bx1 && !bad; ++xx) {
        const int i = yy * W + xx;
        const int ref = img[3 * i] != 0.0f, got = g_mask[i] || g_line_img[3 * i] != 0.0f;
        if (ref != got) {
            if (g_bad < 10) fprintf(stderr, "differs at (x %d, y %d): oracle %d model %d for P (%d,%d) (%d,%d) (%d,%d) (%d,%d), T2 = (%d,%d,%d), apex of T1 %d, in %dx%d\n",
                                    xx, yy, ref, got, P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7], b[0], b[1], b[2], a1, W, H);
            ++g_bad;
            bad = 1;
        }
    }
    for (int yy = by0; yy <= by1 && bx0 <= bx1; ++yy) {
        memset(img + 3 * (yy * W + bx0), 0, sizeof(float) * 3 * (bx1 - bx0 + 1));
        memset(g_line_img + 3 * (yy * W + bx0), 0, sizeof(float) * 3 * (bx1 - bx0 + 1));
        memset(g_mask + yy * W + bx0, 0, (size_t)(bx1 - bx0 + 1));
    }
}

/* T2 over the shared edge opposite vertex a1 of T1, its three vertices in the order number `perm` (0..5) */
static void make_t2(int a1, int perm, int *b) {
    const int s0 = a1 == 0 ? 1 : 0, s1 = a1 == 2 ? 1 : 2;
    const int v[3] = {s0, s1, 3};
    static const int pm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    for (int i = 0; i < 3; ++i) b[i] = v[pm[perm][i]];
}

int main(int argc, char **argv) {
    /* usage: fill_quads_model [always-|noapex-]exhaustive RES LO HI [A1 PERM] | [..]random RES COUNT SEED | one RES x0 y0 .. x3 y3 A1 PERM */
    if (argc < 5) { fprintf(stderr, "usage: %s exhaustive RES LO HI [A1 PERM] | random RES COUNT SEED | one RES 8 coords A1 PERM\n", argv[0]); return 2; }
    if (!strncmp(argv[1], "always-", 7)) { g_always_pair = 1; argv[1] += 7; }
    if (!strncmp(argv[1], "noapex-", 7)) { g_skip_apex_row = 1; argv[1] += 7; }
    if (!strncmp(argv[1], "kernel-", 7)) { g_kernel_form = 1; argv[1] += 7; }       /* the pair as the kernel's two lanes organise it */
    g_W = g_H = atoi(argv[2]);
    float *img = (float *)calloc((size_t)3 * g_W * g_H, sizeof(float));
    g_line_img = (float *)calloc((size_t)3 * g_W * g_H, sizeof(float));
    g_mask = (uint8_t *)calloc((size_t)g_W * g_H, 1);
    int b[3];
    if (!strcmp(argv[1], "one")) {
        int32_t P[8];
        for (int i = 0; i < 8; ++i) P[i] = atoi(argv[3 + i]);
        const int a1 = atoi(argv[11]);
        make_t2(a1, atoi(argv[12]), b);
        check(P, b, a1, img);
    } else if (!strcmp(argv[1], "exhaustive")) {
        const int lo = atoi(argv[3]), hi = atoi(argv[4]);
        /* every position of the four points; which edge of T1 is shared and the vertex order of T2: all 18, or the one given */
        const int a_lo = argc > 6 ? atoi(argv[5]) : 0, a_hi = argc > 6 ? atoi(argv[5]) : 2;
        const int p_lo = argc > 6 ? atoi(argv[6]) : 0, p_hi = argc > 6 ? atoi(argv[6]) : 5;
        int32_t P[8];
        for (P[0] = lo; P[0] <= hi; ++P[0]) for (P[1] = lo; P[1] <= hi; ++P[1])
        for (P[2] = lo; P[2] <= hi; ++P[2]) for (P[3] = lo; P[3] <= hi; ++P[3])
        for (P[4] = lo; P[4] <= hi; ++P[4]) for (P[5] = lo; P[5] <= hi; ++P[5])
        for (P[6] = lo; P[6] <= hi; ++P[6]) for (P[7] = lo; P[7] <= hi; ++P[7])
            for (int a1 = a_lo; a1 <= a_hi; ++a1) for (int perm = p_lo; perm <= p_hi; ++perm) { make_t2(a1, perm, b); check(P, b, a1, img); }
    } else {
        const long count = atol(argv[3]);
        rng_state ^= (uint64_t)atol(argv[4]) * 0x2545F4914F6CDD1Dull;
        const int R = g_W;
        for (long n = 0; n < count; ++n) {
            int32_t P[8];
            const int kind = (int)(rnd() % 8);
            if (kind == 0) {                                           /* anywhere, also well outside */
                for (int i = 0; i < 8; ++i) P[i] = rnd_range(-R / 2, R + R / 2);
            } else if (kind == 1) {                                    /* inside the image */
                for (int i = 0; i < 8; ++i) P[i] = rnd_range(0, R - 1);
            } else if (kind == 2 || kind == 3 || kind == 4) {
                /* a lane marking / road quad: a long thin (2) or wide (3) parallelogram with jitter, any rotation; (4) across the image border */
                const int cx = kind == 4 ? rnd_range(-R / 8, R + R / 8) : rnd_range(R / 8, R - R / 8), cy = kind == 4 ? rnd_range(-R / 8, R + R / 8) : rnd_range(R / 8, R - R / 8);
                const int len = kind == 2 ? R / 6 : R / 3;
                const int ux = rnd_range(-len, len), uy = rnd_range(-len, len);
                const int wd = kind == 2 ? 2 : R / 10 + 1;
                const int vx = rnd_range(-wd, wd), vy = rnd_range(-wd, wd);
                const int q[8] = {cx, cy, cx + ux, cy + uy, cx + ux + vx, cy + uy + vy, cx + vx, cy + vy};        /* ring order */
                /* T1 = (r0, r1, r3), apex of T2 = r2: diagonal r1 - r3 */
                P[0] = q[0]; P[1] = q[1]; P[2] = q[2]; P[3] = q[3]; P[4] = q[6]; P[5] = q[7]; P[6] = q[4]; P[7] = q[5];
                for (int i = 0; i < 8; ++i) P[i] += rnd_range(-1, 1) * (int)(rnd() % 3 == 0);
            } else if (kind == 5) {                                    /* small quads */
                P[0] = rnd_range(-4, R + 4); P[1] = rnd_range(-4, R + 4);
                for (int i = 2; i < 8; ++i) P[i] = P[i & 1] + rnd_range(-6, 6);
            } else if (kind == 6) {                                    /* degenerate: coinciding and collinear points, flat tops and bottoms */
                P[0] = rnd_range(0, R - 1); P[1] = rnd_range(0, R - 1);
                for (int i = 2; i < 8; ++i) P[i] = P[i & 1] + rnd_range(-1, 1) * rnd_range(0, R / 4);
                if (rnd() & 1) { P[6] = P[2 * (int)(rnd() % 3)]; }
                if (rnd() & 1) { P[7] = P[2 * (int)(rnd() % 3) + 1]; }
            } else {                                                   /* long edges beyond the merge limits (exact walk), partly outside */
                for (int i = 0; i < 8; ++i) P[i] = rnd_range(-R / 4, R + R / 4);
                P[4] = P[0] + rnd_range(-3, 3); P[6] = P[2] + rnd_range(-3, 3);
            }
            /* any vertex of T1 as its apex, any vertex order of T2; T1's own order rotated along */
            const int a1 = (int)(rnd() % 3), perm = (int)(rnd() % 6), rot = (int)(rnd() % 3);
            int32_t Q[8];
            for (int i = 0; i < 3; ++i) { const int k = (i + rot) % 3; Q[2 * i] = P[2 * k]; Q[2 * i + 1] = P[2 * k + 1]; }
            Q[6] = P[6]; Q[7] = P[7];
            make_t2(a1, perm, b);
            check(Q, b, a1, img);
        }
    }
    printf("%ld pairs, %ld differ (%ld painted as one, %ld triangle by triangle)\n", g_checked, g_bad, g_pairs_merged, g_pairs_split);
    return g_bad ? 1 : 0;
}
