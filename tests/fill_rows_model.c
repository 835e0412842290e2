/* Test infrastructure: a sequential C model of how the K3 bit-plane kernel (torchdrivesim_amd/csrc/raster.hip, process_batch_bits)
 * paints one triangle, checked here against the oracle's restatement of cv::fillConvexPoly (oracle/tds_oracle.c) triangle by triangle.
 *
 * What is being proved.  cv::fillConvexPoly paints  Line(v2,v0) + Line(v0,v1) + Line(v1,v2) + the scan-converted rows.  The kernel does not
 * draw the outline of an edge that lies inside the image (no clipLine) pixel by pixel: per ROW, the pixels of that edge and the scan-converted
 * span of the row are ONE run of pixels, and its two ends follow from the 16.16 edge chain of the scan conversion by an add and a shift:
 *   y-major edge (|dy| > |dx|): cv::Line paints  x(tau) = x0 + dx tau / |dy|  rounded to nearest, a tie going to the LEFT pixel; the span
 *     ends at floor(chain + 1/2).  They differ only in tie rows, by one pixel, so the left end becomes floor(chain + 1/2 - BIAS);
 *   x-major edge: row tau holds the pixels x with  x(tau - 1/2) < x <= x(tau + 1/2)  (cut at the edge's end points), i.e.
 *     floor(chain - slope/2 + BIAS) + 1 .. floor(chain + slope/2 + BIAS).
 * BIAS must exceed the accumulated rounding of the chain (half a unit of 2^-16 per row) and, together with it, stay below the distance 1/(2|dy|)
 * of a non-tie from a tie.  The chain's slope is trunc(q + 1/2) units of 2^-16: within half a unit of q >= 0 but between 1/2 and 3/2 units above
 * q < 0, per row -- so BIAS = 160 units serves |dy| <= 100.  Longer edges without ties (bias 0, |dy| <= 147), and everything else, as before:
 * edges that need clipping or are too long are drawn by the exact walk.  The rows of the three vertices are painted separately (the runs are cut
 * at the end points there); the rows in between are the kernel's work items.
 *
 * Build + run: see tests/test_fill_rows_model.py.  Exit status 0 = no differing pixel. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern void orc_fill_convex_poly(float *img, int W, int H, const int32_t *pts, int npts, const float *col);
extern void orc_line_px(float *img, int W, int H, int ax, int ay, int bx, int by, const float *col);

enum { DY_BIAS_MAX = 100, DY_NOBIAS_MAX = 147 };
static int BIAS = 160;            /* "nobias-..." modes set it to 0: the check must then find differing pixels (tie rows) */

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int ctz(int v) { return __builtin_ctz((unsigned)v); }

/* OpenCV's 16.16 edge slope: ((xe - xs) * 2 + dy) / (2 * dy) on 16.16 operands, C division */
static int edge_dx(int xs, int xe, int dy) {
    int64_t n = ((int64_t)(xe - xs) << 17) + dy;
    return (int)(n / (2 * (int64_t)dy));
}

typedef struct { int merge, bias, xmajor; } ecls;

static ecls classify(int ax, int ay, int bx, int by, int W, int H) {
    ecls c = {0, 0, 0};
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

/* offsets of the row ends for a chain that follows an edge of class c with 16.16 slope s: left = (x + offL) >> 16, right = (x + offR) >> 16 */
static void offsets(ecls c, int s, int *offL, int *offR) {
    *offL = *offR = 32768;
    if (!c.merge) return;
    if (!c.xmajor) { *offL = 32768 - c.bias; return; }
    const int h = abs(s) >> 1;
    *offL = imin(32768, 65536 - h + c.bias);
    *offR = imax(32768, h + c.bias);
}

/* the pixels of an x-major edge in the row of one of its end points (x0): from x0 towards x0 + d, d = half a row's advance (16.16, signed) */
static void reach(ecls c, int x0, int d, int *L, int *R) {
    if (!c.merge) return;
    *L = imin(*L, x0); *R = imax(*R, x0);
    if (!c.xmajor) return;
    const int v = (int)((((int64_t)x0 << 16) + d + c.bias) >> 16);
    if (d >= 0) *R = imax(*R, v); else *L = imin(*L, v + 1);
}

static uint8_t *g_mask; static int g_W, g_H;
static void paint(int y, int L, int R) {
    if (y < 0 || y >= g_H) return;
    L = imax(L, 0); R = imin(R, g_W - 1);
    for (int x = L; x <= R; ++x) g_mask[y * g_W + x] = 1;
}

static float *g_line_img;            /* scratch image for the exact walk of the edges that are not merged */
static void line_exact(int ax, int ay, int bx, int by) {
    const float one[3] = {1, 1, 1};
    orc_line_px(g_line_img, g_W, g_H, ax, ay, bx, by, one);      /* directed: clipLine depends on the order of the end points */
}

static int half_of(int s) { return s >= 0 ? (s >> 1) : -((-s) >> 1); }     /* signed half slope, magnitude rounded down */

static void model_fill(const int32_t *pts) {
    const int W = g_W, H = g_H;
    int x[3] = {pts[0], pts[2], pts[4]}, y[3] = {pts[1], pts[3], pts[5]};
    /* edges that are not merged: the exact walk (the kernel's edge ring) */
    const int ea[3] = {2, 0, 1}, eb[3] = {0, 1, 2};
    for (int l = 0; l < 3; ++l) {
        ecls c = classify(x[ea[l]], y[ea[l]], x[eb[l]], y[eb[l]], W, H);
        if (!c.merge) line_exact(x[ea[l]], y[ea[l]], x[eb[l]], y[eb[l]]);
    }
    int xmin = imin(x[0], imin(x[1], x[2])), xmax = imax(x[0], imax(x[1], x[2]));
    int ymin = imin(y[0], imin(y[1], y[2])), ymax = imax(y[0], imax(y[1], y[2]));
    if (xmax < 0 || ymax < 0 || xmin >= W || ymin >= H) return;
    /* T, M, B: the vertices by row (any order among equals) */
    int o[3] = {0, 1, 2};
    for (int i = 1; i < 3; ++i) for (int j = i; j > 0 && y[o[j - 1]] > y[o[j]]; --j) { int t = o[j]; o[j] = o[j - 1]; o[j - 1] = t; }
    const int xt = x[o[0]], yt = y[o[0]], xm = x[o[1]], ym = y[o[1]], xb = x[o[2]], yb = y[o[2]];
    const ecls cTM = classify(xt, yt, xm, ym, W, H), cMB = classify(xm, ym, xb, yb, W, H), cTB = classify(xt, yt, xb, yb, W, H);
    if (yt == yb) {                                                    /* one row: three horizontal edges, nothing is scan-converted */
        int L = 0x7fffffff, R = -0x7fffffff;
        if (cTM.merge) { L = imin(L, imin(xt, xm)); R = imax(R, imax(xt, xm)); }
        if (cMB.merge) { L = imin(L, imin(xm, xb)); R = imax(R, imax(xm, xb)); }
        if (cTB.merge) { L = imin(L, imin(xt, xb)); R = imax(R, imax(xt, xb)); }
        if (L <= R) paint(yt, L, R);
        return;
    }
    const int sTB = edge_dx(xt, xb, yb - yt);
    const int sTM = ym > yt ? edge_dx(xt, xm, ym - yt) : 0;
    const int sMB = yb > ym ? edge_dx(xm, xb, yb - ym) : 0;
    int oL_TB, oR_TB, oL_TM, oR_TM, oL_MB, oR_MB;
    offsets(cTB, sTB, &oL_TB, &oR_TB); offsets(cTM, sTM, &oL_TM, &oR_TM); offsets(cMB, sMB, &oL_MB, &oR_MB);
    /* ---- the row of the top vertex / vertices */
    {
        int L, R;
        if (ym > yt) { L = R = xt; reach(cTM, xt, half_of(sTM), &L, &R); }
        else { L = imin(xt, xm); R = imax(xt, xm); reach(cMB, xm, half_of(sMB), &L, &R); }
        reach(cTB, xt, half_of(sTB), &L, &R);
        paint(yt, L, R);
    }
    /* ---- rows strictly between the top and the middle vertex: chains T->M and T->B */
    for (int yy = imax(yt + 1, 0); yy < ym && yy < H; ++yy) {
        const int64_t xa = ((int64_t)xt << 16) + (int64_t)(yy - yt) * sTM, xc = ((int64_t)xt << 16) + (int64_t)(yy - yt) * sTB;
        const int L = (int)imin((int)((xa + oL_TM) >> 16), (int)((xc + oL_TB) >> 16)), R = imax((int)((xa + oR_TM) >> 16), (int)((xc + oR_TB) >> 16));
        paint(yy, L, R);
    }
    /* ---- the row of the middle vertex when it lies strictly between the others */
    if (ym > yt && ym < yb) {
        const int64_t xc = ((int64_t)xt << 16) + (int64_t)(ym - yt) * sTB;
        int L = imin(xm, (int)((xc + oL_TB) >> 16)), R = imax(xm, (int)((xc + oR_TB) >> 16));
        reach(cTM, xm, -half_of(sTM), &L, &R);
        reach(cMB, xm, half_of(sMB), &L, &R);
        paint(ym, L, R);
    }
    /* ---- rows strictly between the middle and the bottom vertex: chains M->B and T->B */
    for (int yy = imax(ym + 1, 0); yy < yb && yy < H; ++yy) {
        if (yy <= yt) continue;                                        /* flat top: the top row was painted above */
        const int64_t xa = ((int64_t)xm << 16) + (int64_t)(yy - ym) * sMB, xc = ((int64_t)xt << 16) + (int64_t)(yy - yt) * sTB;
        const int L = (int)imin((int)((xa + oL_MB) >> 16), (int)((xc + oL_TB) >> 16)), R = imax((int)((xa + oR_MB) >> 16), (int)((xc + oR_TB) >> 16));
        paint(yy, L, R);
    }
    /* ---- the row of the bottom vertex / vertices: never scan-converted, only the merged edges' pixels */
    {
        int L = 0x7fffffff, R = -0x7fffffff;
        reach(cTB, xb, -half_of(sTB), &L, &R);
        if (ym < yb) reach(cMB, xb, -half_of(sMB), &L, &R);
        else {
            reach(cTM, xm, -half_of(sTM), &L, &R);
            if (cMB.merge) { L = imin(L, imin(xm, xb)); R = imax(R, imax(xm, xb)); }     /* the horizontal bottom edge */
        }
        if (L <= R) paint(yb, L, R);
    }
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 16); }
static int rnd_range(int lo, int hi) { return lo + (int)(rnd() % (uint32_t)(hi - lo + 1)); }

static long g_checked, g_bad;
static void check(const int32_t *pts, float *img) {
    const int W = g_W, H = g_H;
    const float one[3] = {1, 1, 1};
    /* both sides paint inside the triangle's bounding box only (checked over the whole image every 4096th time) */
    int bx0 = imax(imin(pts[0], imin(pts[2], pts[4])) - 2, 0), bx1 = imin(imax(pts[0], imax(pts[2], pts[4])) + 2, W - 1);
    int by0 = imax(imin(pts[1], imin(pts[3], pts[5])) - 2, 0), by1 = imin(imax(pts[1], imax(pts[3], pts[5])) + 2, H - 1);
    if ((g_checked & 4095) == 0) { bx0 = by0 = 0; bx1 = W - 1; by1 = H - 1; }
    orc_fill_convex_poly(img, W, H, pts, 3, one);
    model_fill(pts);
    ++g_checked;
    int bad = 0;
    for (int yy = by0; yy <= by1 && !bad; ++yy) for (int xx = bx0; xx <= bx1 && !bad; ++xx) {
        const int i = yy * W + xx;
        const int ref = img[3 * i] != 0.0f, got = g_mask[i] || g_line_img[3 * i] != 0.0f;
        if (ref != got) {
            if (g_bad < 10) fprintf(stderr, "differs at (x %d, y %d): oracle %d model %d for (%d,%d) (%d,%d) (%d,%d) in %dx%d\n", i % W, i / W, ref, got,
                                    pts[0], pts[1], pts[2], pts[3], pts[4], pts[5], W, H);
            ++g_bad;
            bad = 1;
        }
    }
    for (int yy = by0; yy <= by1 && bx0 <= bx1; ++yy) {                 /* leave everything cleared for the next triangle */
        memset(img + 3 * (yy * W + bx0), 0, sizeof(float) * 3 * (bx1 - bx0 + 1));
        memset(g_line_img + 3 * (yy * W + bx0), 0, sizeof(float) * 3 * (bx1 - bx0 + 1));
        memset(g_mask + yy * W + bx0, 0, (size_t)(bx1 - bx0 + 1));
    }
}

int main(int argc, char **argv) {
    /* usage: fill_rows_model exhaustive RES LO HI | random RES COUNT SEED */
    if (argc < 5) { fprintf(stderr, "usage: %s exhaustive RES LO HI | random RES COUNT SEED\n", argv[0]); return 2; }
    g_W = g_H = atoi(argv[2]);
    if (!strncmp(argv[1], "nobias-", 7)) { BIAS = 0; argv[1] += 7; }
    float *img = (float *)calloc((size_t)3 * g_W * g_H, sizeof(float));
    g_line_img = (float *)calloc((size_t)3 * g_W * g_H, sizeof(float));
    g_mask = (uint8_t *)calloc((size_t)g_W * g_H, 1);
    if (!strcmp(argv[1], "one")) {                                     /* one RES x0 y0 x1 y1 x2 y2 */
        int32_t p[6];
        for (int i = 0; i < 6; ++i) p[i] = atoi(argv[3 + i]);
        check(p, img);
        printf("%ld triangles, %ld differ\n", g_checked, g_bad);
        return g_bad ? 1 : 0;
    }
    if (!strcmp(argv[1], "exhaustive")) {
        const int lo = atoi(argv[3]), hi = atoi(argv[4]);
        int32_t p[6];
        for (p[0] = lo; p[0] <= hi; ++p[0]) for (p[1] = lo; p[1] <= hi; ++p[1])
        for (p[2] = lo; p[2] <= hi; ++p[2]) for (p[3] = lo; p[3] <= hi; ++p[3])
        for (p[4] = lo; p[4] <= hi; ++p[4]) for (p[5] = lo; p[5] <= hi; ++p[5]) check(p, img);
    } else {
        const long count = atol(argv[3]);
        rng_state ^= (uint64_t)atol(argv[4]) * 0x2545F4914F6CDD1Dull;
        const int R = g_W;
        for (long n = 0; n < count; ++n) {
            int32_t p[6];
            const int kind = (int)(rnd() % 6);
            if (kind == 0) {                                           /* anywhere, also well outside */
                for (int i = 0; i < 6; ++i) p[i] = rnd_range(-R / 2, R + R / 2);
            } else if (kind == 1) {                                    /* inside the image */
                for (int i = 0; i < 6; ++i) p[i] = rnd_range(0, R - 1);
            } else if (kind == 2 || kind == 3) {                       /* slivers: two vertices close together, the third far away */
                p[0] = rnd_range(-8, R + 8); p[1] = rnd_range(-8, R + 8);
                p[2] = p[0] + rnd_range(-2, 2); p[3] = p[1] + rnd_range(-2, 2);
                const int len = kind == 2 ? R / 8 : R;
                p[4] = p[0] + rnd_range(-len, len); p[5] = p[1] + rnd_range(-len, len);
            } else if (kind == 4) {                                    /* small faces */
                p[0] = rnd_range(-4, R + 4); p[1] = rnd_range(-4, R + 4);
                for (int i = 2; i < 6; ++i) p[i] = p[i & 1] + rnd_range(-12, 12);
            } else {                                                   /* flat tops / bottoms / near-horizontal and near-vertical edges */
                p[0] = rnd_range(0, R - 1); p[1] = rnd_range(0, R - 1);
                p[2] = rnd_range(0, R - 1); p[3] = p[1] + rnd_range(-1, 1);
                p[4] = p[0] + rnd_range(-1, 1); p[5] = rnd_range(0, R - 1);
            }
            /* any order of the vertices */
            const int rot = (int)(rnd() % 3), flip = (int)(rnd() & 1);
            int32_t q[6];
            for (int i = 0; i < 3; ++i) { const int j = (flip ? 2 - i : i), k = (j + rot) % 3; q[2 * i] = p[2 * k]; q[2 * i + 1] = p[2 * k + 1]; }
            check(q, img);
        }
    }
    printf("%ld triangles, %ld differ\n", g_checked, g_bad);
    return g_bad ? 1 : 0;
}
