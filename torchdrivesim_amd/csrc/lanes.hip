// Lane tables and the wrong-way query.
//
//   Simulator.compute_wrong_way            reference simulator.py:607-630
//   -> lanelet_orientation_loss            reference infractions.py:232-304   (triple Python loop over scenes, agents, lanelets)
//   -> find_lanelet_directions             reference lanelet2.py:108-141      (lanelet2.geometry.findWithin2d, boost-python)
//   -> find_direction                      reference lanelet2.py:144-180      (lanelet2.geometry.project / distance)
//
// The reference asks the Lanelet2 C++ library one agent at a time.  Here a map is flattened ONCE on the host into a lane table
// (outline polygon and centre line of every lanelet as float64 arrays, a uniform grid of lanelet indices) and the whole batch is
// answered by one launch, sixteen lanes per agent, in float64 up to the direction angle (Lanelet2 computes in double) and in
// float32 from there on (the reference continues with `torch.tensor(directions)`, infractions.py:283).
//
// Host part of this file: the centre line of a lanelet (lanelet2_core Lanelet.cpp `calculateCenterline`, restated from the
// published source -- the library is not installed here, see torchdrivesim_amd/lanelet2.py for what pins it).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "tds_common.h"

namespace tds {

struct LaneRec {
    int32_t poly_start, poly_n;     // outline ring: left bound, then the right bound reversed (implicitly closed)
    int32_t cl_start, cl_n;         // centre line points
    int32_t flags;                  // bit 0: tagged with an excluded attribute ('parking', infractions.py:21)
    float bx0, by0, bx1, by1;       // bounding box of the outline, rounded outwards
};

struct LaneView {
    const double *poly;             // 2 doubles per point
    const double *cl;               // 3 doubles per point
    const LaneRec *rec;
    const int32_t *cell_start;      // nx*ny + 1
    const int32_t *cell_items;      // lanelet indices
    double ox, oy, inv_cell;
    int nx, ny, n;
    float max_tol;
};

}  // namespace tds

struct tds_lanes {
    tds::LaneView view;
    void *d_poly, *d_cl, *d_rec, *d_cell_start, *d_cell_items;
    int device;
    int64_t bytes;
};

struct tds_laneset {
    tds::LaneView *d_views;
    int n, device;
    float max_tol;
};

using tds::LaneRec;
using tds::LaneView;

// ---------------------------------------------------------------------------------------------------------------
// host: centre line
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct P2 {
    double x, y;
    bool operator==(const P2 &o) const { return x == o.x && y == o.y; }
    bool operator!=(const P2 &o) const { return !(*this == o); }
};

double orient(P2 a, P2 b, P2 c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }
bool on_box(P2 a, P2 b, P2 c) {
    return std::min(a.x, b.x) <= c.x && c.x <= std::max(a.x, b.x) && std::min(a.y, b.y) <= c.y && c.y <= std::max(a.y, b.y);
}
// closed segments p1p2 and q1q2 share a point
bool seg_intersect(P2 p1, P2 p2, P2 q1, P2 q2) {
    double o1 = orient(p1, p2, q1), o2 = orient(p1, p2, q2), o3 = orient(q1, q2, p1), o4 = orient(q1, q2, p2);
    if (((o1 > 0) != (o2 > 0)) && ((o3 > 0) != (o4 > 0)) && o1 != 0 && o2 != 0 && o3 != 0 && o4 != 0) return true;
    return (o1 == 0 && on_box(p1, p2, q1)) || (o2 == 0 && on_box(p1, p2, q2)) || (o3 == 0 && on_box(q1, q2, p1)) ||
           (o4 == 0 && on_box(q1, q2, p2));
}

struct BoundChecker {
    std::vector<P2> left, right;
    P2 entry[2], exit[2];
    // segments of the bound that leave from seg's first point do not count
    static bool crosses(const std::vector<P2> &line, P2 s0, P2 s1) {
        for (size_t i = 0; i + 1 < line.size(); i++)
            if (seg_intersect(s0, s1, line[i], line[i + 1]) && line[i] != s0 && line[i + 1] != s0) return true;
        return false;
    }
    // the inside of the lanelet is on the right of both gates: a segment leaves through a gate when its end lies strictly on the
    // outer side and it meets the gate
    static bool crosses_gate(const P2 *g, P2 s0, P2 s1) {
        bool outside = (g[1].x - g[0].x) * (s1.y - g[0].y) - (g[1].y - g[0].y) * (s1.x - g[0].x) > 0;
        return outside && seg_intersect(s0, s1, g[0], g[1]);
    }
    bool intersects(P2 s0, P2 s1) const {
        return crosses(left, s0, s1) || crosses(right, s0, s1) || crosses_gate(entry, s0, s1) || crosses_gate(exit, s0, s1);
    }
    // a connection between the bounds crosses the bound its SECOND point lies on, other than in that point
    bool second_crosses_bounds(P2 s0, P2 s1, bool is_left) const {
        const std::vector<P2> &line = is_left ? left : right;
        for (size_t i = 0; i + 1 < line.size(); i++)
            if (line[i] != s1 && line[i + 1] != s1 && seg_intersect(s0, s1, line[i], line[i + 1])) return true;
        return false;
    }
};

double dist2d(P2 a, P2 b) { return sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y)); }

// the point of `line` ahead of `cur` that is closest to `other` and whose connection stays inside the lanelet
bool closest_candidate(const BoundChecker &bc, const std::vector<P2> &line, int cur, P2 other, P2 last, bool is_left, int *best,
                       double *best_d) {
    std::vector<int> order;
    for (int k = cur + 1; k < (int)line.size(); k++) order.push_back(k);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return dist2d(line[a], other) < dist2d(line[b], other); });
    bool have = false;
    double d_last_other = dist2d(other, last);
    for (int k : order) {
        double d = dist2d(line[k], other) / 2.0;
        if (have && d - d_last_other > *best_d) break;      // no closer point can follow
        if (have && *best_d <= d) continue;
        P2 centre = {0.5 * (line[k].x + other.x), 0.5 * (line[k].y + other.y)};
        if (!bc.intersects(last, centre) && !bc.second_crosses_bounds(other, line[k], is_left) &&
            !bc.second_crosses_bounds(line[k], other, !is_left)) {
            have = true;
            *best = k;
            *best_d = d;
        }
    }
    return have;
}

}  // namespace

TDS_EXPORT int tds_lanelet_centerline_f64(const double *left, int n_left, const double *right, int n_right, double *out, int *n_out) {
    TDS_CHECK_ARG(n_left >= 0 && n_right >= 0 && out && n_out, "tds_lanelet_centerline_f64: bad arguments");
    *n_out = 0;
    if (n_left == 0 || n_right == 0) return TDS_OK;
    TDS_CHECK_ARG(left && right, "tds_lanelet_centerline_f64: null bound");
    BoundChecker bc;
    for (int i = 0; i < n_left; i++) bc.left.push_back({left[3 * i], left[3 * i + 1]});
    for (int i = 0; i < n_right; i++) bc.right.push_back({right[3 * i], right[3 * i + 1]});
    bc.entry[0] = bc.right.front(), bc.entry[1] = bc.left.front();
    bc.exit[0] = bc.left.back(), bc.exit[1] = bc.right.back();
    int n = 0;
    auto push = [&](int il, int ir) {
        for (int c = 0; c < 3; c++) out[3 * n + c] = 0.5 * (left[3 * il + c] + right[3 * ir + c]);
        n++;
    };
    push(0, 0);
    int il = 0, ir = 0;
    while (il < n_left - 1 || ir < n_right - 1) {
        P2 last = {out[3 * (n - 1)], out[3 * (n - 1) + 1]};
        int kl = -1, kr = -1;
        double dl = 0, dr = 0;
        bool hl = closest_candidate(bc, bc.left, il, bc.right[ir], last, true, &kl, &dl);
        bool hr = closest_candidate(bc, bc.right, ir, bc.left[il], last, false, &kr, &dr);
        if (hl && (!hr || dl <= dr)) {
            push(kl, ir);
            il = kl;
        } else if (hr) {
            push(il, kr);
            ir = kr;
        } else {
            break;
        }
    }
    if (!(il == n_left - 1 && ir == n_right - 1)) push(n_left - 1, n_right - 1);
    *n_out = n;
    return TDS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// host: lane table -> device
// ---------------------------------------------------------------------------------------------------------------
namespace {
inline int lane_cell(double v, double origin, double inv_cell) { return (int)floor((v - origin) * inv_cell); }

template <typename T>
int upload(void **dst, const std::vector<T> &src, int64_t *bytes) {
    size_t nb = std::max<size_t>(src.size(), 1) * sizeof(T);
    TDS_HIP(hipMalloc(dst, nb));
    if (!src.empty()) TDS_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    *bytes += (int64_t)nb;
    return TDS_OK;
}
}  // namespace

TDS_EXPORT int tds_lanes_create(const double *poly_xy, const int32_t *poly_start, const double *cl_xyz, const int32_t *cl_start,
                                const int32_t *flags, int n_lanelets, float cell_size, float max_tolerance, tds_lanes_t **out) {
    TDS_CHECK_ARG(out, "tds_lanes_create: out is null");
    TDS_CHECK_ARG(n_lanelets >= 0 && poly_start && cl_start, "tds_lanes_create: bad arguments");
    TDS_CHECK_ARG(max_tolerance >= 0.f && isfinite(max_tolerance), "tds_lanes_create: max_tolerance must be finite and >= 0");
    if (cell_size <= 0.f) cell_size = 8.f;
    int n_poly = poly_start[n_lanelets], n_cl = cl_start[n_lanelets];
    TDS_CHECK_ARG((n_poly == 0 || poly_xy) && (n_cl == 0 || cl_xyz), "tds_lanes_create: null point array");
    std::vector<LaneRec> rec(n_lanelets);
    double gx0 = INFINITY, gy0 = INFINITY, gx1 = -INFINITY, gy1 = -INFINITY;
    for (int l = 0; l < n_lanelets; l++) {
        LaneRec &r = rec[l];
        r.poly_start = poly_start[l], r.poly_n = poly_start[l + 1] - poly_start[l];
        r.cl_start = cl_start[l], r.cl_n = cl_start[l + 1] - cl_start[l];
        TDS_CHECK_ARG(r.poly_n >= 0 && r.cl_n >= 0, "tds_lanes_create: start arrays must not decrease");
        r.flags = flags ? flags[l] : 0;
        double x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY;
        for (int i = 0; i < r.poly_n; i++) {
            double x = poly_xy[2 * (r.poly_start + i)], y = poly_xy[2 * (r.poly_start + i) + 1];
            TDS_CHECK_ARG(isfinite(x) && isfinite(y), "tds_lanes_create: lanelet %d has a non-finite outline point", l);
            x0 = std::min(x0, x), y0 = std::min(y0, y), x1 = std::max(x1, x), y1 = std::max(y1, y);
        }
        r.bx0 = nextafterf((float)x0, -INFINITY), r.by0 = nextafterf((float)y0, -INFINITY);
        r.bx1 = nextafterf((float)x1, INFINITY), r.by1 = nextafterf((float)y1, INFINITY);
        if (r.poly_n > 0) gx0 = std::min(gx0, x0), gy0 = std::min(gy0, y0), gx1 = std::max(gx1, x1), gy1 = std::max(gy1, y1);
    }
    int device = 0;
    TDS_HIP(hipGetDevice(&device));
    tds_lanes *h = new (std::nothrow) tds_lanes();
    if (!h) return TDS_ENOMEM;
    memset(h, 0, sizeof(*h));
    h->device = device;
    LaneView &v = h->view;
    v.n = n_lanelets;
    v.max_tol = max_tolerance;
    v.inv_cell = 1.0 / (double)cell_size;
    double margin = (double)max_tolerance + 1e-3;
    std::vector<int32_t> cell_start(1, 0), cell_items;
    if (gx0 <= gx1) {
        v.ox = gx0 - margin, v.oy = gy0 - margin;
        v.nx = lane_cell(gx1 + margin, v.ox, v.inv_cell) + 1, v.ny = lane_cell(gy1 + margin, v.oy, v.inv_cell) + 1;
        if ((int64_t)v.nx * v.ny > (int64_t)1 << 26) {
            delete h;
            tds::set_error("tds_lanes_create: the grid would have %d x %d cells", v.nx, v.ny);
            return TDS_ELIMIT;
        }
        std::vector<std::vector<int32_t>> cells((size_t)v.nx * v.ny);
        for (int l = 0; l < n_lanelets; l++) {
            const LaneRec &r = rec[l];
            if (r.poly_n == 0) continue;
            // one extra cell around the grown box: the query computes its cell from a float32 coordinate widened to double
            int cx0 = std::max(0, lane_cell((double)r.bx0 - margin, v.ox, v.inv_cell) - 1), cx1 = std::min(v.nx - 1, lane_cell((double)r.bx1 + margin, v.ox, v.inv_cell) + 1);
            int cy0 = std::max(0, lane_cell((double)r.by0 - margin, v.oy, v.inv_cell) - 1), cy1 = std::min(v.ny - 1, lane_cell((double)r.by1 + margin, v.oy, v.inv_cell) + 1);
            for (int cy = cy0; cy <= cy1; cy++)
                for (int cx = cx0; cx <= cx1; cx++) cells[(size_t)cy * v.nx + cx].push_back(l);
        }
        cell_start.assign((size_t)v.nx * v.ny + 1, 0);
        for (size_t c = 0; c < cells.size(); c++) {
            cell_start[c + 1] = cell_start[c] + (int32_t)cells[c].size();
            cell_items.insert(cell_items.end(), cells[c].begin(), cells[c].end());
        }
    } else {
        v.ox = v.oy = 0, v.nx = v.ny = 0;
    }
    std::vector<double> poly(poly_xy, poly_xy + 2 * (size_t)n_poly), cl(cl_xyz, cl_xyz + 3 * (size_t)n_cl);
    int rc;
    if ((rc = upload(&h->d_poly, poly, &h->bytes)) || (rc = upload(&h->d_cl, cl, &h->bytes)) || (rc = upload(&h->d_rec, rec, &h->bytes)) ||
        (rc = upload(&h->d_cell_start, cell_start, &h->bytes)) || (rc = upload(&h->d_cell_items, cell_items, &h->bytes))) {
        tds_lanes_destroy(h);
        return rc;
    }
    v.poly = (const double *)h->d_poly, v.cl = (const double *)h->d_cl, v.rec = (const LaneRec *)h->d_rec;
    v.cell_start = (const int32_t *)h->d_cell_start, v.cell_items = (const int32_t *)h->d_cell_items;
    *out = h;
    return TDS_OK;
}

TDS_EXPORT int tds_lanes_destroy(tds_lanes_t *h) {
    if (!h) return TDS_OK;
    for (void *p : {h->d_poly, h->d_cl, h->d_rec, h->d_cell_start, h->d_cell_items}) (void)hipFree(p);
    delete h;
    return TDS_OK;
}

TDS_EXPORT int tds_lanes_info(const tds_lanes_t *h, int64_t *info) {
    TDS_CHECK_ARG(h && info, "tds_lanes_info: null argument");
    info[0] = h->view.n, info[1] = h->view.nx, info[2] = h->view.ny, info[3] = h->bytes;
    return TDS_OK;
}

TDS_EXPORT int tds_laneset_create(const tds_lanes_t *const *lanes, int n, tds_laneset_t **out) {
    TDS_CHECK_ARG(lanes && n > 0 && out, "tds_laneset_create: bad arguments");
    std::vector<LaneView> views(n);
    float max_tol = INFINITY;
    for (int i = 0; i < n; i++) {
        TDS_CHECK_ARG(lanes[i], "tds_laneset_create: lane table %d is null", i);
        TDS_CHECK_ARG(lanes[i]->device == lanes[0]->device, "tds_laneset_create: lane tables live on different devices");
        views[i] = lanes[i]->view;
        max_tol = std::min(max_tol, lanes[i]->view.max_tol);
    }
    tds_laneset *s = new (std::nothrow) tds_laneset();
    if (!s) return TDS_ENOMEM;
    s->n = n, s->device = lanes[0]->device, s->max_tol = max_tol, s->d_views = nullptr;
    hipError_t e = hipMalloc((void **)&s->d_views, n * sizeof(LaneView));
    if (e == hipSuccess) e = hipMemcpy(s->d_views, views.data(), n * sizeof(LaneView), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(s->d_views);
        delete s;
        tds::set_error("tds_laneset_create: %s", hipGetErrorString(e));
        return TDS_EHIP;
    }
    *out = s;
    return TDS_OK;
}

TDS_EXPORT int tds_laneset_destroy(tds_laneset_t *s) {
    if (!s) return TDS_OK;
    (void)hipFree(s->d_views);
    delete s;
    return TDS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// device: the query.  An agent is served by a GROUP of 16 lanes (a quarter wavefront): the lanes split the outline edges and the
// centre-line points of a candidate lanelet between them and combine their partial results with xor-shuffles inside the group,
// so that 65 536 agents are 16 384 wavefronts (the first version, one thread per agent with serial loops over up to 250 points
// per lanelet, ran one wave per SIMD and took 1.0 ms at B = 1024 x A = 64).
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int GROUP = 16;

__device__ inline double group_min(double v) {
#pragma unroll
    for (int m = GROUP / 2; m > 0; m >>= 1) v = fmin(v, __shfl_xor(v, m, GROUP));
    return v;
}

// squared distance from (x, y) to the outline ring, 0 when the point is inside (boost::geometry::distance(point, polygon) as used by
// lanelet2 geometry::findWithin2d); edge i runs from point i-1 to point i, lanes take every 16th edge
__device__ double ring_distance2(const double *poly, int n, double x, double y, int g) {
    int parity = 0;
    double best = INFINITY;
    for (int i = g; i < n; i += GROUP) {
        int ia = i == 0 ? n - 1 : i - 1;
        double ax = poly[2 * ia], ay = poly[2 * ia + 1], bx = poly[2 * i], by = poly[2 * i + 1];
        if ((ay > y) != (by > y)) {
            double xc = ax + (y - ay) * (bx - ax) / (by - ay);
            if (x < xc) parity ^= 1;
        }
        double dx = bx - ax, dy = by - ay, l2 = dx * dx + dy * dy;
        double t = l2 > 0 ? ((x - ax) * dx + (y - ay) * dy) / l2 : 0.0;
        t = fmin(fmax(t, 0.0), 1.0);
        double fx = ax + t * dx - x, fy = ay + t * dy - y;
        best = fmin(best, fx * fx + fy * fy);
    }
#pragma unroll
    for (int m = GROUP / 2; m > 0; m >>= 1) parity ^= __shfl_xor(parity, m, GROUP);
    best = group_min(best);
    return parity ? 0.0 : best;
}

__device__ inline bool lex_less(double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); }

// find_direction (lanelet2.py:144-180): false when the two vertices closest to the projection are not neighbours
__device__ bool line_direction(const double *cl, int n, double x, double y, int g, double *dir) {
    // lanelet2.geometry.project(linestring, BasicPoint3d(x, y, 0)): the closest point of the line, first segment on ties
    double best = INFINITY;
    int sb = 0x7fffffff;
    for (int i = g; i + 1 < n; i += GROUP) {
        double ax = cl[3 * i], ay = cl[3 * i + 1], az = cl[3 * i + 2];
        double dx = cl[3 * i + 3] - ax, dy = cl[3 * i + 4] - ay, dz = cl[3 * i + 5] - az;
        double l2 = dx * dx + dy * dy + dz * dz;
        double t = l2 > 0 ? ((x - ax) * dx + (y - ay) * dy + (0.0 - az) * dz) / l2 : 0.0;
        t = fmin(fmax(t, 0.0), 1.0);
        double fx = ax + t * dx, fy = ay + t * dy, fz = az + t * dz;
        double d2 = (fx - x) * (fx - x) + (fy - y) * (fy - y) + fz * fz;
        if (d2 < best) best = d2, sb = i;
    }
#pragma unroll
    for (int m = GROUP / 2; m > 0; m >>= 1) {
        double od = __shfl_xor(best, m, GROUP);
        int oi = __shfl_xor(sb, m, GROUP);
        if (lex_less(od, oi, best, sb)) best = od, sb = oi;
    }
    double px, py, pz;
    {   // the foot point on the winning segment, with the arithmetic of the scan above
        double ax = cl[3 * sb], ay = cl[3 * sb + 1], az = cl[3 * sb + 2];
        double dx = cl[3 * sb + 3] - ax, dy = cl[3 * sb + 4] - ay, dz = cl[3 * sb + 5] - az;
        double l2 = dx * dx + dy * dy + dz * dz;
        double t = l2 > 0 ? ((x - ax) * dx + (y - ay) * dy + (0.0 - az) * dz) / l2 : 0.0;
        t = fmin(fmax(t, 0.0), 1.0);
        px = ax + t * dx, py = ay + t * dy, pz = az + t * dz;
    }
    // the two vertices closest to the projection: the reference's sequential scan (strict <) keeps the two smallest in the order
    // (distance, index)
    double d1 = INFINITY, d2 = INFINITY;
    int i1 = 0x7fffffff, i2 = 0x7fffffff;
    for (int i = g; i < n; i += GROUP) {
        double dx = cl[3 * i] - px, dy = cl[3 * i + 1] - py, dz = cl[3 * i + 2] - pz;
        double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (lex_less(d, i, d1, i1)) {
            d2 = d1, i2 = i1, d1 = d, i1 = i;
        } else if (lex_less(d, i, d2, i2)) {
            d2 = d, i2 = i;
        }
    }
#pragma unroll
    for (int m = GROUP / 2; m > 0; m >>= 1) {
        double e1 = __shfl_xor(d1, m, GROUP), e2 = __shfl_xor(d2, m, GROUP);
        int j1 = __shfl_xor(i1, m, GROUP), j2 = __shfl_xor(i2, m, GROUP);
        if (lex_less(d1, i1, e1, j1)) {                   // mine first; second = the better of my second and the other's first
            if (!lex_less(d2, i2, e1, j1)) d2 = e1, i2 = j1;
        } else {
            if (lex_less(e2, j2, d1, i1)) d2 = e2, i2 = j2; else d2 = d1, i2 = i1;
            d1 = e1, i1 = j1;
        }
    }
    int lo = min(i1, i2), hi = max(i1, i2);
    if (i2 == 0x7fffffff || hi - lo != 1) return false;
    *dir = atan2(cl[3 * hi + 1] - cl[3 * lo + 1], cl[3 * hi] - cl[3 * lo]);
    return true;
}

// utils.normalize_angle on a float32 tensor (utils.py:31-37): (angle + pi) % (2 pi) - pi with torch.remainder
__device__ float normalize_angle_f32(float a) {
    const float PI_F = 3.14159265358979323846f, TWO_PI_F = 6.28318530717958647692f;
    float r = fmodf(a + PI_F, TWO_PI_F);
    if (r != 0.f && r < 0.f) r += TWO_PI_F;
    return r - PI_F;
}

__global__ __launch_bounds__(256) void wrong_way_kernel(const LaneView *views, const int32_t *scene_map, int64_t agents_per_scene,
                                                        const float *xy, const double *xyd, const float *psi,
                                                        const float *offset, const uint8_t *present, float *out, double *dirs,
                                                        double *dists, int32_t *count, uint8_t *status, int max_dirs, int64_t n,
                                                        float tol, float thr) {
    int64_t a = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
    int g = threadIdx.x & (GROUP - 1);
    if (a >= n) return;                               // whole groups leave together
    int64_t scene = a / agents_per_scene;
    int m = scene_map ? scene_map[scene] : 0;
    float loss = 0.f;
    int k = 0;
    bool excluded = false, failed = false;
    if (m >= 0) {
        const LaneView v = views[m];
        double x, y;
        if (xyd) {
            x = xyd[2 * a], y = xyd[2 * a + 1];                          // find_lanelet_directions(x, y) of host doubles
        } else {
            float xf = xy[4 * a], yf = xy[4 * a + 1];                    // float(agent_state[0]), infractions.py:269
            if (offset) xf = xf + offset[2 * scene], yf = yf + offset[2 * scene + 1];      // recenter_offset, infractions.py:271-273
            x = xf, y = yf;
        }
        float ps = psi ? psi[4 * a] : 0.f;
        int cx = (int)floor((x - v.ox) * v.inv_cell), cy = (int)floor((y - v.oy) * v.inv_cell);
        if (cx >= 0 && cy >= 0 && cx < v.nx && cy < v.ny) {              // also false for NaN coordinates
            int c = cy * v.nx + cx;
            double t = (double)tol;
            for (int it = v.cell_start[c]; it < v.cell_start[c + 1]; it++) {
                const LaneRec r = v.rec[v.cell_items[it]];
                if (x < (double)r.bx0 - t || x > (double)r.bx1 + t || y < (double)r.by0 - t || y > (double)r.by1 + t) continue;
                double d = sqrt(ring_distance2(v.poly + 2 * (int64_t)r.poly_start, r.poly_n, x, y, g));
                if (!(d <= t)) continue;
                if (r.cl_n < 2) continue;                                 // lanelet2.py:131-132
                if (r.flags & 1) {                                        // lanelet2.py:133-135: no directions at all
                    excluded = true;
                    break;
                }
                double dir;
                if (!line_direction(v.cl + 3 * (int64_t)r.cl_start, r.cl_n, x, y, g, &dir)) {
                    failed = true;                                        // LaneletError -> loss 0 (infractions.py:290-294)
                    continue;
                }
                float df = (float)dir;
                float delta = normalize_angle_f32(df - ps);
                float l = -cosf(delta) * (fabsf(delta) > thr ? 1.f : 0.f);
                loss = k == 0 ? l : fminf(loss, l);
                if (dirs && k < max_dirs && g == 0) dirs[a * max_dirs + k] = dir, dists[a * max_dirs + k] = d;
                k++;
            }
        }
    }
    if (g != 0) return;
    if (excluded || failed) loss = 0.f, k = excluded ? 0 : k;
    if (present && !present[a]) loss = 0.f;       // `* self.get_present_mask()`, simulator.py:624 (loss is never NaN or infinite)
    if (out) out[a] = loss;
    if (count) count[a] = k;
    if (status) status[a] = (excluded ? 2 : 0) | (failed ? 1 : 0);
}

}  // namespace

TDS_EXPORT int tds_wrong_way_f32(const tds_laneset_t *set, const int32_t *scene_map, int64_t agents_per_scene, const float *state,
                                 const float *recenter_offset, const uint8_t *present, float *out, int64_t n_agents,
                                 float direction_angle_threshold, float lanelet_dist_tolerance, void *stream) {
    TDS_CHECK_ARG(set && out, "tds_wrong_way_f32: null argument");
    TDS_CHECK_ARG(n_agents >= 0 && agents_per_scene > 0, "tds_wrong_way_f32: bad sizes");
    TDS_CHECK_ARG(lanelet_dist_tolerance >= 0.f && lanelet_dist_tolerance <= set->max_tol,
                  "tds_wrong_way_f32: lanelet_dist_tolerance %g exceeds the %g the lane tables were built for", lanelet_dist_tolerance,
                  set->max_tol);
    TDS_CHECK_ARG(scene_map || set->n == 1, "tds_wrong_way_f32: a set of %d lane tables needs scene_map", set->n);
    if (n_agents == 0) return TDS_OK;
    TDS_CHECK_ARG(state, "tds_wrong_way_f32: state is null");
    unsigned blocks = (unsigned)((n_agents * GROUP + 255) / 256);
    hipLaunchKernelGGL(wrong_way_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, set->d_views, scene_map, agents_per_scene, state,
                       (const double *)nullptr, state + 2, recenter_offset, present, out, (double *)nullptr, (double *)nullptr,
                       (int32_t *)nullptr, (uint8_t *)nullptr, 0, n_agents, lanelet_dist_tolerance, direction_angle_threshold);
    TDS_LAUNCH_CHECK("wrong_way_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_lanelet_directions_f64(const tds_laneset_t *set, const int32_t *scene_map, int64_t points_per_scene, const double *xy,
                                          double *dirs, double *dists, int32_t *count, uint8_t *status, int max_dirs, int64_t n_points,
                                          float lanelet_dist_tolerance, void *stream) {
    TDS_CHECK_ARG(set && dirs && dists && count && status, "tds_lanelet_directions_f64: null argument");
    TDS_CHECK_ARG(n_points >= 0 && points_per_scene > 0 && max_dirs > 0, "tds_lanelet_directions_f64: bad sizes");
    TDS_CHECK_ARG(lanelet_dist_tolerance >= 0.f && lanelet_dist_tolerance <= set->max_tol,
                  "tds_lanelet_directions_f64: lanelet_dist_tolerance %g exceeds the %g the lane tables were built for",
                  lanelet_dist_tolerance, set->max_tol);
    TDS_CHECK_ARG(scene_map || set->n == 1, "tds_lanelet_directions_f64: a set of %d lane tables needs scene_map", set->n);
    if (n_points == 0) return TDS_OK;
    TDS_CHECK_ARG(xy, "tds_lanelet_directions_f64: xy is null");
    unsigned blocks = (unsigned)((n_points * GROUP + 255) / 256);
    hipLaunchKernelGGL(wrong_way_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, set->d_views, scene_map, points_per_scene,
                       (const float *)nullptr, xy, (const float *)nullptr, (const float *)nullptr, (const uint8_t *)nullptr, (float *)nullptr,
                       dirs, dists, count, status, max_dirs, n_points, lanelet_dist_tolerance, 4.f);
    TDS_LAUNCH_CHECK("wrong_way_kernel");
    return TDS_OK;
}
