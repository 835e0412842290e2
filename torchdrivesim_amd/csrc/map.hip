// Static map handle (triangle soup binned into a uniform grid, built once per map on the host and uploaded) and
// K2b: offroad = thresholded squared distance from the 4 corners of every agent to the nearest mesh face.
// Reference: simulator.py:1035-1044; infractions.py:86-229 (pure-torch path), which materialises a
// (B*A*4, F, 3, 3) tensor; here each corner walks grid rings outwards until the best distance is proven minimal.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "tds_common.h"

using tds::GridEntry;
using tds::MapView;

// The ring-walk fallback of K2b can be forced (no nearest-face candidate lists) only in the TESTING build of the library
// (tds_testing_set_near_lists, include/tdship.h "testing hooks"); the product never reads the environment.
#ifdef TDS_TESTING
static int g_near_lists = 1;
#define TDS_NEAR_LISTS_ENABLED (g_near_lists != 0)
#define TDS_BVH_BUILD (g_near_lists != 2)          // 2: candidate lists but no hierarchy (points beyond the lists walk grid rings, as before round 5)
TDS_EXPORT int tds_testing_set_near_lists(int enabled) {
    g_near_lists = enabled;
    return TDS_OK;
}
#else
#define TDS_NEAR_LISTS_ENABLED true
#define TDS_BVH_BUILD true
#endif


namespace {
// Host: nearest-face candidate lists (tds::NearView).  For a cell C (its box grown by a rounding margin) and the reference's distance
// d(p, f) (infractions.py:100-170: 0 inside, else the min over the three edges of the clamped point-segment distance):
//   * d(p, f) <= |p - v0(f)|^2 for every p (the clamped foot point of edge v0-v1 is never farther than its start point), so
//     U = min_f max_{corner of C} |corner - v0(f)|^2 bounds the nearest-face distance of every point of C from above;
//   * d(p, f) >= the squared distance between C and the bounding box of f =: lb(C, f).
// Every face that is nearest to some point of C therefore has lb(C, f) <= U; the list holds exactly those, sorted by lb (then large
// faces first), so the walk may stop at the first candidate whose lb is not below the best distance found.
void build_near_lists(const float *verts, const int32_t *faces, int64_t F, float map_ox, float map_oy, float cell, int map_nx, int map_ny,
                      std::vector<int32_t> &cand_start, std::vector<tds::NearCand> &cand, std::vector<GridEntry> &face_entries, float &ox, float &oy,
                      int &nx, int &ny) {
    const float inv = 1.0f / cell;
    // own grid: the map's, grown by about 48 m on every side (less for huge maps)
    int pad = (int)std::ceil(48.0f / cell);
    while (pad > 0 && (int64_t)(map_nx + 2 * pad) * (map_ny + 2 * pad) > (int64_t)(1 << 22)) pad /= 2;
    ox = map_ox - (float)pad * cell; oy = map_oy - (float)pad * cell;
    nx = map_nx + 2 * pad; ny = map_ny + 2 * pad;
    struct FaceBox { double x0, y0, x1, y1, vx, vy, area2; bool ok; };
    std::vector<FaceBox> fb((size_t)F);
    face_entries.assign((size_t)F, GridEntry{0, 0, 0, 0, 0, 0, 0, 0});
    std::vector<int32_t> cs((size_t)nx * ny + 1, 0);
    auto cells_of = [&](const FaceBox &b, int &cx0, int &cx1, int &cy0, int &cy1) {
        cx0 = tds::cell_coord((float)b.x0, ox, inv); cx1 = tds::cell_coord((float)b.x1, ox, inv);
        cy0 = tds::cell_coord((float)b.y0, oy, inv); cy1 = tds::cell_coord((float)b.y1, oy, inv);
    };
    for (int64_t f = 0; f < F; ++f) {
        const float *p0 = verts + 2 * faces[3 * f], *p1 = verts + 2 * faces[3 * f + 1], *p2 = verts + 2 * faces[3 * f + 2];
        FaceBox &b = fb[(size_t)f];
        b.ok = std::isfinite(p0[0]) && std::isfinite(p0[1]) && std::isfinite(p1[0]) && std::isfinite(p1[1]) && std::isfinite(p2[0]) && std::isfinite(p2[1]);
        if (!b.ok) continue;
        GridEntry &e = face_entries[(size_t)f];
        e.x0 = p0[0]; e.y0 = p0[1]; e.x1 = p1[0]; e.y1 = p1[1]; e.x2 = p2[0]; e.y2 = p2[1];
        b.x0 = std::min(p0[0], std::min(p1[0], p2[0])); b.x1 = std::max(p0[0], std::max(p1[0], p2[0]));
        b.y0 = std::min(p0[1], std::min(p1[1], p2[1])); b.y1 = std::max(p0[1], std::max(p1[1], p2[1]));
        b.vx = p0[0]; b.vy = p0[1];
        b.area2 = std::fabs(((double)p1[0] - p0[0]) * ((double)p2[1] - p0[1]) - ((double)p2[0] - p0[0]) * ((double)p1[1] - p0[1]));
        int cx0, cx1, cy0, cy1;
        cells_of(b, cx0, cx1, cy0, cy1);
        for (int cy = cy0; cy <= cy1; ++cy)
            for (int cx = cx0; cx <= cx1; ++cx) cs[(size_t)cy * nx + cx + 1]++;
    }
    for (size_t i = 1; i < cs.size(); ++i) cs[i] += cs[i - 1];
    std::vector<int32_t> ids((size_t)cs.back()), cur(cs.begin(), cs.end() - 1);
    for (int64_t f = 0; f < F; ++f) {
        if (!fb[(size_t)f].ok) continue;
        int cx0, cx1, cy0, cy1;
        cells_of(fb[(size_t)f], cx0, cx1, cy0, cy1);
        for (int cy = cy0; cy <= cy1; ++cy)
            for (int cx = cx0; cx <= cx1; ++cx) ids[(size_t)cur[(size_t)cy * nx + cx]++] = (int32_t)f;
    }
    // a point is assigned to a cell in float arithmetic (cell_coord), so it may lie a hair outside the nominal box
    const double grow = 1e-3 + 1e-6 * (std::fabs((double)ox) + std::fabs((double)oy) + (double)cell * (nx + ny));
    std::vector<int32_t> stamp((size_t)F, -1);
    struct Tmp { double lb, area2; int32_t f; };
    std::vector<Tmp> tmp;
    cand_start.assign((size_t)nx * ny + 1, 0);
    cand.clear();
    for (int cy = 0; cy < ny; ++cy)
        for (int cx = 0; cx < nx; ++cx) {
            const int32_t c = cy * nx + cx;
            const double bx0 = (double)ox + (double)cx * cell - grow, bx1 = (double)ox + (double)(cx + 1) * cell + grow;
            const double by0 = (double)oy + (double)cy * cell - grow, by1 = (double)oy + (double)(cy + 1) * cell + grow;
            double U = INFINITY;
            for (int k = 0;; ++k) {
                const int y0 = std::max(cy - k, 0), y1 = std::min(cy + k, ny - 1), x0 = std::max(cx - k, 0), x1 = std::min(cx + k, nx - 1);
                for (int y = y0; y <= y1; ++y)
                    for (int x = x0; x <= x1; ++x) {
                        if (std::max(std::abs(x - cx), std::abs(y - cy)) != k) continue;
                        for (int32_t i = cs[(size_t)y * nx + x]; i < cs[(size_t)y * nx + x + 1]; ++i) {
                            const FaceBox &b = fb[(size_t)ids[(size_t)i]];
                            const double dx = std::max(std::fabs(b.vx - bx0), std::fabs(b.vx - bx1)), dy = std::max(std::fabs(b.vy - by0), std::fabs(b.vy - by1));
                            U = std::min(U, dx * dx + dy * dy);
                        }
                    }
                // faces not met yet lie entirely outside the box grown by k cells
                const double reach = (double)k * cell;
                if (U <= reach * reach) break;
                if (cx - k <= 0 && cx + k >= nx - 1 && cy - k <= 0 && cy + k >= ny - 1) break;
            }
            tmp.clear();
            if (std::isfinite(U)) {
                const double Um = U * 1.002 + 1e-2, R = std::sqrt(Um);
                const int x0 = std::max((int)std::floor((bx0 - R - ox) / cell) - 1, 0), x1 = std::min((int)std::floor((bx1 + R - ox) / cell) + 1, nx - 1);
                const int y0 = std::max((int)std::floor((by0 - R - oy) / cell) - 1, 0), y1 = std::min((int)std::floor((by1 + R - oy) / cell) + 1, ny - 1);
                for (int y = y0; y <= y1; ++y)
                    for (int x = x0; x <= x1; ++x)
                        for (int32_t i = cs[(size_t)y * nx + x]; i < cs[(size_t)y * nx + x + 1]; ++i) {
                            const int32_t f = ids[(size_t)i];
                            if (stamp[(size_t)f] == c) continue;
                            stamp[(size_t)f] = c;
                            const FaceBox &b = fb[(size_t)f];
                            const double dx = std::max(std::max(b.x0 - bx1, bx0 - b.x1), 0.0), dy = std::max(std::max(b.y0 - by1, by0 - b.y1), 0.0);
                            const double lb = dx * dx + dy * dy;
                            if (lb <= Um) tmp.push_back(Tmp{lb, b.area2, f});
                        }
                std::sort(tmp.begin(), tmp.end(), [](const Tmp &a, const Tmp &b) {
                    if (a.lb != b.lb) return a.lb < b.lb;
                    if (a.area2 != b.area2) return a.area2 > b.area2;
                    return a.f < b.f;
                });
            }
            for (const Tmp &t : tmp) {
                float lb = (float)std::max(t.lb * 0.998 - 1e-3, 0.0);
                if ((double)lb > t.lb * 0.998 - 1e-3 && lb > 0.0f) lb = std::nextafter(lb, 0.0f);
                cand.push_back(tds::NearCand{t.f, lb});
            }
            cand_start[(size_t)c + 1] = (int32_t)cand.size();
        }
}
// Host: the hierarchy over the faces with finite vertices (tds::BvhNode).  A node's faces are cut into up to eight parts by three rounds of
// median splits, each along the longer axis of the part's centroids; parts of at most 8 faces are leaves (one face per lane of a group of the
// off-road kernels), larger ones nodes of their own: the depth stays below 8 for any mesh tds_map_create accepts (fewer than 2^24 faces).
// -> number of inner levels of the hierarchy (0: none)
int build_bvh(const float *verts, const int32_t *faces, int64_t F, std::vector<tds::BvhNode> &nodes, std::vector<int32_t> &idx) {
    struct Box { float x0, y0, x1, y1; };
    std::vector<Box> fb((size_t)F);
    std::vector<float> cx((size_t)F), cy((size_t)F);
    idx.clear(); nodes.clear();
    for (int64_t f = 0; f < F; ++f) {
        const float *p0 = verts + 2 * faces[3 * f], *p1 = verts + 2 * faces[3 * f + 1], *p2 = verts + 2 * faces[3 * f + 2];
        if (!(std::isfinite(p0[0]) && std::isfinite(p0[1]) && std::isfinite(p1[0]) && std::isfinite(p1[1]) && std::isfinite(p2[0]) && std::isfinite(p2[1]))) continue;
        Box &b = fb[(size_t)f];
        b.x0 = std::min(p0[0], std::min(p1[0], p2[0])); b.x1 = std::max(p0[0], std::max(p1[0], p2[0]));
        b.y0 = std::min(p0[1], std::min(p1[1], p2[1])); b.y1 = std::max(p0[1], std::max(p1[1], p2[1]));
        cx[(size_t)f] = 0.5f * (b.x0 + b.x1); cy[(size_t)f] = 0.5f * (b.y0 + b.y1);
        idx.push_back((int32_t)f);
    }
    if (idx.empty()) return 0;
    const Box none = {INFINITY, INFINITY, -INFINITY, -INFINITY};
    auto box_of = [&](int lo, int hi) {
        Box b = none;
        for (int i = lo; i < hi; ++i) {
            const Box &q = fb[(size_t)idx[(size_t)i]];
            b.x0 = std::min(b.x0, q.x0); b.y0 = std::min(b.y0, q.y0); b.x1 = std::max(b.x1, q.x1); b.y1 = std::max(b.y1, q.y1);
        }
        return b;
    };
    // idx[lo, hi) cut in two at the median of the longer axis of its centroids' box (ties by face index: the build is deterministic)
    auto split = [&](int lo, int hi) {
        float mx0 = INFINITY, mx1 = -INFINITY, my0 = INFINITY, my1 = -INFINITY;
        for (int i = lo; i < hi; ++i) {
            mx0 = std::min(mx0, cx[(size_t)idx[(size_t)i]]); mx1 = std::max(mx1, cx[(size_t)idx[(size_t)i]]);
            my0 = std::min(my0, cy[(size_t)idx[(size_t)i]]); my1 = std::max(my1, cy[(size_t)idx[(size_t)i]]);
        }
        const std::vector<float> &key = (mx1 - mx0 >= my1 - my0) ? cx : cy;
        const int mid = lo + (hi - lo) / 2;
        std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int32_t a, int32_t b) { return key[(size_t)a] < key[(size_t)b] || (key[(size_t)a] == key[(size_t)b] && a < b); });
        return mid;
    };
    struct Work { int lo, hi, node, level; };
    std::vector<Work> work;
    int depth = 1;
    nodes.push_back(tds::BvhNode{});
    work.push_back(Work{0, (int)idx.size(), 0, 1});
    while (!work.empty()) {
        const Work w = work.back();
        work.pop_back();
        // the parts of this node: three rounds, each cutting every part of more than 8 faces in two
        int cut[9] = {w.lo, w.hi, 0, 0, 0, 0, 0, 0, 0}, nparts = 1;
        for (int round = 0; round < 3; ++round) {
            int next[9], nn = 0;
            for (int p = 0; p < nparts; ++p) {
                next[nn++] = cut[p];
                if (cut[p + 1] - cut[p] > 8) next[nn++] = split(cut[p], cut[p + 1]);
            }
            next[nn] = cut[nparts];
            for (int p = 0; p <= nn; ++p) cut[p] = next[p];
            nparts = nn;
        }
        tds::BvhNode n;
        for (int k = 0; k < 8; ++k) {
            n.box[k] = make_float4(none.x0, none.y0, none.x1, none.y1);
            n.child[k] = -1;                                   // a leaf without faces
            if (k >= nparts) continue;
            const int lo = cut[k], hi = cut[k + 1];
            const Box b = box_of(lo, hi);
            n.box[k] = make_float4(b.x0, b.y0, b.x1, b.y1);
            if (hi - lo <= 8) {
                n.child[k] = -1 - ((lo << 4) | (hi - lo));
            } else {
                nodes.push_back(tds::BvhNode{});
                work.push_back(Work{lo, hi, (int)nodes.size() - 1, w.level + 1});
                depth = std::max(depth, w.level + 1);
                n.child[k] = (int32_t)nodes.size() - 1;
            }
        }
        nodes[(size_t)w.node] = n;
    }
    return depth;
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------
// host: build + upload
// ------------------------------------------------------------------------------------------------------------
TDS_EXPORT int tds_map_create(const float *verts, const int32_t *faces, const float *face_z, const uint32_t *face_rgb, int64_t V,
                              int64_t F, const float *levels, int n_levels, float cell_size, tds_map_t **out) {
    TDS_CHECK_ARG(out, "tds_map_create: out is null");
    *out = nullptr;
    TDS_CHECK_ARG(V >= 0 && F >= 0, "tds_map_create: negative size");
    TDS_CHECK_ARG((V == 0 || verts) && (F == 0 || faces), "tds_map_create: null mesh arrays");
    TDS_CHECK_ARG(F < (1 << 24), "tds_map_create: more than 2^24 faces");
    TDS_CHECK_ARG((face_z == nullptr) == (face_rgb == nullptr), "tds_map_create: face_z and face_rgb go together");
    if (face_z) {
        TDS_CHECK_ARG(levels && n_levels > 0, "tds_map_create: rendering data needs a level table");
        if (n_levels > 255) { tds::set_error("tds_map_create: %d rendering levels (max 255)", n_levels); return TDS_ELIMIT; }
        for (int i = 1; i < n_levels; ++i)
            TDS_CHECK_ARG(levels[i] < levels[i - 1], "tds_map_create: levels must be strictly descending");
    }
    for (int64_t i = 0; i < 3 * F; ++i) TDS_CHECK_ARG(faces[i] >= 0 && faces[i] < V, "tds_map_create: face index %d out of range", faces[i]);

    float cell = cell_size > 0 ? cell_size : 8.0f;
    float minx = 0, miny = 0, maxx = 0, maxy = 0;
    bool any = false;
    for (int64_t f = 0; f < F; ++f)
        for (int k = 0; k < 3; ++k) {
            float x = verts[2 * faces[3 * f + k]], y = verts[2 * faces[3 * f + k] + 1];
            if (!std::isfinite(x) || !std::isfinite(y)) continue;
            if (!any) { minx = maxx = x; miny = maxy = y; any = true; }
            minx = std::min(minx, x); maxx = std::max(maxx, x); miny = std::min(miny, y); maxy = std::max(maxy, y);
        }
    // Outline edges shared by faces of the same key: Line(p, q) and Line(q, p) paint the same pixels (OpenCV walks left to
    // right), so when two faces with the same key share an edge only one of them has to draw it.  dup[f] bit l marks edge l of
    // face f as a repeat of an edge of an EARLIER face; the rasteriser skips it only when one of the edge's end points passes
    // the trim test, which guarantees that the earlier face is drawn too (it shares both end points).
    std::vector<uint8_t> dup((size_t)F, 0);
    if (face_z) {
        struct EdgeKey { uint32_t a[4]; uint32_t key; bool operator==(const EdgeKey &o) const { return memcmp(this, &o, sizeof(EdgeKey)) == 0; } };
        struct EdgeHash { size_t operator()(const EdgeKey &k) const { size_t h = 1469598103934665603ull; for (int i = 0; i < 4; ++i) h = (h ^ k.a[i]) * 1099511628211ull; return (h ^ k.key) * 1099511628211ull; } };
        std::unordered_map<EdgeKey, int64_t, EdgeHash> seen;
        seen.reserve((size_t)F * 3);
        for (int64_t f = 0; f < F; ++f) {
            int rank = -1;
            for (int l = 0; l < n_levels; ++l) if (levels[l] == face_z[f]) { rank = l + 1; break; }
            if (rank < 0) continue;
            uint32_t key = ((uint32_t)rank << 24) | (face_rgb[f] & 0xFFFFFFu);
            const int ea[3] = {2, 0, 1}, eb[3] = {0, 1, 2};
            for (int l = 0; l < 3; ++l) {
                float p[2] = {verts[2 * faces[3 * f + ea[l]]], verts[2 * faces[3 * f + ea[l]] + 1]};
                float q[2] = {verts[2 * faces[3 * f + eb[l]]], verts[2 * faces[3 * f + eb[l]] + 1]};
                if (p[0] > q[0] || (p[0] == q[0] && p[1] > q[1])) { std::swap(p[0], q[0]); std::swap(p[1], q[1]); }
                EdgeKey ek;
                memcpy(&ek.a[0], &p[0], 4); memcpy(&ek.a[1], &p[1], 4); memcpy(&ek.a[2], &q[0], 4); memcpy(&ek.a[3], &q[1], 4);
                ek.key = key;
                auto it = seen.find(ek);
                if (it == seen.end()) seen.emplace(ek, f);
                else if (it->second != f) dup[(size_t)f] |= (uint8_t)(1u << l);
            }
        }
    }
    // Pairs for the rendering grid (tds::QuadEntry): two faces of the same key that share an edge -- bit-identical end points -- become ONE
    // entry.  Greedy in face order: a face takes the unpaired partner of lowest index over any of its three edges (meshes of triangulated
    // quads list the two halves one after the other).  Nothing about the shape is required: the rasteriser decides pair by pair, in pixel space,
    // whether the rows of the two halves can be painted as one (raster.hip), and draws every triangle with its own vertex order either way.
    std::vector<int64_t> partner((size_t)F, -1);
    int64_t n_pairs = 0;
    if (face_z) {
        struct EdgeKey { uint32_t a[4]; uint32_t key; bool operator==(const EdgeKey &o) const { return memcmp(this, &o, sizeof(EdgeKey)) == 0; } };
        struct EdgeHash { size_t operator()(const EdgeKey &k) const { size_t h = 1469598103934665603ull; for (int i = 0; i < 4; ++i) h = (h ^ k.a[i]) * 1099511628211ull; return (h ^ k.key) * 1099511628211ull; } };
        std::unordered_map<EdgeKey, std::vector<int64_t>, EdgeHash> by_edge;
        by_edge.reserve((size_t)F * 3);
        auto vx = [&](int64_t f, int k) { return verts + 2 * faces[3 * f + k]; };
        auto same = [&](const float *p, const float *q) { return memcmp(p, q, 8) == 0; };
        auto face_key = [&](int64_t f, uint32_t &key) {
            for (int l = 0; l < n_levels; ++l) if (levels[l] == face_z[f]) { key = ((uint32_t)(l + 1) << 24) | (face_rgb[f] & 0xFFFFFFu); return true; }
            return false;
        };
        auto usable = [&](int64_t f) {          // finite, three distinct points
            for (int k = 0; k < 3; ++k) if (!std::isfinite(vx(f, k)[0]) || !std::isfinite(vx(f, k)[1])) return false;
            return !same(vx(f, 0), vx(f, 1)) && !same(vx(f, 1), vx(f, 2)) && !same(vx(f, 0), vx(f, 2));
        };
        auto edge_key = [&](int64_t f, int l, uint32_t key) {
            const float *p = vx(f, l), *q = vx(f, (l + 1) % 3);
            if (p[0] > q[0] || (p[0] == q[0] && p[1] > q[1])) std::swap(p, q);
            EdgeKey ek;
            memcpy(&ek.a[0], p, 8); memcpy(&ek.a[2], q, 8);
            ek.key = key;
            return ek;
        };
        for (int64_t f = 0; f < F; ++f) {
            uint32_t key;
            if (!face_key(f, key) || !usable(f)) continue;
            for (int l = 0; l < 3; ++l) by_edge[edge_key(f, l, key)].push_back(f);
        }
        for (int64_t f = 0; f < F; ++f) {
            uint32_t key;
            if (partner[(size_t)f] >= 0 || !face_key(f, key) || !usable(f)) continue;
            int64_t best = -1;
            for (int l = 0; l < 3; ++l) {
                const float *apex = vx(f, (l + 2) % 3);
                for (int64_t g : by_edge[edge_key(f, l, key)]) {
                    if (g == f || partner[(size_t)g] >= 0 || (best >= 0 && g >= best)) continue;
                    // g's third vertex must be a fourth point
                    int shared = 0, other = -1;
                    for (int k = 0; k < 3; ++k) {
                        if (same(vx(g, k), vx(f, l)) || same(vx(g, k), vx(f, (l + 1) % 3))) ++shared; else other = k;
                    }
                    if (shared == 2 && other >= 0 && !same(vx(g, other), apex)) best = g;
                }
            }
            if (best >= 0) { partner[(size_t)f] = best; partner[(size_t)best] = f; ++n_pairs; }
        }
    }
    std::vector<int32_t> cell_start;
    std::vector<GridEntry> entries;
    std::vector<int32_t> qcell_start;
    std::vector<tds::QuadEntry> qentries;
    int nx = 0, ny = 0;
    float ox = minx, oy = miny, inv = 1.0f / cell;
    for (int attempt = 0; attempt < 24 && any; ++attempt) {
        inv = 1.0f / cell;
        nx = tds::cell_coord(maxx, ox, inv) + 1;
        ny = tds::cell_coord(maxy, oy, inv) + 1;
        bool too_big = nx > 8191 || ny > 8191 || (int64_t)nx * ny > (int64_t)(1 << 26);
        int64_t total = 0;
        if (!too_big) {
            cell_start.assign((size_t)nx * ny + 1, 0);
            for (int64_t f = 0; f < F && !too_big; ++f) {
                float fx0 = INFINITY, fy0 = INFINITY, fx1 = -INFINITY, fy1 = -INFINITY;
                bool ok = true;
                for (int k = 0; k < 3; ++k) {
                    float x = verts[2 * faces[3 * f + k]], y = verts[2 * faces[3 * f + k] + 1];
                    ok &= std::isfinite(x) && std::isfinite(y);
                    fx0 = std::min(fx0, x); fx1 = std::max(fx1, x); fy0 = std::min(fy0, y); fy1 = std::max(fy1, y);
                }
                if (!ok) continue;       // faces with non-finite vertices are never binned (never drawn, never nearest)
                int cx0 = tds::cell_coord(fx0, ox, inv), cx1 = tds::cell_coord(fx1, ox, inv);
                int cy0 = tds::cell_coord(fy0, oy, inv), cy1 = tds::cell_coord(fy1, oy, inv);
                total += (int64_t)(cx1 - cx0 + 1) * (cy1 - cy0 + 1);
                if (total > 24 * F + (1 << 20)) { too_big = true; break; }
                for (int cy = cy0; cy <= cy1; ++cy)
                    for (int cx = cx0; cx <= cx1; ++cx) cell_start[(size_t)cy * nx + cx + 1]++;
            }
        }
        if (too_big) { cell *= 2.0f; continue; }
        for (size_t i = 1; i < cell_start.size(); ++i) cell_start[i] += cell_start[i - 1];
        entries.resize((size_t)total);
        std::vector<int32_t> cursor(cell_start.begin(), cell_start.end() - 1);
        for (int64_t f = 0; f < F; ++f) {
            GridEntry e;
            const float *p0 = verts + 2 * faces[3 * f], *p1 = verts + 2 * faces[3 * f + 1], *p2 = verts + 2 * faces[3 * f + 2];
            e.x0 = p0[0]; e.y0 = p0[1]; e.x1 = p1[0]; e.y1 = p1[1]; e.x2 = p2[0]; e.y2 = p2[1];
            if (!(std::isfinite(e.x0) && std::isfinite(e.y0) && std::isfinite(e.x1) && std::isfinite(e.y1) && std::isfinite(e.x2) &&
                  std::isfinite(e.y2)))
                continue;
            e.key = 0;
            if (face_z) {
                int rank = -1;
                for (int l = 0; l < n_levels; ++l)
                    if (levels[l] == face_z[f]) { rank = l + 1; break; }
                TDS_CHECK_ARG(rank > 0, "tds_map_create: face %lld has level %g which is not in the level table", (long long)f, face_z[f]);
                e.key = ((uint32_t)rank << 24) | (face_rgb[f] & 0xFFFFFFu);
            }
            float fx0 = std::min(e.x0, std::min(e.x1, e.x2)), fx1 = std::max(e.x0, std::max(e.x1, e.x2));
            float fy0 = std::min(e.y0, std::min(e.y1, e.y2)), fy1 = std::max(e.y0, std::max(e.y1, e.y2));
            int cx0 = tds::cell_coord(fx0, ox, inv), cx1 = tds::cell_coord(fx1, ox, inv);
            int cy0 = tds::cell_coord(fy0, oy, inv), cy1 = tds::cell_coord(fy1, oy, inv);
            for (int cy = cy0; cy <= cy1; ++cy)
                for (int cx = cx0; cx <= cx1; ++cx) {
                    e.own = (uint32_t)cx0 | ((uint32_t)cx1 << 13) | (cx > cx0 ? (1u << 26) : 0u) | (cy > cy0 ? (1u << 27) : 0u) |
                            ((uint32_t)dup[(size_t)f] << 29);
                    entries[(size_t)cursor[(size_t)cy * nx + cx]++] = e;
                }
        }
        // Within a cell, large faces first: the nearest-face walk of K2b stops at the first face that contains the query point,
        // and a random point of a cell is most likely inside one of its large faces.  (No consumer depends on the order.)
        for (size_t c = 0; c + 1 < cell_start.size(); ++c)
            std::stable_sort(entries.begin() + cell_start[c], entries.begin() + cell_start[c + 1], [](const GridEntry &a, const GridEntry &b) {
                auto area2 = [](const GridEntry &g) { return std::fabs((g.x1 - g.x0) * (g.y2 - g.y0) - (g.x2 - g.x0) * (g.y1 - g.y0)); };
                return area2(a) > area2(b);
            });
        break;
    }
    if (!any) { nx = ny = 0; cell_start.assign(1, 0); entries.clear(); }
    // ---- the rendering grid with paired faces: the same cells, one entry per lone face or pair and cell of ITS bounding box
    if (face_z && any && nx > 0) {
        auto rank_of = [&](int64_t f) { for (int l = 0; l < n_levels; ++l) if (levels[l] == face_z[f]) return l + 1; return -1; };
        auto build = [&](int64_t f, tds::QuadEntry &q, float &bx0, float &by0, float &bx1, float &by1) {
            const float *p[3] = {verts + 2 * faces[3 * f], verts + 2 * faces[3 * f + 1], verts + 2 * faces[3 * f + 2]};
            for (int k = 0; k < 3; ++k) if (!std::isfinite(p[k][0]) || !std::isfinite(p[k][1])) return false;
            q.x0 = p[0][0]; q.y0 = p[0][1]; q.x1 = p[1][0]; q.y1 = p[1][1]; q.x2 = p[2][0]; q.y2 = p[2][1];
            q.x3 = q.x2; q.y3 = q.y2;
            q.key = ((uint32_t)rank_of(f) << 24) | (face_rgb[f] & 0xFFFFFFu);
            q.flags = (uint32_t)dup[(size_t)f] << 9;
            q.pad = 0;
            bx0 = std::min(q.x0, std::min(q.x1, q.x2)); bx1 = std::max(q.x0, std::max(q.x1, q.x2));
            by0 = std::min(q.y0, std::min(q.y1, q.y2)); by1 = std::max(q.y0, std::max(q.y1, q.y2));
            const int64_t g = partner[(size_t)f];
            if (g >= 0) {
                uint32_t b = 0, in_t2 = 0;
                for (int k = 0; k < 3; ++k) {
                    const float *v = verts + 2 * faces[3 * g + k];
                    int slot = 3;
                    for (int j = 0; j < 3; ++j) if (memcmp(v, p[j], 8) == 0) slot = j;
                    if (slot == 3) { q.x3 = v[0]; q.y3 = v[1]; } else in_t2 |= 1u << slot;
                    b |= (uint32_t)slot << (2 * k);
                }
                const uint32_t a1 = !(in_t2 & 1u) ? 0u : (!(in_t2 & 2u) ? 1u : 2u);
                q.flags |= b | (a1 << 6) | (1u << 8) | ((uint32_t)dup[(size_t)g] << 12);
                bx0 = std::min(bx0, q.x3); bx1 = std::max(bx1, q.x3); by0 = std::min(by0, q.y3); by1 = std::max(by1, q.y3);
            }
            return true;
        };
        qcell_start.assign((size_t)nx * ny + 1, 0);
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<int32_t> cursor;
            if (pass == 1) {
                for (size_t i = 1; i < qcell_start.size(); ++i) qcell_start[i] += qcell_start[i - 1];
                qentries.resize((size_t)qcell_start.back());
                cursor.assign(qcell_start.begin(), qcell_start.end() - 1);
            }
            for (int64_t f = 0; f < F; ++f) {
                if (partner[(size_t)f] >= 0 && partner[(size_t)f] < f) continue;       // the second half of a pair: in its partner's entry
                if (rank_of(f) < 0) continue;
                tds::QuadEntry q;
                float bx0, by0, bx1, by1;
                if (!build(f, q, bx0, by0, bx1, by1)) continue;
                const int cx0 = tds::cell_coord(bx0, ox, inv), cx1 = tds::cell_coord(bx1, ox, inv);
                const int cy0 = tds::cell_coord(by0, oy, inv), cy1 = tds::cell_coord(by1, oy, inv);
                for (int cy = cy0; cy <= cy1; ++cy)
                    for (int cx = cx0; cx <= cx1; ++cx) {
                        if (pass == 0) { qcell_start[(size_t)cy * nx + cx + 1]++; continue; }
                        q.own = (uint32_t)cx0 | ((uint32_t)cx1 << 13) | (cx > cx0 ? (1u << 26) : 0u) | (cy > cy0 ? (1u << 27) : 0u);
                        qentries[(size_t)cursor[(size_t)cy * nx + cx]++] = q;
                    }
            }
        }
    }
    if (qcell_start.empty()) qcell_start.assign((size_t)std::max(nx, 0) * std::max(ny, 0) + 1, 0);

    tds_map *m = new (std::nothrow) tds_map();
    if (!m) { tds::set_error("tds_map_create: out of host memory"); return TDS_ENOMEM; }
    if (entries.size() >= ((size_t)1 << 25)) {            // the raster scan packs an entry index and its grid row into one register (raster.hip: scan_fetch)
        delete m;
        tds::set_error("tds_map_create: %zu grid entries (limit 2^25): use a coarser cell size", entries.size());
        return TDS_ELIMIT;
    }
    m->V = V; m->F = F; m->n_entries = (int64_t)entries.size(); m->n_levels = face_z ? n_levels : 0;
    m->d_entries = nullptr; m->d_cell_start = nullptr;
    m->near = tds::NearView{nullptr, nullptr, nullptr, 0.0f, 0.0f, 0, 0, nullptr, nullptr}; m->n_cand = 0;
    // geometry-only maps are the ones K2b queries: give them nearest-face candidate lists
    std::vector<int32_t> cand_start;
    std::vector<tds::NearCand> cand;
    std::vector<GridEntry> face_entries;
    const bool with_near = !face_z && any && F > 0 && nx > 0 && (int64_t)nx * ny <= (int64_t)(1 << 21) && TDS_NEAR_LISTS_ENABLED;
    float near_ox = 0, near_oy = 0;
    int near_nx = 0, near_ny = 0;
    if (with_near) {
        build_near_lists(verts, faces, F, ox, oy, cell, nx, ny, cand_start, cand, face_entries, near_ox, near_oy, near_nx, near_ny);
        if (cand.size() > ((size_t)1 << 28)) { cand.clear(); cand_start.clear(); }
    }
    std::vector<tds::BvhNode> bvh_nodes;
    std::vector<int32_t> bvh_idx;
    if (with_near && !cand.empty() && TDS_BVH_BUILD) {
        // the kernels' descent keeps a fixed stack per point: a hierarchy deeper than it can hold is not attached (the query then walks the
        // grid rings, slower and exact).  F < 2^24 and eight-way nodes over leaves of up to 8 faces keep the depth below 8.
        const int depth = build_bvh(verts, faces, F, bvh_nodes, bvh_idx);
        if (!tds::bvh_fits_stack(depth)) { bvh_nodes.clear(); bvh_idx.clear(); }
    }
    hipError_t e = hipGetDevice(&m->device);
    size_t be = std::max<size_t>(entries.size(), 1) * sizeof(GridEntry), bc = cell_start.size() * sizeof(int32_t);
    if (e == hipSuccess) e = hipMalloc(&m->d_entries, be);
    if (e == hipSuccess) e = hipMalloc(&m->d_cell_start, bc);
    if (e == hipSuccess && !entries.empty()) e = hipMemcpy(m->d_entries, entries.data(), entries.size() * sizeof(GridEntry), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->d_cell_start, cell_start.data(), bc, hipMemcpyHostToDevice);
    m->d_qentries = nullptr; m->d_qcell_start = nullptr;
    m->n_qentries = (int64_t)qentries.size(); m->n_pairs = n_pairs;
    size_t bq = 0;
    if (face_z) {
        const size_t bqe = std::max<size_t>(qentries.size(), 1) * sizeof(tds::QuadEntry), bqc = qcell_start.size() * sizeof(int32_t);
        if (e == hipSuccess) e = hipMalloc(&m->d_qentries, bqe);
        if (e == hipSuccess) e = hipMalloc(&m->d_qcell_start, bqc);
        if (e == hipSuccess && !qentries.empty()) e = hipMemcpy(m->d_qentries, qentries.data(), qentries.size() * sizeof(tds::QuadEntry), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(m->d_qcell_start, qcell_start.data(), bqc, hipMemcpyHostToDevice);
        bq = bqe + bqc;
    }
    size_t bn = 0;
    if (e == hipSuccess && !cand.empty()) {
        void *dc = nullptr, *ds = nullptr, *df = nullptr;
        const size_t b1 = cand.size() * sizeof(tds::NearCand), b2 = cand_start.size() * sizeof(int32_t), b3 = face_entries.size() * sizeof(GridEntry);
        e = hipMalloc(&dc, b1);
        if (e == hipSuccess) e = hipMalloc(&ds, b2);
        if (e == hipSuccess) e = hipMalloc(&df, b3);
        if (e == hipSuccess) e = hipMemcpy(dc, cand.data(), b1, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ds, cand_start.data(), b2, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(df, face_entries.data(), b3, hipMemcpyHostToDevice);
        m->near = tds::NearView{(const tds::NearCand *)dc, (const int32_t *)ds, (const GridEntry *)df, near_ox, near_oy, near_nx, near_ny, nullptr, nullptr};
        m->n_cand = (int64_t)cand.size();
        bn = b1 + b2 + b3;
        if (e == hipSuccess && !bvh_nodes.empty()) {
            void *dn = nullptr, *di = nullptr;
            const size_t b4 = bvh_nodes.size() * sizeof(tds::BvhNode), b5 = bvh_idx.size() * sizeof(int32_t);
            e = hipMalloc(&dn, b4);
            if (e == hipSuccess) e = hipMalloc(&di, b5);
            if (e == hipSuccess) e = hipMemcpy(dn, bvh_nodes.data(), b4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(di, bvh_idx.data(), b5, hipMemcpyHostToDevice);
            m->near.bvh = (const tds::BvhNode *)dn; m->near.bvh_idx = (const int32_t *)di;
            bn += b4 + b5;
        }
    }
    if (e != hipSuccess) {
        tds::set_error("tds_map_create: %s", hipGetErrorString(e));
        if (m->d_entries) (void)hipFree(m->d_entries);
        if (m->d_cell_start) (void)hipFree(m->d_cell_start);
        if (m->d_qentries) (void)hipFree(m->d_qentries);
        if (m->d_qcell_start) (void)hipFree(m->d_qcell_start);
        if (m->near.cand) (void)hipFree((void *)m->near.cand);
        if (m->near.cand_start) (void)hipFree((void *)m->near.cand_start);
        if (m->near.faces) (void)hipFree((void *)m->near.faces);
        if (m->near.bvh) (void)hipFree((void *)m->near.bvh);
        if (m->near.bvh_idx) (void)hipFree((void *)m->near.bvh_idx);
        delete m;
        return e == hipErrorOutOfMemory ? TDS_ENOMEM : TDS_EHIP;
    }
    m->n_uniq = 0;
    for (const GridEntry &ge : entries) {
        bool seen = false;
        for (int i = 0; i < m->n_uniq && !seen; ++i) seen = m->uniq_keys[i] == ge.key;
        if (seen) continue;
        if (m->n_uniq == 64) { m->n_uniq = -1; break; }
        m->uniq_keys[m->n_uniq++] = ge.key;
    }
    m->bytes = (int64_t)(be + bc + bn + bq);
    m->view.entries = (const GridEntry *)m->d_entries;
    m->view.cell_start = (const int32_t *)m->d_cell_start;
    m->view.qentries = (const tds::QuadEntry *)m->d_qentries;
    m->view.qcell_start = (const int32_t *)m->d_qcell_start;
    m->view.ox = ox; m->view.oy = oy; m->view.inv_cell = inv; m->view.cell = cell;
    m->view.nx = nx; m->view.ny = ny; m->view.n_faces = F;
    *out = m;
    return TDS_OK;
}

TDS_EXPORT int tds_map_destroy(tds_map_t *map) {
    if (!map) return TDS_OK;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != map->device) (void)hipSetDevice(map->device);
    hipError_t e1 = hipFree(map->d_entries), e2 = hipFree(map->d_cell_start);
    if (map->d_qentries) (void)hipFree(map->d_qentries);
    if (map->d_qcell_start) (void)hipFree(map->d_qcell_start);
    if (map->near.cand) (void)hipFree((void *)map->near.cand);
    if (map->near.cand_start) (void)hipFree((void *)map->near.cand_start);
    if (map->near.faces) (void)hipFree((void *)map->near.faces);
    if (map->near.bvh) (void)hipFree((void *)map->near.bvh);
    if (map->near.bvh_idx) (void)hipFree((void *)map->near.bvh_idx);
    if (cur != map->device) (void)hipSetDevice(cur);
    delete map;
    if (e1 != hipSuccess || e2 != hipSuccess) { tds::set_error("tds_map_destroy: hipFree failed"); return TDS_EHIP; }
    return TDS_OK;
}

TDS_EXPORT int tds_map_keys(const tds_map_t *map, uint32_t *keys, int cap, int *n) {
    TDS_CHECK_ARG(map && n && (keys || cap == 0) && cap >= 0, "tds_map_keys: bad arguments");
    *n = map->n_uniq;
    for (int i = 0; i < map->n_uniq && i < cap; ++i) keys[i] = map->uniq_keys[i];
    return TDS_OK;
}

TDS_EXPORT int tds_map_info_ex(const tds_map_t *map, int64_t *info, int n_words) {
    TDS_CHECK_ARG(map && info && n_words >= 0, "tds_map_info_ex: bad arguments");
    const int64_t all[TDS_MAP_INFO_WORDS] = {map->V, map->F, map->view.nx, map->view.ny, map->n_entries, map->bytes, map->n_levels, map->n_cand,
                                             map->n_qentries, map->n_pairs};
    for (int i = 0; i < n_words; ++i) info[i] = i < TDS_MAP_INFO_WORDS ? all[i] : 0;
    return TDS_OK;
}

// eight words, as every caller built against the first header expects (round 5 wrote ten through the same symbol: ADVICE r5)
TDS_EXPORT int tds_map_info(const tds_map_t *map, int64_t *info) { return tds_map_info_ex(map, info, 8); }

// ------------------------------------------------------------------------------------------------------------
// Which scenes of a collated batch share a mesh (mesh.py:172-200 pads every element of a collated batch to the largest one, so the rows
// of two scenes on the same map are identical byte for byte): a 64-bit content hash per batch row and an exact row comparison, so that
// the host builds ONE device map per distinct mesh instead of one per scene.  Both read every row once, at the rate of the HBM stream.
// ------------------------------------------------------------------------------------------------------------
namespace {
constexpr int ROW_CHUNK_WORDS = 8192;          // 4-byte words of a row per workgroup
__device__ __forceinline__ uint64_t mix64(uint64_t z) {          // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
// out[row] += sum over the row's words of mix64(word, position): a sum, so the order in which workgroups arrive does not matter
__global__ void __launch_bounds__(256) rows_hash_kernel(const uint32_t *__restrict__ rows, int64_t row_words, int64_t stride_words, uint64_t seed,
                                                       unsigned long long *__restrict__ out) {
    const int64_t row = blockIdx.y;
    const uint32_t *p = rows + row * stride_words;
    const int64_t w0 = (int64_t)blockIdx.x * ROW_CHUNK_WORDS, w1 = min(w0 + ROW_CHUNK_WORDS, row_words);
    uint64_t h = 0;
    for (int64_t i = w0 + threadIdx.x; i < w1; i += 256) h += mix64(((uint64_t)p[i] | ((uint64_t)(uint32_t)i << 32)) + seed + (uint64_t)(i >> 32));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) h += __shfl_xor((unsigned long long)h, d);
    if ((threadIdx.x & 63) == 0 && h != 0) atomicAdd(out + row, (unsigned long long)h);
}
// equal[row] = 0 where row differs from row rep[row] (the caller presets 1)
__global__ void __launch_bounds__(256) rows_equal_kernel(const uint32_t *__restrict__ rows, int64_t row_words, int64_t stride_words, int64_t row0,
                                                        const int32_t *__restrict__ rep, uint8_t *__restrict__ equal) {
    const int64_t row = row0 + blockIdx.y, other = rep[row];
    if (other == row) return;
    const uint32_t *p = rows + row * stride_words, *q = rows + other * stride_words;
    const int64_t w0 = (int64_t)blockIdx.x * ROW_CHUNK_WORDS, w1 = min(w0 + ROW_CHUNK_WORDS, row_words);
    bool diff = false;
    for (int64_t i = w0 + threadIdx.x; i < w1; i += 256) diff = diff || (p[i] != q[i]);
    if (diff) equal[row] = 0;
}
}  // namespace

TDS_EXPORT int tds_rows_hash_u64(const void *rows, int64_t n_rows, int64_t row_bytes, int64_t row_stride_bytes, uint64_t seed, uint64_t *out, void *stream) {
    TDS_CHECK_ARG(n_rows >= 0 && n_rows < 65536 * 16 && row_bytes >= 0 && (row_bytes & 3) == 0 && (row_stride_bytes & 3) == 0 && row_stride_bytes >= 0,
                  "tds_rows_hash_u64: rows are whole 4-byte words, at most 2^20 of them");
    TDS_CHECK_ARG(n_rows == 0 || (out && (rows || row_bytes == 0)), "tds_rows_hash_u64: null pointer");
    TDS_CHECK_ARG(((uintptr_t)rows & 3) == 0 && ((uintptr_t)out & 7) == 0, "tds_rows_hash_u64: misaligned pointer");
    if (n_rows == 0) return TDS_OK;
    if (tds::zero_async(out, (size_t)n_rows * 8, (hipStream_t)stream) != hipSuccess) { tds::set_error("tds_rows_hash_u64: clearing the output failed"); return TDS_EHIP; }
    if (row_bytes == 0) return TDS_OK;
    const int64_t words = row_bytes / 4, chunks = (words + ROW_CHUNK_WORDS - 1) / ROW_CHUNK_WORDS;
    for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {                              // gridDim.y <= 65535
        const int64_t nr = std::min<int64_t>(65535, n_rows - r0);
        hipLaunchKernelGGL(rows_hash_kernel, dim3((unsigned)chunks, (unsigned)nr), dim3(256), 0, (hipStream_t)stream,
                           (const uint32_t *)rows + r0 * (row_stride_bytes / 4), words, row_stride_bytes / 4, seed, (unsigned long long *)out + r0);
    }
    TDS_LAUNCH_CHECK("rows_hash_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_rows_equal_u8(const void *rows, int64_t n_rows, int64_t row_bytes, int64_t row_stride_bytes, const int32_t *rep, uint8_t *equal, void *stream) {
    TDS_CHECK_ARG(n_rows >= 0 && n_rows < 65536 * 16 && row_bytes >= 0 && (row_bytes & 3) == 0 && (row_stride_bytes & 3) == 0 && row_stride_bytes >= 0,
                  "tds_rows_equal_u8: rows are whole 4-byte words, at most 2^20 of them");
    TDS_CHECK_ARG(n_rows == 0 || (rep && equal && (rows || row_bytes == 0)), "tds_rows_equal_u8: null pointer");
    TDS_CHECK_ARG(((uintptr_t)rows & 3) == 0, "tds_rows_equal_u8: misaligned pointer");
    if (n_rows == 0 || row_bytes == 0) return TDS_OK;
    const int64_t words = row_bytes / 4, chunks = (words + ROW_CHUNK_WORDS - 1) / ROW_CHUNK_WORDS;
    for (int64_t r0 = 0; r0 < n_rows; r0 += 65535)                                // gridDim.y <= 65535
        hipLaunchKernelGGL(rows_equal_kernel, dim3((unsigned)chunks, (unsigned)std::min<int64_t>(65535, n_rows - r0)), dim3(256), 0, (hipStream_t)stream,
                           (const uint32_t *)rows, words, row_stride_bytes / 4, r0, rep, equal);
    TDS_LAUNCH_CHECK("rows_equal_kernel");
    return TDS_OK;
}

// ------------------------------------------------------------------------------------------------------------
// device: offroad
// ------------------------------------------------------------------------------------------------------------
namespace {

constexpr int OBLOCK = 256;

__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) { return (ax * bx + ay * by) + 0.0f; }

// point_line_distance, infractions.py:147-159 (squared)
__device__ __forceinline__ float seg_d2(float px, float py, float ax, float ay, float bx, float by) {
    float ex = bx - ax, ey = by - ay;
    float l2 = dot2(ex, ey, ex, ey);
    float t = dot2(ex, ey, px - ax, py - ay) / (l2 + (float)1e-8);
    float tt = fminf(fmaxf(t, 0.0f), 1.0f);
    tt = (t != t) ? t : tt;                                  // torch.clamp propagates NaN
    float qx = ax + tt * ex, qy = ay + tt * ey;
    float d = dot2(px - qx, py - qy, px - qx, py - qy);
    if (l2 <= (float)1e-8) d = dot2(px - bx, py - by, px - bx, py - by);
    return d;
}

// point_to_mesh_distance_pt for one (point, triangle) in the z = 0 plane, infractions.py:100-170
__device__ __forceinline__ float tri_d2(float px, float py, const GridEntry &e) {
    float cz = (e.x2 - e.x0) * (e.y1 - e.y0) - (e.y2 - e.y0) * (e.x1 - e.x0);
    float norm_normal = sqrtf(cz * cz);
    float p0x = e.x1 - e.x0, p0y = e.y1 - e.y0, p1x = e.x2 - e.x0, p1y = e.y2 - e.y0;
    float p2x = px - e.x0, p2y = py - e.y0;
    float d00 = dot2(p0x, p0y, p0x, p0y), d01 = dot2(p0x, p0y, p1x, p1y), d11 = dot2(p1x, p1y, p1x, p1y);
    float d20 = dot2(p2x, p2y, p0x, p0y), d21 = dot2(p2x, p2y, p1x, p1y);
    float denom = d00 * d11 - d01 * d01 + (float)1e-8;
    float w1 = (d11 * d20 - d01 * d21) / denom;
    float w2 = (d00 * d21 - d01 * d20) / denom;
    float w0 = 1.0f - w1 - w2;
    bool inside = (0.0f <= w0) && (w0 <= 1.0f) && (0.0f <= w1) && (w1 <= 1.0f) && (0.0f <= w2) && (w2 <= 1.0f);
    float area = fabsf(p0x * p1y - p0y * p1x) / 2.0f;
    inside = inside && !(area < (float)5e-3) && (norm_normal > (float)1e-8);
    float e01 = seg_d2(px, py, e.x0, e.y0, e.x1, e.y1);
    float e02 = seg_d2(px, py, e.x0, e.y0, e.x2, e.y2);
    float e12 = seg_d2(px, py, e.x1, e.y1, e.x2, e.y2);
    float dist = fminf(fminf(e01, e02), e12);
    return inside ? 0.0f : dist;
}

// A point is served by a GROUP of OL consecutive lanes (all of them hold the same point); `sub` is the lane's place in its group.
#ifndef TDS_OL
#define TDS_OL 8
#endif
constexpr int OL = TDS_OL;        // lanes per corner (a power of two, <= 16: a wavefront holds whole agents)
__device__ __forceinline__ float group_min(float v) {
#pragma unroll
    for (int d = 1; d < OL; d <<= 1) v = fminf(v, __shfl_xor(v, d));
    return v;
}

// min over all faces of tri_d2, found by walking grid rings outwards from the point's cell.  A face is listed in every
// cell its bounding box touches, so its closest point lies in a listed cell whose box is at least as close as the face:
// cells whose box is not closer than the best distance so far can be skipped, and the walk stops once a whole ring is.
// The lanes of the group share the entries of a cell (entry i + sub of every round) and agree on the minimum after every cell; the minimum
// over faces does not depend on the order in which they are looked at, so the result is that of a sequential walk, bit for bit.
// `stop`: the caller only needs the exact minimum if it exceeds `stop` (F.threshold zeroes everything else), so the walk ends as soon
// as the running minimum is <= stop.
__device__ __forceinline__ float nearest_face_d2(const MapView &m, float px, float py, float stop, int sub) {
    const float inf = __builtin_inff();
    float best = inf;                                        // the group's minimum so far: the same in all its lanes
    if (m.nx <= 0 || !(px == px) || !(py == py) || __builtin_isinf(px) || __builtin_isinf(py)) return best;
    // unclamped cell of the point (float -> int conversion saturates, keep it in a sane range first)
    float fx = fminf(fmaxf((px - m.ox) * m.inv_cell, -1.0e6f), 1.0e6f), fy = fminf(fmaxf((py - m.oy) * m.inv_cell, -1.0e6f), 1.0e6f);
    int cx = (int)floorf(fx), cy = (int)floorf(fy);
    auto visit = [&](int x, int y) {
        // squared distance from the point to the cell's box, shrunk a little so that rounding can only make us visit more
        float bx0 = m.ox + (float)x * m.cell, by0 = m.oy + (float)y * m.cell;
        float ddx = fmaxf(fmaxf(bx0 - px, px - (bx0 + m.cell)), 0.0f), ddy = fmaxf(fmaxf(by0 - py, py - (by0 + m.cell)), 0.0f);
        float cd = (ddx * ddx + ddy * ddy) * 0.998f - 1e-3f;
        if (cd >= best || best <= stop) return;
        const int s = m.cell_start[y * m.nx + x], e = m.cell_start[y * m.nx + x + 1];
        float mine = best;
        for (int i = s + sub; i < e; i += OL) {
            const GridEntry ge = m.entries[i];
            // the face cannot beat the minimum if even its bounding box is farther (same safety shrink as for the cell)
            float fx0 = fminf(ge.x0, fminf(ge.x1, ge.x2)), fx1 = fmaxf(ge.x0, fmaxf(ge.x1, ge.x2));
            float fy0 = fminf(ge.y0, fminf(ge.y1, ge.y2)), fy1 = fmaxf(ge.y0, fmaxf(ge.y1, ge.y2));
            float ex = fmaxf(fmaxf(fx0 - px, px - fx1), 0.0f), ey = fmaxf(fmaxf(fy0 - py, py - fy1), 0.0f);
            if ((ex * ex + ey * ey) * 0.998f - 1e-3f >= mine) continue;
            float d = tri_d2(px, py, ge);
            mine = (d < mine) ? d : mine;                    // a NaN distance never becomes the minimum
        }
        best = group_min(mine);
    };
    int k = max(max(0, max(-cx, cx - (m.nx - 1))), max(-cy, cy - (m.ny - 1)));
    for (;; ++k) {
        // cells at Chebyshev distance exactly k from (cx, cy), clipped to the grid
        int y0 = max(cy - k, 0), y1 = min(cy + k, m.ny - 1);
        int x0 = max(cx - k, 0), x1 = min(cx + k, m.nx - 1);
        for (int y = y0; y <= y1; ++y) {
            if (y == cy - k || y == cy + k) {
                for (int x = x0; x <= x1; ++x) visit(x, y);
            } else {
                if (cx - k >= 0 && cx - k < m.nx) visit(cx - k, y);
                if (k > 0 && cx + k >= 0 && cx + k < m.nx) visit(cx + k, y);
            }
        }
        if (best <= stop) break;
        float bound = (float)k * m.cell * 0.999f;            // everything unvisited is at least this far away
        if (best <= bound * bound) break;
        if (cx - k <= 0 && cx + k >= m.nx - 1 && cy - k <= 0 && cy + k >= m.ny - 1) break;
    }
    return best;
}

// The same minimum by an ordered descent of the hierarchy over the faces (tds::BvhNode): for the points the candidate lists do not cover.
// The group's lanes hold the same point.  At an inner node every lane weighs ONE of the eight children (a 16-byte box each: one 128-byte
// line for the group); the children whose box can still beat the minimum go onto the group's stack (in LDS) farthest first -- a lane finds
// its place by comparing its bound with the other seven -- and the nearest is taken up at once; at a leaf a face per lane.  Boxes are
// shrunk like everywhere in K2b (x 0.998 - 1e-3): a subtree is left out only if every face in it is certainly farther than the minimum so
// far, so the result is the minimum over ALL faces, bit for bit, and `stop` ends the descent as in the other walks.  (Round 5 began with
// two children per node, both weighed by every lane: three times the depth, 0.60 ms for 65 536 strayed agents.)
static_assert(OL == 8, "nearest_face_d2_bvh: a lane per child of a node");
using tds::BVH_STACK;                   // (tds_common.h; tds_map_create refuses to attach a hierarchy the stack cannot hold)
__device__ __forceinline__ float box_lb(float px, float py, float x0, float y0, float x1, float y1) {
    const float ex = fmaxf(fmaxf(x0 - px, px - x1), 0.0f), ey = fmaxf(fmaxf(y0 - py, py - y1), 0.0f);
    return (ex * ex + ey * ey) * 0.998f - 1e-3f;
}
__device__ float nearest_face_d2_bvh(const tds::NearView &nv, float px, float py, float stop, int sub) {
    __shared__ int2 stacks[OBLOCK / OL][BVH_STACK];
    int2 *st = stacks[threadIdx.x / OL];
    const float inf = __builtin_inff();
    const int shift = (int)(threadIdx.x & 63 & ~(OL - 1));                              // the group's first lane within the wavefront
    float best = inf;
    int sp = 0, cur = 0;
    for (;;) {
        if (cur >= 0) {
            const float4 bx = nv.bvh[cur].box[sub];
            const int ch = nv.bvh[cur].child[sub];
            const float lb = box_lb(px, py, bx.x, bx.y, bx.z, bx.w);
            const bool open = lb < best;
            const float key = open ? lb : inf;
            int above = 0;                                                             // open children that will lie above this lane's on the stack
#pragma unroll
            for (int k = 1; k < OL; ++k) {
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
            if (sub < cnt) {
                const GridEntry ge = nv.faces[nv.bvh_idx[first + sub]];
                const float fx0 = fminf(ge.x0, fminf(ge.x1, ge.x2)), fx1 = fmaxf(ge.x0, fmaxf(ge.x1, ge.x2));
                const float fy0 = fminf(ge.y0, fminf(ge.y1, ge.y2)), fy1 = fmaxf(ge.y0, fmaxf(ge.y1, ge.y2));
                if (!(box_lb(px, py, fx0, fy0, fx1, fy1) >= best)) {
                    const float t = tri_d2(px, py, ge);
                    d = (t < inf) ? t : inf;                                          // a NaN distance never becomes the minimum
                }
            }
            best = fminf(best, group_min(d));
        }
        // the next subtree that can still hold a nearer face
        float lb;
        do {
            if (sp == 0 || best <= stop) return best;
            --sp;
            cur = st[sp].x; lb = __int_as_float(st[sp].y);
        } while (lb >= best);
    }
}

// The same minimum from the candidate list of the point's cell (tds::NearView), walked by OL lanes together: candidate i + lane of every
// round, the group's running minimum by three xor-shuffles, ended by the first round whose first candidate has a lower bound that is not
// below it (the list is sorted by lower bound) or as soon as the minimum is <= stop.  The minimum over faces does not depend on the order
// in which they are looked at, so the result is the sequential walk's, bit for bit.  One lane per point (the first version) was a chain of
// dependent loads, candidate -> face -> next candidates, a few to a few dozen rounds long: 0.148 ms at B = 1024 x 64 however little
// arithmetic it is; eight lanes per point: one or two rounds, 0.09 ms.  Points outside the grid take the ring walk.
__device__ __forceinline__ float nearest_face_d2_lists(const MapView &m, const tds::NearView &nv, float px, float py, float stop, int sub) {
    const float inf = __builtin_inff();
    bool ring = nv.cand == nullptr || m.nx <= 0 || !(px == px) || !(py == py) || __builtin_isinf(px) || __builtin_isinf(py);
    int cx = 0, cy = 0;
    if (!ring) {
        const float fx = (px - nv.ox) * m.inv_cell, fy = (py - nv.oy) * m.inv_cell;
        ring = !(fx >= 0.0f && fy >= 0.0f && fx < (float)nv.nx && fy < (float)nv.ny);
        if (!ring) {
            cx = tds::cell_coord(px, nv.ox, m.inv_cell); cy = tds::cell_coord(py, nv.oy, m.inv_cell);
            ring = cx < 0 || cy < 0 || cx >= nv.nx || cy >= nv.ny;
        }
    }
    if (ring) {                                                                         // (uniform within the group: all its lanes hold the same point)
        // beyond the lists' grid: the hierarchy over the faces where the map has one, else the walk over grid rings
        if (nv.bvh != nullptr && px == px && py == py && !__builtin_isinf(px) && !__builtin_isinf(py)) return nearest_face_d2_bvh(nv, px, py, stop, sub);
        return nearest_face_d2(m, px, py, stop, sub);
    }
    const int s = nv.cand_start[cy * nv.nx + cx], e = nv.cand_start[cy * nv.nx + cx + 1];
    float best = inf;                                                                 // the group's minimum so far, the same in all its lanes
    for (int i = s; i < e && best > stop; i += OL) {
        const int j = i + sub;
        tds::NearCand c;
        c.face = 0; c.lb = inf;
        if (j < e) c = nv.cand[j];
        const float lb0 = __shfl(c.lb, (threadIdx.x & 63 & ~(OL - 1)));                // the round's first candidate
        if (lb0 >= best) break;                                                       // sorted by lb: nothing further can be nearer
        float d = inf;
        if (c.lb < best) {
            const GridEntry ge = nv.faces[c.face];
            const float fx0 = fminf(ge.x0, fminf(ge.x1, ge.x2)), fx1 = fmaxf(ge.x0, fmaxf(ge.x1, ge.x2));
            const float fy0 = fminf(ge.y0, fminf(ge.y1, ge.y2)), fy1 = fmaxf(ge.y0, fmaxf(ge.y1, ge.y2));
            const float ex = fmaxf(fmaxf(fx0 - px, px - fx1), 0.0f), ey = fmaxf(fmaxf(fy0 - py, py - fy1), 0.0f);
            if (!((ex * ex + ey * ey) * 0.998f - 1e-3f >= best)) {
                const float t = tri_d2(px, py, ge);
                d = (t < inf) ? t : inf;                                              // a NaN distance never becomes the minimum (as `d < best ? d : best`)
            }
        }
        best = fminf(best, group_min(d));
    }
    return best;
}

// OL lanes per agent corner, 4 consecutive groups = one agent
__global__ void __launch_bounds__(OBLOCK) offroad_kernel(MapView m, tds::NearView nv, const float4 *__restrict__ state, const float2 *__restrict__ lenwid,
                                                         const float2 *__restrict__ sc, const uint8_t *__restrict__ present,
                                                         float *__restrict__ out, int64_t n, float threshold, const MapView *__restrict__ views,
                                                         const tds::NearView *__restrict__ nears, const int32_t *__restrict__ scene_map, int agents_per_scene) {
    const int64_t t = (int64_t)blockIdx.x * OBLOCK + threadIdx.x;
    const int64_t a = t / (4 * OL);
    const int k = (int)((t / OL) & 3), sub = (int)(t & (OL - 1));
    float v = 0.0f;
    if (a < n && views != nullptr) {                                                // one map per scene (tds_offroad_multi_f32)
        const int im = scene_map[a / agents_per_scene];
        m = views[im]; nv = nears[im];
    }
    if (a < n && m.n_faces > 0) {
        float4 s = state[a];
        float2 lw = lenwid[a];
        float2 scv = sc[a];
        const float sx = (k == 0 || k == 3) ? 0.5f : -0.5f, sy = (k < 2) ? 0.5f : -0.5f;     // box2corners_th :285-288
        float x4 = sx * lw.x, y4 = sy * lw.y;
        float px = (x4 * scv.y + y4 * (-scv.x)) + s.x;
        float py = (x4 * scv.x + y4 * scv.y) + s.y;
        float d = nearest_face_d2_lists(m, nv, px, py, fmaxf(threshold, 0.0f), sub);
        d = (d != d) ? 0.0f : d;                                     // nan_to_num :171
        if (__builtin_isinf(d)) d = 3.4028234663852886e38f;
        v = (d > threshold) ? d : 0.0f;                              // F.threshold(d, thr, 0) :172
    }
    // sum of the 4 corners in order (infractions.py:228)
    float v1 = __shfl_down(v, OL), v2 = __shfl_down(v, 2 * OL), v3 = __shfl_down(v, 3 * OL);
    if (a < n && k == 0 && sub == 0) {
        float tot = ((v + v1) + v2) + v3;
        if (present) tot = tot * (present[a] ? 1.0f : 0.0f);         // simulator.py:1044
        out[a] = tot;
    }
}

}  // namespace

TDS_EXPORT int tds_offroad_f32(const tds_map_t *map, const float *state, const float *lenwid, const float *sc, const uint8_t *present,
                               float *out, int64_t n_agents, float threshold, void *stream) {
    TDS_CHECK_ARG(map, "tds_offroad_f32: null map");
    TDS_CHECK_ARG(n_agents >= 0 && n_agents < ((int64_t)1 << 34), "tds_offroad_f32: bad agent count");
    if (n_agents == 0) return TDS_OK;
    TDS_CHECK_ARG(state && lenwid && sc && out, "tds_offroad_f32: null pointer");
    int64_t threads = n_agents * 4 * OL;
    hipLaunchKernelGGL(offroad_kernel, dim3((unsigned)((threads + OBLOCK - 1) / OBLOCK)), dim3(OBLOCK), 0, (hipStream_t)stream, map->view, map->near,
                       (const float4 *)state, (const float2 *)lenwid, (const float2 *)sc, present, out, n_agents, threshold,
                       (const MapView *)nullptr, (const tds::NearView *)nullptr, (const int32_t *)nullptr, 1);
    TDS_LAUNCH_CHECK("offroad_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_offroad_multi_f32(const tds_mapset_t *set, const int32_t *scene_map, int64_t agents_per_scene, const float *state, const float *lenwid,
                                     const float *sc, const uint8_t *present, float *out, int64_t n_agents, float threshold, void *stream) {
    TDS_CHECK_ARG(set && set->n > 0 && scene_map, "tds_offroad_multi_f32: null map set or scene index array");
    TDS_CHECK_ARG(agents_per_scene > 0 && agents_per_scene < (1 << 30), "tds_offroad_multi_f32: bad number of agents per scene");
    TDS_CHECK_ARG(n_agents >= 0 && n_agents < ((int64_t)1 << 34), "tds_offroad_multi_f32: bad agent count");
    if (n_agents == 0) return TDS_OK;
    TDS_CHECK_ARG(state && lenwid && sc && out, "tds_offroad_multi_f32: null pointer");
    int64_t threads = n_agents * 4 * OL;
    hipLaunchKernelGGL(offroad_kernel, dim3((unsigned)((threads + OBLOCK - 1) / OBLOCK)), dim3(OBLOCK), 0, (hipStream_t)stream, MapView{},
                       tds::NearView{nullptr, nullptr, nullptr, 0.0f, 0.0f, 0, 0, nullptr, nullptr}, (const float4 *)state, (const float2 *)lenwid, (const float2 *)sc, present, out, n_agents,
                       threshold, (const MapView *)set->d_views, (const tds::NearView *)set->d_near, scene_map, (int)agents_per_scene);
    TDS_LAUNCH_CHECK("offroad_kernel");
    return TDS_OK;
}

// ---- map sets: device array of the views of several maps, for launches whose scenes have different maps ----------------------------
TDS_EXPORT int tds_mapset_create(const tds_map_t *const *maps, int n, tds_mapset_t **out) {
    TDS_CHECK_ARG(out, "tds_mapset_create: null output");
    *out = nullptr;
    TDS_CHECK_ARG(maps && n > 0 && n < (1 << 24), "tds_mapset_create: need at least one map");
    std::vector<MapView> views((size_t)n);
    std::vector<tds::NearView> nears((size_t)n);
    tds_mapset *s = new (std::nothrow) tds_mapset();
    if (!s) { tds::set_error("tds_mapset_create: out of host memory"); return TDS_ENOMEM; }
    s->n = n; s->device = maps[0] ? maps[0]->device : 0; s->n_levels = maps[0] ? maps[0]->n_levels : 0; s->n_uniq = 0; s->d_views = nullptr; s->d_near = nullptr;
    for (int i = 0; i < n; ++i) {
        if (!maps[i] || maps[i]->device != s->device || maps[i]->n_levels != s->n_levels) {
            delete s;
            tds::set_error("tds_mapset_create: map %d is null, lives on another device or was created with a different level table", i);
            return TDS_EINVAL;
        }
        views[(size_t)i] = maps[i]->view;
        nears[(size_t)i] = maps[i]->near;
        if (maps[i]->n_uniq < 0) s->n_uniq = -1;
        for (int k = 0; s->n_uniq >= 0 && k < maps[i]->n_uniq; ++k) {
            bool seen = false;
            for (int j = 0; j < s->n_uniq && !seen; ++j) seen = s->uniq_keys[j] == maps[i]->uniq_keys[k];
            if (seen) continue;
            if (s->n_uniq == 64) { s->n_uniq = -1; break; }
            s->uniq_keys[s->n_uniq++] = maps[i]->uniq_keys[k];
        }
    }
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != s->device) (void)hipSetDevice(s->device);
    hipError_t e = hipMalloc((void **)&s->d_views, views.size() * sizeof(MapView));
    if (e == hipSuccess) e = hipMemcpy(s->d_views, views.data(), views.size() * sizeof(MapView), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_near, nears.size() * sizeof(tds::NearView));
    if (e == hipSuccess) e = hipMemcpy(s->d_near, nears.data(), nears.size() * sizeof(tds::NearView), hipMemcpyHostToDevice);
    if (cur != s->device) (void)hipSetDevice(cur);
    if (e != hipSuccess) {
        if (s->d_views) (void)hipFree(s->d_views);
        if (s->d_near) (void)hipFree(s->d_near);
        delete s;
        tds::set_error("tds_mapset_create: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? TDS_ENOMEM : TDS_EHIP;
    }
    *out = s;
    return TDS_OK;
}

TDS_EXPORT int tds_mapset_keys(const tds_mapset_t *set, uint32_t *keys, int cap, int *n) {
    TDS_CHECK_ARG(set && n && (keys || cap == 0) && cap >= 0, "tds_mapset_keys: bad arguments");
    *n = set->n_uniq;
    for (int i = 0; i < set->n_uniq && i < cap; ++i) keys[i] = set->uniq_keys[i];
    return TDS_OK;
}

TDS_EXPORT int tds_mapset_destroy(tds_mapset_t *set) {
    if (!set) return TDS_OK;
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != set->device) (void)hipSetDevice(set->device);
    hipError_t e = hipFree(set->d_views);
    if (set->d_near) (void)hipFree(set->d_near);
    if (cur != set->device) (void)hipSetDevice(cur);
    delete set;
    if (e != hipSuccess) { tds::set_error("tds_mapset_destroy: hipFree failed"); return TDS_EHIP; }
    return TDS_OK;
}
