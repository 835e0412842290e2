// K3: bird's-eye-view rasteriser with the reference's CV2 semantics.
// Reference dataflow being replaced: simulator.py:920-1033 -> BirdviewRGBMeshGenerator.generate (mesh.py:1053-1157,
// which concatenates the WHOLE static map per camera) -> BirdviewRenderer.render_frame (rendering/base.py:167-204) ->
// CV2Renderer.render_rgb_mesh (rendering/cv2.py:27-70: shift, trim, z-sort, project + int truncation, cv2.fillConvexPoly).
//
// Common to every path (gfx950): candidate faces come from the map's uniform grid (only the cells under the view are scanned) and
// from the actors' templates, pass the reference's trim test, are projected with the reference's fp32 operation order and truncated
// to int; survivors are compacted into a per-wave LDS queue and rasterised 64 at a time.  The fill reproduces OpenCV's
// FillConvexPoly + 8-connected Line + clipLine in closed form (16.16 fixed-point edge stepping; oracle/tds_oracle.c restates it
// sequentially).  "Later wins" of the painter's algorithm is made order-independent: key = rank << 24 | 0x00RRGGBB, where rank orders
// the rendering levels (lower level drawn later = larger rank).
//
// Two families of kernels, same pixels:
//   * bit planes (the fast path, second half of this file): one bit per pixel and key in LDS, a whole camera per workgroup,
//     spans painted with ds_or_b32, the planes resolved and streamed out at the end;
//   * packed keys (any number of keys): a strip of TW output rows as one u32 per pixel in LDS, painted with ds_max_u32 -- fused
//     (every strip scans the grid) or binned (K3a bins faces per strip, K3b rasterises the lists) -- and the generic kernel for an
//     explicit per-camera RGB mesh.
// Roofline: HBM write, 3*H*W*4 B per camera (fp32) -- DESIGN.md.
#include "tds_common.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>

using tds::GridEntry;
using tds::MapView;

// Ablation switches, work counters and tuning knobs exist only in the TESTING build of the library (libtdship_testing.so, -DTDS_TESTING:
// tools/ and tests/); in the product they are compiled out -- TDS_DBG(x) is the constant 0 there.
#ifdef TDS_TESTING
#define TDS_DBG(x) (x)
#else
#define TDS_DBG(x) 0
#endif

namespace {

constexpr int RWAVES = 4;
constexpr int RBLOCK = RWAVES * 64;
constexpr int QCAP = 64;                          // per-wave face queue (entries): key + 3 packed vertices
constexpr int Q_DW = 4 * QCAP;
constexpr int BLOCK_CAP = 256;                    // per-wave list of (face, block) pairs
constexpr int WAVE_LDS_DW = Q_DW + BLOCK_CAP;
constexpr int NO_SWITCH = 0x7fffffff;
constexpr int COORD_LIMIT = 16000;                // |pixel coordinate| below this: int16 packing and int32 16.16 slopes are exact

struct SceneArgs {
    MapView map;                // the one map of the launch, or (views != nullptr) ...
    const MapView *views;       // ... one map per scene: views[scene_map[b]] (tds_raster_scene_multi)
    const int32_t *scene_map;   // B
    const float4 *state;        // B x N
    const float2 *agent_sc;     // B x N   [sin, cos]
    const float2 *tmpl;         // B x N x 7
    const uint32_t *actor_key;  // B x N x 2, or B x Nc x N x 2 when key_per_cam
    const uint8_t *mask;        // B x Nc x N
    int N, Nc;
    int key_per_cam;            // custom_agent_colors (mesh.py:1092-1099): the actors' keys differ from camera to camera
};
// ... plus per-camera triangles.  A separate type so that the kernels of scenes without them keep their argument layout and
// register budget (the bit-plane kernel sits exactly at its VGPR limit).
struct SceneArgsEx : SceneArgs {
    const float *extra_tri;     // B x Nc x K x 3 x 2: per-camera triangles in world coordinates (waypoint discs, mesh.py:1120-1145)
    const uint32_t *extra_key;  // B x Nc x K, 0 = no triangle
    int K;
};
template <typename SA> struct has_extras { static constexpr bool value = false; };
template <> struct has_extras<SceneArgsEx> { static constexpr bool value = true; };

struct MeshArgs {
    const float *verts;         // n_img x V x 3
    const float *attrs;         // n_img x V x 3
    const int32_t *faces;       // n_img x F x 3
    int64_t V, F;
    float levels[64];           // descending
    int n_levels;
};

struct CommonArgs {
    const float2 *cam_xy, *cam_sc;
    float scale;
    int res;                    // H == W
    int strips;
    int64_t n_img;
    void *out;
    int no_trim;                // trim_mesh_before_rendering = False (cv2.py:15,32): faces are kept whether or not a vertex is in view
    uint32_t *slices;           // optional: bit-slices of the winning key index per pixel, kept for the backward pass (bit-plane kernel
                                // only; layout in include/tdship.h, tds_raster_aux_t)
    int debug;                  // ablation switches for profiling (tds_raster_set_debug): 1 no static, 2 no actors, 4 no store,
                                // 8 no outline edges, 16 no scan conversion
};

struct Camera {
    float cx, cy, s, c;
    float px[4], py[4];         // trim polygon (1.05 x view square) in camera-shifted coordinates
};

// ---------------------------------------------------------------------------------------------------------
// geometry helpers, fp32 with the reference's operation order
// ---------------------------------------------------------------------------------------------------------

// Cameras.reverse_transform_points_screen of the image corners + 1.05 margin (cv2.py:32-40, base.py:117-130)
__device__ inline void make_polygon(Camera &cam, float scale, int res) {
    const float cor[4][2] = {{0.f, 0.f}, {0.f, (float)res}, {(float)res, (float)res}, {(float)res, 0.f}};
    float mn = (float)res / 2.0f;
    float sumx = 0.0f, sumy = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = cor[k][0] - (float)res / 2.0f, y = cor[k][1] - (float)res / 2.0f;
        x = x / mn; y = y / mn;
        x = (-x) / scale; y = (-y) / scale;
        float rx = cam.c * x + (-cam.s) * y;      // rot_mat^T = [[c,-s],[s,c]]
        float ry = cam.s * x + cam.c * y;
        cam.px[k] = rx + 0.0f; cam.py[k] = ry + 0.0f;
        sumx = sumx + cam.px[k]; sumy = sumy + cam.py[k];
    }
    float mx = sumx / 4.0f, my = sumy / 4.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cam.px[k] = mx + (cam.px[k] - mx) * 1.05f;
        cam.py[k] = my + (cam.py[k] - my) * 1.05f;
    }
    // the camera is the same for the whole workgroup: keep it in scalar registers (the values were computed on the vector ALU)
    auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
#pragma unroll
    for (int k = 0; k < 4; ++k) { cam.px[k] = uni(cam.px[k]); cam.py[k] = uni(cam.py[k]); }
    cam.cx = uni(cam.cx); cam.cy = uni(cam.cy); cam.s = uni(cam.s); cam.c = uni(cam.c);
}

// utils.is_inside_polygon :99-122
__device__ inline bool inside_polygon(const Camera &cam, float x, float y) {
    bool all_right = true, all_left = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x0 = cam.px[k], y0 = cam.py[k], x1 = cam.px[(k + 1) & 3], y1 = cam.py[(k + 1) & 3];
        float a = y1 - y0, b = x0 - x1;
        float c = (-a) * x0 - b * y0;
        bool right = ((a * x + b * y) + c) >= 0.0f;
        all_right = all_right && right;
        all_left = all_left && !right;
    }
    return all_right || all_left;
}

// Cameras.transform_points_screen (base.py:102-115) on camera-shifted coordinates, then .to(int32) (cv2.py:48)
__device__ inline void project(const Camera &cam, float scale, int res, float vx, float vy, int &ox, int &oy, float &fx, float &fy) {
    float x = vx - 0.0f, y = vy - 0.0f;
    float rx = cam.c * x + cam.s * y;
    float ry = (-cam.s) * x + cam.c * y;
    rx = (-rx) * scale; ry = (-ry) * scale;
    rx = (rx * (float)res) / 2.0f; ry = (ry * (float)res) / 2.0f;
    rx = rx + (float)res / 2.0f; ry = ry + (float)res / 2.0f;
    ox = (int)rx; oy = (int)ry;
    fx = rx; fy = ry;
}

// truncating division n / d for d > 0, |n| < 2^52: the correctly rounded double quotient can only hit an integer
// when the exact quotient is one, so (long)(double) is exact; one correction step guards the claim.
__device__ inline long long div_trunc(long long n, long long d) {
    long long q = (long long)((double)n / (double)d);
    long long r = n - q * d;
    if (n >= 0) { if (r < 0) --q; else if (r >= d) ++q; }
    else { if (r > 0) ++q; else if (r <= -d) --q; }
    return q;
}

// ---------------------------------------------------------------------------------------------------------
// OpenCV FillConvexPoly (shift 0, 8-connected) for a triangle, in closed form per row
// ---------------------------------------------------------------------------------------------------------
struct Chain { int xs1; int dx1; int ysw; int xs2; int dx2; };     // slopes fit int32 for |dx| < 2^15 pixels

// n / d for n < 2^31, 1 <= d < 2^16 and a quotient of at most 2^16: the fp32 estimate (24-bit operand, v_rcp_f32, one product: relative
// error below 2^-22) is within 0.02 of the exact quotient, so its floor is off by at most one and one correction step in integers makes
// it exact (q and d fit the 24-bit multiplier, q d < 2^32)
__device__ __forceinline__ unsigned udiv_small(unsigned n, unsigned d) {
    unsigned q = (unsigned)((float)n * __builtin_amdgcn_rcpf((float)d));
    int r = (int)(n - __umul24(q, d));
    if (r < 0) --q;
    else if (r >= (int)d) ++q;
    return q;
}

// OpenCV's edge slope ((xe - xs) * 2 + dy) / (2 * dy) in 16.16 fixed point, C truncating division.
// With N = (xe - xs) << 16 this is trunc(N / dy + 1/2):
//   N >= 0: a + (2 r >= dy);   N < 0: -(a - (2 r < dy)),   a, r = divmod(|N|, dy)
// (for N < 0 the case 2|N| <= dy -> 0 of the general formula cannot occur: |N| >= 65536 > dy).  |xe - xs| and dy are below 2^15 (packed
// coordinate range), so divmod(|N|, dy) is two short divisions: |xe - xs| = hi dy + r1, then (r1 << 16) = lo dy + r  ->  a = hi << 16 | lo.
// A 32-bit unsigned division costs the compiler two quarter-rate multiplies and twenty instructions; there are four per face.
__device__ inline int edge_dx(int xs, int xe, int dyy) {
    const int dxs = xe - xs;
    const unsigned D = (unsigned)abs(dxs), d = (unsigned)dyy;
    const unsigned hi = udiv_small(D, d), r1 = D - __umul24(hi, d);
    const unsigned lo = udiv_small(r1 << 16, d), r = (r1 << 16) - __umul24(lo, d);
    const unsigned a = (hi << 16) + lo;
    if (dxs >= 0) return (int)(a + ((2u * r >= d) ? 1u : 0u));
    return -(int)(a - ((2u * r < d) ? 1u : 0u));
}

// chain visiting a0 -> a1 -> a2 (a0 = first top vertex)
__device__ inline Chain make_chain(const int *px, const int *py, int a0, int a1, int a2) {
    Chain c;
    int ymin = py[a0];
    c.ysw = NO_SWITCH; c.xs2 = 0; c.dx2 = 0;
    if (py[a1] > ymin) {
        c.xs1 = px[a0];
        c.dx1 = edge_dx(px[a0], px[a1], py[a1] - ymin);
        if (py[a2] > py[a1]) {
            c.ysw = py[a1];
            c.xs2 = px[a1];
            c.dx2 = edge_dx(px[a1], px[a2], py[a2] - py[a1]);
        }
    } else {
        int dyy = py[a2] - ymin;
        c.xs1 = px[a1];
        c.dx1 = dyy > 0 ? edge_dx(px[a1], px[a2], dyy) : 0;
    }
    return c;
}

__device__ inline long long chain_x(int xs1, int dx1, int ysw, int xs2, int dx2, int ymin, int y) {
    bool second = y >= ysw;
    long long x0 = (long long)(second ? xs2 : xs1) << 16;
    int dx = second ? dx2 : dx1;
    int y0 = second ? ysw : ymin;
    return x0 + (long long)(y - y0) * (long long)dx;
}

// cv::clipLine on int64 (drawing.cpp); returns false if nothing is left
__device__ inline bool clip_line(int W, int H, long long &x1, long long &y1, long long &x2, long long &y2) {
    long long right = W - 1, bottom = H - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// cv::Line, 8-connected, leftToRight; paints the pixels that fall into the strip [X0, X0+TW).
// The Bresenham walk is entered in closed form at the first pixel whose x reaches the strip and left once x passes it
// (x never decreases along the walk): after k steps the minor axis has advanced m_k = floor((2 dmin k + dmaj - 1) / (2 dmaj))
// and the error term is e0 - 2 dmin k + 2 dmaj m_k  (checked exhaustively against the iterative form).
template <int TW>
__device__ inline void draw_line(uint32_t *tile, int H, int W, int X0, int ax, int ay, int bx, int by, uint32_t key) {
    long long x1 = ax, y1 = ay, x2 = bx, y2 = by;
    if ((unsigned long long)x1 >= (unsigned long long)W || (unsigned long long)x2 >= (unsigned long long)W ||
        (unsigned long long)y1 >= (unsigned long long)H || (unsigned long long)y2 >= (unsigned long long)H) {
        if (!clip_line(W, H, x1, y1, x2, y2)) return;
    }
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    int px = (int)x1, py = (int)y1;
    int step_y = 1;
    if (dx < 0) { dx = -dx; dy = -dy; px = (int)x2; py = (int)y2; }
    if (px >= X0 + TW || px + dx < X0) return;                   // the x range [px, px + dx] misses the strip
    if (dy < 0) { dy = -dy; step_y = -1; }
    const bool vert = dy > dx;
    const int dmaj = vert ? dy : dx, dmin = vert ? dx : dy;
    int err = dmaj - (dmin + dmin);
    const int plus_delta = dmaj + dmaj, minus_delta = -(dmin + dmin);
    int k = 0;                                                       // steps already taken
    if (px < X0) {
        const int t = X0 - px;                                       // columns to skip
        if (!vert) {
            k = t;
            int m = (int)(((unsigned)(2 * dmin) * (unsigned)k + (unsigned)dmaj - 1u) / (unsigned)(2 * dmaj));
            err += k * minus_delta + m * plus_delta;
            px += k; py += step_y * m;
        } else {
            // first step count with m_k >= t  (dmin > 0 here because px + dx >= X0 > px)
            k = (int)(((unsigned)(2 * dmaj) * (unsigned)t - (unsigned)dmaj + (unsigned)(2 * dmin)) / (unsigned)(2 * dmin));
            int m = (int)(((unsigned)(2 * dmin) * (unsigned)k + (unsigned)dmaj - 1u) / (unsigned)(2 * dmaj));
            err += k * minus_delta + m * plus_delta;
            py += step_y * k; px += m;
        }
    }
    uint32_t *p = tile + (px - X0) * H + py;
    const int pstep_maj = vert ? step_y : H, pstep_min = vert ? H : step_y;
    for (; k <= dmaj; ++k) {
        if (px >= X0 + TW) break;
        atomicMax(p, key);
        bool neg = err < 0;
        err += minus_delta + (neg ? plus_delta : 0);
        p += pstep_maj + (neg ? pstep_min : 0);
        px += vert ? (neg ? 1 : 0) : 1;
    }
}

// ---------------------------------------------------------------------------------------------------------
// per-wave face queue and batch rasterisation
// ---------------------------------------------------------------------------------------------------------
struct WaveCtx {
    uint32_t *tile;
    uint32_t *q;        // [4][QCAP]: key, then the three vertices packed as (x & 0xffff) | y << 16
    uint32_t *blocks;   // [BLOCK_CAP]: face | block << 8
    int qlen;           // wave-uniform
    int lane;
    int H, W, X0;
    int debug;
};

__device__ inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ inline uint32_t pack_xy(int x, int y) { return ((uint32_t)x & 0xffffu) | ((uint32_t)y << 16); }
__device__ inline int unpack_x(uint32_t p) { return (int)(short)(p & 0xffffu); }
__device__ inline int unpack_y(uint32_t p) { return (int)p >> 16; }

// Faces whose pixel coordinates do not fit the packed fast path (|coord| >= COORD_LIMIT): the sequential OpenCV
// algorithm executed by a single lane, painting only the strip.  Practically never taken (a vertex more than 16000
// pixels away from the image), kept so that the result is exact for any input.
template <int TW>
__device__ __noinline__ void fill_generic(uint32_t *tile, int H, int W, int X0, int x0, int y0, int x1, int y1, int x2, int y2, uint32_t key) {
    const int px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2};
    draw_line<TW>(tile, H, W, X0, px[2], py[2], px[0], py[0], key);
    draw_line<TW>(tile, H, W, X0, px[0], py[0], px[1], py[1], key);
    draw_line<TW>(tile, H, W, X0, px[1], py[1], px[2], py[2], key);
    long long xmin = px[0], xmax = px[0], ymin = py[0], ymax = py[0];
    int imin = 0;
    for (int i = 0; i < 3; ++i) {
        if (py[i] < ymin) { ymin = py[i]; imin = i; }
        if (py[i] > ymax) ymax = py[i];
        if (px[i] > xmax) xmax = px[i];
        if (px[i] < xmin) xmin = px[i];
    }
    if (xmax < 0 || ymax < 0 || xmin >= W || ymin >= H) return;
    if (ymax > H - 1) ymax = H - 1;
    int eidx[2] = {imin, imin}, edi[2] = {1, 2}, eye[2] = {(int)ymin, (int)ymin};
    long long ex[2] = {-65536, -65536}, edx[2] = {0, 0};
    int edges = 3, y = (int)ymin;
    do {
        for (int i = 0; i < 2; ++i) {
            if (y >= eye[i]) {
                int idx0 = eidx[i], idx = idx0 + edi[i];
                if (idx >= 3) idx -= 3;
                for (; edges-- > 0;) {
                    int ty = py[idx];
                    if (ty > y) {
                        long long xs = (long long)px[idx0] << 16, xe = (long long)px[idx] << 16;
                        eye[i] = ty;
                        edx[i] = div_trunc((xe - xs) * 2 + (ty - y), 2ll * (ty - y));
                        ex[i] = xs;
                        eidx[i] = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += edi[i];
                    if (idx >= 3) idx -= 3;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            long long xl = ex[0] < ex[1] ? ex[0] : ex[1], xr = ex[0] < ex[1] ? ex[1] : ex[0];
            long long xx1 = (xl + 32768) >> 16, xx2 = (xr + 32768) >> 16;
            if (xx2 >= 0 && xx1 < W) {
                int s0 = (int)(xx1 < 0 ? 0 : xx1), s1 = (int)(xx2 >= W ? W - 1 : xx2);
                s0 = max(s0, X0); s1 = min(s1, X0 + TW - 1);
                for (int x = s0; x <= s1; ++x) atomicMax(&tile[(x - X0) * H + y], key);
            }
        }
        ex[0] += edx[0];
        ex[1] += edx[1];
    } while (++y <= (int)ymax);
}

struct FaceRows { int ymin, ystart, nrows, imin, xlo, xhi; };

// rows that OpenCV's scan conversion paints for this triangle, and the pixel columns of the strip it can touch
__device__ inline FaceRows face_rows(const int *px, const int *py, int H, int W, int X0, int TW) {
    FaceRows r;
    int ymax = py[0], xmin = px[0], xmax = px[0];
    r.ymin = py[0]; r.imin = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k) {
        if (py[k] < r.ymin) { r.ymin = py[k]; r.imin = k; }
        ymax = max(ymax, py[k]); xmin = min(xmin, px[k]); xmax = max(xmax, px[k]);
    }
    r.ystart = max(r.ymin, 0);
    int yend = min(ymax - 1, H - 1);              // the row of the bottom vertex is never scan-converted (edges run out)
    r.xlo = max(max(xmin, 0), X0);
    r.xhi = min(min(xmax, W - 1), X0 + TW - 1);
    bool scan = !(xmax < 0 || ymax < 0 || xmin >= W || r.ymin >= H) && (r.xlo <= r.xhi);
    r.nrows = scan ? max(0, yend - r.ystart + 1) : 0;
    return r;
}

constexpr int TILE = 8;          // scan conversion work unit: TILE rows x TILE columns of a face's bounding box

// Rasterise the first n (<= 64) faces of the wave's queue.
//   1. lane f sets up face f: rows to paint and the two edge chains of OpenCV's scan conversion (start x, 16.16 slope,
//      switch row) -- one 32-bit division per edge;
//   2. the part of each face's bounding box inside the strip is cut into TILE x TILE tiles; tiles are numbered by a wave
//      prefix sum, their owners write (face, tile) pairs into an LDS list, and the wave paints one tile per lane: the edge
//      data comes from the owner lane by ds_bpermute, rows are stepped like OpenCV does, every span is cut to the tile.
//      Work per lane is bounded by TILE x TILE whatever the face sizes, so slivers and big road triangles mix well;
//   3. the outline edges (Bresenham + OpenCV's clipLine) in segments of SEG pixels, one segment per lane, each segment
//      entered in closed form.
template <int TW>
__device__ __forceinline__ void process_batch(WaveCtx &w, int n) {
    const int lane = w.lane, H = w.H, W = w.W, X0 = w.X0;
    wave_sync();
    // ---- per-face set-up (lane = face)
    uint32_t key = 0, v0 = 0, v1 = 0, v2 = 0;
    int ntiles = 0, ntx = 1;
    FaceRows r = {0, 0, 0, 0, 0, 0};
    Chain a = {0, 0, NO_SWITCH, 0, 0}, b = {0, 0, NO_SWITCH, 0, 0};
    if (lane < n) {
        key = w.q[lane]; v0 = w.q[QCAP + lane]; v1 = w.q[2 * QCAP + lane]; v2 = w.q[3 * QCAP + lane];
        const int px[3] = {unpack_x(v0), unpack_x(v1), unpack_x(v2)}, py[3] = {unpack_y(v0), unpack_y(v1), unpack_y(v2)};
        r = face_rows(px, py, H, W, X0, TW);
        if (r.nrows > 0 && !(TDS_DBG(w.debug) & 16)) {
            int i1 = r.imin == 2 ? 0 : r.imin + 1, i2 = r.imin == 0 ? 2 : r.imin - 1;
            a = make_chain(px, py, r.imin, i1, i2);
            b = make_chain(px, py, r.imin, i2, i1);
            ntx = (r.xhi - r.xlo) / TILE + 1;
            ntiles = ntx * ((r.nrows + TILE - 1) / TILE);
        }
    }
    // ---- scan conversion, one tile per lane
    if (__ballot(ntiles > 0) != 0) {
        int incl = ntiles;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        const int excl = incl - ntiles;
        const int total = __shfl(incl, 63);
        // what a tile needs from its face, packed for the cross-lane fetch
        const int sh_a = (a.xs1 & 0xffff) | (a.xs2 << 16), sh_b = (b.xs1 & 0xffff) | (b.xs2 << 16);
        const int sh_sw = (min(a.ysw, 0x7fff) & 0xffff) | (min(b.ysw, 0x7fff) << 16);
        const int sh_y = (r.ymin & 0xffff) | (r.ystart << 16), sh_n = r.nrows | (ntx << 16), sh_x = r.xlo | (r.xhi << 16);
        for (int base = 0; base < total; base += BLOCK_CAP) {
            int lo = max(excl, base), hi = min(incl, base + BLOCK_CAP);
            for (int idx = lo; idx < hi; ++idx) w.blocks[idx - base] = (uint32_t)lane | ((uint32_t)(idx - excl) << 8);
            wave_sync();
            const int m = min(BLOCK_CAP, total - base);
            for (int i0 = 0; i0 < m; i0 += 64) {
                const int i = i0 + lane;
                const bool live = i < m;
                const uint32_t e = live ? w.blocks[i] : 0u;
                const int f = e & 0xff, t = (int)(e >> 8);
                // all lanes take part in the shuffles
                const int ga = __shfl(sh_a, f), gb = __shfl(sh_b, f), gsw = __shfl(sh_sw, f), gy = __shfl(sh_y, f), gn = __shfl(sh_n, f);
                const int gx = __shfl(sh_x, f), adx1 = __shfl(a.dx1, f), adx2 = __shfl(a.dx2, f), bdx1 = __shfl(b.dx1, f), bdx2 = __shfl(b.dx2, f);
                const uint32_t k2 = (uint32_t)__shfl((int)key, f);
                if (live) {
                    const int ymin = (int)(short)(gy & 0xffff), ystart = gy >> 16, nrows = gn & 0xffff, tnx = gn >> 16;
                    const int ty = t / tnx, tx = t - ty * tnx;
                    const int y0 = ystart + ty * TILE, y1 = min(y0 + TILE, ystart + nrows) - 1;
                    const int c0 = (gx & 0xffff) + tx * TILE, c1 = min(c0 + TILE - 1, gx >> 16);
                    const int aysw = (int)(short)(gsw & 0xffff), bysw = gsw >> 16;           // 0x7fff = no second edge
                    const int axs1 = (int)(short)(ga & 0xffff), axs2 = ga >> 16, bxs1 = (int)(short)(gb & 0xffff), bxs2 = gb >> 16;
                    long long xa = chain_x(axs1, adx1, aysw, axs2, adx2, ymin, y0);
                    long long xb = chain_x(bxs1, bdx1, bysw, bxs2, bdx2, ymin, y0);
                    int da = y0 >= aysw ? adx2 : adx1, db = y0 >= bysw ? bdx2 : bdx1;
                    for (int y = y0; y <= y1; ++y) {
                        long long xl = xa < xb ? xa : xb, xr = xa < xb ? xb : xa;
                        int xx1 = (int)((xl + 32768) >> 16), xx2 = (int)((xr + 32768) >> 16);
                        // OpenCV draws [xx1, xx2] clamped to the image when it is not entirely outside; here cut to the tile
                        int s0 = max(xx1, c0), s1 = min(xx2, c1);
                        uint32_t *p = w.tile + (s0 - X0) * H + y;
                        for (int x = s0; x <= s1; ++x, p += H) atomicMax(p, k2);
                        xa += da; xb += db;
                        if (y + 1 == aysw) { xa = (long long)axs2 << 16; da = adx2; }
                        if (y + 1 == bysw) { xb = (long long)bxs2 << 16; db = bdx2; }
                    }
                }
            }
            wave_sync();
        }
    }
    // ---- outline edges: OpenCV draws Line(v2,v0), Line(v0,v1), Line(v1,v2) before the scan conversion
    if (!(TDS_DBG(w.debug) & 8)) {
#pragma unroll 1
        for (int l = 0; l < 3; ++l) {
            const uint32_t pa = l == 0 ? v2 : (l == 1 ? v0 : v1), pb = l == 0 ? v0 : (l == 1 ? v1 : v2);
            if (lane < n) draw_line<TW>(w.tile, H, W, X0, unpack_x(pa), unpack_y(pa), unpack_x(pb), unpack_y(pb), key);
        }
    }
    wave_sync();
}

// project (fp32, reference order) -> reject faces whose pixel bounding box misses the strip -> trim test
// (>= 1 vertex inside the 1.05x view polygon, cv2.py:32-41).  The cheap exact rejection comes first.
// `ins` receives one bit per vertex that passed the trim test.
__device__ inline bool trim_project(const Camera &cam, float scale, int res, int X0, int TW, const float *sx, const float *sy,
                                    int *px, int *py, unsigned &ins, int no_trim = 0) {
    ins = 0;
    float fx[3], fy[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) project(cam, scale, res, sx[k], sy[k], px[k], py[k], fx[k], fy[k]);
    int xmin = min(px[0], min(px[1], px[2])), xmax = max(px[0], max(px[1], px[2]));
    int ymin = min(py[0], min(py[1], py[2])), ymax = max(py[0], max(py[1], py[2]));
    if (xmax < X0 || xmin >= X0 + TW || ymax < 0 || ymin >= res) return false;
    // untrimmed (cv2.py:32 off): every face goes to fillConvexPoly; one whose pixel bounding box misses the image paints nothing, the others
    // are all candidates of the scan (a triangle that meets the window lies over a scanned grid cell)
    if (no_trim) { ins = 7u; return true; }
    // The trim polygon is the image square scaled by 1.05 about its centre, i.e. pixel coordinates in [-0.025 res, 1.025 res]^2.
    // The two fp32 formulations (pixel coordinates here, the reference's half-plane tests in world coordinates) agree to well
    // below 1e-2 pixel, so only vertices within `band` of the border need the reference's own test (NaNs end up there too).
    const float lo = -0.025f * (float)res, hi = 1.025f * (float)res, band = 0.0625f * fmaxf(1.0f, (float)res * (1.0f / 256.0f));
    unsigned amb = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float mn = fminf(fx[k], fy[k]), mx = fmaxf(fx[k], fy[k]), sum = fx[k] + fy[k];
        const bool fin = sum == sum;                                   // fminf / fmaxf drop NaNs: keep them for the exact test
        const bool in = fin && mn > lo + band && mx < hi - band;
        const bool out = fin && (mn < lo - band || mx > hi + band);
        ins |= in ? (1u << k) : 0u;
        amb |= (!in && !out) ? (1u << k) : 0u;
    }
    if (__builtin_expect(__ballot(amb != 0) != 0, 0)) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if ((amb >> k) & 1u) ins |= inside_polygon(cam, sx[k], sy[k]) ? (1u << k) : 0u;
    }
    return ins != 0;
}

// Which of the three outline edges (l = 0: v2-v0, 1: v0-v1, 2: v1-v2) must be drawn: an edge flagged as the repeat of an
// earlier same-key face's edge is skipped when one of its end points passed the trim test (then that face is drawn as well)
__device__ inline unsigned edge_mask(unsigned dup, unsigned ins) {
    const unsigned e0 = (ins & 5u) ? 1u : 0u, e1 = (ins & 3u) ? 2u : 0u, e2 = (ins & 6u) ? 4u : 0u;    // end points of edge l inside?
    return 7u & ~(dup & (e0 | e1 | e2));
}

template <int TW, typename OutT>
__device__ inline void write_out(const uint32_t *tile, OutT *out, int64_t img, int res, int X0, int tid) {
    const int H = res, W = res;
    const int tw = min(TW, W - X0);
    const int n = tw * H;
    const int64_t plane = (int64_t)W * H;
    OutT *o = out + img * 3 * plane + (int64_t)X0 * H;
    if constexpr (sizeof(OutT) == 4) {
        if ((H & 3) == 0) {
            for (int i = tid * 4; i < n; i += RBLOCK * 4) {
                uint4 k = *(const uint4 *)(tile + i);
                float4 r = make_float4((float)((k.x >> 16) & 255), (float)((k.y >> 16) & 255), (float)((k.z >> 16) & 255), (float)((k.w >> 16) & 255));
                float4 g = make_float4((float)((k.x >> 8) & 255), (float)((k.y >> 8) & 255), (float)((k.z >> 8) & 255), (float)((k.w >> 8) & 255));
                float4 b = make_float4((float)(k.x & 255), (float)(k.y & 255), (float)(k.z & 255), (float)(k.w & 255));
                *(float4 *)(o + i) = r;
                *(float4 *)(o + plane + i) = g;
                *(float4 *)(o + 2 * plane + i) = b;
            }
            return;
        }
    } else {
        if ((H & 15) == 0) {
            for (int i = tid * 16; i < n; i += RBLOCK * 16) {
                uint32_t r[4], g[4], b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint4 k = *(const uint4 *)(tile + i + 4 * q);
                    r[q] = ((k.x >> 16) & 255) | (((k.y >> 16) & 255) << 8) | (((k.z >> 16) & 255) << 16) | (((k.w >> 16) & 255) << 24);
                    g[q] = ((k.x >> 8) & 255) | (((k.y >> 8) & 255) << 8) | (((k.z >> 8) & 255) << 16) | (((k.w >> 8) & 255) << 24);
                    b[q] = (k.x & 255) | ((k.y & 255) << 8) | ((k.z & 255) << 16) | ((k.w & 255) << 24);
                }
                *(uint4 *)(o + i) = make_uint4(r[0], r[1], r[2], r[3]);
                *(uint4 *)(o + plane + i) = make_uint4(g[0], g[1], g[2], g[3]);
                *(uint4 *)(o + 2 * plane + i) = make_uint4(b[0], b[1], b[2], b[3]);
            }
            return;
        }
    }
    for (int i = tid; i < n; i += RBLOCK) {
        uint32_t k = tile[i];
        o[i] = (OutT)((k >> 16) & 255);
        o[plane + i] = (OutT)((k >> 8) & 255);
        o[2 * plane + i] = (OutT)(k & 255);
    }
}

__device__ inline void block_to_image(int64_t nblk, int strips, int64_t &img, int &strip) {
    // blocks are dealt round-robin to the 8 XCDs; give consecutive logical ids to the same XCD so that the strips of
    // one camera (and neighbouring cameras of a scene) share an L2
    int64_t b = blockIdx.x, L = b;
    if ((nblk & 7) == 0) L = (b & 7) * (nblk >> 3) + (b >> 3);
    img = L / strips;
    strip = (int)(L - img * strips);
}

// ---------------------------------------------------------------------------------------------------------
// kernels.  Each kernel is one loop: a producer step yields at most one candidate face per lane (already projected,
// known to touch the strip, trimmed); `drain` appends them to the per-wave queue and rasterises a full queue on the spot.
// There is exactly one inlined copy of process_batch per kernel; all control flow is wave-uniform.
// ---------------------------------------------------------------------------------------------------------
template <int TW>
__device__ __forceinline__ void drain(WaveCtx &w, bool acc, uint32_t key, const int (&px)[3], const int (&py)[3], bool more) {
    // faces outside the packed coordinate range take the exact sequential path, one lane each
    bool big = acc && (max(max(abs(px[0]), abs(px[1])), max(max(abs(px[2]), abs(py[0])), max(abs(py[1]), abs(py[2])))) >= COORD_LIMIT);
    if (__builtin_expect(__ballot(big) != 0, 0)) {
        if (big) fill_generic<TW>(w.tile, w.H, w.W, w.X0, px[0], py[0], px[1], py[1], px[2], py[2], key);
        acc = acc && !big;
    }
    unsigned long long pending = __ballot(acc);
    for (;;) {
        if (pending != 0) {
            const int room = QCAP - w.qlen;
            int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(pending >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pending, 0));
            bool take = acc && ((pending >> w.lane) & 1) && rank < room;
            if (take) {
                int slot = w.qlen + rank;
                w.q[slot] = key;
                w.q[1 * QCAP + slot] = pack_xy(px[0], py[0]);
                w.q[2 * QCAP + slot] = pack_xy(px[1], py[1]);
                w.q[3 * QCAP + slot] = pack_xy(px[2], py[2]);
            }
            unsigned long long taken = __ballot(take);
            w.qlen += __popcll(taken);
            pending &= ~taken;
        }
        if (w.qlen == QCAP || (!more && pending == 0 && w.qlen > 0)) {
            process_batch<TW>(w, w.qlen);
            w.qlen = 0;
        }
        if (pending == 0) break;
    }
}

#define TDS_RASTER_PROLOGUE()                                                                                   \
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];                                             \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                              \
    const int res = c.res, H = res, W = res;                                                                    \
    int64_t img;                                                                                                \
    int strip;                                                                                                  \
    block_to_image(c.n_img * c.strips, c.strips, img, strip);                                                   \
    const int X0 = strip * TW;                                                                                  \
    uint32_t *tile = smem;                                                                                      \
    for (int i = tid * 4; i < TW * H; i += RBLOCK * 4) *(uint4 *)(tile + i) = make_uint4(0, 0, 0, 0);            \
    WaveCtx w;                                                                                                  \
    w.tile = tile;                                                                                              \
    w.q = smem + TW * H + wave * WAVE_LDS_DW;                                                                   \
    w.blocks = w.q + Q_DW;                                                                                      \
    w.qlen = 0; w.lane = lane; w.H = H; w.W = W; w.X0 = X0; w.debug = c.debug;                                  \
    Camera cam;                                                                                                 \
    {                                                                                                           \
        float2 xy = c.cam_xy[img], sc = c.cam_sc[img];                                                          \
        cam.cx = xy.x; cam.cy = xy.y; cam.s = sc.x; cam.c = sc.y;                                               \
        make_polygon(cam, c.scale, res);                                                                        \
    }                                                                                                           \
    __syncthreads();

// ---- wave-level helpers (scan, bit-plane path) ----------------------------------------------------------------------
// Inclusive scans over the 64 lanes on DPP (row_shr 1,2,4,8 inside rows of 16, then row_bcast 15 / 31); all lanes active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_from(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ int wave_scan_add(int v) {
    v += dpp_from<0x111, 0xf>(v); v += dpp_from<0x112, 0xf>(v); v += dpp_from<0x114, 0xf>(v); v += dpp_from<0x118, 0xf>(v);
    v += dpp_from<0x142, 0xa>(v); v += dpp_from<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ int wave_scan_max(int v) {           // values >= 0
    v = max(v, dpp_from<0x111, 0xf>(v)); v = max(v, dpp_from<0x112, 0xf>(v)); v = max(v, dpp_from<0x114, 0xf>(v));
    v = max(v, dpp_from<0x118, 0xf>(v)); v = max(v, dpp_from<0x142, 0xa>(v)); v = max(v, dpp_from<0x143, 0xc>(v));
    return v;
}

// ---- scene producer: actors, then the masked-agent dot, then the static map cells under a pixel window --------------
constexpr int SCAN_EMPTY_ROW = (int)0xffff7fffu;     // cell range lo = 0x7fff, hi = -1
struct ScanState {
    int phase, a0;
    bool masked_seen;
    int cx0, cx1, cy0, nrows, row, chunk, prev_rw;       // wave-uniform: cell rectangle, grid rows to scan, first row of the current block of 64
                                                         // rows, this wave's next chunk of the block's entry list, cell range of the row before the block
    int rw, rex, rbase, rfe;                             // lane r: grid row (block start + r): cell range lo | hi << 16; number of its first entry in the
                                                         // block's list (prefix sum of the rows' entry counts), first entry - that number; end of its first cell
    int rtotal;                                          // entries of the block (wave-uniform)
    // the chunk of entries in flight (scan_fetch): this lane's entry index | its grid row within the block << 25 (-1: none) and data
    bool have;
    int cur_i;
    uint4 pu0, pu1, pu2;                                 // (pu2: the third 16 bytes of a QuadEntry, bit-plane kernels)
    MapView map;                                         // the map of this camera's scene (wave-uniform)
    uint32_t *dyn;                                       // LDS counter that deals the chunks to the waves as they ask (nullptr: round-robin)
    int done;                                            // chunks of the blocks of grid rows already left behind (the counter numbers them all)
};

// Lane r prepares grid row `row0 + r` of the scan: the cells of that row under the window polygon (the pixel window plus a 2 px
// margin, a rotated rectangle in world space) and where their entries start and end.  The entries of consecutive cells of a
// grid row are contiguous, so a row is ONE range of entries.
__device__ __forceinline__ void scan_load_rows(ScanState &st, const MapView &m, const CommonArgs &c, const Camera &cam, int lane, int X0, int TWw,
                                               int row0) {
    const int res = c.res;
    const float half = (float)res / 2.0f;
    const float pxs[2] = {(float)X0 - 2.0f, (float)min(X0 + TWw, res) + 2.0f}, pys[2] = {-2.0f, (float)res + 2.0f};
    float qx[4], qy[4];                                    // corners in order around the rectangle
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x = -((pxs[(k == 1 || k == 2) ? 1 : 0] - half) / half) / c.scale, y = -((pys[k >= 2 ? 1 : 0] - half) / half) / c.scale;
        qx[k] = cam.c * x - cam.s * y + cam.cx; qy[k] = cam.s * x + cam.c * y + cam.cy;
    }
    const float eps = 1e-3f + 1e-6f * (fabsf(cam.cx) + fabsf(cam.cy));
    const int r = row0 + lane;
    st.rw = SCAN_EMPTY_ROW; st.rfe = 0;
    int re0 = 0, re1 = 0;
    if (r < st.nrows) {
        const int cy = st.cy0 + r;
        const float ya = m.oy + (float)cy * m.cell - eps - 1e-4f * m.cell, yb = m.oy + (float)(cy + 1) * m.cell + eps + 1e-4f * m.cell;
        float xmin = 3.0e38f, xmax = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x0 = qx[k], y0 = qy[k], x1 = qx[(k + 1) & 3], y1 = qy[(k + 1) & 3];
            if (y0 >= ya && y0 <= yb) { xmin = fminf(xmin, x0); xmax = fmaxf(xmax, x0); }
            const float dy = y1 - y0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float yl = j == 0 ? ya : yb;
                if ((y0 - yl) * (y1 - yl) <= 0.0f && dy != 0.0f) {
                    const float x = x0 + (yl - y0) / dy * (x1 - x0);
                    xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
                }
            }
        }
        if (xmin <= xmax) {
            const float pad = eps + 1e-4f * m.cell;
            const float f0 = fminf(fmaxf((xmin - pad - m.ox) * m.inv_cell, -1.0f), (float)m.nx), f1 = fminf(fmaxf((xmax + pad - m.ox) * m.inv_cell, -1.0f), (float)m.nx);
            const int lo = max((int)floorf(f0), st.cx0), hi = min((int)floorf(f1), st.cx1);
            if (lo <= hi) {
                const int32_t *cs = m.cell_start + (size_t)cy * m.nx;
                st.rw = lo | (hi << 16);
                re0 = cs[lo]; re1 = cs[hi + 1]; st.rfe = cs[lo + 1];
            }
        }
    }
    // the entry ranges of the block's rows form one list: number the entries once per block (scan_fetch cuts the list into chunks of 64)
    const int cnt = re1 - re0, incl = wave_scan_add(cnt);
    st.rex = incl - cnt;
    st.rbase = re0 - st.rex;
    st.rtotal = __builtin_amdgcn_readlane(incl, 63);
}

// window = pixel columns [X0, X0 + TWw) of the image (the whole image when binning)
// POLY: the walk is over the rendering grid with paired faces (MapView::qentries / qcell_start: the bit-plane kernels); the cells are the same
template <typename SA, bool POLY = false>
__device__ __forceinline__ void scan_init(ScanState &st, const SA &a, const CommonArgs &c, const Camera &cam, int64_t img, int lane, int wave, int X0,
                                          int TWw) {   // wave = index among the cooperating waves
    st.map = a.map;
    if (a.views != nullptr) st.map = a.views[a.scene_map[img / a.Nc]];
    if constexpr (POLY) st.map.cell_start = st.map.qcell_start;
    const MapView &m = st.map;
    const int res = c.res;
    st.phase = (a.N > 0 && !(TDS_DBG(c.debug) & 2)) ? 0 : 2;      // 0 actors, 1 masked-agent dot, 3 per-camera triangles, 2 static map
    if constexpr (POLY) {
        // The scan kernel of the split form reads the first phase through the launch's (always zero, in the product) debug word: a phase the
        // compiler knows at compile time makes it restructure the phases so that scan_faces_kernel needs 368 instead of 112 bytes of scratch per
        // lane and takes 0.98 instead of 0.72 ms at B = 1024 x 64.  (Found because the testing build was the faster one; an opaque register
        // in place of the kernel argument does not have the effect.)
        st.phase = (a.N > 0 && !(c.debug & 2)) ? 0 : 2;
    }
    if constexpr (has_extras<SA>::value) { if (st.phase == 2 && a.K > 0) st.phase = 3; }
    st.a0 = 0; st.masked_seen = false;
    st.cx0 = 0; st.cx1 = -1; st.cy0 = 0; st.nrows = 0; st.row = 0; st.chunk = __builtin_amdgcn_readfirstlane(wave); st.prev_rw = SCAN_EMPTY_ROW;
    st.rw = SCAN_EMPTY_ROW; st.rex = st.rbase = st.rfe = 0; st.rtotal = 0;
    st.have = false; st.cur_i = -1; st.dyn = nullptr; st.done = 0;
    st.pu0 = st.pu1 = st.pu2 = make_uint4(0, 0, 0, 0);
    if (m.nx > 0 && !(TDS_DBG(c.debug) & 1)) {
        // world-space bounding box of the window (2 px margin: int truncation moves a vertex by < 1 px) -> grid cell rectangle
        float wx0 = 3.0e38f, wx1 = -3.0e38f, wy0 = 3.0e38f, wy1 = -3.0e38f;
        const float half = (float)res / 2.0f;
        const float pxs[2] = {(float)X0 - 2.0f, (float)min(X0 + TWw, res) + 2.0f}, pys[2] = {-2.0f, (float)res + 2.0f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float x = -((pxs[i] - half) / half) / c.scale, y = -((pys[j] - half) / half) / c.scale;
                float rx = cam.c * x - cam.s * y + cam.cx, ry = cam.s * x + cam.c * y + cam.cy;
                wx0 = fminf(wx0, rx); wx1 = fmaxf(wx1, rx); wy0 = fminf(wy0, ry); wy1 = fmaxf(wy1, ry);
            }
        const float eps = 1e-3f + 1e-6f * (fabsf(cam.cx) + fabsf(cam.cy));
        float fx0 = fminf(fmaxf((wx0 - eps - m.ox) * m.inv_cell, -1.0f), (float)m.nx), fx1 = fminf(fmaxf((wx1 + eps - m.ox) * m.inv_cell, -1.0f), (float)m.nx);
        float fy0 = fminf(fmaxf((wy0 - eps - m.oy) * m.inv_cell, -1.0f), (float)m.ny), fy1 = fminf(fmaxf((wy1 + eps - m.oy) * m.inv_cell, -1.0f), (float)m.ny);
        st.cx0 = max((int)floorf(fx0), 0); st.cx1 = min((int)floorf(fx1), m.nx - 1);
        st.cy0 = max((int)floorf(fy0), 0);
        const int cy1 = min((int)floorf(fy1), m.ny - 1);
        st.nrows = (st.cx0 <= st.cx1 && st.cy0 <= cy1) ? __builtin_amdgcn_readfirstlane(cy1 - st.cy0 + 1) : 0;
        if (st.nrows > 0) scan_load_rows(st, m, c, cam, lane, X0, TWw, 0);
    }
}

// Move to the next chunk of map entries (if any) and request this lane's entry of it.  The entry ranges of the (up to 64) grid rows of
// the current block form ONE list -- lane r holds row r's range, a wave prefix sum numbers the entries -- which is cut into chunks of 64
// dealt round-robin to the cooperating waves: every chunk but the last of a block is full whatever the rows hold (with a chunk per row
// range the tails of 6 - 9 short ranges per view idled a third of the lanes).  A lane finds the row of its entry by bisection.
template <int NW, bool POLY = false>
__device__ __forceinline__ bool scan_fetch(ScanState &st, const MapView &m, const CommonArgs &c, const Camera &cam, int lane, int X0, int TWw) {
    while (st.row < st.nrows) {
        const int nblk = min(64, st.nrows - st.row);
        const int total = st.rtotal;
        if (st.chunk * 64 < total) {
            const int v = st.chunk * 64 + lane;
            int r = 0;                                        // the last row whose first entry number is <= v (rows without entries share the
#pragma unroll                                                // number of the next one and lose to it)
            for (int step = 32; step >= 1; step >>= 1) {
                if (step < nblk) {                            // wave-uniform: a view spans a handful of grid rows, the upper steps find nothing
                    const int cand = r + step;
                    const int ex = __shfl(st.rex, cand & 63);
                    if (cand < nblk && ex <= v) r = cand;
                }
            }
            const int i = __shfl(st.rbase, r) + v;
            st.cur_i = v < total ? (i | (r << 25)) : -1;
            if (v < total) {
                if constexpr (POLY) {
                    const uint4 *ep = (const uint4 *)(m.qentries + i);
                    st.pu0 = ep[0]; st.pu1 = ep[1]; st.pu2 = ep[2];
                } else {
                    const uint4 *ep = (const uint4 *)(m.entries + i);
                    st.pu0 = ep[0]; st.pu1 = ep[1];
                }
            }
            if (st.dyn != nullptr) {
                // the next chunk goes to whichever wave asks first: the waves of a camera finish together (they meet at a barrier before the
                // stream-out, and with three waves per SIMD an idle one costs issue slots)
                unsigned nxt = 0;
                if (lane == 0) nxt = atomicAdd(st.dyn, 1u);
                st.chunk = __builtin_amdgcn_readfirstlane((int)nxt) - st.done;
            } else st.chunk += NW;
            return true;
        }
        st.chunk -= (total + 63) >> 6;
        st.done += (total + 63) >> 6;
        st.row += 64;
        if (st.row < st.nrows) {                                  // more than 64 grid rows: prepare the next block
            st.prev_rw = __builtin_amdgcn_readlane(st.rw, 63);
            scan_load_rows(st, m, c, cam, lane, X0, TWw, st.row);
        }
    }
    return false;
}

// entry number v of the list that the entry ranges of the current block of grid rows form -> entry index | its grid row within the block << 25
// (-1: the list is shorter).  A lane finds the row by bisection over the rows' first entry numbers (st.rex), see scan_fetch.
__device__ __forceinline__ int scan_locate(const ScanState &st, int nblk, int v) {
    int r = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
        if (step < nblk) {                                // wave-uniform
            const int cand = r + step;
            const int ex = __shfl(st.rex, cand & 63);
            if (cand < nblk && ex <= v) r = cand;
        }
    }
    const int i = __shfl(st.rbase, r) + v;
    return v < st.rtotal ? (i | (r << 25)) : -1;
}

// One entry of the static map's grid (ci = entry index | its grid row within the current block of rows << 25, -1: none; u0, u1 = the
// entry's 32 bytes): the owner rule, then projection and trim.  All lanes take part (shuffles).
__device__ __forceinline__ void scan_candidate(const ScanState &st, const CommonArgs &c, const Camera &cam, int X0, int TWw, int ci, const uint4 &u0,
                                               const uint4 &u1, bool &acc, uint32_t &key, int (&px)[3], int (&py)[3], unsigned &edges) {
    const int i = ci & 0x1ffffff, r = ci >= 0 ? ci >> 25 : 0;
    // what the owner rule needs to know about the entry's grid row (still the rows of the chunk's block: blocks advance in scan_fetch only)
    const int first_end = __shfl(st.rfe, r);                                       // end of the row's first scanned cell
    const int above = __shfl(st.rw, (r + 63) & 63);
    const int pw = r > 0 ? above : st.prev_rw;                                     // cell range of the row above
    const int plo = pw & 0xffff, phi = pw >> 16;
    const bool top = st.row + r == 0;
    if (ci >= 0) {
        const unsigned own = u1.w;              // see GridEntry::own
        // Exactly one of the scanned cells emits the face: in its grid row the first scanned cell of the face's bounding
        // box; among the rows the first one where the bounding box meets the scanned cells (the row above has none).
        const int bx0 = (int)(own & 0x1fffu), bx1 = (int)((own >> 13) & 0x1fffu);
        const bool first_in_row = !(own & (1u << 26)) || (i < first_end);
        const bool none_above = !(own & (1u << 27)) || top || bx1 < plo || bx0 > phi;
        if (first_in_row && none_above && !(TDS_DBG(c.debug) & 1024)) {      // 1024: ablation, walk the grid but project nothing
            unsigned ins = 0;
            float sxv[3] = {__uint_as_float(u0.x) + (-cam.cx), __uint_as_float(u0.z) + (-cam.cx), __uint_as_float(u1.x) + (-cam.cx)};
            float syv[3] = {__uint_as_float(u0.y) + (-cam.cy), __uint_as_float(u0.w) + (-cam.cy), __uint_as_float(u1.y) + (-cam.cy)};
            key = u1.z;
            acc = trim_project(cam, c.scale, c.res, X0, TWw, sxv, syv, px, py, ins, c.no_trim);
            edges = edge_mask(own >> 29, ins);
        }
    }
}

// one producer step: at most one candidate face per lane; returns false when the producer is exhausted
template <int NW = RWAVES, typename SA = SceneArgsEx>      // SceneArgsEx: the per-camera triangle phase is compiled in
__device__ __forceinline__ bool scan_step(ScanState &st, const SA &a, const CommonArgs &c, const Camera &cam, int64_t img, int lane,
                                          int wave, int X0, int TWw, bool &acc, uint32_t &key, int (&px)[3], int (&py)[3], unsigned &edges) {
    const MapView &m = st.map;
    unsigned ins = 0;
    edges = 7u;
    const int res = c.res;
    const int64_t b = img / a.Nc;
    acc = false;
    key = 0;
    if (st.phase == 0) {
        // actors (mesh.py:1071-1103): 7 template vertices per agent, faces [0,1,3],[1,3,2] (body), [4,5,6] (direction).
        // One lane per (agent, face): 21 agents per wave and step (lane 63 idles), dealt round-robin to the cooperating waves;
        // agents are culled by distance before anything else is loaded.
        const int slot = lane / 3, f = lane - 3 * slot;
        const int ag = st.a0 + slot * NW + wave;
        if (lane < 63 && ag < a.N) {
            const int64_t ia = b * a.N + ag;
            const bool on = a.mask[img * a.N + ag] != 0;
            st.masked_seen = st.masked_seen || !on;
            if (on) {
                const float view_r = 1.05f * 1.41421356f / c.scale;        // half diagonal of the trim polygon
                const float4 s = a.state[ia];
                const float2 t0 = a.tmpl[ia * 7];                          // (l/2, w/2): the farthest template vertex
                float reach = view_r + sqrtf(t0.x * t0.x + t0.y * t0.y);
                reach = reach * 1.01f + 0.01f;
                const float ddx = s.x - cam.cx, ddy = s.y - cam.cy;
                if (!(ddx * ddx + ddy * ddy > reach * reach)) {            // NaNs are kept
                    const int64_t ik = a.key_per_cam ? img * a.N + ag : ia;
                    key = a.actor_key[2 * ik + (f == 2 ? 1 : 0)];
                    if (key != 0u) {                                       // key 0: the part does not exist
                        const float2 sc = a.agent_sc[ia];
                        const int v0 = f == 0 ? 0 : (f == 1 ? 1 : 4), v1 = f == 0 ? 1 : (f == 1 ? 3 : 5), v2 = f == 0 ? 3 : (f == 1 ? 2 : 6);
                        const float2 ta = a.tmpl[ia * 7 + v0], tb = a.tmpl[ia * 7 + v1], tc = a.tmpl[ia * 7 + v2];
                        // utils.transform :82-96, then mesh.translate(-cameras.xy) cv2.py:29-31
                        const float fx[3] = {((sc.y * ta.x + (-sc.x) * ta.y) + s.x) + (-cam.cx), ((sc.y * tb.x + (-sc.x) * tb.y) + s.x) + (-cam.cx),
                                             ((sc.y * tc.x + (-sc.x) * tc.y) + s.x) + (-cam.cx)};
                        const float fy[3] = {((sc.x * ta.x + sc.y * ta.y) + s.y) + (-cam.cy), ((sc.x * tb.x + sc.y * tb.y) + s.y) + (-cam.cy),
                                             ((sc.x * tc.x + sc.y * tc.y) + s.y) + (-cam.cy)};
                        acc = trim_project(cam, c.scale, res, X0, TWw, fx, fy, px, py, ins, c.no_trim);
                        edges = edge_mask(f == 1 ? 2u : 0u, ins);      // body faces [0,1,3] and [1,3,2] share the edge 1-3 (edge 1 of the second)
                    }
                }
            }
        }
        st.a0 += 21 * NW;
        if (st.a0 >= a.N) st.phase = 1;
        return true;
    }
    if (st.phase == 1) {
        // every face of a masked agent collapses onto vertex 0 of agent 0 (faces * 0 then concat offsets,
        // mesh.py:1083-1089): a one-pixel dot with the body colour / level of agent 0, under the same trim
        if (__ballot(st.masked_seen) != 0) {
            float4 s0 = a.state[b * a.N];
            float2 sc0 = a.agent_sc[b * a.N];
            float2 t0 = a.tmpl[b * a.N * 7];
            float wx = (sc0.y * t0.x + (-sc0.x) * t0.y) + s0.x, wy = (sc0.x * t0.x + sc0.y * t0.y) + s0.y;
            float fx[3] = {wx + (-cam.cx), wx + (-cam.cx), wx + (-cam.cx)}, fy[3] = {wy + (-cam.cy), wy + (-cam.cy), wy + (-cam.cy)};
            acc = (lane == 0) && trim_project(cam, c.scale, res, X0, TWw, fx, fy, px, py, ins, c.no_trim);
            key = a.actor_key[2 * (a.key_per_cam ? img * a.N : b * a.N)];
        }
        st.phase = 2;
        if constexpr (has_extras<SA>::value) { st.a0 = 0; if (a.K > 0) st.phase = 3; }
        return true;
    }
    if constexpr (has_extras<SA>::value) if (st.phase == 3) {
        // per-camera triangles, already in world coordinates (the host applies generate()'s transform, mesh.py:1120-1145): one lane each
        const int t = st.a0 + wave * 64 + lane;
        if (t < a.K) {
            key = a.extra_key[img * a.K + t];
            if (key != 0u) {
                const float2 *v = (const float2 *)a.extra_tri + (img * a.K + t) * 3;
                const float2 va = v[0], vb = v[1], vc = v[2];
                const float fx[3] = {va.x + (-cam.cx), vb.x + (-cam.cx), vc.x + (-cam.cx)}, fy[3] = {va.y + (-cam.cy), vb.y + (-cam.cy), vc.y + (-cam.cy)};
                acc = trim_project(cam, c.scale, res, X0, TWw, fx, fy, px, py, ins, c.no_trim);
                edges = edge_mask(0u, ins);
            }
        }
        st.a0 += 64 * NW;
        if (st.a0 >= a.K) st.phase = 2;
        return true;
    }
    // static map: the cells of one grid row under the window are ONE contiguous range of entries; chunks of 64 consecutive
    // entries are dealt round-robin to the waves.  The entries of the NEXT chunk are requested before this one is returned, so
    // that their latency is hidden behind the rasterisation of the queue.
    if (!st.have) st.have = scan_fetch<NW>(st, m, c, cam, lane, X0, TWw);
    if (!st.have) return false;
    scan_candidate(st, c, cam, X0, TWw, st.cur_i, st.pu0, st.pu1, acc, key, px, py, edges);
    st.have = scan_fetch<NW>(st, m, c, cam, lane, X0, TWw);
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// The producer of the bit-plane kernels: faces come as POLYS -- one triangle, or two triangles of the same key that share an edge
// (tds_common.h: QuadEntry; the actors' body is such a pair too) -- fetched, projected and trimmed once.  A poly RECORD is what the
// queues and the lists of the split form hold: a flags word and four packed pixel vertices,
//   T1 = (P0, P1, P2), T2 = (P[b0], P[b1], P[b2]) -- each triangle in its own vertex order (cv::fillConvexPoly draws Line(v2,v0),
//   Line(v0,v1), Line(v1,v2) and cv::clipLine depends on the direction) --
//   flags: bits 0..3 plane index (set when the record is queued), 4..6 / 7..9 outline edges of T1 / T2 that have to be drawn (edge_mask),
//          10..15 b0, b1, b2, 16..17 a1 = the vertex of T1 that T2 does not have, bit 18: there is a T2.
// The reference trims face by face (cv2.py:32-41: a face is kept iff one of ITS vertices is in view): a pair of which only one triangle
// survives becomes a lone triangle (T2 re-slotted as T1).
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t PF_HAS2 = 1u << 18;
__device__ __forceinline__ uint32_t sel4(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t i) {
    return i == 0u ? v0 : (i == 1u ? v1 : (i == 2u ? v2 : v3));
}

// qf: QuadEntry::flags (b0..b2, a1, bit 8 has T2, dup bits of T1 at 9, of T2 at 12).  -> accepted; P = packed record vertices, pf = record
// flags (without the plane index), big = a coordinate outside the packed range (the record cannot hold the poly: exact sequential path)
__device__ inline bool trim_project_poly(const Camera &cam, float scale, int res, int X0, int TW, const float *sx, const float *sy, uint32_t qf,
                                         uint32_t (&P)[4], uint32_t &pf, bool &big, int (&px)[4], int (&py)[4], int no_trim) {
    const bool has2 = (qf & 256u) != 0u;
    float fx[4], fy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) project(cam, scale, res, sx[k], sy[k], px[k], py[k], fx[k], fy[k]);
    if (!has2) { px[3] = px[2]; py[3] = py[2]; fx[3] = fx[2]; fy[3] = fy[2]; }
    // one bounding-box reject for the whole entry (a triangle that misses the window paints nothing there either way)
    const int xmin = min(min(px[0], px[1]), min(px[2], px[3])), xmax = max(max(px[0], px[1]), max(px[2], px[3]));
    const int ymin = min(min(py[0], py[1]), min(py[2], py[3])), ymax = max(max(py[0], py[1]), max(py[2], py[3]));
    pf = 0u; big = false;
    if (xmax < X0 || xmin >= X0 + TW || ymax < 0 || ymin >= res) return false;
    unsigned ins = 15u;
    if (!no_trim) {
        // as trim_project: pixel-space test away from the border of the 1.05 x view, the reference's own half-plane test near it
        const float lo = -0.025f * (float)res, hi = 1.025f * (float)res, band = 0.0625f * fmaxf(1.0f, (float)res * (1.0f / 256.0f));
        unsigned amb = 0;
        ins = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float mn = fminf(fx[k], fy[k]), mx = fmaxf(fx[k], fy[k]), sum = fx[k] + fy[k];
            const bool fin = sum == sum;
            const bool in = fin && mn > lo + band && mx < hi - band;
            const bool out = fin && (mn < lo - band || mx > hi + band);
            ins |= in ? (1u << k) : 0u;
            amb |= (!in && !out) ? (1u << k) : 0u;
        }
        if (__builtin_expect(__ballot(amb != 0) != 0, 0)) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((amb >> k) & 1u) ins |= inside_polygon(cam, sx[k], sy[k]) ? (1u << k) : 0u;
        }
    }
    const uint32_t b0 = qf & 3u, b1 = (qf >> 2) & 3u, b2 = (qf >> 4) & 3u, a1 = (qf >> 6) & 3u;
    const bool acc1 = (ins & 7u) != 0u, acc2 = has2 && (ins & (15u & ~(1u << a1))) != 0u;
    if (!acc1 && !acc2) return false;
    big = max(max(max(abs(px[0]), abs(px[1])), max(abs(px[2]), abs(px[3]))), max(max(abs(py[0]), abs(py[1])), max(abs(py[2]), abs(py[3])))) >= COORD_LIMIT;
    const unsigned ins2 = ((ins >> b0) & 1u) | (((ins >> b1) & 1u) << 1) | (((ins >> b2) & 1u) << 2);
    const uint32_t e1 = edge_mask((qf >> 9) & 7u, ins & 7u), e2 = edge_mask((qf >> 12) & 7u, ins2);
    if (acc1) {
        pf = (e1 << 4) | (acc2 ? ((e2 << 7) | ((qf & 63u) << 10) | (a1 << 16) | PF_HAS2) : 0u);
    } else {
        // only T2 is kept: it becomes the record's T1, in its own vertex order
        const int qx[3] = {(int)sel4(px[0], px[1], px[2], px[3], b0), (int)sel4(px[0], px[1], px[2], px[3], b1), (int)sel4(px[0], px[1], px[2], px[3], b2)};
        const int qy[3] = {(int)sel4(py[0], py[1], py[2], py[3], b0), (int)sel4(py[0], py[1], py[2], py[3], b1), (int)sel4(py[0], py[1], py[2], py[3], b2)};
#pragma unroll
        for (int k = 0; k < 3; ++k) { px[k] = qx[k]; py[k] = qy[k]; }
        pf = e2 << 4;
    }
    if (!(pf & PF_HAS2)) { px[3] = px[2]; py[3] = py[2]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) P[k] = pack_xy(px[k], py[k]);
    return true;
}

// one entry of the rendering grid with paired faces (see scan_candidate): the owner rule, then projection and trim of the poly
__device__ __forceinline__ void scan_candidate_poly(const ScanState &st, const CommonArgs &c, const Camera &cam, int X0, int TWw, int ci, const uint4 &u0,
                                                    const uint4 &u1, const uint4 &u2, bool &acc, uint32_t &key, uint32_t (&P)[4], uint32_t &pf, bool &big,
                                                    int (&px)[4], int (&py)[4]) {
    const int i = ci & 0x1ffffff, r = ci >= 0 ? ci >> 25 : 0;
    const int first_end = __shfl(st.rfe, r);
    const int above = __shfl(st.rw, (r + 63) & 63);
    const int pw = r > 0 ? above : st.prev_rw;
    const int plo = pw & 0xffff, phi = pw >> 16;
    const bool top = st.row + r == 0;
    if (ci >= 0) {
        const unsigned own = u2.y;
        const int bx0 = (int)(own & 0x1fffu), bx1 = (int)((own >> 13) & 0x1fffu);
        const bool first_in_row = !(own & (1u << 26)) || (i < first_end);
        const bool none_above = !(own & (1u << 27)) || top || bx1 < plo || bx0 > phi;
        if (first_in_row && none_above && !(TDS_DBG(c.debug) & 1024)) {
            const float sxv[4] = {__uint_as_float(u0.x) + (-cam.cx), __uint_as_float(u0.z) + (-cam.cx), __uint_as_float(u1.x) + (-cam.cx), __uint_as_float(u1.z) + (-cam.cx)};
            const float syv[4] = {__uint_as_float(u0.y) + (-cam.cy), __uint_as_float(u0.w) + (-cam.cy), __uint_as_float(u1.y) + (-cam.cy), __uint_as_float(u1.w) + (-cam.cy)};
            key = u2.x;
            acc = trim_project_poly(cam, c.scale, c.res, X0, TWw, sxv, syv, u2.z, P, pf, big, px, py, c.no_trim);
        }
    }
}

// one producer step of the bit-plane kernels: at most one poly per lane; returns false when the producer is exhausted
template <int NW = RWAVES, typename SA = SceneArgsEx>
__device__ __forceinline__ bool scan_step_poly(ScanState &st, const SA &a, const CommonArgs &c, const Camera &cam, int64_t img, int lane, int wave, int X0,
                                               int TWw, bool &acc, uint32_t &key, uint32_t (&P)[4], uint32_t &pf, bool &big, int (&px)[4], int (&py)[4]) {
    const MapView &m = st.map;
    const int res = c.res;
    const int64_t b = img / a.Nc;
    acc = false; key = 0; pf = 0; big = false;
    if (st.phase == 0) {
        // actors (mesh.py:1071-1103): 7 template vertices per agent, faces [0,1,3], [1,3,2] (the body: ONE poly, the two faces share the edge
        // 1-3) and [4,5,6] (direction).  One lane per (agent, poly): 32 agents per wave and step, dealt round-robin to the cooperating waves;
        // agents are culled by distance before anything else is loaded.
        const int slot = lane >> 1, f = lane & 1;
        const int ag = st.a0 + slot * NW + wave;
        if (ag < a.N) {
            const int64_t ia = b * a.N + ag;
            const bool on = a.mask[img * a.N + ag] != 0;
            st.masked_seen = st.masked_seen || !on;
            if (on) {
                const float view_r = 1.05f * 1.41421356f / c.scale;
                const float4 s = a.state[ia];
                const float2 t0 = a.tmpl[ia * 7];
                float reach = view_r + sqrtf(t0.x * t0.x + t0.y * t0.y);
                reach = reach * 1.01f + 0.01f;
                const float ddx = s.x - cam.cx, ddy = s.y - cam.cy;
                if (!(ddx * ddx + ddy * ddy > reach * reach)) {
                    const int64_t ik = a.key_per_cam ? img * a.N + ag : ia;
                    key = a.actor_key[2 * ik + f];
                    if (key != 0u) {
                        const float2 sc = a.agent_sc[ia];
                        // body: T1 = [0,1,3] in slots 0..2, vertex 2 in slot 3, T2 = [1,3,2] = slots (1,2,3); direction: [4,5,6]
                        const float2 ta = a.tmpl[ia * 7 + (f ? 4 : 0)], tb = a.tmpl[ia * 7 + (f ? 5 : 1)], tc = a.tmpl[ia * 7 + (f ? 6 : 3)], td = a.tmpl[ia * 7 + (f ? 6 : 2)];
                        // utils.transform :82-96, then mesh.translate(-cameras.xy) cv2.py:29-31
                        const float fx[4] = {((sc.y * ta.x + (-sc.x) * ta.y) + s.x) + (-cam.cx), ((sc.y * tb.x + (-sc.x) * tb.y) + s.x) + (-cam.cx),
                                             ((sc.y * tc.x + (-sc.x) * tc.y) + s.x) + (-cam.cx), ((sc.y * td.x + (-sc.x) * td.y) + s.x) + (-cam.cx)};
                        const float fy[4] = {((sc.x * ta.x + sc.y * ta.y) + s.y) + (-cam.cy), ((sc.x * tb.x + sc.y * tb.y) + s.y) + (-cam.cy),
                                             ((sc.x * tc.x + sc.y * tc.y) + s.y) + (-cam.cy), ((sc.x * td.x + sc.y * td.y) + s.y) + (-cam.cy)};
                        // T2's edge 1 (its v0 - v1 = template vertices 1 - 3) repeats T1's
                        const uint32_t qf = f ? 0u : (1u | (2u << 2) | (3u << 4) | (0u << 6) | 256u | (2u << 12));
                        acc = trim_project_poly(cam, c.scale, res, X0, TWw, fx, fy, qf, P, pf, big, px, py, c.no_trim);
                    }
                }
            }
        }
        st.a0 += 32 * NW;
        if (st.a0 >= a.N) st.phase = 1;
        return true;
    }
    if (st.phase == 1) {
        // every face of a masked agent collapses onto vertex 0 of agent 0 (mesh.py:1083-1089): a one-pixel dot, see scan_step
        if (__ballot(st.masked_seen) != 0) {
            float4 s0 = a.state[b * a.N];
            float2 sc0 = a.agent_sc[b * a.N];
            float2 t0 = a.tmpl[b * a.N * 7];
            float wx = (sc0.y * t0.x + (-sc0.x) * t0.y) + s0.x, wy = (sc0.x * t0.x + sc0.y * t0.y) + s0.y;
            const float fx[4] = {wx + (-cam.cx), wx + (-cam.cx), wx + (-cam.cx), wx + (-cam.cx)}, fy[4] = {wy + (-cam.cy), wy + (-cam.cy), wy + (-cam.cy), wy + (-cam.cy)};
            const bool ok = trim_project_poly(cam, c.scale, res, X0, TWw, fx, fy, 0u, P, pf, big, px, py, c.no_trim);
            acc = (lane == 0) && ok;
            key = a.actor_key[2 * (a.key_per_cam ? img * a.N : b * a.N)];
        }
        st.phase = 2;
        if constexpr (has_extras<SA>::value) { st.a0 = 0; if (a.K > 0) st.phase = 3; }
        return true;
    }
    if constexpr (has_extras<SA>::value) if (st.phase == 3) {
        // per-camera triangles, already in world coordinates: one lane each, lone triangles
        const int t = st.a0 + wave * 64 + lane;
        if (t < a.K) {
            key = a.extra_key[img * a.K + t];
            if (key != 0u) {
                const float2 *v = (const float2 *)a.extra_tri + (img * a.K + t) * 3;
                const float2 va = v[0], vb = v[1], vc = v[2];
                const float fx[4] = {va.x + (-cam.cx), vb.x + (-cam.cx), vc.x + (-cam.cx), vc.x + (-cam.cx)}, fy[4] = {va.y + (-cam.cy), vb.y + (-cam.cy), vc.y + (-cam.cy), vc.y + (-cam.cy)};
                acc = trim_project_poly(cam, c.scale, res, X0, TWw, fx, fy, 0u, P, pf, big, px, py, c.no_trim);
            }
        }
        st.a0 += 64 * NW;
        if (st.a0 >= a.K) st.phase = 2;
        return true;
    }
    // static map: as scan_step, over the rendering grid with paired faces
    if (!st.have) st.have = scan_fetch<NW, true>(st, m, c, cam, lane, X0, TWw);
    if (!st.have) return false;
    scan_candidate_poly(st, c, cam, X0, TWw, st.cur_i, st.pu0, st.pu1, st.pu2, acc, key, P, pf, big, px, py);
    st.have = scan_fetch<NW, true>(st, m, c, cam, lane, X0, TWw);
    return true;
}

// Fused single-pass kernel: one workgroup per (camera, strip); every strip scans the grid itself.  Used when the caller
// gives no workspace; also the semantics reference for the binned kernel below.
template <int TW, typename OutT>
__global__ void __launch_bounds__(RBLOCK, 4) raster_scene_kernel(SceneArgsEx a, CommonArgs c) {
    TDS_RASTER_PROLOGUE()
    ScanState st;
    scan_init(st, a, c, cam, img, lane, wave, X0, TW);
    for (;;) {
        bool acc;
        uint32_t key;
        int px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
        unsigned edges;
        const bool more = scan_step(st, a, c, cam, img, lane, wave, X0, TW, acc, key, px, py, edges);
        drain<TW>(w, acc, key, px, py, more);
        if (!more) break;
    }
    __syncthreads();
    if (!(TDS_DBG(c.debug) & 4)) write_out<TW, OutT>(tile, (OutT *)c.out, img, res, X0, tid);
}

// ---- fast path: two kernels ---------------------------------------------------------------------------------
// K3a bin_faces_kernel: ONE WAVE per camera scans the grid once for the whole image (trim, projection) and appends every
// surviving face (16 B: key + three packed vertices) to the list of each strip it touches; no workgroup barriers.
// K3b raster_scene_list_kernel: one workgroup per (camera, strip) rasterises that strip's list.
// Scratch layout: counts[n_img * strips] (uint32) followed by lists[n_img * strips * caps] (uint4).
// A strip whose list overflowed, or that met a face outside the packed coordinate range, is poisoned (count > caps) and
// scans the grid for itself in K3b, so the result never depends on the capacity.
constexpr int BIN_WAVES = 4;          // cameras per workgroup of K3a
constexpr int MAX_STRIPS = 128;

__global__ void __launch_bounds__(BIN_WAVES * 64) bin_faces_kernel(SceneArgsEx a, CommonArgs c, int tw, uint32_t *__restrict__ counts,
                                                                     uint4 *__restrict__ lists, int caps) {
    __shared__ uint32_t cnt_s[BIN_WAVES][MAX_STRIPS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t img = (int64_t)blockIdx.x * BIN_WAVES + wv;
    if (img >= c.n_img) return;                                        // wave-uniform; no barriers in this kernel
    const int res = c.res, W = res, strips = c.strips;
    uint32_t *cnt = cnt_s[wv];
    for (int i = lane; i < strips; i += 64) cnt[i] = 0;
    Camera cam;
    {
        float2 xy = c.cam_xy[img], sc = c.cam_sc[img];
        cam.cx = xy.x; cam.cy = xy.y; cam.s = sc.x; cam.c = sc.y;
        make_polygon(cam, c.scale, res);
    }
    wave_sync();
    uint4 *mine = lists + (size_t)img * strips * caps;
    ScanState st;
    scan_init(st, a, c, cam, img, lane, 0, 0, W);
    for (;;) {
        bool acc;
        uint32_t key;
        int px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
        unsigned edges;
        const bool more = scan_step<1>(st, a, c, cam, img, lane, 0, 0, W, acc, key, px, py, edges);
        if (acc) {
            int xmin = min(px[0], min(px[1], px[2])), xmax = max(px[0], max(px[1], px[2]));
            bool big = max(max(abs(px[0]), abs(px[1])), max(max(abs(px[2]), abs(py[0])), max(abs(py[1]), abs(py[2])))) >= COORD_LIMIT;
            int s0 = max(xmin, 0) / tw, s1 = min(xmax, W - 1) / tw;
            uint4 e = make_uint4(key, pack_xy(px[0], py[0]), pack_xy(px[1], py[1]), pack_xy(px[2], py[2]));
            for (int s = s0; s <= s1; ++s) {
                if (big) { atomicAdd(&cnt[s], 0x40000000u); continue; }      // poison: this strip scans for itself
                uint32_t slot = atomicAdd(&cnt[s], 1u);
                if (slot < (uint32_t)caps) mine[(size_t)s * caps + slot] = e;
            }
        }
        if (!more) break;
    }
    wave_sync();
    for (int i = lane; i < strips; i += 64) counts[img * strips + i] = cnt[i];
}

template <int TW, typename OutT>
__global__ void __launch_bounds__(RBLOCK, 4) raster_scene_list_kernel(SceneArgsEx a, CommonArgs c, const uint32_t *__restrict__ counts,
                                                                       const uint4 *__restrict__ lists, int caps) {
    TDS_RASTER_PROLOGUE()
    const uint32_t n = counts[img * c.strips + strip];
    if (n <= (uint32_t)caps) {
        const uint4 *lst = lists + ((size_t)img * c.strips + strip) * caps;
        for (uint32_t i0 = wave * 64;; i0 += RBLOCK) {
            const bool more = i0 < n;
            bool acc = more && (i0 + lane < n);
            uint4 e = acc ? lst[i0 + lane] : make_uint4(0, 0, 0, 0);
            int px[3] = {unpack_x(e.y), unpack_x(e.z), unpack_x(e.w)}, py[3] = {unpack_y(e.y), unpack_y(e.z), unpack_y(e.w)};
            drain<TW>(w, acc, e.x, px, py, more);
            if (!more) break;
        }
    } else {
        ScanState st;
        scan_init(st, a, c, cam, img, lane, wave, X0, TW);
        for (;;) {
            bool acc;
            uint32_t key;
            int px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
            unsigned edges;
        const bool more = scan_step(st, a, c, cam, img, lane, wave, X0, TW, acc, key, px, py, edges);
            drain<TW>(w, acc, key, px, py, more);
            if (!more) break;
        }
    }
    __syncthreads();
    if (!(TDS_DBG(c.debug) & 4)) write_out<TW, OutT>(tile, (OutT *)c.out, img, res, X0, tid);
}

// =========================================================================================================
// Bit-plane fast path.  A scene uses only a handful of distinct keys (rank + colour: road, lane kinds, vehicle, direction
// ...).  Instead of one u32 per pixel, the workgroup keeps ONE BIT per pixel and key: plane[k][y][x / 32].  Painting a
// span is then one or two ds_or_b32 whatever its length (bits run along OpenCV's x, the span direction), a whole 256x256
// image fits in 8 KiB per key so a single workgroup renders the entire camera (no strips, every face is set up once), and
// the final pass resolves, per pixel, the highest key whose bit is set.  Equal keys paint the same bit, so -- exactly as
// with ds_max on packed keys -- the result does not depend on the order in which faces are processed.
// =========================================================================================================
constexpr int MAX_KEYS = 15;              // key indices 1..15 fit 4 bits (0 = background)
constexpr int EQCAP = 128;                    // ring of queued outline edges per wave (power of two, >= 2 * 64)
constexpr int BITS_WAVE_LDS_DW = Q_DW + 64 + 2 * EQCAP;   // per wave: face queue + owner markers + edge ring

struct KeyTable { uint32_t key[16]; int n; };      // ascending = painter order (later wins)

struct BitCtx {
    uint32_t *planes;   // [K][wpr][H]: plane, word column, row (see paint_span_bits)
    uint32_t *q;        // [4][QCAP]: plane index | outline edges to draw << 4, then the three packed vertices of a triangle
    uint32_t *slots;    // [64] owner markers of wave_owner()
    uint32_t *eq;       // [2][EQCAP] ring of outline edges waiting to be drawn: end points + plane index (pack_xyk)
    int qlen, lane, H, W, X0, TWp, wpr, debug, gen, eq_head, eq_count;
};

// end point of a queued edge: x, y in 15 bits each (|coordinate| < COORD_LIMIT) and two bits of the plane index
__device__ __forceinline__ uint32_t pack_xyk(int x, int y, uint32_t k2) { return (k2 & 3u) | (((uint32_t)x & 0x7fffu) << 2) | ((uint32_t)y << 17); }
__device__ __forceinline__ int unpack_xk(uint32_t p) { return (int)(p << 15) >> 17; }
__device__ __forceinline__ int unpack_yk(uint32_t p) { return (int)p >> 17; }

// bits [s0, s1] (strip-local columns) of one row of one plane.  Layout: plane[k][word column][row] -- the rows of a word column are
// consecutive dwords, so the lanes that paint neighbouring row chunks of one face (same word column, rows 4 apart) fall on different LDS
// banks (the row-major layout put them 32 dwords apart: one bank), and the write-out reads 4 rows of a word with one ds_read_b128.
// `rowp` points at (word column 0, row y); the words of a row are H dwords apart.
__device__ __forceinline__ void paint_span_bits(uint32_t *rowp, int H, int s0, int s1) {
    const int w0 = s0 >> 5, w1 = s1 >> 5;
    const uint32_t m0 = 0xffffffffu << (s0 & 31), m1 = 0xffffffffu >> (31 - (s1 & 31));
    atomicOr(rowp + __umul24((unsigned)w0, (unsigned)H), w0 == w1 ? (m0 & m1) : m0);
    if (w1 > w0) {
        atomicOr(rowp + __umul24((unsigned)w1, (unsigned)H), m1);
        // words in between become all ones (a volatile store through the generic pointer would become a flat store that waits for
        // every outstanding global load of the wave: the OR stays in the LDS pipe)
        for (int wd = w0 + 1; wd < w1; ++wd) atomicOr(rowp + __umul24((unsigned)wd, (unsigned)H), 0xffffffffu);
    }
}

// cv::Line into a bit plane (same walk as draw_line; consecutive pixels that share a word are merged into one ds_or)
// Only the steps kb..ke of the walk are painted (a line is cut into segments that are spread over the lanes).
__device__ inline void draw_line_bits(uint32_t *plane, int H, int W, int X0, int TWp, int wpr, int ax, int ay, int bx, int by, int kb = 0,
                                      int ke = 0x7fffffff) {
    long long x1 = ax, y1 = ay, x2 = bx, y2 = by;
    if ((unsigned long long)x1 >= (unsigned long long)W || (unsigned long long)x2 >= (unsigned long long)W ||
        (unsigned long long)y1 >= (unsigned long long)H || (unsigned long long)y2 >= (unsigned long long)H) {
        if (!clip_line(W, H, x1, y1, x2, y2)) return;
    }
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    int px = (int)x1, py = (int)y1;
    int step_y = 1;
    if (dx < 0) { dx = -dx; dy = -dy; px = (int)x2; py = (int)y2; }
    if (px >= X0 + TWp || px + dx < X0) return;
    if (dy < 0) { dy = -dy; step_y = -1; }
    const bool vert = dy > dx;
    const int dmaj = vert ? dy : dx, dmin = vert ? dx : dy;
    int err = dmaj - (dmin + dmin);
    const int plus_delta = dmaj + dmaj, minus_delta = -(dmin + dmin);
    if (kb > dmaj) return;
    int k = kb;
    if (px < X0) {                                                   // first step whose x reaches the strip
        const int t = X0 - px;
        const int ks = !vert ? t : (int)(((unsigned)(2 * dmaj) * (unsigned)t - (unsigned)dmaj + (unsigned)(2 * dmin)) / (unsigned)(2 * dmin));
        k = max(k, ks);
    }
    if (k > 0) {                                                     // enter the walk at step k in closed form
        int m = (int)(((unsigned)(2 * dmin) * (unsigned)k + (unsigned)dmaj - 1u) / (unsigned)(2 * dmaj));
        err += k * minus_delta + m * plus_delta;
        if (!vert) { px += k; py += step_y * m; }
        else { py += step_y * k; px += m; }
    }
    const int kend = min(ke, dmaj);
    const int lim = X0 + TWp;
    if (vert) {
        // y-major: one pixel per row, every step lands in another word
        uint32_t *rowp = plane + py;
        const int rstep = step_y;
        for (; k <= kend && px < lim; ++k) {
            const int lx = px - X0;
            atomicOr(rowp + (lx >> 5) * H, 1u << (lx & 31));
            const bool neg = err < 0;
            err += minus_delta + (neg ? plus_delta : 0);
            rowp += rstep;
            px += neg ? 1 : 0;
        }
    } else {
        // x-major: consecutive pixels of a row that share a word are merged into one ds_or
        int cur = -1;
        uint32_t mask = 0;
        int rowoff = py;
        const int rstep = step_y;
        for (; k <= kend && px < lim; ++k) {
            const int lx = px - X0;
            const int addr = rowoff + (lx >> 5) * H;
            const uint32_t bit = 1u << (lx & 31);
            if (addr != cur) {
                if (mask) atomicOr(plane + cur, mask);
                cur = addr; mask = bit;
            } else {
                mask |= bit;
            }
            const bool neg = err < 0;
            err += minus_delta + (neg ? plus_delta : 0);
            px += 1;
            rowoff += neg ? rstep : 0;
        }
        if (mask) atomicOr(plane + cur, mask);
    }
}

// exact sequential path for faces outside the packed coordinate range (see fill_generic)
__device__ __noinline__ void fill_generic_bits(uint32_t *plane, int H, int W, int X0, int TWp, int wpr, int x0, int y0, int x1, int y1, int x2,
                                               int y2) {
    const int px[3] = {x0, x1, x2}, py[3] = {y0, y1, y2};
    draw_line_bits(plane, H, W, X0, TWp, wpr, px[2], py[2], px[0], py[0]);
    draw_line_bits(plane, H, W, X0, TWp, wpr, px[0], py[0], px[1], py[1]);
    draw_line_bits(plane, H, W, X0, TWp, wpr, px[1], py[1], px[2], py[2]);
    long long xmin = px[0], xmax = px[0], ymin = py[0], ymax = py[0];
    int imin = 0;
    for (int i = 0; i < 3; ++i) {
        if (py[i] < ymin) { ymin = py[i]; imin = i; }
        if (py[i] > ymax) ymax = py[i];
        if (px[i] > xmax) xmax = px[i];
        if (px[i] < xmin) xmin = px[i];
    }
    if (xmax < 0 || ymax < 0 || xmin >= W || ymin >= H) return;
    if (ymax > H - 1) ymax = H - 1;
    int eidx[2] = {imin, imin}, edi[2] = {1, 2}, eye[2] = {(int)ymin, (int)ymin};
    long long ex[2] = {-65536, -65536}, edx[2] = {0, 0};
    int edges = 3, y = (int)ymin;
    do {
        for (int i = 0; i < 2; ++i) {
            if (y >= eye[i]) {
                int idx0 = eidx[i], idx = idx0 + edi[i];
                if (idx >= 3) idx -= 3;
                for (; edges-- > 0;) {
                    int ty = py[idx];
                    if (ty > y) {
                        long long xs = (long long)px[idx0] << 16, xe = (long long)px[idx] << 16;
                        eye[i] = ty;
                        edx[i] = div_trunc((xe - xs) * 2 + (ty - y), 2ll * (ty - y));
                        ex[i] = xs;
                        eidx[i] = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += edi[i];
                    if (idx >= 3) idx -= 3;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            long long xl = ex[0] < ex[1] ? ex[0] : ex[1], xr = ex[0] < ex[1] ? ex[1] : ex[0];
            long long xx1 = (xl + 32768) >> 16, xx2 = (xr + 32768) >> 16;
            if (xx2 >= 0 && xx1 < W) {
                int s0 = (int)(xx1 < 0 ? 0 : xx1), s1 = (int)(xx2 >= W ? W - 1 : xx2);
                s0 = max(s0, X0); s1 = min(s1, X0 + TWp - 1);
                if (s0 <= s1) paint_span_bits(plane + y, H, s0 - X0, s1 - X0);
            }
        }
        ex[0] += edx[0];
        ex[1] += edx[1];
    } while (++y <= (int)ymax);
}

// Work items are numbered by a prefix sum over their owners (lane j owns the items [excl_j, incl_j)).  For the window of 64
// items starting at `base`, returns in lane i the owner of item base + i: every owner whose first item falls into the window
// drops a marker (generation tag | lane) at that position of a 64-entry LDS array, the owner of the window's first item is
// found with a ballot, and a max-scan spreads the markers (stale markers carry an older tag and lose).
__device__ __forceinline__ int wave_owner(uint32_t *slots, int &gen, int lane, bool has, int excl, int incl, int base) {
    ++gen;
    const int tag = gen << 6;
    const int rel = excl - base;
    if (has && (unsigned)rel < 64u) slots[rel] = (uint32_t)(tag | lane);
    const unsigned long long cm = __ballot(has && excl <= base && base < incl);
    const int carry = cm ? (int)__ffsll((long long)cm) - 1 : 0;
    wave_sync();
    int m = (int)slots[lane];
    if (lane == 0) m = max(m, tag | carry);
    m = wave_scan_max(m);
    return m & 63;
}

// trunc((double)u * (double)v / (double)d) of cv::clipLine for |u v| < 2^30, 0 < |u| <= |d| < 2^16 (exact: see div_trunc)
__device__ __forceinline__ int clip_quot(int u, int v, int d) {
    const int n = __mul24(u, v);
    const unsigned q = udiv_small((unsigned)abs(n), (unsigned)abs(d));
    return ((n < 0) != (d < 0)) ? -(int)q : (int)q;
}

// cv::clipLine for end points within the packed coordinate range (|coordinate| < COORD_LIMIT); same steps as clip_line above
__device__ inline bool clip_line_small(int W, int H, int &x1, int &y1, int &x2, int &y2) {
    const int right = W - 1, bottom = H - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += clip_quot(a - y1, x2 - x1, y2 - y1);
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += clip_quot(a - y2, x2 - x1, y2 - y1);
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += clip_quot(a - x1, y2 - y1, x2 - x1);
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += clip_quot(a - x2, y2 - y1, x2 - x1);
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// Rasterise the first n (<= 64) faces of the wave's queue into the bit planes (process_batch_bits, below):
//   1. lane f sets up face f: vertices by row, the three 16.16 slopes of OpenCV's scan conversion (one short division pair per edge),
//      which outline edges are merged into the rows and which have to be walked exactly; it paints the rows of the three vertices;
//   2. the rows between the vertices are cut into items of CHUNK consecutive rows of one part (top..middle, middle..bottom), numbered by
//      wave prefix sums and mapped back to their owners by wave_owner(); a lane enters its item in closed form and then advances four
//      running sums, one span (one or two ds_or) per row;
//   3. the outline edges that are walked exactly (OpenCV: Line(v2,v0), Line(v0,v1), Line(v1,v2); edges shared with an earlier same-key
//      face are skipped) wait in a per-wave ring and are set up 64 at a time, one per lane: clipLine, left-to-right order, then rewritten
//      top-down so that the pixels of row tau of the walk are a closed form of tau (below);
//   4. items of VCHUNK / HCHUNK rows of such an edge: the pixels of cv::Line in one row are one run, painted like a span.
// Bresenham in closed form.  After the left-to-right swap the walk starts at (px, py), dx >= 0, and after k steps the minor axis has
// advanced m_k = floor((2 dmin k + dmaj - 1) / (2 dmaj)) (see draw_line).  Seen from the TOP end point (x0, ytop), rows tau = 0..|dy|,
// x moving by sgn = +1 (walk goes down) or -1 (walk goes up):
//   y-major: the pixel of row tau is x0 + sgn floor((2 dx tau + c) / (2 |dy|)),  c = |dy| - 1 (sgn > 0) or |dy| (sgn < 0);
//   x-major: row tau holds the offsets G_tau .. min(G_{tau+1} - 1, dx),  G_0 = 0,  G_tau = floor((2 dx tau + c2) / (2 |dy|)),
//            c2 = 2 |dy| - dx - (sgn < 0).
// Both are floor((N0 + tau * 2 dx) / D): stepping tau adds divmod(2 dx, D) = (ia, ib) to (quotient, remainder) with one carry.
// (checked exhaustively against the iterative walk in tests/test_oracle_fill.py::test_line_rows_closed_form)
// optional work counters (profiling hook tds_raster_get_stats of the testing build; active with debug flag 128)
#ifdef TDS_TESTING
__device__ unsigned long long g_stats[16];
#define TDS_STAT_LANES(W, I, V) do { if ((W).debug & 128) { const unsigned long long v_ = (unsigned long long)(V); if (v_) atomicAdd(&g_stats[I], v_); } } while (0)
#define TDS_STAT(W, I, V) do { if (((W).debug & 128) && (W).lane == 0) atomicAdd(&g_stats[I], (unsigned long long)(V)); } while (0)
#else
#define TDS_STAT_LANES(W, I, V) do { } while (0)
#define TDS_STAT(W, I, V) do { } while (0)
#endif
#ifndef TDS_FCHUNK
#define TDS_FCHUNK 4
#endif
#ifndef TDS_VCHUNK
#define TDS_VCHUNK 6
#endif
#ifndef TDS_HCHUNK
#define TDS_HCHUNK 3
#endif
constexpr int CHUNK = TDS_FCHUNK;     // rows per item: scan conversion
constexpr int VCHUNK = TDS_VCHUNK;    // rows per item: y-major outline edges (one pixel per row)
constexpr int HCHUNK = TDS_HCHUNK;    // rows per item: x-major outline edges (one run per row)

// ---- outline edges merged into the rows -------------------------------------------------------------------------------------
// cv::fillConvexPoly paints Line(v2,v0) + Line(v0,v1) + Line(v1,v2) + the scan-converted rows.  For an edge that lies inside the image
// (cv::Line does not clip it) the pixels of the edge in a row and the scan-converted span of that row are ONE run of pixels, and its
// ends follow from the 16.16 edge chain of the scan conversion by an add and a shift -- the edge is never walked:
//   y-major (|dy| > |dx|): cv::Line paints x(tau) = x0 + dx tau / |dy| rounded to nearest, a tie going to the LEFT pixel (either direction
//     of the walk); the span ends at floor(chain + 1/2).  Left end: floor(chain + 1/2 - BIAS), right end unchanged.
//   x-major: row tau holds the pixels x with x(tau - 1/2) < x <= x(tau + 1/2), cut at the edge's end points:
//     floor(chain - slope/2 + BIAS) + 1 .. floor(chain + slope/2 + BIAS), merged with the span by min / max.
// The chain strays from the exact line: OpenCV's slope is trunc(q + 1/2) in units of 2^-16, i.e. within 1/2 unit of q for q >= 0 but
// between 1/2 and 3/2 units ABOVE it for q < 0, per row.  The half-row positions of the exact line are multiples of 1/(2|dy|): BIAS must
// exceed the accumulated error and, with it, stay below 1/(2|dy|) -- 160 units for |dy| <= 100 (3/2 * 100 < 160, 160 + 150 < 32768 / 100).
// Edges of up to 147 rows whose half-row positions are never integers (no ties) need no bias (3/2 * 147 < 32768 / 147).  Everything
// else -- edges that cv::clipLine would touch, longer edges -- is walked exactly (edge ring below).  The rows of the three vertices are
// painted apart, by the face's own lane (the runs are cut at the end points there); the rows in between are work items.
// tests/fill_rows_model.c is this rule as sequential C, checked triangle by triangle against the oracle's cv::fillConvexPoly
// (exhaustively on small grids, millions of random triangles at 64..1024 pixels).
constexpr int MERGE_BIAS = 160, MERGE_DY_BIAS = 100, MERGE_DY_NOBIAS = 147;

// bit 0: the edge is merged into the rows, bit 1: x-major, bit 2: biased
// oca, ocb: outcodes of the end points (vertex_outcode), 0 = inside the image
__device__ __forceinline__ unsigned vertex_outcode(int x, int y, int W, int H) {      // cv::clipLine's: 1 left, 2 right, 4 above, 8 below
    return ((unsigned)x >> 31) | (((unsigned)(W - 1 - x) >> 31) << 1) | (((unsigned)y >> 31) << 2) | (((unsigned)(H - 1 - y) >> 31) << 3);
}
__device__ __forceinline__ unsigned edge_class(int ax, int ay, int bx, int by, unsigned oca, unsigned ocb) {
    const bool inside = (oca | ocb) == 0u;
    const int adx = abs(bx - ax), ady = abs(by - ay);
    const bool xmaj = adx >= ady, shortish = ady <= MERGE_DY_BIAS;
    bool merge = inside && shortish;
    // longer edges (101 .. 147 rows) are merged when they have no ties; hardly any edge is that long, so the test sits behind a wave-uniform branch
    const bool longer = inside && !shortish && ady <= MERGE_DY_NOBIAS;
    if (__ballot(longer) != 0) {
        const int fx = __ffs(adx), fy = __ffs(ady);                   // 1 + trailing zeros, 0 for 0
        const bool tiefree = xmaj ? (fx <= fy) : (adx == 0 || fx >= fy);
        merge = merge || (longer && tiefree);
    }
    return (merge ? 1u : 0u) | (xmaj ? 2u : 0u) | (shortish ? 4u : 0u);
}
// row ends of a chain that follows an edge of class `cls` with 16.16 slope s: left = (x + offL) >> 16, right = (x + offR) >> 16
__device__ __forceinline__ void edge_offsets(unsigned cls, int s, int &offL, int &offR) {
    // selects, no branches: neighbouring lanes hold edges of every class
    const int bias = (cls & 4u) ? MERGE_BIAS : 0;
    const int h = abs(s) >> 1;
    const bool mx = (cls & 3u) == 3u;
    const int yl = 32768 - ((cls & 1u) ? bias : 0);
    const int xl = min(32768, 65536 - h + bias), xr = max(32768, h + bias);
    offL = mx ? xl : yl;
    offR = mx ? xr : 32768;
}
// the pixels of a merged edge in the row of its end point x0: x0 itself and, x-major, the run from x0 towards x0 + d (d = half a row's
// advance along the edge, 16.16, signed)
__device__ __forceinline__ void edge_reach(unsigned cls, int x0, int d, int &L, int &R) {
    if (cls & 1u) {
        L = min(L, x0); R = max(R, x0);
        if (cls & 2u) {
            const int v = ((x0 << 16) + d + ((cls & 4u) ? MERGE_BIAS : 0)) >> 16;
            if (d >= 0) R = max(R, v); else L = min(L, v + 1);
        }
    }
}
__device__ __forceinline__ int half_slope(int s) { return s >= 0 ? (s >> 1) : -((-s) >> 1); }

__device__ __forceinline__ void process_batch_bits(BitCtx &w, int n, bool flush) {
    const int lane = w.lane, H = w.H, W = w.W, X0 = w.X0, TWp = w.TWp, wpr = w.wpr;
    const int Xhi = min(W, X0 + TWp) - 1;                              // last column of the strip (X0 = first)
    wave_sync();
    if (TDS_DBG(w.debug) & 512) return;                       // ablation: no per-face set-up either
    // ---- per-face set-up (lane = face): vertices by row (T, M, B), the three 16.16 slopes, the rows of the vertices painted on the spot,
    //      the rows in between described as two parts (T..M: chains T->M and T->B; M..B: chains M->B and T->B)
    // what a lane keeps of its face: the packed top and middle vertices, the three slopes, the row counts of the two parts (n1 | n2 << 16)
    // and flags = plane index | classes of T->M, M->B, T->B << 4, 7, 10 | outline edges to walk exactly << 13
    int pT = 0, pM = 0, sTB = 0, sTM = 0, sMB = 0, n12 = 0, flags = 0;
    if (lane < n) {
        const uint32_t q0 = w.q[lane];
        const uint32_t v0 = w.q[QCAP + lane], v1 = w.q[2 * QCAP + lane], v2 = w.q[3 * QCAP + lane];
        const uint32_t kidx = q0 & 15u, em = (q0 >> 4) & 7u;
        flags = (int)kidx;
        // packed vertices (x & 0xffff | y << 16) compare like (y, x): sorted by row with three instructions
        const int i0 = (int)v0, i1 = (int)v1, i2 = (int)v2;
        pT = min(i0, min(i1, i2));
        const int pB = max(i0, max(i1, i2));
        pM = max(min(i0, i1), min(max(i0, i1), i2));
        const int xt = unpack_x((uint32_t)pT), yt = unpack_y((uint32_t)pT), xm = unpack_x((uint32_t)pM), ym = unpack_y((uint32_t)pM);
        const int xb = unpack_x((uint32_t)pB), yb = unpack_y((uint32_t)pB);
        const int xmin = min(xt, min(xm, xb)), xmax = max(xt, max(xm, xb));
        const bool hit = !(xmax < X0 || xmin > Xhi || yb < 0 || yt >= H);
        unsigned cTM = 0, cMB = 0, cTB = 0;
        if (hit && !(TDS_DBG(w.debug) & 8)) {
            const unsigned ocT = vertex_outcode(xt, yt, W, H), ocM = vertex_outcode(xm, ym, W, H), ocB = vertex_outcode(xb, yb, W, H);
            cTM = edge_class(xt, yt, xm, ym, ocT, ocM); cMB = edge_class(xm, ym, xb, yb, ocM, ocB); cTB = edge_class(xt, yt, xb, yb, ocT, ocB);
            // Edges that are not merged and not entirely on one outer side of the image (both outcodes share a bit: cv::clipLine leaves nothing)
            // are walked exactly, in OpenCV's order and direction (l = 0: v2-v0, 1: v0-v1, 2: v1-v2; clipLine depends on the direction).
            // Edge l lies opposite vertex (l + 1) % 3; T->M lies opposite B, M->B opposite T, T->B opposite M (vertices that coincide have
            // edges of the same class).
            const unsigned wTM = (!(cTM & 1u) && !(ocT & ocM)) ? 1u : 0u, wMB = (!(cMB & 1u) && !(ocM & ocB)) ? 1u : 0u, wTB = (!(cTB & 1u) && !(ocT & ocB)) ? 1u : 0u;
            unsigned ring = 0;
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                const int po = (int)(l == 0 ? v1 : (l == 1 ? v2 : v0));
                ring |= (po == pB ? wTM : (po == pT ? wMB : wTB)) << l;
            }
            flags |= (int)((cTM << 4) | (cMB << 7) | (cTB << 10) | ((em & ring) << 13));
        }
        if (hit) {
            uint32_t *pl = w.planes + (size_t)__umul24(__umul24(kidx, (unsigned)H), (unsigned)wpr);
            auto paint_row = [&](int y, int L, int R) {
                const int s0 = max(L, X0), s1 = min(R, Xhi);
                if ((unsigned)y < (unsigned)H && s0 <= s1) paint_span_bits(pl + y, H, s0 - X0, s1 - X0);
            };
            if (yt == yb) {
                // all in one row: nothing is scan-converted, the merged (horizontal) edges are spans
                int L = 0x7fffffff, R = -0x7fffffff;
                if (cTM & 1u) { L = min(L, min(xt, xm)); R = max(R, max(xt, xm)); }
                if (cMB & 1u) { L = min(L, min(xm, xb)); R = max(R, max(xm, xb)); }
                if (cTB & 1u) { L = min(L, min(xt, xb)); R = max(R, max(xt, xb)); }
                paint_row(yt, L, R);
            } else {
                sTB = edge_dx(xt, xb, yb - yt);
                sTM = ym > yt ? edge_dx(xt, xm, ym - yt) : 0;
                sMB = yb > ym ? edge_dx(xm, xb, yb - ym) : 0;
                if (!(TDS_DBG(w.debug) & 16)) {
                    // the row of the top vertex (two of them: the span between them): both chains start here
                    {
                        int L = xt, R = xt;
                        if (ym > yt) edge_reach(cTM, xt, half_slope(sTM), L, R);
                        else { L = min(xt, xm); R = max(xt, xm); edge_reach(cMB, xm, half_slope(sMB), L, R); }
                        edge_reach(cTB, xt, half_slope(sTB), L, R);
                        paint_row(yt, L, R);
                    }
                    // the row of the middle vertex when it lies strictly between the others: T->M ends, M->B starts, T->B passes
                    if (ym > yt && ym < yb) {
                        int oL, oR;
                        edge_offsets(cTB, sTB, oL, oR);
                        const int xc = (xt << 16) + (ym - yt) * sTB;
                        int L = min(xm, (xc + oL) >> 16), R = max(xm, (xc + oR) >> 16);
                        edge_reach(cTM, xm, -half_slope(sTM), L, R);
                        edge_reach(cMB, xm, half_slope(sMB), L, R);
                        paint_row(ym, L, R);
                    }
                    // the row of the bottom vertex is never scan-converted: the last pixels of the merged edges that end there
                    {
                        int L = 0x7fffffff, R = -0x7fffffff;
                        edge_reach(cTB, xb, -half_slope(sTB), L, R);
                        if (ym < yb) edge_reach(cMB, xb, -half_slope(sMB), L, R);
                        else {
                            edge_reach(cTM, xm, -half_slope(sTM), L, R);
                            if (cMB & 1u) { L = min(L, min(xm, xb)); R = max(R, max(xm, xb)); }      // the horizontal bottom edge
                        }
                        paint_row(yb, L, R);
                    }
                    // the rows in between: part 1 = rows yt+1 .. ym-1, part 2 = rows ym+1 .. yb-1, cut to the image
                    const int n1 = max(0, min(ym - 1, H - 1) - max(yt + 1, 0) + 1), n2 = max(0, min(yb - 1, H - 1) - max(ym + 1, 0) + 1);
                    n12 = n1 | (n2 << 16);
                }
            }
        }
    }
    TDS_STAT(w, 0, 1); TDS_STAT(w, 1, n);
    // ---- the rows between the vertices: items of CHUNK rows of one part ----
    {
        const int nch = ((n12 & 0xffff) + CHUNK - 1) / CHUNK + ((n12 >> 16) + CHUNK - 1) / CHUNK;
        const int incl = wave_scan_add(nch), excl = incl - nch;
        const int total = __builtin_amdgcn_readlane(incl, 63);
        TDS_STAT(w, 2, total); TDS_STAT(w, 3, (total + 63) / 64);
        TDS_STAT_LANES(w, 4, (n12 & 0xffff) + (n12 >> 16)); TDS_STAT_LANES(w, 15, nch > 0 ? 1 : 0);
        for (int base = 0; base < total; base += 64) {
            const int f = wave_owner(w.slots, w.gen, lane, nch > 0, excl, incl, base);
            const int gT = __shfl(pT, f), gM = __shfl(pM, f), gsTB = __shfl(sTB, f), gsTM = __shfl(sTM, f), gsMB = __shfl(sMB, f);
            const int gn = __shfl(n12, f), gf = __shfl(flags, f), ex = __shfl(excl, f);
            if (base + lane < total) {
                const int t = base + lane - ex, n1c = ((gn & 0xffff) + CHUNK - 1) / CHUNK;
                const bool second = t >= n1c;
                const int pA = second ? gM : gT, sA = second ? gsMB : gsTM;
                const int ys = max(unpack_y((uint32_t)pA) + 1, 0), nr = second ? gn >> 16 : gn & 0xffff;
                const int y0 = ys + CHUNK * (second ? t - n1c : t), y1 = min(y0 + CHUNK, ys + nr) - 1;
                const unsigned cA = ((unsigned)gf >> (second ? 7 : 4)) & 7u, cB = ((unsigned)gf >> 10) & 7u;
                int oLa, oRa, oLb, oRb;
                edge_offsets(cA, sA, oLa, oRa);
                edge_offsets(cB, gsTB, oLb, oRb);
                const int xa = (unpack_x((uint32_t)pA) << 16) + (y0 - unpack_y((uint32_t)pA)) * sA;
                const int xb = (unpack_x((uint32_t)gT) << 16) + (y0 - unpack_y((uint32_t)gT)) * gsTB;
                int la = xa + oLa, ra = xa + oRa, lb = xb + oLb, rb = xb + oRb;
                uint32_t *rowp = w.planes + (size_t)(__umul24(__umul24((unsigned)(gf & 15), (unsigned)H), (unsigned)wpr) + (unsigned)y0);
#pragma unroll
                for (int i = 0; i < CHUNK; ++i) {
                    if (y0 + i <= y1) {
                        // OpenCV draws the span clamped to the image unless it lies entirely outside; the merged edges lie inside
                        const int s0 = max(min(la, lb) >> 16, X0), s1 = min(max(ra, rb) >> 16, Xhi);
                        if (s0 <= s1) paint_span_bits(rowp, H, s0 - X0, s1 - X0);
                    }
                    la += sA; ra += sA; lb += gsTB; rb += gsTB;
                    rowp += 1;
                }
            }
        }
    }
    // ---- outline edges that are walked exactly ----
    // They go through a per-wave ring of EQCAP entries (two packed end points + plane index) that lives across batches: 64 of them are
    // taken at a time, so that the per-edge set-up and the row items below run on full waves.
    // (ablations of the testing build: 8 = no edge classes and no walk -- every edge unmerged, nothing walked --, 256 = the classes stay, only the walk goes)
    if (!(TDS_DBG(w.debug) & (8 | 256))) {
        TDS_STAT_LANES(w, 5, __popc(((unsigned)flags >> 13) & 7u));
#pragma unroll 1
        for (int l = 0; l < 4; ++l) {
            if (l < 3) {
                // push edge l (0: v2-v0, 1: v0-v1, 2: v1-v2) of every face that has to walk it
                const bool has = (((unsigned)flags >> (13 + l)) & 1u) != 0;
                const unsigned long long bm = __ballot(has);
                if (bm == 0) continue;
                if (has) {
                    const uint32_t v0 = w.q[QCAP + lane], v1 = w.q[2 * QCAP + lane], v2 = w.q[3 * QCAP + lane];
                    const uint32_t pa = l == 0 ? v2 : (l == 1 ? v0 : v1), pb = l == 0 ? v0 : (l == 1 ? v1 : v2);
                    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
                    const int slot = (w.eq_head + w.eq_count + rank) & (EQCAP - 1);
                    w.eq[slot] = pack_xyk(unpack_x(pa), unpack_y(pa), (uint32_t)flags & 3u);
                    w.eq[EQCAP + slot] = pack_xyk(unpack_x(pb), unpack_y(pb), ((uint32_t)flags >> 2) & 3u);
                }
                w.eq_count += __popcll(bm);
                wave_sync();
            }
            while (w.eq_count >= 64 || (l == 3 && flush && w.eq_count > 0)) {
            const int take = min(w.eq_count, 64);
            const bool valid = lane < take;
            const int slot = (w.eq_head + lane) & (EQCAP - 1);
            const uint32_t pa = valid ? w.eq[slot] : 0u, pb = valid ? w.eq[EQCAP + slot] : 0u;
            w.eq_head = (w.eq_head + take) & (EQCAP - 1);
            w.eq_count -= take;
            const int ek = (int)((pa & 3u) | ((pb & 3u) << 2));
            TDS_STAT(w, 6, 1);
            int x1 = unpack_xk(pa), y1 = unpack_yk(pa), x2 = unpack_xk(pb), y2 = unpack_yk(pb);
            bool ok = valid;
            const bool outside = (unsigned)x1 >= (unsigned)W || (unsigned)x2 >= (unsigned)W || (unsigned)y1 >= (unsigned)H || (unsigned)y2 >= (unsigned)H;
            if (__ballot(ok && outside) != 0) {
                if (ok && outside) ok = clip_line_small(W, H, x1, y1, x2, y2);
            }
            int dx = x2 - x1, dy = y2 - y1, sx = x1, sy = y1;
            if (dx < 0) { dx = -dx; dy = -dy; sx = x2; sy = y2; }
            ok = ok && !(sx >= X0 + TWp || sx + dx < X0);                    // the x range misses the strip
            const int ady = abs(dy);
            const bool up = dy < 0, vert = ady > dx;
            const int x0 = up ? sx + dx : sx, ytop = up ? sy - ady : sy;
            // x-major edges: per-row increment of (quotient, remainder) = divmod(2 dx, 2 |dy|)
            int ia = 0, ib = 0;
            if (!vert && ady > 0) { ia = (int)udiv_small((unsigned)dx, (unsigned)ady); ib = 2 * (dx - __mul24(ia, ady)); }
            const int e1 = (ytop & 0xffff) | (x0 << 16), e2 = dx | (ady << 16), e3 = ia | (ib << 16), e4 = ek | (up ? 16 : 0);
            // ---- y-major edges: one pixel per row, items of VCHUNK rows
            {
                const int nch = (ok && vert) ? (ady + VCHUNK) / VCHUNK : 0;
                const int rincl = wave_scan_add(nch), rexcl = rincl - nch;
                const int rtotal = __builtin_amdgcn_readlane(rincl, 63);
                TDS_STAT(w, 7, rtotal); TDS_STAT(w, 8, (rtotal + 63) / 64);
                TDS_STAT_LANES(w, 9, (ok && vert) ? ady + 1 : 0); TDS_STAT_LANES(w, 13, (ok && vert) ? 1 : 0);
                for (int rbase = 0; rbase < rtotal; rbase += 64) {
                    const int e = wave_owner(w.slots, w.gen, lane, nch > 0, rexcl, rincl, rbase);
                    const int g1 = __shfl(e1, e), g2 = __shfl(e2, e), g4 = __shfl(e4, e), gx = __shfl(rexcl, e);
                    const bool live = rbase + lane < rtotal;
                    const int tau0 = VCHUNK * (rbase + lane - gx);
                    const int gdx = g2 & 0xffff, gady = g2 >> 16, D = 2 * gady, A = 2 * gdx;
                    const bool gup = (g4 & 16) != 0;
                    int q = 0, rem = gady - (gup ? 0 : 1);                  // row 0: N = c < D
                    if (__ballot(live && tau0 > 0) != 0) {
                        const unsigned N = (unsigned)(__mul24(A, tau0) + rem);
                        q = (int)udiv_small(N, (unsigned)max(D, 1));
                        rem = (int)N - __mul24(q, D);
                    }
                    if (live) {
                        const int gx0 = g1 >> 16, ytop_e = (int)(short)(g1 & 0xffff);
                        const int nrow = min(VCHUNK, gady - tau0 + 1);
                        uint32_t *rowp = w.planes + (size_t)(__umul24(__umul24((unsigned)(g4 & 15), (unsigned)H), (unsigned)wpr) + (unsigned)(ytop_e + tau0));
#pragma unroll
                        for (int i = 0; i < VCHUNK; ++i) {
                            const int lx = (gup ? gx0 - q : gx0 + q) - X0;
                            if (i < nrow && (unsigned)lx < (unsigned)TWp) atomicOr(rowp + __umul24((unsigned)(lx >> 5), (unsigned)H), 1u << (lx & 31));
                            rem += A;
                            const bool carry = rem >= D;
                            rem -= carry ? D : 0;
                            q += carry ? 1 : 0;
                            rowp += 1;
                        }
                    }
                }
            }
            // ---- x-major edges: one run per row, items of HCHUNK rows
            {
                const int nch = (ok && !vert) ? (ady + HCHUNK) / HCHUNK : 0;
                const int rincl = wave_scan_add(nch), rexcl = rincl - nch;
                const int rtotal = __builtin_amdgcn_readlane(rincl, 63);
                TDS_STAT(w, 10, rtotal); TDS_STAT(w, 11, (rtotal + 63) / 64);
                TDS_STAT_LANES(w, 12, (ok && !vert) ? ady + 1 : 0); TDS_STAT_LANES(w, 14, (valid && outside) ? 1 : 0);
                for (int rbase = 0; rbase < rtotal; rbase += 64) {
                    const int e = wave_owner(w.slots, w.gen, lane, nch > 0, rexcl, rincl, rbase);
                    const int g1 = __shfl(e1, e), g2 = __shfl(e2, e), g3 = __shfl(e3, e), g4 = __shfl(e4, e), gx = __shfl(rexcl, e);
                    if (rbase + lane < rtotal) {
                        const int tau0 = HCHUNK * (rbase + lane - gx);
                        const int gdx = g2 & 0xffff, gady = g2 >> 16, D = 2 * gady;
                        const int gia = g3 & 0xffff, gib = (int)((unsigned)g3 >> 16);
                        const bool gup = (g4 & 16) != 0;
                        // state: q = G_{tau+1} = floor(N / D), rem = N - q D; the row runs from lo = G_tau to q - 1
                        int q, rem = 0, lo = 0;
                        if (gady == 0) {                                     // horizontal: one row, offsets 0..dx
                            q = gdx + 1;
                        } else {
                            const unsigned N = (unsigned)(__mul24(2 * gdx, tau0) + gdx + D - (gup ? 1 : 0));
                            q = (int)udiv_small(N, (unsigned)D);
                            rem = (int)N - __mul24(q, D);
                            if (tau0 > 0) lo = q - gia - (rem < gib ? 1 : 0);
                        }
                        const int gx0 = g1 >> 16, ytop_e = (int)(short)(g1 & 0xffff);
                        const int nrow = min(HCHUNK, gady - tau0 + 1);
                        uint32_t *rowp = w.planes + (size_t)(__umul24(__umul24((unsigned)(g4 & 15), (unsigned)H), (unsigned)wpr) + (unsigned)(ytop_e + tau0));
#pragma unroll
                        for (int i = 0; i < HCHUNK; ++i) {
                            if (i < nrow) {
                                const int h0 = min(q - 1, gdx);
                                const int xs = gup ? gx0 - h0 : gx0 + lo, xe = gup ? gx0 - lo : gx0 + h0;
                                const int s0 = max(xs, X0), s1 = min(xe, X0 + TWp - 1);
                                if (s0 <= s1) paint_span_bits(rowp, H, s0 - X0, s1 - X0);
                            }
                            lo = q;
                            rem += gib;
                            const bool carry = rem >= D;
                            rem -= carry ? D : 0;
                            q += gia + (carry ? 1 : 0);
                            rowp += 1;
                        }
                    }
                }
            }
            }       // while: 64 queued edges at a time
        }
    }
    wave_sync();
}

// The short set-up for SMALL faces (K3r sorts them out chunk by chunk: all three vertices inside the image, in one row or in two adjacent rows).  Such a
// face has nothing between its vertex rows to scan-convert, every outline edge lies inside the image and is short, i.e. merged into the rows
// (edge_class: 1 | x-major << 1 | 4) and nothing goes to the edge ring -- what is left of process_batch_bits is the painting of the vertex
// rows, with the SAME expressions (edge_reach over the half slopes of OpenCV's 16.16 chains); the slope over one row,
// edge_dx(xs, xe, 1) = trunc((xe - xs) * 65536 + 1/2) with C's truncation, is (xe - xs) << 16 for xe >= xs and ((xe - xs) << 16) + 1 below.
// lane = face; `e` = the face's list entry (plane | outline edges << 4, three packed vertices); faces that miss the strip are skipped.
__device__ __forceinline__ int slope_one_row(int xs, int xe) { const int d = xe - xs; return d >= 0 ? d << 16 : (int)(((unsigned)d << 16) + 1u); }
// TALL = false: faces of one or two rows; TALL = true: up to four rows (yb - yt <= 3) -- the one or two rows between the top and the bottom
// vertex are painted by the lane too, each either the row of the middle vertex or a row of part 1 (chains T->M, T->B) / part 2 (M->B, T->B)
// with the expressions of process_batch_bits' row items.  Which of the two serves a launch is decided by the resolution (wave-uniform): the
// taller variant costs every chunk of faces about twice the instructions and pays where faces of three and four rows are many (128 x 128: 38 %
// of the faces on top of the 32 % of one and two rows; 64 x 64: 8 % on top of 66 %).
template <bool TALL>
__device__ __forceinline__ void process_small_bits(BitCtx &w, bool valid, uint32_t plane, uint32_t v0, uint32_t v1, uint32_t v2) {
    const int H = w.H, X0 = w.X0, wpr = w.wpr;
    const int Xhi = min(w.W, X0 + w.TWp) - 1;
    if (!valid) return;
    const int i0 = (int)v0, i1 = (int)v1, i2 = (int)v2;
    const int pT = min(i0, min(i1, i2)), pB = max(i0, max(i1, i2)), pM = max(min(i0, i1), min(max(i0, i1), i2));
    const int xt = unpack_x((uint32_t)pT), yt = unpack_y((uint32_t)pT), xm = unpack_x((uint32_t)pM), ym = unpack_y((uint32_t)pM);
    const int xb = unpack_x((uint32_t)pB), yb = unpack_y((uint32_t)pB);
    const int xmin = min(xt, min(xm, xb)), xmax = max(xt, max(xm, xb));
    if (xmax < X0 || xmin > Xhi) return;
    uint32_t *pl = w.planes + (size_t)__umul24(__umul24(plane & 15u, (unsigned)H), (unsigned)wpr);
    auto paint_row = [&](int y, int L, int R) {
        const int s0 = max(L, X0), s1 = min(R, Xhi);
        if (s0 <= s1) paint_span_bits(pl + y, H, s0 - X0, s1 - X0);
    };
    if (yt == yb) { paint_row(yt, xmin, xmax); return; }         // one row: the three (horizontal, merged) edges are one span
    // classes of the merged short edges: 1 | x-major << 1 | biased (edge_class with both end points inside the image and |dy| <= 100)
    auto cls = [](int ax, int ay, int bx, int by) { return 5u | ((abs(bx - ax) >= abs(by - ay)) ? 2u : 0u); };
    const unsigned cTM = cls(xt, yt, xm, ym), cMB = cls(xm, ym, xb, yb), cTB = cls(xt, yt, xb, yb);
    int sTB, sTM, sMB;
    if constexpr (TALL) {
        sTB = edge_dx(xt, xb, yb - yt);
        sTM = ym > yt ? edge_dx(xt, xm, ym - yt) : 0;
        sMB = yb > ym ? edge_dx(xm, xb, yb - ym) : 0;
    } else {
        sTB = slope_one_row(xt, xb); sTM = ym > yt ? slope_one_row(xt, xm) : 0; sMB = yb > ym ? slope_one_row(xm, xb) : 0;
    }
    {
        int L = xt, R = xt;
        if (ym > yt) edge_reach(cTM, xt, half_slope(sTM), L, R);
        else { L = min(xt, xm); R = max(xt, xm); edge_reach(cMB, xm, half_slope(sMB), L, R); }
        edge_reach(cTB, xt, half_slope(sTB), L, R);
        paint_row(yt, L, R);
    }
    if constexpr (TALL) {
        int oLb, oRb;
        edge_offsets(cTB, sTB, oLb, oRb);
#pragma unroll
        for (int j = 1; j <= 2; ++j) {
            const int y = yt + j;
            if (y >= yb) break;
            const int xc = (xt << 16) + j * sTB;                     // the chain T->B in this row
            if (y == ym) {
                // the row of the middle vertex: T->M ends, M->B starts, T->B passes
                int L = min(xm, (xc + oLb) >> 16), R = max(xm, (xc + oRb) >> 16);
                edge_reach(cTM, xm, -half_slope(sTM), L, R);
                edge_reach(cMB, xm, half_slope(sMB), L, R);
                paint_row(y, L, R);
            } else {
                const bool second = y > ym;                          // part 2: the other chain is M->B, else T->M
                const int sA = second ? sMB : sTM, xA = second ? xm : xt, yA = second ? ym : yt;
                int oLa, oRa;
                edge_offsets(second ? cMB : cTM, sA, oLa, oRa);
                const int xa = (xA << 16) + (y - yA) * sA;
                paint_row(y, min(xa + oLa, xc + oLb) >> 16, max(xa + oRa, xc + oRb) >> 16);
            }
        }
    }
    {
        int L = 0x7fffffff, R = -0x7fffffff;
        edge_reach(cTB, xb, -half_slope(sTB), L, R);
        if (ym < yb) edge_reach(cMB, xb, -half_slope(sMB), L, R);
        else {
            edge_reach(cTM, xm, -half_slope(sTM), L, R);
            L = min(L, min(xm, xb)); R = max(R, max(xm, xb));          // the horizontal bottom edge (merged)
        }
        paint_row(yb, L, R);
    }
}

// plane of a key = its position in the ascending table (in LDS: broadcast reads)
__device__ __forceinline__ int key_plane(const uint32_t *keys, int K, uint32_t key) {
    int k = 0;
#pragma unroll 1
    for (int i = 0; i < K; ++i) k += (keys[i] < key) ? 1 : 0;
    return k;
}

// BIG: faces outside the packed coordinate range may come along (they take the exact sequential path); the list kernel never sees one
template <bool BIG = true>
__device__ __forceinline__ void drain_bits(BitCtx &w, int k, bool acc, unsigned edges, const int (&px)[3], const int (&py)[3], bool more) {
    if constexpr (BIG) {
        bool big = acc && (max(max(abs(px[0]), abs(px[1])), max(max(abs(px[2]), abs(py[0])), max(abs(py[1]), abs(py[2])))) >= COORD_LIMIT);
        if (__builtin_expect(__ballot(big) != 0, 0)) {
            if (big) fill_generic_bits(w.planes + (size_t)k * w.H * w.wpr, w.H, w.W, w.X0, w.TWp, w.wpr, px[0], py[0], px[1], py[1], px[2], py[2]);
            acc = acc && !big;
        }
    }
    unsigned long long pending = __ballot(acc);
    for (;;) {
        if (pending != 0) {
            const int room = QCAP - w.qlen;
            int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(pending >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pending, 0));
            bool take = acc && ((pending >> w.lane) & 1) && rank < room;
            if (take) {
                int slot = w.qlen + rank;
                w.q[slot] = (uint32_t)k | (edges << 4);           // plane index | outline edges to draw
                w.q[1 * QCAP + slot] = pack_xy(px[0], py[0]);
                w.q[2 * QCAP + slot] = pack_xy(px[1], py[1]);
                w.q[3 * QCAP + slot] = pack_xy(px[2], py[2]);
            }
            unsigned long long taken = __ballot(take);
            w.qlen += __popcll(taken);
            pending &= ~taken;
        }
        if (w.qlen == QCAP || (!more && pending == 0)) {          // the last call also empties the edge queue
            process_batch_bits(w, w.qlen, !more && pending == 0);
            w.qlen = 0;
        }
        if (pending == 0) break;
    }
}

// Queue the triangles of the poly records K3r reads from its camera's list (at most one poly per lane: one or two triangles, each in its own
// vertex order; acc1 / acc2: which of them still have to be rasterised) and rasterise the queue whenever it is full.  `flags`: the poly
// record's.  The rasteriser works triangle by triangle (painting the rows of a pair as one is exact -- tests/fill_quads_model.c -- and slower:
// DESIGN_HISTORY.md, appendix R1).
__device__ __forceinline__ void drain_poly(BitCtx &w, bool acc1, bool acc2, uint32_t flags, const uint32_t (&P)[4], bool more) {
    const uint32_t b0 = (flags >> 10) & 3u, b1 = (flags >> 12) & 3u, b2 = (flags >> 14) & 3u;
    // two rounds: the first triangles of all polys, then the second ones (a triangle of the queue is plane | outline edges << 4 + three vertices)
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const bool mine = h == 0 ? acc1 : acc2;
        const bool last = more ? false : (h == 1);
        if (h == 1 && __ballot(mine) == 0) { if (!more) { process_batch_bits(w, w.qlen, true); w.qlen = 0; } break; }
        const uint32_t q0 = (flags & 15u) | (((flags >> (h ? 7 : 4)) & 7u) << 4);
        const uint32_t v0 = h ? sel4(P[0], P[1], P[2], P[3], b0) : P[0], v1 = h ? sel4(P[0], P[1], P[2], P[3], b1) : P[1], v2 = h ? sel4(P[0], P[1], P[2], P[3], b2) : P[2];
        unsigned long long pending = __ballot(mine);
        for (;;) {
            if (pending != 0) {
                const int room = QCAP - w.qlen;
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(pending >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pending, 0));
                const bool take = mine && ((pending >> w.lane) & 1) && rank < room;
                if (take) {
                    const int slot = w.qlen + rank;
                    w.q[slot] = q0;
                    w.q[1 * QCAP + slot] = v0;
                    w.q[2 * QCAP + slot] = v1;
                    w.q[3 * QCAP + slot] = v2;
                }
                const unsigned long long taken = __ballot(take);
                w.qlen += __popcll(taken);
                pending &= ~taken;
            }
            if (w.qlen == QCAP || (last && pending == 0)) {          // the last call also empties the edge queue
                process_batch_bits(w, w.qlen, last && pending == 0);
                w.qlen = 0;
            }
            if (pending == 0) break;
        }
    }
}

// Resolve the planes (highest key wins) and stream the strip out; one item = 4 consecutive rows x 32 columns.
// The K planes of a row are first reduced to NB bit-slices of the winning key's index (0 = background).  Pixels leave in PAIRS of
// vertically adjacent rows: the 2*NB index bits of a pair address a table of ready-made output values (two floats per channel), so
// that a pixel costs no arithmetic after its index has been extracted.  Extraction: slice i rotated left by i, masked to every 8th
// column and OR-ed together puts the complete pair index of 4 columns (phase, phase+8, ...) into one word.
template <int NB, typename OutT> struct PairTab;
template <int NB> struct PairTab<NB, float> { using E = float2; };
template <int NB> struct PairTab<NB, uint8_t> { using E = uint32_t; };        // low 16 bits: the two bytes of a pair

__device__ __forceinline__ uint32_t rotl32(uint32_t v, int n) { return __builtin_rotateleft32(v, (unsigned)n & 31u); }

template <int BBLOCK, int NB, typename OutT, bool EMIT>
__device__ __forceinline__ void write_out_bits(const uint32_t *planes, const typename PairTab<NB, OutT>::E *tab, int K, OutT *out, int64_t img, int res,
                                      int X0, int TWp, int wpr, int tid, uint32_t *slices) {
    using E = typename PairTab<NB, OutT>::E;
    constexpr int P = 1 << (2 * NB);
    const int H = res, W = res;
    const int64_t plane_px = (int64_t)W * H;
    OutT *o = out + img * 3 * plane_px;
    const int cols = min(TWp, W - X0);
    if constexpr (sizeof(OutT) == 1 && NB <= 3 && !EMIT) {
        // uint8 output, at most seven keys: no table in LDS at all.  The colours of the eight key indices are two dwords per channel, and
        // v_perm_b32 looks four pixels up at once: its selector bytes are the key indices of four rows of a column.  Per four columns: the
        // index bits of a row are gathered into one byte per column (slice b rotated by b, masked to every 8th column), the 4 x 4 bytes of
        // four rows are transposed with eight more v_perm, and a lane stores four rows of a column per channel.  Strips tall enough to give
        // every thread an item of EIGHT rows x 32 columns take those: 8-byte stores, half the store instructions (with fewer items than
        // threads the 4-row items keep more lanes busy).  uint8 256 x 256: the stream-out was 0.65 of 5.4 ms, now 0.3.
        if ((H & 3) == 0) {
            uint32_t tlo[3], thi[3];                                 // colour bytes of the key indices 0..3 / 4..7 per channel (index 0 = background)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int i = 1; i < 8; ++i) {
                    const uint32_t v = (uint32_t)tab[ch * P + i] & 255u;           // pair (index i, background): the low byte is index i's value
                    if (i < 4) lo |= v << (8 * i); else hi |= v << (8 * (i - 4));
                }
                tlo[ch] = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo); thi[ch] = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
            }
            const bool full = (cols & 31) == 0;
            char *ob[3] = {(char *)o, (char *)(o + plane_px), (char *)(o + 2 * plane_px)};
            auto run = [&](auto rows_tag) {
                constexpr int ROWS = decltype(rows_tag)::value, HALVES = ROWS / 4;
                const int groups = H / ROWS;
                for (int item = tid; item < groups * wpr; item += BBLOCK) {
                    const int rg = item % groups, xw = item / groups, y0 = rg * ROWS;
                    if (xw * 32 >= cols) continue;
                    uint32_t s[NB][ROWS], cov[ROWS];
#pragma unroll
                    for (int j = 0; j < ROWS; ++j) { cov[j] = 0; for (int b = 0; b < NB; ++b) s[b][j] = 0; }
                    for (int k = K - 1; k >= 0; --k) {               // wave-uniform
                        const int idx = k + 1;
                        const uint4 *pw = (const uint4 *)(planes + ((size_t)k * wpr + xw) * H + y0);
                        uint32_t wds[ROWS];
#pragma unroll
                        for (int h = 0; h < HALVES; ++h) { const uint4 wq = pw[h]; wds[4 * h] = wq.x; wds[4 * h + 1] = wq.y; wds[4 * h + 2] = wq.z; wds[4 * h + 3] = wq.w; }
#pragma unroll
                        for (int j = 0; j < ROWS; ++j) {
                            const uint32_t sn = wds[j] & ~cov[j];
                            cov[j] |= wds[j];
#pragma unroll
                            for (int b = 0; b < NB; ++b) s[b][j] |= ((idx >> b) & 1) ? sn : 0u;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < ROWS; ++j)
#pragma unroll
                        for (int b = 1; b < NB; ++b) s[b][j] = rotl32(s[b][j], b);
                    const int ncol = min(32, cols - xw * 32);
                    const uint32_t off0 = (uint32_t)((X0 + xw * 32) * H + y0), colb = (uint32_t)H;
                    auto emit8 = [&](auto check) {
                        constexpr bool CHECK = decltype(check)::value;
#pragma unroll
                        for (int ph = 0; ph < 8; ++ph) {
                            uint32_t B[ROWS];                        // B[j]: byte m = key index of (row j, column ph + 8 m)
#pragma unroll
                            for (int j = 0; j < ROWS; ++j) {
                                uint32_t v = 0;
#pragma unroll
                                for (int b = 0; b < NB; ++b) v |= s[b][j] & (0x01010101u << ((ph + b) & 7));
                                B[j] = rotl32(v, 32 - ph);
                            }
                            uint32_t C[HALVES][4];                   // C[h][m]: byte r = key index of (row 4 h + r, column ph + 8 m)
#pragma unroll
                            for (int h = 0; h < HALVES; ++h) {
                                const uint32_t a_lo = __builtin_amdgcn_perm(B[4 * h + 1], B[4 * h + 0], 0x05010400u), a_hi = __builtin_amdgcn_perm(B[4 * h + 1], B[4 * h + 0], 0x07030602u);
                                const uint32_t b_lo = __builtin_amdgcn_perm(B[4 * h + 3], B[4 * h + 2], 0x05010400u), b_hi = __builtin_amdgcn_perm(B[4 * h + 3], B[4 * h + 2], 0x07030602u);
                                C[h][0] = __builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u); C[h][1] = __builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u);
                                C[h][2] = __builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u); C[h][3] = __builtin_amdgcn_perm(b_hi, a_hi, 0x07060302u);
                            }
#pragma unroll
                            for (int m = 0; m < 4; ++m) {
                                const int pcol = ph + 8 * m;
                                if (CHECK && pcol >= ncol) continue;
                                const uint32_t off = off0 + (uint32_t)pcol * colb;
#pragma unroll
                                for (int ch = 0; ch < 3; ++ch) {
                                    if constexpr (ROWS == 8) {
                                        typedef uint32_t vu2 __attribute__((ext_vector_type(2)));
                                        const vu2 v = {__builtin_amdgcn_perm(thi[ch], tlo[ch], C[0][m]), __builtin_amdgcn_perm(thi[ch], tlo[ch], C[1][m])};
                                        __builtin_nontemporal_store(v, (vu2 *)(ob[ch] + off));
                                    } else {
                                        __builtin_nontemporal_store(__builtin_amdgcn_perm(thi[ch], tlo[ch], C[0][m]), (uint32_t *)(ob[ch] + off));
                                    }
                                }
                            }
                        }
                    };
                    if (full) emit8(std::false_type{}); else emit8(std::true_type{});
                }
            };
            if ((H & 7) == 0 && (H >> 3) * wpr >= BBLOCK) run(std::integral_constant<int, 8>{}); else run(std::integral_constant<int, 4>{});
            return;
        }
    }
    if ((H & 3) == 0) {
        const int quads = H >> 2;
        const bool full = (cols & 31) == 0;                          // wave-uniform: no partial 32-column word
        for (int item = tid; item < quads * wpr; item += BBLOCK) {
            // A wave takes 64 consecutive row quads of one word column: its plane reads are consecutive 16-byte pieces of LDS (conflict-free
            // ds_read_b128) and a store instruction writes one run of 1 KiB.
            const int rq = item % quads, xw = item / quads, y0 = rq * 4;
            if (xw * 32 >= cols) continue;
            uint32_t s[NB][4], cov[4] = {0, 0, 0, 0};
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[b][j] = 0;
            for (int k = K - 1; k >= 0; --k) {                       // wave-uniform
                const int idx = k + 1;
                // four rows of one word column: 16 contiguous bytes (H and y0 are multiples of 4), consecutive row quads in consecutive lanes
                const uint4 w4 = *(const uint4 *)(planes + ((size_t)k * wpr + xw) * H + y0);
                const uint32_t wds[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t wd = wds[j];
                    uint32_t sn = wd & ~cov[j];
                    cov[j] |= wd;
#pragma unroll
                    for (int b = 0; b < NB; ++b) s[b][j] |= ((idx >> b) & 1) ? sn : 0u;
                }
            }
            if constexpr (EMIT) {
                // the resolved index of these 4 rows x 32 columns, as NB bit-slices: 16 B per slice, 64 B apart per (word column, row quad)
                uint4 *dst = (uint4 *)(slices + ((((size_t)img * (size_t)((W + 31) >> 5) + (size_t)((X0 >> 5) + xw)) * (size_t)quads + (size_t)rq) << 4));
#pragma unroll
                for (int b = 0; b < NB; ++b) dst[b] = make_uint4(s[b][0], s[b][1], s[b][2], s[b][3]);
                // the unused slices are written too (zeros): every 64-byte record is then stored completely, consecutive lanes fill whole cache
                // lines and nothing has to be merged with old memory contents (partial records made this instantiation 12 % slower than the plain one)
#pragma unroll
                for (int b = NB; b < 4; ++b) dst[b] = make_uint4(0u, 0u, 0u, 0u);
            }
            uint32_t R0[2 * NB], R1[2 * NB];                         // pair (rows 0,1) and pair (rows 2,3)
#pragma unroll
            for (int i = 0; i < 2 * NB; ++i) {
                R0[i] = rotl32(i < NB ? s[i % NB][0] : s[i % NB][1], i);
                R1[i] = rotl32(i < NB ? s[i % NB][2] : s[i % NB][3], i);
            }
            const int ncol = min(32, cols - xw * 32);
            // byte offsets from the (uniform) image base fit 32 bits: 3 * 4096^2 * 4 B at most
            const uint32_t off0 = (uint32_t)(((X0 + xw * 32) * H + y0) * (int)sizeof(OutT)), colb = (uint32_t)(H * (int)sizeof(OutT));
            const char *ob0 = (const char *)o, *ob1 = (const char *)(o + plane_px), *ob2 = (const char *)(o + 2 * plane_px);
            auto emit = [&](auto check) {
                constexpr bool CHECK = decltype(check)::value;
#pragma unroll
                for (int ph = 0; ph < 8; ++ph) {
                    uint32_t A0 = 0, A1 = 0;
#pragma unroll
                    for (int i = 0; i < 2 * NB; ++i) {
                        const uint32_t Mx = 0x01010101u << ((ph + i) & 7);      // every 8th column, bit i of its pair index
                        A0 |= R0[i] & Mx;
                        A1 |= R1[i] & Mx;
                    }
                    A0 = rotl32(A0, 32 - ph);
                    A1 = rotl32(A1, 32 - ph);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int p = ph + 8 * m;
                        if (CHECK && p >= ncol) continue;
                        const uint32_t i01 = (A0 >> (8 * m)) & (uint32_t)(P - 1), i23 = (A1 >> (8 * m)) & (uint32_t)(P - 1);
                        const uint32_t off = off0 + (uint32_t)p * colb;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            const E lo = tab[ch * P + i01], hi = tab[ch * P + i23];
                            char *dst = (char *)(ch == 0 ? ob0 : (ch == 1 ? ob1 : ob2)) + off;
                            // streamed out, never read again by this kernel: non-temporal, so that the grid entries keep their place in L2
                            if constexpr (sizeof(OutT) == 4) { typedef float vf4 __attribute__((ext_vector_type(4))); const vf4 v = {lo.x, lo.y, hi.x, hi.y}; __builtin_nontemporal_store(v, (vf4 *)dst); }
                            else __builtin_nontemporal_store(lo | (hi << 16), (uint32_t *)dst);
                        }
                    }
                }
            };
            if (full) emit(std::false_type{}); else emit(std::true_type{});
        }
        return;
    }
    for (int i = tid; i < H * cols; i += BBLOCK) {                   // odd resolutions: one pixel per thread
        const int lx = i / H, y = i - lx * H;
        int idx = 0;
        for (int k = K - 1; k >= 0 && idx == 0; --k)
            if ((planes[((size_t)k * wpr + (lx >> 5)) * H + y] >> (lx & 31)) & 1) idx = k + 1;
        const int64_t off = (int64_t)(X0 + lx) * H + y;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const E e = tab[ch * P + idx];
            if constexpr (sizeof(OutT) == 4) o[ch * plane_px + off] = e.x;
            else o[ch * plane_px + off] = (OutT)(e & 255u);
        }
    }
}

template <int NB, typename OutT>
constexpr int pair_tab_dw() { return 3 * (1 << (2 * NB)) * (int)sizeof(typename PairTab<NB, OutT>::E) / 4; }

// Work distribution of the bit-plane kernel.  The launch is PERSISTENT: a fixed number of workgroups (a few per CU), each of which takes
// one (camera, strip) after the other from a queue until none is left.  There is one queue per XCD -- a contiguous eighth of the launch, so
// that neighbouring cameras share an L2 for grid cells and every XCD streams into its own region of the output (scenes dealt round-robin to
// the XCDs cost 15 %) -- and a workgroup whose own queue has run dry STEALS from the queues of the other XCDs.  Why: the write path is not
// symmetric.  With equal shares the even XCDs of this part finish their eighth 9 % (float32: 6.5 against 7.1 ms; on a "slow" output
// allocation 7.2 against 8.3 ms) before the odd ones, which then write the tail at half the aggregate bandwidth (tools/xcd_finish_times.py;
// the imbalance is in the store stream: a launch that rasterises nothing shows it, a launch that stores nothing does not).
// `queue`: 8 counters, zero at launch: the last 64 bytes of the CALLER's workspace, cleared in stream order right before the launch (the
// library owns no device memory besides the map handles).  A workgroup reads the XCD it runs on from the XCC_ID hardware register.
// Without a workspace there is no queue and the same kernel runs one workgroup per item (claim_work).
constexpr int BITS_FIXED_DW = 20;             // [0..14] ascending key table, [15] next chunk of the grid scan, [16] the work item taken next,
                                              // [17] queues this workgroup has found empty
// one thread: take the next item -- own queue first, then the others in cyclic order -- and leave it in state[0] (-1: nothing left);
// state[1] = queues of that order already found empty by this workgroup.  Everything a workgroup carries from item to item lives in
// LDS: the kernel has no register to spare
__device__ __noinline__ void claim_work(uint32_t *queue, int nblk, uint32_t *state, const uint32_t *only, int strips) {
    if (queue == nullptr) {
        // no queue (the caller gave no workspace): ONE item per workgroup, by blockIdx -- the launch then has one workgroup per item.
        // state[1] doubles as "this workgroup has had its item"
        int64_t img;
        int strip;
        block_to_image(nblk, strips, img, strip);
        state[0] = state[1] == 0 ? (uint32_t)(img * strips + strip) : 0xffffffffu;
        state[1] = 1;
        return;
    }
    if (only != nullptr) {
        // the launch that follows the split form: only the cameras K3s marked (only[0] of them, listed from only[1]), one queue
        const uint32_t j = atomicAdd(&queue[0], 1u), n = only[0] * (uint32_t)strips;
        state[0] = j < n ? only[1 + j / (uint32_t)strips] * (uint32_t)strips + j % (uint32_t)strips : 0xffffffffu;
        return;
    }
    // which XCD this workgroup really runs on: the XCC_ID hardware register (id 20, bits 3:0 on gfx950), not blockIdx.x & 7 -- workgroups
    // are dealt round-robin to the XCDs, but where the round starts is the dispatcher's business once another stream's kernel is in flight
    const int nq = (nblk & 7) == 0 ? 8 : 1, per = nq == 8 ? nblk >> 3 : nblk;
    const int xcd = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & (unsigned)(nq - 1));
    int tried = (int)state[1], item = -1;
    for (; tried < nq; ++tried) {
        const int v = (xcd + tried) & (nq - 1);
        const uint32_t i = atomicAdd(&queue[v], 1u);
        if (i < (uint32_t)per) { item = v * per + (int)i; break; }
    }
    state[1] = (uint32_t)tried;
    state[0] = (uint32_t)item;
}

// The instantiations that run three workgroups per CU are persistent (work queues above; also 4 % on the uint8 mode: no gap between
// the end of one workgroup and the start of the next on its slot).  The 128-VGPR instantiations are not: the loop over the items costs
// them another 64 bytes of spills per lane, which outweighs it (128 x 128: 3.95 -> 4.17 ms); they take ONE item, by blockIdx.
template <int BWAVES, int NB, typename OutT, typename SA, bool EMIT = false, int MINWG = 3>
__global__ void __launch_bounds__(BWAVES * 64, BWAVES == 4 ? MINWG : 4) raster_scene_bits_kernel(SA a, CommonArgs c, KeyTable kt, int TWp, uint32_t *queue, const uint32_t *only) {
    using E = typename PairTab<NB, OutT>::E;
    constexpr int BBLOCK = BWAVES * 64, P = 1 << (2 * NB);
    constexpr bool PERSIST = BWAVES == 4 && MINWG == 3;              // (the 8-wave instantiation as a persistent launch: 224 instead of 160 B of scratch, 6 % slower)
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int res = c.res, H = res, W = res, wpr = TWp >> 5, K = kt.n;
    const int plane_dw = K * H * wpr;
    // LDS: everything of fixed size first (addresses known at compile time), the planes (K * H * wpr words) last
    E *tab = (E *)smem;                                                  // [3][P] output values of an index pair
    uint32_t *lkeys = smem + pair_tab_dw<NB, OutT>();                     // [16] ascending key table, [16] the work item taken next
    uint32_t *planes = lkeys + BITS_FIXED_DW + BWAVES * BITS_WAVE_LDS_DW;
    // the launch's work items: (camera, strip) pairs, as eight queues of consecutive items when they divide evenly, else as one
    if constexpr (PERSIST) {
        if (tid == 0) { lkeys[17] = 0; claim_work(queue, (int)(c.n_img * c.strips), lkeys + 16, only, c.strips); }
    }
#ifdef TDS_TESTING
    // debug flag 4096: when does each XCD start and finish its share of the launch?  (wall clock, 100 MHz; blocks are dealt round-robin to the XCDs)
    if ((c.debug & 4096) && tid == 0) atomicMax(&g_stats[8 + (blockIdx.x & 7)], ~(unsigned long long)wall_clock64());
#endif
    if (tid < 16) {
        uint32_t kv = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 16; ++i) kv = (tid == i) ? kt.key[i] : kv;     // kt lives in SGPRs: no dynamic indexing
        lkeys[tid] = kv;                                                   // entry 15 is no key (K <= 15): the next chunk of the grid scan, set per item
    }
    __syncthreads();
    for (int e = tid; e < 3 * P; e += BBLOCK) {
        const int ch = e / P, pr = e - ch * P, ilo = pr & ((1 << NB) - 1), ihi = pr >> NB;
        const uint32_t klo = (ilo >= 1 && ilo <= K) ? lkeys[ilo - 1] : 0u, khi = (ihi >= 1 && ihi <= K) ? lkeys[ihi - 1] : 0u;
        const int sh = 16 - 8 * ch;
        const uint32_t vlo = (klo >> sh) & 255u, vhi = (khi >> sh) & 255u;
        if constexpr (sizeof(OutT) == 4) tab[e] = make_float2((float)vlo, (float)vhi);
        else tab[e] = vlo | (vhi << 8);
    }
    BitCtx w;
    w.planes = planes;
    w.q = lkeys + BITS_FIXED_DW + wave * BITS_WAVE_LDS_DW;
    w.slots = w.q + Q_DW;
    w.eq = w.slots + 64;
    w.lane = lane; w.H = H; w.W = W; w.TWp = TWp; w.wpr = wpr; w.debug = c.debug;
#pragma unroll 1
    for (;;) {
        int64_t img;
        int X0;
        if constexpr (PERSIST) {
            __syncthreads();                                                 // the next item is known; every wave has left the planes of the previous one
            const int item = __builtin_amdgcn_readfirstlane((int)lkeys[16]);
            if (item < 0) break;
            img = item / c.strips;
            X0 = (item - (int)img * c.strips) * TWp;
        } else {
            int strip;
            block_to_image(c.n_img * c.strips, c.strips, img, strip);
            X0 = strip * TWp;
        }
        for (int i = tid * 4; i < plane_dw; i += BBLOCK * 4) *(uint4 *)(planes + i) = make_uint4(0, 0, 0, 0);
        if (tid == 15) lkeys[15] = (uint32_t)BWAVES;
        // uint8 output (bound by instruction issue): waves that rasterise take precedence over waves of other workgroups that are streaming out;
        // float32 output is bound by the write stream and loses by the raise (DESIGN_HISTORY.md, appendix R2)
        if constexpr (sizeof(OutT) != 4) __builtin_amdgcn_s_setprio(1);
        w.eq_head = 0; w.eq_count = 0;
        w.slots[lane] = 0;
        w.qlen = 0; w.X0 = X0; w.gen = 0;
        // Scale and resolution are the same for every item, and the compiler would compute what the item prologue derives from them (the
        // corners of the trim polygon, the window of the grid scan: a dozen divisions) once, before the loop over the items -- and then
        // hold the results in registers through the rasterisation of every item, where the kernel has none to spare.  The empty asm makes
        // them per-item values.
        CommonArgs ci = c;
        if constexpr (PERSIST) {
            int sb = __float_as_int(c.scale), rb = c.res;
            asm volatile("" : "+s"(sb), "+s"(rb));
            ci.scale = __int_as_float(sb); ci.res = rb;
        }
        Camera cam;
        {
            float2 xy = c.cam_xy[img], sc = c.cam_sc[img];
            cam.cx = xy.x; cam.cy = xy.y; cam.s = sc.x; cam.c = sc.y;
            make_polygon(cam, ci.scale, ci.res);
        }
        __syncthreads();
        // (the fused kernel walks the grid of single triangles: the 48-byte entries of the grid with paired faces cost it four more registers for
        // the chunk in flight, which it does not have -- DESIGN_HISTORY.md, appendix R3; the split form's scan kernel takes the pairs)
        ScanState st;
        scan_init(st, a, ci, cam, img, lane, wave, X0, TWp);
        st.dyn = lkeys + 15;
        for (;;) {
            bool acc;
            uint32_t key;
            int px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
            unsigned edges;
            const bool more = scan_step<BWAVES, SA>(st, a, ci, cam, img, lane, wave, X0, TWp, acc, key, px, py, edges);
            drain_bits(w, key_plane(lkeys, K, key), acc, edges, px, py, more);
            if (!more) break;
        }
        // the next item is taken now (every wave read the current one two barriers ago): the round trip of the atomic hides behind the stream-out
        if constexpr (PERSIST) { if (tid == 0) claim_work(queue, (int)(c.n_img * c.strips), lkeys + 16, only, c.strips); }
        __syncthreads();
        if constexpr (sizeof(OutT) != 4) __builtin_amdgcn_s_setprio(0);
        if (!(TDS_DBG(c.debug) & 4)) write_out_bits<BBLOCK, NB, OutT, EMIT>(planes, tab, K, (OutT *)c.out, img, res, X0, TWp, wpr, tid, c.slices);
        if constexpr (!PERSIST) break;
    }
#ifdef TDS_TESTING
    if ((c.debug & 4096) && lane == 0) atomicMax(&g_stats[blockIdx.x & 7], (unsigned long long)wall_clock64());
#endif
}

// =========================================================================================================
// The split form of the bit-plane path.  Wherever the launch is bound by instruction issue and latency rather than by the write stream
// (uint8 output, resolutions below 256 x 256) the fused kernel above pays for carrying the grid scan -- its state, the camera, the trim
// test, the faces that wait for room in the queue -- through the rasterisation: registers the row loops then spill.  Here
//   K3s  scan_faces_kernel        one wavefront per camera walks the grid and the actors ONCE and writes the accepted faces of the whole image,
//                                 projected and classified (plane index | outline edges to draw, three packed vertices: 16 bytes), to a
//                                 list in the caller's workspace;
//   K3r  raster_list_bits_kernel  one workgroup per (camera, strip) reads the list and rasterises: no scan state, and strips no longer
//                                 cost a grid scan each, so the strip width is chosen for occupancy alone.
// A camera whose list overflowed (more than `caps` faces) or that met a face outside the packed coordinate range is marked in `counts`
// and rendered by the fused kernel afterwards (a launch over the marked cameras only), so the result never depends on the capacity.
// =========================================================================================================
constexpr int SCAN_WAVES = 4;                 // cameras per workgroup of K3s
#ifndef TDS_SMALL_TALL_RES
#define TDS_SMALL_TALL_RES 112
#endif
constexpr int SMALL_TALL_RES = TDS_SMALL_TALL_RES;      // from this resolution on K3r's short path takes faces of up to four rows (below: two)
#ifndef TDS_SCAN_DEPTH
#define TDS_SCAN_DEPTH 1
#endif
#ifndef TDS_SCAN_OCC
#define TDS_SCAN_OCC 5
#endif
constexpr int SCAN_DEPTH = TDS_SCAN_DEPTH;    // chunks of 64 grid entries whose loads K3s keeps in flight
constexpr uint32_t LIST_POISON = 0xffffffffu;

// (SCAN_DEPTH chunks in flight x TDS_SCAN_OCC waves per SIMD: the sweep is DESIGN_HISTORY.md, appendix R4; 1 x 5 is its optimum and
// tests/test_kernel_resources.py holds the kernel to the 112 bytes of scratch it has there)
template <typename SA>
__global__ void __launch_bounds__(SCAN_WAVES * 64, TDS_SCAN_OCC) scan_faces_kernel(SA a, CommonArgs c, KeyTable kt, uint32_t *__restrict__ counts, uint4 *__restrict__ lists,
                                                                     uint32_t *__restrict__ lists3, int caps, uint32_t *__restrict__ poisoned) {
    __shared__ uint32_t lkeys[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 16) {
        uint32_t kv = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 16; ++i) kv = ((int)threadIdx.x == i) ? kt.key[i] : kv;
        lkeys[threadIdx.x] = kv;
    }
    __syncthreads();
    const int64_t img = (int64_t)blockIdx.x * SCAN_WAVES + wv;
    if (img >= c.n_img) return;                                        // wave-uniform; no further barriers
    const int res = c.res, K = kt.n;
    Camera cam;
    {
        float2 xy = c.cam_xy[img], sc = c.cam_sc[img];
        cam.cx = xy.x; cam.cy = xy.y; cam.s = sc.x; cam.c = sc.y;
        make_polygon(cam, c.scale, res);
    }
    // the list of this camera: poly records (scan_step_poly) -- flags | plane index, P0, P1, P2 in `lists`, P3 in `lists3`
    uint4 *mine = lists + (size_t)img * caps;
    uint32_t *mine3 = lists3 + (size_t)img * caps;
    // (actors and the masked-agent dot run as in the kernel without per-camera triangles -- the SceneArgs steps, whatever SA is -- and the
    // per-camera triangles get a loop of their own below: as a fourth phase of the producer's state machine they cost this kernel 400
    // instead of 112 bytes of scratch per lane)
    const SceneArgs &ab = a;
    ScanState st;
    scan_init<SceneArgs, true>(st, ab, c, cam, img, lane, 0, 0, res);
    int count = 0;
    bool poison = false;
    auto emit = [&](bool acc, uint32_t key, uint32_t pf, const uint32_t (&P)[4], bool big) {
        poison = poison || (__ballot(acc && big) != 0);
        const unsigned long long bm = __ballot(acc);
        if (bm != 0) {
            const int k = key_plane(lkeys, K, key);
            const int slot = count + __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
            if (acc && slot < caps) { mine[slot] = make_uint4((uint32_t)k | pf, P[0], P[1], P[2]); mine3[slot] = P[3]; }
            count += __popcll(bm);
        }
    };
    // actors, the masked-agent dot: the producer steps of the fused kernel, up to the static map
    while (st.phase != 2) {
        bool acc, big;
        uint32_t key, pf, P[4] = {0, 0, 0, 0};
        int px[4], py[4];
        (void)scan_step_poly<1, SceneArgs>(st, ab, c, cam, img, lane, 0, 0, res, acc, key, P, pf, big, px, py);
        emit(acc, key, pf, P, big);
    }
    // per-camera triangles, already in world coordinates (scan_step_poly, phase 3): one lane each, lone triangles
    if constexpr (has_extras<SA>::value) {
        for (int t0 = 0; t0 < a.K; t0 += 64) {
            bool acc = false, big = false;
            uint32_t key = 0, pf = 0, P[4] = {0, 0, 0, 0};
            int px[4], py[4];
            const int t = t0 + lane;
            if (t < a.K) {
                key = a.extra_key[img * a.K + t];
                if (key != 0u) {
                    const float2 *v = (const float2 *)a.extra_tri + (img * a.K + t) * 3;
                    const float2 va = v[0], vb = v[1], vc = v[2];
                    const float fx[4] = {va.x + (-cam.cx), vb.x + (-cam.cx), vc.x + (-cam.cx), vc.x + (-cam.cx)}, fy[4] = {va.y + (-cam.cy), vb.y + (-cam.cy), vc.y + (-cam.cy), vc.y + (-cam.cy)};
                    acc = trim_project_poly(cam, c.scale, res, 0, res, fx, fy, 0u, P, pf, big, px, py, c.no_trim);
                }
            }
            emit(acc, key, pf, P, big);
        }
    }
    // The static map.  With nothing to rasterise between two chunks of entries the walk would wait for every load (1.2 us each):
    // the entries are taken SCAN_DEPTH chunks at a time, all their loads in flight together.
    if (!(TDS_DBG(c.debug) & 1)) {
        const MapView &m = st.map;
        while (st.row < st.nrows) {
            const int nblk = min(64, st.nrows - st.row);
            for (int base = 0; base < st.rtotal; base += 64 * SCAN_DEPTH) {
                int ci[SCAN_DEPTH];
                uint4 u0[SCAN_DEPTH], u1[SCAN_DEPTH], u2[SCAN_DEPTH];
#pragma unroll
                for (int u = 0; u < SCAN_DEPTH; ++u) {
                    ci[u] = (base + 64 * u < st.rtotal) ? scan_locate(st, nblk, base + 64 * u + lane) : -1;
                    u0[u] = u1[u] = u2[u] = make_uint4(0, 0, 0, 0);
                    if (ci[u] >= 0) {
                        const uint4 *ep = (const uint4 *)(m.qentries + (ci[u] & 0x1ffffff));
                        u0[u] = ep[0]; u1[u] = ep[1]; u2[u] = ep[2];
                    }
                }
#pragma unroll
                for (int u = 0; u < SCAN_DEPTH; ++u) {
                    if (base + 64 * u >= st.rtotal) break;                   // wave-uniform
                    bool acc = false, big = false;
                    uint32_t key = 0, pf = 0, P[4] = {0, 0, 0, 0};
                    int px[4], py[4];
                    scan_candidate_poly(st, c, cam, 0, res, ci[u], u0[u], u1[u], u2[u], acc, key, P, pf, big, px, py);
                    emit(acc, key, pf, P, big);
                }
            }
            st.row += 64;
            if (st.row < st.nrows) {                                         // more than 64 grid rows: the next block
                st.prev_rw = __builtin_amdgcn_readlane(st.rw, 63);
                scan_load_rows(st, m, c, cam, lane, 0, res, st.row);
            }
        }
    }
    poison = poison || count > caps;
    if (lane == 0) {
        counts[img] = poison ? LIST_POISON : (uint32_t)count;
        if (poison) poisoned[1 + atomicAdd(poisoned, 1u)] = (uint32_t)img;          // poisoned[0] = how many, then which
    }
}

// K3r.  75 - 81 VGPRs, no scratch: LDS, not registers, sets how many workgroups share a CU.  NOT a persistent launch (the loop over the items
// costs 30 VGPRs and a fifth of the waves).  BWAVES: wavefronts per workgroup -- a small image holds too few chunks of faces to keep four waves in
// step between the barriers of a workgroup, so small images get two waves per workgroup and more workgroups.  (Sweeps: DESIGN_HISTORY.md, appendix R5.)
template <int NB, typename OutT, int BWAVES>
__global__ void __launch_bounds__(BWAVES * 64) raster_list_bits_kernel(CommonArgs c, KeyTable kt, int TWp, const uint32_t *__restrict__ counts,
                                                                      const uint4 *__restrict__ lists, const uint32_t *__restrict__ lists3, int caps) {
    using E = typename PairTab<NB, OutT>::E;
    constexpr int BBLOCK = BWAVES * 64, P = 1 << (2 * NB);
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int res = c.res, H = res, W = res, wpr = TWp >> 5, K = kt.n;
    const int plane_dw = K * H * wpr;
    E *tab = (E *)smem;
    uint32_t *lkeys = smem + pair_tab_dw<NB, OutT>();
    uint32_t *planes = lkeys + BITS_FIXED_DW + BWAVES * BITS_WAVE_LDS_DW;
    int64_t img;
    int strip;
    block_to_image(c.n_img * c.strips, c.strips, img, strip);
    const uint32_t n = counts[img];
    if (n == LIST_POISON) return;                                      // uniform over the workgroup: the fused kernel renders this camera
    const int X0 = strip * TWp;
    for (int i = tid * 4; i < plane_dw; i += BBLOCK * 4) *(uint4 *)(planes + i) = make_uint4(0, 0, 0, 0);
    if (tid < 16) {
        uint32_t kv = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < 16; ++i) kv = (tid == i) ? kt.key[i] : kv;
        lkeys[tid] = tid == 15 ? (uint32_t)BWAVES : kv;                    // entry 15: the next chunk of the lists
    }
    __syncthreads();
    for (int e = tid; e < 3 * P; e += BBLOCK) {
        const int ch = e / P, pr = e - ch * P, ilo = pr & ((1 << NB) - 1), ihi = pr >> NB;
        const uint32_t klo = (ilo >= 1 && ilo <= K) ? lkeys[ilo - 1] : 0u, khi = (ihi >= 1 && ihi <= K) ? lkeys[ihi - 1] : 0u;
        const int sh = 16 - 8 * ch;
        const uint32_t vlo = (klo >> sh) & 255u, vhi = (khi >> sh) & 255u;
        if constexpr (sizeof(OutT) == 4) tab[e] = make_float2((float)vlo, (float)vhi);
        else tab[e] = vlo | (vhi << 8);
    }
    BitCtx w;
    w.planes = planes;
    w.q = lkeys + BITS_FIXED_DW + wave * BITS_WAVE_LDS_DW;
    w.slots = w.q + Q_DW;
    w.eq = w.slots + 64;
    w.eq_head = 0; w.eq_count = 0;
    w.slots[lane] = 0;
    w.qlen = 0; w.lane = lane; w.H = H; w.W = W; w.X0 = X0; w.TWp = TWp; w.wpr = wpr; w.debug = c.debug; w.gen = 0;
    __syncthreads();
    const uint4 *lst = lists + (size_t)img * caps;
    const uint32_t *lst3 = lists3 + (size_t)img * caps;
    const int Xhi = min(W, X0 + TWp) - 1;
    int chunk = __builtin_amdgcn_readfirstlane(wave);                  // chunks of 64 faces, dealt to the waves as they ask
    for (;;) {
        const uint32_t i0 = (uint32_t)chunk * 64u;
        const bool more = i0 < n;
        bool acc = more && (i0 + (uint32_t)lane < n);
        uint4 e = make_uint4(0, 0, 0, 0);
        uint32_t e3 = 0;
        if (acc) { e = lst[i0 + lane]; e3 = lst3[i0 + lane]; }
        if (more) {
            unsigned nxt = 0;
            if (lane == 0) nxt = atomicAdd(lkeys + 15, 1u);
            chunk = __builtin_amdgcn_readfirstlane((int)nxt);
        }
        const uint32_t P[4] = {e.y, e.z, e.w, e3};
        const bool has2 = (e.x & PF_HAS2) != 0u;
        const uint32_t b0 = (e.x >> 10) & 3u, b1 = (e.x >> 12) & 3u, b2 = (e.x >> 14) & 3u;
        const uint32_t T[3] = {sel4(P[0], P[1], P[2], P[3], b0), sel4(P[0], P[1], P[2], P[3], b1), sel4(P[0], P[1], P[2], P[3], b2)};     // the second triangle
        // per triangle: does it touch this strip, and is it SMALL -- its three vertices inside the image, in one row or in two adjacent rows
        // (up to four from SMALL_TALL_RES on): two thirds of the faces of a 64 x 64 view, a third at 128 x 128.  Small triangles are painted on
        // the spot by their lane (process_small_bits: nothing to scan-convert, no edge to clip or walk); only the others go through the queue
        // and the full set-up.
        const int span = res >= SMALL_TALL_RES ? 3 : 1;                    // wave-uniform
        auto classify = [&](uint32_t v0, uint32_t v1, uint32_t v2, bool on, bool &small) {
            const int x0 = unpack_x(v0), x1 = unpack_x(v1), x2 = unpack_x(v2), y0 = unpack_y(v0), y1 = unpack_y(v1), y2 = unpack_y(v2);
            const int xmin = min(x0, min(x1, x2)), xmax = max(x0, max(x1, x2)), ymin = min(y0, min(y1, y2)), ymax = max(y0, max(y1, y2));
            if (c.strips > 1) on = on && !(xmax < X0 || xmin > Xhi);       // triangles that miss this strip are not queued
            small = on && !(TDS_DBG(c.debug) & 32768) && xmin >= 0 && xmax < W && ymin >= 0 && ymax < H && ymax - ymin <= span;
            return on;
        };
        bool small1, small2;
        const bool on1 = classify(P[0], P[1], P[2], acc, small1), on2 = classify(T[0], T[1], T[2], acc && has2, small2);
        if (res >= SMALL_TALL_RES) {
            process_small_bits<true>(w, small1, e.x, P[0], P[1], P[2]);
            process_small_bits<true>(w, small2, e.x, T[0], T[1], T[2]);
        } else {
            process_small_bits<false>(w, small1, e.x, P[0], P[1], P[2]);
            process_small_bits<false>(w, small2, e.x, T[0], T[1], T[2]);
        }
        drain_poly(w, on1 && !small1, on2 && !small2, e.x, P, more);
        if (!more) break;
    }
    __syncthreads();
    if (!(TDS_DBG(c.debug) & 4)) write_out_bits<BBLOCK, NB, OutT, false>(planes, tab, K, (OutT *)c.out, img, res, X0, TWp, wpr, tid, nullptr);
}

inline int bits_index_bits(int K) { return K <= 3 ? 2 : (K <= 7 ? 3 : 4); }
inline size_t bits_lds_bytes(int K, int res, int twp, int nwaves, int out_mode) {
    size_t plane_dw = ((size_t)K * res * (twp / 32) + 3) & ~(size_t)3;
    size_t P = (size_t)1 << (2 * bits_index_bits(K));
    size_t tab_dw = 3 * P * (out_mode == TDS_OUT_F32 ? 2 : 1);
    return (plane_dw + tab_dw + BITS_FIXED_DW + (size_t)nwaves * BITS_WAVE_LDS_DW) * 4;
}
int g_bits_waves = 4;
int g_list_waves = 0;            // waves per workgroup of K3r (0: by image size; testing hook)
int g_list_lds_kb = 40;          // K3r picks the widest strip whose workgroup needs at most this much LDS (testing hook)

// The work queues of a persistent bit-plane launch: 8 counters (64 bytes) that must be zero when the kernel starts.  They live at the
// END of the caller's workspace (tds_raster_scene_workspace_bytes provides for them) and are cleared in stream order right before the
// launch.  Two launches that may overlap in time must not share a
// workspace; that already holds for the face lists.  The library allocates nothing per call.
constexpr int64_t QUEUE_BYTES = 64;
// (cleared by tds::zero_async, a kernel: see there why not hipMemsetAsync)
// splits the caller's workspace: -> queue (nullptr when there is no room for one), `bytes` is cut to what is left for the lists
inline uint32_t *workspace_queue(void *workspace, int64_t &bytes) {
    if (workspace == nullptr || bytes < QUEUE_BYTES + 64) return nullptr;
    const int64_t off = (bytes - QUEUE_BYTES) & ~(int64_t)63;
    bytes = off;
    return (uint32_t *)((char *)workspace + off);
}
// workgroups of a persistent launch: exactly as many as are resident at a time -- three per CU (168 VGPRs: three waves per SIMD), fewer when
// the LDS of one exceeds a third of the CU's 160 KiB.  (A surplus of waiting workgroups costs: DESIGN_HISTORY.md, appendix R6; fewer than the
// resident number, or head / tail items in strips, do not pay either: profiles/r06_tail_attempts.log.)
#ifdef TDS_TESTING
int g_debug = 0;
#endif
int persistent_grid(int64_t items, size_t lds_bytes, hipStream_t stream) {
    constexpr int MAX_DEVICES = 64;
    static std::atomic<int> cus_of[MAX_DEVICES];          // per device: a process may drive several (zero-initialised: not looked up yet)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = -1;
    if (dev >= 0) cus = cus_of[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        int n = 0;
        cus = (dev >= 0 && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        if (dev >= 0) cus_of[dev].store(cus, std::memory_order_relaxed);
    }
    // a stream confined to a part of the CUs (tds_stream_create): the launch is sized for those
    if (stream != nullptr) {
        uint32_t mask[16] = {0};
        if (hipExtStreamGetCUMask(stream, 16, mask) == hipSuccess) {
            int n = 0;
            for (int i = 0; i < 16; ++i) n += __builtin_popcount(mask[i]);
            if (n > 0 && n < cus) cus = n;
        } else {
            (void)hipGetLastError();
        }
    }
    const size_t granted = (lds_bytes + 2047) & ~(size_t)2047;                  // LDS is granted in 2 KiB steps
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(3, (size_t)(160 * 1024) / std::max<size_t>(granted, 1)));
    // (testing build, debug flags 65536 / 131072: 8 / 4 workgroups per CU -- the surplus of round 3)
    const int64_t cap = (int64_t)cus * ((TDS_DBG(g_debug) & 65536) ? 8 : ((TDS_DBG(g_debug) & 131072) ? 4 : per_cu));
    return (int)(items < cap ? items : cap);
}

// Generic path: arbitrary per-camera RGB mesh, every face is a candidate (no grid).
template <int TW, typename OutT>
__global__ void __launch_bounds__(RBLOCK, 4) raster_mesh_kernel(MeshArgs a, CommonArgs c) {
    TDS_RASTER_PROLOGUE()
    const float *V = a.verts + img * a.V * 3, *A = a.attrs + img * a.V * 3;
    const int32_t *Fp = a.faces + img * a.F * 3;
    for (int64_t f0 = (int64_t)wave * 64;; f0 += RBLOCK) {
        const bool more = f0 < a.F;
        int64_t f = f0 + lane;
        bool acc = false;
        uint32_t key = 0;
        int px[3] = {0, 0, 0}, py[3] = {0, 0, 0};
        if (more && f < a.F) {
            int v0 = Fp[3 * f], v1 = Fp[3 * f + 1], v2 = Fp[3 * f + 2];
            float sxv[3] = {V[3 * v0] + (-cam.cx), V[3 * v1] + (-cam.cx), V[3 * v2] + (-cam.cx)};
            float syv[3] = {V[3 * v0 + 1] + (-cam.cy), V[3 * v1 + 1] + (-cam.cy), V[3 * v2 + 1] + (-cam.cy)};
            unsigned ins;
            acc = trim_project(cam, c.scale, res, X0, TW, sxv, syv, px, py, ins, c.no_trim);
            if (acc) {
                float z = V[3 * v0 + 2];                                         // level of the first vertex, cv2.py:44-46
                int rank = 0;
                for (int l = 0; l < a.n_levels; ++l) rank += (a.levels[l] >= z) ? 1 : 0;   // 1 + index in the descending table
                rank = max(rank, 1);
                uint32_t rgb = 0;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {                                 // cv2.py:50
                    float q = floorf((A[3 * v0 + ch] * (float)(1.0 - 1e-3)) * 256.0f);
                    rgb = (rgb << 8) | ((uint32_t)(int)q & 255u);
                }
                key = ((uint32_t)rank << 24) | rgb;
            }
        }
        drain<TW>(w, acc, key, px, py, more);
        if (!more) break;
    }
    __syncthreads();
    if (!(TDS_DBG(c.debug) & 4)) write_out<TW, OutT>(tile, (OutT *)c.out, img, res, X0, tid);
}

inline size_t lds_bytes(int tw, int res) { return ((size_t)tw * res + (size_t)RWAVES * WAVE_LDS_DW) * sizeof(uint32_t); }

// strip width: the widest of {64,32,16,8} whose tile leaves room for two workgroups per CU (160 KiB LDS), else one
inline int pick_tw(int res) {
    const int cands[4] = {64, 32, 16, 8};
    for (int tw : cands)
        if (lds_bytes(tw, res) <= 80 * 1024) return tw;
    for (int tw : cands)
        if (lds_bytes(tw, res) <= 160 * 1024) return tw;
    return 0;
}

int g_force_tw = 0;

}  // namespace

#ifdef TDS_TESTING
// ---- testing build only (include/tdship.h, section "testing hooks"): absent from libtdship.so --------------------------------
// force the strip width (0 = automatic)
TDS_EXPORT int tds_raster_set_strip_width(int tw) {
    TDS_CHECK_ARG(tw == 0 || tw == 8 || tw == 16 || tw == 32 || tw == 64 || tw == 96 || tw == 128, "strip width must be 0, 8, 16, 32, 64, 96 or 128");
    g_force_tw = tw;
    return TDS_OK;
}

// waves per workgroup of the bit-plane kernel (4 or 8)
TDS_EXPORT int tds_raster_set_bits_waves(int n) {
    TDS_CHECK_ARG(n == 4 || n == 8, "waves per workgroup must be 4 or 8");
    g_bits_waves = n;
    return TDS_OK;
}

// read and reset the work counters of the bit-plane kernel (debug flag 128)
TDS_EXPORT int tds_raster_get_stats(unsigned long long *out16) {
    TDS_CHECK_ARG(out16, "tds_raster_get_stats: null output");
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stats), sizeof(zero)) != hipSuccess) { tds::set_error("tds_raster_get_stats: copy failed"); return TDS_EHIP; }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stats), zero, sizeof(zero)) != hipSuccess) { tds::set_error("tds_raster_get_stats: reset failed"); return TDS_EHIP; }
    return TDS_OK;
}

// K3r (the list rasteriser of the split bit-plane path): the LDS a workgroup may take in KiB, which sets the strip width
TDS_EXPORT int tds_raster_set_list_lds(int lds_kb) {
    TDS_CHECK_ARG(lds_kb >= 16 && lds_kb <= 150, "tds_raster_set_list_lds: 16..150 KiB");
    g_list_lds_kb = lds_kb;
    return TDS_OK;
}

// K3r: waves per workgroup (2 or 4; 0 = chosen by the size of a strip)
TDS_EXPORT int tds_raster_set_list_waves(int waves) {
    TDS_CHECK_ARG(waves == 0 || waves == 2 || waves == 4, "tds_raster_set_list_waves: 0, 2 or 4");
    g_list_waves = waves;
    return TDS_OK;
}

// ablation switches, see CommonArgs::debug
TDS_EXPORT int tds_raster_set_debug(int flags) {
    g_debug = flags;
    return TDS_OK;
}

#endif  // TDS_TESTING

#define TDS_LAUNCH_RASTER(KERNEL, ARGS)                                                                                        \
    do {                                                                                                                       \
        size_t lds = lds_bytes(tw, res);                                                                                       \
        dim3 grid((unsigned)(n_img * cm.strips));                                                                              \
        auto launch = [&](auto kern) {                                                                                         \
            if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL(kern, grid, dim3(RBLOCK), lds, (hipStream_t)stream, ARGS, cm);                                  \
        };                                                                                                                     \
        if (out_mode == TDS_OUT_F32) {                                                                                         \
            if (tw == 64) launch(KERNEL<64, float>); else if (tw == 32) launch(KERNEL<32, float>);                             \
            else if (tw == 16) launch(KERNEL<16, float>); else launch(KERNEL<8, float>);                                       \
        } else {                                                                                                               \
            if (tw == 64) launch(KERNEL<64, uint8_t>); else if (tw == 32) launch(KERNEL<32, uint8_t>);                         \
            else if (tw == 16) launch(KERNEL<16, uint8_t>); else launch(KERNEL<8, uint8_t>);                                   \
        }                                                                                                                      \
    } while (0)

static int common_checks(const char *fn, int64_t n_img, int res, int out_mode, const void *out, int &tw) {
    TDS_CHECK_ARG(n_img >= 0, "%s: negative image count", fn);
    TDS_CHECK_ARG(res > 0 && res <= 4096, "%s: resolution %d out of range (1..4096)", fn, res);
    TDS_CHECK_ARG(out_mode == TDS_OUT_F32 || out_mode == TDS_OUT_U8, "%s: unknown output mode %d", fn, out_mode);
    TDS_CHECK_ARG(out || n_img == 0, "%s: null output", fn);
    tw = (g_force_tw && g_force_tw <= 64) ? g_force_tw : pick_tw(res);
    if (tw == 0 || lds_bytes(tw, res) > 160 * 1024) { tds::set_error("%s: resolution %d does not fit the LDS tile", fn, res); return TDS_ELIMIT; }
    TDS_CHECK_ARG(n_img * ((res + tw - 1) / tw) < ((int64_t)1 << 31), "%s: too many strips", fn);
    return TDS_OK;
}

namespace {
constexpr int DEFAULT_CAPS = 512;         // faces per strip list that the recommended workspace provides
constexpr int LIST_CAPS = 2048;           // faces per camera list of the split bit-plane path that the recommended workspace provides
// The split form (K3s + K3r) serves resolutions up to these; above, the fused launch hides the scan behind its write stream or its row loops.
// ONE sweep on one build and box (round 6, tools/split_threshold_sweep.sh, B = 1024 x 64, ms per call, fused / split -- each pair from one process;
// profiles/r06_split_threshold_sweep.log):
//   float32   96 4.00 / 2.72   104 4.09 / 3.02   112 4.13 / 3.24   116 4.14 / 3.98   120 4.19 / 4.46   124 4.18 / 4.20   128 4.12 / 3.24   136 4.33 / 5.38
//            144 4.36 / 3.93   152 4.46 / 5.03   160 4.48 / 4.28   168 5.11 / 6.20   176 4.80 / 5.02   192 4.97 / 5.48
//   uint8    128 3.99 / 2.91   160 4.16 / 3.38   176 4.51 / 3.70   192 4.50 / 4.14   208 4.69 / 4.33   216 4.65 / 4.38   224 4.94 / 5.18 (7.35 once the strip no longer
//            holds the whole image)   256 5.09 / 5.64
// float32: K3r is bound by its write stream from about 116 x 116 on and pays for output columns (res x 4 bytes) that straddle 64-byte sectors, so above
// SPLIT_ANY_RES_F32 the split form is taken only where res is a multiple of 16 (split_serves).
constexpr int SPLIT_MAX_RES_F32 = 160, SPLIT_ANY_RES_F32 = 116, SPLIT_MAX_RES_U8 = 216;
inline bool split_serves(int res, bool f32) {
    if (!f32) return res <= SPLIT_MAX_RES_U8;
    return res <= SPLIT_ANY_RES_F32 || (res <= SPLIT_MAX_RES_F32 && (res & 15) == 0);
}
inline int64_t ws_bytes_for(int64_t n_img, int strips, int caps) {
    return n_img * strips * ((int64_t)caps * (int64_t)sizeof(uint4) + (int64_t)sizeof(uint32_t));
}
}  // namespace

TDS_EXPORT int tds_raster_index_slices_bytes(int64_t n_img, int res, int64_t *bytes) {
    TDS_CHECK_ARG(bytes, "tds_raster_index_slices_bytes: null output");
    TDS_CHECK_ARG(n_img >= 0 && res > 0 && res <= 4096, "tds_raster_index_slices_bytes: bad sizes");
    *bytes = (res & 3) ? 0 : n_img * (int64_t)((res + 31) / 32) * (int64_t)(res / 4) * 64;
    return TDS_OK;
}

TDS_EXPORT int tds_raster_scene_workspace_bytes(int64_t n_img, int res, int64_t *bytes) {
    TDS_CHECK_ARG(bytes, "tds_raster_scene_workspace_bytes: null output");
    TDS_CHECK_ARG(res > 0 && res <= 4096 && n_img >= 0, "tds_raster_scene_workspace_bytes: bad arguments");
    int tw = (g_force_tw && g_force_tw <= 64) ? g_force_tw : pick_tw(res);
    *bytes = 0;
    if (tw != 0 && (res + tw - 1) / tw <= MAX_STRIPS) *bytes = ws_bytes_for(n_img, (res + tw - 1) / tw, DEFAULT_CAPS);
    // the split bit-plane path: a marker list, a count and a list of LIST_CAPS faces (16 bytes each) per camera
    // (only where the split form can be chosen for either output type; the testing build can force it anywhere)
    const int64_t lists = (((n_img + 1) * 4 + 255) & ~(int64_t)255) + ((n_img * 4 + 255) & ~(int64_t)255) + n_img * LIST_CAPS * 16;
#ifndef TDS_TESTING
    if (split_serves(res, true) || split_serves(res, false))
#endif
    if (lists > *bytes) *bytes = lists;
    *bytes = ((*bytes + 63) & ~(int64_t)63) + QUEUE_BYTES + 64;          // + the work queues of a persistent launch, at the end (workspace_queue)
    return TDS_OK;
}

TDS_EXPORT int tds_raster_scene_workspace_bytes_for(int64_t n_img, int res, int out_mode, int n_keys, int64_t *bytes) {
    TDS_CHECK_ARG(bytes, "tds_raster_scene_workspace_bytes_for: null output");
    TDS_CHECK_ARG(res > 0 && res <= 4096 && n_img >= 0, "tds_raster_scene_workspace_bytes_for: bad arguments");
    TDS_CHECK_ARG(out_mode == TDS_OUT_F32 || out_mode == TDS_OUT_U8, "tds_raster_scene_workspace_bytes_for: unknown output mode %d", out_mode);
    if (n_keys < 0 || n_keys > MAX_KEYS) return tds_raster_scene_workspace_bytes(n_img, res, bytes);
    // the bit-plane kernels: face lists where the split form can run for this output type, the work queues of the persistent launch always
    int64_t lists = 0;
#ifndef TDS_TESTING
    if (split_serves(res, out_mode == TDS_OUT_F32))
#endif
        lists = (((n_img + 1) * 4 + 255) & ~(int64_t)255) + ((n_img * 4 + 255) & ~(int64_t)255) + n_img * LIST_CAPS * 16;
    *bytes = ((lists + 63) & ~(int64_t)63) + QUEUE_BYTES + 64;
    return TDS_OK;
}

namespace {
// what the scene kernels need to know about the static map(s) of a launch
struct MapSource {
    MapView one;                    // the single map, or ...
    const MapView *views;           // ... device array of per-scene maps
    const int32_t *scene_map;       // device, B indices into `views`
    const uint32_t *uniq_keys;      // host: distinct face keys (union over the maps)
    int n_uniq;                     // -1: too many for the bit-plane path
    bool renders;                   // created with rendering data
};
int raster_scene_impl(const MapSource &ms, const float *state, const float *agent_sc, const float *tmpl, const uint32_t *actor_key, const uint8_t *mask,
                      const float *cam_xy, const float *cam_sc, int64_t B, int64_t Nc, int64_t N, float scale, int res, int out_mode, void *out,
                      void *workspace, int64_t workspace_bytes, const uint32_t *actor_keys, int n_actor_keys, int actor_key_per_camera, const float *extra_tri, const uint32_t *extra_key, int64_t n_extra, tds_raster_aux_t *aux, void *stream);
}  // namespace

TDS_EXPORT int tds_raster_scene(const tds_map_t *map, const float *state, const float *agent_sc, const float *tmpl,
                                const uint32_t *actor_key, const uint8_t *mask, const float *cam_xy, const float *cam_sc, int64_t B,
                                int64_t Nc, int64_t N, float scale, int res, int out_mode, void *out, void *workspace,
                                int64_t workspace_bytes, const uint32_t *actor_keys, int n_actor_keys, int actor_key_per_camera, const float *extra_tri, const uint32_t *extra_key, int64_t n_extra, tds_raster_aux_t *aux, void *stream) {
    TDS_CHECK_ARG(map, "tds_raster_scene: null map");
    MapSource ms;
    ms.one = map->view; ms.views = nullptr; ms.scene_map = nullptr; ms.uniq_keys = map->uniq_keys; ms.n_uniq = map->n_uniq;
    ms.renders = map->n_levels > 0 || map->view.nx == 0;
    return raster_scene_impl(ms, state, agent_sc, tmpl, actor_key, mask, cam_xy, cam_sc, B, Nc, N, scale, res, out_mode, out, workspace, workspace_bytes,
                             actor_keys, n_actor_keys, actor_key_per_camera, extra_tri, extra_key, n_extra, aux, stream);
}

TDS_EXPORT int tds_raster_scene_multi(const tds_mapset_t *set, const int32_t *scene_map, const float *state, const float *agent_sc, const float *tmpl,
                                      const uint32_t *actor_key, const uint8_t *mask, const float *cam_xy, const float *cam_sc, int64_t B,
                                      int64_t Nc, int64_t N, float scale, int res, int out_mode, void *out, void *workspace,
                                      int64_t workspace_bytes, const uint32_t *actor_keys, int n_actor_keys, int actor_key_per_camera, const float *extra_tri, const uint32_t *extra_key, int64_t n_extra, tds_raster_aux_t *aux, void *stream) {
    TDS_CHECK_ARG(set && set->n > 0, "tds_raster_scene_multi: null or empty map set");
    TDS_CHECK_ARG(scene_map || B == 0, "tds_raster_scene_multi: null scene -> map index array");
    MapSource ms;
    ms.one = tds::MapView{}; ms.views = set->d_views; ms.scene_map = scene_map; ms.uniq_keys = set->uniq_keys; ms.n_uniq = set->n_uniq;
    ms.renders = set->n_levels > 0;
    return raster_scene_impl(ms, state, agent_sc, tmpl, actor_key, mask, cam_xy, cam_sc, B, Nc, N, scale, res, out_mode, out, workspace, workspace_bytes,
                             actor_keys, n_actor_keys, actor_key_per_camera, extra_tri, extra_key, n_extra, aux, stream);
}

namespace {
int raster_scene_impl(const MapSource &ms, const float *state, const float *agent_sc, const float *tmpl, const uint32_t *actor_key, const uint8_t *mask,
                      const float *cam_xy, const float *cam_sc, int64_t B, int64_t Nc, int64_t N, float scale, int res, int out_mode, void *out,
                      void *workspace, int64_t workspace_bytes, const uint32_t *actor_keys, int n_actor_keys, int actor_key_per_camera, const float *extra_tri, const uint32_t *extra_key, int64_t n_extra, tds_raster_aux_t *aux, void *stream) {
    TDS_CHECK_ARG(B >= 0 && Nc >= 0 && N >= 0 && N < (1 << 20), "tds_raster_scene: bad sizes");
    if (aux) { aux->n_keys = 0; aux->index_bits = 0; }
    TDS_CHECK_ARG(ms.renders, "tds_raster_scene: the map was created without rendering data");
    int64_t n_img = B * Nc;
    int tw = 0;
    int rc = common_checks("tds_raster_scene", n_img, res, out_mode, out, tw);
    if (rc != TDS_OK) return rc;
    if (n_img == 0) return TDS_OK;
    TDS_CHECK_ARG(cam_xy && cam_sc, "tds_raster_scene: null camera arrays");
    TDS_CHECK_ARG(N == 0 || (state && agent_sc && tmpl && actor_key && mask), "tds_raster_scene: null agent arrays");
    TDS_CHECK_ARG(scale > 0.0f, "tds_raster_scene: scale must be positive");
    TDS_CHECK_ARG(workspace_bytes >= 0 && (workspace || workspace_bytes == 0), "tds_raster_scene: bad workspace");
    TDS_CHECK_ARG(workspace == nullptr || ((uintptr_t)workspace & 15) == 0, "tds_raster_scene: the workspace must be 16-byte aligned");
    uint32_t *const ws_queue = workspace_queue(workspace, workspace_bytes);          // the tail of the workspace: work queues of a persistent launch
    SceneArgsEx a;
    a.map = ms.one; a.views = ms.views; a.scene_map = ms.scene_map; a.state = (const float4 *)state; a.agent_sc = (const float2 *)agent_sc; a.tmpl = (const float2 *)tmpl;
    a.actor_key = actor_key; a.mask = mask; a.N = (int)N; a.Nc = (int)Nc; a.key_per_cam = actor_key_per_camera ? 1 : 0;
    TDS_CHECK_ARG(n_extra >= 0 && n_extra < (1 << 20) && (n_extra == 0 || (extra_tri && extra_key)), "tds_raster_scene: bad per-camera triangle arrays");
    a.extra_tri = extra_tri; a.extra_key = extra_key; a.K = (int)n_extra;
    CommonArgs cm;
    cm.cam_xy = (const float2 *)cam_xy; cm.cam_sc = (const float2 *)cam_sc; cm.scale = scale; cm.res = res;
    cm.strips = (res + tw - 1) / tw; cm.n_img = n_img; cm.out = out; cm.slices = nullptr; cm.debug = TDS_DBG(g_debug);
    cm.no_trim = (aux && (aux->flags & TDS_RASTER_NO_TRIM)) ? 1 : 0;
    const bool want_slices = aux && aux->index_slices;
    if (want_slices) {
        int64_t need = 0;
        (void)tds_raster_index_slices_bytes(n_img, res, &need);
        TDS_CHECK_ARG(need > 0 && out_mode == TDS_OUT_F32, "tds_raster_scene: index slices need a float32 image whose resolution is a multiple of 4");
        TDS_CHECK_ARG(aux->index_slices_bytes >= need, "tds_raster_scene: index slices buffer too small (%lld < %lld bytes)",
                      (long long)aux->index_slices_bytes, (long long)need);
    }
    // fastest path: bit planes, when the scene uses at most MAX_KEYS distinct keys and the caller listed the actors' keys
    if (((N == 0 && n_extra == 0) || (actor_keys && n_actor_keys > 0)) && ms.n_uniq >= 0 && !(TDS_DBG(g_debug) & 64)) {
        KeyTable kt;
        kt.n = 0;
        bool ok = true;
        auto add = [&](uint32_t key) {
            for (int i = 0; i < kt.n; ++i) if (kt.key[i] == key) return;
            if (kt.n == MAX_KEYS) { ok = false; return; }
            kt.key[kt.n++] = key;
        };
        for (int i = 0; i < ms.n_uniq; ++i) add(ms.uniq_keys[i]);
        for (int i = 0; i < ((N > 0 || n_extra > 0) ? n_actor_keys : 0); ++i) add(actor_keys[i]);
        for (int i = kt.n; i < 16; ++i) kt.key[i] = 0xffffffffu;
        if (ok && kt.n > 0) {
            for (int i = 1; i < kt.n; ++i)                                   // ascending (insertion sort)
                for (int j = i; j > 0 && kt.key[j - 1] > kt.key[j]; --j) { uint32_t t = kt.key[j]; kt.key[j] = kt.key[j - 1]; kt.key[j - 1] = t; }
            // strip width: the whole (32-padded) image if the planes fit 64 KiB, else the widest multiple of 32 that does
            int twp = (res + 31) & ~31;
            while (twp > 32 && (size_t)kt.n * res * (twp / 8) > 64 * 1024) twp -= 32;
            // ... in strips of EQUAL width (nine keys at 256 x 256: 192 + 64 columns left the second strip's workgroups a third of the work and the first
            // one's two per CU -- 10.4 ms; 128 + 128 at three per CU: profiles/r06_more_keys.log)
            // and, where one more strip lets three workgroups share a CU instead of two, in one more (ten keys: 3 x 96 columns).
            if (twp < ((res + 31) & ~31) && !(TDS_DBG(g_debug) & 1048576)) {      // (1048576: testing, the widest strips that fit)
                int n_strips = (res + twp - 1) / twp;
                auto equal_width = [&](int n) { return (((res + n - 1) / n) + 31) & ~31; };
                twp = equal_width(n_strips);
                if (bits_lds_bytes(kt.n, res, twp, g_bits_waves, out_mode) > 52 * 1024 && equal_width(n_strips + 1) >= 64 &&
                    bits_lds_bytes(kt.n, res, equal_width(n_strips + 1), g_bits_waves, out_mode) <= 52 * 1024 && !(TDS_DBG(g_debug) & 2097152))      // (2097152: testing, not one more)
                    twp = equal_width(n_strips + 1);
            }
            int nwv = g_bits_waves;
            // Three workgroups per CU need at most 52 KiB each (160 KiB of LDS, allocated in 2 KiB steps): five keys at 256 x 256 just fit.
            // A whole image that does not (six keys and more: two agent types, traffic lights, waypoints) -- measured at B = 1024 x 64, 256 x 256,
            // six / seven keys, float32 | uint8 ms (profiles/r06_more_keys.log):
            //     two half-image strips, four 128-VGPR workgroups per CU (each strip scans the grid for itself)   7.79 / 7.86 | 6.88 / 6.98
            //     the whole image, two 4-wave workgroups per CU (persistent)                                       7.62 / 7.71 | 6.90 / 6.99
            //     the whole image, two 8-WAVE workgroups per CU (sixteen waves per CU instead of eight)            7.61 / 7.63 | 6.25 / 6.32
            // so: eight waves on the whole image where two such workgroups fit a CU (80 KiB each: up to seven keys at 256 x 256), else half strips
            // (eight keys: 7.72 against 7.75 for the whole image in 4-wave workgroups; five keys at 320 x 320: 2.76 against 2.84; and differentiable
            // calls, whose index slices the 4-wave kernel writes).
            if (twp == ((res + 31) & ~31) && twp >= 128 && bits_lds_bytes(kt.n, res, twp, nwv, out_mode) > 52 * 1024 && !(TDS_DBG(g_debug) & 262144)) {      // (262144: testing, the whole image in 4-wave workgroups, two per CU)
                const int half = ((twp / 2) + 31) & ~31;
                if (nwv == 4 && !want_slices && bits_lds_bytes(kt.n, res, twp, 8, out_mode) <= 80 * 1024 && !(TDS_DBG(g_debug) & 524288)) nwv = 8;          // (524288: testing, never eight waves)
                else if (bits_lds_bytes(kt.n, res, half, nwv, out_mode) <= 52 * 1024) twp = half;
            }
            if (g_force_tw >= 32 && g_force_tw < twp) twp = g_force_tw;     // tuning hook
            size_t lds = bits_lds_bytes(kt.n, res, twp, nwv, out_mode);
            if (lds <= 150 * 1024) {
                CommonArgs cb = cm;
                cb.strips = (res + twp - 1) / twp;
                cb.slices = want_slices ? aux->index_slices : nullptr;
                if (aux) { aux->n_keys = kt.n; aux->index_bits = bits_index_bits(kt.n); for (int i = 0; i < 16; ++i) aux->keys[i] = i < kt.n ? kt.key[i] : 0u; }
                const int nb = bits_index_bits(kt.n);
                bool four_per_cu = lds <= 40 * 1024 && !(TDS_DBG(g_debug) & 2048);      // see MINWG (2048: ablation, the 170-VGPR kernel)
                // ---- the split form (K3s + K3r, above) where the launch is not bound by the write stream: uint8 output, resolutions below 256 ----
                const uint32_t *only = nullptr;
                {
                    const bool f32 = out_mode == TDS_OUT_F32;
                    // (where the split form pays: the sweep beside SPLIT_MAX_RES_*)
                    bool split = !want_slices && nwv == 4 && workspace != nullptr && split_serves(res, f32);
                    if (TDS_DBG(g_debug) & 8192) split = false;                              // testing: the fused kernel everywhere
                    if (TDS_DBG(g_debug) & 16384) split = !want_slices && nwv == 4 && workspace != nullptr;      // testing: the split form everywhere
                    const size_t off_counts = (((size_t)n_img + 1) * 4 + 255) & ~(size_t)255;
                    const size_t off_lists = (off_counts + (size_t)n_img * 4 + 255) & ~(size_t)255;
                    // a poly record is 20 bytes: 16 in `lists` (flags, P0, P1, P2), 4 in `lists3` (P3)
                    int64_t caps = split && (size_t)workspace_bytes > off_lists ? (((int64_t)workspace_bytes - (int64_t)off_lists) / (n_img * 20)) & ~(int64_t)3 : 0;
                    if (caps > 8192) caps = 8192;
                    if (split && caps >= 128) {
                        uint32_t *poisoned = (uint32_t *)workspace, *counts = (uint32_t *)((char *)workspace + off_counts);
                        uint4 *lists = (uint4 *)((char *)workspace + off_lists);
                        uint32_t *lists3 = (uint32_t *)(lists + (size_t)n_img * (size_t)caps);
                        // strip width of K3r: the widest multiple of 32 columns whose workgroup stays within the LDS budget (four workgroups per CU)
                        // waves per workgroup of K3r: by the pixels of a strip (testing hook: g_list_waves)
                        int lw = g_list_waves;
                        int tws = (res + 31) & ~31;
                        // (two waves up to 104 x 104 float32 / 128 x 128 uint8 pixels of a strip, four above: the sweep is DESIGN_HISTORY.md, appendix R7)
                        if (lw == 0) lw = (int64_t)res * tws <= (f32 ? 104 * 104 : 128 * 128) ? 2 : 4;
                        while (tws > 32 && bits_lds_bytes(kt.n, res, tws, lw, out_mode) > (size_t)g_list_lds_kb * 1024) tws -= 32;
                        const size_t lds_s = bits_lds_bytes(kt.n, res, tws, lw, out_mode);
                        // (more keys than the sweep's five can push a large image over the LDS budget: K3r in several strips per camera loses to
                        // the fused kernel from about 160 x 160 on -- uint8 224: 7.35 against 4.94 ms -- unless the testing build forces the form)
                        const bool narrowed = tws < ((res + 31) & ~31) && res >= 160 && !(TDS_DBG(g_debug) & 16384);
                        if (lds_s <= 150 * 1024 && !narrowed) {
                            if (aux) { aux->n_keys = kt.n; aux->index_bits = nb; for (int i = 0; i < 16; ++i) aux->keys[i] = i < kt.n ? kt.key[i] : 0u; }
                            if (tds::zero_async(poisoned, 4, (hipStream_t)stream) != hipSuccess) { tds::set_error("tds_raster_scene: clearing the workspace failed"); return TDS_EHIP; }
                            CommonArgs cs = cm;
                            cs.strips = 1; cs.slices = nullptr;
                            const dim3 sgrid((unsigned)((n_img + SCAN_WAVES - 1) / SCAN_WAVES));
                            const SceneArgs sbase = a;
                            auto launch_s = [&](auto kern, const auto &args) { hipLaunchKernelGGL(kern, sgrid, dim3(SCAN_WAVES * 64), 0, (hipStream_t)stream, args, cs, kt, counts, lists, lists3, (int)caps, poisoned); };
                            if (a.K != 0) launch_s(scan_faces_kernel<SceneArgsEx>, a);
                            else launch_s(scan_faces_kernel<SceneArgs>, sbase);
                            TDS_LAUNCH_CHECK("scan_faces_kernel");
                            CommonArgs cr = cm;
                            cr.strips = (res + tws - 1) / tws; cr.slices = nullptr;
                            const dim3 rgrid((unsigned)(n_img * cr.strips));
                            auto launch_r = [&](auto kern) {
                                if (lds_s > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);
                                hipLaunchKernelGGL(kern, rgrid, dim3(lw * 64), lds_s, (hipStream_t)stream, cr, kt, tws, (const uint32_t *)counts, (const uint4 *)lists, (const uint32_t *)lists3, (int)caps);
                            };
#define TDS_LIST_DISPATCH_W(T, W) do { if (nb == 2) launch_r(raster_list_bits_kernel<2, T, W>); else if (nb == 3) launch_r(raster_list_bits_kernel<3, T, W>); else launch_r(raster_list_bits_kernel<4, T, W>); } while (0)
#define TDS_LIST_DISPATCH(T) do { if (lw == 2) TDS_LIST_DISPATCH_W(T, 2); else TDS_LIST_DISPATCH_W(T, 4); } while (0)
                            if (f32) TDS_LIST_DISPATCH(float); else TDS_LIST_DISPATCH(uint8_t);
#undef TDS_LIST_DISPATCH
#undef TDS_LIST_DISPATCH_W
                            TDS_LAUNCH_CHECK("raster_list_bits_kernel");
                            only = poisoned;            // what follows: the fused kernel, over the cameras K3s marked (normally none)
                        }
                    }
                }
                // the instantiations for three workgroups per CU are persistent launches: their workgroups take (camera, strip) items from per-XCD
                // queues (see claim_work); the others get one workgroup per item
                if (only != nullptr) four_per_cu = false;          // the launch over the marked cameras takes its items from a queue: a persistent instantiation
                // (without a workspace there is no queue: the same kernels then run one workgroup per item)
                const bool persist = nwv == 4 && (!four_per_cu || cb.slices != nullptr) && ws_queue != nullptr;
                uint32_t *queue = nullptr;
                if (persist) {
                    queue = ws_queue;
                    if (tds::zero_async(queue, (size_t)QUEUE_BYTES, (hipStream_t)stream) != hipSuccess) { tds::set_error("tds_raster_scene: clearing the work queues failed"); return TDS_EHIP; }
                }
                dim3 grid((unsigned)(persist ? persistent_grid(only != nullptr ? 512 : n_img * cb.strips, lds, (hipStream_t)stream) : n_img * cb.strips));
                const SceneArgs base = a;
                auto launch_b = [&](auto kern) {            // scenes without per-camera triangles
                    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipLaunchKernelGGL(kern, grid, dim3(nwv * 64), lds, (hipStream_t)stream, base, cb, kt, twp, queue, only);
                };
                auto launch = [&](auto kern) {
                    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    hipLaunchKernelGGL(kern, grid, dim3(nwv * 64), lds, (hipStream_t)stream, a, cb, kt, twp, queue, only);
                };
#define TDS_BITS_DISPATCH(T)                                                                                                   \
    do {                                                                                                                       \
        if (nwv == 4 && four_per_cu && a.K == 0) { if (nb == 2) launch_b(raster_scene_bits_kernel<4, 2, T, SceneArgs, false, 4>); else if (nb == 3) launch_b(raster_scene_bits_kernel<4, 3, T, SceneArgs, false, 4>); else launch_b(raster_scene_bits_kernel<4, 4, T, SceneArgs, false, 4>); } \
        else if (nwv == 4 && four_per_cu) { if (nb == 2) launch(raster_scene_bits_kernel<4, 2, T, SceneArgsEx, false, 4>); else if (nb == 3) launch(raster_scene_bits_kernel<4, 3, T, SceneArgsEx, false, 4>); else launch(raster_scene_bits_kernel<4, 4, T, SceneArgsEx, false, 4>); } \
        else if (nwv == 4 && a.K == 0) { if (nb == 2) launch_b(raster_scene_bits_kernel<4, 2, T, SceneArgs>); else if (nb == 3) launch_b(raster_scene_bits_kernel<4, 3, T, SceneArgs>); else launch_b(raster_scene_bits_kernel<4, 4, T, SceneArgs>); } \
        else if (nwv == 4) { if (nb == 2) launch(raster_scene_bits_kernel<4, 2, T, SceneArgsEx>); else if (nb == 3) launch(raster_scene_bits_kernel<4, 3, T, SceneArgsEx>); else launch(raster_scene_bits_kernel<4, 4, T, SceneArgsEx>); } \
        else { if (nb == 2) launch(raster_scene_bits_kernel<8, 2, T, SceneArgsEx>); else if (nb == 3) launch(raster_scene_bits_kernel<8, 3, T, SceneArgsEx>); else launch(raster_scene_bits_kernel<8, 4, T, SceneArgsEx>); } \
    } while (0)
                if (cb.slices != nullptr) {
                    // differentiable calls: float32, four waves; the instantiation that also stores the index slices
                    if (nwv != 4) { tds::set_error("tds_raster_scene: index slices need the 4-wave bit-plane kernel"); return TDS_ELIMIT; }
                    if (a.K == 0) { if (nb == 2) launch_b(raster_scene_bits_kernel<4, 2, float, SceneArgs, true>); else if (nb == 3) launch_b(raster_scene_bits_kernel<4, 3, float, SceneArgs, true>); else launch_b(raster_scene_bits_kernel<4, 4, float, SceneArgs, true>); }
                    else { if (nb == 2) launch(raster_scene_bits_kernel<4, 2, float, SceneArgsEx, true>); else if (nb == 3) launch(raster_scene_bits_kernel<4, 3, float, SceneArgsEx, true>); else launch(raster_scene_bits_kernel<4, 4, float, SceneArgsEx, true>); }
                } else if (out_mode == TDS_OUT_F32) TDS_BITS_DISPATCH(float); else TDS_BITS_DISPATCH(uint8_t);
#undef TDS_BITS_DISPATCH
                TDS_LAUNCH_CHECK("raster_scene_bits_kernel");
                return TDS_OK;
            }
        }
    }
    if (want_slices) {
        tds::set_error("tds_raster_scene: index slices are produced by the bit-plane kernel only (at most %d distinct keys, listed by the caller)", MAX_KEYS);
        return TDS_ELIMIT;
    }
    // general path: bin once per camera (K3a), then rasterise per strip from the lists (K3b)
    if (workspace && !(TDS_DBG(g_debug) & 32) && cm.strips <= MAX_STRIPS) {
        int64_t caps = (workspace_bytes / (n_img * cm.strips) - (int64_t)sizeof(uint32_t)) / (int64_t)sizeof(uint4);
        if (caps > 4096) caps = 4096;
        if (caps >= 64) {
            uint32_t *counts = (uint32_t *)workspace;
            size_t off = ((size_t)n_img * cm.strips * sizeof(uint32_t) + 255) & ~(size_t)255;
            if ((int64_t)(off + (size_t)n_img * cm.strips * caps * sizeof(uint4)) > workspace_bytes) --caps;
            uint4 *lists = (uint4 *)((char *)workspace + off);
            hipLaunchKernelGGL(bin_faces_kernel, dim3((unsigned)((n_img + BIN_WAVES - 1) / BIN_WAVES)), dim3(BIN_WAVES * 64), 0,
                               (hipStream_t)stream, a, cm, tw, counts, lists, (int)caps);
            TDS_LAUNCH_CHECK("bin_faces_kernel");
            size_t lds = lds_bytes(tw, res);
            dim3 grid((unsigned)(n_img * cm.strips));
            auto launch = [&](auto kern) {
                if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(kern, grid, dim3(RBLOCK), lds, (hipStream_t)stream, a, cm, (const uint32_t *)counts, (const uint4 *)lists, (int)caps);
            };
            if (out_mode == TDS_OUT_F32) {
                if (tw == 64) launch(raster_scene_list_kernel<64, float>); else if (tw == 32) launch(raster_scene_list_kernel<32, float>);
                else if (tw == 16) launch(raster_scene_list_kernel<16, float>); else launch(raster_scene_list_kernel<8, float>);
            } else {
                if (tw == 64) launch(raster_scene_list_kernel<64, uint8_t>); else if (tw == 32) launch(raster_scene_list_kernel<32, uint8_t>);
                else if (tw == 16) launch(raster_scene_list_kernel<16, uint8_t>); else launch(raster_scene_list_kernel<8, uint8_t>);
            }
            TDS_LAUNCH_CHECK("raster_scene_list_kernel");
            return TDS_OK;
        }
    }
    TDS_LAUNCH_RASTER(raster_scene_kernel, a);
    TDS_LAUNCH_CHECK("raster_scene_kernel");
    return TDS_OK;
}
}  // namespace

TDS_EXPORT int tds_raster_mesh(const float *verts, const float *attrs, const int32_t *faces, int64_t n_img, int64_t V, int64_t F,
                               const float *cam_xy, const float *cam_sc, const float *levels, int n_levels, float scale, int res,
                               int out_mode, void *out, int flags, void *stream) {
    TDS_CHECK_ARG(V >= 0 && F >= 0, "tds_raster_mesh: negative size");
    int tw = 0;
    int rc = common_checks("tds_raster_mesh", n_img, res, out_mode, out, tw);
    if (rc != TDS_OK) return rc;
    if (n_img == 0) return TDS_OK;
    TDS_CHECK_ARG(cam_xy && cam_sc, "tds_raster_mesh: null camera arrays");
    TDS_CHECK_ARG(F == 0 || (verts && attrs && faces && levels), "tds_raster_mesh: null mesh arrays");
    TDS_CHECK_ARG(scale > 0.0f, "tds_raster_mesh: scale must be positive");
    if (n_levels > 64) { tds::set_error("tds_raster_mesh: %d rendering levels (max 64 on the generic path)", n_levels); return TDS_ELIMIT; }
    MeshArgs a;
    a.verts = verts; a.attrs = attrs; a.faces = faces; a.V = V; a.F = F; a.n_levels = n_levels;
    for (int i = 0; i < 64; ++i) a.levels[i] = i < n_levels ? levels[i] : -3.0e38f;
    for (int i = 1; i < n_levels; ++i) TDS_CHECK_ARG(levels[i] < levels[i - 1], "tds_raster_mesh: levels must be strictly descending");
    CommonArgs cm;
    cm.cam_xy = (const float2 *)cam_xy; cm.cam_sc = (const float2 *)cam_sc; cm.scale = scale; cm.res = res;
    cm.strips = (res + tw - 1) / tw; cm.n_img = n_img; cm.out = out; cm.slices = nullptr; cm.debug = TDS_DBG(g_debug);
    cm.no_trim = (flags & TDS_RASTER_NO_TRIM) ? 1 : 0;
    TDS_LAUNCH_RASTER(raster_mesh_kernel, a);
    TDS_LAUNCH_CHECK("raster_mesh_kernel");
    return TDS_OK;
}
