// Shared host/device helpers of libtdship (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/tdship.h"

#define TDS_EXPORT extern "C" __attribute__((visibility("default")))

namespace tds {

void set_error(const char *fmt, ...);

#define TDS_CHECK_ARG(cond, ...)                       \
    do {                                               \
        if (!(cond)) {                                 \
            tds::set_error(__VA_ARGS__);               \
            return TDS_EINVAL;                         \
        }                                              \
    } while (0)

#define TDS_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            tds::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return TDS_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

// kernel launch check (no sync): picks up invalid configuration / missing device code immediately
#define TDS_LAUNCH_CHECK(name)                                                                 \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) {                                                                \
            tds::set_error("launch of %s failed: %s", name, hipGetErrorString(e_));            \
            return TDS_EHIP;                                                                   \
        }                                                                                      \
    } while (0)

constexpr int WAVE = 64;

// Zeroing device words in stream order with a KERNEL, never with hipMemsetAsync: on ROCm 7.0 a memset node of a captured graph is not
// ordered before the kernel node that follows it (a replayed raster launch found its work queues uncleared: tests/test_gpu_graph.py),
// kernel nodes are.  `words` 4-byte words at a 4-byte aligned address.
static __global__ void zero_words_kernel(uint32_t *p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
inline hipError_t zero_async(void *p, size_t bytes, hipStream_t stream) {
    const size_t words = bytes / 4;
    if (words == 0) return hipSuccess;
    const unsigned blocks = (unsigned)(words < 256 ? 1 : (words / 256 < 2048 ? words / 256 : 2048));
    hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(words < 256 ? 64 : 256), 0, stream, (uint32_t *)p, words);
    return hipGetLastError();
}

// ---- static map handle (device side view is MapView) -------------------------------------------------------
struct GridEntry {          // 32 bytes, one per (cell, face whose bounding box touches the cell)
    float x0, y0, x1, y1, x2, y2;   // world coordinates of the three vertices
    uint32_t key;                   // rank << 24 | 0x00RRGGBB  (0 when the map carries no rendering data)
    uint32_t own;                   // owner rule of the rasteriser's grid scan (raster.hip: scan_step) + outline flags:
                                    //   bits 0..12 / 13..25: first / last grid column of the face's bounding box;
                                    //   bit 26: this cell is not the first column of the box; bit 27: not its first row;
                                    //   bits 29..31: outline edge l (0: v2-v0, 1: v0-v1, 2: v1-v2) is also an edge of an earlier
                                    //   face with the same key (identical end points)
};

// A face of the RENDERING grid (maps with rendering data; raster.hip, bit-plane kernels): one triangle, or TWO triangles of the same key that
// share an edge -- road and lane-marking meshes are triangulated quads (Town01: 95 - 100 % of the faces of a view come in such pairs).  The
// pair is fetched, projected and trimmed once (four vertices instead of six) and the rows between its vertex rows are painted once
// (raster.hip: process_batch_bits; tests/fill_quads_model.c proves the row rule).  Every triangle is still drawn with the reference's
// per-face semantics (rendering/cv2.py:44-59: one cv2.fillConvexPoly per face): the vertex ORDER of both triangles is kept.
struct QuadEntry {          // 48 bytes, one per (cell, face or pair whose bounding box touches the cell)
    float x0, y0, x1, y1, x2, y2;   // T1 = (v0, v1, v2), world coordinates, in the face's own vertex order
    float x3, y3;                   // the vertex of T2 that T1 does not have (unused for a lone triangle)
    uint32_t key;                   // rank << 24 | 0x00RRGGBB
    uint32_t own;                   // as GridEntry::own, bits 0..27, for the bounding box of the whole entry
    uint32_t flags;                 // bits 0..5: T2 = (v[b0], v[b1], v[b2]), two bits each, in T2's own vertex order; bits 6..7: the vertex of T1
                                    // that T2 does not have; bit 8: there is a T2; bits 9..11 / 12..14: outline edge l of T1 / T2 repeats an edge
                                    // of an earlier face of the same key (GridEntry::own bits 29..31)
    uint32_t pad;
};

struct MapView {
    const GridEntry *entries;
    const int32_t *cell_start;      // nx*ny + 1
    float ox, oy, inv_cell, cell;
    int nx, ny;
    int64_t n_faces;
    const QuadEntry *qentries;      // the rendering grid with paired faces (null without rendering data): same cells, its own ranges
    const int32_t *qcell_start;     // nx*ny + 1
};

// Nearest-face candidate lists of K2b (maps created without rendering data): for every grid cell the faces that can be the nearest one
// of SOME point of the cell, sorted by a lower bound of their squared distance to the cell.  The off-road query of a point inside the
// grid is one linear walk of its cell's list instead of a walk over grid rings.
struct NearCand {
    int32_t face;               // index into NearView::faces
    float lb;                   // lower bound (rounded down, with a safety margin) of the squared distance from any point of the cell
};
// ... and a bounding-volume hierarchy over the same faces for the points the lists do not serve (agents that strayed more than the margin
// beyond the mesh): an ordered nearest-neighbour descent, pruned by the same conservative box bounds as every other walk of K2b, is the
// exact minimum over all faces in a dozen steps wherever the point lies (the walk over grid rings took 2.5 ms once random steps had
// scattered the agents and 124 ms with every agent a thousand kilometres away).
struct BvhNode {                // 160 bytes, eight children: a lane of the off-road kernels' groups of eight weighs one child each
    float4 box[8];              // (x0, y0, x1, y1) of child k; an empty slot holds (+inf, +inf, -inf, -inf): infinitely far from every point
    int32_t child[8];           // >= 0: an inner node; < 0: a leaf, -1 - (first << 4 | count): `count` (<= 8) faces from bvh_idx[first]
};
// entries of the per-point stack of the ordered descent (map.hip: nearest_face_d2_bvh, backward.hip: nearest_face_d2_grad_bvh).  A visit of an
// inner node pops one entry and pushes at most 8, so a hierarchy of `depth` inner levels needs at most 7 * depth + 1 entries; the builder
// measures its depth and tds_map_create attaches the hierarchy only if that fits (else the query falls back to the walk over grid rings).
constexpr int BVH_STACK = 64;
inline bool bvh_fits_stack(int depth) { return 7 * depth + 8 <= BVH_STACK; }
struct NearView {
    const NearCand *cand;       // null: no lists (maps with rendering data, empty maps)
    const int32_t *cand_start;  // nx*ny + 1
    const GridEntry *faces;     // one entry per face (key / own unused)
    float ox, oy;               // the lists have their own grid: the map's cell size, grown by a margin around the mesh so that agents
    int nx, ny;                 // that left the map's bounding box are still served
    const BvhNode *bvh;         // node 0 = root; null: none
    const int32_t *bvh_idx;     // face indices (into `faces`) in leaf order
};

// the ONE definition of "which cell does this coordinate fall in" -- used by the host builder and by the kernels
// (IEEE binary32, one rounding per operation on both sides: the library is built with -ffp-contract=off)
__host__ __device__ inline int cell_coord(float v, float origin, float inv_cell) {
    return (int)floorf((v - origin) * inv_cell);
}

}  // namespace tds

// several maps of one device for launches whose scenes have different maps (tds_mapset_create)
struct tds_mapset {
    tds::MapView *d_views;      // device array [n]
    tds::NearView *d_near;      // device array [n] (entries with null lists fall back to the ring walk)
    int n, device, n_levels;
    uint32_t uniq_keys[64];     // union of the maps' distinct face keys
    int n_uniq;                 // -1: more than 64
};

struct tds_map {
    tds::MapView view;
    void *d_entries;
    void *d_cell_start;
    void *d_qentries;           // the rendering grid with paired faces (QuadEntry), null without rendering data
    void *d_qcell_start;
    int64_t n_qentries, n_pairs;
    tds::NearView near;         // K2b candidate lists (device pointers; all null when absent)
    int64_t n_cand;
    int device;
    int64_t V, F, n_entries, bytes;
    int n_levels;
    uint32_t uniq_keys[64];     // distinct face keys of the map (bit-plane fast path of K3)
    int n_uniq;                 // -1: more than 64
};
