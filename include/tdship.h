/*
 * tdship.h -- C ABI of libtdship.so, the MI355X (gfx950) implementation of the torchdrivesim hot path.
 *
 * The reference (inverted-ai/torchdrivesim v0.2.3) is pure Python on torch and has no FFI of its own; each
 * entry point below names the reference function whose inner loop it replaces (file:line into the reference
 * checkout).  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory owned by the caller (torch tensors, `.data_ptr()`), except
 *     in tds_map_create / tds_grid_*(), which take HOST arrays (map preparation happens once per map);
 *   - tensors are dense, row-major, fp32 / int32 / uint8 as stated; shapes are given in comments;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); kernels are enqueued on it and the call
 *     returns without synchronising; calls are re-entrant per device;
 *   - return value: 0 on success, a negative TDS_E* code otherwise (never throws); tds_last_error() returns the
 *     message for the calling thread.  The Python host raises RuntimeError on any non-zero code, so
 *     BirdviewRenderer.render_frame's `except RuntimeError` (rendering/base.py:190-201) still applies;
 *   - sin/cos of agent and camera headings are INPUTS ([sin, cos] pairs), computed by the caller with torch on the
 *     same device, exactly where the reference calls torch.sin/torch.cos (simulator.py:940, utils.py:40-53,
 *     _iou_utils.py:290-291); all remaining arithmetic is IEEE-754 binary32 with one rounding per operation in the
 *     reference's order (no FMA contraction) so integer / boolean outputs can be compared bit for bit.
 */
#ifndef TDSHIP_H
#define TDSHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDS_ABI_VERSION 1

#define TDS_OK 0
#define TDS_EINVAL (-1)     /* bad argument (null pointer, negative size, unsupported resolution ...) */
#define TDS_EHIP (-2)       /* a HIP runtime call or kernel launch failed */
#define TDS_ENOMEM (-3)
#define TDS_ELIMIT (-4)     /* a documented capacity was exceeded (e.g. more than 255 rendering levels) */

int tds_version(void);
/* copies the calling thread's last error message (NUL terminated) into buf; returns its length */
int tds_last_error(char *buf, size_t n);

/* ------------------------------------------------------------------------------------------------------------
 * K1  kinematic models                                                                   kinematic.py:328-523
 * state / out: n x 4 [x, y, psi, v]  (n = B*A agents);  `out` must not alias `state` (the reference replaces the
 * state tensor, it never mutates it: kinematic.py:477, pack_state = torch.stack).
 * ---------------------------------------------------------------------------------------------------------- */

/* KinematicBicycle.step (kinematic.py:462-477) and BicycleNoReversing.step (:513-523, no_reversing != 0).
 * action: n x 2 normalised [acceleration, steering];  lr: n. */
int tds_bicycle_step_f32(const float *state, const float *action, const float *lr, float *out, int64_t n,
                         float dt, float max_acc, float max_steer, int left_handed, int no_reversing, void *stream);
/* backward of the above: grad_out n x 4 -> grad_state n x 4, grad_action n x 2, grad_lr n (any may be NULL) */
int tds_bicycle_step_bwd_f32(const float *state, const float *action, const float *lr, const float *grad_out,
                             float *grad_state, float *grad_action, float *grad_lr, int64_t n, float dt, float max_acc,
                             float max_steer, int left_handed, int no_reversing, void *stream);

/* SimpleKinematicModel.step (:362-367) / OrientedKinematicModel.step (:384-389, oriented != 0).
 * action: n x 4 normalised, norm: 4 host floats [max_dx, max_dx, max_dpsi, max_dv]. */
int tds_simple_step_f32(const float *state, const float *action, float *out, int64_t n, float dt, const float *norm,
                        int oriented, void *stream);
int tds_simple_step_bwd_f32(const float *state, const float *action, const float *grad_out, float *grad_state,
                            float *grad_action, int64_t n, float dt, const float *norm, int oriented, void *stream);

/* UnicycleModel.step -- named by the north star, absent from the reference: action n x 2 normalised
 * [acceleration, yaw rate]: v += a dt; x += v cos(psi) dt; y += v sin(psi) dt; psi += w dt. */
int tds_unicycle_step_f32(const float *state, const float *action, float *out, int64_t n, float dt, float max_acc,
                          float max_yaw_rate, void *stream);
int tds_unicycle_step_bwd_f32(const float *state, const float *action, const float *grad_out, float *grad_state,
                              float *grad_action, int64_t n, float dt, float max_acc, float max_yaw_rate, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * K2a  collisions                     simulator.py:1064-1109,1161-1194; _iou_utils.py:42-367; infractions.py:378-545
 * ---------------------------------------------------------------------------------------------------------- */
#define TDS_METRIC_IOU 0
#define TDS_METRIC_DISCS 1

/* Simulator.compute_collision for all exposed agents of all scenes at once.
 *   boxes   B x N x 5  [x, y, length, width, psi]   all agents = A exposed agents followed by N-A NPCs
 *   sc      B x N x 2  [sin, cos] of the angle the metric uses: psi (iou) or psi + pi/2*(width > length) (discs)
 *   present B x N      uint8 (already restricted to the requested agent types, simulator.py:1086-1089)
 *   out     B x A      collision_i = sum_j o_ij present_j - max_j o_ij present_j   
 *   overlap B x A      uint64 bit j set iff o_ij present_j > 0, j != i   (NULL to skip; requires N <= 64)
 *   partner B x A      int32 arg-max_j!=i of o_ij present_j, -1 if none   (NULL to skip)   [new outputs, SURVEY R8] */
int tds_collision_f32(const float *boxes, const float *sc, const uint8_t *present, float *out, uint64_t *overlap,
                      int32_t *partner, int64_t B, int64_t A, int64_t N, int metric, void *stream);
/* backward: grad_out B x A -> grad_boxes B x N x 5 (psi column = 0), grad_sc B x N x 2; both are OVERWRITTEN */
int tds_collision_bwd_f32(const float *boxes, const float *sc, const uint8_t *present, const float *grad_out,
                          float *grad_boxes, float *grad_sc, int64_t B, int64_t A, int64_t N, int metric, void *stream);

/* The `nograd` metric (simulator.py:1111-1149 -> infractions.py:352-375, 429-500; the reference asks shapely for
 * `intersection(...).area != 0` pair by pair on the host): out B x A float64 = number of OTHER present agents whose rectangle shares
 * area with agent i's, 0 for an absent agent.  boxes B x A x 5, sc B x A x 2 = [sin psi, cos psi], present B x A uint8.  Corners as
 * infractions.rectangle_vertices in float32; the predicate (no separating edge line, touching does not count) in float64. */
int tds_overlap_count_f32(const float *boxes, const float *sc, const uint8_t *present, double *out, int64_t B, int64_t A, void *stream);

/* iou_differentiable (infractions.py:307-324) / collision_detection_with_discs (:503-545), element-wise over n pairs.
 * box1, box2: n x 5; sc1, sc2: n x 2 as above. */
int tds_pairwise_overlap_f32(const float *box1, const float *sc1, const float *box2, const float *sc2, float *out,
                             int64_t n, int metric, void *stream);

/* collision_detection_with_discs(box1, box2, num_discs) for a number of discs other than the default 5 (infractions.py:378-426,
 * 503-545): odd, 3 .. 25 (torch.cdist changes its formulation above 25 points) */
int tds_pairwise_discs_f32(const float *box1, const float *sc1, const float *box2, const float *sc2, float *out, int64_t n,
                           int num_discs, void *stream);

/* box2corners_th (_iou_utils.py:270-299): box n x 5, sc n x 2 -> corners n x 4 x 2 */
int tds_box2corners_f32(const float *box, const float *sc, float *corners, int64_t n, void *stream);

/* StandardSensingObservationNoise.get_noisy_present_mask (observation_noise.py:89-132, utils.line_circle_intersection :139-187):
 *   state B x E x 4 (exposed agents first, then NPCs), size B x E x 2, present B x E uint8
 *   out   B x A x E uint8: present[b,e] and no other entity o (o != e, o != a) whose disc of radius width/2 touches the segment
 *         from ego a to entity e */
int tds_occlusion_mask_f32(const float *state, const float *size, const uint8_t *present, uint8_t *out, int64_t B, int64_t A, int64_t E,
                           void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Static map handle: triangle mesh + uniform grid, built once per map and per device.
 * Replaces the per-call `mesh.expand(...)` / `RGBMesh.concat([background.expand(Nc), ...])` dataflow of
 * infractions.py:220-226 and mesh.py:1147-1156.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct tds_map tds_map_t;

/* HOST inputs: verts V x 2, faces F x 3 (indices into verts; padded faces [0,0,0] allowed, mesh.py:69),
 * face_z F (rendering level of the face's FIRST vertex, cv2.py:44-46), face_rgb F (colour of the first vertex already
 * quantised as cv2.py:50: 0x00RRGGBB), levels: n_levels distinct rendering levels sorted DESCENDING that contain every
 * face_z and every actor level that will be rendered with this map (<= 255).  face_z / face_rgb / levels may be NULL
 * for a map that is only used by tds_offroad_f32; such a map also gets per-cell nearest-face candidate lists (exact),
 * which make the off-road query one short linear walk, and a bounding-volume hierarchy over its faces for points beyond the lists' grid.  cell_size <= 0 selects the default (8 m).
 * The handle lives on the CURRENT HIP device. */
int tds_map_create(const float *verts, const int32_t *faces, const float *face_z, const uint32_t *face_rgb, int64_t V,
                   int64_t F, const float *levels, int n_levels, float cell_size, tds_map_t **out);
int tds_map_destroy(tds_map_t *map);
/* info[0..7] = V, F, grid nx, grid ny, number of grid entries, device bytes held, n_levels, number of nearest-face candidates
 * (EIGHT words, as in every release of this header but round 5's, which wrote ten through this symbol: see INTEGRATION.md, "ABI notes") */
int tds_map_info(const tds_map_t *map, int64_t *info);
/* the first `n_words` of: the eight words above, then [8] entries of the rendering grid (lone faces + pairs, tds_common.h: QuadEntry),
 * [9] pairs of same-key faces that share an edge; words beyond TDS_MAP_INFO_WORDS read 0 */
#define TDS_MAP_INFO_WORDS 10
int tds_map_info_ex(const tds_map_t *map, int64_t *info, int n_words);
/* the distinct face keys of the map (HOST array of `cap` entries; *n receives their number, -1 when there are more than 64): with the caller's
 * actor keys they decide which rasteriser serves a launch -- at most 15 keys in all: the bit-plane kernels -- and with it how much scratch
 * tds_raster_scene can use (tds_raster_scene_workspace_bytes_for) */
int tds_map_keys(const tds_map_t *map, uint32_t *keys, int cap, int *n);

/* Map sets: batches whose scenes have DIFFERENT meshes (the reference supports them as a collated, padded mesh batch, mesh.py:69,
 * 113-123).  A set is a device array of the views of several maps of ONE device that were created with the SAME `levels` table; the
 * maps must outlive the set.  `scene_map` (device, B int32) of the *_multi entry points says which map of the set scene b uses. */
typedef struct tds_mapset tds_mapset_t;
int tds_mapset_create(const tds_map_t *const *maps, int n, tds_mapset_t **out);
int tds_mapset_destroy(tds_mapset_t *set);
int tds_mapset_keys(const tds_mapset_t *set, uint32_t *keys, int cap, int *n);      /* tds_map_keys over the union of the set's maps */

/* Which scenes of a collated batch share a mesh.  The reference pads every element of a collated mesh batch to the largest one
 * (mesh.py:172-200 `pad`, :113-123 / :232-245 `collate`), so two scenes on the same map hold identical rows, byte for byte; it then carries
 * B private copies through every batch operation (simulator.py:444-511).  These two let the host group the B rows of a DEVICE tensor by content
 * without a copy to the host, so that it builds one tds_map_t per DISTINCT mesh:
 *   tds_rows_hash_u64   out[r] = 64-bit content hash of row r (rows of `row_bytes` bytes -- a multiple of 4 -- `row_stride_bytes` apart);
 *                       `seed` selects the hash function (two seeds: 128 bits);
 *   tds_rows_equal_u8   equal[r] = 0 where row r differs from row rep[r] (the caller presets `equal` to 1): the exact confirmation of a
 *                       grouping by hash.
 * rows, out, rep, equal: device pointers. */
int tds_rows_hash_u64(const void *rows, int64_t n_rows, int64_t row_bytes, int64_t row_stride_bytes, uint64_t seed, uint64_t *out, void *stream);
int tds_rows_equal_u8(const void *rows, int64_t n_rows, int64_t row_bytes, int64_t row_stride_bytes, const int32_t *rep, uint8_t *equal, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * K2b  offroad              simulator.py:1035-1044; infractions.py:86-229 (pure-torch path, squared distances)
 *   state B x A x 4, lenwid B x A x 2, sc B x A x 2 ([sin,cos] of psi), present B x A uint8 or NULL,
 *   out B x A = sum over the 4 corners of threshold(min over faces of squared distance) [* present]
 * ---------------------------------------------------------------------------------------------------------- */
int tds_offroad_f32(const tds_map_t *map, const float *state, const float *lenwid, const float *sc,
                    const uint8_t *present, float *out, int64_t n_agents, float threshold, void *stream);
/* backward: grad_out n -> grad_state n x 4 (x, y columns; psi, v = 0), grad_lenwid n x 2, grad_sc n x 2 (overwritten) */
int tds_offroad_bwd_f32(const tds_map_t *map, const float *state, const float *lenwid, const float *sc,
                        const uint8_t *present, const float *grad_out, float *grad_state, float *grad_lenwid,
                        float *grad_sc, int64_t n_agents, float threshold, void *stream);

/* the same for scenes with different maps: agent a belongs to scene a / agents_per_scene, which uses map scene_map[scene] of the set */
int tds_offroad_multi_f32(const tds_mapset_t *set, const int32_t *scene_map, int64_t agents_per_scene, const float *state, const float *lenwid,
                          const float *sc, const uint8_t *present, float *out, int64_t n_agents, float threshold, void *stream);
int tds_offroad_multi_bwd_f32(const tds_mapset_t *set, const int32_t *scene_map, int64_t agents_per_scene, const float *state,
                              const float *lenwid, const float *sc, const uint8_t *present, const float *grad_out, float *grad_state,
                              float *grad_lenwid, float *grad_sc, int64_t n_agents, float threshold, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * K3  bird's-eye-view rasteriser, CV2 semantics
 *     simulator.py:920-1033 -> mesh.py:1053-1157 -> rendering/base.py:167-204 -> rendering/cv2.py:27-70
 * ---------------------------------------------------------------------------------------------------------- */
#define TDS_OUT_F32 0      /* reference-faithful float32 image with values 0..255 */
#define TDS_OUT_U8 1       /* same values as uint8 (separate mode, 4x fewer bytes) */

/* Fused scene path used by Simulator.render / render_egocentric: the static map comes from the handle, the actor
 * mesh is generated on the fly from agent state (never materialised per camera).
 *   state     B x N x 4   all agents (exposed + NPCs)
 *   agent_sc  B x N x 2   [sin, cos] of psi
 *   tmpl      B x N x 7 x 2   actor template verts in the agent frame (mesh.py:911-996), built once by the host
 *   actor_key B x N x 2   uint32 (rank << 24 | 0x00RRGGBB) for (body, direction) faces, rank = 1 + index of the
 *                         part's rendering level in the map's `levels`; a key of 0 removes the part (quads without a direction
 *                         triangle, e.g. the stop lines of traffic controls, mesh.py:1007-1035)
 *   mask      B x Nc x N  uint8, present & rendering mask (simulator.py:946-948)
 *   cam_xy    B x Nc x 2, cam_sc B x Nc x 2 [sin, cos]
 *   scale = 2 / fov (rendering/base.py:149), res = H = W
 *   out       B x Nc x 3 x H x W   float32 or uint8 (out_mode)
 * Equal-level faces of different colour are ordered by packed colour (documented tie-break).
 *
 * workspace: optional DEVICE scratch (tds_raster_scene_workspace_bytes) for the two-kernel forms -- a first kernel scans the map grid
 * once per camera and lists the surviving faces, a second one rasterises from the lists: per-strip lists for the packed-key kernels
 * (more than 15 keys), one list of up to 1 636 faces or pairs of faces per camera (20 bytes each) for the bit-plane kernels up to 160 x 160 (float32; above 116 x 116 only multiples of 16) / 216 x 216 (uint8),
 * where that form is faster than the fused kernel (above, the fused kernel runs and the recommended size holds no lists).  A list that overflows only sends its camera (or strip) to the kernel that scans for
 * itself: the pixels never depend on the size of the scratch.  With workspace == NULL every launch scans the grid itself (same pixels,
 * slower at low resolutions).  The scratch carries no state between calls.
 * Launch shape of the bit-plane kernels: the fused kernel at three workgroups per CU is a PERSISTENT launch: its workgroups take cameras
 * from per-XCD work queues, 64 bytes at the END of the workspace that the call clears in stream order right before the launch (a one-wave
 * kernel).  The library owns no device memory besides the handles of tds_map_create & co. and allocates nothing in
 * any per-call entry point, so every call can be captured into a HIP graph.  Two calls that may run at the same time (different
 * streams) need a workspace each.  With workspace == NULL the same kernel runs one workgroup per camera (about 1 % slower at 256 x 256).
 * The workspace must be 16-byte aligned.
 *
 * actor_keys: optional HOST array of the distinct values occurring in `actor_key` (one per agent type and part).  When it is
 * given and the scene (map + actors) uses at most 16 distinct keys, the bit-plane kernel is used: one bit per pixel and key in
 * LDS, a whole camera per workgroup, spans painted with one ds_or.  Same pixels as the other paths.
 *
 * actor_key_per_camera != 0: `actor_key` is B x Nc x N x 2 -- every camera sees its own colours, the fused form of generate()'s
 * custom_agent_colors (mesh.py:1092-1099; only the body faces take the custom colour there).
 *
 * extra_tri / extra_key / n_extra: optional per-camera triangles, B x Nc x n_extra x 3 x 2 float32 WORLD coordinates and
 * B x Nc x n_extra uint32 keys (0 = no triangle) -- the fused form of generate()'s waypoint discs (mesh.py:1120-1145), which differ
 * from camera to camera.  Their keys must be listed in `actor_keys` for the bit-plane kernel to be used. */
/* aux (may be NULL): optional outputs for a later backward pass (tds_raster_scene_bwd_idx_f32).
 *   index_slices  DEVICE buffer of tds_raster_index_slices_bytes(B * Nc, res) bytes, or NULL.  When given, the call must be servable by
 *                 the bit-plane kernel (float32 output, res a multiple of 4, at most 15 distinct keys with `actor_keys` listed) -- else
 *                 TDS_ELIMIT -- and receives, per pixel, the 1-based position of the winning key in `keys` (0 = background) as bit-slices:
 *                 uint32 [camera][x / 32][y / 4][slice 0..3][y % 4], bit x % 32; 64 B per (word column, row quad), slices >= index_bits
 *                 are zero.  x, y = OpenCV pixel coordinates = the last two axes of `out`.
 *   keys, n_keys, index_bits   filled by the call (HOST): the ascending key table of the launch (n_keys = 0 when another kernel ran)
 *   flags         input: TDS_RASTER_NO_TRIM draws every face as the reference does with trim_mesh_before_rendering = False */
#define TDS_RASTER_NO_TRIM 1   /* trim_mesh_before_rendering = False (rendering/cv2.py:15,32-41): keep faces without a vertex in view */
typedef struct tds_raster_aux {
    uint32_t *index_slices;
    int64_t index_slices_bytes;
    uint32_t keys[16];
    int32_t n_keys;
    int32_t index_bits;
    int32_t flags;              /* in: TDS_RASTER_* */
} tds_raster_aux_t;
int tds_raster_index_slices_bytes(int64_t n_img, int res, int64_t *bytes);

int tds_raster_scene(const tds_map_t *map, const float *state, const float *agent_sc, const float *tmpl,
                     const uint32_t *actor_key, const uint8_t *mask, const float *cam_xy, const float *cam_sc,
                     int64_t B, int64_t Nc, int64_t N, float scale, int res, int out_mode, void *out, void *workspace,
                     int64_t workspace_bytes, const uint32_t *actor_keys, int n_actor_keys, int actor_key_per_camera,
                     const float *extra_tri, const uint32_t *extra_key, int64_t n_extra, tds_raster_aux_t *aux, void *stream);
/* tds_raster_scene for scenes with different maps: scene b is drawn over map scene_map[b] of the set (one launch for the batch) */
int tds_raster_scene_multi(const tds_mapset_t *set, const int32_t *scene_map, const float *state, const float *agent_sc, const float *tmpl,
                           const uint32_t *actor_key, const uint8_t *mask, const float *cam_xy, const float *cam_sc, int64_t B, int64_t Nc,
                           int64_t N, float scale, int res, int out_mode, void *out, void *workspace, int64_t workspace_bytes,
                           const uint32_t *actor_keys, int n_actor_keys, int actor_key_per_camera,
                           const float *extra_tri, const uint32_t *extra_key, int64_t n_extra, tds_raster_aux_t *aux, void *stream);
/* recommended scratch size for n_img = B * Nc cameras at this resolution (0 if the fast path cannot be used): enough for every path */
int tds_raster_scene_workspace_bytes(int64_t n_img, int res, int64_t *bytes);
/* the same for a launch of which the caller knows the number of distinct keys (map keys + actor keys, tds_map_keys): with at most 15 the
 * bit-plane kernels run, which use the face lists up to 160 x 160 (float32; above 116 x 116 only multiples of 16) / 216 x 216 (uint8) only and above nothing but the 64 bytes of
 * work queues -- 128 bytes instead of 32 KB per camera at 256 x 256.  n_keys < 0 or > 15: as tds_raster_scene_workspace_bytes. */
int tds_raster_scene_workspace_bytes_for(int64_t n_img, int res, int out_mode, int n_keys, int64_t *bytes);

/* ------------------------------------------------------------------------------------------------------------
 * Output buffers with spread-out physical pages (no reference counterpart: rendering/cv2.py:52 allocates a numpy image per call).
 *
 * The rasteriser is bound by the HBM write stream, and what a write stream reaches on MI355X depends on the PHYSICAL pages under the
 * buffer: a large hipMalloc is served at 1, 15/16 or 7/8 of the rate for as long as it lives, whatever its virtual address and whatever
 * the store pattern (about one 51.5 GB allocation in three is at 7/8); a buffer whose physical pages are spread over twice its size
 * hardly ever is (one of 100 probed, at 15/16; DESIGN_HISTORY.md section 4, tools/alloc_probe.hip).  tds_buffer_create builds such a buffer: chunks of 8 MiB created alternately
 * with spacer chunks that are released once the buffer is mapped (hipMemCreate / hipMemMap; needs twice the size free while it runs,
 * and falls back to dense chunks when that is not there).  Below 256 MiB, or with TDS_BUFFER_DENSE, it is one hipMalloc.
 * These are explicit create / destroy calls like tds_map_create: no per-step entry point allocates.
 * tds_torch_alloc / tds_torch_free have the signatures torch.cuda.memory.CUDAPluggableAllocator binds
 * (void *(size_t, int device, stream), void(void *, size_t, int device, stream)): a torch memory pool over them hands such buffers to
 * `torch.empty`, cached and stream-ordered by torch's allocator like any other block (torchdrivesim_amd/rendering/hip.py: image_pool).
 * ---------------------------------------------------------------------------------------------------------- */
#define TDS_BUFFER_DENSE 1      /* plain hipMalloc whatever the size */
typedef struct tds_buffer tds_buffer_t;
int tds_buffer_create(int64_t bytes, int device, int flags, tds_buffer_t **out);
void *tds_buffer_ptr(const tds_buffer_t *buf);                         /* device pointer, valid until tds_buffer_destroy */
int tds_buffer_info(const tds_buffer_t *buf, int64_t *bytes, int64_t *chunks, int *spread);      /* any output may be NULL */
int tds_buffer_destroy(tds_buffer_t *buf);
void *tds_torch_alloc(size_t size, int device, void *stream);          /* NULL on failure */
void tds_torch_free(void *ptr, size_t size, int device, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Streams confined to a part of the device's CUs (no reference counterpart).
 *
 * The persistent raster launch holds every CU it may use until its last image is out, so nothing runs beside it -- unless it is kept off
 * a few CUs.  tds_stream_create makes a stream whose kernels only run on the CUs of `cu_mask` (hipExtStreamCreateWithCUMask: bit i of the
 * mask is CU i / 8 of XCD i % 8 on MI355X; `n_words` 32-bit words, 8 for 256 CUs; NULL = all CUs).  A loop renders on a stream that leaves
 * four CUs per XCD out and computes its metrics on a stream confined to those 32: the write-bound launch loses nothing (tools/cu_mask_probe.hip)
 * and the metrics run beside it (Simulator.overlap_infractions = 'reserved').  tds_raster_scene sizes its persistent launch for the CUs of the
 * stream it is given.  Explicit create / destroy, like the other handles.
 * ---------------------------------------------------------------------------------------------------------- */
int tds_stream_create(int device, const uint32_t *cu_mask, int n_words, void **stream);
int tds_stream_destroy(int device, void *stream);
int tds_device_cu_count(int device, int *cus);
/* Where do the kernels of `stream` run?  Enqueues `n` one-wave workgroups that each keep their CU busy for a moment and write
 * XCC_ID << 16 | (HW_ID bits 15..8: SE, SH, CU) of the CU they ran on to places[blockIdx] (device memory, n words).  With n of a few
 * thousand every CU the stream may use shows up.  The caller synchronises and reads; the mask layout documented above is an observation of
 * one MI355X in SPX mode, and this is how the host verifies it on the device at hand before it relies on it (_ops.reserved_streams). */
int tds_stream_places(void *stream, uint32_t *places, int n);


/* Backward of tds_raster_scene with respect to the poses of the actors and cameras.  The CV2 backend of the reference has no
 * gradient (rendering/cv2.py:27-70 runs in numpy); this one is build-defined (edge sampling of the actors' outlines against the
 * forward image, DESIGN.md "K3 backward") and plays the role of the pytorch3d backend's soft-blend gradient
 * (rendering/pytorch3d.py:57-119) for policy learning through the renderer.
 *   image, grad_out  B x Nc x 3 x H x W float32: the forward output (TDS_OUT_F32) and the incoming gradient
 *   grad_agent       B x Nc x N x 4   [d/dx, d/dy, d/dsin(psi), d/dcos(psi)] of actor n as seen by camera c (caller sums over c)
 *   grad_cam         B x Nc x 4       [d/dcx, d/dcy, d/dsin, d/dcos] of the camera: every colour boundary of the image moves with it
 *   grad_tmpl        optional (NULL to skip), B x Nc x N x 7 x 2: d/d(template vertex v) of actor n as seen by camera c -- the actor's
 *                    outline in its own frame (mesh.py:911-996).  The caller sums over c and chains to the actor's length and width
 *                    through the construction of the template.
 * All outputs are overwritten. */
int tds_raster_scene_bwd_f32(const float *state, const float *agent_sc, const float *tmpl, const uint8_t *mask, const float *cam_xy,
                             const float *cam_sc, const float *image, const float *grad_out, int64_t B, int64_t Nc, int64_t N,
                             float scale, int res, float *grad_agent, float *grad_cam, float *grad_tmpl, void *stream);

/* The same gradient computed from the forward's key-index slices (tds_raster_aux_t) instead of the forward image: the incoming gradient
 * is read only next to colour boundaries.  keys / n_keys: HOST, the key table the forward launch reported.
 * grad_color (optional, NULL to skip): B x Nc x 16 x 4 -- entry [i][ch] (ch 0..2 = R, G, B; [3] = 0) is the gradient with respect to
 * channel ch of the colour shown for key index i (0 = background, i >= 1: keys[i - 1]) in that camera: the sum of grad_out over the
 * pixels whose winning key it is (exact: the image is colour[index] pixel by pixel).  Asking for it reads grad_out in full.
 * grad_out_stride: floats between the gradient images of consecutive cameras -- 3 res^2 for a dense B x Nc x 3 x H x W gradient, 0 when
 * ONE 3 x H x W image is the gradient of every camera (losses like image.sum() or a fixed linear read-out hand back a broadcast). */
int tds_raster_scene_bwd_idx_f32(const float *state, const float *agent_sc, const float *tmpl, const uint8_t *mask, const float *cam_xy,
                                 const float *cam_sc, const uint32_t *index_slices, const uint32_t *keys, int n_keys, const float *grad_out,
                                 int64_t grad_out_stride, int64_t B, int64_t Nc, int64_t N, float scale, int res, float *grad_agent,
                                 float *grad_cam, float *grad_color, float *grad_tmpl, void *stream);

/* Generic BirdviewRenderer.render_rgb_mesh (rendering/base.py:206-212) for an arbitrary per-camera RGB mesh:
 *   verts n_img x V x 3 (x, y, z), attrs n_img x V x 3 in [0,1], faces n_img x F x 3 int32,
 *   levels: n_levels HOST floats sorted descending containing every z in use (<= 255)
 *   out   n_img x 3 x H x W  (CHW, i.e. already permuted as render_frame returns it, base.py:202-203)
 *   flags TDS_RASTER_* (0: the trim rule of cv2.py:32-41 is applied) */
int tds_raster_mesh(const float *verts, const float *attrs, const int32_t *faces, int64_t n_img, int64_t V, int64_t F,
                    const float *cam_xy, const float *cam_sc, const float *levels, int n_levels, float scale, int res,
                    int out_mode, void *out, int flags, void *stream);

/* ------------------------------------------------------------------------------------------------------------
 * Wrong-way query           simulator.py:607-630 (compute_wrong_way); infractions.py:232-304 (lanelet_orientation_loss);
 *                           lanelet2.py:108-180 (find_lanelet_directions, find_direction)
 * The reference keeps a lanelet2.core.LaneletMap (Lanelet2 C++ objects, not part of the reference's sources) and queries it agent by
 * agent.  Here a map is flattened once into a LANE TABLE and a batch is one kernel launch.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct tds_lanes tds_lanes_t;
typedef struct tds_laneset tds_laneset_t;

/* Centre line of one lanelet (`lanelet.centerline`, lanelet2.py:128): HOST arrays, bounds n x 3 float64 in travel order, out must hold
 * (n_left + n_right + 1) x 3 doubles, *n_out = number of points written (0 when a bound is empty). */
int tds_lanelet_centerline_f64(const double *left, int n_left, const double *right, int n_right, double *out, int *n_out);

/* HOST inputs: poly_xy P x 2 float64 = outline rings (left bound followed by the reversed right bound), lanelet l owns points
 * poly_start[l] .. poly_start[l+1]; cl_xyz C x 3 float64 = centre lines, cl_start likewise; flags (n_lanelets, may be NULL): bit 0 =
 * the lanelet carries one of the tags to exclude (infractions.py:21).  cell_size <= 0 selects the default (8 m).  Queries may use
 * any lanelet_dist_tolerance up to max_tolerance.  The handle lives on the CURRENT HIP device. */
int tds_lanes_create(const double *poly_xy, const int32_t *poly_start, const double *cl_xyz, const int32_t *cl_start,
                     const int32_t *flags, int n_lanelets, float cell_size, float max_tolerance, tds_lanes_t **out);
int tds_lanes_destroy(tds_lanes_t *lanes);
/* info[0..3] = lanelets, grid nx, grid ny, device bytes held */
int tds_lanes_info(const tds_lanes_t *lanes, int64_t *info);
/* the lane tables of a batch (`lanelet_maps: List[Optional[LaneletMap]]`, infractions.py:232): a device array of views; the tables must
 * outlive the set */
int tds_laneset_create(const tds_lanes_t *const *lanes, int n, tds_laneset_t **out);
int tds_laneset_destroy(tds_laneset_t *set);

/* lanelet_orientation_loss [* present]:
 *   state n x 4 (x, y, psi, v); agent a belongs to scene a / agents_per_scene, which uses table scene_map[scene] of the set
 *   (device int32; a negative entry = `None`, loss 0; NULL = table 0 for every scene); recenter_offset (scenes x 2 or NULL) is added to
 *   the position (infractions.py:271-273); present (n uint8 or NULL) multiplies the result (simulator.py:624).
 *   out n = min over the lanelets within lanelet_dist_tolerance of -cos(d) * [|d| > direction_angle_threshold],
 *   d = normalize_angle(lanelet direction - psi); 0 without such a lanelet, when one of them carries an excluded tag, or when
 *   find_direction fails (LaneletError, infractions.py:290-294). */
int tds_wrong_way_f32(const tds_laneset_t *set, const int32_t *scene_map, int64_t agents_per_scene, const float *state,
                      const float *recenter_offset, const uint8_t *present, float *out, int64_t n_agents,
                      float direction_angle_threshold, float lanelet_dist_tolerance, void *stream);
/* find_lanelet_directions for a batch of points (xy n x 2 float64, as the reference passes host doubles): dirs / dists n x max_dirs
 * float64 (direction and distance of the first max_dirs lanelets found, in table order), count n (number found, may exceed max_dirs; 0 when excluded), status n uint8 (bit 0: find_direction
 * failed for some lanelet, bit 1: a lanelet with an excluded tag is within tolerance) */
int tds_lanelet_directions_f64(const tds_laneset_t *set, const int32_t *scene_map, int64_t points_per_scene, const double *xy,
                               double *dirs, double *dists, int32_t *count, uint8_t *status, int max_dirs, int64_t n_points,
                               float lanelet_dist_tolerance, void *stream);

/* ---- testing hooks ------------------------------------------------------------------------------------------------------------
 * NOT part of the product: libtdship.so exports none of these.  They exist in libtdship_testing.so, the same sources compiled with
 * -DTDS_TESTING, which tools/ (ablations, work counters) and a few tests (forcing the slow code paths) load instead. */
#ifdef TDS_TESTING
/* force the LDS strip width of K3 (0 = automatic, else 8 .. 128 output rows) */
int tds_raster_set_strip_width(int tw);
/* waves per workgroup of the bit-plane raster kernel (4 or 8) */
int tds_raster_set_bits_waves(int n);
/* K3r, the list rasteriser of the split bit-plane path: LDS budget per workgroup in KiB (sets the strip width) */
int tds_raster_set_list_lds(int lds_kb);
/* K3r: waves per workgroup (2 or 4; 0 = chosen by the size of a strip) */
int tds_raster_set_list_waves(int waves);
/* ablation switches of K3: 1 no static map, 2 no actors, 4 no store, 8 no outline edges, 16 no scan conversion, 32 no binned path,
 * 64 no bit planes, 128 work counters on, 512 no per-face set-up (nothing is painted), 1024 walk the grid but project nothing,
 * 2048 the 170-VGPR instantiation everywhere, 4096 the work counters hold per-XCD finish [0..7] and ~start [8..15] wall clocks (100 MHz),
 * 8192 never the split form (K3s + K3r), 16384 the split form wherever a workspace allows it, 262144 / 524288 an image whose planes do not fit three
 * workgroups per CU (six and more keys at 256 x 256) whole in 4-wave workgroups / in half-image strips instead of whole in 8-wave ones, 1048576 / 2097152
 * strips as wide as the LDS allows instead of equal ones / never one strip more than necessary (nine and more keys), 32768 K3r without its short path for
 * small faces, 65536 / 131072 the persistent launch with 8 / 4 workgroups per CU (round 3's surplus) instead of the resident number */
int tds_raster_set_debug(int flags);
/* read and reset the 16 work counters of the bit-plane kernel */
int tds_raster_get_stats(unsigned long long *out16);
/* 0: maps created from now on carry no nearest-face candidate lists (K2b then walks grid rings) */
int tds_testing_set_near_lists(int enabled);
#endif

#ifdef __cplusplus
}
#endif
#endif /* TDSHIP_H */
