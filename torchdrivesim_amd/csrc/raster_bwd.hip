// K3 backward: gradient of the rendered bird's-eye view with respect to the poses of the actors and of the cameras.
//
// The reference's CV2 backend has no gradient (numpy; rendering/cv2.py:27-70) and its differentiable backends (pytorch3d soft
// blend, rendering/pytorch3d.py:57-119) are not runnable here, so this backward is BUILD-DEFINED and documented in
// the rendered image is piecewise constant, and the derivative of a pixel-integrated loss  L = sum_p f(p) I(p)  with
// respect to a parameter that moves a polygon is the boundary integral (Reynolds transport; "edge sampling")
//
//      dL/dtheta = integral over the polygon's boundary of  f(s) (I_in(s) - I_out(s)) (n(s) . ds/dtheta) dl
//
// with n the outward normal.  The kernel evaluates it for the outlines of the actors (body rectangle, direction triangle;
// mesh.py:911-996) as drawn into each camera: samples one pixel apart along every edge, I_in / I_out read from the FORWARD image
// just inside / just outside the drawn edge, f = the incoming gradient averaged over the same two pixels.  Occlusion needs no special
// case: where another actor covers the edge both sides show the same colour and the term vanishes.
// Gradients produced: actor position (x, y) and heading (through [sin, cos], the form the forward consumes) per (camera, actor) --
// the host sums over cameras -- and camera position / heading: a camera move shifts the WHOLE image rigidly, so that gradient is the
// same integral over every colour boundary of the image (neighbouring pixel pairs), static map included.  Optionally the gradient with
// respect to the seven TEMPLATE vertices of every actor (its outline in its own frame, mesh.py:911-996): a sample at parameter u of the edge
// t_a -> t_b moves with (1 - u) dt_a + u dt_b, so the same weights A0, A1 give d/dt_a and d/dt_b; the host chains them to the actor's
// length and width through the template's construction.
#include "tds_common.h"
#include <algorithm>

namespace {

#ifndef TDS_BW_BLOCK
#define TDS_BW_BLOCK 256
#endif
constexpr int BW_BLOCK = TDS_BW_BLOCK;
#ifndef TDS_BW_OCC
#define TDS_BW_OCC 4          // workgroups per CU the index kernel's registers are cut for
#endif
#ifndef TDS_BW_QCOLS
#define TDS_BW_QCOLS 32       // columns of a slab whose boundary cells are queued at a time (32: the whole slab, up to 2 048 cells; 16: 1 024)
#endif
constexpr int BW_QCOLS = TDS_BW_QCOLS, BW_QCELLS = 64 * BW_QCOLS;
constexpr int LANES_PER_AGENT = 8;          // 4 body edges + 3 direction-triangle edges (+1 idle)
// Where the colours on the two sides of an edge are read.  The forward draws the polygon through the TRUNCATED vertices with OpenCV's
// rules (fill + outline): in continuous pixel coordinates that is the polygon through the centres of the vertex pixels, grown by about
// half a pixel.  Samples therefore run along the edge between the vertex-pixel centres; the inner tap sits 0.25 px inside it, the outer
// tap 1.25 px outside (beyond the grown border, 0.5 .. 0.71 px).
constexpr float SIDE_IN = 0.25f, SIDE_OUT = 1.25f;

struct BwdArgs {
    const float4 *state;        // B x N
    const float2 *agent_sc;     // B x N   [sin, cos]
    const float2 *tmpl;         // B x N x 7
    const uint8_t *mask;        // B x Nc x N
    const float2 *cam_xy, *cam_sc;   // B x Nc
    const float *image;         // B x Nc x 3 x res x res   (forward output)
    const float *grad_out;      // same shape
    float *grad_agent;          // B x Nc x N x 4   [d/dx, d/dy, d/dsin, d/dcos]
    float *grad_cam;            // B x Nc x 4       [d/dcx, d/dcy, d/dsin, d/dcos]
    float2 *grad_tmpl;          // B x Nc x N x 7   d/d(template vertex), or nullptr
    int N, Nc, res;
    float scale;
};

__global__ void __launch_bounds__(BW_BLOCK) raster_scene_bwd_kernel(BwdArgs a) {
    __shared__ float cam_part[(BW_BLOCK / 64) * 4];
    const int64_t img = blockIdx.x;
    const int64_t b = img / a.Nc;
    const int tid = threadIdx.x, sub = tid & (LANES_PER_AGENT - 1);
    const int res = a.res;
    const float k = a.scale * (float)res * 0.5f, half = (float)res * 0.5f;
    const float2 cxy = a.cam_xy[img], csc = a.cam_sc[img];
    const float cs = csc.x, cc = csc.y;
    const int64_t plane = (int64_t)res * res;
    const float *I = a.image + img * 3 * plane, *G = a.grad_out + img * 3 * plane;
    __syncthreads();
    float gcam[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float view_r = 1.05f * 1.41421356f / a.scale;
    for (int j0 = 0; j0 < a.N; j0 += BW_BLOCK / LANES_PER_AGENT) {
        const int j = j0 + tid / LANES_PER_AGENT;
        float g[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float2 gta = make_float2(0.0f, 0.0f), gtb = make_float2(0.0f, 0.0f);          // this edge's share of d/dt_a and d/dt_b
        if (j < a.N && sub < 7 && a.mask[img * a.N + j] != 0) {
            const int64_t ia = b * a.N + j;
            const float4 s = a.state[ia];
            const float2 sc = a.agent_sc[ia];
            const float2 t0 = a.tmpl[ia * 7];
            const float reach = view_r + sqrtf(t0.x * t0.x + t0.y * t0.y) + 2.0f / a.scale;
            const float ddx = s.x - cxy.x, ddy = s.y - cxy.y;
            if (ddx * ddx + ddy * ddy <= reach * reach) {
                // edge end points (template space) and the centroid of their polygon
                const int ea = sub < 4 ? sub : sub, eb = sub < 4 ? ((sub + 1) & 3) : (sub == 6 ? 4 : sub + 1);
                const float2 ta = a.tmpl[ia * 7 + ea], tb = a.tmpl[ia * 7 + eb];
                float2 ctr;
                if (sub < 4) {
                    const float2 q1 = a.tmpl[ia * 7 + 1], q2 = a.tmpl[ia * 7 + 2], q3 = a.tmpl[ia * 7 + 3];
                    ctr = make_float2(0.25f * (t0.x + q1.x + q2.x + q3.x), 0.25f * (t0.y + q1.y + q2.y + q3.y));
                } else {
                    const float2 q4 = a.tmpl[ia * 7 + 4], q5 = a.tmpl[ia * 7 + 5], q6 = a.tmpl[ia * 7 + 6];
                    ctr = make_float2((q4.x + q5.x + q6.x) / 3.0f, (q4.y + q5.y + q6.y) / 3.0f);
                }
                // template -> world (relative to the camera) -> continuous pixel coordinates:  p = -k Rc (w - cam) + res/2
                auto to_rel = [&](float2 t) { return make_float2(sc.y * t.x - sc.x * t.y + s.x - cxy.x, sc.x * t.x + sc.y * t.y + s.y - cxy.y); };
                auto to_pix = [&](float2 v) { return make_float2(-k * (cc * v.x + cs * v.y) + half, -k * (-cs * v.x + cc * v.y) + half); };
                const float2 va = to_rel(ta), vb = to_rel(tb), pc = to_pix(to_rel(ctr));
                float2 pa = to_pix(va), pb = to_pix(vb);
                pa = make_float2(floorf(pa.x) + 0.5f, floorf(pa.y) + 0.5f);       // centres of the vertex pixels
                pb = make_float2(floorf(pb.x) + 0.5f, floorf(pb.y) + 0.5f);
                const float ex = pb.x - pa.x, ey = pb.y - pa.y, len = sqrtf(ex * ex + ey * ey);
                if (len > 1e-6f && len < 1e5f) {
                    float nx = ey / len, ny = -ex / len;
                    if (nx * (0.5f * (pa.x + pb.x) - pc.x) + ny * (0.5f * (pa.y + pb.y) - pc.y) < 0.0f) { nx = -nx; ny = -ny; }
                    const int ns = max(1, min(4096, (int)ceilf(len)));
                    const float dl = len / (float)ns;
                    float A0 = 0.0f, A1 = 0.0f;
                    for (int si = 0; si < ns; ++si) {
                        const float u = ((float)si + 0.5f) / (float)ns;
                        const float px = pa.x + u * ex, py = pa.y + u * ey;
                        const float xi = floorf(px - SIDE_IN * nx), yi = floorf(py - SIDE_IN * ny), xo = floorf(px + SIDE_OUT * nx), yo = floorf(py + SIDE_OUT * ny);
                        if (xi < 0.0f || yi < 0.0f || xo < 0.0f || yo < 0.0f || xi >= (float)res || yi >= (float)res || xo >= (float)res || yo >= (float)res)
                            continue;
                        const int64_t oi = (int64_t)xi * res + (int64_t)yi, oo = (int64_t)xo * res + (int64_t)yo;      // out[ch][x][y]
                        float D = 0.0f;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch)
                            D += 0.5f * (G[ch * plane + oi] + G[ch * plane + oo]) * (I[ch * plane + oi] - I[ch * plane + oo]);
                        const float wgt = D * dl;
                        A0 += wgt * (1.0f - u); A1 += wgt * u;
                    }
                    // n . dp/dtheta for the parameters, with M = -k Rc (a similarity): M q = -k (cc qx + cs qy, -cs qx + cc qy)
                    auto nM = [&](float qx, float qy) { return -k * (nx * (cc * qx + cs * qy) + ny * (-cs * qx + cc * qy)); };
                    const float At = A0 + A1;
                    const float Tx = A0 * ta.x + A1 * tb.x, Ty = A0 * ta.y + A1 * tb.y;       // weighted template point
                    g[0] = At * nM(1.0f, 0.0f);
                    g[1] = At * nM(0.0f, 1.0f);
                    g[2] = nM(-Ty, Tx);                 // d w / d sin_j = (-ty, tx)
                    g[3] = nM(Tx, Ty);                  // d w / d cos_j = ( tx, ty)
                    const float qx = nM(sc.y, sc.x), qy = nM(-sc.x, sc.y);       // d w / d t.x = (cos_j, sin_j), d w / d t.y = (-sin_j, cos_j)
                    gta = make_float2(A0 * qx, A0 * qy); gtb = make_float2(A1 * qx, A1 * qy);
                }
            }
        }
        if (a.grad_tmpl != nullptr) {
            // vertex v = sub starts edge v and ends the edge before it in its polygon (body 0..3, direction triangle 4..6)
            const int prev = (tid & ~(LANES_PER_AGENT - 1)) + (sub < 4 ? ((sub + 3) & 3) : (sub == 4 ? 6 : sub - 1));
            const float bx = __shfl(gtb.x, prev & 63), by = __shfl(gtb.y, prev & 63);
            if (j < a.N && sub < 7) a.grad_tmpl[(img * a.N + j) * 7 + sub] = make_float2(gta.x + bx, gta.y + by);
        }
        // sum the edges of an agent (8 consecutive lanes) and store; agents out of sight store zeros
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            g[q] += __shfl_xor(g[q], 1); g[q] += __shfl_xor(g[q], 2); g[q] += __shfl_xor(g[q], 4);
        }
        if (j < a.N && sub == 0) *(float4 *)(a.grad_agent + (img * a.N + j) * 4) = make_float4(g[0], g[1], g[2], g[3]);
    }
    // Camera: when the camera moves, EVERYTHING in the image moves rigidly (static map and actors alike), so its gradient is the
    // same boundary integral taken over every colour boundary of the image, i.e. over neighbouring pixel pairs:
    //   dL/dtheta = - sum_pairs f (I_B - I_A) (n . dp/dtheta),   dp/dcx = k (cc, -cs),  dp/dcy = k (cs, cc),
    //   dp/dcos = Rc^T (p - centre),  dp/dsin = (rot -90) of it      (p = -k Rc (w - cam) + centre)
    // Memory: this pass reads the forward image and the incoming gradient once (2 x 786 432 B per camera at 256 x 256: the algorithmic
    // bytes of the kernel).  A thread owns four neighbouring pixels of the fastest axis (one float4 per channel and array) and walks down a
    // band of rows keeping the previous row in registers; the pixel to the right of its four comes from the next lane.
    {
        float Sx = 0.0f, Sy = 0.0f, Cc = 0.0f, Cs = 0.0f;
        if ((res & 3) == 0 && res >= 8) {
            const int CG = res >> 2;                                     // column groups per row
            const int NB = max(1, min(res, BW_BLOCK / CG));              // row bands
            const int RB = (res + NB - 1) / NB;
            const int lane = tid & 63;
            for (int item = tid; item < ((CG * NB + 63) & ~63); item += BW_BLOCK) {      // whole waves iterate together (shuffles below)
                const bool live = item < CG * NB;
                const int cg = live ? item % CG : 0, band = live ? item / CG : 0;
                const int jc = cg * 4, r0 = band * RB, r1 = live ? min(res, r0 + RB) : r0;
                float4 pi[3], pg[3];
                if (r0 < r1) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        pi[ch] = *(const float4 *)(I + ch * plane + (int64_t)r0 * res + jc);
                        pg[ch] = *(const float4 *)(G + ch * plane + (int64_t)r0 * res + jc);
                    }
                }
                for (int rr = 0; rr < RB; ++rr) {                        // every lane of the wave runs RB rows (`on` masks the short last band)
                    const int i = r0 + rr;
                    const bool on = i < r1;
                    // pairs along the fastest axis (pixel y): the fifth pixel is the first one of the next lane's group
                    float ni[3], ng[3];
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        ni[ch] = __shfl_down(pi[ch].x, 1);
                        ng[ch] = __shfl_down(pg[ch].x, 1);
                    }
                    const bool has_right = on && cg + 1 < CG;
                    if (has_right && lane == 63) {                           // the neighbour group sits in the next wave
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            ni[ch] = I[ch * plane + (int64_t)i * res + jc + 4];
                            ng[ch] = G[ch * plane + (int64_t)i * res + jc + 4];
                        }
                    }
                    if (on) {
                        const float dxr = (float)i + 0.5f - half;
                        float D[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            D[0] += 0.5f * (pg[ch].x + pg[ch].y) * (pi[ch].y - pi[ch].x);
                            D[1] += 0.5f * (pg[ch].y + pg[ch].z) * (pi[ch].z - pi[ch].y);
                            D[2] += 0.5f * (pg[ch].z + pg[ch].w) * (pi[ch].w - pi[ch].z);
                            D[3] += 0.5f * (pg[ch].w + ng[ch]) * (ni[ch] - pi[ch].w);
                        }
                        if (!has_right) D[3] = 0.0f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float dy = (float)(jc + q + 1) - half;
                            Sy += D[q]; Cc += D[q] * (cs * dxr + cc * dy); Cs += -D[q] * (cc * dxr - cs * dy);
                        }
                    }
                    // pairs along the slow axis (pixel x): this row and the next one
                    if (on && i + 1 < res) {
                        float4 qi[3], qg[3];
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            qi[ch] = *(const float4 *)(I + ch * plane + (int64_t)(i + 1) * res + jc);
                            qg[ch] = *(const float4 *)(G + ch * plane + (int64_t)(i + 1) * res + jc);
                        }
                        float D[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            D[0] += 0.5f * (pg[ch].x + qg[ch].x) * (qi[ch].x - pi[ch].x);
                            D[1] += 0.5f * (pg[ch].y + qg[ch].y) * (qi[ch].y - pi[ch].y);
                            D[2] += 0.5f * (pg[ch].z + qg[ch].z) * (qi[ch].z - pi[ch].z);
                            D[3] += 0.5f * (pg[ch].w + qg[ch].w) * (qi[ch].w - pi[ch].w);
                        }
                        const float dx = (float)(i + 1) - half;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float dy = (float)(jc + q) + 0.5f - half;
                            Sx += D[q]; Cc += D[q] * (cc * dx - cs * dy); Cs += D[q] * (cs * dx + cc * dy);
                        }
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) pi[ch] = qi[ch], pg[ch] = qg[ch];
                    }
                }
            }
        } else {
        for (int idx = tid; idx < res * res; idx += BW_BLOCK) {
            const int i = idx / res, j = idx - i * res;                       // out[ch][i][j]: i = pixel x, j = pixel y
            const int64_t o0 = (int64_t)i * res + j;
            if (i + 1 < res) {
                const int64_t o1 = o0 + res;
                float D = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) D += 0.5f * (G[ch * plane + o0] + G[ch * plane + o1]) * (I[ch * plane + o1] - I[ch * plane + o0]);
                const float dx = (float)(i + 1) - half, dy = (float)j + 0.5f - half;
                Sx += D; Cc += D * (cc * dx - cs * dy); Cs += D * (cs * dx + cc * dy);
            }
            if (j + 1 < res) {
                const int64_t o1 = o0 + 1;
                float D = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) D += 0.5f * (G[ch * plane + o0] + G[ch * plane + o1]) * (I[ch * plane + o1] - I[ch * plane + o0]);
                const float dx = (float)i + 0.5f - half, dy = (float)(j + 1) - half;
                Sy += D; Cc += D * (cs * dx + cc * dy); Cs += -D * (cc * dx - cs * dy);
            }
        }
        }
        gcam[0] = -k * (cc * Sx - cs * Sy);
        gcam[1] = -k * (cs * Sx + cc * Sy);
        gcam[2] = -Cs;
        gcam[3] = -Cc;
    }
    // fixed order: a butterfly inside every wave, then the waves' partial sums one after the other (no atomics: the result does not depend
    // on which wave finishes first)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = gcam[q];
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if ((tid & 63) == 0) cam_part[(tid >> 6) * 4 + q] = v;
    }
    __syncthreads();
    if (tid < 4) {
        float v = 0.0f;
        for (int wv = 0; wv < BW_BLOCK / 64; ++wv) v += cam_part[wv * 4 + tid];
        a.grad_cam[img * 4 + tid] = v;
    }
}


// ---------------------------------------------------------------------------------------------------------
// The same gradient from the forward's KEY-INDEX SLICES instead of the forward image (tds_raster_aux_t::index_slices: per pixel the
// 1-based position of the winning key, as bit-slices, 3 % of the image's bytes).  The image is a function of that index alone --
// I[ch](p) = colour[ch][idx(p)] -- so every term  f (I_B - I_A)  vanishes unless the two pixels differ in idx, and colour boundaries are
// found 32 pixel pairs at a time by XOR-ing slice words.  The incoming gradient is then read only where a boundary is: 16-byte pieces
// (four pixels of the output's fastest axis), a row of the image per wave instruction when every lane takes part.
//   index_slices  uint32 [camera][x / 32][y / 4][slice][y % 4], bit x % 32     (x, y: OpenCV pixel coordinates; out[ch][x][y])
// One workgroup per camera.  Camera part: a wave owns a slab of 32 columns x, lane = row quad; it walks the columns, keeps the previous
// column's gradient in registers (pairs along x) and takes the first row of the next quad from the next lane (pairs along y).
// ---------------------------------------------------------------------------------------------------------
__device__ inline void wave_sync_bwd() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#ifndef TDS_BWD_ABLATE
#define TDS_BWD_ABLATE 0      // tuning builds: 1 no actor pass, 2 no cell processing, 4 no cell queue either
#endif

struct BwdIdxArgs {
    const float4 *state;
    const float2 *agent_sc;
    const float2 *tmpl;
    const uint8_t *mask;
    const float2 *cam_xy, *cam_sc;
    const uint32_t *slices;
    const float *grad_out;
    float *grad_agent, *grad_cam;
    float2 *grad_tmpl;          // optional: B x Nc x N x 7, d/d(template vertex)
    float *grad_color;          // optional: B x Nc x 16 x 4: per key index (0 = background) the sum of the incoming gradient per channel
    int N, Nc, res, nb;         // nb: slices in use (index bits)
    int64_t gstride;            // floats between the gradient images of consecutive cameras (3 res^2; 0 = one image shared by all)
    float scale;
    uint32_t keys[16];          // ascending key table of the forward launch
    int n_keys;
};

template <int NB>
__global__ void __launch_bounds__(BW_BLOCK, TDS_BW_OCC) raster_scene_bwd_idx_kernel(BwdIdxArgs a) {
    __shared__ float cam_part[(BW_BLOCK / 64) * 4];
    __shared__ float4 col_tab[16];                                       // colour of key index i (w unused)
    // dynamic LDS: per wave the owner records + cell queue of the camera pass; then (only when grad_color is asked for) reused as
    // [16 keys][3 channels][BW_BLOCK threads] private sums of the colour pass
    extern __shared__ __attribute__((aligned(16))) float col_priv[];
    const int64_t img = blockIdx.x;
    const int64_t b = img / a.Nc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int res = a.res, quads = res >> 2, wprT = (res + 31) >> 5;
    const float k = a.scale * (float)res * 0.5f, half = (float)res * 0.5f;
    const float2 cxy = a.cam_xy[img], csc = a.cam_sc[img];
    const float cs = csc.x, cc = csc.y;
    const int64_t plane = (int64_t)res * res;
    const float *G = a.grad_out + img * a.gstride;
    const uint32_t *S = a.slices + (size_t)img * wprT * quads * 16;
    if (tid < 16) {
        uint32_t key = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) key = (tid == i + 1 && i < a.n_keys) ? a.keys[i] : key;       // a.keys lives in SGPRs
        col_tab[tid] = make_float4((float)((key >> 16) & 255u), (float)((key >> 8) & 255u), (float)(key & 255u), 0.0f);
    }
    __syncthreads();
    auto idx_at = [&](int x, int y) {
        const uint32_t *w = S + (((size_t)(x >> 5) * quads + (size_t)(y >> 2)) << 4) + (y & 3);
        int idx = 0;
#pragma unroll
        for (int q = 0; q < NB; ++q) idx |= (int)((w[4 * q] >> (x & 31)) & 1u) << q;
        return idx;
    };
    float gcam[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float view_r = 1.05f * 1.41421356f / a.scale;
    // ---- actors: the definition of raster_scene_bwd_kernel (same samples, same taps), the two taps of a sample read the key index instead
    // of the image.  Few of a scene's agents are in a camera's view: thread j culls agent j and the survivors go to an LDS list; a whole
    // wave then takes one listed agent at a time -- 8 lanes per outline edge, which share the edge's samples -- so that a 33-pixel edge
    // costs five dependent round trips to memory instead of 33.
    __shared__ int vis_list[BW_BLOCK];
    __shared__ int vis_count;
    for (int j0 = 0; j0 < a.N; j0 += BW_BLOCK) {
        if (tid == 0) vis_count = 0;
        __syncthreads();
        const int jc = j0 + tid;
        if (jc < a.N) {
            bool vis = false;
            if (!(TDS_BWD_ABLATE & 1) && a.mask[img * a.N + jc] != 0) {
                const int64_t ia = b * a.N + jc;
                const float4 s = a.state[ia];
                const float2 t0 = a.tmpl[ia * 7];
                const float reach = view_r + sqrtf(t0.x * t0.x + t0.y * t0.y) + 2.0f / a.scale;
                const float ddx = s.x - cxy.x, ddy = s.y - cxy.y;
                vis = ddx * ddx + ddy * ddy <= reach * reach;
            }
            if (vis) vis_list[atomicAdd(&vis_count, 1)] = jc;
            else {                                                                                             // out of sight or masked
                *(float4 *)(a.grad_agent + (img * a.N + jc) * 4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (a.grad_tmpl != nullptr)
                    for (int v = 0; v < 7; ++v) a.grad_tmpl[(img * a.N + jc) * 7 + v] = make_float2(0.0f, 0.0f);
            }
        }
        __syncthreads();
        const int nvis = vis_count;
        for (int vi = wave; vi < nvis; vi += BW_BLOCK / 64) {
            const int j = vis_list[vi];
            const int e = lane >> 3, sl = lane & 7;                       // edge 0..6 (7 idles), position among the edge's 8 lanes
            float g[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            float2 gta = make_float2(0.0f, 0.0f), gtb = make_float2(0.0f, 0.0f);      // this edge's share of d/dt_a and d/dt_b (all 8 lanes of the edge)
            if (e < 7) {
                const int64_t ia = b * a.N + j;
                const float4 s = a.state[ia];
                const float2 sc = a.agent_sc[ia];
                const float2 t0 = a.tmpl[ia * 7];
                const int ea = e, eb = e < 4 ? ((e + 1) & 3) : (e == 6 ? 4 : e + 1);
                const float2 ta = a.tmpl[ia * 7 + ea], tb = a.tmpl[ia * 7 + eb];
                float2 ctr;
                if (e < 4) {
                    const float2 q1 = a.tmpl[ia * 7 + 1], q2 = a.tmpl[ia * 7 + 2], q3 = a.tmpl[ia * 7 + 3];
                    ctr = make_float2(0.25f * (t0.x + q1.x + q2.x + q3.x), 0.25f * (t0.y + q1.y + q2.y + q3.y));
                } else {
                    const float2 q4 = a.tmpl[ia * 7 + 4], q5 = a.tmpl[ia * 7 + 5], q6 = a.tmpl[ia * 7 + 6];
                    ctr = make_float2((q4.x + q5.x + q6.x) / 3.0f, (q4.y + q5.y + q6.y) / 3.0f);
                }
                auto to_rel = [&](float2 t) { return make_float2(sc.y * t.x - sc.x * t.y + s.x - cxy.x, sc.x * t.x + sc.y * t.y + s.y - cxy.y); };
                auto to_pix = [&](float2 v) { return make_float2(-k * (cc * v.x + cs * v.y) + half, -k * (-cs * v.x + cc * v.y) + half); };
                const float2 va = to_rel(ta), vb = to_rel(tb), pc = to_pix(to_rel(ctr));
                float2 pa = to_pix(va), pb = to_pix(vb);
                pa = make_float2(floorf(pa.x) + 0.5f, floorf(pa.y) + 0.5f);
                pb = make_float2(floorf(pb.x) + 0.5f, floorf(pb.y) + 0.5f);
                const float ex = pb.x - pa.x, ey = pb.y - pa.y, len = sqrtf(ex * ex + ey * ey);
                if (len > 1e-6f && len < 1e5f) {
                    float nx = ey / len, ny = -ex / len;
                    if (nx * (0.5f * (pa.x + pb.x) - pc.x) + ny * (0.5f * (pa.y + pb.y) - pc.y) < 0.0f) { nx = -nx; ny = -ny; }
                    const int ns = max(1, min(4096, (int)ceilf(len)));
                    const float dl = len / (float)ns;
                    float A0 = 0.0f, A1 = 0.0f;
                    for (int si = sl; si < ns; si += 8) {
                        const float u = ((float)si + 0.5f) / (float)ns;
                        const float px = pa.x + u * ex, py = pa.y + u * ey;
                        const float xi = floorf(px - SIDE_IN * nx), yi = floorf(py - SIDE_IN * ny), xo = floorf(px + SIDE_OUT * nx), yo = floorf(py + SIDE_OUT * ny);
                        if (xi < 0.0f || yi < 0.0f || xo < 0.0f || yo < 0.0f || xi >= (float)res || yi >= (float)res || xo >= (float)res || yo >= (float)res)
                            continue;
                        const int ki = idx_at((int)xi, (int)yi), ko = idx_at((int)xo, (int)yo);
                        if (ki == ko) continue;
                        const float4 ci = col_tab[ki], co = col_tab[ko];
                        const int64_t oi = (int64_t)xi * res + (int64_t)yi, oo = (int64_t)xo * res + (int64_t)yo;
                        const float D = 0.5f * (G[oi] + G[oo]) * (ci.x - co.x) + 0.5f * (G[plane + oi] + G[plane + oo]) * (ci.y - co.y) +
                                        0.5f * (G[2 * plane + oi] + G[2 * plane + oo]) * (ci.z - co.z);
                        const float wgt = D * dl;
                        A0 += wgt * (1.0f - u); A1 += wgt * u;
                    }
                    // the edge's samples were shared by 8 lanes
                    A0 += __shfl_xor(A0, 1); A0 += __shfl_xor(A0, 2); A0 += __shfl_xor(A0, 4);
                    A1 += __shfl_xor(A1, 1); A1 += __shfl_xor(A1, 2); A1 += __shfl_xor(A1, 4);
                    auto nM = [&](float qx, float qy) { return -k * (nx * (cc * qx + cs * qy) + ny * (-cs * qx + cc * qy)); };
                    const float At = A0 + A1;
                    const float Tx = A0 * ta.x + A1 * tb.x, Ty = A0 * ta.y + A1 * tb.y;
                    if (sl == 0) {
                        g[0] = At * nM(1.0f, 0.0f);
                        g[1] = At * nM(0.0f, 1.0f);
                        g[2] = nM(-Ty, Tx);
                        g[3] = nM(Tx, Ty);
                    }
                    const float qx = nM(sc.y, sc.x), qy = nM(-sc.x, sc.y);
                    gta = make_float2(A0 * qx, A0 * qy); gtb = make_float2(A1 * qx, A1 * qy);
                }
            }
            if (a.grad_tmpl != nullptr) {
                // vertex v = e starts edge e and ends the edge before it in its polygon (see raster_scene_bwd_kernel)
                const int pe = e < 4 ? ((e + 3) & 3) : (e == 4 ? 6 : e - 1);
                const float bx = __shfl(gtb.x, (pe & 7) * 8), by = __shfl(gtb.y, (pe & 7) * 8);
                if (e < 7 && sl == 0) a.grad_tmpl[(img * a.N + j) * 7 + e] = make_float2(gta.x + bx, gta.y + by);
            }
            // sum the seven edges (their first lanes hold the values, every other lane holds zero)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                for (int d = 1; d < 64; d <<= 1) g[q] += __shfl_xor(g[q], d);
            if (lane == 0) *(float4 *)(a.grad_agent + (img * a.N + j) * 4) = make_float4(g[0], g[1], g[2], g[3]);
        }
    }
    // ---- camera: every colour boundary of the image (see raster_scene_bwd_kernel for the formulas)
    // Only about one in seven (quad, column) cells of a Town01 view touches a boundary, but nine columns in ten do somewhere along their
    // 256 rows: walking the columns with a lane per quad would keep the wave busy on empty cells.  Instead every lane lists the columns of
    // its quad that hold the FIRST pixel of a differing pair -- (x, y)-(x+1, y) or (x, y)-(x, y+1) -- into a per-wave LDS queue, the slice
    // words go to LDS beside it, and the wave then takes the queued cells 64 at a time, whoever they belong to.
    {
        float Sx = 0.0f, Sy = 0.0f, Cc = 0.0f, Cs = 0.0f;
        constexpr int OWN_DW = (NB * 5 + 1 + 3) & ~3;                         // per quad: its NB x 4 slice words, row 0 of the quad below, bit 0 of the next word column (16-byte records)
        uint32_t *wl = (uint32_t *)col_priv + wave * (64 * OWN_DW + BW_QCELLS / 2);    // per wave: 64 owner records, then up to BW_QCELLS cells as uint16
        unsigned short *cells = (unsigned short *)(wl + 64 * OWN_DW);
        const int qblocks = (quads + 63) >> 6;
        for (int slab = wave; slab < wprT * qblocks; slab += BW_BLOCK / 64) {
            const int xw = slab / qblocks, rq0 = (slab - xw * qblocks) * 64, rq = rq0 + lane;
            const bool live = rq < quads;
            const int xbase = xw * 32;
            const int ncol = min(32, res - xbase);                      // columns of this word that exist
            const bool has_next_word = xw + 1 < wprT;
            uint32_t s[NB][4], nxt = 0, below[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                uint4 v = make_uint4(0, 0, 0, 0), vn = make_uint4(0, 0, 0, 0);
                if (live) {
                    v = *(const uint4 *)(S + (((size_t)xw * quads + (size_t)rq) << 4) + 4 * q);
                    if (has_next_word) vn = *(const uint4 *)(S + (((size_t)(xw + 1) * quads + (size_t)rq) << 4) + 4 * q);
                }
                s[q][0] = v.x; s[q][1] = v.y; s[q][2] = v.z; s[q][3] = v.w;
                nxt |= ((vn.x & 1u) | ((vn.y & 1u) << 1) | ((vn.z & 1u) << 2) | ((vn.w & 1u) << 3)) << (4 * q);
            }
            const bool has_below = live && rq + 1 < quads;
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                below[q] = (uint32_t)__shfl_down((int)s[q][0], 1);
                if (lane == 63 && has_below) below[q] = S[(((size_t)xw * quads + (size_t)(rq + 1)) << 4) + 4 * q];
            }
            const uint32_t colmask = ncol == 32 ? 0xffffffffu : ((1u << ncol) - 1u);
            uint32_t xvalid = colmask >> 1;                               // pair (x, x+1) inside this word's existing columns ...
            if (has_next_word && ncol == 32) xvalid |= 0x80000000u;       // ... or reaching into the next word
            uint32_t m = 0;                                               // bit x: some pair starting in column x of this quad differs
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t dx = 0, dy = 0;
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    dx |= s[q][j] ^ ((s[q][j] >> 1) | (((nxt >> (4 * q + j)) & 1u) << 31));
                    dy |= s[q][j] ^ (j < 3 ? s[q][j + 1] : below[q]);
                }
                m |= (dx & xvalid) | ((j < 3 || has_below) ? (dy & colmask) : 0u);
            }
            if (!live || (TDS_BWD_ABLATE & 4)) m = 0;
            // owner records and the queue of cells
            wave_sync_bwd();
#pragma unroll
            for (int q = 0; q < NB; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wl[lane * OWN_DW + q * 4 + j] = s[q][j];
                wl[lane * OWN_DW + NB * 4 + q] = below[q];
            }
            wl[lane * OWN_DW + NB * 4 + NB] = nxt;
            // The queue is filled COLUMN by column (a ballot per column: which row quads have a boundary there), not lane by lane: neighbouring
            // entries are then neighbouring row quads of one column, i.e. 16-byte pieces of the SAME 64-byte sector of the gradient (out[ch][x][y]:
            // four row quads per sector), and the lanes of a gather share their sectors instead of touching up to 64 different ones.
            for (int xq = 0; xq < 32; xq += BW_QCOLS) {                       // BW_QCOLS columns of the slab at a time: what the queue holds
            int total = 0;
#pragma unroll
            for (int xx = 0; xx < BW_QCOLS; ++xx) {
                const int x = xq + xx;
                const bool mine = (m >> x) & 1u;
                const unsigned long long bq = __ballot(mine);
                if (mine) cells[total + __builtin_amdgcn_mbcnt_hi((unsigned)(bq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bq, 0))] = (unsigned short)((lane << 5) | x);
                total += __popcll(bq);
            }
            wave_sync_bwd();
            for (int base = 0; base < ((TDS_BWD_ABLATE & 2) ? 0 : total); base += 64) {
                const int it = base + lane;
                if (it >= total) continue;
                const int cell = cells[it], o = cell >> 5, x = cell & 31;
                const uint32_t *rec = wl + o * OWN_DW;
                const int rqo = rq0 + o, y0 = rqo * 4;
                int i0[4], i1[4], ib = 0;
                const uint32_t nx = rec[NB * 4 + NB];
#pragma unroll
                for (int j = 0; j < 4; ++j) { i0[j] = 0; i1[j] = 0; }
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const uint4 w4 = *(const uint4 *)(rec + 4 * q);           // records are 16-byte aligned
                    const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        i0[j] |= (int)((w[j] >> x) & 1u) << q;
                        i1[j] |= (int)(x < 31 ? ((w[j] >> (x + 1)) & 1u) : ((nx >> (4 * q + j)) & 1u)) << q;
                    }
                    ib |= (int)((rec[NB * 4 + q] >> x) & 1u) << q;
                }
                const bool xok = xbase + x + 1 < res && (x < 31 || has_next_word), bok = rqo + 1 < quads;
                const float *gp = G + (int64_t)(xbase + x) * res + y0;
                const float4 a0 = *(const float4 *)gp, a1 = *(const float4 *)(gp + plane), a2 = *(const float4 *)(gp + 2 * plane);
                const float g0[4][3] = {{a0.x, a1.x, a2.x}, {a0.y, a1.y, a2.y}, {a0.z, a1.z, a2.z}, {a0.w, a1.w, a2.w}};
                const bool anyx = xok && (i0[0] != i1[0] || i0[1] != i1[1] || i0[2] != i1[2] || i0[3] != i1[3]);
                float4 c0[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) c0[j] = col_tab[i0[j]];
                if (anyx) {
                    const float4 b0 = *(const float4 *)(gp + res), b1 = *(const float4 *)(gp + res + plane), b2 = *(const float4 *)(gp + res + 2 * plane);
                    const float g1[4][3] = {{b0.x, b1.x, b2.x}, {b0.y, b1.y, b2.y}, {b0.z, b1.z, b2.z}, {b0.w, b1.w, b2.w}};
                    const float dx = (float)(xbase + x + 1) - half;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (i0[j] == i1[j]) continue;
                        const float4 c1 = col_tab[i1[j]];
                        const float D = 0.5f * (g0[j][0] + g1[j][0]) * (c1.x - c0[j].x) + 0.5f * (g0[j][1] + g1[j][1]) * (c1.y - c0[j].y) +
                                        0.5f * (g0[j][2] + g1[j][2]) * (c1.z - c0[j].z);
                        const float dy = (float)(y0 + j) + 0.5f - half;
                        Sx += D; Cc += D * (cc * dx - cs * dy); Cs += D * (cs * dx + cc * dy);
                    }
                }
                const float dxr = (float)(xbase + x) + 0.5f - half;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (i0[j] == i0[j + 1]) continue;
                    const float D = 0.5f * (g0[j][0] + g0[j + 1][0]) * (c0[j + 1].x - c0[j].x) + 0.5f * (g0[j][1] + g0[j + 1][1]) * (c0[j + 1].y - c0[j].y) +
                                    0.5f * (g0[j][2] + g0[j + 1][2]) * (c0[j + 1].z - c0[j].z);
                    const float dy = (float)(y0 + j + 1) - half;
                    Sy += D; Cc += D * (cs * dxr + cc * dy); Cs += -D * (cc * dxr - cs * dy);
                }
                if (bok && i0[3] != ib) {
                    const float4 cb = col_tab[ib];
                    const float D = 0.5f * (g0[3][0] + gp[4]) * (cb.x - c0[3].x) + 0.5f * (g0[3][1] + gp[plane + 4]) * (cb.y - c0[3].y) +
                                    0.5f * (g0[3][2] + gp[2 * plane + 4]) * (cb.z - c0[3].z);
                    const float dy = (float)(y0 + 4) - half;
                    Sy += D; Cc += D * (cs * dxr + cc * dy); Cs += -D * (cc * dxr - cs * dy);
                }
            }
            wave_sync_bwd();                                              // the queue is refilled for the next columns
            }
        }
        gcam[0] = -k * (cc * Sx - cs * Sy);
        gcam[1] = -k * (cs * Sx + cc * Sy);
        gcam[2] = -Cs;
        gcam[3] = -Cc;
    }
    // fixed order: a butterfly inside every wave, then the waves' partial sums one after the other (no atomics: the result does not depend
    // on which wave finishes first)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = gcam[q];
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if ((tid & 63) == 0) cam_part[(tid >> 6) * 4 + q] = v;
    }
    __syncthreads();
    if (tid < 4) {
        float v = 0.0f;
        for (int wv = 0; wv < BW_BLOCK / 64; ++wv) v += cam_part[wv * 4 + tid];
        a.grad_cam[img * 4 + tid] = v;
    }
    // ---- colours (optional): dL/dcolour[key][ch] = sum of the incoming gradient over the pixels that show the key -- exact, the image
    // being colour[key index] pixel by pixel.  This pass reads the whole gradient.  Every thread sums runs of equal index along x in
    // registers and adds finished runs to its private column of an LDS table (no atomics: the order of the additions is fixed).
    if (a.grad_color != nullptr) {
        __syncthreads();                                                 // the camera pass is done with the LDS it shares
        for (int e = tid; e < 48 * BW_BLOCK; e += BW_BLOCK) col_priv[e] = 0.0f;
        const int qblocks = (quads + 63) >> 6;
        for (int slab = wave; slab < wprT * qblocks; slab += BW_BLOCK / 64) {
            const int xw = slab / qblocks, rq = (slab - xw * qblocks) * 64 + lane;
            if (rq >= quads) continue;
            const int y0 = rq * 4, xbase = xw * 32, ncol = min(32, res - xbase);
            uint32_t s[NB][4];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const uint4 v = *(const uint4 *)(S + (((size_t)xw * quads + (size_t)rq) << 4) + 4 * q);
                s[q][0] = v.x; s[q][1] = v.y; s[q][2] = v.z; s[q][3] = v.w;
            }
            int run[4] = {0, 0, 0, 0};
            float acc[4][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            for (int x = 0; x < ncol; ++x) {
                const float *gp = G + (int64_t)(xbase + x) * res + y0;
                const float4 g0 = *(const float4 *)gp, g1 = *(const float4 *)(gp + plane), g2 = *(const float4 *)(gp + 2 * plane);
                const float gv[4][3] = {{g0.x, g1.x, g2.x}, {g0.y, g1.y, g2.y}, {g0.z, g1.z, g2.z}, {g0.w, g1.w, g2.w}};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int v = 0;
#pragma unroll
                    for (int q = 0; q < NB; ++q) v |= (int)((s[q][j] >> x) & 1u) << q;
                    if (v != run[j]) {
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) { col_priv[(run[j] * 3 + ch) * BW_BLOCK + tid] += acc[j][ch]; acc[j][ch] = 0.0f; }
                        run[j] = v;
                    }
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) acc[j][ch] += gv[j][ch];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) col_priv[(run[j] * 3 + ch) * BW_BLOCK + tid] += acc[j][ch];
        }
        __syncthreads();
        // 48 rows of BW_BLOCK partial sums: one wave per row, four values per lane, then a butterfly
        for (int row = wave; row < 48; row += BW_BLOCK / 64) {
            float v = 0.0f;
            for (int e = lane; e < BW_BLOCK; e += 64) v += col_priv[row * BW_BLOCK + e];
            for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
            if (lane == 0) a.grad_color[img * 64 + (row / 3) * 4 + (row % 3)] = v;
        }
        if (tid < 16) a.grad_color[img * 64 + tid * 4 + 3] = 0.0f;
    }
}

}  // namespace

TDS_EXPORT int tds_raster_scene_bwd_f32(const float *state, const float *agent_sc, const float *tmpl, const uint8_t *mask, const float *cam_xy,
                                        const float *cam_sc, const float *image, const float *grad_out, int64_t B, int64_t Nc, int64_t N,
                                        float scale, int res, float *grad_agent, float *grad_cam, float *grad_tmpl, void *stream) {
    TDS_CHECK_ARG(B >= 0 && Nc >= 0 && N >= 0 && N < (1 << 20), "tds_raster_scene_bwd_f32: bad sizes");
    TDS_CHECK_ARG(res > 0 && res <= 4096, "tds_raster_scene_bwd_f32: resolution out of range");
    TDS_CHECK_ARG(scale > 0.0f, "tds_raster_scene_bwd_f32: scale must be positive");
    const int64_t n_img = B * Nc;
    if (n_img == 0) return TDS_OK;
    TDS_CHECK_ARG(n_img < (1ll << 31), "tds_raster_scene_bwd_f32: too many cameras");
    TDS_CHECK_ARG(cam_xy && cam_sc && image && grad_out && grad_cam, "tds_raster_scene_bwd_f32: null array");
    TDS_CHECK_ARG(N == 0 || (state && agent_sc && tmpl && mask && grad_agent), "tds_raster_scene_bwd_f32: null agent array");
    BwdArgs a;
    a.state = (const float4 *)state; a.agent_sc = (const float2 *)agent_sc; a.tmpl = (const float2 *)tmpl; a.mask = mask;
    a.cam_xy = (const float2 *)cam_xy; a.cam_sc = (const float2 *)cam_sc; a.image = image; a.grad_out = grad_out;
    a.grad_agent = grad_agent; a.grad_cam = grad_cam; a.grad_tmpl = (float2 *)grad_tmpl; a.N = (int)N; a.Nc = (int)Nc; a.res = res; a.scale = scale;
    hipLaunchKernelGGL(raster_scene_bwd_kernel, dim3((unsigned)n_img), dim3(BW_BLOCK), 0, (hipStream_t)stream, a);
    TDS_LAUNCH_CHECK("raster_scene_bwd_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_raster_scene_bwd_idx_f32(const float *state, const float *agent_sc, const float *tmpl, const uint8_t *mask, const float *cam_xy,
                                            const float *cam_sc, const uint32_t *index_slices, const uint32_t *keys, int n_keys, const float *grad_out,
                                            int64_t grad_out_stride, int64_t B, int64_t Nc, int64_t N, float scale, int res, float *grad_agent,
                                            float *grad_cam, float *grad_color, float *grad_tmpl, void *stream) {
    TDS_CHECK_ARG(B >= 0 && Nc >= 0 && N >= 0 && N < (1 << 20), "tds_raster_scene_bwd_idx_f32: bad sizes");
    TDS_CHECK_ARG(res > 0 && res <= 4096 && (res & 3) == 0, "tds_raster_scene_bwd_idx_f32: the resolution must be a multiple of 4");
    TDS_CHECK_ARG(scale > 0.0f, "tds_raster_scene_bwd_idx_f32: scale must be positive");
    TDS_CHECK_ARG(keys && n_keys >= 1 && n_keys <= 15, "tds_raster_scene_bwd_idx_f32: the key table of the forward launch (1..15 keys) is required");
    const int64_t n_img = B * Nc;
    if (n_img == 0) return TDS_OK;
    TDS_CHECK_ARG(n_img < (1ll << 31), "tds_raster_scene_bwd_idx_f32: too many cameras");
    TDS_CHECK_ARG(cam_xy && cam_sc && index_slices && grad_out && grad_cam, "tds_raster_scene_bwd_idx_f32: null array");
    TDS_CHECK_ARG(N == 0 || (state && agent_sc && tmpl && mask && grad_agent), "tds_raster_scene_bwd_idx_f32: null agent array");
    BwdIdxArgs a;
    a.state = (const float4 *)state; a.agent_sc = (const float2 *)agent_sc; a.tmpl = (const float2 *)tmpl; a.mask = mask;
    a.cam_xy = (const float2 *)cam_xy; a.cam_sc = (const float2 *)cam_sc; a.slices = index_slices; a.grad_out = grad_out;
    a.grad_agent = grad_agent; a.grad_cam = grad_cam; a.grad_color = grad_color; a.grad_tmpl = (float2 *)grad_tmpl; a.N = (int)N; a.Nc = (int)Nc; a.res = res; a.scale = scale;
    TDS_CHECK_ARG(grad_out_stride == 0 || grad_out_stride >= 3ll * res * res, "tds_raster_scene_bwd_idx_f32: grad_out_stride must be 0 or at least 3 res^2");
    a.gstride = grad_out_stride;
    a.n_keys = n_keys;
    for (int i = 0; i < 16; ++i) a.keys[i] = i < n_keys ? keys[i] : 0u;
    a.nb = n_keys <= 3 ? 2 : (n_keys <= 7 ? 3 : 4);                 // as bits_index_bits of the forward
    // camera pass: per wave 64 owner records (NB * 5 + 1 words each, padded to 16 bytes) + 2048 uint16 cells; colour pass: 48 x BW_BLOCK floats
    const size_t lds_cam = (size_t)(BW_BLOCK / 64) * (64 * ((a.nb * 5 + 1 + 3) & ~3) + BW_QCELLS / 2) * sizeof(uint32_t);
    const size_t lds = std::max(lds_cam, grad_color ? (size_t)48 * BW_BLOCK * sizeof(float) : (size_t)0);
    if (a.nb == 2) hipLaunchKernelGGL(raster_scene_bwd_idx_kernel<2>, dim3((unsigned)n_img), dim3(BW_BLOCK), lds, (hipStream_t)stream, a);
    else if (a.nb == 3) hipLaunchKernelGGL(raster_scene_bwd_idx_kernel<3>, dim3((unsigned)n_img), dim3(BW_BLOCK), lds, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(raster_scene_bwd_idx_kernel<4>, dim3((unsigned)n_img), dim3(BW_BLOCK), lds, (hipStream_t)stream, a);
    TDS_LAUNCH_CHECK("raster_scene_bwd_idx_kernel");
    return TDS_OK;
}
