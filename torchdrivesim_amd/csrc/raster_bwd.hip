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
// same integral over every colour boundary of the image (neighbouring pixel pairs), static map included.  NOT produced: gradients
// with respect to actor sizes (templates) and colours.
#include "tds_common.h"

namespace {

constexpr int BW_BLOCK = 256;
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
    int N, Nc, res;
    float scale;
};

__global__ void __launch_bounds__(BW_BLOCK) raster_scene_bwd_kernel(BwdArgs a) {
    __shared__ float cam_acc[4];
    const int64_t img = blockIdx.x;
    const int64_t b = img / a.Nc;
    const int tid = threadIdx.x, sub = tid & (LANES_PER_AGENT - 1);
    const int res = a.res;
    const float k = a.scale * (float)res * 0.5f, half = (float)res * 0.5f;
    const float2 cxy = a.cam_xy[img], csc = a.cam_sc[img];
    const float cs = csc.x, cc = csc.y;
    const int64_t plane = (int64_t)res * res;
    const float *I = a.image + img * 3 * plane, *G = a.grad_out + img * 3 * plane;
    if (tid < 4) cam_acc[tid] = 0.0f;
    __syncthreads();
    float gcam[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float view_r = 1.05f * 1.41421356f / a.scale;
    for (int j0 = 0; j0 < a.N; j0 += BW_BLOCK / LANES_PER_AGENT) {
        const int j = j0 + tid / LANES_PER_AGENT;
        float g[4] = {0.0f, 0.0f, 0.0f, 0.0f};
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
                }
            }
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
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = gcam[q];
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        if ((tid & 63) == 0) atomicAdd(&cam_acc[q], v);
    }
    __syncthreads();
    if (tid < 4) a.grad_cam[img * 4 + tid] = cam_acc[tid];
}

}  // namespace

TDS_EXPORT int tds_raster_scene_bwd_f32(const float *state, const float *agent_sc, const float *tmpl, const uint8_t *mask, const float *cam_xy,
                                        const float *cam_sc, const float *image, const float *grad_out, int64_t B, int64_t Nc, int64_t N,
                                        float scale, int res, float *grad_agent, float *grad_cam, void *stream) {
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
    a.grad_agent = grad_agent; a.grad_cam = grad_cam; a.N = (int)N; a.Nc = (int)Nc; a.res = res; a.scale = scale;
    hipLaunchKernelGGL(raster_scene_bwd_kernel, dim3((unsigned)n_img), dim3(BW_BLOCK), 0, (hipStream_t)stream, a);
    TDS_LAUNCH_CHECK("raster_scene_bwd_kernel");
    return TDS_OK;
}
