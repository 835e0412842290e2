// K1: kinematic model steps over coalesced (n,4) state tensors, one thread per agent, float4 / float2 accesses.
// Reference: torchdrivesim/kinematic.py:328-523 (KinematicBicycle.step :462-477, BicycleNoReversing.step :513-523,
// SimpleKinematicModel.step :362-367, OrientedKinematicModel.step :384-389).  Roofline: 44 B/agent, launch bound.
#include "tds_common.h"

namespace {

constexpr int KBLOCK = 256;

__global__ void __launch_bounds__(KBLOCK) bicycle_step_kernel(const float4 *__restrict__ state, const float2 *__restrict__ action,
                                                              const float *__restrict__ lr, float4 *__restrict__ out, int64_t n,
                                                              float dt, float max_acc, float max_steer, int left_handed,
                                                              int no_reversing) {
    int64_t i = (int64_t)blockIdx.x * KBLOCK + threadIdx.x;
    if (i >= n) return;
    float4 s = state[i];
    float2 a2 = action[i];
    float a = a2.x * max_acc;                 // denormalize_action, kinematic.py:459-460
    float beta = a2.y * max_steer;
    if (no_reversing) {                       // :513-523
        bool rev = (s.w + a * dt) < 0.0f;
        float macc = rev ? (-s.w) / dt : a;
        a = (macc / max_acc) * max_acc;       // normalize then denormalize, as the reference does
        beta = (beta / max_steer) * max_steer;
    }
    if (left_handed) beta = -beta;            // :466-467
    float v = s.w + a * dt;
    float pb = s.z + beta;
    float x = s.x + (v * cosf(pb)) * dt;
    float y = s.y + (v * sinf(pb)) * dt;
    float psi = s.z + ((v / lr[i]) * sinf(beta)) * dt;   // no angle wrap (:475)
    out[i] = make_float4(x, y, psi, v);
}

__global__ void __launch_bounds__(KBLOCK) bicycle_step_bwd_kernel(const float4 *__restrict__ state, const float2 *__restrict__ action,
                                                                  const float *__restrict__ lr, const float4 *__restrict__ gout,
                                                                  float4 *__restrict__ gstate, float2 *__restrict__ gaction,
                                                                  float *__restrict__ glr, int64_t n, float dt, float max_acc,
                                                                  float max_steer, int left_handed, int no_reversing) {
    int64_t i = (int64_t)blockIdx.x * KBLOCK + threadIdx.x;
    if (i >= n) return;
    float4 s = state[i];
    float2 a2 = action[i];
    float4 g = gout[i];
    float l = lr[i];
    float a = a2.x * max_acc;
    float beta = a2.y * max_steer;
    bool rev = false;
    if (no_reversing) {
        rev = (s.w + a * dt) < 0.0f;
        if (rev) a = (-s.w) / dt;
    }
    float sgn = left_handed ? -1.0f : 1.0f;
    beta = sgn * beta;
    float v = s.w + a * dt;
    float pb = s.z + beta;
    float cpb = cosf(pb), spb = sinf(pb), sb = sinf(beta), cb = cosf(beta);
    // d loss / d v'
    float gv = g.w + g.x * cpb * dt + g.y * spb * dt + g.z * (sb * dt / l);
    float gang = g.x * (-v * spb * dt) + g.y * (v * cpb * dt);   // through psi + beta
    float gpsi = g.z + gang;
    float gbeta = gang + g.z * (v / l) * cb * dt;
    // v' = v + a*dt ; a = act*max_acc, or a = -v/dt when reversing (then dv'/dv = 0, dv'/dact = 0)
    float gv_in = rev ? 0.0f : gv;
    float ga = rev ? 0.0f : gv * dt * max_acc;
    if (gstate) gstate[i] = make_float4(g.x, g.y, gpsi, gv_in);
    if (gaction) gaction[i] = make_float2(ga, gbeta * sgn * max_steer);
    if (glr) glr[i] = g.z * (-(v / (l * l)) * sb * dt);
}

struct Norm4 { float v[4]; };

__global__ void __launch_bounds__(KBLOCK) simple_step_kernel(const float4 *__restrict__ state, const float4 *__restrict__ action,
                                                             float4 *__restrict__ out, int64_t n, float dt, Norm4 nm, int oriented) {
    int64_t i = (int64_t)blockIdx.x * KBLOCK + threadIdx.x;
    if (i >= n) return;
    float4 s = state[i];
    float4 a = action[i];
    if (oriented) {                           // utils.rotate :56-69, [[c,-s],[s,c]] @ (a.x, a.y)
        float c = cosf(s.z), sn = sinf(s.z);
        float r0 = c * a.x + (-sn) * a.y;
        float r1 = sn * a.x + c * a.y;
        a.x = r0; a.y = r1;
    }
    out[i] = make_float4(s.x + (a.x * nm.v[0]) * dt, s.y + (a.y * nm.v[1]) * dt, s.z + (a.z * nm.v[2]) * dt,
                         s.w + (a.w * nm.v[3]) * dt);
}

__global__ void __launch_bounds__(KBLOCK) simple_step_bwd_kernel(const float4 *__restrict__ state, const float4 *__restrict__ action,
                                                                 const float4 *__restrict__ gout, float4 *__restrict__ gstate,
                                                                 float4 *__restrict__ gaction, int64_t n, float dt, Norm4 nm, int oriented) {
    int64_t i = (int64_t)blockIdx.x * KBLOCK + threadIdx.x;
    if (i >= n) return;
    float4 s = state[i];
    float4 a = action[i];
    float4 g = gout[i];
    float gr0 = g.x * nm.v[0] * dt, gr1 = g.y * nm.v[1] * dt;   // grads w.r.t. the (rotated) xy action
    float gpsi = g.z;
    float gax = gr0, gay = gr1;
    if (oriented) {
        float c = cosf(s.z), sn = sinf(s.z);
        gax = c * gr0 + sn * gr1;
        gay = -sn * gr0 + c * gr1;
        // d r0/d psi = -sn*a.x - c*a.y ; d r1/d psi = c*a.x - sn*a.y
        gpsi += gr0 * (-sn * a.x - c * a.y) + gr1 * (c * a.x - sn * a.y);
    }
    if (gstate) gstate[i] = make_float4(g.x, g.y, gpsi, g.w);
    if (gaction) gaction[i] = make_float4(gax, gay, g.z * nm.v[2] * dt, g.w * nm.v[3] * dt);
}

__global__ void __launch_bounds__(KBLOCK) unicycle_step_kernel(const float4 *__restrict__ state, const float2 *__restrict__ action,
                                                               float4 *__restrict__ out, int64_t n, float dt, float max_acc, float max_w) {
    int64_t i = (int64_t)blockIdx.x * KBLOCK + threadIdx.x;
    if (i >= n) return;
    float4 s = state[i];
    float2 a2 = action[i];
    float v = s.w + (a2.x * max_acc) * dt;
    float x = s.x + (v * cosf(s.z)) * dt;
    float y = s.y + (v * sinf(s.z)) * dt;
    float psi = s.z + (a2.y * max_w) * dt;
    out[i] = make_float4(x, y, psi, v);
}

__global__ void __launch_bounds__(KBLOCK) unicycle_step_bwd_kernel(const float4 *__restrict__ state, const float2 *__restrict__ action,
                                                                   const float4 *__restrict__ gout, float4 *__restrict__ gstate,
                                                                   float2 *__restrict__ gaction, int64_t n, float dt, float max_acc, float max_w) {
    int64_t i = (int64_t)blockIdx.x * KBLOCK + threadIdx.x;
    if (i >= n) return;
    float4 s = state[i];
    float2 a2 = action[i];
    float4 g = gout[i];
    float v = s.w + (a2.x * max_acc) * dt;
    float c = cosf(s.z), sn = sinf(s.z);
    float gv = g.w + g.x * c * dt + g.y * sn * dt;
    float gpsi = g.z + g.x * (-v * sn * dt) + g.y * (v * c * dt);
    if (gstate) gstate[i] = make_float4(g.x, g.y, gpsi, gv);
    if (gaction) gaction[i] = make_float2(gv * dt * max_acc, g.z * dt * max_w);
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + KBLOCK - 1) / KBLOCK)); }

}  // namespace

#define TDS_KIN_ARGS_OK(n) TDS_CHECK_ARG((n) >= 0 && (n) < ((int64_t)1 << 31) * KBLOCK, "agent count %lld out of range", (long long)(n))

TDS_EXPORT int tds_bicycle_step_f32(const float *state, const float *action, const float *lr, float *out, int64_t n, float dt,
                                    float max_acc, float max_steer, int left_handed, int no_reversing, void *stream) {
    TDS_KIN_ARGS_OK(n);
    if (n == 0) return TDS_OK;
    TDS_CHECK_ARG(state && action && lr && out, "tds_bicycle_step_f32: null pointer");
    TDS_CHECK_ARG(state != out, "tds_bicycle_step_f32: out must not alias state");
    hipLaunchKernelGGL(bicycle_step_kernel, grid_for(n), dim3(KBLOCK), 0, (hipStream_t)stream, (const float4 *)state,
                       (const float2 *)action, lr, (float4 *)out, n, dt, max_acc, max_steer, left_handed, no_reversing);
    TDS_LAUNCH_CHECK("bicycle_step_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_bicycle_step_bwd_f32(const float *state, const float *action, const float *lr, const float *grad_out,
                                        float *grad_state, float *grad_action, float *grad_lr, int64_t n, float dt, float max_acc,
                                        float max_steer, int left_handed, int no_reversing, void *stream) {
    TDS_KIN_ARGS_OK(n);
    if (n == 0) return TDS_OK;
    TDS_CHECK_ARG(state && action && lr && grad_out, "tds_bicycle_step_bwd_f32: null pointer");
    hipLaunchKernelGGL(bicycle_step_bwd_kernel, grid_for(n), dim3(KBLOCK), 0, (hipStream_t)stream, (const float4 *)state,
                       (const float2 *)action, lr, (const float4 *)grad_out, (float4 *)grad_state, (float2 *)grad_action, grad_lr, n,
                       dt, max_acc, max_steer, left_handed, no_reversing);
    TDS_LAUNCH_CHECK("bicycle_step_bwd_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_simple_step_f32(const float *state, const float *action, float *out, int64_t n, float dt, const float *norm,
                                   int oriented, void *stream) {
    TDS_KIN_ARGS_OK(n);
    if (n == 0) return TDS_OK;
    TDS_CHECK_ARG(state && action && out && norm, "tds_simple_step_f32: null pointer");
    TDS_CHECK_ARG(state != out, "tds_simple_step_f32: out must not alias state");
    Norm4 nm{{norm[0], norm[1], norm[2], norm[3]}};
    hipLaunchKernelGGL(simple_step_kernel, grid_for(n), dim3(KBLOCK), 0, (hipStream_t)stream, (const float4 *)state,
                       (const float4 *)action, (float4 *)out, n, dt, nm, oriented);
    TDS_LAUNCH_CHECK("simple_step_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_simple_step_bwd_f32(const float *state, const float *action, const float *grad_out, float *grad_state,
                                       float *grad_action, int64_t n, float dt, const float *norm, int oriented, void *stream) {
    TDS_KIN_ARGS_OK(n);
    if (n == 0) return TDS_OK;
    TDS_CHECK_ARG(state && action && grad_out && norm, "tds_simple_step_bwd_f32: null pointer");
    Norm4 nm{{norm[0], norm[1], norm[2], norm[3]}};
    hipLaunchKernelGGL(simple_step_bwd_kernel, grid_for(n), dim3(KBLOCK), 0, (hipStream_t)stream, (const float4 *)state,
                       (const float4 *)action, (const float4 *)grad_out, (float4 *)grad_state, (float4 *)grad_action, n, dt, nm, oriented);
    TDS_LAUNCH_CHECK("simple_step_bwd_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_unicycle_step_f32(const float *state, const float *action, float *out, int64_t n, float dt, float max_acc,
                                     float max_yaw_rate, void *stream) {
    TDS_KIN_ARGS_OK(n);
    if (n == 0) return TDS_OK;
    TDS_CHECK_ARG(state && action && out, "tds_unicycle_step_f32: null pointer");
    TDS_CHECK_ARG(state != out, "tds_unicycle_step_f32: out must not alias state");
    hipLaunchKernelGGL(unicycle_step_kernel, grid_for(n), dim3(KBLOCK), 0, (hipStream_t)stream, (const float4 *)state,
                       (const float2 *)action, (float4 *)out, n, dt, max_acc, max_yaw_rate);
    TDS_LAUNCH_CHECK("unicycle_step_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_unicycle_step_bwd_f32(const float *state, const float *action, const float *grad_out, float *grad_state,
                                         float *grad_action, int64_t n, float dt, float max_acc, float max_yaw_rate, void *stream) {
    TDS_KIN_ARGS_OK(n);
    if (n == 0) return TDS_OK;
    TDS_CHECK_ARG(state && action && grad_out, "tds_unicycle_step_bwd_f32: null pointer");
    hipLaunchKernelGGL(unicycle_step_bwd_kernel, grid_for(n), dim3(KBLOCK), 0, (hipStream_t)stream, (const float4 *)state,
                       (const float2 *)action, (const float4 *)grad_out, (float4 *)grad_state, (float2 *)grad_action, n, dt, max_acc,
                       max_yaw_rate);
    TDS_LAUNCH_CHECK("unicycle_step_bwd_kernel");
    return TDS_OK;
}
