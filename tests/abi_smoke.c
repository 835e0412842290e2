/*
 * A plain C caller of libtdship.so (no Python, no torch): proves that the drop-in boundary is the C ABI of include/tdship.h alone.
 * Built with gcc by tests/test_abi.py::test_c_program_through_the_abi (GPU box) and checked for compilation on the CPU side.
 *   gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/abi_smoke.c -o abi_smoke -Ltorchdrivesim_amd/lib -ltdship -L/opt/rocm/lib -lamdhip64 -lm
 * 1. KinematicBicycle.step on the survey's known answer (kinematic.py:462-477): (1, 2, 0.5, 3), action (0.4, -0.2), lr 1.5, dt 0.1
 *    -> (1.31449008, 2.05912733, 0.43407637, 3.2);
 * 2. a scene of three boxes through tds_collision_f32 (IoU): two identical overlapping boxes and a far one -> (1, 1, 0) ... sum - max = (0 + ...)
 * 3. error behaviour: a null pointer is refused with TDS_EINVAL and a message, nothing is thrown across the boundary.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "tdship.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(void) {
    if (tds_version() != TDS_ABI_VERSION) { printf("FAIL version %d\n", tds_version()); return 1; }
    const float state[4] = {1.0f, 2.0f, 0.5f, 3.0f}, action[2] = {0.4f, -0.2f}, lr[1] = {1.5f};
    const float want[4] = {1.31449008f, 2.05912733f, 0.43407637f, 3.2f};
    float *d_state, *d_action, *d_lr, *d_out, out[4];
    CHECK_HIP(hipMalloc((void **)&d_state, sizeof state));
    CHECK_HIP(hipMalloc((void **)&d_action, sizeof action));
    CHECK_HIP(hipMalloc((void **)&d_lr, sizeof lr));
    CHECK_HIP(hipMalloc((void **)&d_out, sizeof out));
    CHECK_HIP(hipMemcpy(d_state, state, sizeof state, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_action, action, sizeof action, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_lr, lr, sizeof lr, hipMemcpyHostToDevice));
    int rc = tds_bicycle_step_f32(d_state, d_action, d_lr, d_out, 1, 0.1f, 5.0f, 1.5707963267948966f, 0, 0, NULL);
    if (rc != TDS_OK) { char msg[256]; tds_last_error(msg, sizeof msg); printf("FAIL step rc=%d %s\n", rc, msg); return 1; }
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i)
        if (fabsf(out[i] - want[i]) > 1e-5f * fmaxf(1.0f, fabsf(want[i]))) { printf("FAIL state[%d] = %.8f, want %.8f\n", i, out[i], want[i]); return 1; }

    /* three boxes [x, y, length, width, psi]: 0 and 1 coincide, 2 is far away; sc = [sin psi, cos psi]; everybody present */
    const float boxes[15] = {0, 0, 4, 2, 0, 0, 0, 4, 2, 0, 50, 50, 4, 2, 0}, sc[6] = {0, 1, 0, 1, 0, 1};
    const uint8_t present[3] = {1, 1, 1};
    float *d_boxes, *d_sc, *d_coll, coll[3];
    uint8_t *d_present;
    CHECK_HIP(hipMalloc((void **)&d_boxes, sizeof boxes));
    CHECK_HIP(hipMalloc((void **)&d_sc, sizeof sc));
    CHECK_HIP(hipMalloc((void **)&d_present, sizeof present));
    CHECK_HIP(hipMalloc((void **)&d_coll, sizeof coll));
    CHECK_HIP(hipMemcpy(d_boxes, boxes, sizeof boxes, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_sc, sc, sizeof sc, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_present, present, sizeof present, hipMemcpyHostToDevice));
    rc = tds_collision_f32(d_boxes, d_sc, d_present, d_coll, NULL, NULL, 1, 3, 3, TDS_METRIC_IOU, NULL);
    if (rc != TDS_OK) { char msg[256]; tds_last_error(msg, sizeof msg); printf("FAIL collision rc=%d %s\n", rc, msg); return 1; }
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(coll, d_coll, sizeof coll, hipMemcpyDeviceToHost));
    /* sum_j IoU - max_j IoU: agents 0 and 1 see (1 + 1 + 0) - 1 = 1, agent 2 sees (0 + 0 + 1) - 1 = 0 (simulator.py:1105-1108) */
    if (fabsf(coll[0] - 1.0f) > 1e-5f || fabsf(coll[1] - 1.0f) > 1e-5f || coll[2] != 0.0f) {
        printf("FAIL collision = %.6f %.6f %.6f\n", coll[0], coll[1], coll[2]);
        return 1;
    }

    /* an output buffer with spread-out physical pages (512 MiB: 64 chunks of 8 MiB), written and read back, and a small dense one */
    tds_buffer_t *buf = NULL, *small = NULL;
    int64_t nbytes = 0, chunks = 0;
    int spread = 0;
    rc = tds_buffer_create((int64_t)512 << 20, 0, 0, &buf);
    if (rc != TDS_OK || tds_buffer_info(buf, &nbytes, &chunks, &spread) != TDS_OK || !spread || chunks != 64 || nbytes != ((int64_t)512 << 20)) {
        char m[256]; tds_last_error(m, sizeof m); printf("FAIL tds_buffer_create rc=%d spread=%d chunks=%lld %s\n", rc, spread, (long long)chunks, m); return 1;
    }
    unsigned char *bp = (unsigned char *)tds_buffer_ptr(buf), back[4] = {0, 0, 0, 0};
    CHECK_HIP(hipMemset(bp, 0x5a, (size_t)512 << 20));
    CHECK_HIP(hipMemcpy(back, bp + ((size_t)512 << 20) - 4, 4, hipMemcpyDeviceToHost));
    if (back[0] != 0x5a || back[3] != 0x5a) { printf("FAIL the buffer does not hold what was written\n"); return 1; }
    if (tds_buffer_create(4096, 0, 0, &small) != TDS_OK || tds_buffer_info(small, NULL, NULL, &spread) != TDS_OK || spread) { printf("FAIL small buffer\n"); return 1; }
    if (tds_buffer_destroy(small) != TDS_OK || tds_buffer_destroy(buf) != TDS_OK) { printf("FAIL tds_buffer_destroy\n"); return 1; }

    rc = tds_bicycle_step_f32(NULL, d_action, d_lr, d_out, 1, 0.1f, 5.0f, 1.5707963f, 0, 0, NULL);
    char msg[256] = "";
    int len = tds_last_error(msg, sizeof msg);
    if (rc != TDS_EINVAL || len <= 0 || strlen(msg) == 0) { printf("FAIL error path rc=%d msg='%s'\n", rc, msg); return 1; }
    printf("abi smoke ok: step (%.6f %.6f %.6f %.6f), collision (%.3f %.3f %.3f), refusal '%s'\n", out[0], out[1], out[2], out[3], coll[0], coll[1],
           coll[2], msg);
    return 0;
}
