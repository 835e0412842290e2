// What does FETCH_SIZE (rocprofv3 --pmc) count for GATHERS of 16-byte pieces?  MI355X_MICROARCH.md: on gfx950 the counter reports half the bytes
// of a wide coalesced streaming read (128-byte requests tallied at 64 bytes) and "other access widths are uncalibrated: calibrate on a known
// byte count in your own access pattern".  The raster backward (raster_bwd.hip) reads 16-byte pieces of scattered 64-byte sectors; round 4
// doubled its raw FETCH_SIZE (4.05 -> 8.1 GB) and concluded that half of every 128-byte line is fetched in vain.  Four patterns over a buffer
// far larger than the 256 MiB Infinity Cache, each lane one float4:
//   dense   consecutive float4s (every byte read)                          expected bytes: N
//   sector  one float4 per 64-byte sector, every sector                    sectors touched: N        lines touched: N
//   line    one float4 per 128-byte line (every other sector)              sectors touched: N / 2    lines touched: N
//   sparse  one float4 per 128-byte line, the lanes of a wave 1 KiB apart  sectors touched: N / 2    lines touched: N     (no two lanes share a line)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/gather_calibrate tools/gather_calibrate.hip ; tools/gather_calibrate.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void gather_kernel(const float4 *buf, float *out, size_t n_items, size_t stride16, size_t lane_stride16, int sparse) {
    float acc = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (size_t)gridDim.x * blockDim.x) {
        size_t at = i * stride16;
        if (sparse) {
            // wave w reads 64 pieces 1 KiB apart, the next wave starts one line further: every line is still read exactly once
            const size_t w = i >> 6, l = i & 63;
            const size_t group = w / 8, in_group = w % 8;                   // 8 waves x 64 lanes cover 64 KiB = 512 lines
            at = group * (64 * 64) + l * 64 + in_group * 8;                 // units of 16 bytes: 1 KiB = 64, a line = 8
        }
        const float4 v = buf[at];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char **argv) {
    const char *mode = argc > 1 ? argv[1] : "dense";
    const size_t bytes = (size_t)4 << 30;
    float4 *buf; float *out;
    CK(hipMalloc((void **)&buf, bytes)); CK(hipMalloc((void **)&out, 64));
    CK(hipMemset(buf, 0, bytes));
    size_t stride = 1, items = bytes / 16;
    int sparse = 0;
    if (!strcmp(mode, "sector")) { stride = 4; items = bytes / 64; }
    if (!strcmp(mode, "line")) { stride = 8; items = bytes / 128; }
    if (!strcmp(mode, "sparse")) { stride = 8; items = bytes / 128; sparse = 1; }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(gather_kernel, dim3(256 * 16), dim3(256), 0, 0, buf, out, items, stride, (size_t)0, sparse);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2) printf("%s: %zu pieces of 16 B over %.2f GiB in %.3f ms\n", mode, items, bytes / 1073741824.0, ms);
    }
    return 0;
}
