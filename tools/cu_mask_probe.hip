// Can the metric kernels run BESIDE the persistent raster launch if a few CUs are kept free of it?  
// The persistent launch holds three workgroups on every CU until the last image is out (52 KiB of LDS and 168 VGPRs each: nothing else
// fits beside them), so kernels of another stream only get the slots they grab in the first microseconds.  A stream created with
// hipExtStreamCreateWithCUMask keeps its kernels off the masked-out CUs.  This probe answers, on the box:
//   1. which physical CU (XCC, SE, SH, CU of HW_ID) a mask bit stands for -- every bit of the first 40 cleared in turn;
//   2. what the write pattern of the raster launch (tools/alloc_probe.hip: stream_kernel) loses when 1 / 2 / 4 CUs per XCD are masked out;
//   3. whether a small kernel on a second stream that may ONLY use the masked-out CUs runs at its own pace beside the launch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/cu_mask_probe tools/cu_mask_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CK(x)                                                                                        \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)

constexpr int IMG_BYTES = 3 * 256 * 256 * 4;
typedef float vf4 __attribute__((ext_vector_type(4)));

// every workgroup reports where it ran (XCC_ID << 16 | SE, SH, CU bits of HW_ID) after keeping its CU busy for a while
__global__ void where_kernel(uint32_t *out, int spin) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);         // HW_ID bits 15:0 ... size field = width - 1
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xffu);          // CU_ID 11:8, SH_ID 12, SE_ID 15:13
}

// the write pattern of the persistent raster launch (see tools/alloc_probe.hip)
__global__ void __launch_bounds__(256, 3) stream_kernel(char *out, int64_t n_img, uint32_t *queue, float value) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);
    const int64_t per = n_img >> 3;
    __shared__ int64_t s_img;
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            int64_t img = -1;
            for (int t = 0; t < 8 && img < 0; ++t) {
                const int q = (xcd + t) & 7;
                const uint32_t i = atomicAdd(&queue[q], 1u);
                if ((int64_t)i < per) img = q * per + (int64_t)i;
            }
            s_img = img;
        }
        __syncthreads();
        const int64_t img = s_img;
        if (img < 0) break;
        lds[tid] = (uint32_t)img;
        char *o = out + img * IMG_BYTES;
        const vf4 v = {value, value + (float)lds[tid ^ 1] * 0.0f, value, value};
        for (int pass = 0; pass < 2; ++pass) {
            const int xw = wave + 4 * pass;
            for (int ph = 0; ph < 8; ++ph)
                for (int m = 0; m < 4; ++m) {
                    const uint32_t off = (uint32_t)((xw * 32 + ph + 8 * m) * 256 + lane * 4) * 4u;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) __builtin_nontemporal_store(v, (vf4 *)(o + (size_t)ch * 262144 + off));
                }
        }
    }
}

// a stand-in for the metric kernels: n workgroups of 256 threads, each a chain of dependent loads + some arithmetic (latency-bound like K2b)
__global__ void metric_kernel(const float *table, float *out, int rounds) {
    int i = (blockIdx.x * 256 + threadIdx.x) & 0xfffff;
    float acc = 0.0f;
    for (int r = 0; r < rounds; ++r) {
        const float v = table[i];
        acc += v;
        i = (i * 1664525 + 1013904223 + (int)v) & 0xfffff;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static float elapsed(hipEvent_t a, hipEvent_t b) { float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int section = argc > 1 ? atoi(argv[1]) : 0;          // 0 all, 1 the bit -> CU listing only, 2 timings only
    int cus = 0;
    CK(hipSetDevice(0));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int words = (cus + 31) / 32;
    printf("%d CUs, mask of %d words\n", cus, words);
    uint32_t *d_where = nullptr;
    const int NW = 16384;
    CK(hipMalloc((void **)&d_where, NW * 4));
    std::vector<uint32_t> h(NW);
    auto used = [&](hipStream_t s) {
        CK(hipMemsetAsync(d_where, 0xff, NW * 4, s));
        hipLaunchKernelGGL(where_kernel, dim3(NW), dim3(64), 0, s, d_where, 200);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d_where, NW * 4, hipMemcpyDeviceToHost));
        return std::set<uint32_t>(h.begin(), h.end());
    };
    auto masked_stream = [&](const std::vector<uint32_t> &mask) {
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
        return s;
    };
    const std::vector<uint32_t> all(words, 0xffffffffu);
    hipStream_t s_all = masked_stream(all);
    const std::set<uint32_t> every = used(s_all);
    std::map<int, int> per_xcc;
    for (uint32_t v : every) per_xcc[v >> 16]++;
    printf("full mask: %zu distinct (XCC, SE/SH/CU) places;", every.size());
    for (auto &kv : per_xcc) printf(" XCC%d:%d", kv.first, kv.second);
    printf("\n");
    // 1. which place does bit i stand for?
    printf("bit -> missing place (xcc:hwid bits 15..8) :");
    for (int bit = 0; section != 2 && bit < std::min(cus, 40); ++bit) {
        std::vector<uint32_t> m = all;
        m[bit / 32] &= ~(1u << (bit % 32));
        hipStream_t s = masked_stream(m);
        const std::set<uint32_t> u = used(s);
        printf(" %d->", bit);
        int miss = 0;
        for (uint32_t v : every) if (!u.count(v)) { printf("%s%u:%02x", miss ? "," : "", v >> 16, v & 0xffu); ++miss; }
        if (!miss) printf("none");
        CK(hipStreamDestroy(s));
    }
    printf("\n");
    if (section == 1) return 0;
    // candidate reservations: k CUs per XCD, assuming bit i belongs to XCC (i % 8) (checked by the listing above)
    const size_t n_img = 65536, bytes = n_img * (size_t)IMG_BYTES;
    char *buf = nullptr;
    CK(hipMalloc((void **)&buf, bytes));
    uint32_t *queue = nullptr;
    CK(hipMalloc((void **)&queue, 64));
    float *table = nullptr, *mout = nullptr;
    CK(hipMalloc((void **)&table, (1 << 20) * 4));
    CK(hipMemset(table, 0, (1 << 20) * 4));
    CK(hipMalloc((void **)&mout, 8192 * 256 * 4));
    CK(hipFuncSetAttribute((const void *)stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 52 * 1024));
    hipEvent_t e0, e1, m0, m1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&m0)); CK(hipEventCreate(&m1));
    auto raster = [&](hipStream_t s, int reps, float &best, float &mean) {
        best = 1e30f; mean = 0;
        for (int r = 0; r <= reps; ++r) {
            CK(hipMemsetAsync(queue, 0, 64, s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(stream_kernel, dim3(cus * 8), dim3(256), 52 * 1024 - 64, s, buf, (int64_t)n_img, queue, (float)r);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            const float ms = elapsed(e0, e1);
            if (r > 0) { best = std::min(best, ms); mean += ms / reps; }
        }
    };
    float best, mean;
    raster(s_all, 5, best, mean);
    printf("write pattern of the raster launch, all CUs: best %.3f mean %.3f ms\n", best, mean);
    for (int k : {1, 2, 4}) {
        std::vector<uint32_t> keep = all, rest(words, 0u);
        for (int j = 0; j < 8 * k; ++j) { keep[j / 32] &= ~(1u << (j % 32)); rest[j / 32] |= 1u << (j % 32); }     // bits 0 .. 8k-1: k per XCC if bit i -> XCC i % 8
        hipStream_t s_keep = masked_stream(keep), s_rest = masked_stream(rest);
        const std::set<uint32_t> ur = used(s_rest);
        std::map<int, int> px;
        for (uint32_t v : ur) px[v >> 16]++;
        printf("reserve bits 0..%d: the reserved stream sees %zu places (", 8 * k - 1, ur.size());
        for (auto &kv : px) printf("XCC%d:%d ", kv.first, kv.second);
        printf("), the raster stream %zu", used(s_keep).size());
        raster(s_keep, 5, best, mean);
        printf(" | raster alone: best %.3f mean %.3f ms", best, mean);
        // the metric stand-in alone on the reserved CUs, then beside the raster launch
        for (int blocks : {1024, 8192}) {
            CK(hipEventRecord(m0, s_rest));
            hipLaunchKernelGGL(metric_kernel, dim3(blocks), dim3(256), 0, s_rest, (const float *)table, mout, 64);
            CK(hipEventRecord(m1, s_rest));
            CK(hipEventSynchronize(m1));
            const float alone = elapsed(m0, m1);
            CK(hipEventRecord(m0, s_all));
            hipLaunchKernelGGL(metric_kernel, dim3(blocks), dim3(256), 0, s_all, (const float *)table, mout, 64);
            CK(hipEventRecord(m1, s_all));
            CK(hipEventSynchronize(m1));
            const float alone_all = elapsed(m0, m1);
            printf(" | metric stand-in %d blocks: %.3f ms on all CUs, %.3f alone on the reserved;", blocks, alone_all, alone);
            // beside the raster launch: with the surplus workgroups of the product's persistent launch (8 per CU launched, 3 resident: the
            // rest WAITS in the dispatcher) and with exactly as many workgroups as fit the allowed CUs (nothing waits)
            for (int grid : {cus * 8, 3 * (cus - 8 * k)}) {
                float rb = 0, mb = 0;
                for (int r = 0; r < 4; ++r) {
                    CK(hipMemsetAsync(queue, 0, 64, s_keep));
                    CK(hipEventRecord(e0, s_keep));
                    hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 52 * 1024 - 64, s_keep, buf, (int64_t)n_img, queue, (float)r);
                    CK(hipEventRecord(e1, s_keep));
                    CK(hipEventRecord(m0, s_rest));
                    hipLaunchKernelGGL(metric_kernel, dim3(blocks), dim3(256), 0, s_rest, (const float *)table, mout, 64);
                    CK(hipEventRecord(m1, s_rest));
                    CK(hipEventSynchronize(e1));
                    CK(hipEventSynchronize(m1));
                    rb += elapsed(e0, e1) / 4; mb += elapsed(m0, m1) / 4;
                }
                printf(" raster grid %d: metric %.3f ms beside the launch, the launch %.3f;", grid, mb, rb);
            }
        }
        printf("\n");
        CK(hipStreamDestroy(s_keep)); CK(hipStreamDestroy(s_rest));
    }
    // for comparison: the same pair without any mask (what round 3 measured: the small kernel crawls, the launch stretches)
    for (int grid : {cus * 8, cus * 3}) {
        hipStream_t s2;
        CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, -1));
        float rb = 0, mb = 0;
        printf("grid %d: ", grid);
        for (int r = 0; r < 4; ++r) {
            CK(hipMemsetAsync(queue, 0, 64, s_all));
            CK(hipEventRecord(e0, s_all));
            hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 52 * 1024 - 64, s_all, buf, (int64_t)n_img, queue, (float)r);
            CK(hipEventRecord(e1, s_all));
            CK(hipEventRecord(m0, s2));
            hipLaunchKernelGGL(metric_kernel, dim3(1024), dim3(256), 0, s2, (const float *)table, mout, 64);
            CK(hipEventRecord(m1, s2));
            CK(hipEventSynchronize(e1));
            CK(hipEventSynchronize(m1));
            rb += elapsed(e0, e1) / 4; mb += elapsed(m0, m1) / 4;
        }
        printf("no masks, second stream of high priority: metric stand-in 1024 blocks %.3f ms beside the launch, the launch %.3f ms\n", mb, rb);
    }
    return 0;
}
