// What is a "slow" output allocation?  (DESIGN_HISTORY.md section 4)
//
// A stand-alone reproduction of the WRITE pattern of the headline raster launch -- no rasterisation, no torch: 768 resident workgroups
// (three per CU, held there by 52 KiB of LDS each), every XCD streams into its own contiguous eighth of a 51.5 GB buffer, a workgroup
// writes one 786 432-byte image after the other in the store order of write_out_bits (1 KiB per wave instruction, 96 instructions per
// item, non-temporal) -- timed into buffers obtained in different ways:
//     malloc        hipMalloc
//     contiguous    hipExtMallocWithFlags(hipDeviceMallocContiguous)
//     uncached      hipExtMallocWithFlags(hipDeviceMallocUncached)
//     vmm:A:C       hipMemAddressReserve(alignment 2^A) + hipMemCreate in chunks of 2^C bytes (C = 0: one handle) + hipMemMap
// and with variants of the pattern (--pattern): 0 as the kernel, 1 = every workgroup starts its image at its own column (the low address
// bits of concurrently written lines differ), 2 = images dealt round-robin to the XCDs, 3 = plain front-to-back fill by all workgroups.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/alloc_probe tools/alloc_probe.hip
//   tools/_build/alloc_probe [--images 65536] [--hold 4] [--reps 3] method[,method...] [pattern,pattern...]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                                        \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)

constexpr int IMG_BYTES = 3 * 256 * 256 * 4;

typedef float vf4 __attribute__((ext_vector_type(4)));

// pattern bits: 1 rotate the start column per workgroup, 2 round-robin images over XCDs, 4 front-to-back fill
__global__ void __launch_bounds__(256, 3) stream_kernel(char *out, int64_t n_img, int pattern, uint32_t *queue, float value) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (pattern & 4) {
        // all workgroups together, front to back: what fill_ does
        const int64_t n16 = n_img * (IMG_BYTES / 16);
        const vf4 v = {value, value, value, value};
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n16; i += (int64_t)gridDim.x * 256) __builtin_nontemporal_store(v, (vf4 *)out + i);
        return;
    }
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
                const int64_t ck = pattern >> 8;                 // > 0: the XCDs' regions interleaved in chunks of this many images
                if ((int64_t)i < per) img = ck > 0 ? ((int64_t)i / ck * 8 + q) * ck + (int64_t)i % ck : ((pattern & 2) ? (int64_t)i * 8 + q : q * per + (int64_t)i);
            }
            s_img = img;
        }
        __syncthreads();
        const int64_t img = s_img;
        if (img < 0) break;
        lds[tid] = (uint32_t)img;                                     // the LDS is there for the occupancy only
        char *o = out + img * IMG_BYTES;
        const int rot = (pattern & 1) ? (int)((img * 37) & 255) : 0;
        const vf4 v = {value, value + (float)lds[tid ^ 1] * 0.0f, value, value};
        for (int pass = 0; pass < 2; ++pass) {
            const int xw = wave + 4 * pass;
            for (int ph = 0; ph < 8; ++ph)
                for (int m = 0; m < 4; ++m) {
                    const int col = ((xw * 32 + ph + 8 * m) + rot) & 255;
                    const uint32_t off = (uint32_t)(col * 256 + lane * 4) * 4u;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) __builtin_nontemporal_store(v, (vf4 *)(o + (size_t)ch * 262144 + off));
                }
        }
    }
}

// the front-to-back fill, `rounds` times over the same range inside ONE launch: a rate for ranges too small to time launch by launch
__global__ void __launch_bounds__(256) refill_kernel(vf4 *out, int64_t n16, int rounds) {
    for (int r = 0; r < rounds; ++r) {
        const float f = (float)r;
        const vf4 v = {f, f, f, f};
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) __builtin_nontemporal_store(v, out + i);
    }
}

struct Buf {
    void *ptr = nullptr;
    size_t bytes = 0, mapped = 0;
    int kind = 0;                                  // 0 hipMalloc / hipExtMalloc, 1 vmm
    std::vector<hipMemGenericAllocationHandle_t> handles;
    size_t reserved = 0;
};

static bool alloc_buf(const std::string &method, size_t bytes, int dev, Buf &b) {
    b = Buf();
    b.bytes = bytes;
    if (method == "malloc") return hipMalloc(&b.ptr, bytes) == hipSuccess;
    if (method == "contiguous") return hipExtMallocWithFlags(&b.ptr, bytes, hipDeviceMallocContiguous) == hipSuccess;
    if (method == "uncached") return hipExtMallocWithFlags(&b.ptr, bytes, hipDeviceMallocUncached) == hipSuccess;
    if (method == "finegrained") return hipExtMallocWithFlags(&b.ptr, bytes, hipDeviceMallocFinegrained) == hipSuccess;
    if (method.rfind("vmms", 0) == 0) {
        // chunks of 2^C bytes whose PHYSICAL addresses are spread out: between two chunks of the buffer K - 1 spacer chunks are created
        // (and released once the buffer is mapped); S = 1 also maps the chunks in a shuffled order
        int c = 26, K = 2, S = 0;
        sscanf(method.c_str(), "vmms:%d:%d:%d", &c, &K, &S);
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        const size_t chunk = (size_t)1 << c;
        const size_t n = (bytes + chunk - 1) / chunk, total = n * chunk;
        b.kind = 1;
        b.reserved = total;
        if (hipMemAddressReserve(&b.ptr, total, 0, nullptr, 0) != hipSuccess) { b.ptr = nullptr; return false; }
        std::vector<hipMemGenericAllocationHandle_t> spacers;
        bool ok = true;
        for (size_t i = 0; i < n && ok; ++i) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
            b.handles.push_back(h);
            for (int k = 1; k < K; ++k) {
                hipMemGenericAllocationHandle_t sp;
                if (hipMemCreate(&sp, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
                spacers.push_back(sp);
            }
        }
        std::vector<size_t> slot(b.handles.size());
        for (size_t i = 0; i < slot.size(); ++i) slot[i] = i;
        if (S) { uint64_t r = 88172645463325252ull; for (size_t i = slot.size(); i > 1; --i) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; std::swap(slot[i - 1], slot[r % i]); } }
        for (size_t i = 0; i < b.handles.size() && ok; ++i) {
            if (hipMemMap((char *)b.ptr + slot[i] * chunk, chunk, 0, b.handles[i], 0) != hipSuccess) ok = false;
        }
        b.mapped = ok ? total : 0;
        for (auto sp : spacers) (void)hipMemRelease(sp);
        if (!ok) return false;
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        return hipMemSetAccess(b.ptr, total, &acc, 1) == hipSuccess;
    }
    if (method.rfind("vmm", 0) == 0) {
        int a = 21, c = 0;
        sscanf(method.c_str(), "vmm:%d:%d", &a, &c);
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) gran = 2u << 20;
        const size_t chunk = c == 0 ? 0 : ((size_t)1 << c);
        const size_t total = (bytes + gran - 1) / gran * gran;
        b.kind = 1;
        b.reserved = total;
        if (hipMemAddressReserve(&b.ptr, total, (size_t)1 << a, nullptr, 0) != hipSuccess) { b.ptr = nullptr; return false; }
        size_t off = 0;
        while (off < total) {
            const size_t n = chunk == 0 ? total : (total - off < chunk ? total - off : chunk);
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, n, &prop, 0) != hipSuccess) return false;
            b.handles.push_back(h);
            if (hipMemMap((char *)b.ptr + off, n, 0, h, 0) != hipSuccess) return false;
            off += n;
            b.mapped = off;
        }
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = dev;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        return hipMemSetAccess(b.ptr, total, &acc, 1) == hipSuccess;
    }
    fprintf(stderr, "unknown method %s\n", method.c_str());
    exit(2);
}

static void free_buf(Buf &b) {
    if (b.kind == 0) { if (b.ptr) (void)hipFree(b.ptr); }
    else {
        if (b.ptr && b.mapped) (void)hipMemUnmap(b.ptr, b.mapped);
        for (auto h : b.handles) (void)hipMemRelease(h);
        if (b.ptr) (void)hipMemAddressFree(b.ptr, b.reserved);
    }
    b = Buf();
}

int main(int argc, char **argv) {
    int64_t n_img = 65536;
    int hold = 4, reps = 3;
    int64_t sub = 0;
    std::vector<std::string> methods, pats;
    auto split = [](const char *s) { std::vector<std::string> v; std::string cur; for (; *s; ++s) { if (*s == ',') { v.push_back(cur); cur.clear(); } else cur += *s; } v.push_back(cur); return v; };
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--images")) n_img = atoll(argv[++i]);
        else if (!strcmp(argv[i], "--hold")) hold = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--sub")) sub = atoll(argv[++i]);            // also: the fill pattern over sub-ranges of this many images
        else if (methods.empty()) methods = split(argv[i]);
        else pats = split(argv[i]);
    }
    if (methods.empty()) methods = {"malloc"};
    if (pats.empty()) pats = {"0"};
    int dev = 0, cus = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    const size_t bytes = (size_t)n_img * IMG_BYTES;
    printf("device: %d CUs, %.1f of %.1f GB free; buffer %.2f GB (%lld images), %d held at a time, %d timed launches each (best / mean)\n", cus, free_b / 1e9, total_b / 1e9,
           bytes / 1e9, (long long)n_img, hold, reps);
    uint32_t *queue = nullptr;
    CK(hipMalloc((void **)&queue, 64));
    CK(hipFuncSetAttribute((const void *)stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 52 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](void *ptr, int pattern, float &first, float &best, float &mean) {
        best = 1e30f; mean = 0.f;
        for (int r = 0; r <= reps; ++r) {
            CK(hipMemsetAsync(queue, 0, 64, 0));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(stream_kernel, dim3((pattern & 4) ? cus * 12 : cus * 8), dim3(256), (pattern & 4) ? 1024 : 52 * 1024 - 64, 0, (char *)ptr, n_img, pattern, queue, (float)r);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r == 0) first = ms; else { best = ms < best ? ms : best; mean += ms / reps; }
        }
    };
    for (const auto &m : methods) {
        if (m.rfind("remap", 0) == 0) {
            // ONE physical allocation (hipMemCreate), mapped again and again at virtual addresses base + O * 2 MiB, base aligned to 2^A:
            // the same physical pages, only the alignment of (virtual - physical) -- what bounds the PTE fragment size -- changes
            int a = 36;
            sscanf(m.c_str(), "remap:%d", &a);
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = dev;
            size_t gran = 0;
            if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) gran = 2u << 20;
            size_t gmin = 0;
            (void)hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
            const size_t total = (bytes + gran - 1) / gran * gran;
            const size_t span = total + ((size_t)1 << a) + ((size_t)2048 << 21);
            void *res = nullptr;
            CK(hipMemAddressReserve(&res, span, 0, nullptr, 0));
            const uintptr_t base = ((uintptr_t)res + (((uintptr_t)1 << a) - 1)) & ~(((uintptr_t)1 << a) - 1);
            printf("remap: granularity recommended %zu minimum %zu; reservation %p, base %#llx (2^%d-aligned)\n", gran, gmin, res, (unsigned long long)base, a);
            std::vector<hipMemGenericAllocationHandle_t> hs(hold);
            for (int i = 0; i < hold; ++i) CK(hipMemCreate(&hs[i], total, &prop, 0));
            hipMemAccessDesc acc = {};
            acc.location.type = hipMemLocationTypeDevice;
            acc.location.id = dev;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            const int offs[] = {0, 1, 2, 3, 4, 6, 8, 16, 32, 64, 128, 256, 512, 1024, 0, 1};
            for (int i = 0; i < hold; ++i)
                for (int o : offs) {
                    void *va = (void *)(base + ((uintptr_t)o << 21));
                    CK(hipMemMap(va, total, 0, hs[i], 0));
                    CK(hipMemSetAccess(va, total, &acc, 1));
                    printf("handle #%d at base + %4d x 2 MiB (va 2^%d-aligned)", i, o, o ? __builtin_ctzll((uintptr_t)va) : a);
                    for (const auto &ps : pats) {
                        float first, best, mean;
                        run(va, atoi(ps.c_str()), first, best, mean);
                        printf(" | pat %s first %.2f best %.3f mean %.3f ms", ps.c_str(), first, best, mean);
                    }
                    printf("\n");
                    fflush(stdout);
                    CK(hipMemUnmap(va, total));
                }
            for (int i = 0; i < hold; ++i) CK(hipMemRelease(hs[i]));
            CK(hipMemAddressFree(res, span));
            continue;
        }
        if (m.rfind("assemble", 0) == 0) {
            // Can a fast buffer be BUILT?  N physical pieces of one eighth of the buffer each (what one XCD writes), every piece timed by itself
            // (the pattern of the launch over that piece alone), then whole buffers mapped out of the 8 fastest / the 8 slowest / a mix
            int N = 40;
            sscanf(m.c_str(), "assemble:%d", &N);
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = dev;
            const size_t piece = bytes / 8;                            // 6.44 GB: a multiple of 2 MiB for 65536 images
            if (piece % ((size_t)2 << 20)) { printf("assemble: piece size is no multiple of 2 MiB\n"); continue; }
            hipMemAccessDesc acc = {};
            acc.location.type = hipMemLocationTypeDevice;
            acc.location.id = dev;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            std::vector<hipMemGenericAllocationHandle_t> hs;
            std::vector<float> ms_piece;
            void *va1 = nullptr;
            CK(hipMemAddressReserve(&va1, piece, 0, nullptr, 0));
            for (int i = 0; i < N; ++i) {
                hipMemGenericAllocationHandle_t h;
                if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) break;
                hs.push_back(h);
                CK(hipMemMap(va1, piece, 0, h, 0));
                CK(hipMemSetAccess(va1, piece, &acc, 1));
                float first, best, mean;
                const int64_t save = n_img;
                n_img = save / 8;
                run(va1, 0, first, best, mean);
                n_img = save;
                ms_piece.push_back(best);
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(va1, piece));
            }
            CK(hipMemAddressFree(va1, piece));
            const int got = (int)hs.size();
            printf("assemble: %d pieces of %.2f GB; the launch pattern over each piece alone, ms:", got, piece / 1e9);
            for (int i = 0; i < got; ++i) printf(" %.3f", ms_piece[i]);
            printf("\n");
            std::vector<int> order(got);
            for (int i = 0; i < got; ++i) order[i] = i;
            std::sort(order.begin(), order.end(), [&](int x, int y) { return ms_piece[x] < ms_piece[y]; });
            auto build = [&](const char *label, std::vector<int> idx) {
                void *va = nullptr;
                CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
                for (int k = 0; k < 8; ++k) CK(hipMemMap((char *)va + (size_t)k * piece, piece, 0, hs[idx[k]], 0));
                CK(hipMemSetAccess(va, bytes, &acc, 1));
                printf("%-28s pieces", label);
                for (int k = 0; k < 8; ++k) printf(" %d(%.2f)", idx[k], ms_piece[idx[k]]);
                for (const auto &ps : pats) {
                    float first, best, mean;
                    run(va, atoi(ps.c_str()), first, best, mean);
                    printf(" | pat %s best %.3f mean %.3f ms", ps.c_str(), best, mean);
                }
                printf("\n");
                fflush(stdout);
                CK(hipDeviceSynchronize());
                CK(hipMemUnmap(va, bytes));
                CK(hipMemAddressFree(va, bytes));
            };
            if (got >= 24) {
                build("8 fastest", std::vector<int>(order.begin(), order.begin() + 8));
                build("next 8 fastest", std::vector<int>(order.begin() + 8, order.begin() + 16));
                build("8 slowest", std::vector<int>(order.end() - 8, order.end()));
                std::vector<int> mix;
                for (int k = 0; k < 4; ++k) { mix.push_back(order[k]); mix.push_back(order[got - 1 - k]); }
                build("4 fastest + 4 slowest, mixed", mix);
                std::vector<int> rev(order.begin(), order.begin() + 8);
                std::reverse(rev.begin(), rev.end());
                build("8 fastest, reversed", rev);
                build("first 8 allocated", {0, 1, 2, 3, 4, 5, 6, 7});
                build("8 fastest again", std::vector<int>(order.begin(), order.begin() + 8));
            }
            for (auto h : hs) CK(hipMemRelease(h));
            continue;
        }
        if (m.rfind("chunks", 0) == 0) {
            // a map of the device memory: as many hipMalloc'ed chunks of 2^C bytes as fit (all held), each filled `R` times inside one launch
            int c = 30, R = 16;
            sscanf(m.c_str(), "chunks:%d:%d", &c, &R);
            const size_t cb = (size_t)1 << c;
            CK(hipMemGetInfo(&free_b, &total_b));
            const int n = (int)((free_b - ((size_t)6 << 30)) / cb);
            std::vector<void *> ps(n, nullptr);
            int got = 0;
            for (; got < n; ++got) if (hipMalloc(&ps[got], cb) != hipSuccess) break;
            printf("chunks: %d x %.3f GB held, each filled %d times in one launch; TB/s x 100, in allocation order:\n", got, cb / 1e9, R);
            for (int pass = 0; pass < 2; ++pass) {
                for (int i = 0; i < got; ++i) {
                    CK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(refill_kernel, dim3(cus * 12), dim3(256), 0, 0, (vf4 *)ps[i], (int64_t)(cb / 16), R);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("%d%s", (int)((double)cb * R / (ms * 1e-3) / 1e10 + 0.5), (i % 32 == 31 || i == got - 1) ? "\n" : " ");
                }
                printf("-- pass %d done; virtual addresses of the first chunks: %p %p %p\n", pass, ps[0], got > 1 ? ps[1] : nullptr, got > 2 ? ps[2] : nullptr);
            }
            fflush(stdout);
            for (int i = 0; i < got; ++i) (void)hipFree(ps[i]);
            continue;
        }
        std::vector<Buf> bufs(hold);
        int got = 0;
        for (int i = 0; i < hold; ++i) {
            if (!alloc_buf(m, bytes, dev, bufs[i])) { printf("%-14s allocation %d FAILED (%s)\n", m.c_str(), i, hipGetErrorString(hipGetLastError())); free_buf(bufs[i]); break; }
            ++got;
        }
        for (int i = 0; i < got; ++i) {
            const uintptr_t p = (uintptr_t)bufs[i].ptr;
            printf("%-14s #%d va %#014llx (2^%d-aligned)", m.c_str(), i, (unsigned long long)p, __builtin_ctzll(p));
            for (const auto &ps : pats) {
                float first, best, mean;
                run(bufs[i].ptr, atoi(ps.c_str()), first, best, mean);
                printf(" | pat %s first %.2f best %.3f mean %.3f ms = %.2f TB/s", ps.c_str(), first, best, mean, bytes / (best * 1e-3) / 1e12);
            }
            printf("\n");
            if (sub > 0) {
                // where in the buffer is the time lost?  the front-to-back fill over consecutive sub-ranges, 8 launches each, TB/s
                printf("    fill over sub-ranges of %lld images (%.2f GB), TB/s:", (long long)sub, sub * (double)IMG_BYTES / 1e9);
                for (int64_t lo = 0; lo + sub <= n_img; lo += sub) {
                    char *p0 = (char *)bufs[i].ptr + lo * IMG_BYTES;
                    hipLaunchKernelGGL(stream_kernel, dim3(cus * 12), dim3(256), 1024, 0, p0, sub, 4, queue, 1.0f);
                    CK(hipEventRecord(e0, 0));
                    for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(stream_kernel, dim3(cus * 12), dim3(256), 1024, 0, p0, sub, 4, queue, (float)r);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    printf(" %.2f", 8.0 * sub * IMG_BYTES / (ms * 1e-3) / 1e12);
                }
                printf("\n");
            }
            fflush(stdout);
        }
        for (int i = 0; i < got; ++i) free_buf(bufs[i]);
    }
    return 0;
}
