// Output buffers whose PHYSICAL pages are spread out (include/tdship.h: tds_buffer_*, tds_torch_alloc / tds_torch_free).
//
// Why the library has an allocator at all.  The rasteriser is bound by the HBM write stream (51.5 GB per launch at B = 1024 x A = 64 x 256 x 256),
// and on MI355X what a write stream reaches depends on the PHYSICAL pages under the buffer: a 51.5 GB hipMalloc is served at one of three
// rates for as long as it lives -- 7.03 ms per launch, x 16/15 (7.45 ms) or x 8/7 (8.05 ms) -- whatever the virtual address (one
// hipMemCreate handle mapped at thirteen alignments between 2 MiB and 64 GiB: the same time everywhere, so it is neither the address nor
// the PTE fragment size), whatever the store pattern (contiguous eighths per XCD, eighths interleaved in chunks of 6 MB .. 3 GB, a rotated
// column order, a plain front-to-back fill: a slow buffer stays at its 8/7), about one allocation in three.  The ratios read like a load
// imbalance over 16 units of the memory system that a physically contiguous range of this size can fall into and a scattered one cannot:
// 1 GiB contiguous chunks all fill at 5.8 TB/s while the driver's scattered leftovers reach 7.0, and buffers assembled from chunks whose
// physical addresses are spread over TWICE the range were fast in 18 of 18 probe runs and in 99 of 100 ring candidates of 50 fresh bench starts -- one at 15/16 -- (tools/alloc_probe.hip, profiles/r04_alloc_probe.log, r04_fresh_starts.log;
// DESIGN_HISTORY.md section 4).  So a buffer is built from chunks of 8 MiB created alternately with spacer chunks that are released once the buffer is
// mapped (hipMemCreate / hipMemMap): the chunks end up about 16 MiB apart in physical memory and the holes go back to the driver.
//
// The reference allocates its image per call (rendering/cv2.py:52: np.zeros); here the Python host routes the image allocations of HipRenderer
// through a torch memory pool that is served by tds_torch_alloc / tds_torch_free, so that `render_egocentric()` without `out=` gets such a
// buffer too, cached by torch's allocator like any other block.
#include <mutex>
#include <unordered_map>
#include <vector>

#include "tds_common.h"

struct tds_buffer {
    void *ptr = nullptr;
    size_t bytes = 0, reserved = 0, mapped = 0;
    int device = 0;
    int spread = 0;                                 // 1: chunks with spacers, 0: plain hipMalloc
    std::vector<hipMemGenericAllocationHandle_t> handles;
};

namespace {
constexpr size_t CHUNK = (size_t)8 << 20;           // 8 MiB (8 / 64 MiB / 1 GiB chunks were all fast; 8 MiB was the fastest: 7.00 - 7.02 ms)
constexpr size_t SPREAD_MIN = (size_t)256 << 20;    // below this a buffer is one hipMalloc (a launch into it is too short to tell)

void release(tds_buffer *b) {
    if (b->spread) {
        if (b->ptr && b->mapped) (void)hipMemUnmap(b->ptr, b->mapped);
        for (auto h : b->handles) (void)hipMemRelease(h);
        if (b->ptr) (void)hipMemAddressFree(b->ptr, b->reserved);
    } else if (b->ptr) {
        (void)hipFree(b->ptr);
    }
    b->ptr = nullptr;
    b->handles.clear();
}

// -> TDS_OK and a mapped buffer, or an error code with nothing held
int create_spread(tds_buffer *b, size_t bytes, int device) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    // the chunk size and the 2 MiB alignment of the range must be multiples of what the device maps at (2 MiB on MI355X with this driver):
    // asked once per call, so that another ASIC / driver takes the dense path at once instead of after thousands of failed hipMemCreate
    size_t gran = 0;
    hipError_t eg = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (eg != hipSuccess || gran == 0 || CHUNK % gran != 0) {
        (void)hipGetLastError();
        tds::set_error("tds_buffer_create: the device maps memory in units of %zu bytes, which do not divide the %zu-byte chunks", gran, CHUNK);
        return TDS_ELIMIT;
    }
    const size_t align = gran > ((size_t)2 << 20) ? gran : ((size_t)2 << 20);
    const size_t n = (bytes + CHUNK - 1) / CHUNK, total = n * CHUNK;
    b->spread = 1;
    b->reserved = total;
    b->bytes = bytes;
    b->device = device;
    hipError_t e = hipMemAddressReserve(&b->ptr, total, align, nullptr, 0);
    if (e != hipSuccess) { b->ptr = nullptr; tds::set_error("tds_buffer_create: reserving %zu bytes of address space failed: %s", total, hipGetErrorString(e)); return TDS_EHIP; }
    std::vector<hipMemGenericAllocationHandle_t> spacers;
    spacers.reserve(n);
    b->handles.reserve(n);
    bool spacing = true;
    unsigned long long rng = 0x9e3779b97f4a7c15ull ^ (unsigned long long)bytes;
    for (size_t i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, CHUNK, &prop, 0);
        if (e != hipSuccess && !spacers.empty()) {
            // out of memory with the spacers held: give them back and go on without (the rest of the buffer is then as dense as a hipMalloc)
            for (auto sp : spacers) (void)hipMemRelease(sp);
            spacers.clear();
            spacing = false;
            (void)hipGetLastError();
            e = hipMemCreate(&h, CHUNK, &prop, 0);
        }
        if (e != hipSuccess) break;
        b->handles.push_back(h);
        // 0, 1 or 2 spacers after every chunk (one on average), never a regular pattern: with exactly one spacer after every chunk a buffer built in
        // pristine memory sits at a fixed 16 MiB stride, i.e. never on the odd 8 MiB slots -- and the first buffer of a process was the one that
        // came out slow (7.1 - 8.1 ms in four of four starts on one box)
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        for (unsigned k = (unsigned)((rng >> 33) % 3u); spacing && k > 0; --k) {
            hipMemGenericAllocationHandle_t sp;
            if (hipMemCreate(&sp, CHUNK, &prop, 0) == hipSuccess) spacers.push_back(sp);
            else { spacing = false; (void)hipGetLastError(); }
        }
    }
    for (size_t i = 0; e == hipSuccess && i < n; ++i) {
        e = hipMemMap((char *)b->ptr + i * CHUNK, CHUNK, 0, b->handles[i], 0);
        if (e == hipSuccess) b->mapped = (i + 1) * CHUNK;
    }
    for (auto sp : spacers) (void)hipMemRelease(sp);
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(b->ptr, total, &acc, 1);
    }
    if (e != hipSuccess) {
        tds::set_error("tds_buffer_create: %zu bytes in chunks of %zu failed: %s", bytes, CHUNK, hipGetErrorString(e));
        (void)hipGetLastError();
        release(b);
        return e == hipErrorOutOfMemory ? TDS_ENOMEM : TDS_EHIP;
    }
    return TDS_OK;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

TDS_EXPORT int tds_buffer_create(int64_t bytes, int device, int flags, tds_buffer_t **out) {
    TDS_CHECK_ARG(out, "tds_buffer_create: null output");
    *out = nullptr;
    TDS_CHECK_ARG(bytes > 0, "tds_buffer_create: %lld bytes", (long long)bytes);
    TDS_CHECK_ARG((flags & ~TDS_BUFFER_DENSE) == 0, "tds_buffer_create: unknown flags %#x", flags);
    DeviceGuard guard(device);
    if (!guard.ok) { tds::set_error("tds_buffer_create: no device %d", device); return TDS_EINVAL; }
    tds_buffer *b = new tds_buffer();
    int rc = TDS_OK;
    if ((flags & TDS_BUFFER_DENSE) || (size_t)bytes < SPREAD_MIN) {
        b->bytes = (size_t)bytes;
        b->device = device;
        hipError_t e = hipMalloc(&b->ptr, (size_t)bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            tds::set_error("tds_buffer_create: hipMalloc of %lld bytes failed: %s", (long long)bytes, hipGetErrorString(e));
            rc = e == hipErrorOutOfMemory ? TDS_ENOMEM : TDS_EHIP;
        }
    } else {
        rc = create_spread(b, (size_t)bytes, device);
        if (rc == TDS_ELIMIT) {
            // no usable mapping granularity: one hipMalloc (the placement is then the driver's)
            *b = tds_buffer();
            b->bytes = (size_t)bytes;
            b->device = device;
            hipError_t e = hipMalloc(&b->ptr, (size_t)bytes);
            rc = e == hipSuccess ? TDS_OK : (e == hipErrorOutOfMemory ? TDS_ENOMEM : TDS_EHIP);
            if (e != hipSuccess) { (void)hipGetLastError(); tds::set_error("tds_buffer_create: hipMalloc of %lld bytes failed: %s", (long long)bytes, hipGetErrorString(e)); }
        }
    }
    if (rc != TDS_OK) { delete b; return rc; }
    *out = b;
    return TDS_OK;
}

TDS_EXPORT void *tds_buffer_ptr(const tds_buffer_t *buf) { return buf ? buf->ptr : nullptr; }

TDS_EXPORT int tds_buffer_info(const tds_buffer_t *buf, int64_t *bytes, int64_t *chunks, int *spread) {
    TDS_CHECK_ARG(buf, "tds_buffer_info: null buffer");
    if (bytes) *bytes = (int64_t)buf->bytes;
    if (chunks) *chunks = (int64_t)buf->handles.size();
    if (spread) *spread = buf->spread;
    return TDS_OK;
}

TDS_EXPORT int tds_buffer_destroy(tds_buffer_t *buf) {
    if (!buf) return TDS_OK;
    DeviceGuard guard(buf->device);
    release(buf);
    delete buf;
    return TDS_OK;
}

// ---- the two entry points a torch.cuda.memory.CUDAPluggableAllocator binds (signatures fixed by torch: plain C types) ---------------------
namespace {
std::mutex g_mu;
std::unordered_map<void *, tds_buffer *> g_live;
}  // namespace

TDS_EXPORT void *tds_torch_alloc(size_t size, int device, void *stream) {
    (void)stream;
    tds_buffer_t *b = nullptr;
    if (tds_buffer_create((int64_t)(size ? size : 1), device, 0, &b) != TDS_OK &&
        tds_buffer_create((int64_t)(size ? size : 1), device, TDS_BUFFER_DENSE, &b) != TDS_OK)       // the virtual-memory calls failed: one hipMalloc
        return nullptr;                                                                              // torch reports the failure (out of memory)
    std::lock_guard<std::mutex> lock(g_mu);
    g_live[b->ptr] = b;
    return b->ptr;
}

TDS_EXPORT void tds_torch_free(void *ptr, size_t size, int device, void *stream) {
    (void)size; (void)device; (void)stream;
    tds_buffer *b = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_live.find(ptr);
        if (it == g_live.end()) return;
        b = it->second;
        g_live.erase(it);
    }
    (void)tds_buffer_destroy(b);
}

// ---- streams confined to a part of the CUs (include/tdship.h) ------------------------------------------------------------------------------
TDS_EXPORT int tds_device_cu_count(int device, int *cus) {
    TDS_CHECK_ARG(cus, "tds_device_cu_count: null output");
    int n = 0;
    TDS_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
    *cus = n;
    return TDS_OK;
}

TDS_EXPORT int tds_stream_create(int device, const uint32_t *cu_mask, int n_words, void **stream) {
    TDS_CHECK_ARG(stream, "tds_stream_create: null output");
    *stream = nullptr;
    TDS_CHECK_ARG(n_words >= 0 && n_words <= 64 && (cu_mask || n_words == 0), "tds_stream_create: bad mask");
    DeviceGuard guard(device);
    if (!guard.ok) { tds::set_error("tds_stream_create: no device %d", device); return TDS_EINVAL; }
    hipStream_t s = nullptr;
    if (n_words == 0) {
        TDS_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    } else {
        bool any = false;
        for (int i = 0; i < n_words; ++i) any = any || cu_mask[i] != 0u;
        TDS_CHECK_ARG(any, "tds_stream_create: the mask selects no CU");
        TDS_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask));
    }
    *stream = (void *)s;
    return TDS_OK;
}

namespace {
// every workgroup reports the CU it ran on after keeping it busy for a while (so that the dispatcher has to use them all)
__global__ void places_kernel(uint32_t *out, int spin) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);          // HW_ID bits 15:0 (CU_ID 11:8, SH_ID 12, SE_ID 15:13)
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;              // XCC_ID
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xffu);
}
}  // namespace

TDS_EXPORT int tds_stream_places(void *stream, uint32_t *places, int n) {
    TDS_CHECK_ARG(places && n > 0 && n <= (1 << 20), "tds_stream_places: bad arguments");
    hipLaunchKernelGGL(places_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, places, 200);
    TDS_LAUNCH_CHECK("places_kernel");
    return TDS_OK;
}

TDS_EXPORT int tds_stream_destroy(int device, void *stream) {
    if (!stream) return TDS_OK;
    DeviceGuard guard(device);
    TDS_HIP(hipStreamDestroy((hipStream_t)stream));
    return TDS_OK;
}
