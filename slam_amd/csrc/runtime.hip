// runtime.hip -- device plumbing behind the C-ABI (memory, streams, events).
#include <algorithm>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.hpp"

namespace slam {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    (void)hipGetLastError();
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? SLAM_E_NO_DEVICE : SLAM_E_HIP;
}

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no usable HIP device (hipGetDeviceCount: %s, count %d): this library has no CPU path",
                  hipGetErrorString(e), n);
        return SLAM_E_NO_DEVICE;
    }
    return SLAM_OK;
}

namespace {
struct PoolBlock {
    void  *p;
    size_t cap;
    int    dev;
};
std::mutex                         g_pool_mu;
std::vector<PoolBlock>            *g_pool_free = nullptr;   // never destroyed: the HIP runtime may be gone at exit
std::unordered_map<void *, PoolBlock> *g_pool_live = nullptr;
size_t                             g_pool_cached = 0;
constexpr size_t kPoolMaxCached = 512u << 20, kPoolMaxBlocks = 64;

size_t pool_round(size_t bytes)
{
    size_t c = 256;
    while (c < bytes) c += c >= 4096 ? c / 4 : c; // doubling up to 4 KB, then steps of a quarter
    return (c + 255) & ~(size_t)255;
}
} // namespace

void *pool_alloc(size_t bytes)
{
    const size_t want = pool_round(bytes ? bytes : 1);
    int          dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        set_error("pool_alloc: no current HIP device");
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool_free) {
            g_pool_free = new std::vector<PoolBlock>();
            g_pool_live = new std::unordered_map<void *, PoolBlock>();
        }
        int best = -1;
        for (int i = 0; i < (int)g_pool_free->size(); ++i) {
            const PoolBlock &b = (*g_pool_free)[i];
            if (b.dev == dev && b.cap >= want && b.cap <= 2 * want && (best < 0 || b.cap < (*g_pool_free)[best].cap)) best = i;
        }
        if (best >= 0) {
            const PoolBlock b = (*g_pool_free)[best];
            g_pool_free->erase(g_pool_free->begin() + best);
            g_pool_cached -= b.cap;
            (*g_pool_live)[b.p] = b;
            return b.p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { // give the cached blocks back and try once more
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("pool_alloc: hipMalloc of %zu bytes failed (%s)", want, hipGetErrorString(e));
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    (*g_pool_live)[p] = PoolBlock{p, want, dev};
    return p;
}

void pool_free(void *p)
{
    if (!p) return;
    PoolBlock b{p, 0, 0};
    bool      keep = false;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool_live) return;
        auto it = g_pool_live->find(p);
        if (it == g_pool_live->end()) return; // not ours
        b = it->second;
        g_pool_live->erase(it);
        keep = g_pool_free->size() < kPoolMaxBlocks && g_pool_cached + b.cap <= kPoolMaxCached;
        if (keep) {
            g_pool_free->push_back(b);
            g_pool_cached += b.cap;
        }
    }
    if (!keep) (void)hipFree(b.p);
}

void pool_trim()
{
    std::vector<PoolBlock> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool_free) return;
        drop.swap(*g_pool_free);
        g_pool_cached = 0;
    }
    for (const PoolBlock &b : drop) (void)hipFree(b.p);
}

hipStream_t build_stream()
{
    static std::mutex  mu;
    static hipStream_t streams[64] = {};
    int                dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(mu);
    if (!streams[dev]) {
        int least = 0, greatest = 0; // hipDeviceGetStreamPriorityRange(&least, &greatest): greatest is the numerically LOWER one
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
            hipStreamCreateWithPriority(&streams[dev], hipStreamNonBlocking, greatest) != hipSuccess) {
            (void)hipGetLastError();
            streams[dev] = nullptr;
        }
    }
    return streams[dev];
}

// a small pinned host buffer per host thread for the build's read-back (allocated once, never freed: the
// runtime may be gone when thread-local destructors run)
void *pinned_scratch(size_t bytes)
{
    static thread_local void  *p = nullptr;
    static thread_local size_t cap = 0;
    if (bytes > cap) {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(bytes, 8192);
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            return nullptr;
        }
        cap = want;
    }
    return p;
}

namespace {
constexpr size_t      kPinnedBlock = 16384;
std::mutex            g_pinned_mu;
std::vector<void *>  *g_pinned_free = nullptr; // never destroyed (see the pool)
} // namespace

void *pinned_block_get(size_t bytes)
{
    if (bytes > kPinnedBlock) return nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        if (g_pinned_free && !g_pinned_free->empty()) {
            void *p = g_pinned_free->back();
            g_pinned_free->pop_back();
            return p;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, kPinnedBlock, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void pinned_block_put(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    if (!g_pinned_free) g_pinned_free = new std::vector<void *>();
    g_pinned_free->push_back(p);
}

} // namespace slam

using namespace slam;

extern "C" {

const char *slam_last_error(void) { return g_err; }
const char *slam_version(void) { return "slam_mi355x 0.1 (gfx950)"; }

int slam_device_count(int *n)
{
    SLAM_REQUIRE(n, SLAM_E_INVALID, "slam_device_count: null out pointer");
    *n = 0;
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *n = 0;
    }
    return SLAM_OK;
}

int slam_set_device(int ordinal)
{
    SLAM_TRY(require_device());
    SLAM_HIP(hipSetDevice(ordinal));
    return SLAM_OK;
}

int slam_device_info(char *name, int name_len, int *compute_units, size_t *hbm_bytes)
{
    SLAM_TRY(require_device());
    int dev = 0;
    SLAM_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    SLAM_HIP(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    }
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return SLAM_OK;
}

int slam_malloc(void **dptr, size_t bytes)
{
    SLAM_REQUIRE(dptr, SLAM_E_INVALID, "slam_malloc: null out pointer");
    SLAM_TRY(require_device());
    *dptr = nullptr;
    SLAM_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return SLAM_OK;
}

int slam_free(void *dptr)
{
    if (!dptr) return SLAM_OK;
    SLAM_HIP(hipFree(dptr));
    return SLAM_OK;
}

int slam_memset(void *dptr, int value, size_t bytes, slam_stream_t stream)
{
    SLAM_TRY(require_device());
    SLAM_HIP(hipMemsetAsync(dptr, value, bytes, as_stream(stream)));
    return SLAM_OK;
}

int slam_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, slam_stream_t stream)
{
    SLAM_TRY(require_device());
    if (!bytes) return SLAM_OK;
    SLAM_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    SLAM_HIP(hipStreamSynchronize(as_stream(stream)));
    return SLAM_OK;
}

int slam_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, slam_stream_t stream)
{
    SLAM_TRY(require_device());
    if (!bytes) return SLAM_OK;
    SLAM_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    SLAM_HIP(hipStreamSynchronize(as_stream(stream)));
    return SLAM_OK;
}

int slam_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, slam_stream_t stream)
{
    SLAM_TRY(require_device());
    if (!bytes) return SLAM_OK;
    SLAM_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return SLAM_OK;
}

int slam_host_alloc(void **hptr, size_t bytes)
{
    SLAM_REQUIRE(hptr, SLAM_E_INVALID, "slam_host_alloc: null out pointer");
    SLAM_TRY(require_device());
    *hptr = nullptr;
    SLAM_HIP(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return SLAM_OK;
}

int slam_host_is_pinned(const void *hptr)
{
    if (!hptr || require_device() != SLAM_OK) return 0;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, hptr) != hipSuccess) {
        (void)hipGetLastError(); // (an ordinary host pointer is "invalid value" to the runtime)
        return 0;
    }
    return a.type == hipMemoryTypeHost ? 1 : 0;
}

int slam_host_free(void *hptr)
{
    if (!hptr) return SLAM_OK;
    SLAM_HIP(hipHostFree(hptr));
    return SLAM_OK;
}

int slam_memcpy_h2d_async(void *dst_dev, const void *src_pinned, size_t bytes, slam_stream_t stream)
{
    SLAM_TRY(require_device());
    if (!bytes) return SLAM_OK;
    SLAM_HIP(hipMemcpyAsync(dst_dev, src_pinned, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return SLAM_OK;
}

int slam_memcpy_d2h_async(void *dst_pinned, const void *src_dev, size_t bytes, slam_stream_t stream)
{
    SLAM_TRY(require_device());
    if (!bytes) return SLAM_OK;
    SLAM_HIP(hipMemcpyAsync(dst_pinned, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return SLAM_OK;
}

int slam_stream_wait_event(slam_stream_t stream, slam_event_t ev)
{
    SLAM_REQUIRE(ev, SLAM_E_INVALID, "slam_stream_wait_event: null event");
    SLAM_HIP(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)ev, 0));
    return SLAM_OK;
}

int slam_stream_create(slam_stream_t *stream)
{
    SLAM_REQUIRE(stream, SLAM_E_INVALID, "slam_stream_create: null out pointer");
    SLAM_TRY(require_device());
    hipStream_t s;
    SLAM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (slam_stream_t)s;
    return SLAM_OK;
}

int slam_stream_create_with_priority(slam_stream_t *stream, int priority)
{
    SLAM_REQUIRE(stream, SLAM_E_INVALID, "slam_stream_create_with_priority: null out pointer");
    SLAM_TRY(require_device());
    int least = 0, greatest = 0;
    SLAM_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int   p = priority > 0 ? greatest : (priority < 0 ? least : (least + greatest) / 2);
    hipStream_t s;
    SLAM_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, p));
    *stream = (slam_stream_t)s;
    return SLAM_OK;
}

int slam_stream_create_reserving_cus(slam_stream_t *stream, int reserve_per_xcd)
{
    SLAM_REQUIRE(stream && reserve_per_xcd >= 0, SLAM_E_INVALID, "slam_stream_create_reserving_cus: bad arguments");
    // (reserve_per_xcd = 0 is not slam_stream_create: the mask names every CU, and what is left of a masked stream is its own
    // hardware queue -- this runtime deals unmasked streams over a few shared ones)
    SLAM_TRY(require_device());
    int dev = 0, n_cu = 0;
    SLAM_HIP(hipGetDevice(&dev));
    SLAM_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    constexpr int kXcd = 8; // MI355X: bit i of a CU mask is a CU of XCD i % 8 (measured: tools/exp/cumask.hip)
    if (reserve_per_xcd > 0) {
        // which bit is which XCD was measured on ONE layout: an unpartitioned gfx950 of 256 CUs (SPX).  A partition (CPX: 32 CUs of one
        // XCD) or another part would pass a divisibility check and silently keep the wrong CUs free
        hipDeviceProp_t prop;
        SLAM_HIP(hipGetDeviceProperties(&prop, dev));
        SLAM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0 && n_cu == 256, SLAM_E_UNSUPPORTED,
                     "slam_stream_create_reserving_cus: the CU-mask layout is known for an unpartitioned gfx950 (256 CUs, 8 XCDs) only; this is %s with %d CUs",
                     prop.gcnArchName, n_cu);
    }
    SLAM_REQUIRE(n_cu > 0 && n_cu % kXcd == 0 && reserve_per_xcd < n_cu / kXcd, SLAM_E_INVALID,
                 "slam_stream_create_reserving_cus: cannot keep %d CUs per XCD free on a device of %d CUs", reserve_per_xcd, n_cu);
    const int             words = (n_cu + 31) / 32, off = kXcd * reserve_per_xcd;
    std::vector<uint32_t> mask((size_t)words, 0u);
    for (int i = off; i < n_cu; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
    hipStream_t s;
    SLAM_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask.data()));
    *stream = (slam_stream_t)s;
    return SLAM_OK;
}

int slam_stream_destroy(slam_stream_t stream)
{
    if (!stream) return SLAM_OK;
    SLAM_HIP(hipStreamDestroy(as_stream(stream)));
    return SLAM_OK;
}

int slam_stream_synchronize(slam_stream_t stream)
{
    SLAM_TRY(require_device());
    SLAM_HIP(hipStreamSynchronize(as_stream(stream)));
    return SLAM_OK;
}

int slam_device_synchronize(void)
{
    SLAM_TRY(require_device());
    SLAM_HIP(hipDeviceSynchronize());
    return SLAM_OK;
}

// A sequence of the library's asynchronous calls on one stream, recorded once and replayed with a single
// launch (a hipGraph): for launch-bound loops such as one registration + map-update step per batch.  Every
// call between begin and end must be stream-ordered on `stream` (no host buffers, no synchronising calls)
// and must have run once before, so that scratch buffers exist.
int slam_graph_begin_capture(slam_stream_t stream)
{
    SLAM_REQUIRE(stream, SLAM_E_INVALID, "slam_graph_begin_capture: a created stream is required");
    SLAM_TRY(require_device());
    SLAM_HIP(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
    return SLAM_OK;
}

int slam_graph_end_capture(slam_stream_t stream, slam_graph_t *out)
{
    SLAM_REQUIRE(stream && out, SLAM_E_INVALID, "slam_graph_end_capture: bad arguments");
    *out = nullptr;
    hipGraph_t g = nullptr;
    SLAM_HIP(hipStreamEndCapture(as_stream(stream), &g));
    hipGraphExec_t  ex = nullptr;
    const hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    SLAM_HIP(e);
    *out = (slam_graph_t)ex;
    return SLAM_OK;
}

int slam_graph_launch(slam_graph_t graph, slam_stream_t stream)
{
    SLAM_REQUIRE(graph, SLAM_E_INVALID, "slam_graph_launch: null graph");
    SLAM_HIP(hipGraphLaunch((hipGraphExec_t)graph, as_stream(stream)));
    return SLAM_OK;
}

int slam_graph_destroy(slam_graph_t graph)
{
    if (!graph) return SLAM_OK;
    SLAM_HIP(hipGraphExecDestroy((hipGraphExec_t)graph));
    return SLAM_OK;
}

int slam_event_create(slam_event_t *ev)
{
    SLAM_REQUIRE(ev, SLAM_E_INVALID, "slam_event_create: null out pointer");
    SLAM_TRY(require_device());
    hipEvent_t e;
    SLAM_HIP(hipEventCreate(&e));
    *ev = (slam_event_t)e;
    return SLAM_OK;
}

int slam_event_destroy(slam_event_t ev)
{
    if (!ev) return SLAM_OK;
    SLAM_HIP(hipEventDestroy((hipEvent_t)ev));
    return SLAM_OK;
}

int slam_event_record(slam_event_t ev, slam_stream_t stream)
{
    SLAM_HIP(hipEventRecord((hipEvent_t)ev, as_stream(stream)));
    return SLAM_OK;
}

int slam_event_synchronize(slam_event_t ev)
{
    SLAM_HIP(hipEventSynchronize((hipEvent_t)ev));
    return SLAM_OK;
}

int slam_event_query(slam_event_t ev, int *done)
{
    SLAM_REQUIRE(done, SLAM_E_INVALID, "slam_event_query: null out pointer");
    const hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        *done = 0;
        return SLAM_OK;
    }
    SLAM_HIP(e);
    *done = 1;
    return SLAM_OK;
}

int slam_event_elapsed_ms(slam_event_t start, slam_event_t stop, float *ms)
{
    SLAM_REQUIRE(ms, SLAM_E_INVALID, "slam_event_elapsed_ms: null out pointer");
    SLAM_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SLAM_OK;
}

} // extern "C"
