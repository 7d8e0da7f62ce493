// common.hpp -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "slam_mi355x.h"

namespace slam {

void set_error(const char *fmt, ...);
int  hip_fail(hipError_t e, const char *what, const char *file, int line);
// SLAM_OK once a HIP device is usable, SLAM_E_NO_DEVICE otherwise (and every
// compute entry point returns that: there is no CPU path in this library).
int  require_device();

inline hipStream_t as_stream(slam_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Device-memory pool for the short-lived buffers of a handle (the reference constructs its matcher per
// scan, icpTools.cpp:187: a hipMalloc/hipFree pair per buffer would cost more than the match).  pool_alloc
// returns a cached block of at least `bytes` or allocates one; pool_free puts it back -- the caller guarantees
// that no enqueued work still uses it (hipFree would have synchronised; the pool does not).
void *pool_alloc(size_t bytes);
void  pool_free(void *p);
void  pool_trim(); // hipFree of everything cached
// One pinned host buffer per host thread (grown on demand, never freed), for a call's staging and read-back;
// valid until the same thread asks again.
void *pinned_scratch(size_t bytes);
// Pinned blocks that outlive the call (an index build's plan travels back while the caller goes on): taken from and
// returned to a free list, never given back to the runtime.  At most 16 KB each.
void *pinned_block_get(size_t bytes);
void  pinned_block_put(void *p);
// The stream slam_icp_create[_dev] builds its index on: one per device, created on first use at the
// highest priority level and never destroyed.  A level of its own, because HIP deals the streams of one level over a
// few hardware queues shared in creation order: on the default stream the build's twenty short kernels and its host
// wait queued behind whatever registration shared that queue (0.4 ms).  Null on failure (callers fall back to
// the default stream).
hipStream_t build_stream();

} // namespace slam

#define SLAM_HIP(expr)                                                          \
    do {                                                                        \
        hipError_t e__ = (expr);                                                \
        if (e__ != hipSuccess) return slam::hip_fail(e__, #expr, __FILE__, __LINE__); \
    } while (0)

#define SLAM_REQUIRE(cond, code, ...)  \
    do {                               \
        if (!(cond)) {                 \
            slam::set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

#define SLAM_TRY(expr)              \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != SLAM_OK) return rc__; \
    } while (0)
