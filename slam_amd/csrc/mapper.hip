// mapper.hip -- the streaming form of the path (BASELINE config 5): chunks of scans go from pinned host
// memory through registration into the occupancy grid on three HIP streams, the ICP target is a sliding window
// of the scans registered so far, and every few chunks the grid is merged over the GPUs and finalized.
//
// Stands where scan_registration and local_mapper run one callback per scan (scan_registration.cpp:109-199:
// setSceneCloud -> doICPMatch against the target it keeps, :139-159; local_mapper.cpp:65-130: addToMap at
// 50 Hz).  Per chunk k, with two device slots:
//   copy stream : pinned chunk k -> HBM                                            -> event copied
//   icp  streams: wait copied -> slam_icp_fit_batch_dev against the current target -> decimate the registered
//   (two, chunks  points into the window ring                                       -> event registered
//    alternate)
//   grid stream : wait registered -> slam_grid_set_pose (rolling window, mls.cpp:408-479) + Bresenham update
//                 -> every merge_every chunks: dirty-row merge over the ranks (hook installed by
//                 slam_mapper_use_comm, slam_mi355x_rccl.h), fold into the accumulator, finalize -> event mapped
// so that the copy of chunk k+1, the registration of chunk k and the map update of chunk k-1 overlap.
// Every rebuild_every chunks the target is rebuilt on the device (icp_build.hip) from the prior map (optional) plus the
// thinned points of the last window_chunks chunks (the north-star's "sliding-window local map ... accepting staleness").
// The whole rebuild -- thinning, extent, plan, cell index, halo lists -- is ENQUEUED on a stream of its own with the counts
// it produces left on the device (build_index_begin): no host wait and no thread.  The producer keeps pushing against the
// old target and adopts the new one at the first push that finds the build's plan back in pinned memory (strict_window: it
// waits for it).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#include <vector>

#include "icp_model.hpp"

namespace slam {
namespace icp { // icp.hip
bool takes_spread_form(const slam_icp *h, int n_scans);
// slam_icp_create_dev in two halves (build_index_begin / _finish): begin enqueues the index build of a model of up to cap_*
// points per class (d_cnt: int[2] on the device, in stream order; null = exactly cap_*) on st and returns a handle that may
// only be passed to create_ready / create_finish / slam_icp_destroy; finish makes it usable (its one host wait) or destroys it.
int  create_begin(const double *d_ga, int cap_ga, const double *d_nga, int cap_nga, const int *d_cnt, const slam_icp_params *params,
                  hipStream_t st, slam_icp **out, bool beside);
bool create_ready(slam_icp *h);
int  create_finish(slam_icp *h);
} // namespace icp
} // namespace slam

using namespace slam;

namespace {

// registered points of a chunk, every stride-th of each class, into the window ring (map frame, f64 xy)
__global__ __launch_bounds__(256) void window_points_kernel(const double2 *pts, const int *scan_off, const int *scan_nga,
                                                            const int *ga_before, int n_scans, int n_points, const double *R,
                                                            const double *t, int stride_ga, int stride_nga, double2 *out_ga,
                                                            double2 *out_nga)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_points) return;
    int lo = 0, hi = n_scans - 1; // scan of point i: the last s with scan_off[s] <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (scan_off[mid] <= i)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int  s = lo, j = i - scan_off[s], g = scan_nga[s];
    const bool is_ga = j < g;
    const int  rank = is_ga ? ga_before[s] + j : (scan_off[s] - ga_before[s]) + (j - g);
    const int  stride = is_ga ? stride_ga : stride_nga;
    if (rank % stride) return;
    const double2 P = pts[i];
    const double *Rs = R + 4 * (size_t)s, *ts = t + 2 * (size_t)s;
    double2       q;
    q.x = __dadd_rn(__dadd_rn(__dmul_rn(Rs[0], P.x), __dmul_rn(Rs[1], P.y)), ts[0]); // icpPointToPoint.cpp:69-70 (kept f64)
    q.y = __dadd_rn(__dadd_rn(__dmul_rn(Rs[2], P.x), __dmul_rn(Rs[3], P.y)), ts[1]);
    (is_ga ? out_ga : out_nga)[rank / stride] = q;
}

// ---- thinning of the window: at most one point per cell of a lattice of pitch thin_res over the grid's extent and
// class -- the point with the lowest rank in window order (oldest chunk first, scan order inside): deterministic,
// and a wall seen by a thousand scans stays one point per cell instead of a thousand (pcl::VoxelGrid keeps the
// centroid, icpTools.cpp:620-633; a measured point is kept here so that the map holds only measurements).
struct Segs { // the window's points of one class, in window order
    const double2 *p[8];
    int            n[8];
    int            count, total;
};
struct ThinGeom {
    double x0, y0, inv;
    int    nx, ny;
};

__device__ inline bool seg_point(const Segs &s, int i, double2 *out)
{
    int k = 0;
    while (k < s.count - 1 && i >= s.n[k]) i -= s.n[k], ++k;
    *out = s.p[k][i];
    return true;
}

__device__ inline int thin_cell(const ThinGeom &g, const double2 q)
{
    const double fx = floor((q.x - g.x0) * g.inv), fy = floor((q.y - g.y0) * g.inv);
    if (!(fx >= 0.0 && fx < (double)g.nx && fy >= 0.0 && fy < (double)g.ny)) return -1; // outside the grid (or NaN)
    return (int)fy * g.nx + (int)fx;
}

// (The thinning kernels are launched over at most kThinGrid workgroups that stride over the window: with one workgroup per 256
// points -- 4300 of them -- a rebuild that starts in the same microsecond as a registration kept refilling every CU the
// registration had not reached yet with small workgroups, kernel after kernel, and the registration's last workgroups, which need
// a whole CU each, were placed only when the thinning was through: that registration took 490-550 us instead of 370.)
constexpr int kThinGrid = 256;

__global__ __launch_bounds__(256) void thin_min_kernel(Segs s, ThinGeom g, unsigned *lat)
{
  for (int i = blockIdx.x * 256 + threadIdx.x; i < s.total; i += gridDim.x * 256) {
    double2 q;
    seg_point(s, i, &q);
    const int c = thin_cell(g, q);
    // Look first: the lattice only ever goes down, so a cell that already shows a lower rank -- however stale the look -- cannot
    // be won by this point, and its atomic would change nothing.  The window is 1.1 M points on a few ten thousand wall cells, in
    // window order: all but the first few points of a cell lose, and without the look their atomics queue up on those cells'
    // words one after the other (112 + 132 us for the two classes alone on the chip, round 4).
    if (c >= 0 && __hip_atomic_load(&lat[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (unsigned)i) atomicMin(&lat[c], (unsigned)i);
  }
}

// PASS 0: winners per block; PASS 1: the winners, every stride-th, written in rank order -- the stride from the total the
// scan left on the device (more cells than the target may hold: every stride-th), the class's count (prior + kept) with it
template <int PASS>
__global__ __launch_bounds__(256) void thin_pick_kernel(Segs s, ThinGeom g, unsigned *lat, unsigned *block_count,
                                                        const unsigned *block_off, const unsigned *total, int cap, int prior,
                                                        int *count_out, double2 *out)
{
    __shared__ unsigned s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_blocks = (s.total + 255) / 256;
    // a workgroup takes the 256-point blocks vb = blockIdx.x, blockIdx.x + gridDim.x, ...: counts and offsets stay per block
    for (int vb = blockIdx.x; vb < n_blocks; vb += gridDim.x) {
        const int i = vb * 256 + (int)threadIdx.x;
        double2   q = make_double2(0.0, 0.0);
        bool      win = false;
        if (i < s.total) {
            seg_point(s, i, &q);
            const int c = thin_cell(g, q);
            win = c >= 0 && lat[c] == (unsigned)i;
        }
        const unsigned long long m = __ballot(win);
        if (lane == 0) s_w[wave] = (unsigned)__popcll(m);
        __syncthreads();
        if (PASS == 0) {
            if (threadIdx.x == 0) block_count[vb] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        } else {
            const unsigned kept = *total, stride = max(1u, (kept + (unsigned)cap - 1u) / (unsigned)max(cap, 1));
            if (vb == 0 && threadIdx.x == 0) *count_out = prior + (int)((kept + stride - 1u) / stride);
            unsigned k = block_off[vb] + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; ++w) k += s_w[w];
            if (win && k % stride == 0) out[k / stride] = q;
            // the winner leaves its cell as it found the lattice: nothing else reads this cell for a match with ITS rank, and a
            // loser that looks later sees "no rank of mine" either way -- the next rebuild needs no 4 MB fill per class
            if (win) lat[thin_cell(g, q)] = 0xffffffffu;
        }
        __syncthreads(); // s_w is written again for the next block
    }
}

__global__ void set_counts_kernel(int *cnt, int a, int b) { cnt[0] = a, cnt[1] = b; }

// exclusive prefix of the block counts (a few thousand), total behind the last; one workgroup of four wavefronts (one per
// SIMD, few registers: it runs beside a registration workgroup instead of waiting for a CU to come free)
constexpr int kThinScan = 256;
__global__ __launch_bounds__(kThinScan) void thin_scan_kernel(const unsigned *cnt, int n, unsigned *off, unsigned *total)
{
    __shared__ unsigned s_wave[kThinScan / 64], s_carry;
    const int           tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += kThinScan) {
        const int      i = base + tid;
        const unsigned v = i < n ? cnt[i] : 0u;
        unsigned       x = v;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        unsigned before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < n) off[i] = before + x - v;
        __syncthreads();
        if (tid == kThinScan - 1) s_carry = before + x;
        __syncthreads();
    }
    if (tid == 0) *total = s_carry;
}

struct Slot {
    double  *d_pts = nullptr, *d_R = nullptr, *d_t = nullptr;
    int32_t *d_off = nullptr, *d_nga = nullptr, *d_gab = nullptr;
    // pinned staging the producer fills
    double  *h_pts = nullptr, *h_R = nullptr, *h_t = nullptr;
    int32_t *h_off = nullptr, *h_nga = nullptr, *h_gab = nullptr;
    hipEvent_t copied = nullptr, fitted = nullptr, registered = nullptr, mapped = nullptr; // fitted: behind the registration
                                                                                          // launch; registered: poses on the host
    bool     busy = false;
    int      n_scans = 0;
};

constexpr int kMaxSlots = 8;

struct WindowEntry {
    double2 *ga = nullptr, *nga = nullptr; // device, room for max_points each
    int      n_ga = 0, n_nga = 0;
    long     chunk = -1;                   // which chunk lies here
    hipEvent_t ready = nullptr;            // recorded behind the kernel that filled it
};

} // namespace

struct slam_mapper {
    slam_mapper_params prm;
    slam_grid_t       *grid = nullptr;
    slam_icp_t        *target = nullptr, *retired = nullptr;
    hipEvent_t         target_used[2] = {nullptr, nullptr}, retired_used[2] = {nullptr, nullptr}; // behind the last launch on each
                                                                                                  // registration stream that read the handle
    hipStream_t        copy = nullptr, icp_s[2] = {nullptr, nullptr}, grid_s = nullptr; // chunks alternate over the two icp streams
    hipStream_t        build_s = nullptr;      // the sliding target's rebuilds
    hipStream_t        post_s = nullptr;       // what follows a registration and nothing on the registration stream waits for:
                                               // the window's points, the poses' way back to the host
    Slot               slot[kMaxSlots];
    int                n_slots = 5;
    bool               two_lanes = false;      // chunks alternate over two registration streams
    int                next = 0;
    long               chunks = 0, merges = 0, rebuilds = 0, last_rebuild = -1;
    std::vector<WindowEntry> window;
    std::vector<double> prior_ga, prior_nga; // host copies of the model given at create
    double            *d_model_ga = nullptr, *d_model_nga = nullptr; // [prior | window part] per class
    size_t             cap_model = 0;          // points reserved (both classes)
    unsigned          *d_thin = nullptr;      // [nx*ny] lowest window rank per lattice cell (one class after the other); all ones
                                              // between rebuilds: every rebuild's winners put their cells back (thin_pick_kernel<1>)
    bool               thin_dirty = false;    // a rebuild was abandoned half-way: fill the lattice before the next one
    unsigned          *d_thin_blk = nullptr;  // [2][blocks + blocks + 1] winners per block, their prefix, the total
    size_t             cap_thin_blk = 0;
    slam_mapper_merge_fn merge_begin = nullptr, merge_finish = nullptr;
    void              *merge_ctx = nullptr;
    bool               merge_pending = false;
    int                last_rows[2] = {0, -1};
    double             rebuild_ms = 0;         // host time spent enqueueing rebuilds and waiting for their plans
    int                device = 0, device_at_create = 0;
    int                max_lag = 0;            // pushes a build may stay un-adopted (0: every rebuild is waited for at once)
    slam_icp_t        *building = nullptr;     // the target whose build is enqueued and not yet adopted
    long               building_chunk = 0;     // m->chunks when it was begun
    int               *d_cnt = nullptr;        // points per class of the model being built (the thinning leaves them here)
    size_t             model_prior[2] = {0, 0}; // prior points resident at the head of d_model_ga / d_model_nga
};

namespace {

// Host-side trace of the mapper's calls (measurement build only, SLAM_MAPPER_TRACE=<file>): a label and a monotonic time per
// mark, written out by slam_mapper_finish; tools/xtrace_periods.py turns it into per-chunk periods.  This is what showed the
// producer standing still inside slam_grid_raycast_scans_dev whenever a chunk was the largest so far.
#ifdef SLAM_MEASURE
std::vector<std::pair<const char *, double>> g_trace;
inline void xt(const char *label)
{
    static const bool on = getenv("SLAM_MAPPER_TRACE") != nullptr;
    if (!on) return;
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    g_trace.push_back({label, ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3});
}
// ... and the DEVICE's side of it without a profiler (rocprofv3 slows the host's calls enough to make this pipeline
// host-bound, which it is not otherwise): a timing event per mark, recorded on the stream the work goes to; written out
// with the host marks as "DEV <label> <us since the first device mark>".
std::vector<std::pair<const char *, hipEvent_t>> g_dev;
inline void dt(const char *label, hipStream_t st)
{
    static const bool on = getenv("SLAM_MAPPER_TRACE") != nullptr;
    if (!on) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess || hipEventRecord(e, st) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    g_dev.push_back({label, e});
}
void xt_dump()
{
    const char *path = getenv("SLAM_MAPPER_TRACE");
    if (!path || g_trace.empty()) return;
    if (FILE *f = fopen(path, "w")) {
        for (size_t i = 0; i < g_trace.size(); ++i)
            fprintf(f, "%-18s %12.1f  +%.1f\n", g_trace[i].first, g_trace[i].second - g_trace[0].second,
                    i ? g_trace[i].second - g_trace[i - 1].second : 0.0);
        (void)hipDeviceSynchronize();
        for (size_t i = 0; i < g_dev.size(); ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, g_dev[0].second, g_dev[i].second) != hipSuccess) (void)hipGetLastError();
            fprintf(f, "DEV %-18s %12.1f\n", g_dev[i].first, ms * 1e3);
        }
        fclose(f);
    }
    for (auto &d : g_dev) (void)hipEventDestroy(d.second);
    g_dev.clear();
    g_trace.clear();
}
#else
inline void xt(const char *) {}
inline void dt(const char *, hipStream_t) {}
inline void xt_dump() {}
#endif

#define MAP_HIP(expr) SLAM_HIP(expr)

void retire_now(slam_mapper *m)
{
    if (!m->retired) return;
    for (hipEvent_t e : m->retired_used)
        if (e) (void)hipEventSynchronize(e);
    slam::icp::destroy_unsynchronised(m->retired);
    m->retired = nullptr;
}

// decimation that keeps about target_points / (2 * window) points of a class per chunk
int stride_for(const slam_mapper *m, int n)
{
    if (m->prm.thin_res > 0) return 1; // thinned over the whole window at rebuild time
    const int per_chunk = std::max(64, m->prm.target_points / std::max(2 * m->prm.window_chunks, 1));
    return std::max(1, (n + per_chunk - 1) / per_chunk);
}

// one class of the window thinned into out[] (room for cap points); the class's count (prior + kept) goes to *count_out
int thin_class(slam_mapper *m, const std::vector<const WindowEntry *> &use, int cls, int cap, int prior, int *count_out, double2 *out,
               hipStream_t st)
{
    Segs s;
    memset(&s, 0, sizeof s);
    for (const WindowEntry *w : use) {
        const int n = cls ? w->n_nga : w->n_ga;
        if (!n) continue;
        s.p[s.count] = cls ? w->nga : w->ga;
        s.n[s.count] = n;
        s.total += n;
        ++s.count;
    }
    if (!s.total) return SLAM_OK; // (the count stays the prior's: set_counts_kernel)
    ThinGeom g;
    int      gx = 0, gy = 0;
    double   res = 0;
    SLAM_TRY(slam_grid_info(m->grid, &gx, &gy, &res, nullptr, nullptr));
    g.inv = 1.0 / m->prm.thin_res;
    g.nx = (int)std::ceil(gx * res * g.inv);
    g.ny = (int)std::ceil(gy * res * g.inv);
    g.x0 = -0.5 * gx * res; // the grid's extent in the map frame (mls.h:167-175: origin -res*size/2)
    g.y0 = -0.5 * gy * res;
    const size_t cells = (size_t)g.nx * g.ny;
    const int    blocks = (s.total + 255) / 256;
    if (!m->d_thin) {
        MAP_HIP(hipMalloc((void **)&m->d_thin, 4 * cells));
        m->thin_dirty = true;
    }
    if (m->thin_dirty) {
        MAP_HIP(hipMemsetAsync(m->d_thin, 0xff, 4 * cells, st));
        m->thin_dirty = false;
    }
    if ((size_t)(2 * blocks + 1) > m->cap_thin_blk) {
        if (m->d_thin_blk) {
            MAP_HIP(hipStreamSynchronize(st)); // (grows once or twice in a mapper's life)
            (void)hipFree(m->d_thin_blk);
        }
        m->cap_thin_blk = (size_t)(2 * blocks + 1) * 2;
        MAP_HIP(hipMalloc((void **)&m->d_thin_blk, 4 * m->cap_thin_blk));
    }
    unsigned *cnt = m->d_thin_blk, *off = cnt + blocks, *total = off + blocks;
    // (the lattice is all ones: filled when it was allocated, and every rebuild's winners put it back, thin_pick_kernel<1>)
    const int grid = std::min(blocks, kThinGrid);
    hipLaunchKernelGGL(thin_min_kernel, dim3(grid), dim3(256), 0, st, s, g, m->d_thin);
    hipLaunchKernelGGL((thin_pick_kernel<0>), dim3(grid), dim3(256), 0, st, s, g, m->d_thin, cnt, off, total, cap, prior, count_out, out);
    hipLaunchKernelGGL(thin_scan_kernel, dim3(1), dim3(kThinScan), 0, st, cnt, blocks, off, total);
    hipLaunchKernelGGL((thin_pick_kernel<1>), dim3(grid), dim3(256), 0, st, s, g, m->d_thin, cnt, off, total, cap, prior, count_out, out);
    MAP_HIP(hipGetLastError());
    return SLAM_OK;
}

// the window as the next target sees it: the newest window_chunks chunks, oldest first
void collect_window(slam_mapper *m, std::vector<const WindowEntry *> &use)
{
    use.clear();
    for (const WindowEntry &w : m->window)
        if (w.chunk >= 0) use.push_back(&w);
    std::sort(use.begin(), use.end(), [](const WindowEntry *a, const WindowEntry *b) { return a->chunk < b->chunk; });
    if ((int)use.size() > m->prm.window_chunks) use.erase(use.begin(), use.end() - m->prm.window_chunks); // the newest W
}

// the producer's thread: the new target replaces the current one; the one before last is destroyed
void adopt_target(slam_mapper *m, slam_icp_t *fresh)
{
    if (!fresh) return;
    retire_now(m); // the handle before last: its launches ended chunks ago
    m->retired = m->target;
    std::swap(m->retired_used[0], m->target_used[0]);
    std::swap(m->retired_used[1], m->target_used[1]);
    m->target = fresh;
    ++m->rebuilds;
}

// Enqueues on st the target for the window as it is now: prior map (optional) + the window, thinned, and the build of its
// index.  Everything it reads is waited for in stream order (the window entries: behind their events); the host waits for nothing.
int begin_rebuild(slam_mapper *m, hipStream_t st)
{
    std::vector<const WindowEntry *> use;
    collect_window(m, use);
    if (use.empty()) return SLAM_OK; // nothing registered yet: keep the current target
    const auto   t0 = std::chrono::steady_clock::now();
    const size_t p_ga = m->prm.keep_prior ? m->prior_ga.size() / 2 : 0, p_nga = m->prm.keep_prior ? m->prior_nga.size() / 2 : 0;
    const bool   thin = m->prm.thin_res > 0;
    size_t       w_ga = 0, w_nga = 0; // room for the window's part of each class
    for (const WindowEntry *w : use) w_ga += (size_t)w->n_ga, w_nga += (size_t)w->n_nga;
    if (thin) w_ga = w_nga = (size_t)std::max(64, m->prm.target_points / 2); // one point per lattice cell and class, at most this many
    const size_t cap_ga = p_ga + w_ga, cap_nga = p_nga + w_nga;
    if (!thin && cap_ga + cap_nga < 5) return SLAM_OK;
    for (const WindowEntry *w : use) MAP_HIP(hipStreamWaitEvent(st, w->ready, 0));
    dt("rebuild>", st);
    if (cap_ga + cap_nga > m->cap_model || !m->d_model_ga) {
        if (m->d_model_ga) {
            MAP_HIP(hipStreamSynchronize(st)); // an earlier build may still read the old block (grows once or twice)
            pool_free(m->d_model_ga);
        }
        m->cap_model = (cap_ga + cap_nga) * 2;
        m->d_model_ga = static_cast<double *>(pool_alloc(16 * m->cap_model));
        if (!m->d_model_ga) return SLAM_E_NOMEM;
        m->model_prior[0] = m->model_prior[1] = (size_t)-1;
    }
    if (!m->d_cnt) MAP_HIP(hipMalloc((void **)&m->d_cnt, 2 * sizeof(int)));
    // class GA at the head of the block, class NGA behind GA's reservation; the prior's points go in when the layout changes
    double *d_ga = m->d_model_ga, *d_nga = m->d_model_ga + 2 * cap_ga;
    if (m->model_prior[0] != p_ga || m->model_prior[1] != p_nga || m->d_model_nga != d_nga) {
        if (p_ga) MAP_HIP(hipMemcpyAsync(d_ga, m->prior_ga.data(), 16 * p_ga, hipMemcpyHostToDevice, st));
        if (p_nga) MAP_HIP(hipMemcpyAsync(d_nga, m->prior_nga.data(), 16 * p_nga, hipMemcpyHostToDevice, st));
        m->model_prior[0] = p_ga;
        m->model_prior[1] = p_nga;
        m->d_model_nga = d_nga;
    }
    const int *d_cnt = nullptr;
    if (thin) {
        hipLaunchKernelGGL(set_counts_kernel, dim3(1), dim3(1), 0, st, m->d_cnt, (int)p_ga, (int)p_nga);
        int rc = thin_class(m, use, 0, (int)w_ga, (int)p_ga, m->d_cnt + 0, reinterpret_cast<double2 *>(d_ga + 2 * p_ga), st);
        if (rc == SLAM_OK) rc = thin_class(m, use, 1, (int)w_nga, (int)p_nga, m->d_cnt + 1, reinterpret_cast<double2 *>(d_nga + 2 * p_nga), st);
        if (rc != SLAM_OK) {
            m->thin_dirty = true; // (a class's passes may have stopped between marking the lattice and putting it back)
            return rc;
        }
        d_cnt = m->d_cnt;
    } else {
        size_t o_ga = p_ga, o_nga = p_nga;
        for (const WindowEntry *w : use) {
            if (w->n_ga) MAP_HIP(hipMemcpyAsync(d_ga + 2 * o_ga, w->ga, 16 * (size_t)w->n_ga, hipMemcpyDeviceToDevice, st));
            if (w->n_nga) MAP_HIP(hipMemcpyAsync(d_nga + 2 * o_nga, w->nga, 16 * (size_t)w->n_nga, hipMemcpyDeviceToDevice, st));
            o_ga += (size_t)w->n_ga;
            o_nga += (size_t)w->n_nga;
        }
    }
    // (a thinned window has no long halo lists: the build's small-LDS variants, which run beside the registrations' workgroups)
    SLAM_TRY(slam::icp::create_begin(d_ga, (int)cap_ga, d_nga, (int)cap_nga, d_cnt, &m->prm.icp, st, &m->building, thin));
    dt("rebuild<", st);
    m->building_chunk = m->chunks;
    m->rebuild_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return SLAM_OK;
}

// The build in flight becomes the target: when its plan is back, or -- block -- after waiting for it.  A window that came
// to fewer than five points (icp.cpp:38-43) leaves the current target in place.
int adopt_build(slam_mapper *m, bool block)
{
    if (!m->building) return SLAM_OK;
    if (!block && !slam::icp::create_ready(m->building)) return SLAM_OK;
    const auto  t0 = std::chrono::steady_clock::now();
    slam_icp_t *fresh = m->building;
    m->building = nullptr;
    const int rc = slam::icp::create_finish(fresh); // (destroys the handle when it fails)
    m->rebuild_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc == SLAM_E_TOO_FEW_MODEL_POINTS) return SLAM_OK;
    SLAM_TRY(rc);
    adopt_target(m, fresh);
    return SLAM_OK;
}

int finish_merge(slam_mapper *m)
{
    if (!m->merge_pending) return SLAM_OK;
    m->merge_pending = false;
    int lo = 0, hi = -1;
    SLAM_TRY(m->merge_finish(m->merge_ctx, m->grid, (slam_stream_t)m->grid_s, &lo, &hi));
    m->last_rows[0] = lo;
    m->last_rows[1] = hi;
    SLAM_TRY(slam_grid_fold(m->grid, lo, hi, (slam_stream_t)m->grid_s));
    SLAM_TRY(slam_grid_finalize(m->grid, (slam_stream_t)m->grid_s));
    ++m->merges;
    return SLAM_OK;
}

} // namespace

extern "C" {

void slam_mapper_default_params(slam_mapper_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->grid_size_x = p->grid_size_y = 2000;
    p->resolution = 0.05;
    slam_grid_default_params(&p->grid);
    p->grid.min_cluster_points = 20; // local_mapper.cpp:86
    slam_icp_default_params(&p->icp);
    p->indist = 5.0;                 // icpTools.cpp:188
    p->max_scans = 256;
    p->max_points = 256 * 1081;
    p->window_chunks = 0;
    p->rebuild_every = 1;
    p->target_points = 2 * 19999;    // ICP_MAX_PTS per class, icpTools.h:21
    p->keep_prior = 0;
    p->merge_every = 0;
    p->pipelined = 1;
    p->strict_window = 0;
    p->thin_res = 0.0;
    p->slots = 0;
    p->background_rebuild = 1;
    p->registration_streams = 0;
}

int slam_mapper_create(const slam_mapper_params *params, const double *m_ga, int n_ga, const double *m_nga, int n_nga,
                       slam_mapper_t **out)
{
    SLAM_REQUIRE(params && out, SLAM_E_INVALID, "slam_mapper_create: bad arguments");
    *out = nullptr;
    SLAM_REQUIRE(params->max_scans > 0 && params->max_points > 0 && params->window_chunks >= 0 && params->window_chunks <= 8 &&
                     params->rebuild_every >= 1 && params->merge_every >= 0 && params->thin_res >= 0,
                 SLAM_E_INVALID, "slam_mapper_create: bad parameters");
    SLAM_TRY(require_device());
    slam_mapper *m = new (std::nothrow) slam_mapper();
    SLAM_REQUIRE(m, SLAM_E_NOMEM, "slam_mapper_create: out of host memory");
    m->prm = *params;
    {
        // Two registrations are in flight (two streams): with two scans per workgroup a chunk of up to two scans per CU
        // holds half the CUs it would hold otherwise, and the chunk after it starts beside it instead of behind its
        // slowest scan (256-scan chunks: 0.41 -> 0.36 ms).  Below half a scan per CU two chunks fit side by side anyway.
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            params->pipelined && (params->registration_streams == 2 || (!params->window_chunks && params->registration_streams == 0)) &&
            m->prm.icp.pair_scans == 0 && 2 * params->max_scans > cus)
            m->prm.icp.pair_scans = 2;
        // A sliding target's chunks register one scan per workgroup whatever their size: a pair workgroup fills its CU's register
        // file, and the rebuild's thirty short kernels -- written to run BESIDE a registration workgroup -- would each wait for a
        // whole launch to end.  Measured (round 5, tools/exp/c5_chunk.sh): chunks of 512 scans, which the library pairs by default
        // (two scans per CU), 2.3-2.6 ms per chunk with rebuilds of 5-15 ms that the producer ends up waiting for; one scan per
        // workgroup: see DESIGN.md 6.
        else if (params->window_chunks && params->registration_streams != 2 && m->prm.icp.pair_scans == 0)
            m->prm.icp.pair_scans = -1;
        (void)hipGetLastError();
    }
    int rc = slam_grid_create(params->grid_size_x, params->grid_size_y, params->resolution, &m->prm.grid, &m->grid);
    // (chunks differ in size: a raycast that had to grow its scratch would free the old block, and a free waits for the device)
    if (rc == SLAM_OK) rc = slam_grid_reserve(m->grid, params->max_points);
    if (rc == SLAM_OK) rc = slam_icp_create(m_ga, n_ga, m_nga, n_nga, &m->prm.icp, &m->target); // the prior map: the first target
    auto hip = [&](hipError_t e) {
        if (rc == SLAM_OK && e != hipSuccess) rc = hip_fail(e, "slam_mapper_create", __FILE__, __LINE__);
    };
    if (rc == SLAM_OK) {
        m->prior_ga.assign(m_ga, m_ga + 2 * (size_t)n_ga);
        m->prior_nga.assign(m_nga, m_nga + 2 * (size_t)n_nga);
        (void)hipGetDevice(&m->device_at_create);
        if (params->pipelined) {
            // HIP deals the streams of a process over a few hardware queues: a handful for the default priority level,
            // dealt in creation order over everything the application made before, and (as far as the timings tell) one
            // each for the high and the low level.  Two streams on one queue run one after the other (measured: three
            // default-level streams 0.42 ms per chunk alone, 0.54 and 0.65 ms with one and two idle application streams
            // made first; the index build's stream on the registration stream's level: rebuilds 1.3 -> 3.2 ms).  So: the
            // two registration streams on the default level, the copies on the high level (which they share with the
            // index build's stream: both are short), the grid update on the low level.  The levels themselves make no
            // measurable difference to the kernels (tools/pipeline_experiment.py).
            int least = 0, greatest = 0;
            hip(hipDeviceGetStreamPriorityRange(&least, &greatest));
            const int mid = (least + greatest) / 2;
            hip(hipStreamCreateWithPriority(&m->copy, hipStreamNonBlocking, greatest));
            m->two_lanes = params->registration_streams == 2 || (params->registration_streams == 0 && !params->window_chunks);
            hip(hipStreamCreateWithPriority(&m->icp_s[0], hipStreamNonBlocking, mid));
            hip(hipStreamCreateWithPriority(&m->grid_s, hipStreamNonBlocking, least));
            if (m->two_lanes) hip(hipStreamCreateWithPriority(&m->icp_s[1], hipStreamNonBlocking, mid));
            if (!m->two_lanes) m->icp_s[1] = m->icp_s[0]; // a sliding target registers its chunks one after the other (slam_mapper_push)
            // The sliding target's rebuilds on a stream of their own, on the grid update's level: about thirty short launches, none
            // of which anything waits for.  Measured on config 5: own stream on the low level 0.460 ms per chunk, default level
            // 0.469, the registration stream itself 0.52 (the chain sits between two registrations while the raycast of the chunk
            // before holds the CUs), the copy stream 0.75 (the next chunk's copy queues behind it).
            // Round 4: "a stream of its own" was not a QUEUE of its own.  The runtime deals a process's streams over a few
            // hardware queues, and the kernel trace of config 5 showed every rebuild kernel on the queue of the copy stream
            // (profiles/r04_config5_queues.txt): the next chunk's host-to-device copies -- which its registration waits for --
            // stood behind the rebuild's thirty launches, a 0.25 ms hole in the registrations per rebuild (0.5 ms under the
            // profiler).  A stream made with a CU mask gets a hardware queue that no other stream is put on
            // (tools/exp/queues2.hip); the mask here names every CU.
            {
                int n_cu = 0;
                hip(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, m->device_at_create));
                std::vector<uint32_t> all((size_t)std::max((n_cu + 31) / 32, 1), 0u);
                for (int i = 0; i < n_cu; ++i) all[(size_t)i / 32] |= 1u << (i % 32);
                // the stream of what follows a registration (slam_mapper_push): a queue of its own, like the rebuild's
                if (hipExtStreamCreateWithCUMask(&m->post_s, (uint32_t)all.size(), all.data()) != hipSuccess) {
                    (void)hipGetLastError();
                    hip(hipStreamCreateWithPriority(&m->post_s, hipStreamNonBlocking, greatest));
                }
            }
            if (params->window_chunks) {
                int n_cu = 0;
                hip(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, m->device_at_create));
                std::vector<uint32_t> all((size_t)std::max((n_cu + 31) / 32, 1), 0u);
                for (int i = 0; i < n_cu; ++i) all[(size_t)i / 32] |= 1u << (i % 32);
                bool masked = true;
#ifdef SLAM_MEASURE
                if (const char *e = getenv("SLAM_MAPPER_BUILD_STREAM")) masked = strcmp(e, "low") != 0; // A/B: the low priority level's stream
#endif
                if (!masked || hipExtStreamCreateWithCUMask(&m->build_s, (uint32_t)all.size(), all.data()) != hipSuccess) {
                    (void)hipGetLastError();
                    hip(hipStreamCreateWithPriority(&m->build_s, hipStreamNonBlocking, least));
                }
            }
        } else {
            hip(hipStreamCreateWithFlags(&m->copy, hipStreamNonBlocking));
            m->icp_s[0] = m->icp_s[1] = m->grid_s = m->build_s = m->copy;
        }
        for (int k = 0; k < 2; ++k) {
            hip(hipEventCreateWithFlags(&m->target_used[k], hipEventDisableTiming));
            hip(hipEventCreateWithFlags(&m->retired_used[k], hipEventDisableTiming));
        }
        const size_t np = (size_t)params->max_points, ns = (size_t)params->max_scans;
        m->n_slots = params->slots >= 2 && params->slots <= kMaxSlots ? params->slots : 5;
        for (int k = 0; k < m->n_slots; ++k) {
            Slot &b = m->slot[k];
            hip(hipMalloc((void **)&b.d_pts, 16 * np));
            hip(hipMalloc((void **)&b.d_off, 4 * (ns + 1)));
            hip(hipMalloc((void **)&b.d_nga, 4 * ns));
            hip(hipMalloc((void **)&b.d_gab, 4 * (ns + 1)));
            hip(hipMalloc((void **)&b.d_R, 32 * ns));
            hip(hipMalloc((void **)&b.d_t, 16 * ns));
            hip(hipHostMalloc((void **)&b.h_pts, 16 * np, hipHostMallocDefault));
            hip(hipHostMalloc((void **)&b.h_off, 4 * (ns + 1), hipHostMallocDefault));
            hip(hipHostMalloc((void **)&b.h_nga, 4 * ns, hipHostMallocDefault));
            hip(hipHostMalloc((void **)&b.h_gab, 4 * (ns + 1), hipHostMallocDefault));
            hip(hipHostMalloc((void **)&b.h_R, 32 * ns, hipHostMallocDefault));
            hip(hipHostMalloc((void **)&b.h_t, 16 * ns, hipHostMallocDefault));
            hip(hipEventCreateWithFlags(&b.copied, hipEventDisableTiming));
            hip(hipEventCreateWithFlags(&b.fitted, hipEventDisableTiming));
            hip(hipEventCreateWithFlags(&b.registered, hipEventDisableTiming));
            hip(hipEventCreateWithFlags(&b.mapped, hipEventDisableTiming));
        }
        hip(hipGetDevice(&m->device));
        if (!m->build_s) m->build_s = m->icp_s[0];
        if (!m->post_s) m->post_s = m->icp_s[0];
        // the window keeps one entry more than it uses: the newest is still being written when a rebuild looks -- and as
        // many more as pushes may pass while a background rebuild reads its entries (it is waited for after max_lag)
        const bool bg = params->window_chunks && params->background_rebuild && !params->strict_window;
        m->max_lag = bg ? std::min(std::max(params->rebuild_every, 1), 4) : 0;
        m->window.resize(params->window_chunks ? (size_t)params->window_chunks + 1 + (size_t)m->max_lag : 0);
        const size_t per = params->thin_res > 0 ? (size_t)params->max_points
                                                : (size_t)std::max(64, params->target_points / std::max(2 * params->window_chunks, 1)) + 8;
        for (WindowEntry &w : m->window) {
            hip(hipMalloc((void **)&w.ga, 16 * per));
            hip(hipMalloc((void **)&w.nga, 16 * per));
            hip(hipEventCreateWithFlags(&w.ready, hipEventDisableTiming));
        }
    }
    if (rc != SLAM_OK) {
        slam_mapper_destroy(m);
        return rc;
    }
    *out = m;
    return SLAM_OK;
}

void slam_mapper_destroy(slam_mapper_t *m)
{
    if (!m) return;
    (void)hipDeviceSynchronize();
    if (m->building) slam_icp_destroy(m->building);
    for (Slot &b : m->slot) {
        for (void *p : {(void *)b.d_pts, (void *)b.d_off, (void *)b.d_nga, (void *)b.d_gab, (void *)b.d_R, (void *)b.d_t})
            if (p) (void)hipFree(p);
        for (void *p : {(void *)b.h_pts, (void *)b.h_off, (void *)b.h_nga, (void *)b.h_gab, (void *)b.h_R, (void *)b.h_t})
            if (p) (void)hipHostFree(p);
        for (hipEvent_t e : {b.copied, b.fitted, b.registered, b.mapped})
            if (e) (void)hipEventDestroy(e);
    }
    for (WindowEntry &w : m->window) {
        if (w.ga) (void)hipFree(w.ga);
        if (w.nga) (void)hipFree(w.nga);
        if (w.ready) (void)hipEventDestroy(w.ready);
    }
    if (m->d_model_ga) pool_free(m->d_model_ga);
    for (void *p : {(void *)m->d_thin, (void *)m->d_thin_blk, (void *)m->d_cnt})
        if (p) (void)hipFree(p);

    if (m->target) slam_icp_destroy(m->target);
    if (m->retired) slam_icp_destroy(m->retired);
    if (m->grid) slam_grid_destroy(m->grid);
    for (int k = 0; k < 2; ++k) {
        if (m->target_used[k]) (void)hipEventDestroy(m->target_used[k]);
        if (m->retired_used[k]) (void)hipEventDestroy(m->retired_used[k]);
    }
    const bool one = m->icp_s[0] == m->copy;
    if (m->copy) (void)hipStreamDestroy(m->copy);
    if (!one && m->icp_s[0]) (void)hipStreamDestroy(m->icp_s[0]);
    if (!one && m->icp_s[1] && m->icp_s[1] != m->icp_s[0]) (void)hipStreamDestroy(m->icp_s[1]);
    if (!one && m->grid_s) (void)hipStreamDestroy(m->grid_s);
    if (m->build_s && m->build_s != m->copy && m->build_s != m->icp_s[0] && m->build_s != m->grid_s) (void)hipStreamDestroy(m->build_s);
    if (m->post_s && m->post_s != m->copy && m->post_s != m->icp_s[0] && m->post_s != m->grid_s) (void)hipStreamDestroy(m->post_s);
    delete m;
}

int slam_mapper_chunk_buffers(slam_mapper_t *m, int slot, double **pts, int32_t **scan_off, int32_t **scan_nga, double **R0,
                              double **t0)
{
    SLAM_REQUIRE(m && slot >= 0 && slot < m->n_slots, SLAM_E_INVALID, "slam_mapper_chunk_buffers: bad arguments");
    Slot &b = m->slot[slot];
    if (pts) *pts = b.h_pts;
    if (scan_off) *scan_off = b.h_off;
    if (scan_nga) *scan_nga = b.h_nga;
    if (R0) *R0 = b.h_R;
    if (t0) *t0 = b.h_t;
    return SLAM_OK;
}

int slam_mapper_next_slot(slam_mapper_t *m, int *slot)
{
    SLAM_REQUIRE(m && slot, SLAM_E_INVALID, "slam_mapper_next_slot: bad arguments");
    *slot = m->next;
    return SLAM_OK;
}

int slam_mapper_slots(slam_mapper_t *m, int *n_slots)
{
    SLAM_REQUIRE(m && n_slots, SLAM_E_INVALID, "slam_mapper_slots: bad arguments");
    *n_slots = m->n_slots;
    return SLAM_OK;
}

int slam_mapper_set_merge(slam_mapper_t *m, slam_mapper_merge_fn begin, slam_mapper_merge_fn finish, void *ctx)
{
    SLAM_REQUIRE(m && ((begin && finish) || (!begin && !finish)), SLAM_E_INVALID, "slam_mapper_set_merge: bad arguments");
    m->merge_begin = begin;
    m->merge_finish = finish;
    m->merge_ctx = ctx;
    if (begin) SLAM_TRY(slam_grid_enable_accumulator(m->grid));
    return SLAM_OK;
}

int slam_mapper_push(slam_mapper_t *m, int n_scans, int n_points, double window_x, double window_y, int *slot_out)
{
    SLAM_REQUIRE(m && n_scans > 0 && n_points > 0, SLAM_E_INVALID, "slam_mapper_push: bad arguments");
    SLAM_REQUIRE(n_scans <= m->prm.max_scans && n_points <= m->prm.max_points, SLAM_E_INVALID,
                 "slam_mapper_push: chunk of %d scans / %d points exceeds the reservation (%d / %d)", n_scans, n_points,
                 m->prm.max_scans, m->prm.max_points);
    // (everything that can refuse the chunk comes before anything is consumed: a push that fails leaves the slot, its
    // pinned buffers and slam_mapper_next_slot() as they were, and the caller may push the same chunk again)
    const int s = m->next;
    Slot &b = m->slot[s];
    SLAM_REQUIRE(!b.busy, SLAM_E_INVALID, "slam_mapper_push: slot %d still holds a chunk that was not waited for", s);
    SLAM_REQUIRE(b.h_off[0] == 0 && b.h_off[n_scans] == n_points, SLAM_E_INVALID, "slam_mapper_push: scan_off does not span the chunk");
    // points of class GA before each scan (the window's decimation ranks points per class)
    b.h_gab[0] = 0;
    for (int k = 0; k < n_scans; ++k) b.h_gab[k + 1] = b.h_gab[k] + b.h_nga[k];
    const int n_ga = b.h_gab[n_scans], n_nga = n_points - n_ga;

    // ---- sliding target: rebuilt before this chunk's registration is enqueued
    // (enqueued on the rebuild's own stream; a build still un-adopted is waited for only when the next one is due or after
    // max_lag pushes)
    xt("push");
    const bool due = m->prm.window_chunks && m->chunks > 0 && m->chunks - std::max<long>(m->last_rebuild, 0) >= m->prm.rebuild_every;
    if (m->building) SLAM_TRY(adopt_build(m, due || m->chunks - m->building_chunk >= std::max(m->max_lag, 1)));
    xt("adopted");
    if (due) {
        SLAM_TRY(begin_rebuild(m, m->build_s));
        xt("begun");
        m->last_rebuild = m->chunks;
        if (!m->max_lag) SLAM_TRY(adopt_build(m, true)); // strict_window / background_rebuild = 0: this chunk meets the new target
    }

    // ---- copy
    dt("copy>", m->copy);
    MAP_HIP(hipMemcpyAsync(b.d_pts, b.h_pts, 16 * (size_t)n_points, hipMemcpyHostToDevice, m->copy));
    MAP_HIP(hipMemcpyAsync(b.d_off, b.h_off, 4 * (size_t)(n_scans + 1), hipMemcpyHostToDevice, m->copy));
    MAP_HIP(hipMemcpyAsync(b.d_nga, b.h_nga, 4 * (size_t)n_scans, hipMemcpyHostToDevice, m->copy));
    MAP_HIP(hipMemcpyAsync(b.d_gab, b.h_gab, 4 * (size_t)(n_scans + 1), hipMemcpyHostToDevice, m->copy));
    MAP_HIP(hipMemcpyAsync(b.d_R, b.h_R, 32 * (size_t)n_scans, hipMemcpyHostToDevice, m->copy));
    MAP_HIP(hipMemcpyAsync(b.d_t, b.h_t, 16 * (size_t)n_scans, hipMemcpyHostToDevice, m->copy));
    MAP_HIP(hipEventRecord(b.copied, m->copy));
    dt("copy<", m->copy);
    xt("copied");
    // ---- register
    // chunks alternate over the two registration streams; the spread form (a handful of scans) takes one call at a time
    // (and so do the chunks of a sliding target: a chunk registered beside its predecessor meets a window that is a chunk
    // staler, and every further chunk in flight costs more than the overlap gains -- config 5: 0.54 ms per chunk on one
    // stream, 0.51 / 0.53 / 0.56 on two with three / four / five chunks in flight)
    const int   lane = (!m->two_lanes || slam::icp::takes_spread_form(m->target, n_scans)) ? 0 : (int)(m->chunks & 1);
    hipStream_t icp_s = m->icp_s[lane];
    MAP_HIP(hipStreamWaitEvent(icp_s, b.copied, 0));
    dt("fit>", icp_s);
    SLAM_TRY(slam_icp_fit_batch_dev(m->target, b.d_pts, b.d_off, b.d_nga, n_scans, b.d_R, b.d_t, m->prm.indist, nullptr, nullptr,
                                    (slam_stream_t)icp_s));
    MAP_HIP(hipEventRecord(m->target_used[lane], icp_s));
    dt("fit<", icp_s);
    // With ONE registration stream (a sliding target) everything else a chunk needs after its registration goes to a stream of its
    // own behind this event: on the registration stream the window kernel and the two copies were three dispatches between one
    // registration and the next, 30-40 us in which the chip registered nothing (round 4: 0.413 -> 0.406 ms per chunk of config 5).
    // With two registration streams in turn the gap on one is covered by the other, and the map update that starts the moment a
    // registration ends takes CUs from the next one's workgroups (measured: 0.335-0.348 -> 0.357-0.377 ms per chunk): there the
    // copies stay where they were and the map update waits for them.
    MAP_HIP(hipEventRecord(b.fitted, icp_s));
    hipStream_t post_s = m->two_lanes ? icp_s : m->post_s;
    if (post_s != icp_s) MAP_HIP(hipStreamWaitEvent(post_s, b.fitted, 0));
    if (m->prm.window_chunks) {
        WindowEntry &w = m->window[(size_t)(m->chunks % (long)m->window.size())];
        const int    sg = stride_for(m, n_ga), sn = stride_for(m, n_nga);
        // (no wait for a rebuild in flight: it reads the entries of the window_chunks chunks before the push it was begun at,
        // the ring holds 1 + max_lag entries more than that, and a build is adopted -- complete -- before max_lag pushes have passed)
        hipLaunchKernelGGL(window_points_kernel, dim3((n_points + 255) / 256), dim3(256), 0, post_s,
                           reinterpret_cast<const double2 *>(b.d_pts), b.d_off, b.d_nga, b.d_gab, n_scans, n_points, b.d_R, b.d_t, sg, sn,
                           w.ga, w.nga);
        MAP_HIP(hipGetLastError());
        w.n_ga = (n_ga + sg - 1) / sg;
        w.n_nga = (n_nga + sn - 1) / sn;
        w.chunk = m->chunks;
        MAP_HIP(hipEventRecord(w.ready, post_s));
    }
    // the registered poses go back to the slot's pinned pose buffers on THIS stream (12 KB): slam_mapper_wait then needs no
    // stream of its own -- round 2 read them back on the copy stream and synchronised it, which during a background rebuild
    // (whose launches share that stream) held the producer for the whole rebuild: a 0.4 ms hole in the registrations per rebuild
    MAP_HIP(hipMemcpyAsync(b.h_R, b.d_R, 32 * (size_t)n_scans, hipMemcpyDeviceToHost, post_s));
    MAP_HIP(hipMemcpyAsync(b.h_t, b.d_t, 16 * (size_t)n_scans, hipMemcpyDeviceToHost, post_s));
    MAP_HIP(hipEventRecord(b.registered, post_s));
    xt("fit enqueued");
    // ---- the previous chunk's merge, now that this chunk's registration is in the queue ahead of the wait
    SLAM_TRY(finish_merge(m));
    // ---- map
    MAP_HIP(hipStreamWaitEvent(m->grid_s, m->two_lanes ? b.registered : b.fitted, 0)); // (one lane: the poses are in HBM behind the
                                                                                      // registration, no need to wait for their copy)
    SLAM_TRY(slam_grid_set_pose(m->grid, window_x, window_y, (slam_stream_t)m->grid_s)); // MLS::setPose, mls.cpp:408-479
    dt("raycast>", m->grid_s);
    SLAM_TRY(slam_grid_raycast_scans_dev(m->grid, b.d_pts, b.d_off, n_scans, n_points, b.d_R, b.d_t, (slam_stream_t)m->grid_s));
    dt("raycast<", m->grid_s);
    ++m->chunks;
    if (m->prm.merge_every && m->chunks % m->prm.merge_every == 0) {
        if (m->merge_begin) {
            int lo = 0, hi = -1;
            SLAM_TRY(m->merge_begin(m->merge_ctx, m->grid, (slam_stream_t)m->grid_s, &lo, &hi));
            m->merge_pending = true;
        } else {
            SLAM_TRY(slam_grid_finalize(m->grid, (slam_stream_t)m->grid_s)); // one GPU: the periodic part is the occupancy output
            ++m->merges;
        }
    }
    MAP_HIP(hipEventRecord(b.mapped, m->grid_s));
    xt("grid enqueued");
    b.busy = true;
    b.n_scans = n_scans;
    m->next = (s + 1) % m->n_slots;
    if (slot_out) *slot_out = s;
    return SLAM_OK;
}

int slam_mapper_wait(slam_mapper_t *m, int slot, double *R_out, double *t_out)
{
    SLAM_REQUIRE(m && slot >= 0 && slot < m->n_slots, SLAM_E_INVALID, "slam_mapper_wait: bad arguments");
    Slot &b = m->slot[slot];
    if (!b.busy) return SLAM_OK;
    xt("wait");
    MAP_HIP(hipEventSynchronize(b.registered));
    xt("wait: registered"); // the poses are in the slot's pinned buffers (slam_mapper_push)
    if (R_out) memcpy(R_out, b.h_R, 32 * (size_t)b.n_scans);
    if (t_out) memcpy(t_out, b.h_t, 16 * (size_t)b.n_scans);
    MAP_HIP(hipEventSynchronize(b.mapped)); // the slot's device buffers are free again
    xt("wait: mapped");
    b.busy = false;
    return SLAM_OK;
}

int slam_mapper_finish(slam_mapper_t *m)
{
    SLAM_REQUIRE(m, SLAM_E_INVALID, "null handle");
    xt_dump();
    SLAM_TRY(adopt_build(m, true));
    SLAM_TRY(finish_merge(m));
    if (m->merge_begin) { // whatever was added since the last merge
        int lo = 0, hi = -1;
        SLAM_TRY(m->merge_begin(m->merge_ctx, m->grid, (slam_stream_t)m->grid_s, &lo, &hi));
        m->merge_pending = true;
        SLAM_TRY(finish_merge(m));
    } else {
        SLAM_TRY(slam_grid_finalize(m->grid, (slam_stream_t)m->grid_s));
    }
    MAP_HIP(hipStreamSynchronize(m->copy));
    MAP_HIP(hipStreamSynchronize(m->icp_s[0]));
    MAP_HIP(hipStreamSynchronize(m->icp_s[1]));
    MAP_HIP(hipStreamSynchronize(m->post_s));
    MAP_HIP(hipStreamSynchronize(m->grid_s));
    return SLAM_OK;
}

int slam_mapper_grid(slam_mapper_t *m, slam_grid_t **grid)
{
    SLAM_REQUIRE(m && grid, SLAM_E_INVALID, "slam_mapper_grid: bad arguments");
    *grid = m->grid;
    return SLAM_OK;
}

int slam_mapper_target(slam_mapper_t *m, slam_icp_t **icp)
{
    SLAM_REQUIRE(m && icp, SLAM_E_INVALID, "slam_mapper_target: bad arguments");
    *icp = m->target;
    return SLAM_OK;
}

int slam_mapper_stats(slam_mapper_t *m, long *chunks, long *merges, long *rebuilds, double *rebuild_ms, int last_merge_rows[2])
{
    SLAM_REQUIRE(m, SLAM_E_INVALID, "null handle");
    if (chunks) *chunks = m->chunks;
    if (merges) *merges = m->merges;
    if (rebuilds) *rebuilds = m->rebuilds;
    if (rebuild_ms) *rebuild_ms = m->rebuild_ms;
    if (last_merge_rows) last_merge_rows[0] = m->last_rows[0], last_merge_rows[1] = m->last_rows[1];
    return SLAM_OK;
}

} // extern "C"
