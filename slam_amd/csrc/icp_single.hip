// icp_single.hip -- few scans at a time, each spread over many workgroups of ONE persistent launch.
//
// The reference matches scan by scan (scan_registration.cpp:139-159 -> CCICP::doICPMatch ->
// IcpPointToPoint::fit, icp.cpp:80-122): one scan, one model, <= max_iter dependent steps.  One workgroup per
// scan (icp.hip) would leave 255 of 256 CUs idle for that, and with a model too large for LDS every query is a
// chain of dependent loads from L2.  Here a scan's points are dealt over up to n_cu workgroups that stay
// resident for all iterations:
//   * a query is searched by 64 lanes (scans of up to 4096 points; 16 lanes beyond): 8 row groups of 8 lanes
//     fetch the rows of a ring level together (nn_search_rows), so a level is two or three dependent round
//     trips instead of two per row -- the north-star "one wavefront per scan point" is right for THIS case;
//   * per iteration every workgroup leaves its nine sums as 18 self-validating 8-byte granules
//     {iteration + 1, half a double} (sc1 stores, no fence: cdna_hip_programming.md guideline 16, form R2);
//     wavefront 0 of EVERY workgroup polls all granules of its scan (sc1 loads), adds them in a fixed order and
//     solves -- so every workgroup holds the same new pose bit for bit, there is no broadcast and no second
//     exchange: one all-gather per iteration is the only synchronisation, about 3 us on this chip;
//   * granules are double-buffered by iteration parity: a workgroup can run at most one exchange ahead of the
//     slowest (it needs everyone's granules of iteration k to start k + 1);
//   * a model that fits LDS is copied there by every workgroup once (the cell index of icp.hip, 16-bit starts).
// All workgroups of a scan must be resident together (the grid is at most one workgroup per CU), and nothing guarantees
// that: a second spread launch (another handle, another process on the GPU) or a persistent kernel that holds the CUs
// can leave each launch with a share of its workgroups, spinning on the ones that never start.  So every spin is bounded
// by the wall clock (s_memrealtime) -- the FIRST exchange, which is what proves the scan's workgroups resident together,
// by milliseconds -- a workgroup that gives up raises the scan's abort word, which every other spin of the scan looks
// at, and workgroup 0 of a scan that did not finish flags it in redo[]: the caller enqueues the one-workgroup-per-scan
// form behind this launch for exactly the flagged scans (icp.hip: no dependency between workgroups there), from the
// initial pose the spread form has left untouched.  A fit always returns a pose (Icp::fit, icp.cpp:80-114).
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "icp_search.hpp"

using namespace slam;
using namespace slam::icp;

namespace {

constexpr int      kSB = 512, kSW = kSB / 64;      // threads / wavefronts of a spread workgroup (256 VGPRs per lane: no spills)
constexpr int      kDenseCell = 64;                // points in the fullest cell from which a query gets 64 lanes instead of 16
constexpr int      kGranPerWg = 2 * kNumAcc;       // 18 granules: hi and lo half of nine doubles
constexpr int      kFirstWaitUs = 5000;                 // the first exchange of a scan (slam_icp_params::spread_wait_us)
constexpr unsigned long long kSpinTicks = 200000000ull; // 2 s: later exchanges (everyone was resident at the first)

typedef unsigned long long __attribute__((address_space(1))) gu64;

__device__ inline void granule_store(unsigned long long *g, unsigned tag, unsigned value)
{
    __hip_atomic_store((gu64 *)g, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline unsigned long long granule_load(const unsigned long long *g)
{
    return __hip_atomic_load((gu64 *)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// workgroups that share scan `n`'s points when a query takes G lanes and the launch has `parts` per scan
__host__ __device__ inline int active_parts(int n, int G, int parts)
{
    const int want = (int)(((long long)n * G + kSB - 1) / kSB);
    return want < 1 ? 1 : (want > parts ? parts : want);
}

// The iterations of one scan as seen by one of its workgroups.  Returns false when an exchange timed out.
template <int G, typename StartT, int MODE>
__device__ inline bool spread_iterations(const ModelView &mv, const FitArgs &fa, const IndexPtrs<StartT> &ix,
                                         unsigned char *smem, unsigned long long *gran /* [2][parts][18] of this scan */,
                                         int *abort_word, unsigned long long first_ticks, float2 *qstate, int qcap, int parts, int s,
                                         int off, int n, int nga, FitState &fs)
{
    double   *partial = reinterpret_cast<double *>(smem);   // [kSW][kNumAcc]
    double   *bc = partial + kSW * kNumAcc;                  // [8] new pose, delta, n_corr
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int part = blockIdx.x;
    const int n_act = active_parts(n, G, parts);
    constexpr int kPerPass = kSB / G;
    // this workgroup's points: passes of kPerPass consecutive points, dealt round-robin over the active workgroups
    double r00 = fs.r00, r01 = fs.r01, r10 = fs.r10, r11 = fs.r11, t0 = fs.t0, t1 = fs.t1, delta = fs.delta;
    int    iters = 0, n_corr = 0;
    bool   ok = true;
    float  move_r = 0.0f, move_t = 0.0f; // how far the last step moved a query at most: move_r * (|x| + |y|) + move_t
    for (int iter = 0; iter < fa.max_iter; ++iter) {
        double acc[kNumAcc];
#pragma unroll
        for (int k = 0; k < kNumAcc; ++k) acc[k] = 0.0;
        const Pose T = {r00, r01, r10, r11, t0, t1};
        if (fa.step_pose && part == 0 && tid == 0) {
            double *sp = fa.step_pose + 6 * (size_t)s;
            sp[0] = r00, sp[1] = r01, sp[2] = r10, sp[3] = r11, sp[4] = t0, sp[5] = t1;
        }
        for (int p0 = part * kPerPass; p0 < n; p0 += n_act * kPerPass) {
            const int p = p0 + tid / G, lig = tid % G;
            if (p < n) {
                const int cls = MODE == SLAM_ICP_P2L ? 1 : (p < nga ? 0 : 1); // a point-to-line model is one class
                if (MODE == SLAM_ICP_P2L || mv.n_cls[cls] > 3) { // icpPointToPoint.cpp:59,93
                    float         qx, qy;
                    const double2 P = fa.pts[off + p];
                    transform_query(T, P, qx, qy);
                    // what last iteration's search of this scene point left: its neighbour and the radius it proved empty
                    const bool stateful = off + p < qcap;
                    Seed       seed = {-1, 0.0f};
                    if (stateful && iter > 0) {
                        const float2 st = qstate[off + p];
                        seed.empty = st.x;
                        seed.pos = __float_as_int(st.y);
                    }
                    const float move = move_r * (fabsf((float)P.x) + fabsf((float)P.y) + 1.0e-3f) + move_t;
                    float       empty = 0.0f;
                    const double gate = MODE == SLAM_ICP_P2L ? (double)INFINITY : fa.indist; // icpPointToPlane.cpp:55-77: no gate
                    const Best   b = nn_search_rows<G, StartT>(ix, mv, cls, qx, qy, lig, gate, seed, move, empty);
                    if (lig == 0) {
                        if (MODE == SLAM_ICP_P2L) {
                            if (b.pos >= 0)
                                add_p2l(ix.pts[mv.base[1] + b.pos], reinterpret_cast<const double2 *>(mv.normals)[b.oidx], qx, qy, acc);
                        } else if (b.pos >= 0 && (double)b.d < fa.indist) {
                            add_p2p<StartT>(ix, mv, cls, b, qx, qy, acc); // :76
                        }
                        if (stateful) qstate[off + p] = make_float2(empty, __int_as_float(b.pos));
                    }
                }
            }
        }
        {
            const double v8 = wave_sum8(acc), v9 = wave_sum(acc[8]);
            double      *my = partial + wave * kNumAcc;
            if ((lane & 7) == 0) my[lane >> 3] = v8;
            if (lane == 0) my[8] = v9;
        }
        __syncthreads();
        if (wave == 0) {
            // this workgroup's nine sums (fixed order over its wavefronts) go out as granules ...
            const unsigned      tag = (unsigned)iter + 1u;
            unsigned long long *gbuf = gran + (size_t)(iter & 1) * parts * kGranPerWg;
            if (lane < kNumAcc) {
                double mine = 0.0;
                for (int w = 0; w < kSW; ++w) mine += partial[w * kNumAcc + lane];
                unsigned long long *g = gbuf + (size_t)part * kGranPerWg + 2 * lane;
                granule_store(g, tag, (unsigned)__double2hiint(mine));
                granule_store(g + 1, tag, (unsigned)__double2loint(mine));
            }
            // ... and every workgroup gathers all of them: lane l takes workgroups l, l + 64, ... in turn
            double                   tot[kNumAcc];
            const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
#pragma unroll
            for (int k = 0; k < kNumAcc; ++k) tot[k] = 0.0;
            for (int w = lane; w < n_act; w += 64) {
                const unsigned long long *g = gbuf + (size_t)w * kGranPerWg;
                unsigned long long        v[kGranPerWg];
                for (;;) {
                    bool all = true;
#pragma unroll
                    for (int k = 0; k < kGranPerWg; ++k) {
                        v[k] = granule_load(g + k);
                        all &= (unsigned)(v[k] >> 32) == tag;
                    }
                    if (all && !(iter == 0 && first_ticks == 0)) break; // (spread_wait_us < 0: every scan is handed over, for the tests)
                    if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                        __builtin_amdgcn_s_memrealtime() - t_begin >= (iter == 0 ? first_ticks : kSpinTicks)) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
#pragma unroll
                for (int k = 0; k < kNumAcc; ++k) tot[k] += __hiloint2double((int)(unsigned)v[2 * k], (int)(unsigned)v[2 * k + 1]);
            }
            ok = __all(ok);
            if (!ok && lane == 0) __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // the others stop spinning
            const double v8 = wave_sum8(tot), v9 = wave_sum(tot[8]);
            double       S[kNumAcc];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                S[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v8), 8 * k),
                                        __builtin_amdgcn_readlane(__double2loint(v8), 8 * k));
            S[8] = v9;
            double o[6] = {r00, r01, r10, r11, t0, t1};
            int    nc_out = n; // point-to-line: every template point has a correspondence
            const double d_out = MODE == SLAM_ICP_P2L ? p2l_step(S, o) : p2p_step(S, mv, o, nc_out);
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 6; ++k) bc[k] = o[k];
                bc[6] = d_out;
                bc[7] = ok ? (double)nc_out : -1.0;
            }
        }
        __syncthreads();
        if (bc[7] < 0.0) return false; // uniform: an exchange gave up
        {
            // |q_new - q_old| <= |R_new - R_old|_F |p| + |t_new - t_old|, rounded up generously (the float store of q
            // adds an ulp of the coordinate: covered by the lattice margin the search subtracts)
            const double n00 = uniform(bc[0]), n01 = uniform(bc[1]), n10 = uniform(bc[2]), n11 = uniform(bc[3]);
            const double n4 = uniform(bc[4]), n5 = uniform(bc[5]);
            const double dr = sqrt((n00 - r00) * (n00 - r00) + (n01 - r01) * (n01 - r01) + (n10 - r10) * (n10 - r10) +
                                   (n11 - r11) * (n11 - r11));
            const double dt = sqrt((n4 - t0) * (n4 - t0) + (n5 - t1) * (n5 - t1));
            move_r = (float)dr * 1.0001f + 1.0e-7f;
            move_t = (float)dt * 1.0001f + 1.0e-7f;
            r00 = n00, r01 = n01, r10 = n10, r11 = n11, t0 = n4, t1 = n5;
        }
        delta = uniform(bc[6]);
        n_corr = (int)uniform(bc[7]);
        ++iters;
        if (fa.trace && part == 0 && tid == 0) {
            double *tr = fa.trace + ((size_t)s * fa.max_iter + iter) * 8;
            tr[0] = r00, tr[1] = r01, tr[2] = r10, tr[3] = r11, tr[4] = t0, tr[5] = t1;
            tr[6] = delta;
            tr[7] = (double)n_corr;
        }
        if (delta < fa.min_delta) break; // icp.cpp:119-121
        // (bc and partial are next written behind the next iteration's first barrier, which every reader of this
        // iteration's values reaches only after reading them)
    }
    fs.r00 = r00, fs.r01 = r01, fs.r10 = r10, fs.r11 = r11, fs.t0 = t0, fs.t1 = t1;
    fs.delta = delta;
    fs.iters = iters;
    fs.n_corr = n_corr;
    return true;
}

// grid (parts, n_scans); a workgroup whose scan does not need it exits at once
template <typename StartT, bool LDS, int MODE>
__global__ __launch_bounds__(kSB) void icp_fit_spread_kernel(ModelView mv, FitArgs fa, unsigned long long *gran, int *flags /* [2][n_scans] abort | redo */,
                                                                unsigned long long first_ticks, float2 *qstate, int qcap, int wide_max)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.y, parts = (int)gridDim.x, part = blockIdx.x;
    const int off = fa.scan_off[s];
    const int n = fa.scan_off[s + 1] - off;
    const int nga = fa.scan_nga[s];
    if (n < 5 || fa.max_iter <= 0) { // icp.cpp:100-103: R, t untouched -- the pose the fit was given
        if (blockIdx.x == 0 && threadIdx.x < 6 && fa.R0 != fa.R) {
            if (threadIdx.x < 4)
                fa.R[4 * s + threadIdx.x] = fa.R0[4 * s + threadIdx.x];
            else
                fa.t[2 * s + threadIdx.x - 4] = fa.t0[2 * s + threadIdx.x - 4];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && fa.result) {
            fa.result[s].iters = 0;
            fa.result[s].n_corr = 0;
            fa.result[s].delta = 0.0;
        }
        return;
    }
    const bool wide = n <= wide_max; // 64 lanes per query up to here, 16 beyond
    if (part >= active_parts(n, wide ? 64 : 16, parts)) return;
    const unsigned char *base = mv.blob;
    if (LDS) {
        const uint4 *src = reinterpret_cast<const uint4 *>(mv.blob);
        uint4       *dst = reinterpret_cast<uint4 *>(smem + kScratchBytes);
        for (unsigned i = threadIdx.x; i < mv.blob_bytes / 16u; i += kSB) dst[i] = src[i];
        base = smem + kScratchBytes;
        __syncthreads();
    }
    const IndexPtrs<StartT> ix = make_ptrs<StartT>(base, mv);
    FitState                fs;
    fs.r00 = uniform(fa.R0[4 * s + 0]);
    fs.r01 = uniform(fa.R0[4 * s + 1]);
    fs.r10 = uniform(fa.R0[4 * s + 2]);
    fs.r11 = uniform(fa.R0[4 * s + 3]);
    fs.t0 = uniform(fa.t0[2 * s + 0]);
    fs.t1 = uniform(fa.t0[2 * s + 1]);
    fs.delta = 0.0;
    fs.iters = 0;
    fs.n_corr = 0;
    fs.hand_over = false;
    unsigned long long *g = gran + (size_t)s * 2 * parts * kGranPerWg;
    const bool          ok = wide ? spread_iterations<64, StartT, MODE>(mv, fa, ix, smem, g, flags + s, first_ticks, qstate, qcap, parts, s, off, n, nga, fs)
                                  : spread_iterations<16, StartT, MODE>(mv, fa, ix, smem, g, flags + s, first_ticks, qstate, qcap, parts, s, off, n, nga, fs);
    if (part == 0 && threadIdx.x == 0) {
        // workgroup 0 decides: if IT saw every exchange through, the pose is complete whatever the others did afterwards
        if (!ok) flags[gridDim.y + s] = 1; // redo: the one-workgroup form takes this scan, from the pose left untouched here
        if (ok) {
            fa.R[4 * s + 0] = fs.r00;
            fa.R[4 * s + 1] = fs.r01;
            fa.R[4 * s + 2] = fs.r10;
            fa.R[4 * s + 3] = fs.r11;
            fa.t[2 * s + 0] = fs.t0;
            fa.t[2 * s + 1] = fs.t1;
        }
        if (fa.result) {
            fa.result[s].iters = ok ? fs.iters : -1; // -1: an exchange gave up; overwritten by the launch that redoes the scan
            fa.result[s].n_corr = fs.n_corr;
            fa.result[s].delta = fs.delta;
        }
    }
}

} // namespace

namespace slam {
namespace icp {

// How many scans at most go through the spread form on a chip with n_cu CUs (a quarter of them: at least four
// workgroups per scan), and how many workgroups each scan gets
int spread_parts(int n_scans, int n_cu) { return std::max(1, n_cu / std::max(n_scans, 1)); }

// Spread launches of one process run one after the other on the device, whatever streams and handles they come from: two
// of them in flight would each hold a share of the CUs and wait for workgroups that cannot start (they would give up after
// kFirstTicks and be redone -- correct, and milliseconds late).  The order costs the host an event wait and an event record per
// launch; launches of other processes are beyond it, the give-up path is what covers those.
namespace {
std::mutex  g_spread_mu;
hipEvent_t  g_spread_done[16] = {};
} // namespace

int launch_fit_spread(slam_icp *h, const FitArgs &fa, int n_scans, hipStream_t st, const int **redo_flags)
{
    int dev = 0, n_cu = 0;
    SLAM_HIP(hipGetDevice(&dev));
    SLAM_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const int    parts = spread_parts(n_scans, std::max(n_cu, 1));
    const size_t gran_bytes = sizeof(unsigned long long) * 2 * (size_t)parts * kGranPerWg * (size_t)n_scans;
    const size_t flag_bytes = sizeof(int) * 2 * (size_t)n_scans;
    SLAM_TRY(h->w_single.reserve(gran_bytes + flag_bytes));
    SLAM_HIP(hipMemsetAsync(h->w_single.p, 0, gran_bytes + flag_bytes, st)); // tag 0 = nothing published; no abort, no redo
    unsigned long long *gran = static_cast<unsigned long long *>(h->w_single.p);
    int                *flags = reinterpret_cast<int *>(static_cast<unsigned char *>(h->w_single.p) + gran_bytes);
    *redo_flags = flags + n_scans;
    h->d_last_redo = flags + n_scans;
    std::lock_guard<std::mutex> lk(g_spread_mu);
    hipEvent_t &done = g_spread_done[dev & 15];
    // (a stream that is being captured into a hipGraph may neither wait for an event of uncaptured work nor lend its own to
    // other streams: a captured spread launch is ordered by its graph alone -- if it ever meets another one, the redo path covers it)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    const bool ordered = cap == hipStreamCaptureStatusNone;
    if (!done)
        SLAM_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    else if (ordered)
        SLAM_HIP(hipStreamWaitEvent(st, done, 0)); // behind the spread launch before this one, on whatever stream it went
    // per scene point, what its last search left for the next (positions beyond the buffer search unseeded)
    const int qcap = std::max(h->spread_points_hint, 1 << 16);
    SLAM_TRY(h->w_state.reserve(sizeof(float2) * (size_t)qcap));
    float2    *qstate = static_cast<float2 *>(h->w_state.p);
    const dim3 grid(parts, n_scans);
    const int                wait_us = h->prm.spread_wait_us > 0 ? h->prm.spread_wait_us : (h->prm.spread_wait_us < 0 ? 0 : kFirstWaitUs);
    const unsigned long long first_ticks = 100ull * (unsigned long long)wait_us; // s_memrealtime counts at 100 MHz
    // Lanes per query.  A 2-D map holds a handful of points per cell: 16 lanes take a query's cells in one or two
    // steps, a 1081-point scan is 34 workgroups, and the exchange between them is the larger part of an iteration --
    // fewer workgroups, cheaper exchange (0.17 against 0.22 ms for 20 iterations).  A lidar cloud holds hundreds of
    // points per cell near the sensor: there a query wants the whole wavefront (config 3: 0.46 against 1.03 ms).
    int        wide_max = h->max_cell_points > kDenseCell ? 4096 : 0;
#ifdef SLAM_MEASURE
    if (const char *e = getenv("SLAM_SPREAD_WIDE_MAX")) wide_max = atoi(e);
#endif
    const bool p2l = h->prm.mode == SLAM_ICP_P2L; // the nine sums and the solve differ, nothing else
    if (h->in_lds) {
        auto kern = p2l ? icp_fit_spread_kernel<uint16_t, true, SLAM_ICP_P2L> : icp_fit_spread_kernel<uint16_t, true, SLAM_ICP_P2P>;
        SLAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        hipLaunchKernelGGL(kern, grid, dim3(kSB), h->lds_bytes, st, h->mv, fa, gran, flags, first_ticks, qstate, qcap, wide_max);
    } else if (h->start32) {
        auto kern = p2l ? icp_fit_spread_kernel<uint32_t, false, SLAM_ICP_P2L> : icp_fit_spread_kernel<uint32_t, false, SLAM_ICP_P2P>;
        hipLaunchKernelGGL(kern, grid, dim3(kSB), kScratchBytes, st, h->mv, fa, gran, flags, first_ticks, qstate, qcap, wide_max);
    } else {
        auto kern = p2l ? icp_fit_spread_kernel<uint16_t, false, SLAM_ICP_P2L> : icp_fit_spread_kernel<uint16_t, false, SLAM_ICP_P2P>;
        hipLaunchKernelGGL(kern, grid, dim3(kSB), kScratchBytes, st, h->mv, fa, gran, flags, first_ticks, qstate, qcap, wide_max);
    }
    SLAM_HIP(hipGetLastError());
    if (ordered) SLAM_HIP(hipEventRecord(done, st));
    return SLAM_OK;
}

} // namespace icp
} // namespace slam
