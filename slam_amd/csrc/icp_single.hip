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

// Measurement build only (SLAM_SPREAD_STAMPS=1): per workgroup of scan 0 and iteration, 100 MHz wall-clock stamps at
// [0] the iteration's start, [1] wavefront 0 through its searches, [2] the workgroup through them (first barrier), [3] the
// exchange complete, [4] the solve done, [5] the new pose known to the workgroup (second barrier), [6] staging a tile (tile form)
#ifdef SLAM_MEASURE
constexpr int kSpreadStampSlots = 16; // ... [7] searches through L2, [8..10] points / cell-table entries / rows of the tile staged, [11] the staging's ticks
#define SPREAD_STAMP(k)                                                                                                  \
    do {                                                                                                                 \
        if (sstamps && s == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 0)                                     \
            sstamps[((size_t)blockIdx.x * fa.max_iter + iter) * kSpreadStampSlots + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define SPREAD_STAMP(k) do { } while (0)
#endif

// the 18 granules of one workgroup (144 bytes, 16-byte aligned) as NINE 16-byte loads of the same scope as granule_load, issued
// together and waited for once: the gather is bound by the loads a lane has to issue per poll (twice the loads, tried for lanes that
// hold two workgroups, cost config 3 four per cent), and a granule validates itself -- a pair torn between its halves is two granules
__device__ inline void granule_load18(const unsigned long long *g, unsigned long long v[18])
{
    uint4 r0, r1, r2, r3, r4, r5, r6, r7, r8;
    asm volatile("global_load_dwordx4 %0, %9, off sc1\n\t"
                 "global_load_dwordx4 %1, %9, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %9, off offset:32 sc1\n\t"
                 "global_load_dwordx4 %3, %9, off offset:48 sc1\n\t"
                 "global_load_dwordx4 %4, %9, off offset:64 sc1\n\t"
                 "global_load_dwordx4 %5, %9, off offset:80 sc1\n\t"
                 "global_load_dwordx4 %6, %9, off offset:96 sc1\n\t"
                 "global_load_dwordx4 %7, %9, off offset:112 sc1\n\t"
                 "global_load_dwordx4 %8, %9, off offset:128 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(r8)
                 : "v"(g)
                 : "memory");
    const uint4 r[9] = {r0, r1, r2, r3, r4, r5, r6, r7, r8};
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        v[2 * k] = ((unsigned long long)r[k].y << 32) | r[k].x;
        v[2 * k + 1] = ((unsigned long long)r[k].w << 32) | r[k].z;
    }
}
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

// One exchange, by wavefront 0 of a workgroup: its nine sums (fixed order over its wavefronts) go out as granules, the granules of all
// the scan's workgroups come in, the step is solved (the same bits in every workgroup) and left in bc[0..7] = new pose, delta,
// correspondences (-1: the exchange gave up).  Returns false when it gave up.
template <int MODE>
__device__ inline bool spread_exchange(const ModelView &mv, const FitArgs &fa, const double *partial, int rows, double *bc, unsigned long long *gran,
                                       int *abort_word, unsigned long long first_ticks, int iter, int part, int parts, int n_act, int n, int s,
                                       const double pose[6], long long *sstamps)
{
    const int lane = (int)threadIdx.x & 63;
    bool      ok = true;
    // this workgroup's nine sums go out as granules ...
    const unsigned      tag = fa.spread_tag + (unsigned)iter + 1u;
    unsigned long long *gbuf = gran + (size_t)(iter & 1) * parts * kGranPerWg;
    if (lane < kNumAcc) {
        double mine = 0.0;
        for (int w = 0; w < rows; ++w) mine += partial[w * kNumAcc + lane];
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
            static_assert(kGranPerWg == 18, "granule_load18");
            granule_load18(g, v);
#pragma unroll
            for (int k = 0; k < kGranPerWg; ++k) all &= (unsigned)(v[k] >> 32) == tag;
            if (all && !(iter == 0 && first_ticks == 0)) break; // (spread_wait_us < 0: every scan is handed over, for the tests)
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)fa.spread_tag ||
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
    if (!ok && lane == 0) __hip_atomic_store(abort_word, (int)fa.spread_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // the others stop spinning
    SPREAD_STAMP(3);
    const double v8 = wave_sum8(tot), v9 = wave_sum(tot[8]);
    double       S[kNumAcc];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        S[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v8), 8 * k),
                                __builtin_amdgcn_readlane(__double2loint(v8), 8 * k));
    S[8] = v9;
    double o[6] = {pose[0], pose[1], pose[2], pose[3], pose[4], pose[5]};
    int    nc_out = n; // point-to-line: every template point has a correspondence
    const double d_out = MODE == SLAM_ICP_P2L ? p2l_step(S, o) : p2p_step(S, mv, o, nc_out);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) bc[k] = o[k];
        bc[6] = d_out;
        bc[7] = ok ? (double)nc_out : -1.0;
    }
    SPREAD_STAMP(4);
    return ok;
}

// The iterations of one scan as seen by one of its workgroups.  Returns false when an exchange timed out.
template <int G, typename StartT, int MODE>
__device__ inline bool spread_iterations(const ModelView &mv, const FitArgs &fa, const IndexPtrs<StartT> &ix,
                                         unsigned char *smem, unsigned long long *gran /* [2][parts][18] of this scan */,
                                         int *abort_word, unsigned long long first_ticks, float2 *qstate, int qcap, int parts, int s,
                                         int off, int n, int nga, FitState &fs, long long *sstamps)
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
    float  move_r = 0.0f, move_t = 0.0f; // how far the last step moved a query at most: move_r * (|x| + |y|) + move_t
    for (int iter = 0; iter < fa.max_iter; ++iter) {
        double acc[kNumAcc];
#pragma unroll
        for (int k = 0; k < kNumAcc; ++k) acc[k] = 0.0;
        const Pose T = {r00, r01, r10, r11, t0, t1};
        SPREAD_STAMP(0);
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
        SPREAD_STAMP(1);
        {
            const double v8 = wave_sum8(acc), v9 = wave_sum(acc[8]);
            double      *my = partial + wave * kNumAcc;
            if ((lane & 7) == 0) my[lane >> 3] = v8;
            if (lane == 0) my[8] = v9;
        }
        __syncthreads();
        SPREAD_STAMP(2);
        if (wave == 0) {
            const double pose[6] = {r00, r01, r10, r11, t0, t1};
            spread_exchange<MODE>(mv, fa, partial, kSW, bc, gran, abort_word, first_ticks, iter, part, parts, n_act, n, s, pose, sstamps);
        }
        __syncthreads();
        SPREAD_STAMP(5);
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

// a ballot restricted to the G lanes of this lane's group, bit 0 = the group's first lane
template <int G>
__device__ inline unsigned long long group_bits(unsigned long long m, int group_base)
{
    if constexpr (G < 64)
        return (m >> group_base) & ((1ull << G) - 1ull);
    else
        return m;
}

// An unseeded search that looks at the inlier gate's square first.  The ring search finds a neighbour that is near in two or three
// levels and pays five levels of two dependent round trips each for a query that has none within the gate (a scene cloud seen from the
// next pose: half of config 3's queries, every iteration they are searched in) -- here the extents of ALL rows of the square of
// ceil(gate radius / pitch) cells around the query arrive in ONE round trip: nothing there = no correspondence (icpPointToPoint.cpp:76)
// and a radius proved empty; a few points there = their scan is the whole search (every point within the gate's radius of the query is
// in the square).  Returns false for a square with many points (a neighbour is near: the ring search) and for exact ties.
template <int G, typename StartT>
__device__ inline bool nn_search_gate(const IndexPtrs<StartT> &ix, const ModelView &mv, int cls, float qx, float qy, int lig, double gate, float gate_r,
                                      Best &b, float &empty_out)
{
    constexpr int  kRows = G == 64 ? 2 : 4; // rows of the square per lane
    const Lattice &L = mv.lat;
    const StartT  *start = ix.start[cls];
    const float2  *pts = ix.pts + mv.base[cls];
    const StartT  *oidx = ix.oidx + mv.base[cls];
    b.d = FLT_MAX, b.pos = -1, b.oidx = 0xffffffffu;
    empty_out = 0.0f;
    if (mv.n_cls[cls] <= 0) return true;
    // (two cells beyond the gate: the radius an empty square proves is what the point's certificate lives on while the fit moves it)
    const int R = (int)ceilf((gate_r + L.margin) * L.inv_h) + 2;
    if (2 * R + 1 > kRows * G) return false;
    const float fx = (qx - L.x0) * L.inv_h, fy = (qy - L.y0) * L.inv_h;
    const int   cx = clampi(ifloor(fx), 0, L.nx - 1), cy = clampi(ifloor(fy), 0, L.ny - 1);
    const int   x_lo = max(cx - R, 0), x_hi = min(cx + R, L.nx - 1), y_lo = max(cy - R, 0), y_hi = min(cy + R, L.ny - 1);
    const int   group_base = ((int)threadIdx.x & 63) & ~(G - 1);
    int         a[kRows], e[kRows], mine = 0;
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        const int y = y_lo + lig + j * G;
        a[j] = e[j] = 0;
        if (y <= y_hi) {
            a[j] = (int)start[y * L.nx + x_lo];
            e[j] = (int)start[y * L.nx + x_hi + 1];
        }
        mine += e[j] - a[j];
    }
    int total = mine;
#pragma unroll
    for (int o = 1; o < G; o <<= 1) total += __shfl_xor(total, o);
    const float edge = (float)R * L.h - L.margin; // every point outside the square is farther than this
    if (total == 0) {
        empty_out = edge;
        return true;
    }
    if (total > 6 * G) return false;
    bool tie = false;
#pragma unroll
    for (int j = 0; j < kRows; ++j) {
        unsigned long long rows = group_bits<G>(__ballot(e[j] > a[j]), group_base);
        while (rows) {
            const int src = __builtin_ctzll(rows);
            rows &= rows - 1;
            const int A = G == 64 ? __builtin_amdgcn_readlane(a[j], src) : __shfl(a[j], src, G);
            const int E = G == 64 ? __builtin_amdgcn_readlane(e[j], src) : __shfl(e[j], src, G);
            scan_range_rt<StartT, false>(b, tie, pts, oidx, A, E, lig, G, qx, qy);
        }
    }
    group_min_lean<G>(b, tie);
    if (tie) return false;
    b.oidx = b.pos >= 0 ? (unsigned)oidx[b.pos] : 0xffffffffu;
    // within the gate the best of the square is the nearest point of the class; beyond it, it is a seed and a bound
    const float dn = b.d < FLT_MAX ? __fsqrt_rn(b.d) * 0.999999f : 1.0e30f;
    empty_out = (double)b.d < gate ? dn : fminf(dn, edge);
    return true;
}

// ---- The tile form (round 6): a scan against a model that does not fit LDS -- the reference's own operating point: one cloud per
// callback (scan_registration.cpp:139-159), one fit per match (icpTools.cpp:187-188), up to 19 999 points per class
// (icpTools.h:21).  In the form above every search of every iteration is a chain of dependent loads from L2, a scene point's
// state goes through global memory, and the iteration waits for the slowest query of the slowest workgroup (config 3: 18 of
// 21.5 us per iteration, a query against cells of hundreds of wall points; the mean workgroup needs 7).  Here
//   * every workgroup ranks the scan's points along a Morton curve of the lattice under the initial pose (class first) and
//     takes a contiguous run of ranks: its scene points are neighbours in space whatever order the caller's array has (a
//     voxel filter's output has none);
//   * its points and their search state -- last neighbour (position and coordinates), radius proved empty -- live in LDS
//     slots for the whole fit;
//   * after the first step it stages into LDS the TILE of the index its own queries can reach: per class the rectangle of
//     lattice cells around every query's disk {seed distance, capped by the inlier gate} dilated by a multiple of the
//     query's last move -- per lattice row one contiguous span of the sorted array, a cell table of local 16-bit starts, the
//     row's offset back to global positions;
//   * a seeded search whose disk lies inside the rectangle runs on the tile alone: every cell the disk touches is there, so
//     the result is the global search's (a query that has left its rectangle, has no seed yet, or meets an exact distance tie
//     takes the search above on the index in L2, and the workgroup stages again before the next iteration, a bounded number
//     of times).  Exactness is the seeded search's: the seed is a model point, the nearest one is no farther, and the cells
//     of the disk are all visited.
constexpr int kTileMaxN = 2048;     // scene points of a scan in this form (every workgroup ranks all of them: n^2 / 512 compares per thread)
constexpr int kTileSlots = 128;     // scene points of one workgroup
constexpr int kTilePtsMax = 14336;  // model points of a workgroup's tiles, both classes
constexpr int kTileStartMax = 12288; // cell-table entries, both classes
constexpr int kTileRowsMax = 512;   // lattice rows, both classes
constexpr int kTileRestage = 4;     // stagings after the first

struct TileLds {
    double         *partial, *bc;
    int            *hdr;      // [32]
    double2        *P;        // [kTileSlots] scene points
    float2         *sxy;      // last neighbour's coordinates
    int            *spos;     // ... and position in the sorted array (class-relative), -1: none
    float          *sempty;   // radius proved empty around scq (where the point's last search ran), 0: nothing known
    float2         *scq;
    int            *sidx;     // index of the scene point in the scan
    double2        *snrm;     // point-to-line: the normal of the neighbour at snpos
    int            *snpos;
    int            *rloff;    // [kTileRowsMax + 1] first local point of a tile row
    int            *rdelta;   // [kTileRowsMax] class-relative global position minus local position
    int            *rga;      // [kTileRowsMax] scratch of the staging
    unsigned short *tstart;   // [kTileStartMax]
    float2         *tpts;     // [kTilePtsMax]; the keys of the ranking before the first staging
};
constexpr unsigned kTileHeadBytes = (kSW * 4u * kNumAcc + 8u) * 8u + 64u; // partials, broadcast block
static_assert(kTileHeadBytes % 16u == 0, "alignment of the tile form's LDS blocks");
constexpr unsigned kTileLdsBytes = kTileHeadBytes + 128u + kTileSlots * (16u + 8u + 8u + 4u + 4u + 4u + 16u + 4u) + (kTileRowsMax + 1u + 3u) * 4u + kTileRowsMax * 8u +
                                   kTileStartMax * 2u + kTilePtsMax * 8u;
static_assert(kTileLdsBytes <= 160u * 1024u, "the tile form's LDS");

__device__ inline TileLds tile_lds(unsigned char *smem)
{
    TileLds t;
    unsigned char *p = smem;
    t.partial = reinterpret_cast<double *>(p); // [kSW * 4][kNumAcc]: a row per query lane (four per wavefront at 16 lanes per query)
    t.bc = t.partial + kSW * 4 * kNumAcc;
    p += kTileHeadBytes;
    t.hdr = reinterpret_cast<int *>(p);
    p += 128;
    t.P = reinterpret_cast<double2 *>(p);
    p += 16 * kTileSlots;
    t.snrm = reinterpret_cast<double2 *>(p);
    p += 16 * kTileSlots;
    t.sxy = reinterpret_cast<float2 *>(p);
    p += 8 * kTileSlots;
    t.scq = reinterpret_cast<float2 *>(p);
    p += 8 * kTileSlots;
    t.spos = reinterpret_cast<int *>(p);
    p += 4 * kTileSlots;
    t.sempty = reinterpret_cast<float *>(p);
    p += 4 * kTileSlots;
    t.sidx = reinterpret_cast<int *>(p);
    p += 4 * kTileSlots;
    t.snpos = reinterpret_cast<int *>(p);
    p += 4 * kTileSlots;
    t.rloff = reinterpret_cast<int *>(p);
    p += 4 * (kTileRowsMax + 4);
    t.rdelta = reinterpret_cast<int *>(p);
    p += 4 * kTileRowsMax;
    t.rga = reinterpret_cast<int *>(p);
    p += 4 * kTileRowsMax;
    t.tstart = reinterpret_cast<unsigned short *>(p);
    p += 2 * kTileStartMax;
    t.tpts = reinterpret_cast<float2 *>(p);
    return t;
}

// ... for a model whose index lies in LDS (behind kScratchBytes, icp_fit_spread_kernel): reduction block and header in the scratch in
// front of it, the slots behind it; the "tile" is the index itself (tstart = its cell tables, tpts = its points)
static_assert(kTileHeadBytes + 128u <= kScratchBytes, "the spread form's scratch holds the tile form's reduction block and header");
constexpr unsigned kSlotBytes = 16u + 16u + 8u + 8u + 4u + 4u + 4u + 4u;
__device__ inline TileLds tile_lds_whole(unsigned char *smem, const ModelView &mv, int slots)
{
    TileLds        t;
    unsigned char *p = smem;
    t.partial = reinterpret_cast<double *>(p);
    t.bc = t.partial + kSW * 4 * kNumAcc;
    p += kTileHeadBytes;
    t.hdr = reinterpret_cast<int *>(p);
    unsigned char *blob = smem + kScratchBytes;
    t.tstart = reinterpret_cast<unsigned short *>(blob);
    t.tpts = reinterpret_cast<float2 *>(blob + mv.off_pts);
    t.rloff = t.rdelta = t.rga = nullptr;
    p = blob + ((mv.blob_bytes + 15u) & ~15u);
    t.P = reinterpret_cast<double2 *>(p);
    p += 16 * slots;
    t.snrm = reinterpret_cast<double2 *>(p);
    p += 16 * slots;
    t.sxy = reinterpret_cast<float2 *>(p);
    p += 8 * slots;
    t.scq = reinterpret_cast<float2 *>(p);
    p += 8 * slots;
    t.spos = reinterpret_cast<int *>(p);
    p += 4 * slots;
    t.sempty = reinterpret_cast<float *>(p);
    p += 4 * slots;
    t.sidx = reinterpret_cast<int *>(p);
    p += 4 * slots;
    t.snpos = reinterpret_cast<int *>(p);
    return t;
}

// hdr words
enum { kHdrRect = 0 /* [2][4] x_lo y_lo x_hi y_hi */, kHdrMiss = 8, kHdrValid = 9 /* [2] */, kHdrRow0 = 11 /* [2] */, kHdrSoff = 13 /* [2] */, kHdrShift = 15 /* column-group shift of class 0 | of class 1 << 8 */, kHdrScan = 16 /* [8] wave totals */, kHdrDebug = 24 /* measurement build: searches through L2 this iteration */ };

struct TileCls {
    int X0, Y0, X1, Y1, row0, soff, pitch, valid, shift; // cell-table entry g of a row = the start of column X0 + (g << shift)
    int tp_off, tp_max, whole; // the class's points begin at tpts[tp_off], tp_max of them less one; whole: the tile is the class's whole
                               // index as it lies in LDS (a model that fits: local positions are global ones, no row offsets)
};

__device__ inline int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ inline TileCls tile_cls(const int *hdr, int c)
{
    TileCls t;
    t.X0 = uniform_i(hdr[kHdrRect + 4 * c + 0]);
    t.Y0 = uniform_i(hdr[kHdrRect + 4 * c + 1]);
    t.X1 = uniform_i(hdr[kHdrRect + 4 * c + 2]);
    t.Y1 = uniform_i(hdr[kHdrRect + 4 * c + 3]);
    t.row0 = uniform_i(hdr[kHdrRow0 + c]);
    t.soff = uniform_i(hdr[kHdrSoff + c]);
    t.shift = (uniform_i(hdr[kHdrShift]) >> (8 * c)) & 0xff;
    t.pitch = ((t.X1 - t.X0) >> t.shift) + 2;
    t.valid = uniform_i(hdr[kHdrValid + c]);
    t.tp_off = 0, t.tp_max = kTilePtsMax - 1, t.whole = 0;
    return t;
}

// 10 + 10 bits interleaved
__device__ inline unsigned morton10(unsigned x, unsigned y)
{
    auto spread = [](unsigned v) {
        v &= 0x3ffu;
        v = (v | (v << 8)) & 0x00ff00ffu;
        v = (v | (v << 4)) & 0x0f0f0f0fu;
        v = (v | (v << 2)) & 0x33333333u;
        v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    return spread(x) | (spread(y) << 1);
}

// exclusive prefix sum of one int per thread of the workgroup (kSB threads), total through *total; two barriers
__device__ inline int block_scan_excl(int v, int *wave_tot /* [kSW] in LDS */, int *total)
{
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    int       inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kSW; ++w) {
        const int t = wave_tot[w];
        base += w < wave ? t : 0;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// A span of the tile by `lanes` lanes of a group (`sub` = this lane's place among them): candidates at LOCAL positions; seed-aware
// like scan_range_rt (the running best may be met again: no tie).  bdelta follows the row of the lane's best.
__device__ inline void tile_scan(Best &b, bool &tie, int &bdelta, const float2 *tpts, int a, int e, int delta, int sub, int lanes, float qx, float qy)
{
    const int before = b.pos;
    int       i = a + sub;
    float     d2nd = FLT_MAX;
    for (; i + 3 * lanes < e; i += 4 * lanes) {
        const float2 m0 = tpts[i], m1 = tpts[i + lanes], m2 = tpts[i + 2 * lanes], m3 = tpts[i + 3 * lanes];
        const float  d0 = dist2(m0, qx, qy), d1 = dist2(m1, qx, qy), d2 = dist2(m2, qx, qy), d3 = dist2(m3, qx, qy);
        scan_step(b, d2nd, d0, i);
        scan_step(b, d2nd, d1, i + lanes);
        scan_step(b, d2nd, d2, i + 2 * lanes);
        scan_step(b, d2nd, d3, i + 3 * lanes);
    }
    for (; i < e; i += lanes) scan_step(b, d2nd, dist2(tpts[i], qx, qy), i);
    tie |= scan_tie(b, d2nd); // (the best this call was given is a point of another span, or the seed's distance one ulp up)
    bdelta = b.pos != before ? delta : bdelta;
}

// The seeded search on the tile, for a seed at squared distance d0 from the query.  0: done -- b.d, b.pos (class-relative GLOBAL
// position), m (the neighbour), empty_out as the search on the index would leave them; 1: the seed's disk is not inside the tile's
// rectangle; 2: an exact tie (the exact search decides).
template <int G, typename TS = unsigned short /* an entry of the cell tables: 16 bits in a tile and in an LDS index, StartT in HBM/L2 */>
__device__ inline int tile_search(const TileCls &tc, const TileLds &tl, const Lattice &L, float qx, float qy, int lig, float d0, int seed_pos,
                                  float2 seed_xy, Best &b, float2 &m, float &empty_out)
{
    const TS *tstart = reinterpret_cast<const TS *>(tl.tstart);
    const float rad = disk_radius(d0);
    const float fx = (qx - L.x0) * L.inv_h, fy = (qy - L.y0) * L.inv_h;
    const float R = (rad + L.margin) * L.inv_h;
    const int   x_lo = max(0, ifloor(fx - R)), x_hi = min(L.nx - 1, ifloor(fx + R));
    const int   y_lo = max(0, ifloor(fy - R)), y_hi = min(L.ny - 1, ifloor(fy + R));
    if (!(rad < 1.0e30f) || x_lo < tc.X0 || x_hi > tc.X1 || y_lo < tc.Y0 || y_hi > tc.Y1 || x_lo > x_hi || y_lo > y_hi) return 1;
    const int group_base = ((int)threadIdx.x & 63) & ~(G - 1);
    b.d = ulp_above(d0); // the seed, met again in its cell, is an update and no tie (icp_search.hpp, ulp_above)
    b.pos = -2;
    b.oidx = 0xffffffffu;
    bool tie = false;
    int  bdelta = 0;
    if (y_hi - y_lo < 4) {
        // The usual case -- a neighbour a cell or two away, a disk of at most four lattice rows -- as straight code: a pass is a chain
        // of latencies (a wavefront alone on its SIMD takes as long for it as two sharing one), so what counts is how many LDS round
        // trips and branches lie in a row.  Here: the rows' extents, all at once; the first candidate of every row, all at once; then
        // whatever a row has beyond one candidate per lane (only cells of stacked wall points do).
        int a[4], e[4], dl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int   y = y_lo + r;
            const float dy = fmaxf(fmaxf((float)y - fy, fy - (float)(y + 1)), 0.0f);
            const float half = __builtin_amdgcn_sqrtf(fmaxf(R * R - dy * dy, 0.0f)) + 1.0e-3f;
            const int   xa = max(x_lo, ifloor(fx - half)), xb = min(x_hi, ifloor(fx + half));
            const bool  on = (y <= y_hi) & (xa <= xb);
            const int   tr = min(y, y_hi) - tc.Y0, base = tc.soff + tr * tc.pitch;
            // (a row without cells under the disk reads its first entry twice: an empty span)
            a[r] = (int)tstart[base + (on ? (xa - tc.X0) >> tc.shift : 0)];
            e[r] = (int)tstart[base + (on ? ((xb - tc.X0) >> tc.shift) + 1 : 0)];
            dl[r] = tc.whole ? 0 : tl.rdelta[tc.row0 + tr];
        }
        float  d2nd = FLT_MAX;
        float2 c[4];
        const float2 *tp = tl.tpts + tc.tp_off;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = tp[min(a[r] + lig, tc.tp_max)];
#pragma unroll
        for (int r = 0; r < 4; ++r) scan_step(b, d2nd, a[r] + lig < e[r] ? dist2(c[r], qx, qy) : FLT_MAX, a[r] + lig);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int i = a[r] + lig + G;
            for (; i + 3 * G < e[r]; i += 4 * G) {
                const float2 m0 = tp[i], m1 = tp[i + G], m2 = tp[i + 2 * G], m3 = tp[i + 3 * G];
                const float  d_0 = dist2(m0, qx, qy), d_1 = dist2(m1, qx, qy), d_2 = dist2(m2, qx, qy), d_3 = dist2(m3, qx, qy);
                scan_step(b, d2nd, d_0, i);
                scan_step(b, d2nd, d_1, i + G);
                scan_step(b, d2nd, d_2, i + 2 * G);
                scan_step(b, d2nd, d_3, i + 3 * G);
            }
            for (; i < e[r]; i += G) scan_step(b, d2nd, dist2(tp[i], qx, qy), i);
        }
        tie = scan_tie(b, d2nd);
        if (G == 64) {
            group_min_lean<16>(b, tie);
            float bd = b.d;
            int   bp = b.pos, bt = (int)tie;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float od = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b.d), 16 * r));
                const int   op = __builtin_amdgcn_readlane(b.pos, 16 * r), ot = __builtin_amdgcn_readlane((int)tie, 16 * r);
                if (r == 0) {
                    bd = od, bp = op, bt = ot;
                } else {
                    bt |= ot | (int)((od == bd) & (op != bp) & (op >= 0) & (bp >= 0));
                    const bool take = (od < bd) | ((od == bd) & (op >= 0) & ((bp < 0) | (op < bp)));
                    bd = take ? od : bd;
                    bp = take ? op : bp;
                }
            }
            b.d = bd, b.pos = bp, tie = bt != 0;
        } else {
            group_min_lean<G>(b, tie);
        }
        if (tie) return 2;
        if (b.pos < 0) {
            b.d = d0;
            b.pos = seed_pos;
            m = seed_xy;
        } else {
            m = tp[b.pos];
            // the winner's row: the spans are disjoint runs of the tile
            int d = dl[0];
#pragma unroll
            for (int r = 1; r < 4; ++r) d = ((b.pos >= a[r]) & (b.pos < e[r])) ? dl[r] : d;
            b.pos += d;
        }
        empty_out = __fsqrt_rn(b.d) * 0.999999f;
        return 0;
    }
    const int nrows = y_hi - y_lo + 1;
    int       lpr_log = 0;
    while (lpr_log < 4 && (nrows << (lpr_log + 1)) <= G) ++lpr_log;
    const int lpr = 1 << lpr_log, slot = lig >> lpr_log, sub = lig & (lpr - 1), slots = G >> lpr_log;
    for (int y0 = y_lo; y0 <= y_hi; y0 += slots) {
        const int y = y0 + slot;
        int       a = 0, e = 0, dl = 0;
        if (y <= y_hi) {
            // the row's cells under the disk: a point of lattice row y is at least dy rows from the query, so within R of it only
            // if no farther than sqrt(R^2 - dy^2) columns (a wall that the disk merely touches leaves one or two cells per row where
            // the disk's bounding square holds metres of it)
            const float dy = fmaxf(fmaxf((float)y - fy, fy - (float)(y + 1)), 0.0f);
            const float half = __builtin_amdgcn_sqrtf(fmaxf(R * R - dy * dy, 0.0f)) + 1.0e-3f;
            const int   xa = max(x_lo, ifloor(fx - half)), xb = min(x_hi, ifloor(fx + half));
            if (xa <= xb) {
                const int r = y - tc.Y0;
                const int base = tc.soff + r * tc.pitch;
                a = (int)tstart[base + ((xa - tc.X0) >> tc.shift)];     // (column groups: a superset of the cells wanted)
                e = (int)tstart[base + ((xb - tc.X0) >> tc.shift) + 1];
                dl = tc.whole ? 0 : tl.rdelta[tc.row0 + r];
            }
        }
        const bool heavy = e - a > 8 * lpr;
        if (!heavy) tile_scan(b, tie, bdelta, tl.tpts + tc.tp_off, a, e, dl, sub, lpr, qx, qy);
        unsigned long long hm = group_bits<G>(__ballot(heavy && sub == 0), group_base);
        while (hm) {
            const int src = __builtin_ctzll(hm);
            hm &= hm - 1;
            int A, E, D;
            if (G == 64) {
                A = __builtin_amdgcn_readlane(a, src), E = __builtin_amdgcn_readlane(e, src), D = __builtin_amdgcn_readlane(dl, src);
            } else {
                A = __shfl(a, src, G), E = __shfl(e, src, G), D = __shfl(dl, src, G);
            }
            tile_scan(b, tie, bdelta, tl.tpts + tc.tp_off, A, E, D, lig, G, qx, qy);
        }
    }
    const int mine = b.pos;
    if (G == 64) {
        // within the rows of 16 on the DPP path, then the four rows' results through scalar registers (two ds_bpermute steps
        // of three values each are two more LDS round trips in a chain that is nothing but latency)
        group_min_lean<16>(b, tie);
        float bd = b.d;
        int   bp = b.pos, bt = (int)tie;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float od = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b.d), 16 * r));
            const int   op = __builtin_amdgcn_readlane(b.pos, 16 * r), ot = __builtin_amdgcn_readlane((int)tie, 16 * r);
            if (r == 0) {
                bd = od, bp = op, bt = ot;
            } else {
                bt |= ot | (int)((od == bd) & (op != bp) & (op >= 0) & (bp >= 0));
                const bool take = (od < bd) | ((od == bd) & (op >= 0) & ((bp < 0) | (op < bp)));
                bd = take ? od : bd;
                bp = take ? op : bp;
            }
        }
        b.d = bd, b.pos = bp, tie = bt != 0;
    } else {
        group_min_lean<G>(b, tie);
    }
    if (tie) return 2;
    if (b.pos < 0) {
        // nothing nearer than the seed in the cells of its disk
        b.d = d0;
        b.pos = seed_pos;
        m = seed_xy;
    } else {
        const unsigned long long who = group_bits<G>(__ballot(mine == b.pos), group_base);
        const int src = __builtin_ctzll(who);
        const int dl = G == 64 ? __builtin_amdgcn_readlane(bdelta, src) : __shfl(bdelta, src, G);
        m = tl.tpts[tc.tp_off + b.pos];
        b.pos += dl;
    }
    empty_out = __fsqrt_rn(b.d) * 0.999999f; // the class has no point nearer than its nearest
    return 0;
}

// Stages the tiles of both classes for the workgroup's slots at pose T.  Whole workgroup; ends with a barrier.
template <typename StartT, int MODE>
__device__ inline void tile_stage(const TileLds &tl, const ModelView &mv, const IndexPtrs<StartT> &ix, const Pose &T, int cnt, int nga, double gate,
                                  float move_r, float move_t, float slack_moves, float slack_cells)
{
    const Lattice &L = mv.lat;
    const int      tid = threadIdx.x;
    int           *hdr = tl.hdr;
    if (tid < 2) {
        hdr[kHdrRect + 4 * tid + 0] = 0x7fffffff, hdr[kHdrRect + 4 * tid + 1] = 0x7fffffff;
        hdr[kHdrRect + 4 * tid + 2] = -0x7fffffff, hdr[kHdrRect + 4 * tid + 3] = -0x7fffffff;
        hdr[kHdrValid + tid] = 0;
    }
    __syncthreads();
    // the rectangle of every class: cells of the disk of each query whose seed is an inlier, dilated (a seed beyond the gate is
    // searched on the index, where the radius proved empty usually spares it the search altogether)
    if (tid < cnt && tl.spos[tid] >= 0) {
        const int     cls = MODE == SLAM_ICP_P2L ? 1 : (tl.sidx[tid] < nga ? 0 : 1);
        const double2 P = tl.P[tid];
        float         qx, qy;
        transform_query(T, P, qx, qy);
        const float d0 = dist2(tl.sxy[tid], qx, qy);
        const float rad = disk_radius(d0);
        const float move = move_r * (fabsf((float)P.x) + fabsf((float)P.y) + 1.0e-3f) + move_t;
        const float pad = (rad + L.margin + slack_moves * move) * L.inv_h + slack_cells;
        const float fx = (qx - L.x0) * L.inv_h, fy = (qy - L.y0) * L.inv_h;
        if ((double)d0 < gate && fx - pad < 1.0e9f && fx + pad > -1.0e9f && fy - pad < 1.0e9f && fy + pad > -1.0e9f) { // (NaN: no part in the rectangle)
            atomicMin(&hdr[kHdrRect + 4 * cls + 0], ifloor(fx - pad));
            atomicMin(&hdr[kHdrRect + 4 * cls + 1], ifloor(fy - pad));
            atomicMax(&hdr[kHdrRect + 4 * cls + 2], ifloor(fx + pad));
            atomicMax(&hdr[kHdrRect + 4 * cls + 3], ifloor(fy + pad));
        }
    }
    __syncthreads();
    int X0[2], Y0[2], X1[2], Y1[2], H[2], ok[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        X0[c] = max(0, hdr[kHdrRect + 4 * c + 0]), Y0[c] = max(0, hdr[kHdrRect + 4 * c + 1]);
        X1[c] = min(L.nx - 1, hdr[kHdrRect + 4 * c + 2]), Y1[c] = min(L.ny - 1, hdr[kHdrRect + 4 * c + 3]);
        ok[c] = X0[c] <= X1[c] && Y0[c] <= Y1[c] && mv.n_cls[c] > 0;
        H[c] = ok[c] ? Y1[c] - Y0[c] + 1 : 0;
    }
    // what fits: rows and cell-table entries of class 0, then of class 1 behind it.  A table too wide for its share takes one entry
    // per 2, 4, ... columns (a search then reads whole column groups: more candidates, the same result)
    int soff[2] = {0, 0}, shift[2] = {0, 0}, pitch[2] = {2, 2};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (ok[c] && H[0] * (c ? 1 : 0) + H[c] > kTileRowsMax) ok[c] = 0, H[c] = 0;
        if (ok[c]) {
            const int room = (c ? kTileStartMax - soff[1] : (ok[1] ? kTileStartMax * 3 / 4 : kTileStartMax)) / H[c]; // entries per row
            while (shift[c] < 8 && ((X1[c] - X0[c]) >> shift[c]) + 2 > room) ++shift[c];
            pitch[c] = ((X1[c] - X0[c]) >> shift[c]) + 2;
            if (pitch[c] > room) ok[c] = 0, H[c] = 0;
        }
        if (c == 0) soff[1] = ok[0] ? H[0] * pitch[0] : 0;
    }
    // the rows' spans of the sorted array
    const int Ht = H[0] + H[1];
    int       my_cnt = 0, my_ga = 0;
    if (tid < Ht) {
        const int     c = tid < H[0] ? 0 : 1, r = tid - (c ? H[0] : 0);
        const StartT *st = ix.start[c] + (size_t)(Y0[c] + r) * L.nx;
        my_ga = (int)st[X0[c]];
        my_cnt = (int)st[X1[c] + 1] - my_ga;
    }
    // class totals (block_scan_excl's two barriers order the header reads above before the writes below)
    int       total = 0;
    const int excl = block_scan_excl(my_cnt, hdr + kHdrScan, &total);
    if (tid < Ht) {
        tl.rloff[tid] = excl;
        tl.rga[tid] = my_ga;
    }
    if (tid == 0) tl.rloff[Ht] = total;
    __syncthreads();
    const int tot0 = H[0] > 0 ? tl.rloff[H[0]] : 0; // points of class 0's rows (class 1's follow)
    int       shift1 = 0;                           // class 1's local positions move down when class 0 is dropped
    if (ok[0] && tot0 > kTilePtsMax) ok[0] = 0, shift1 = tot0;
    if (ok[1] && (ok[0] ? total : total - tot0) > kTilePtsMax) ok[1] = 0;
    __syncthreads();
    if (tid < Ht) {
        const int c = tid < H[0] ? 0 : 1;
        const int lo = tl.rloff[tid] - (c ? shift1 : 0);
        tl.rdelta[tid] = tl.rga[tid] - lo;
    }
    if (tid == 0) {
        hdr[kHdrMiss] = 0; // (every wavefront has read it: it did so before the first barrier above)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            hdr[kHdrRect + 4 * c + 0] = X0[c], hdr[kHdrRect + 4 * c + 1] = Y0[c], hdr[kHdrRect + 4 * c + 2] = X1[c], hdr[kHdrRect + 4 * c + 3] = Y1[c];
            hdr[kHdrValid + c] = ok[c];
            hdr[kHdrRow0 + c] = c ? H[0] : 0;
            hdr[kHdrSoff + c] = soff[c];
        }
        hdr[kHdrShift] = shift[0] | (shift[1] << 8);
    }
    __syncthreads();
    // cell tables and points of the classes that fit
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (!ok[c]) continue;
        const int     row0 = c ? H[0] : 0, pc = pitch[c], sc = shift[c];
        const StartT *st = ix.start[c];
        const int ncell = H[c] * pc;
        for (int k0 = tid; k0 < ncell; k0 += 4 * kSB) {
            int v[4], r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = min(k0 + u * kSB, ncell - 1);
                r[u] = k / pc;
                v[u] = (int)st[(size_t)(Y0[c] + r[u]) * L.nx + min(X0[c] + ((k - r[u] * pc) << sc), X1[c] + 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u * kSB < ncell) tl.tstart[soff[c] + k0 + u * kSB] = (unsigned short)(v[u] - tl.rdelta[row0 + r[u]]);
        }
        const int     sh = c ? shift1 : 0;
        const int     first = tl.rloff[row0] - sh, last = tl.rloff[row0 + H[c]] - sh;
        const float2 *gp = ix.pts + mv.base[c];
        const int    *rl = tl.rloff + row0;
        // Eight CONSECUTIVE points per thread and trip: one bisection (nine dependent LDS reads) for the row of the first, a walk to the
        // rows of the others (one compare each; a step only where a row ends inside the eight), eight loads in flight.  Lanes 64 bytes
        // apart cost the L1 more lines per load than lanes side by side would -- and a bisection per point (the first form) cost five
        // times more: a tile of 10 000 points took 10 us to copy, most of config 3's first staging.
        constexpr int kU = 8;
        for (int k0 = first + kU * tid; k0 < last; k0 += kU * kSB) {
            int row = 0;
#pragma unroll
            for (int step = kTileRowsMax / 2; step > 0; step >>= 1) {
                const int cand = row + step;
                if (cand < H[c] && rl[cand] - sh <= k0) row = cand;
            }
            int    next = rl[row + 1] - sh, delta = tl.rdelta[row0 + row];
            float2 v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int k = min(k0 + u, last - 1); // (past the end: the last point again, not stored)
                while (k >= next) {                  // (rows without points start where the next one does: skipped one by one)
                    ++row;
                    next = rl[row + 1] - sh;
                    delta = tl.rdelta[row0 + row];
                }
                v[u] = gp[k + delta];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u)
                if (k0 + u < last) tl.tpts[k0 + u] = v[u];
        }
    }
    __syncthreads();
}

// WHOLE (a model whose index fits LDS; ix points into the copy there): the tile is the index itself -- nothing is ranked (any deal of
// the points will do) and nothing staged; what the form brings is the rest: the points and their search state in LDS slots behind the
// index instead of global memory, the seeded search as straight code, certificates that last (config 1's own usage, one
// 1081-beam scan against the 10 k map: 7.5 -> ... us per iteration).  slot_room: bytes of LDS behind the index.
// WHOLE = 2: the index where it lies in HBM/L2, for models of a few points per cell (staging tiles costs what they save there): the
// same slots, certificates and straight-line seeded search, its three dependent round trips through L2 instead of LDS.
template <int G, typename StartT, int MODE, int WHOLE = 0>
__device__ inline bool tile_iterations(const ModelView &mv, const FitArgs &fa, const IndexPtrs<StartT> &ix, unsigned char *smem,
                                       unsigned long long *gran, int *abort_word, unsigned long long first_ticks, int parts, int n_act, int s,
                                       int off, int n, int nga, FitState &fs, float slack_moves, float slack_cells, int tile_dbg, long long *sstamps)
{
    TileLds        tl = WHOLE == 1 ? tile_lds_whole(smem, mv, (n + n_act - 1) / n_act) : tile_lds(smem);
    if (WHOLE == 2) {
        tl.tstart = reinterpret_cast<unsigned short *>(const_cast<unsigned char *>(mv.blob));
        tl.tpts = reinterpret_cast<float2 *>(const_cast<unsigned char *>(mv.blob + mv.off_pts));
    }
    double        *partial = tl.partial, *bc = tl.bc;
    const Lattice &L = mv.lat;
    const int      tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int      part = blockIdx.x;
    constexpr int  kPerPass = kSB / G;
    const int      Q = (n + n_act - 1) / n_act;              // scene points per workgroup
    const int      lo = part * Q, cnt = max(0, min(n, lo + Q) - lo);
    double r00 = fs.r00, r01 = fs.r01, r10 = fs.r10, r11 = fs.r11, t0 = fs.t0, t1 = fs.t1, delta = fs.delta;
    const double gate = MODE == SLAM_ICP_P2L ? (double)INFINITY : fa.indist; // icpPointToPlane.cpp:55-77: no gate
    const float  gate_r = MODE == SLAM_ICP_P2L ? 1.0e30f : ulp_above(ulp_above((float)sqrt(fmax(fa.indist, 0.0)))); // every inlier is nearer
    if (WHOLE) { // the points as they come
        if (tid < cnt) {
            tl.sidx[tid] = lo + tid;
            tl.P[tid] = fa.pts[off + lo + tid];
            tl.spos[tid] = -1;
            tl.sempty[tid] = 0.0f;
            tl.scq[tid] = make_float2(0.0f, 0.0f);
            tl.snpos[tid] = -1;
        }
        if (tid == 0) tl.hdr[kHdrMiss] = 0, tl.hdr[kHdrDebug] = 0, tl.hdr[26] = 0, tl.hdr[27] = 0;
        __syncthreads();
    } else
    // ---- the scan's points along the Morton curve of the lattice under the initial pose, class first
    {
        unsigned *keys = reinterpret_cast<unsigned *>(tl.tpts);
        int       bits = 0;
        while (((max(L.nx, L.ny) - 1) >> bits) > 1023) ++bits;
        const Pose T = {r00, r01, r10, r11, t0, t1};
        const int  n4 = (n + 3) & ~3;
        for (int i = tid; i < n4; i += kSB) {
            unsigned key = 0xffffffffu;
            if (i < n) {
                float qx, qy;
                transform_query(T, fa.pts[off + i], qx, qy);
                const int cx = clampi(ifloor((qx - L.x0) * L.inv_h), 0, L.nx - 1) >> bits;
                const int cy = clampi(ifloor((qy - L.y0) * L.inv_h), 0, L.ny - 1) >> bits;
                const unsigned cls = MODE == SLAM_ICP_P2L ? 1u : (i < nga ? 0u : 1u);
                key = (((cls << 20) | morton10((unsigned)cx, (unsigned)cy)) << 11) | (unsigned)i;
            }
            keys[i] = key;
        }
        __syncthreads();
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
        for (int i = tid; i < n; i += kSB) {
            const unsigned me = keys[i];
            int            rank = 0;
            for (int j = 0; j < n4 / 4; ++j) {
                const uint4 o = k4[j];
                rank += (int)(o.x < me) + (int)(o.y < me) + (int)(o.z < me) + (int)(o.w < me);
            }
            if (rank >= lo && rank < lo + cnt) tl.sidx[rank - lo] = i;
        }
        __syncthreads();
        if (tid < cnt) {
            tl.P[tid] = fa.pts[off + tl.sidx[tid]];
            tl.spos[tid] = -1;
            tl.sempty[tid] = 0.0f;
            tl.scq[tid] = make_float2(0.0f, 0.0f);
            tl.snpos[tid] = -1;
        }
        if (tid < 2) tl.hdr[kHdrValid + tid] = 0;
        if (tid == 0) tl.hdr[kHdrMiss] = 0, tl.hdr[kHdrDebug] = 0, tl.hdr[26] = 0, tl.hdr[27] = 0;
        __syncthreads();
    }
    int      iters = 0, n_corr = 0, restages = 0, missed = 0;
    bool     staged = false;
    float    move_r = 0.0f, move_t = 0.0f;
    TileCls  tc0, tc1;
    if (WHOLE) {
        tc0.X0 = tc0.Y0 = 0, tc0.X1 = L.nx - 1, tc0.Y1 = L.ny - 1, tc0.row0 = 0, tc0.shift = 0, tc0.pitch = L.nx, tc0.whole = 1;
        tc1 = tc0;
        tc0.soff = (int)(mv.off_start[0] / sizeof(StartT)), tc1.soff = (int)(mv.off_start[1] / sizeof(StartT));
        tc0.tp_off = mv.base[0], tc1.tp_off = mv.base[1];
        tc0.tp_max = max(mv.n_cls[0] - 1, 0), tc1.tp_max = max(mv.n_cls[1] - 1, 0);
        tc0.valid = mv.n_cls[0] > 0, tc1.valid = mv.n_cls[1] > 0;
        staged = true; // (nothing to stage, ever)
    } else {
        tc0 = tile_cls(tl.hdr, 0), tc1 = tile_cls(tl.hdr, 1);
    }
    for (int iter = 0; iter < fa.max_iter; ++iter) {
        double acc[kNumAcc];
#pragma unroll
        for (int k = 0; k < kNumAcc; ++k) acc[k] = 0.0;
        const Pose T = {r00, r01, r10, r11, t0, t1};
        SPREAD_STAMP(0);
        if (fa.step_pose && part == 0 && tid == 0) {
            double *sp = fa.step_pose + 6 * (size_t)s;
            sp[0] = r00, sp[1] = r01, sp[2] = r10, sp[3] = r11, sp[4] = t0, sp[5] = t1;
        }
        // (re)stage: after the first step, and when a query has left its rectangle
        if (!WHOLE && iter >= 1 && (!staged || (missed != 0 && restages < kTileRestage))) {
            if (staged) ++restages;
            staged = true;
#ifdef SLAM_MEASURE
            if (tile_dbg & 12) { // what a staging costs warm: first the same tile (2) or one three metres off (4), then the real one
                Pose Tx = T;
                if (tile_dbg & 8) Tx.t0 += 3.0, Tx.t1 += 3.0;
                tile_stage<StartT, MODE>(tl, mv, ix, Tx, cnt, nga, tile_dbg & 8 ? (double)INFINITY : gate, move_r, move_t, slack_moves, slack_cells);
                SPREAD_STAMP(1);
            }
#endif
#ifdef SLAM_MEASURE
            const unsigned long long stage_t0 = __builtin_amdgcn_s_memrealtime();
#endif
            tile_stage<StartT, MODE>(tl, mv, ix, T, cnt, nga, gate, move_r, move_t, slack_moves, slack_cells);
            tc0 = tile_cls(tl.hdr, 0), tc1 = tile_cls(tl.hdr, 1);
            SPREAD_STAMP(6);
#ifdef SLAM_MEASURE
            if (sstamps && s == 0 && tid == 0) {
                long long *st = sstamps + ((size_t)blockIdx.x * fa.max_iter + iter) * kSpreadStampSlots;
                const int  rows = (tc0.valid ? tc0.Y1 - tc0.Y0 + 1 : 0) + (tc1.valid ? tc1.Y1 - tc1.Y0 + 1 : 0);
                st[8] = rows > 0 ? tl.rloff[(tc1.valid ? tc1.row0 + tc1.Y1 - tc1.Y0 + 1 : tc0.Y1 - tc0.Y0 + 1)] : 0;
                st[9] = (tc0.valid ? (tc0.Y1 - tc0.Y0 + 1) * tc0.pitch : 0) + (tc1.valid ? (tc1.Y1 - tc1.Y0 + 1) * tc1.pitch : 0);
                st[10] = rows | (tc0.shift << 16) | (tc1.shift << 24);
                st[11] = (long long)(__builtin_amdgcn_s_memrealtime() - stage_t0);
            }
#endif
        }
        for (int k0 = 0; k0 < cnt; k0 += kPerPass) {
            const int k = k0 + tid / G, lig = tid % G;
            if (k < cnt) {
                const int cls = MODE == SLAM_ICP_P2L ? 1 : (tl.sidx[k] < nga ? 0 : 1); // a point-to-line model is one class
                if (MODE == SLAM_ICP_P2L || mv.n_cls[cls] > 3) { // icpPointToPoint.cpp:59,93
                    float         qx, qy;
                    const double2 P = tl.P[k];
                    transform_query(T, P, qx, qy);
                    const int    spos = iter > 0 ? tl.spos[k] : -1;
                    const float  sempty = iter > 0 ? tl.sempty[k] : 0.0f;
                    const float2 sxy = tl.sxy[k];
                    Best         b;
                    b.d = FLT_MAX, b.pos = -1, b.oidx = 0xffffffffu;
                    float2 m = make_float2(0.0f, 0.0f);
                    float  empty = 0.0f;
                    bool   done = false, far = false;
                    const TileCls &tc = cls ? tc1 : tc0;
#ifdef SLAM_MEASURE
                    const unsigned long long q_t0 = sstamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
                    unsigned                 q_how = 0u; // 1: gate square, 2: rows on the index, 4: tile
#endif
                    // every model point lies inside the lattice's rectangle: a query farther from it than the inlier gate has no
                    // correspondence whatever its nearest point is (icpPointToPoint.cpp:76) -- no search, no state (a scene cloud
                    // seen from the next pose has parts the target never saw: a third of config 3's queries)
                    if (MODE != SLAM_ICP_P2L) {
                        const float ox = fmaxf(fmaxf(L.x0 - qx, qx - (L.x0 + (float)L.nx * L.h)), 0.0f);
                        const float oy = fmaxf(fmaxf(L.y0 - qy, qy - (L.y0 + (float)L.ny * L.h)), 0.0f);
                        const float od = __builtin_amdgcn_sqrtf(ox * ox + oy * oy) * 0.9999f - 4.0f * L.margin;
                        if (od > 0.0f && (double)od * (double)od >= gate) done = true, far = true;
                    }
                    if (!done && spos >= 0) {
                        const float d0 = dist2(sxy, qx, qy);
                        if ((double)d0 < gate) { // the seed is an inlier: the nearest point lies in the seed's disk
                            const int st = !tc.valid ? 1
                                           : (WHOLE == 2 ? tile_search<G, StartT>(tc, tl, L, qx, qy, lig, d0, spos, sxy, b, m, empty)
                                                         : tile_search<G>(tc, tl, L, qx, qy, lig, d0, spos, sxy, b, m, empty));

                            done = st == 0;
                            if (st == 1 && lig == 0) tl.hdr[kHdrMiss] = 1; // the workgroup stages again before the next iteration
                        }
                    }
                    bool searched = done && !far;
                    if (!done) {
                        // The class has no point within sempty of scq, where this point's last search ran, and the query is `disp` from
                        // there now: if what is left of the radius is still beyond the inlier gate, the nearest point is no
                        // correspondence whatever it is (icpPointToPoint.cpp:76) -- no read at all, and the state stays as it is (the
                        // radius is spent by the NET displacement, not by the steps' lengths added up: it lasts the whole fit where the
                        // search's own bookkeeping, nn_search_rows_impl, spends it in a few iterations).  Also for a query outside the
                        // lattice: the radius is one around the query itself.
                        const float2 cq = tl.scq[k];
                        const float  ddx = qx - cq.x, ddy = qy - cq.y;
                        const float  disp = __builtin_amdgcn_sqrtf(ddx * ddx + ddy * ddy) * 1.0001f + L.margin;
                        const float  D = sempty - disp - 2.0f * L.margin;
                        if (!(sempty > 0.0f && D > 0.0f && (double)D * (double)D >= gate)) {
                            searched = true;
                            bool have = false;
                            if (MODE != SLAM_ICP_P2L && spos < 0) have = nn_search_gate<G, StartT>(ix, mv, cls, qx, qy, lig, gate, gate_r, b, empty);
                            if (!have) {
                                const Seed seed = {spos, sempty};
                                b = nn_search_rows<G, StartT>(ix, mv, cls, qx, qy, lig, gate, seed, disp, empty);
                            }
                            if (b.pos >= 0) m = ix.pts[mv.base[cls] + b.pos];
#ifdef SLAM_MEASURE
                            if (lig == 0) atomicAdd(&tl.hdr[kHdrDebug], have ? 0x10000 : 1);
                            q_how |= have ? 1u : 2u;
#endif
                        }
                    }
#ifdef SLAM_MEASURE
                    if (sstamps && s == 0 && lig == 0) { // the slowest query of the workgroup this iteration: ticks, slot, how it was searched
                        if (done && !far) q_how |= 4u;
                        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - q_t0;
                        atomicMax(reinterpret_cast<unsigned long long *>(&tl.hdr[26]), (dt << 32) | ((unsigned long long)k << 8) | (unsigned)(cls << 3) | q_how);
                    }
#endif
                    if (lig == 0) {
                        if (MODE == SLAM_ICP_P2L) {
                            if (b.pos >= 0) {
                                double2 nrm;
                                if (tl.snpos[k] == b.pos) {
                                    nrm = tl.snrm[k];
                                } else {
                                    nrm = reinterpret_cast<const double2 *>(mv.normals)[(unsigned)ix.oidx[mv.base[1] + b.pos]];
                                    tl.snrm[k] = nrm;
                                    tl.snpos[k] = b.pos;
                                }
                                add_p2l(m, nrm, qx, qy, acc);
                            }
                        } else if (b.pos >= 0 && (double)b.d < fa.indist) {
                            add_p2p_xy(mv, m, qx, qy, acc); // :76
                        }
                        if (searched) { // (a point spared its search keeps what its last search left)
                            tl.spos[k] = b.pos;
                            tl.sxy[k] = m;
                            tl.sempty[k] = empty;
                            tl.scq[k] = make_float2(qx, qy);
                        }
                    }
                }
            }
        }
        SPREAD_STAMP(1);
        if ((lane & (G - 1)) == 0) { // only a query's first lane holds sums: one row of partials each, no reduction across the wavefront
            double *my = partial + (wave * (64 / G) + lane / G) * kNumAcc;
#pragma unroll
            for (int k = 0; k < kNumAcc; ++k) my[k] = acc[k];
        }
        __syncthreads();
        SPREAD_STAMP(2);
#ifdef SLAM_MEASURE
        if (sstamps && s == 0 && tid == 0) {
            long long *st = sstamps + ((size_t)blockIdx.x * fa.max_iter + iter) * kSpreadStampSlots;
            st[7] = tl.hdr[kHdrDebug];
            tl.hdr[kHdrDebug] = 0;
            const unsigned long long key = *reinterpret_cast<unsigned long long *>(&tl.hdr[26]);
            *reinterpret_cast<unsigned long long *>(&tl.hdr[26]) = 0ull;
            const int kq = (int)((key >> 8) & 0xffffu);
            st[12] = (long long)key;
            st[13] = (long long)(((unsigned long long)__float_as_uint(tl.scq[kq].y) << 32) | __float_as_uint(tl.scq[kq].x));
            st[14] = tl.spos[kq];

        }
#endif
        if (wave == 0) {
            const double pose[6] = {r00, r01, r10, r11, t0, t1};
            spread_exchange<MODE>(mv, fa, partial, kSW * (64 / G), bc, gran, abort_word, first_ticks, iter, part, parts, n_act, n, s, pose, sstamps);
        }
        __syncthreads();
        SPREAD_STAMP(5);
        if (bc[7] < 0.0) return false; // uniform: an exchange gave up
        missed = uniform_i(tl.hdr[kHdrMiss]);
        {
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
    }
    fs.r00 = r00, fs.r01 = r01, fs.r10 = r10, fs.r11 = r11, fs.t0 = t0, fs.t1 = t1;
    fs.delta = delta;
    fs.iters = iters;
    fs.n_corr = n_corr;
    return true;
}

// workgroup 0 of a scan writes what the fit found
__device__ inline void spread_finish(const FitArgs &fa, const FitState &fs, int *flags, int s, bool ok)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // workgroup 0 decides: if IT saw every exchange through, the pose is complete whatever the others did afterwards
        flags[gridDim.y + s] = ok ? 0 : 1; // redo: the one-workgroup form takes this scan, from the pose left untouched here (written
                                           // either way: nothing clears the flags between launches)
        if (fa.redo_mirror) fa.redo_mirror[s] = ok ? 0 : 1;
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

// grid (parts, n_scans); a workgroup whose scan does not need it exits at once
template <typename StartT, bool LDS, int MODE>
__global__ __launch_bounds__(kSB) void icp_fit_spread_kernel(ModelView mv, FitArgs fa, unsigned long long *gran, int *flags /* [2][n_scans] abort | redo */,
                                                                unsigned long long first_ticks, float2 *qstate, int qcap, int wide_max, int tile_on, int tile_lanes, int slot_room,
                                                                float slack_moves, float slack_cells, long long *sstamps)
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
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            flags[gridDim.y + s] = 0; // nothing to redo
            if (fa.result) {
                fa.result[s].iters = 0;
                fa.result[s].n_corr = 0;
                fa.result[s].delta = 0.0;
            }
        }
        return;
    }
    const bool wide = n <= wide_max; // 64 lanes per query up to here, 16 beyond
    // The scan's form and its workgroups (the decision is the scan's: the same in all of them).  The tile form deals the points so
    // that tile_lanes lanes of a workgroup have one in a pass.  All 512 (round 6, tools/exp/spread_lanes.sh): with 256 -- four of the
    // eight wavefronts, one per SIMD, twice the workgroups -- a wavefront's pass is no shorter (3.9 against 4.1 us: it is a chain
    // of latencies, not of issue slots) and the exchange between twice the workgroups costs 1.5 us more per iteration (config 3:
    // 237 -> 266 us per fit; 128 lanes: 326)
    int  n_act = active_parts(n, wide ? 64 : 16, parts);
    bool tiled = false;
    if (!LDS && (tile_on & 3) == 1) { // tiles staged into LDS
        const int want = (int)(((long long)n * (wide ? 64 : 16) + tile_lanes - 1) / tile_lanes);
        const int na = want < 1 ? 1 : (want > parts ? parts : want);
        if (n <= kTileMaxN && (n + na - 1) / na <= kTileSlots) tiled = true, n_act = na;
    }
    if (!LDS && (tile_on & 3) == 2 && (n + n_act - 1) / n_act <= kTileSlots) tiled = true; // the index where it lies, slots in LDS
    if (LDS && tile_on) { // the index in LDS is its own tile: the form's slots go behind it, if they fit
        const int q = (n + n_act - 1) / n_act;
        if (q <= kSB && (int)kSlotBytes * q <= slot_room) tiled = true;
    }
    if (part >= n_act) return;
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
    if (tiled) {
        bool okt;
        if (LDS)
            okt = wide ? tile_iterations<64, StartT, MODE, LDS ? 1 : 0>(mv, fa, ix, smem, g, flags + s, first_ticks, parts, n_act, s, off, n, nga, fs, slack_moves, slack_cells, tile_on, sstamps)
                       : tile_iterations<16, StartT, MODE, LDS ? 1 : 0>(mv, fa, ix, smem, g, flags + s, first_ticks, parts, n_act, s, off, n, nga, fs, slack_moves, slack_cells, tile_on, sstamps);
        else if ((tile_on & 3) == 2)
            okt = wide ? tile_iterations<64, StartT, MODE, LDS ? 1 : 2>(mv, fa, ix, smem, g, flags + s, first_ticks, parts, n_act, s, off, n, nga, fs, slack_moves, slack_cells, tile_on, sstamps)
                       : tile_iterations<16, StartT, MODE, LDS ? 1 : 2>(mv, fa, ix, smem, g, flags + s, first_ticks, parts, n_act, s, off, n, nga, fs, slack_moves, slack_cells, tile_on, sstamps);
        else
            okt = wide ? tile_iterations<64, StartT, MODE, LDS ? 1 : 0>(mv, fa, ix, smem, g, flags + s, first_ticks, parts, n_act, s, off, n, nga, fs, slack_moves, slack_cells, tile_on, sstamps)
                       : tile_iterations<16, StartT, MODE, LDS ? 1 : 0>(mv, fa, ix, smem, g, flags + s, first_ticks, parts, n_act, s, off, n, nga, fs, slack_moves, slack_cells, tile_on, sstamps);
        spread_finish(fa, fs, flags, s, okt);
        return;
    }
    const bool          ok = wide ? spread_iterations<64, StartT, MODE>(mv, fa, ix, smem, g, flags + s, first_ticks, qstate, qcap, parts, s, off, n, nga, fs, sstamps)
                                  : spread_iterations<16, StartT, MODE>(mv, fa, ix, smem, g, flags + s, first_ticks, qstate, qcap, parts, s, off, n, nga, fs, sstamps);
    spread_finish(fa, fs, flags, s, ok);
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
    // Scratch of the launch: [abort | redo] words per scan at fixed places in front, the granules behind.  Nothing is filled between
    // launches (a fill was a launch of its own in front of every fit): a granule counts when it carries THIS launch's tag, an abort
    // word when it holds this launch's tag base, and the redo words are written by every scan's first workgroup either way.  Tag
    // bases only grow (by max_iter + 2 per launch); the buffer is cleared when it is made, grows, or the 32 bits run out.
    const size_t flag_bytes = sizeof(int) * 2 * (size_t)std::max(n_cu, n_scans);
    const size_t gran_bytes = sizeof(unsigned long long) * 2 * (size_t)parts * kGranPerWg * (size_t)n_scans;
    const size_t had = h->w_single.cap;
    SLAM_TRY(h->w_single.reserve(flag_bytes + gran_bytes));
    const unsigned span = (unsigned)std::max(fa.max_iter, 0) + 2u;
    // (a stream that is being captured into a hipGraph: see below for the order; for the tags it means that the launch will run
    // again and again with the SAME base -- last replay's granules would count as this one's.  A captured launch takes the fill in
    // front of it into the graph and tags from a range of its own, the upper half, that no uncaptured launch uses.)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    const bool     ordered = cap == hipStreamCaptureStatusNone;
    constexpr unsigned kCapturedTag = 0x80000000u;
    if (!ordered || h->w_single.cap != had || h->spread_tag + span >= kCapturedTag || h->spread_tag == 0) {
        SLAM_HIP(hipMemsetAsync(h->w_single.p, 0, h->w_single.cap, st));
        if (ordered) h->spread_tag = 1;
    }
    FitArgs fa_tagged = fa;
    fa_tagged.spread_tag = ordered ? h->spread_tag : kCapturedTag;
    if (ordered) h->spread_tag += span;
    int                *flags = static_cast<int *>(h->w_single.p);
    unsigned long long *gran = reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(h->w_single.p) + flag_bytes);
    *redo_flags = flags + n_scans;
    std::lock_guard<std::mutex> lk(g_spread_mu);
    hipEvent_t &done = g_spread_done[dev & 15];
    // (a stream that is being captured into a hipGraph may neither wait for an event of uncaptured work nor lend its own to
    // other streams: a captured spread launch is ordered by its graph alone -- if it ever meets another one, the redo path covers it)
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
    long long *sstamps = nullptr;
#ifdef SLAM_MEASURE
    if (const char *e = getenv("SLAM_SPREAD_WIDE_MAX")) wide_max = atoi(e);
    if (getenv("SLAM_SPREAD_STAMPS")) {
        const size_t sb = sizeof(long long) * (size_t)parts * (size_t)std::max(fa.max_iter, 1) * kSpreadStampSlots;
        SLAM_TRY(h->w_stamps.reserve(sb));
        SLAM_HIP(hipMemsetAsync(h->w_stamps.p, 0, sb, st));
        sstamps = static_cast<long long *>(h->w_stamps.p);
        h->n_stamps = -parts; // (negative: spread stamps, [parts][max_iter][kSpreadStampSlots])
        h->spread_stamp_iters = std::max(fa.max_iter, 1);
    }
#endif
    const bool p2l = h->prm.mode == SLAM_ICP_P2L; // the nine sums and the solve differ, nothing else
    // the tile form (an index in HBM/L2 only): how far a tile reaches beyond its queries' disks -- slack_moves times a query's last
    // move plus slack_cells lattice cells
    // Which form (slam_icp_params::spread_tile; 0 = by the model).  Tiles staged into LDS pay where a search through L2 is long: cells
    // of hundreds of points (a lidar cloud's stacked wall points; the same test that gives a query 64 lanes).  On a model of a few
    // points per cell a search through L2 is a few dependent loads and staging costs what the tiles save (2 x 19 999 room points, one
    // 1081-beam scan: 187 us per fit in the plain form, 209 with tiles): there the index stays where it is and the form brings its
    // slots, certificates and straight-line seeded search (form 2).  An index in LDS is its own tile.
    const int form = h->prm.spread_tile > 0 ? std::min(h->prm.spread_tile, 2) : (h->max_cell_points > kDenseCell ? 1 : 2);
    int       tile_on = h->prm.spread_tile < 0 ? 0 : (h->in_lds ? 1 : form);
    float slack_moves = 3.0f, slack_cells = 1.0f;
    int   tile_lanes = kSB; // lanes of a workgroup with a scene point in a pass (icp_fit_spread_kernel)
#ifdef SLAM_MEASURE
    if (const char *e = getenv("SLAM_SPREAD_TILE")) tile_on = atoi(e); // 0 off, 1 tiles, 2 the index where it lies; +4 / +8: the first staging done twice
    if (const char *e = getenv("SLAM_TILE_SLACK_MOVES")) slack_moves = (float)atof(e);
    if (const char *e = getenv("SLAM_TILE_SLACK_CELLS")) slack_cells = (float)atof(e);
    if (const char *e = getenv("SLAM_TILE_LANES")) tile_lanes = std::min(std::max(atoi(e), 64), (int)kSB);
#endif
    const size_t global_lds = (tile_on & 3) == 1 ? (size_t)kTileLdsBytes
                              : ((tile_on & 3) == 2 ? (size_t)(kTileHeadBytes + 128u + kTileSlots * kSlotBytes) : (size_t)kScratchBytes);
    // an index in LDS: the tile form's slots behind it, as many as the CU's 160 KB leave (64 bytes per scene point of a workgroup)
    const size_t lds_top = ((h->lds_bytes + 15) & ~(size_t)15);
    const int    slot_room = h->in_lds && tile_on && lds_top < kLdsTotal ? (int)std::min<size_t>(kLdsTotal - lds_top, (size_t)kSlotBytes * kSB) : 0;
    if (h->in_lds) {
        auto kern = p2l ? icp_fit_spread_kernel<uint16_t, true, SLAM_ICP_P2L> : icp_fit_spread_kernel<uint16_t, true, SLAM_ICP_P2P>;
        SLAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        SLAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_top + slot_room)));
        hipLaunchKernelGGL(kern, grid, dim3(kSB), lds_top + slot_room, st, h->mv, fa_tagged, gran, flags, first_ticks, qstate, qcap, wide_max, tile_on, kSB, slot_room,
                           0.0f, 0.0f, sstamps);
    } else if (h->start32) {
        auto kern = p2l ? icp_fit_spread_kernel<uint32_t, false, SLAM_ICP_P2L> : icp_fit_spread_kernel<uint32_t, false, SLAM_ICP_P2P>;
        SLAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)global_lds));
        hipLaunchKernelGGL(kern, grid, dim3(kSB), global_lds, st, h->mv, fa_tagged, gran, flags, first_ticks, qstate, qcap, wide_max, tile_on, tile_lanes, 0, slack_moves,
                           slack_cells, sstamps);
    } else {
        auto kern = p2l ? icp_fit_spread_kernel<uint16_t, false, SLAM_ICP_P2L> : icp_fit_spread_kernel<uint16_t, false, SLAM_ICP_P2P>;
        SLAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)global_lds));
        hipLaunchKernelGGL(kern, grid, dim3(kSB), global_lds, st, h->mv, fa_tagged, gran, flags, first_ticks, qstate, qcap, wide_max, tile_on, tile_lanes, 0, slack_moves,
                           slack_cells, sstamps);
    }
    SLAM_HIP(hipGetLastError());
    if (ordered) SLAM_HIP(hipEventRecord(done, st));
    return SLAM_OK;
}

} // namespace icp
} // namespace slam
