// icp.hip -- class-constrained 2-D ICP on gfx950 behind the C-ABI.
//
// Reference path (all under /root/reference/ccicp2d):
//   Icp::Icp            src/icp.cpp:26-70        model f64 -> f32, two kd-trees
//   Icp::fit/fitIterate src/icp.cpp:80-122       <= max_iter x fitStep, stop on delta < min_delta
//   IcpPointToPoint::fitStep src/icpPointToPoint.cpp:33-172
//   KDTree::n_nearest   src/kdtree.cpp:378-391   exact 1-NN in float
//
// MI355X design (DESIGN.md "ICP kernel"):
//   * one workgroup (16 wavefronts) per scan, resident for ALL iterations: the
//     pose never leaves registers, two s_barriers per iteration, no host round
//     trip (the reference's loop-carried dependency is per scan, scans are
//     independent: SURVEY 8(a) I6);
//   * the model is held in LDS as a uniform-cell index (points sorted by cell,
//     row-major, per class) instead of two kd-trees: a (2r+1)^2 neighbourhood is
//     (2r+1) CONTIGUOUS spans, so the lanes that share a scene point read
//     consecutive float2 (conflict-free ds_read_b64) with no pointer chasing;
//   * G lanes of a wavefront cooperate on one scene point (G = 64 is the
//     north-star "one wavefront per scan point"; the default is measured);
//   * the per-iteration normal-equation sums (9 doubles) are reduced on the DPP
//     cross-lane path inside the wavefront and once through LDS across the
//     wavefronts (fixed order: bitwise reproducible); wavefront 0 solves the 2x2
//     (or 3x3) system and broadcasts the pose through LDS (every wavefront
//     solving for itself was measured slower: DESIGN.md 4.1);
//   * batches pick their form by size: a handful of scans are each spread over
//     many workgroups of one persistent launch (icp_single.hip), from two scans
//     per CU on two scans share a workgroup and its LDS index
//     (icp_fit_pair_kernel), one workgroup per scan in between;
//   * the float distance is fl(fl(dx*dx)+fl(dy*dy)) with contraction off and
//     the query is (float)(double transform), as icpPointToPoint.cpp:69-70 and
//     kdtree.cpp:610-612 compute them, so the correspondence set is the
//     reference's (ties: lowest original index).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"

#include "icp_search.hpp"

using namespace slam;
using namespace slam::icp;

// In-kernel stamps, the two-launch schedule with events and the environment knobs of the round-1 sweeps exist
// only in a measurement build (python -m slam_amd.build --measure: -DSLAM_MEASURE, a library of its own that
// the tools/ scripts load); the shipped library reads no environment variable and executes no stamp.
#ifndef SLAM_SEED_RING
#define SLAM_SEED_RING 1
#endif
#ifndef SLAM_SEED_EMPTY
#define SLAM_SEED_EMPTY 0
#endif
constexpr bool kSeedEmpty = SLAM_SEED_EMPTY != 0; // ... and skip the cells inside the disk the last search proved empty
// The ring-search form of the batch kernels starts a query's search from last iteration's neighbour of the same scene
// point (exact: nn_search_seeded).  Measured on config 2 (round 2): 0.394 -> 0.385 ms; with the cells
// inside the radius the last search proved empty skipped as well (SLAM_SEED_EMPTY): 0.398 ms -- the bookkeeping costs
// more than the skipped cells, so that half stays off here (the spread form, icp_single.hip, uses both).
constexpr bool kSeedRing = SLAM_SEED_RING != 0;
// Round 4, measured and NOT kept (off; -DSLAM_SEED_CHAIN=1 in a measurement build brings it back): a lane's points as CONSECUTIVE
// beams -- lane group g takes points g * passes + k -- so that a search without a seed of its own (every search of iteration 0;
// the passes past the three whose seeds stay in registers) starts from the neighbour the lane has just found for the beam
// before, of the same class.  Exact like every seed, and slower: 0.495 against 0.410 ms for 256 scans one per workgroup,
// 0.868 against 0.639 ms in pairs (profiles/r04_seed_chain_ab.txt).  With consecutive lane groups per pass a wavefront's 32
// queries are 32 adjacent beams -- the same few cells, the same ring depth, LDS reads of neighbouring words; with consecutive
// beams per lane they are every second or third beam of a sector twice as wide, and what the seeds save in iterations 0-1 the
// divergence costs in all of them.
#ifndef SLAM_SEED_CHAIN
#define SLAM_SEED_CHAIN 0
#endif
constexpr bool kSeedChain = kSeedRing && SLAM_SEED_CHAIN != 0;
// Round 4, measured and NOT kept (off; -DSLAM_LIST_SEED=1 in a measurement build brings it back): the LIST form starting a query's
// sweep at the entry it ended on in the iteration before instead of bisecting the list for the query's key.  Exact (the window
// around any start passes the same key-distance tests), and worth 1-2 % (256 scans in pairs 0.622 against 0.634 ms, 1024 scans
// 1.207 against 1.219) for one to three spilled registers in the point-to-point pair kernels: the bisection is four or five
// LDS reads of a chain that has a dozen, not the third of it that the cycle stamps suggested.
#ifndef SLAM_LIST_SEED
#define SLAM_LIST_SEED 0
#endif
constexpr bool kListSeed = SLAM_LIST_SEED != 0;
// The ring passes scan cell ranges in the packed-distance, second-best form (icp_search.hpp, `PK`): round 4, -4.5 % per launch.
constexpr bool kTunedScan = true;
#ifdef SLAM_MEASURE
#define SLAM_STAMPS(fa) ((fa).stamps != nullptr)
#else
#define SLAM_STAMPS(fa) false
#endif

namespace {

// List-sweep mode: the query's own cell of the list lattice, one sweep over its halo list, certified when
// the best distance is below the halo radius.  Returns false (undecided) otherwise.
struct ListPtrs {
    const float2         *pts;
    const unsigned short *start[2];
    const unsigned       *axis[2];
};

__device__ inline ListPtrs make_list_ptrs(const unsigned char *base, const ModelView &mv)
{
    ListPtrs lp;
    lp.pts = reinterpret_cast<const float2 *>(base + mv.loff_pts);
    lp.start[0] = reinterpret_cast<const unsigned short *>(base + mv.loff_start[0]);
    lp.start[1] = reinterpret_cast<const unsigned short *>(base + mv.loff_start[1]);
    lp.axis[0] = reinterpret_cast<const unsigned *>(base + mv.loff_axis[0]);
    lp.axis[1] = reinterpret_cast<const unsigned *>(base + mv.loff_axis[1]);
    return lp;
}

// the same with the direction vector already in registers (one multiply-add pair per key, no selects)
__device__ inline float list_key_u(float ux, float uy, float x, float y) { return __fadd_rn(__fmul_rn(ux, x), __fmul_rn(uy, y)); }

// Entries examined on either side of the refined start, branch-free, and further steps on either side (a loop the whole wavefront
// runs while any of its lanes needs it) before the query is left to the cooperative round.  Re-tuned in round 4 once a candidate
// had become four instructions cheaper: a window of 7 / 9 / 11 / 13 entries -> 0.573 / 0.564 / 0.560 / 0.561 ms per launch of 256
// scans in pairs (one scan per workgroup 0.373 / 0.368 / 0.366 / 0.363); 2 steps instead of 3: the same.
constexpr int kListWin = 5;
constexpr int kListWalk = 3;

// An exact tie is noticed through the SECOND best of everything examined (one v_med3_f32 per candidate; a tie is second == best at
// the end) -- round 4: the flag this replaced (set whenever a candidate equalled the best so far: a compare, an AND and an OR per
// candidate) cost 2.5 % of a pair launch.  Either way a tie sends the query to the exact pass.
__device__ inline bool list_search(Best &b, float2 &m, const ListPtrs &lp, const ModelView &mv, int cls, float qx, float qy, int seed = -1)
{
    const Lattice &L = mv.llat;
    b.d = FLT_MAX;
    b.oidx = 0xffffffffu;
    b.pos = -1;
    const float fx = (qx - L.x0) * L.inv_h, fy = (qy - L.y0) * L.inv_h;
    if (!(fx >= 0.f && fx < (float)L.nx && fy >= 0.f && fy < (float)L.ny)) return false; // outside (or NaN)
    const int             cx = min((int)fx, L.nx - 1), cy = min((int)fy, L.ny - 1), c = cy * L.nx + cx;
    const unsigned short *start = lp.start[cls];
    const float2         *pts = lp.pts + mv.lbase[cls];
    const int             a = (int)start[c], e = (int)start[c + 1], n = e - a;
    if (n <= 0) return false;
    const int   dir = (int)((lp.axis[cls][c >> 4] >> (2 * (c & 15))) & 3u);
    const float ux = dir == 1 ? 0.0f : 1.0f, uy = dir == 0 ? 0.0f : (dir == 3 ? -1.0f : 1.0f);
    const float kq = list_key_u(ux, uy, qx, qy);
    // A key difference dk bounds the distance from below: |p - q| >= |dk| along an axis, >= |dk| / sqrt(2) along
    // a diagonal, where the rounded sums also cost a few ulp (keps).  far(dk) = "nothing at this key distance or
    // beyond can beat or tie the best".
    const float kscale = dir < 2 ? 1.0f : 0.4999f, keps = dir < 2 ? 0.0f : mv.lkeps;
    auto        far = [&](float dk, float best) {
        const float lb = fmaxf(fabsf(dk) - keps, 0.0f);
        return __fmul_rn(__fmul_rn(lb, lb), kscale) > best;
    };
    // start: the first entry whose key is not below the query's (binary search: the keys of a list bent around
    // a corner are far from evenly spaced, an interpolated start can be dozens of entries off)
    int g = seed;
    if (!(kListSeed && seed >= a && seed < e)) { // (a seed from another cell's list: the query has crossed a cell border)
        int blo = a, bhi = e;
        while (blo < bhi) {
            const int    mid = (blo + bhi) >> 1;
            const float2 pm = pts[mid];
            if (list_key_u(ux, uy, pm.x, pm.y) < kq)
                blo = mid + 1;
            else
                bhi = mid;
        }
        g = min(blo, e - 1);
    }
    const int lo = max(a, g - kListWin), hi = min(e - 1, g + kListWin);
    float d = FLT_MAX, d2nd = FLT_MAX, klo = 0.f, khi = 0.f;
    int   pos = -1;
#pragma unroll
    for (int j = 0; j <= 2 * kListWin; ++j) {
        const int    i = lo + j;
        const bool   ok = i <= hi;
        const float2 p = pts[min(i, hi)];
        const float  dj = ok ? dist2(p, qx, qy) : FLT_MAX;
        const float  kj = list_key_u(ux, uy, p.x, p.y);
        if (j == 0) klo = kj;
        khi = ok ? kj : khi;
        d2nd = __builtin_amdgcn_fmed3f(d, dj, d2nd);
        const bool up = dj < d;
        d = up ? dj : d;
        pos = up ? i : pos;
    }
    // beyond an end whose key distance alone rules it out (on the far side of the query) nothing can beat or
    // tie the best; otherwise walk on from that end
    const float dl = klo - kq, dr = khi - kq;
    bool        Lft = (lo > a) & !((dl < 0.f) & far(dl, d));
    bool        Rgt = (hi < e - 1) & !((dr > 0.f) & far(dr, d));
    int         il = lo - 1, ir = hi + 1;
    // a bounded walk: a query that needs more is left undecided rather than holding its wavefront -- and at the
    // barrier its workgroup -- back
    int budget = kListWalk;
    while ((Lft | Rgt) & (budget > 0)) {
        --budget;
        const float2 ml = pts[max(il, a)], mr = pts[min(ir, e - 1)];
        {
            const float dk = list_key_u(ux, uy, ml.x, ml.y) - kq;
            const bool  in = Lft & !((dk < 0.f) & far(dk, d));
            const float dd = in ? dist2(ml, qx, qy) : FLT_MAX;
            d2nd = __builtin_amdgcn_fmed3f(d, dd, d2nd);
            const bool up = dd < d;
            d = up ? dd : d;
            pos = up ? il : pos;
            --il;
            Lft = in & (il >= a);
        }
        {
            const float dk = list_key_u(ux, uy, mr.x, mr.y) - kq;
            const bool  in = Rgt & !((dk > 0.f) & far(dk, d));
            const float dd = in ? dist2(mr, qx, qy) : FLT_MAX;
            d2nd = __builtin_amdgcn_fmed3f(d, dd, d2nd);
            const bool up = dd < d;
            d = up ? dd : d;
            pos = up ? ir : pos;
            ++ir;
            Rgt = in & (ir < e);
        }
    }
    b.d = d;
    b.pos = pos;
    if (d2nd == d || pos < 0 || (Lft | Rgt) || !(d < mv.cert2)) return false;
    m = pts[pos];
    return true;
}

// The whole halo list of the query's cell scanned by kCoop lanes (the list is in LDS; a handful of entries a lane):
// the cooperative round's first try for the scan's tail and for queries whose bounded walk gave up.
__device__ inline bool list_scan_coop(Best &b, float2 &m, const ListPtrs &lp, const ModelView &mv, int cls, float qx, float qy,
                                   int sub)
{
    const Lattice &L = mv.llat;
    const float    fx = (qx - L.x0) * L.inv_h, fy = (qy - L.y0) * L.inv_h;
    const bool     inside = fx >= 0.f && fx < (float)L.nx && fy >= 0.f && fy < (float)L.ny;
    const int      cx = inside ? min((int)fx, L.nx - 1) : 0, cy = inside ? min((int)fy, L.ny - 1) : 0, c = cy * L.nx + cx;
    const float2  *pts = lp.pts + mv.lbase[cls];
    const int      a = (int)lp.start[cls][c], e = inside ? (int)lp.start[cls][c + 1] : a;
    float          d = FLT_MAX;
    int            pos = -1;
    bool           tie = false;
    for (int i = a + sub; i < e; i += kCoop) {
        const float di = dist2(pts[i], qx, qy);
        tie |= (di == d);
        const bool up = di < d;
        d = up ? di : d;
        pos = up ? i : pos;
    }
    b.d = d;
    b.pos = pos;
    b.oidx = 0xffffffffu;
    group_min_lean<kCoop>(b, tie);
    d = b.d;
    pos = b.pos;
    if (tie || pos < 0 || !(d < mv.cert2)) return false;
    m = pts[pos];
    return true;
}

// The threads that work on one scan (TeamDims, icp_model.hpp).  sync() is the workgroup's s_barrier when the team is the
// workgroup; a smaller team meets at a counter in LDS (gfx950 has no named barriers): every wavefront adds one and
// sleeps until the count of the team's wavefronts times the barriers passed is reached.  LDS traffic of a wavefront
// is complete before its add (release) and the others' is visible after the wait (acquire).
template <int TB>
struct Team {
    int       tid;   // thread within the team
    unsigned *bar;   // TB < kBlock: the team's barrier word in LDS
    unsigned  epoch; // wavefront arrivals that end the next barrier
    __device__ inline void sync()
    {
        if (TB == kBlock) {
            __syncthreads();
        } else {
            epoch += (unsigned)(TB / 64);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if ((tid & 63) == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
};

// List-sweep kernel, one pass over the team's TB points from p0, one lane per point.  A query the list cannot
// decide is not searched by its own lane (the whole wavefront, and at the barrier the whole workgroup,
// would wait for a few lanes): its offset goes to the wavefront's region of a queue in LDS (fixed regions:
// the order does not depend on timing, so sums stay bitwise reproducible) that all wavefronts drain
// together afterwards (drain_queue).
template <int MODE>
__device__ inline void list_pass(const ListPtrs &lp, const ModelView &mv, const FitArgs &fa, const Pose &T, int n, int nga,
                                 int p0, const double2 P, double acc[kNumAcc], unsigned *wave_cnt, unsigned short *queue,
                                 int &fell_back, int tid, int &lseed)
{
    const int  lane = tid & 63, wave = tid >> 6;
    const int  p = p0 + tid;
    const int  cls = MODE == SLAM_ICP_P2L ? 1 : (p < nga ? 0 : 1);                  // a point-to-line model is one class
    const bool valid = p < n && (MODE == SLAM_ICP_P2L || mv.n_cls[cls] > 3); // icpPointToPoint.cpp:59,93
    bool       done = true;
    if (valid) {
        float  qx, qy;
        Best   b;
        float2 m;
        transform_query(T, P, qx, qy);
        done = list_search(b, m, lp, mv, cls, qx, qy, lseed);
        lseed = done ? b.pos : -1;
        if (MODE == SLAM_ICP_P2L) { // icpPointToPlane.cpp:55-77: every template point, no gate
            if (done) add_p2l(m, mv.lnormals[mv.lbase[1] + b.pos], qx, qy, acc);
        } else if (done && (double)b.d < fa.indist) {
            add_p2p_xy(mv, m, qx, qy, acc); // :76
        }
        fell_back += done ? 0 : 1;
    }
    const unsigned long long need = __ballot(!done);
    if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(need);
    if (!done) queue[wave * 64 + __popcll(need & ((1ull << lane) - 1ull))] = (unsigned short)tid;
}

// All wavefronts take kCoopPerWave points at a time and search each with kCoop lanes: first the `tail`
// points past the pass (block_search, then the ring search if that does not decide), then the queued
// ones (ring search).  Entry e of the queue lives in the region of the wavefront whose inclusive count
// prefix first exceeds e.
template <typename StartT, bool LISTS, int TB, int MODE>
__device__ inline void drain_queue(const IndexPtrs<StartT> &ix, const ListPtrs &lp, const ModelView &mv, const FitArgs &fa,
                                   const Pose &T, int off, int n, int nga, int p0, double acc[kNumAcc], int tail,
                                   const unsigned *wave_cnt, const unsigned short *queue, int tid)
{
    constexpr int kW = TeamDims<TB>::kW, kCoopBlock = TeamDims<TB>::kCoopBlock; // kW <= 16: one count per lane of a DPP row
    const int lane = tid & 63, wave = tid >> 6;
    const int mine = (int)wave_cnt[lane % kW];
    int       incl = mine; // inclusive prefix over the 16 wavefronts, in every DPP row
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true); // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
    const int queued = __builtin_amdgcn_readlane(incl, kW - 1);
    const int total = tail + queued;
    for (int base = 0; base < total; base += kCoopBlock) {
        const int g = lane / kCoop;
        const int e = base + wave * kCoopPerWave + g; // wavefront-major: a short round keeps few wavefronts busy (measured
                                                      // faster than dealing it round-robin over all sixteen)
        // wavefront region holding queue entry e - tail: the first whose inclusive prefix exceeds it
        int w = 0, excl = 0;
#pragma unroll
        for (int j = 0; j < kW; ++j) {
            const int  incl_j = __builtin_amdgcn_readlane(incl, j); // the same in every lane: a scalar
            const bool below = incl_j <= e - tail;
            w += below ? 1 : 0;
            excl = below ? incl_j : excl;
        }
        if (e < total) {
            const bool is_tail = e < tail;
            const int  p = p0 + (is_tail ? TB + e : (int)queue[min(w, kW - 1) * 64 + (e - tail - excl)]);
            const int  cls = MODE == SLAM_ICP_P2L ? 1 : (p < nga ? 0 : 1);
            if (MODE == SLAM_ICP_P2L || mv.n_cls[cls] > 3) { // icpPointToPoint.cpp:59,93
                const double gate = MODE == SLAM_ICP_P2L ? (double)INFINITY : fa.indist; // icpPointToPlane.cpp has no gate
                float        qx, qy;
                transform_query(T, fa.pts[off + p], qx, qy);
                Best   b;
                float2 m;
                bool   have_m = false;
                if (LISTS) {
                    have_m = list_scan_coop(b, m, lp, mv, cls, qx, qy, lane % kCoop);
                    if (!have_m) b = nn_search<kCoop, StartT>(ix, mv, cls, qx, qy, lane % kCoop, gate);
                } else {
                    b = nn_search<kCoop, StartT>(ix, mv, cls, qx, qy, lane % kCoop, gate);
                }
                if (MODE == SLAM_ICP_P2L) {
                    if (lane % kCoop == 0 && b.pos >= 0) {
                        double2 nrm;
                        if (have_m) {
                            nrm = mv.lnormals[mv.lbase[1] + b.pos];
                        } else {
                            m = ix.pts[mv.base[1] + b.pos];
                            nrm = reinterpret_cast<const double2 *>(mv.normals)[b.oidx];
                        }
                        add_p2l(m, nrm, qx, qy, acc);
                    }
                } else if (lane % kCoop == 0 && b.pos >= 0 && (double)b.d < fa.indist) { // :76
                    if (!have_m) m = ix.pts[mv.base[cls] + b.pos];
                    add_p2p_xy(mv, m, qx, qy, acc);
                }
            }
        }
    }
}

// One scene point, searched by GG lanes (sub = lane within that group); the
// group's lane 0 adds the correspondence to its running sums.
template <int GG, typename StartT, int MODE, bool SEEDED = false, bool PK = false>
__device__ inline void accumulate_point(const IndexPtrs<StartT> &ix, const ModelView &mv, const FitArgs &fa,
                                        const Pose &T, const double2 P, bool is_ga, int sub, double acc[kNumAcc],
                                        int &far, int *seed = nullptr, float *empty = nullptr, float move_r = 0.f, float move_t = 0.f)
{
    float qx, qy;
    transform_query(T, P, qx, qy);
    if (MODE == SLAM_ICP_P2P) {
        const int cls = is_ga ? 0 : 1;
        if (mv.n_cls[cls] > 3) { // icpPointToPoint.cpp:59,93
            Best b;
            if (SEEDED) { // last iteration's neighbour of this scene point prunes the search from the start
                const float move = move_r * (fabsf((float)P.x) + fabsf((float)P.y) + 1.0e-3f) + move_t;
                float       e_out;
                b = nn_search_seeded<GG, StartT, PK>(ix, mv, cls, qx, qy, sub, fa.indist, *seed, kSeedEmpty ? *empty : 0.0f, move, e_out);
                *seed = b.pos;
                *empty = e_out;
            } else {
                b = nn_search<GG, StartT, PK>(ix, mv, cls, qx, qy, sub, fa.indist);
            }
            if (sub == 0 && b.pos >= 0 && (double)b.d < fa.indist) add_p2p<StartT>(ix, mv, cls, b, qx, qy, acc); // :76
            far += (sub == 0 && !(b.pos >= 0 && b.d < mv.cert2)) ? 1 : 0; // beyond what the halo lists certify
        }
    } else {
        // icpPointToPlane.cpp:55-77: single class (all model points are class 1 of the index, oidx = all-index), no inlier gate
        Best b;
        if (SEEDED) {
            float e_out;
            b = nn_search_seeded<GG, StartT, PK>(ix, mv, 1, qx, qy, sub, (double)INFINITY, *seed, 0.0f, 0.0f, e_out);
            *seed = b.pos;
        } else {
            b = nn_search<GG, StartT, PK>(ix, mv, 1, qx, qy, sub, (double)INFINITY);
        }
        if (sub == 0 && b.pos >= 0)
            add_p2l(ix.pts[mv.base[1] + b.pos], reinterpret_cast<const double2 *>(mv.normals)[b.oidx], qx, qy, acc);
        far += (sub == 0 && !(b.pos >= 0 && b.d < mv.cert2)) ? 1 : 0; // beyond what the halo lists certify
    }
}

// accumulate_point for an index in HBM/L2 with the wavefront's model tile in LDS (icp_search.hpp): called by EVERY lane of the
// wavefront (`valid` = the lane has a point), because the staging is the wavefront's.  A seeded query whose disk touches at most
// 3 x 3 cells is searched in the tile; everything else -- no seed yet, a large disk, the other class, a block too large to
// stage, an exact tie -- takes the ordinary searches on the index in L2.  Same neighbour either way.  The seed is last
// iteration's neighbour: its position AND its coordinates (registers: no load to learn the disk).
template <int GG, typename StartT, int MODE, bool PK>
__device__ __forceinline__ void accumulate_point_tile(const WaveTile &wt, TileState &ts, const IndexPtrs<StartT> &ix, const ModelView &mv,
                                                      const FitArgs &fa, const Pose &T, const double2 P, bool valid, bool is_ga, int sub,
                                                      double acc[kNumAcc], int &far, int &seed, float2 &seed_xy, bool allow)
{
    static_assert(PK, "the tile scan is written for the packed, second-best form");
    float qx, qy;
    transform_query(T, P, qx, qy);
    const int    cls = MODE == SLAM_ICP_P2L ? 1 : (is_ga ? 0 : 1);
    const bool   ok = valid && (MODE == SLAM_ICP_P2L || mv.n_cls[cls] > 3); // icpPointToPoint.cpp:59,93
    const double gate = MODE == SLAM_ICP_P2L ? (double)INFINITY : fa.indist;
    Best         b;
    b.d = FLT_MAX;
    b.oidx = 0xffffffffu;
    b.pos = -1;
    SeedBox bx = {0, -1, 0, -1, false};
    if (ok && seed >= 0) {
        b.d = ulp_above(dist2(seed_xy, qx, qy));
        b.pos = seed;
        bx = seed_box(mv.lat, qx, qy, b.d);
    }
    const bool in_tile = wave_tile_ready<StartT>(wt, ts, ix, mv, bx.small && allow, cls, bx);
    if (!ok) return;
    float2 m;
    bool   have_m = false;
    if (in_tile) {
        bool tie = false;
        wave_tile_scan<GG, PK>(b, m, tie, wt, ts, bx, sub, qx, qy);
        have_m = !tie;
        if (tie) { // rare: the plain exact search, as in nn_search_seeded
            bool unused = false;
            b = nn_search_impl<GG, StartT, true>(ix, mv, cls, qx, qy, sub, gate, unused);
        } else if (MODE == SLAM_ICP_P2L) {
            b.oidx = (unsigned)ix.oidx[mv.base[1] + b.pos]; // (the normals are indexed by it)
        }
    } else {
        float e_out;
        b = nn_search_seeded<GG, StartT, PK>(ix, mv, cls, qx, qy, sub, gate, seed, 0.0f, 0.0f, e_out);
    }
    if (!have_m && b.pos >= 0) m = ix.pts[mv.base[cls] + b.pos];
    seed = b.pos;
    seed_xy = m;
    if (MODE == SLAM_ICP_P2P) {
        if (sub == 0 && b.pos >= 0 && (double)b.d < fa.indist) add_p2p_xy(mv, m, qx, qy, acc); // :76
    } else if (sub == 0 && b.pos >= 0) {
        add_p2l(m, reinterpret_cast<const double2 *>(mv.normals)[b.oidx], qx, qy, acc);
    }
    far += (sub == 0 && !(b.pos >= 0 && b.d < mv.cert2)) ? 1 : 0;
}

// One pass of the workgroup over kBlock/GG consecutive scene points from p0.
template <int GG, typename StartT, int MODE>
__device__ inline void point_pass(const IndexPtrs<StartT> &ix, const ModelView &mv, const FitArgs &fa,
                                  const Pose &T, int off, int n, int nga, int p0, double acc[kNumAcc], int tid)
{
    const int p = p0 + tid / GG;
    int       far = 0;
    if (p < n) accumulate_point<GG, StartT, MODE, false, kTunedScan>(ix, mv, fa, T, fa.pts[off + p], p < nga, tid % GG, acc, far);
}

// The same with the lane's point already in registers (the first kHoist passes: a lane meets the same
// points in every iteration, so they are loaded once per scan, not once per iteration).
template <int GG, typename StartT, int MODE, bool PK = false>
__device__ inline void point_pass_reg(const IndexPtrs<StartT> &ix, const ModelView &mv, const FitArgs &fa,
                                      const Pose &T, int n, int nga, int p, const double2 P, double acc[kNumAcc],
                                      int &far, int tid, int *seed = nullptr, float *empty = nullptr, float move_r = 0.f, float move_t = 0.f)
{
    if (p < n) {
        if (seed)
            accumulate_point<GG, StartT, MODE, true, PK>(ix, mv, fa, T, P, p < nga, tid % GG, acc, far, seed, empty, move_r, move_t);
        else
            accumulate_point<GG, StartT, MODE, false, PK>(ix, mv, fa, T, P, p < nga, tid % GG, acc, far);
    }
}

// One workgroup = one scan, all iterations.  MODE: SLAM_ICP_P2P / SLAM_ICP_P2L.
// G = lanes per scene point; G = 0 picks it per pass: one lane per point while
// more than half a workgroup of points is left, then the widest group that
// still covers the rest in one pass (a 1081-point scan is 1024 points at G = 1
// plus 57 points at G = 16), so no pass runs nearly empty.
// Iterations fs.iters .. max_iter-1 of the workgroup's scan in one search form (SWEEP 0: ring search with G lanes
// per point on the cell index `ix`; SWEEP 2: list sweeps on the halo lists `lp`, the undecided few on `ix`).
// phase 1 stops from fa.switch_iter on with fs.hand_over set as soon as the list form can take over (see the guard).
template <int G, typename StartT, int MODE, int SWEEP, int TB = kBlock, bool TILE = false>
__device__ __forceinline__ void fit_iterations(const ModelView &mv, const FitArgs &fa, const IndexPtrs<StartT> &ix,
                                               const ListPtrs &lp, unsigned char *smem /* the team's scratch */, int s, int off,
                                               int n, int nga, int phase, FitState &fs, Team<TB> &tm)
{
    static_assert(!TILE || (TB == icp::kBlock && SWEEP == 0 && G >= 1), "wave tiles: the ring form of a whole workgroup");
    constexpr int      kWaves = TeamDims<TB>::kW, kBlock = TB, kCoopPerBlock = TeamDims<TB>::kCoopBlock; // of the TEAM, from here on
    constexpr unsigned kReduceBytes = TeamDims<TB>::kReduce;
    double *partial = reinterpret_cast<double *>(smem); // [2][kWaves][kNumAcc]
    double *bcast = partial + 2 * kWaves * kNumAcc;     // [2][8] new pose, delta, n_corr
    unsigned       *wave_cnt = reinterpret_cast<unsigned *>(smem + kReduceBytes); // [kWaves] undecided queries of the pass
    unsigned short *queue = reinterpret_cast<unsigned short *>(smem + kReduceBytes + 4 * kWaves); // [kWaves][64]
    const int tid = tm.tid;
    const int lane = tid & 63, wave = tid >> 6;
    // TILE: behind the scratch, one region per wavefront for the model tile of its pass (icp_search.hpp)
    const WaveTile wtile = wave_tile_at(smem + TeamDims<TB>::kScratch + (TILE ? (unsigned)wave * kWaveTileBytes : 0u));
    const int iter_begin = fs.iters;
    double    r00 = fs.r00, r01 = fs.r01, r10 = fs.r10, r11 = fs.r11, t0 = fs.t0, t1 = fs.t1, delta = fs.delta;
    int       iters = fs.iters, n_corr = fs.n_corr;

    // the lane's points of the first passes, loaded once
    constexpr int kPerPass = kBlock / (G > 0 ? G : 1);
    // point of this lane in ring pass k: consecutive beams per lane group (kSeedChain), else consecutive lane groups per pass
    // (the first n % kPerPass lane groups take one beam more than the others: the last pass is as short as it was with
    // consecutive lane groups per pass -- 55 points in two wavefronts for a 1081-beam scan -- not a pass of every wavefront)
    const int     n_full = n / kPerPass, n_rest = n % kPerPass;
    // TILE: the two passes of a WAVEFRONT are neighbours -- it takes 64 / G adjacent beams in pass 0 and the next 64 / G in pass 1,
    // so that one tile in LDS covers both (a scan of more than one pass; the same points per pass as a workgroup either way)
    const bool    tile_pairs = TILE && G > 0 && n > kPerPass;
    const auto    ring_point = [&](int k) {
        const int g = tid / (G > 0 ? G : 1);
        if (tile_pairs && k < 2) return (2 * wave + k) * (64 / (G > 0 ? G : 1)) + (lane / (G > 0 ? G : 1));
        if (!kSeedChain) return k * kPerPass + g;
        return (k < n_full || (k == n_full && g < n_rest)) ? g * n_full + min(g, n_rest) + k : n; // n: none
    };
    const auto    hoisted = [&](int k) {
        const int p = SWEEP ? k * kPerPass + tid : ring_point(k);
        return (G > 0 && p < n) ? fa.pts[off + p] : make_double2(0.0, 0.0);
    };
    // named, not an array: stays in registers (the list-sweep kernel has one full pass per 1024 points: one is enough)
    // (round 3, measured and not kept: a team of half a workgroup needs five ring passes for a 1081-point scan and seeds only
    // the three it keeps in registers; one pass of points kept and the seeds of all five -- nine registers instead of fifteen
    // -- made no difference with two lanes per point (0.620 / 0.630 / 1.195 ms for 256 / 512 / 1024 scans against 0.627 /
    // 0.626 / 1.199) and cost 3 % with one)
    // (the list form of a pair's team with 32-bit starts -- the cell index of a large model, read from HBM by the undecided few --
    // is the one instantiation that does not fit 128 registers with the pass's point kept: there the point is loaded per
    // iteration, which ended its 3 spilled registers / 16 bytes of scratch per lane, round 4)
    constexpr bool kKeepPoint = !(SWEEP && sizeof(StartT) == 4 && TB < icp::kBlock);
    const double2  Pc0 = kKeepPoint ? hoisted(0) : make_double2(0.0, 0.0), Pc1 = SWEEP ? Pc0 : hoisted(1), Pc2 = SWEEP ? Pc0 : hoisted(2);

    TileState ts = {0, 0, -1, -1, 0, false, false, false};               // TILE: the wavefront's tile, kept over the iterations
    float2    sxy0 = make_float2(0.f, 0.f), sxy1 = make_float2(0.f, 0.f); // ... and the coordinates of the seeds of its two passes
    int   sd0 = -1, sd1 = -1, sd2 = -1; // ring form: last iteration's neighbour of the lane's point in each hoisted pass
    float em0 = 0.f, em1 = 0.f, em2 = 0.f, move_r = 0.f, move_t = 0.f; // ... the radius it proved empty; the last step's size
    bool hand_over = false;
    iters = iter_begin;
    if (n >= 5) { // icp.cpp:100-103
        for (int iter = iter_begin; iter < fa.max_iter; ++iter) {
            double acc[kNumAcc];
#pragma unroll
            for (int k = 0; k < kNumAcc; ++k) acc[k] = 0.0;
            const Pose T = {r00, r01, r10, r11, t0, t1};
            if (fa.step_pose && tid == 0) {
                // (the scan index is the same in every lane of a wavefront; said so HERE, the address is formed in scalar
                // registers inside this rarely taken block -- left to itself the compiler forms it once per kernel in vector
                // registers, which in the pair kernel, where the scan depends on the team, was what went to scratch)
                double *sp = fa.step_pose + 6 * (size_t)__builtin_amdgcn_readfirstlane(s);
                sp[0] = r00;
                sp[1] = r01;
                sp[2] = r10;
                sp[3] = r11;
                sp[4] = t0;
                sp[5] = t1;
            }
            long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c_mid = 0;
            int       fell_back = 0; // diagnostic: sweep queries of this lane that went to the ring search
            int       far = 0;       // queries of this lane whose neighbour is beyond the halo lists' certified radius
            if (SLAM_STAMPS(fa)) c0 = __builtin_amdgcn_s_memtime();

            int pass = 0, chain = -1; // chain: the neighbour found in the pass before (kSeedChain)
            for (int p0 = 0; p0 < n; ++pass) {
                int rem = n - p0;
                if (SWEEP) {
                    // a pass of kBlock points, then the cooperative rounds: the queries the pass left
                    // undecided and, when fewer than kCoopPerBlock points remain after it, those too
                    int tail = 0;
                    if (rem > kCoopPerBlock) {
                        const double2 P = (pass == 0 && kKeepPoint) ? Pc0 : (p0 + tid < n ? fa.pts[off + p0 + tid] : Pc0);
                        // (the lane's entry of the iteration before, per pass: the ring form's seed registers are free here)
                        int ls = kListSeed ? (pass == 0 ? sd0 : (pass == 1 ? sd1 : -1)) : -1;
                        list_pass<MODE>(lp, mv, fa, T, n, nga, p0, P, acc, wave_cnt, queue, fell_back, tid, ls);
                        if (kListSeed) sd0 = pass == 0 ? ls : sd0, sd1 = pass == 1 ? ls : sd1;
                        tail = rem - kBlock;
                        tail = tail > 0 && tail <= kCoopPerBlock ? tail : 0;
                    } else { // a scan shorter than one cooperative round
                        if (lane == 0) wave_cnt[wave] = 0;
                        tail = rem;
                        p0 -= kBlock; // the tail is addressed as p0 + kBlock + i
                    }
                    tm.sync();
                    if (SLAM_STAMPS(fa)) c_mid = __builtin_amdgcn_s_memtime();
                    drain_queue<StartT, SWEEP == 2, TB, MODE>(ix, lp, mv, fa, T, off, n, nga, p0, acc, tail, wave_cnt, queue, tid);
                    p0 += kBlock + tail;
                    if (p0 < n) tm.sync(); // the queue is reused by the next pass
                } else if (G > 0) {
                    const int p = ring_point(pass);
                    double2   P = pass == 0 ? Pc0 : (pass == 1 ? Pc1 : Pc2);
                    if (pass >= kHoist) P = fa.pts[off + min(p, n - 1)];
                    if (kSeedRing && (pass < kHoist || kSeedChain)) {
                        // (the per-pass state is selected by value: a pointer into the three would put them on the stack)
                        int   sd = pass == 0 ? sd0 : (pass == 1 ? sd1 : (pass == 2 ? sd2 : -1));
                        float em = pass == 0 ? em0 : (pass == 1 ? em1 : em2);
                        if (kSeedChain) {
                            // no seed of its own: the neighbour just found for the beam before (the lane's point of the pass before), if
                            // that beam is of the same class -- a seed is a position in the class's sorted array
                            const bool same_cls = MODE == SLAM_ICP_P2L || ((p - 1 < nga) == (p < nga));
                            if (sd < 0 && pass > 0 && same_cls) sd = chain;
                        }
                        if (TILE && pass < 2) {
                            float2 sxy = pass == 0 ? sxy0 : sxy1;
                            accumulate_point_tile<(G > 0 ? G : 1), StartT, MODE, kTunedScan>(wtile, ts, ix, mv, fa, T, P, p < n, p < nga, tid % (G > 0 ? G : 1), acc, far, sd, sxy, iter >= iter_begin + kTileFirstIter);
                            sxy0 = pass == 0 ? sxy : sxy0, sxy1 = pass == 1 ? sxy : sxy1;
                        } else
                            point_pass_reg<(G > 0 ? G : 1), StartT, MODE, kTunedScan>(ix, mv, fa, T, n, nga, p, P, acc, far, tid, &sd, &em, move_r, move_t);
                        sd0 = pass == 0 ? sd : sd0, sd1 = pass == 1 ? sd : sd1, sd2 = pass == 2 ? sd : sd2;
                        if (kSeedEmpty) em0 = pass == 0 ? em : em0, em1 = pass == 1 ? em : em1, em2 = pass == 2 ? em : em2;
                        chain = p < n ? sd : -1;
                    } else {
                        point_pass_reg<(G > 0 ? G : 1), StartT, MODE, kTunedScan>(ix, mv, fa, T, n, nga, p, P, acc, far, tid);
                    }
                    p0 += kBlock / (G > 0 ? G : 1);
                } else if (rem * 2 > kBlock) {
                    point_pass<1, StartT, MODE>(ix, mv, fa, T, off, n, nga, p0, acc, tid);
                    p0 += kBlock;
                } else if (rem * 4 > kBlock) {
                    point_pass<2, StartT, MODE>(ix, mv, fa, T, off, n, nga, p0, acc, tid);
                    p0 += kBlock / 2;
                } else if (rem * 8 > kBlock) {
                    point_pass<4, StartT, MODE>(ix, mv, fa, T, off, n, nga, p0, acc, tid);
                    p0 += kBlock / 4;
                } else if (rem * 16 > kBlock) {
                    point_pass<8, StartT, MODE>(ix, mv, fa, T, off, n, nga, p0, acc, tid);
                    p0 += kBlock / 8;
                } else {
                    point_pass<16, StartT, MODE>(ix, mv, fa, T, off, n, nga, p0, acc, tid);
                    p0 += kBlock / 16;
                }
            }

            const bool at_switch = phase == 1 && iter + 1 >= fa.switch_iter && iter + 1 < fa.max_iter;
            if (at_switch) { // how many queries the list sweeps could not certify right now
                for (int o = 32; o > 0; o >>= 1) far += __shfl_xor(far, o);
                if (lane == 0) wave_cnt[wave] = (unsigned)far;
            }
            if (SLAM_STAMPS(fa)) c1 = __builtin_amdgcn_s_memtime();
            // wavefront reduction on the DPP path, then LDS across the 16 wavefronts
            double *my = partial + ((iter & 1) * kWaves + wave) * kNumAcc;
            {
                const double v8 = wave_sum8(acc), v9 = wave_sum(acc[8]);
                if ((lane & 7) == 0) my[lane >> 3] = v8;
                if (lane == 0) my[8] = v9;
            }
            if (SLAM_STAMPS(fa)) c2 = __builtin_amdgcn_s_memtime();
            tm.sync();
            if (SLAM_STAMPS(fa)) c3 = __builtin_amdgcn_s_memtime();

            // wavefront 0 alone adds the 16 partials (fixed order: bitwise reproducible)
            // and solves; the others wait at the barrier below and read the new pose.
            double *bc = bcast + (iter & 1) * 8;
            if (wave == 0) {
                const double *all = partial + (iter & 1) * kWaves * kNumAcc;
                double        mine = 0.0;
                if (lane < kNumAcc) {
                    // (all loads first, then the additions in the same fixed order: left as one loop the compiler waits for each
                    // pair of partials before it adds them -- kWaves / 2 LDS round trips on the one wavefront everybody waits for)
                    double v[kWaves];
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) v[w] = all[w * kNumAcc + lane];
                    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): every partial has arrived
#pragma unroll
                    for (int w = 0; w < kWaves; ++w) mine += v[w];
                }
                double S[kNumAcc];
#pragma unroll
                for (int k = 0; k < kNumAcc; ++k)
                    S[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), k),
                                            __builtin_amdgcn_readlane(__double2loint(mine), k));

                double o[6] = {r00, r01, r10, r11, t0, t1};
                double d_out = 0.0;
                int    nc_out = 0;
                if (MODE == SLAM_ICP_P2P) {
                    d_out = p2p_step(S, mv, o, nc_out);
                } else {
                    nc_out = n; // every template point has a correspondence (icpPointToPlane.cpp:55-77)
                    d_out = p2l_step(S, o);
                }
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) bc[k] = o[k];
                    bc[6] = d_out;
                    bc[7] = (double)nc_out;
                }
                if (at_switch) {
                    // The hand-over decision is taken HERE, between the two barriers of the iteration: every
                    // wavefront wrote its count before the first, none writes the next iteration's before the
                    // second.  The total goes out through slot 0 of this iteration's partial sums, which
                    // wavefront 0 has finished reading and nobody writes again before the barriers of iteration
                    // + 1 (so every wavefront reads ONE value and the workgroup decides uniformly).
                    unsigned far_all = 0;
                    for (int w = 0; w < kWaves; ++w) far_all += wave_cnt[w];
                    if (lane == 0) partial[(iter & 1) * kWaves * kNumAcc] = (double)far_all;
                }
            }
            tm.sync();
            if (kSeedRing && kSeedEmpty && !SWEEP) { // |q_new - q_old| <= |dR|_F |p| + |dt| (rounded up)
                const double n00 = uniform(bc[0]), n01 = uniform(bc[1]), n10 = uniform(bc[2]), n11 = uniform(bc[3]);
                const double n4 = uniform(bc[4]), n5 = uniform(bc[5]);
                move_r = (float)sqrt((n00 - r00) * (n00 - r00) + (n01 - r01) * (n01 - r01) + (n10 - r10) * (n10 - r10) + (n11 - r11) * (n11 - r11)) * 1.0001f + 1.0e-7f;
                move_t = (float)sqrt((n4 - t0) * (n4 - t0) + (n5 - t1) * (n5 - t1)) * 1.0001f + 1.0e-7f;
            }
            r00 = uniform(bc[0]); // the same in every lane: keep the pose in scalar registers
            r01 = uniform(bc[1]);
            r10 = uniform(bc[2]);
            r11 = uniform(bc[3]);
            t0 = uniform(bc[4]);
            t1 = uniform(bc[5]);
            delta = uniform(bc[6]);
            n_corr = (int)uniform(bc[7]);
            ++iters;
            if (SLAM_STAMPS(fa)) {
                for (int o = 32; o > 0; o >>= 1) fell_back += __shfl_xor(fell_back, o);
                if ((tid & 63) == 0) {
                    const long long c4 = __builtin_amdgcn_s_memtime();
                    long long *st = fa.stamps + ((size_t)s * icp::kWaves + wave) * kStampSlots;
                    st[0] += c1 - c0;
                    st[1] += c2 - c1;
                    st[2] += c3 - c2;
                    st[3] += c4 - c3;
                    if (iter >= 6) st[4] += fell_back; // queries the sweeps left undecided, late iterations
                    if (iter >= 6 && c_mid) st[7] += c1 - c_mid; // cooperative rounds, late iterations
                    if (iter == 0) st[5] += c1 - c0;  // search time of the first iteration
                    if (iter >= 6) st[6] += c1 - c0;  // search time of the late iterations
                    if (c_mid) st[8] += c1 - c_mid;        // of which the cooperative rounds (sweep kernel)
                }
            }
            if (fa.trace && tid == 0) {
                double *tr = fa.trace + ((size_t)__builtin_amdgcn_readfirstlane(s) * fa.max_iter + iter) * 8;
                tr[0] = r00;
                tr[1] = r01;
                tr[2] = r10;
                tr[3] = r11;
                tr[4] = t0;
                tr[5] = t1;
                tr[6] = delta;
                tr[7] = (double)n_corr;
            }
            if (delta < fa.min_delta) break; // icp.cpp:119-121
            if (at_switch) {
                // hand over unless too many queries are still far from the map (outliers, a scan that has not
                // settled): each of them would cost the list form a ring search from HBM per iteration
                const unsigned far_all = (unsigned)uniform(partial[(iter & 1) * kWaves * kNumAcc]);
                if (far_all * (unsigned)fa.far_div <= (unsigned)n) {
                    hand_over = true; // the list sweeps continue from here
                    break;
                }
            }
        }
    }

    fs.r00 = r00;
    fs.r01 = r01;
    fs.r10 = r10;
    fs.r11 = r11;
    fs.t0 = t0;
    fs.t1 = t1;
    fs.delta = delta;
    fs.iters = iters;
    fs.n_corr = n_corr;
    fs.hand_over = hand_over;
}

// what a finished workgroup leaves behind: the pose in place, the result record, the hand-over state
__device__ inline void store_fit(const FitArgs &fa, int s, const FitState &fs, int phase)
{
    fa.R[4 * s + 0] = fs.r00;
    fa.R[4 * s + 1] = fs.r01;
    fa.R[4 * s + 2] = fs.r10;
    fa.R[4 * s + 3] = fs.r11;
    fa.t[2 * s + 0] = fs.t0;
    fa.t[2 * s + 1] = fs.t1;
    if (fa.result) {
        fa.result[s].iters = fs.iters;
        fa.result[s].n_corr = fs.n_corr;
        fa.result[s].delta = fs.delta;
    }
    if (phase == 1) fa.state[s] = fs.hand_over ? fs.iters : -1;
}

__device__ inline FitState load_fit(const FitArgs &fa, int s, int iter_begin)
{
    FitState fs;
    fs.r00 = uniform(fa.R0[4 * s + 0]);
    fs.r01 = uniform(fa.R0[4 * s + 1]);
    fs.r10 = uniform(fa.R0[4 * s + 2]);
    fs.r11 = uniform(fa.R0[4 * s + 3]);
    fs.t0 = uniform(fa.t0[2 * s + 0]);
    fs.t1 = uniform(fa.t0[2 * s + 1]);
    fs.delta = 0.0;
    fs.iters = iter_begin;
    fs.n_corr = 0;
    fs.hand_over = false;
    return fs;
}

template <int G, bool LDS, typename StartT, int MODE, int SWEEP, bool TILE = false>
__global__ __launch_bounds__(kBlock) void icp_fit_kernel(ModelView mv, FitArgs fa)
{
    static_assert(!(TILE && LDS), "a wave tile is for an index that is NOT in LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr unsigned kScratch = kScratchBytes;
    static_assert(kScratch % 16 == 0, "scratch keeps the blob 16-B aligned");
    const int s = blockIdx.x;
    if (fa.only && fa.only[s] == 0) return; // uniform for the workgroup
    const int off = fa.scan_off[s];
    const int n = fa.scan_off[s + 1] - off;
    const int nga = fa.scan_nga[s];
    int       iter_begin = 0;
    if (fa.phase == 2) {
        iter_begin = fa.state[s];
        if (iter_begin < 0) return; // finished in the ring-search launch (uniform for the workgroup)
    }
    const unsigned char *base = mv.blob;
    if (LDS) {
        fill_lds(smem + kScratch, mv.blob, mv.blob_bytes);
        base = smem + kScratch;
        __syncthreads();
    }
    const IndexPtrs<StartT> ix = make_ptrs<StartT>(base, mv);
    ListPtrs                lp = {};
    if (SWEEP == 2) { // the halo lists live in LDS; the cell index above stays in HBM/L2 for the undecided few
        fill_lds(smem + kScratch, mv.lblob, mv.lblob_bytes);
        lp = make_list_ptrs(smem + kScratch, mv);
        __syncthreads();
    }
    FitState     fs = load_fit(fa, s, iter_begin);
    Team<kBlock> tm = {(int)threadIdx.x, nullptr, 0u};
    fit_iterations<G, StartT, MODE, SWEEP, kBlock, TILE>(mv, fa, ix, lp, smem, s, off, n, nga, fa.phase, fs, tm);
    if (threadIdx.x == 0) store_fit(fa, s, fs, fa.phase);
}

// Both forms in one launch (the default for batches, point-to-point, index and lists each fitting LDS): the ring
// search for the first fa.switch_iter iterations, then -- without a barrier over the batch -- the workgroup swaps
// the halo lists into LDS over the cell index and carries on in list form; a scan the lists cannot take (see
// fit_iterations) stays in ring form to the end.  StartT is the cell index's type in HBM (list form).
template <typename StartT, int MODE>
__global__ __launch_bounds__(kBlock) void icp_fit_fused_kernel(ModelView mv, FitArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr unsigned kScratch = kScratchBytes;
    const int s = blockIdx.x;
    if (fa.only && fa.only[s] == 0) return; // uniform for the workgroup
    const int off = fa.scan_off[s];
    const int n = fa.scan_off[s + 1] - off;
    const int nga = fa.scan_nga[s];
    fill_lds(smem + kScratch, mv.blob, mv.blob_bytes);
    __syncthreads();
    FitState     fs = load_fit(fa, s, 0);
    Team<kBlock> tm = {(int)threadIdx.x, nullptr, 0u};
    {
        const IndexPtrs<uint16_t> ix = make_ptrs<uint16_t>(smem + kScratch, mv); // an index in LDS has 16-bit starts
        const ListPtrs            none = {};
        fit_iterations<2, uint16_t, MODE, 0>(mv, fa, ix, none, smem, s, off, n, nga, 1, fs, tm);
    }
    if (fs.hand_over) { // uniform for the workgroup
        __syncthreads();
        fill_lds(smem + kScratch, mv.lblob, mv.lblob_bytes);
        const ListPtrs          lp = make_list_ptrs(smem + kScratch, mv);
        const IndexPtrs<StartT> ix = make_ptrs<StartT>(mv.blob, mv);
        __syncthreads();
        fit_iterations<1, StartT, MODE, 2>(mv, fa, ix, lp, smem, s, off, n, nga, 2, fs, tm);
    }
    if (threadIdx.x == 0) store_fit(fa, s, fs, 0);
}

// TWO scans per workgroup, one LDS index: each half of the workgroup (a team of 8 wavefronts, Team<512>) runs the fused
// schedule of icp_fit_fused_kernel on a scan of its own.  A scan spends a third (ring form) to a half (list form) of an
// iteration in its serial part -- the slowest wavefront of the pass, the sums across wavefronts, the solve by one
// wavefront -- with most of its wavefronts asleep at a barrier; here the other scan's wavefronts issue in those slots.
// The teams meet only where the LDS changes hands: both must be through with the cell index before the halo lists
// overwrite it.  GR = lanes per scene point in the ring form.
template <typename StartT, int MODE, int GR>
__global__ __launch_bounds__(kBlock) void icp_fit_pair_kernel(ModelView mv, FitArgs fa, int n_scans)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int      TB = kBlock / 2;
    constexpr unsigned kTeamScratch = TeamDims<TB>::kScratch, kScratch = 2 * kTeamScratch;
    __shared__ int     any_lists;
    // (the team is the same for every lane of a wavefront; telling the compiler so -- readfirstlane -- moves the scan's
    // descriptors and loop bounds into scalar registers, frees twenty VGPRs and ends the 28-byte scratch spill, and is
    // SLOWER: 138 scalar registers then spill instead, in the loops; 0.611 -> 0.636 ms for 512 scans)
    const int          team = (int)threadIdx.x / TB;
    unsigned char     *tsm = smem + team * kTeamScratch;
    Team<TB>           tm = {(int)threadIdx.x % TB, reinterpret_cast<unsigned *>(tsm + kTeamScratch - 16), 0u};
    if (tm.tid == 0) *tm.bar = 0u;
    if (threadIdx.x == 0) any_lists = 0;
    const int  s = 2 * (int)blockIdx.x + team;
    const bool active = s < n_scans && !(fa.only && fa.only[s] == 0); // an odd batch leaves the last workgroup one idle team
    const int  off = active ? fa.scan_off[s] : 0;
    const int  n = active ? fa.scan_off[s + 1] - off : 0;
    const int  nga = active ? fa.scan_nga[s] : 0;
    fill_lds(smem + kScratch, mv.blob, mv.blob_bytes);
    __syncthreads();
    FitState fs = load_fit(fa, active ? s : 0, 0);
    if (active) {
        const IndexPtrs<uint16_t> ix = make_ptrs<uint16_t>(smem + kScratch, mv); // an index in LDS has 16-bit starts
        const ListPtrs            none = {};
        fit_iterations<GR, uint16_t, MODE, 0, TB>(mv, fa, ix, none, tsm, s, off, n, nga, 1, fs, tm);
    }
    const bool lists = active && fs.hand_over; // uniform for the team
    if (lists && tm.tid == 0) any_lists = 1;
    __syncthreads(); // nobody reads the cell index in LDS any more
    if (any_lists) {
        fill_lds(smem + kScratch, mv.lblob, mv.lblob_bytes);
        __syncthreads();
    }
    if (lists) {
        const ListPtrs          lp = make_list_ptrs(smem + kScratch, mv);
        const IndexPtrs<StartT> ix = make_ptrs<StartT>(mv.blob, mv);
        fit_iterations<1, StartT, MODE, 2, TB>(mv, fa, ix, lp, tsm, s, off, n, nga, 2, fs, tm);
    }
    if (active && tm.tid == 0) store_fit(fa, s, fs, 0);
}

// IcpPointToPoint::getEdgeWeight, icpPointToPoint.cpp:233-316 (with dy = ax - bx
// of :262), over the correspondences of the last executed fitStep, i.e. those
// found from `pose` = R,t as they stood when that step began.  One workgroup.
template <typename StartT>
__global__ __launch_bounds__(kBlock) void icp_edge_weight_kernel(ModelView mv, const double2 *pts, int n, int nga,
                                                                 const double *pose, double indist, double *eW)
{
    __shared__ double red[kWaves][8];
    __shared__ double D[3], tot[8];
    const IndexPtrs<StartT> ix = make_ptrs<StartT>(mv.blob, mv);
    const int  tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Pose T = {pose[0], pose[1], pose[2], pose[3], pose[4], pose[5]};

    auto block_sum = [&](double v[], int cnt) {
        for (int k = 0; k < cnt; ++k) {
            const double w = wave_sum(v[k]);
            if (lane == 0) red[wave][k] = w;
        }
        __syncthreads();
        if (tid < cnt) {
            double s = 0.0;
            for (int w = 0; w < kWaves; ++w) s += red[w][tid];
            tot[tid] = s;
        }
        __syncthreads();
    };

    for (int phase = 0; phase < 2; ++phase) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = tid; p < n; p += kBlock) {
            const double2 P = pts[p];
            const float qx = (float)__dadd_rn(__dadd_rn(__dmul_rn(T.r00, P.x), __dmul_rn(T.r01, P.y)), T.t0);
            const float qy = (float)__dadd_rn(__dadd_rn(__dmul_rn(T.r10, P.x), __dmul_rn(T.r11, P.y)), T.t1);
            const int   cls = p < nga ? 0 : 1;
            if (mv.n_cls[cls] <= 3) continue;
            const Best b = nn_search<1, StartT>(ix, mv, cls, qx, qy, 0, indist);
            if (b.pos < 0 || !((double)b.d < indist)) continue;
            const float2 m = ix.pts[mv.base[cls] + b.pos];
            const double ax = (double)m.x, ay = (double)m.y, bx = (double)qx, by = (double)qy;
            const double x = (ax + bx) / 2.0, y = (ay + by) / 2.0;
            if (phase == 0) {
                const double dx = ax - bx, dy = ax - bx; // :261-262, as written there
                acc[0] += 1.0;
                acc[1] += x;
                acc[2] += y;
                acc[3] += x * x + y * y;
                acc[4] += dx;
                acc[5] += dy;
                acc[6] += -y * dx + x * dy;
            } else {
                const double tx = (ax - bx - D[0] + y * D[2]);
                const double ty = (ay - by - D[1] - x * D[2]);
                acc[0] += tx * tx + ty * ty;
            }
        }
        block_sum(acc, phase == 0 ? 7 : 1);
        if (phase == 0) {
            if (tid == 0) {
                // D = inv(MM) * MZ through the Gauss-Jordan solve (Matrix::inv -> solve, matrix.cpp:393)
                double A[9] = {tot[0], 0, -tot[2], 0, tot[0], tot[1], -tot[2], tot[1], tot[3]};
                double b[3] = {tot[4], tot[5], tot[6]};
                solve3(A, b);
                D[0] = b[0];
                D[1] = b[1];
                D[2] = b[2];
                eW[0] = tot[0]; // keep MM until ss is known
                eW[1] = tot[1];
                eW[2] = tot[2];
                eW[3] = tot[3];
            }
            __syncthreads();
        } else if (tid == 0) {
            const double nc = eW[0], sx = eW[1], sy = eW[2], xpy = eW[3];
            const double ss = tot[0] / (2 * nc - 3);
            const double sconst = 1.0 / ss;
            const double MM[9] = {nc, 0, -sy, 0, nc, sx, -sy, sx, xpy};
            for (int k = 0; k < 9; ++k) eW[k] = MM[k] * sconst;
        }
    }
}

// KDTree::n_nearest(q, 1): G lanes per query, index read from HBM/L2.
template <int G, typename StartT>
__global__ __launch_bounds__(256) void icp_nearest_kernel(ModelView mv, int cls, const float2 *q, int n,
                                                          float *dis, int *idx)
{
    const IndexPtrs<StartT> ix = make_ptrs<StartT>(mv.blob, mv);
    const int gid = (blockIdx.x * 256 + threadIdx.x) / G;
    const int sub = threadIdx.x % G;
    if (gid >= n) return;
    const float2 qq = q[gid];
    const Best   b = nn_search<G, StartT>(ix, mv, cls, qq.x, qq.y, sub, INFINITY);
    if (sub == 0) {
        dis[gid] = b.pos >= 0 ? b.d : 1.0e38f; // kdtree.cpp:325 "infinity"
        idx[gid] = b.pos >= 0 ? (int)b.oidx : -1;
    }
}

// ---------------------------------------------------------------- host side

} // namespace

namespace {

template <int G, bool LDS, typename StartT, int MODE, int SWEEP = 0>
int launch_fit_t(slam_icp *h, const FitArgs &fa, int n_scans, hipStream_t st)
{
    auto kern = icp_fit_kernel<G, LDS, StartT, MODE, SWEEP>;
    const size_t lds = h->lds_bytes;
    if (lds > 48 * 1024)
        SLAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(n_scans), dim3(kBlock), lds, st, h->mv, fa);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

template <int MODE>
int launch_fit_sweep(slam_icp *h, const FitArgs &fa, int n_scans, hipStream_t st)
{
    if (h->sweep == 2 && h->have_lists) { // lists in LDS, cell index from HBM/L2
        auto         kern = h->start32 ? icp_fit_kernel<1, false, uint32_t, MODE, 2> : icp_fit_kernel<1, false, uint16_t, MODE, 2>;
        const size_t lds = h->list_lds_bytes;
        SLAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(n_scans), dim3(kBlock), lds, st, h->mv, fa);
        SLAM_HIP(hipGetLastError());
        return SLAM_OK;
    }
    set_error("the list-sweep kernel needs halo lists, and they did not fit LDS for this model");
    return SLAM_E_UNSUPPORTED;
}

// the ring form on an index in HBM/L2 with the wavefronts' model tiles in the otherwise empty LDS (two lanes per point: a
// wavefront's pass is 32 adjacent beams, whose 3 x 3 blocks one tile holds)
template <typename StartT, int MODE>
int launch_fit_tiled(slam_icp *h, const FitArgs &fa, int n_scans, hipStream_t st)
{
    auto         kern = icp_fit_kernel<2, false, StartT, MODE, 0, true>;
    const size_t lds = kScratchBytes + (size_t)kWaves * kWaveTileBytes;
    static_assert(kScratchBytes + kWaves * kWaveTileBytes <= kLdsTotal, "sixteen wave tiles beside the scratch");
    SLAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(n_scans), dim3(kBlock), lds, st, h->mv, fa);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

template <int G, int MODE>
int launch_fit_g(slam_icp *h, const FitArgs &fa, int n_scans, hipStream_t st)
{
    if (h->in_lds) return launch_fit_t<G, true, uint16_t, MODE>(h, fa, n_scans, st);
    if constexpr (G == 2) {
        if (h->prm.wave_tiles > 0 && fa.phase == 0)
            return h->start32 ? launch_fit_tiled<uint32_t, MODE>(h, fa, n_scans, st) : launch_fit_tiled<uint16_t, MODE>(h, fa, n_scans, st);
    }
    if (h->start32) return launch_fit_t<G, false, uint32_t, MODE>(h, fa, n_scans, st);
    return launch_fit_t<G, false, uint16_t, MODE>(h, fa, n_scans, st);
}

template <int MODE>
int launch_fit_m(slam_icp *h, const FitArgs &fa, int n_scans, hipStream_t st)
{
    // point-to-line (not what the reference builds) comes in the default width only; the other widths
    // exist for the point-to-point measurements of DESIGN.md 4.1
    if constexpr (MODE == SLAM_ICP_P2L) {
        return launch_fit_g<2, MODE>(h, fa, n_scans, st);
    } else {
        if (h->sweep) return launch_fit_sweep<SLAM_ICP_P2P>(h, fa, n_scans, st);
        switch (h->G) {
        case 0: return launch_fit_g<0, MODE>(h, fa, n_scans, st);
        case 1: return launch_fit_g<1, MODE>(h, fa, n_scans, st);
        case 2: return launch_fit_g<2, MODE>(h, fa, n_scans, st);
        case 4: return launch_fit_g<4, MODE>(h, fa, n_scans, st);
        case 8: return launch_fit_g<8, MODE>(h, fa, n_scans, st);
        case 16: return launch_fit_g<16, MODE>(h, fa, n_scans, st);
        case 32: return launch_fit_g<32, MODE>(h, fa, n_scans, st);
        case 64: return launch_fit_g<64, MODE>(h, fa, n_scans, st);
        }
    }
    set_error("lanes_per_point must be 0 (per-pass choice) or one of 1,2,4,8,16,32,64 (got %d)", h->G);
    return SLAM_E_INVALID;
}

// lanes per scene point in the ring form of a pair: one is a measurement of the point-to-point kernel;
// point-to-line pairs come with two
template <typename StartT, int MODE>
auto pair_kernel(int lanes) -> void (*)(ModelView, FitArgs, int)
{
    if constexpr (MODE == SLAM_ICP_P2P) {
        if (lanes == 1) return icp_fit_pair_kernel<StartT, MODE, 1>;
    }
    return icp_fit_pair_kernel<StartT, MODE, 2>;
}

// The default schedule of a batch, either solver: first iterations by the ring search (index in LDS), the rest by the
// list sweeps (halo lists in LDS), one launch -- every workgroup swaps its LDS contents when its own scan gets there.
template <int MODE>
int launch_fit_fused(slam_icp *h, FitArgs &fa, int n_scans, hipStream_t st)
{
    fa.switch_iter = h->switch_iter;
    // two scans per workgroup where the batch leaves CUs to spare for it (slam_icp_params::pair_scans)
    const size_t pair_lds = std::max(h->lds_bytes, h->list_lds_bytes) - kScratchBytes + 2 * TeamDims<kBlock / 2>::kScratch;
    // Library default: from two scans per CU on.  Measured on config 2's scans (tools/pair_time.py): 256 scans on 256 CUs
    // 0.39 ms alone against 0.61 in pairs (half the CUs idle); 512 scans 0.72 against 0.61; 1024 scans 1.31 against 1.16.
    const int pair = h->pair > 0 ? h->pair : (h->pair == 0 && n_scans >= 2 * h->n_cu ? 2 : 0);
    if (pair && pair_lds + 64 <= kLdsTotal && n_scans >= 2) { // (+ the kernel's one static word, with its alignment)
        auto kern = h->start32 ? pair_kernel<uint32_t, MODE>(pair) : pair_kernel<uint16_t, MODE>(pair);
        SLAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_lds));
        hipLaunchKernelGGL(kern, dim3((n_scans + 1) / 2), dim3(kBlock), pair_lds, st, h->mv, fa, n_scans);
        SLAM_HIP(hipGetLastError());
        return SLAM_OK;
    }
    auto         kern = h->start32 ? icp_fit_fused_kernel<uint32_t, MODE> : icp_fit_fused_kernel<uint16_t, MODE>;
    const size_t lds = std::max(h->lds_bytes, h->list_lds_bytes);
    SLAM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(n_scans), dim3(kBlock), lds, st, h->mv, fa);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int launch_fit(slam_icp *h, const FitArgs &fa_in, int n_scans, hipStream_t st)
{
    if (n_scans <= 0) return SLAM_OK;
    FitArgs fa = fa_in;
    fa.state = nullptr;
    fa.phase = 0;
    fa.switch_iter = 0;
    fa.far_div = h->far_div;
    if (h->prm.mode == SLAM_ICP_P2L) {
        SLAM_REQUIRE(h->mv.normals, SLAM_E_INVALID, "point-to-line mode needs model normals");
        // the same schedule as point-to-point (the nine sums differ, nothing else); a model whose index or lists do not fit
        // LDS runs the ring search for every iteration
        if (h->two_phase && h->have_lists && h->in_lds && h->mv.lnormals) return launch_fit_fused<SLAM_ICP_P2L>(h, fa, n_scans, st);
        return launch_fit_m<SLAM_ICP_P2L>(h, fa, n_scans, st);
    }
    if (h->two_phase && h->have_lists && h->in_lds && !h->phase_events && !h->split_launch)
        return launch_fit_fused<SLAM_ICP_P2P>(h, fa, n_scans, st);
    if (h->two_phase && h->have_lists) {
        // the same schedule as two launches (SLAM_ICP_SPLIT=1, or while the phases are being timed): the second
        // launch starts when the slowest scan of the first has handed over
        SLAM_TRY(h->w_state.reserve(sizeof(int) * (size_t)n_scans));
        fa.state = static_cast<int *>(h->w_state.p);
        fa.switch_iter = h->switch_iter;
        fa.phase = 1;
        const int keep = h->sweep;
        if (h->phase_events) {
            if (h->ev_pending) { // fold the previous call's times in before its events are reused
                float a = 0, b = 0;
                SLAM_HIP(hipEventSynchronize(h->ev[2]));
                SLAM_HIP(hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
                SLAM_HIP(hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
                h->phase_ms[0] += a;
                h->phase_ms[1] += b;
                ++h->phase_calls;
                h->ev_pending = false;
            }
            for (auto &e : h->ev)
                if (!e) SLAM_HIP(hipEventCreate(&e));
            SLAM_HIP(hipEventRecord(h->ev[0], st));
        }
        h->sweep = 0;
        int rc = launch_fit_m<SLAM_ICP_P2P>(h, fa, n_scans, st);
        if (h->phase_events) SLAM_HIP(hipEventRecord(h->ev[1], st));
        h->sweep = 2;
        fa.phase = 2;
        fa.R0 = fa.R; // the second launch carries on from what the first left
        fa.t0 = fa.t;
        if (rc == SLAM_OK) rc = launch_fit_m<SLAM_ICP_P2P>(h, fa, n_scans, st);
        if (h->phase_events) {
            SLAM_HIP(hipEventRecord(h->ev[2], st));
            h->ev_pending = true;
        }
        h->sweep = keep;
        return rc;
    }
    return launch_fit_m<SLAM_ICP_P2P>(h, fa, n_scans, st);
}

} // namespace

namespace slam {
namespace icp {
// the caller knows that no enqueued work uses the handle any more (the mapper waits on an event instead of the device)
void destroy_unsynchronised(slam_icp *icp)
{
    if (!icp) return;
    release_index(icp);
    for (auto &e : icp->ev)
        if (e) (void)hipEventDestroy(e);
    if (icp->d_normals) pool_free(icp->d_normals);
    if (icp->d_lnormals) pool_free(icp->d_lnormals);
    for (DevBuf *b : {&icp->w_pts, &icp->w_stamps, &icp->w_ew, &icp->w_state, &icp->w_single}) b->release();
    delete icp;
}
} // namespace icp
} // namespace slam

namespace slam {
namespace icp {

// Batches of so few scans run in the spread form (icp_single.hip): one persistent launch whose workgroups of a scan wait
// for one another, with scratch that belongs to the handle -- a handle takes ONE such call at a time (batches in the
// workgroup-per-scan forms may be in flight on several streams at once: they keep nothing in the handle).
bool takes_spread_form(const slam_icp *h, int n_scans)
{
    const int spread_max = h->prm.spread_scans > 0 ? std::min(h->prm.spread_scans, h->n_cu)
                                                   : (h->prm.spread_scans < 0 ? 0 : h->n_cu / kSpreadMinParts);
    return n_scans >= 1 && n_scans <= spread_max && h->prm.lanes_per_point == 0 && !h->skip_spread;
}

} // namespace icp
} // namespace slam

extern "C" {

void slam_icp_default_params(slam_icp_params *p)
{
    if (!p) return;
    p->max_iter = 20;      // icp.cpp:27
    p->min_delta = 1e-6;   // icp.cpp:27
    p->mode = SLAM_ICP_P2P;
    p->normals_k = 10;
    p->lanes_per_point = 0;
    p->cell_size = 0.0;
    p->force_global = 0;
    p->build_on_host = 0;
    p->first_iterations = 0;
    p->far_div = 0;
    p->split_launch = 0;
    p->spread_scans = 0;
    p->pair_scans = 0;
    p->spread_wait_us = 0;
    p->wave_tiles = 0;
    p->spread_tile = 0;
    p->list_min_halo = 0.0;
}

} // extern "C"

namespace {

// the handle of Icp::Icp (icp.cpp:26-70) before its index exists
int icp_new(const slam_icp_params *params, slam_icp **out)
{
    *out = nullptr;
    SLAM_TRY(require_device());
    slam_icp *h = new (std::nothrow) slam_icp();
    SLAM_REQUIRE(h, SLAM_E_NOMEM, "slam_icp_create: out of host memory");
    if (params)
        h->prm = *params;
    else
        slam_icp_default_params(&h->prm);
    // 0 = library default (ring search with 2 lanes per point, then list sweeps);
    // N > 0 = ring search with N lanes per point; -1 = ring search, lanes chosen per pass
    h->sweep = h->prm.lanes_per_point == -2 ? 2 : 0;
    h->G = h->prm.lanes_per_point > 0 ? h->prm.lanes_per_point : (h->prm.lanes_per_point == -1 ? 0 : 2);
    h->two_phase = h->prm.lanes_per_point == 0; // either solver (the point-to-line lists carry a normal per entry)
    if (h->prm.first_iterations > 0) h->switch_iter = h->prm.first_iterations;
    // Point-to-line steps settle a scan within the lists' certified radius in two or three iterations (no classes, no gate; the
    // hand-over itself still waits for the guard, far_div): round 5, tools/exp/switch_pairs.py and switch_far_p2l.py, config 2's
    // scans, 512 in pairs 0.447 / 0.447 / 0.449 / 0.456 / 0.467 / 0.480 / 0.492 / 0.506 ms for 1 / 2 / 3 / 4 / 6 / 8 / 10 / 12 first
    // iterations (far_div 8 ... 128: the same), one per workgroup 0.306 / 0.309 / 0.313 / 0.320 for 4 / 6 / 8 / 10; point-to-point is
    // flat from 3 to 10 at far_div 16-32 and slower below (tools/exp/switch_far.py): it keeps its 10.
    else if (h->prm.mode == SLAM_ICP_P2L) h->switch_iter = 2;
    if (h->prm.far_div > 0) h->far_div = h->prm.far_div;
    h->split_launch = h->prm.split_launch != 0;
    h->pair = h->prm.pair_scans > 0 ? std::min(h->prm.pair_scans, 2) : (h->prm.pair_scans < 0 ? -1 : 0);
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            h->n_cu = cus;
    }
#ifdef SLAM_MEASURE
    if (const char *e = getenv("SLAM_ICP_SPLIT")) h->split_launch = atoi(e) != 0;
    if (const char *e = getenv("SLAM_ICP_CELL")) h->prm.cell_size = atof(e); // measurements: overrides the pitch
    if (const char *e = getenv("SLAM_ICP_SWITCH_ITER")) h->switch_iter = atoi(e);
    if (const char *e = getenv("SLAM_ICP_FAR_DIV")) h->far_div = std::max(atoi(e), 1);
#endif
    int rc = SLAM_OK;
    if (h->prm.mode == SLAM_ICP_P2L && h->split_launch) {
        // (the point-to-line solver has the one-launch schedule only: a request for the two-launch form is refused, not ignored)
        set_error("split_launch is a point-to-point measurement form: SLAM_ICP_P2L runs both search forms in one launch");
        rc = SLAM_E_UNSUPPORTED;
    }
    if (h->G & (h->G - 1) || h->G > 64) {
        set_error("lanes_per_point must be one of 1,2,4,8,16,32,64 (got %d)", h->G);
        rc = SLAM_E_INVALID;
    }
    if (rc != SLAM_OK) {
        slam_icp_destroy(h);
        return rc;
    }
    *out = h;
    return SLAM_OK;
}

// Icp::Icp, icp.cpp:26-70; the model arrays are host memory unless on_device
int icp_create(const double *m_ga, int n_ga, const double *m_nga, int n_nga, const slam_icp_params *params, bool on_device,
               slam_icp_t **out)
{
    SLAM_REQUIRE(out, SLAM_E_INVALID, "slam_icp_create: null out pointer");
    *out = nullptr;
    SLAM_REQUIRE(n_ga >= 0 && n_nga >= 0 && (m_ga || n_ga == 0) && (m_nga || n_nga == 0),
                 SLAM_E_INVALID, "slam_icp_create: bad model arrays");
    // icp.cpp:38-43 "LIBICP works only with at least 5 model points"
    SLAM_REQUIRE(n_ga + n_nga >= 5, SLAM_E_TOO_FEW_MODEL_POINTS,
                 "LIBICP works only with at least 5 model points (got %d)", n_ga + n_nga);
    slam_icp *h = nullptr;
    SLAM_TRY(icp_new(params, &h));
    int rc;
    // (point-to-line: the build itself merges the classes into one index and makes the normals, icp_build.hip)
    rc = build_index(h, m_ga, n_ga, m_nga, n_nga, on_device);
    if (rc != SLAM_OK) {
        slam_icp_destroy(h);
        return rc;
    }
    *out = h;
    return SLAM_OK;
}

} // namespace

namespace slam {
namespace icp {

int create_begin(const double *d_ga, int cap_ga, const double *d_nga, int cap_nga, const int *d_cnt, const slam_icp_params *params,
                 hipStream_t st, slam_icp **out, bool beside)
{
    SLAM_REQUIRE(out && cap_ga >= 0 && cap_nga >= 0 && (d_ga || cap_ga == 0) && (d_nga || cap_nga == 0), SLAM_E_INVALID,
                 "create_begin: bad model arrays");
    slam_icp *h = nullptr;
    SLAM_TRY(icp_new(params, &h));
    h->build_beside = beside;
    const int rc = build_index_begin(h, d_ga, cap_ga, d_nga, cap_nga, d_cnt, true, st);
    if (rc != SLAM_OK) {
        slam_icp_destroy(h);
        return rc;
    }
    *out = h;
    return SLAM_OK;
}

bool create_ready(slam_icp *h) { return build_index_ready(h); }

int create_finish(slam_icp *h)
{
    const int rc = build_index_finish(h);
    if (rc != SLAM_OK) slam_icp_destroy(h);
    return rc;
}

} // namespace icp
} // namespace slam

extern "C" {

int slam_icp_create(const double *m_ga, int n_ga, const double *m_nga, int n_nga,
                    const slam_icp_params *params, slam_icp_t **out)
{
    return icp_create(m_ga, n_ga, m_nga, n_nga, params, false, out);
}

int slam_icp_create_dev(const double *d_m_ga, int n_ga, const double *d_m_nga, int n_nga,
                        const slam_icp_params *params, slam_icp_t **out)
{
    return icp_create(d_m_ga, n_ga, d_m_nga, n_nga, params, true, out);
}

void slam_icp_destroy(slam_icp_t *icp)
{
    if (!icp) return;
    // the buffers go back to the library's pool, not to hipFree (which would wait for the device itself)
    (void)hipDeviceSynchronize();
    slam::icp::destroy_unsynchronised(icp);
}

int slam_icp_build_info(slam_icp_t *icp, int *on_device, double ms[4])
{
    SLAM_REQUIRE(icp, SLAM_E_INVALID, "null handle");
    if (on_device) *on_device = icp->built_on_device ? 1 : 0;
    if (ms)
        for (int k = 0; k < 4; ++k) ms[k] = icp->build_ms[k];
    return SLAM_OK;
}

int slam_icp_index_blob(slam_icp_t *icp, int which, void *buf, size_t cap, size_t *bytes)
{
    SLAM_REQUIRE(icp && (which == 0 || which == 1), SLAM_E_INVALID, "slam_icp_index_blob: bad arguments");
    const void  *src = which == 0 ? icp->d_blob : (icp->have_lists ? icp->d_lblob : nullptr);
    const size_t n = which == 0 ? icp->mv.blob_bytes : (icp->have_lists ? icp->mv.lblob_bytes : 0);
    if (bytes) *bytes = n;
    if (!buf || n == 0) return SLAM_OK;
    SLAM_REQUIRE(cap >= n, SLAM_E_INVALID, "slam_icp_index_blob: buffer of %zu bytes, blob has %zu", cap, n);
    SLAM_HIP(hipMemcpy(buf, src, n, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

int slam_icp_set_max_iterations(slam_icp_t *icp, int val)
{
    SLAM_REQUIRE(icp, SLAM_E_INVALID, "null handle");
    icp->prm.max_iter = val;
    return SLAM_OK;
}

int slam_icp_set_min_delta(slam_icp_t *icp, double val)
{
    SLAM_REQUIRE(icp, SLAM_E_INVALID, "null handle");
    icp->prm.min_delta = val;
    return SLAM_OK;
}

int slam_icp_set_subsampling_step(slam_icp_t *icp, int val)
{
    SLAM_REQUIRE(icp, SLAM_E_INVALID, "null handle");
    icp->sub_step = val;
    return SLAM_OK;
}

int slam_icp_fit_batch_dev(slam_icp_t *icp, const double *d_pts, const int32_t *d_scan_off,
                           const int32_t *d_scan_nga, int n_scans, double *d_R, double *d_t,
                           double indist, slam_icp_result *d_result, double *d_trace,
                           slam_stream_t stream)
{
    return slam_icp_fit_batch_from_dev(icp, d_pts, d_scan_off, d_scan_nga, n_scans, d_R, d_t, d_R, d_t, indist, d_result, d_trace, stream);
}

int slam_icp_fit_batch_from_dev(slam_icp_t *icp, const double *d_pts, const int32_t *d_scan_off,
                                const int32_t *d_scan_nga, int n_scans, const double *d_R0, const double *d_t0,
                                double *d_R, double *d_t, double indist, slam_icp_result *d_result, double *d_trace,
                                slam_stream_t stream)
{
    SLAM_REQUIRE(icp && d_scan_off && d_scan_nga && d_R0 && d_t0 && d_R && d_t && n_scans >= 0, SLAM_E_INVALID,
                 "slam_icp_fit_batch_dev: bad arguments");
    SLAM_TRY(require_device());
    FitArgs fa;
    fa.pts = reinterpret_cast<const double2 *>(d_pts);
    fa.scan_off = d_scan_off;
    fa.scan_nga = d_scan_nga;
    fa.R = d_R;
    fa.t = d_t;
    fa.R0 = d_R0;
    fa.t0 = d_t0;
    fa.result = d_result;
    fa.trace = d_trace;
    fa.max_iter = icp->prm.max_iter;
    fa.min_delta = icp->prm.min_delta;
    fa.indist = indist;
    fa.step_pose = icp->want_step_pose ? reinterpret_cast<double *>(static_cast<unsigned char *>(icp->w_pts.p) + icp->step_pose_off) : nullptr;
    fa.stamps = nullptr;
#ifdef SLAM_MEASURE
    if (getenv("SLAM_ICP_STAMPS")) {
        SLAM_TRY(icp->w_stamps.reserve((size_t)n_scans * kWaves * kStampSlots * sizeof(long long)));
        SLAM_HIP(hipMemsetAsync(icp->w_stamps.p, 0, (size_t)n_scans * kWaves * kStampSlots * sizeof(long long), as_stream(stream)));
        fa.stamps = static_cast<long long *>(icp->w_stamps.p);
        icp->n_stamps = n_scans * kWaves;
    }
#endif
    // few scans (one, in the reference's own usage): each scan spread over many workgroups of one persistent launch
    fa.only = nullptr;
    fa.spread_tag = 0;
    fa.redo_mirror = icp->redo_mirror;
    if (takes_spread_form(icp, n_scans) && !fa.stamps) {
        // ... and behind it the workgroup-per-scan form for the scans whose workgroups did not all become resident together
        // (another spread launch or a persistent kernel holding CUs: icp_single.hip): its workgroups find their scan's flag
        // clear and exit -- a few microseconds -- unless the scan has to be redone.  A fit always completes (icp.cpp:80-114).
        SLAM_TRY(launch_fit_spread(icp, fa, n_scans, as_stream(stream), &fa.only));
        return launch_fit(icp, fa, n_scans, as_stream(stream));
    }
    return launch_fit(icp, fa, n_scans, as_stream(stream));
}

int slam_icp_fit(slam_icp_t *icp, const double *t_ga, int n_tga, const double *t_nga, int n_tnga,
                 double R[4], double t[2], double indist, slam_icp_result *result)
{
    SLAM_REQUIRE(icp && R && t && n_tga >= 0 && n_tnga >= 0, SLAM_E_INVALID,
                 "slam_icp_fit: bad arguments");
    SLAM_REQUIRE((t_ga || n_tga == 0) && (t_nga || n_tnga == 0), SLAM_E_INVALID,
                 "slam_icp_fit: null template array");
    if (result) {
        result->iters = 0;
        result->n_corr = 0;
        result->delta = 0.0;
    }
    // icp.cpp:100-103 "ERROR: Total template has N points" -> return, R,t untouched
    SLAM_REQUIRE(n_tga + n_tnga >= 5, SLAM_E_TOO_FEW_SCENE_POINTS, "Total template has %d points",
                 n_tga + n_tnga);
    SLAM_TRY(require_device());
    const int n = n_tga + n_tnga;
    // One block in HBM and one pinned block on the host, the same layout: the template's points, then a header
    //   [scan_off 0, n | nGA | pad] [R t] [result] [pose of the last executed step]
    // so that a fit is one copy in and two launches (the reference's fit() is synchronous too).  What comes back -- pose, result,
    // whether the spread form had to hand the scan to the one-workgroup form -- the kernels write into the pinned block themselves
    // (device-visible host memory, a handful of posted stores at the end of a fit): no copy command behind the fit.
    const size_t pts_bytes = 16 * (size_t)n, hdr = pts_bytes, o_pose_in = hdr + 16, o_res = o_pose_in + 48, o_step = o_res + 16,
                 o_redo = o_step + 48, o_pose_out = o_redo + 16, o_res_out = o_pose_out + 48, total = o_res_out + 16;
    SLAM_TRY(icp->w_pts.reserve(o_redo));
    unsigned char *hp = static_cast<unsigned char *>(pinned_scratch(total));
    SLAM_REQUIRE(hp, SLAM_E_NOMEM, "slam_icp_fit: no pinned staging memory");
    unsigned char *dp = static_cast<unsigned char *>(icp->w_pts.p);
    if (n_tga) memcpy(hp, t_ga, 16 * (size_t)n_tga);
    if (n_tnga) memcpy(hp + 16 * (size_t)n_tga, t_nga, 16 * (size_t)n_tnga);
    int32_t *hh = reinterpret_cast<int32_t *>(hp + hdr);
    hh[0] = 0, hh[1] = n, hh[2] = n_tga, hh[3] = 0;
    memcpy(hp + o_pose_in, R, 32);
    memcpy(hp + o_pose_in + 32, t, 16);
    memcpy(hp + o_pose_out, hp + o_pose_in, 48); // (a fit without a correspondence leaves R, t as they were: icp.cpp:120)
    slam_icp_result *h_res = reinterpret_cast<slam_icp_result *>(hp + o_res_out);
    memset(h_res, 0, sizeof *h_res);
    h_res->iters = -2; // no kernel leaves this
    int *h_redo = reinterpret_cast<int *>(hp + o_redo);
    *h_redo = 0;
    hipStream_t st = nullptr;
    SLAM_HIP(hipMemcpyAsync(dp, hp, o_res, hipMemcpyHostToDevice, st));
    icp->want_step_pose = true;
    icp->step_pose_off = o_step;
    icp->spread_points_hint = n;
    icp->skip_spread = icp->spread_backoff > 0; // the spread form lost its CUs a moment ago: not again right away
    if (icp->skip_spread) --icp->spread_backoff;
    icp->redo_mirror = h_redo;
    const int rc_fit = slam_icp_fit_batch_from_dev(icp, reinterpret_cast<double *>(dp), reinterpret_cast<int32_t *>(dp + hdr),
                                                   reinterpret_cast<int32_t *>(dp + hdr + 8), 1,
                                                   reinterpret_cast<double *>(dp + o_pose_in), reinterpret_cast<double *>(dp + o_pose_in + 32),
                                                   reinterpret_cast<double *>(hp + o_pose_out), reinterpret_cast<double *>(hp + o_pose_out + 32),
                                                   indist, h_res, nullptr, st);
    icp->want_step_pose = false;
    icp->skip_spread = false;
    icp->redo_mirror = nullptr;
    SLAM_TRY(rc_fit);
    icp->last_n = n;
    icp->last_nga = n_tga;
    icp->last_indist = indist;
    icp->have_last = true;
    SLAM_HIP(hipStreamSynchronize(st));
    slam_icp_result res;
    memcpy(&res, h_res, sizeof res);
    if (*h_redo != 0) icp->spread_backoff = 16; // redone by the one-workgroup form: 5 ms late
    // (a scan whose spread-form workgroups did not become resident together was redone by the one-workgroup form inside the
    // call above: there is no outcome without a pose)
    SLAM_REQUIRE(res.iters >= 0, SLAM_E_HIP, "slam_icp_fit: the registration kernels left no result (iters = %d)", res.iters);
    memcpy(R, hp + o_pose_out, 32);
    memcpy(t, hp + o_pose_out + 32, 16);
    if (result) *result = res;
    return SLAM_OK;
}

int slam_icp_nearest_dev(slam_icp_t *icp, int cls, const float *d_query_xy, int n, float *d_dis,
                         int32_t *d_idx, slam_stream_t stream)
{
    SLAM_REQUIRE(icp && (cls == 0 || cls == 1) && n >= 0 && d_dis && d_idx, SLAM_E_INVALID,
                 "slam_icp_nearest_dev: bad arguments");
    SLAM_TRY(require_device());
    if (n == 0) return SLAM_OK;
    constexpr int G = 8;
    const int     blocks = (int)(((size_t)n * G + 255) / 256);
    const float2 *q = reinterpret_cast<const float2 *>(d_query_xy);
    if (icp->start32)
        hipLaunchKernelGGL((icp_nearest_kernel<G, uint32_t>), dim3(blocks), dim3(256), 0, as_stream(stream),
                           icp->mv, cls, q, n, d_dis, d_idx);
    else
        hipLaunchKernelGGL((icp_nearest_kernel<G, uint16_t>), dim3(blocks), dim3(256), 0, as_stream(stream),
                           icp->mv, cls, q, n, d_dis, d_idx);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int slam_icp_get_normals(slam_icp_t *icp, double *normals_xy)
{
    SLAM_REQUIRE(icp && normals_xy, SLAM_E_INVALID, "slam_icp_get_normals: bad arguments");
    SLAM_REQUIRE(icp->d_normals, SLAM_E_INVALID, "normals exist only in SLAM_ICP_P2L mode");
    const size_t n = (size_t)icp->mv.n_cls[0] + icp->mv.n_cls[1];
    SLAM_HIP(hipMemcpy(normals_xy, icp->d_normals, 2 * n * sizeof(double), hipMemcpyDeviceToHost));
    return SLAM_OK;
}

int slam_icp_get_edge_weight(slam_icp_t *icp, double eW[9])
{
    SLAM_REQUIRE(icp && eW, SLAM_E_INVALID, "slam_icp_get_edge_weight: bad arguments");
    SLAM_REQUIRE(icp->have_last, SLAM_E_INVALID, "getEdgeWeight: no slam_icp_fit() call to take correspondences from");
    SLAM_REQUIRE(icp->prm.mode == SLAM_ICP_P2P, SLAM_E_UNSUPPORTED, "getEdgeWeight belongs to IcpPointToPoint");
    SLAM_TRY(require_device());
    SLAM_TRY(icp->w_ew.reserve(9 * sizeof(double)));
    const double2 *pts = static_cast<const double2 *>(icp->w_pts.p);
    const double  *pose = reinterpret_cast<const double *>(static_cast<const unsigned char *>(icp->w_pts.p) + icp->step_pose_off);
    double        *d_ew = static_cast<double *>(icp->w_ew.p);
    if (icp->start32)
        hipLaunchKernelGGL((icp_edge_weight_kernel<uint32_t>), dim3(1), dim3(kBlock), 0, nullptr, icp->mv, pts,
                           icp->last_n, icp->last_nga, pose, icp->last_indist, d_ew);
    else
        hipLaunchKernelGGL((icp_edge_weight_kernel<uint16_t>), dim3(1), dim3(kBlock), 0, nullptr, icp->mv, pts,
                           icp->last_n, icp->last_nga, pose, icp->last_indist, d_ew);
    SLAM_HIP(hipGetLastError());
    SLAM_HIP(hipMemcpy(eW, d_ew, 9 * sizeof(double), hipMemcpyDeviceToHost));
    return SLAM_OK;
}

#ifdef SLAM_MEASURE // declared in include/slam_mi355x_measure.h
// diagnostic (not in the public header): per wavefront, mean cycles in the four phases, sweep
// fall-backs, and the search cycles of iterations 0-3 of the last batch launched with SLAM_ICP_STAMPS=1 in the environment
// diagnostic (not in the public header): with `on`, the default schedule runs as two launches (ring search,
// list sweeps) timed separately (three events per call); out = mean ms of either over the calls since the
// last query
int slam_icp_debug_phase_events(slam_icp_t *icp, int on)
{
    SLAM_REQUIRE(icp, SLAM_E_INVALID, "null handle");
    icp->phase_events = on != 0;
    icp->phase_ms[0] = icp->phase_ms[1] = 0;
    icp->phase_calls = 0;
    icp->ev_pending = false;
    return SLAM_OK;
}

int slam_icp_debug_phase_ms(slam_icp_t *icp, double out[2], int *calls)
{
    SLAM_REQUIRE(icp && out, SLAM_E_INVALID, "null argument");
    if (icp->ev_pending) {
        float a = 0, b = 0;
        SLAM_HIP(hipEventSynchronize(icp->ev[2]));
        SLAM_HIP(hipEventElapsedTime(&a, icp->ev[0], icp->ev[1]));
        SLAM_HIP(hipEventElapsedTime(&b, icp->ev[1], icp->ev[2]));
        icp->phase_ms[0] += a;
        icp->phase_ms[1] += b;
        ++icp->phase_calls;
        icp->ev_pending = false;
    }
    const int n = icp->phase_calls > 0 ? icp->phase_calls : 1;
    out[0] = icp->phase_ms[0] / n;
    out[1] = icp->phase_ms[1] / n;
    if (calls) *calls = icp->phase_calls;
    return SLAM_OK;
}

// diagnostic: the raw stamps, [scan][wavefront][kStampSlots] ticks; returns the number of rows through *rows
int slam_icp_debug_stamps_raw(slam_icp_t *icp, long long *out, int cap_rows, int *rows)
{
    SLAM_REQUIRE(icp && out && rows && icp->n_stamps > 0, SLAM_E_INVALID, "no stamps collected");
    SLAM_HIP(hipDeviceSynchronize());
    const int n = std::min(cap_rows, icp->n_stamps);
    SLAM_HIP(hipMemcpy(out, icp->w_stamps.p, (size_t)n * kStampSlots * sizeof(long long), hipMemcpyDeviceToHost));
    *rows = n;
    return SLAM_OK;
}

int slam_icp_debug_stamps(slam_icp_t *icp, double out[9])
{
    SLAM_REQUIRE(icp && out && icp->n_stamps > 0, SLAM_E_INVALID, "no stamps collected");
    SLAM_HIP(hipDeviceSynchronize());
    std::vector<long long> v((size_t)icp->n_stamps * kStampSlots);
    SLAM_HIP(hipMemcpy(v.data(), icp->w_stamps.p, v.size() * sizeof(long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < kStampSlots; ++k) out[k] = 0;
    for (int i = 0; i < icp->n_stamps; ++i)
        for (int k = 0; k < kStampSlots; ++k) out[k] += (double)v[(size_t)i * kStampSlots + k];
    for (int k = 0; k < kStampSlots; ++k) out[k] /= icp->n_stamps;
    return SLAM_OK;
}

// diagnostic: the spread form's stamps of the last launch made with SLAM_SPREAD_STAMPS=1: [parts][iters][16] ticks of 10 ns
int slam_icp_debug_spread_stamps(slam_icp_t *icp, long long *out, size_t cap, int *parts, int *iters)
{
    SLAM_REQUIRE(icp && out && parts && iters && icp->n_stamps < 0, SLAM_E_INVALID, "no spread stamps collected");
    SLAM_HIP(hipDeviceSynchronize());
    const size_t n = (size_t)(-icp->n_stamps) * (size_t)icp->spread_stamp_iters * 16;
    SLAM_REQUIRE(cap >= n, SLAM_E_INVALID, "buffer too small: %zu stamps", n);
    SLAM_HIP(hipMemcpy(out, icp->w_stamps.p, n * sizeof(long long), hipMemcpyDeviceToHost));
    *parts = -icp->n_stamps;
    *iters = icp->spread_stamp_iters;
    return SLAM_OK;
}

#endif // SLAM_MEASURE

int slam_icp_index_info(slam_icp_t *icp, int *nx, int *ny, double *cell, int *in_lds, size_t *lds_bytes,
                        int *lanes_per_point)
{
    SLAM_REQUIRE(icp, SLAM_E_INVALID, "null handle");
    if (nx) *nx = icp->mv.lat.nx;
    if (ny) *ny = icp->mv.lat.ny;
    if (cell) *cell = icp->mv.lat.h;
    if (in_lds) *in_lds = icp->in_lds ? 1 : 0;
    if (lds_bytes) *lds_bytes = icp->lds_bytes;
    if (lanes_per_point) *lanes_per_point = icp->G;
    return SLAM_OK;
}

int slam_icp_list_info(slam_icp_t *icp, int *two_forms, int *first_iterations, double *pitch, double *halo,
                       double *certified_radius, size_t *list_bytes)
{
    SLAM_REQUIRE(icp, SLAM_E_INVALID, "null handle");
    const bool have = icp->have_lists;
    if (two_forms) *two_forms = (icp->two_phase && have) ? 1 : 0;
    if (first_iterations) *first_iterations = icp->switch_iter;
    if (pitch) *pitch = have ? icp->mv.llat.h : 0.0;
    if (halo) *halo = have ? (double)icp->mv.lpad * icp->mv.llat.h : 0.0;
    if (certified_radius) *certified_radius = have ? std::sqrt((double)icp->mv.cert2) : 0.0;
    if (list_bytes) *list_bytes = have ? icp->mv.lblob_bytes : 0;
    return SLAM_OK;
}

} // extern "C"
