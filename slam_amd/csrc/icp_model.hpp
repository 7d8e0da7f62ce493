// icp_model.hpp -- the model index of the ICP (what slam_icp_create builds and the fit kernels read) and the
// handle behind slam_icp_t.  Shared by icp.hip (batch kernels, C-ABI), icp_build.hip (index build on the
// device) and icp_single.hip (one scan against a large model).
#pragma once
#include <cfloat>
#include <cstdint>

#include "common.hpp"

namespace slam {
namespace icp {

constexpr int kBlock = 1024;          // threads per scan workgroup
constexpr int kWaves = kBlock / 64;
constexpr int kNumAcc = 9;            // doubles reduced per iteration
constexpr int kStampSlots = 9;        // diagnostic stamps per wavefront
constexpr int kHoist = 3;             // passes whose points stay in registers across iterations
constexpr unsigned kLdsTotal = 160u * 1024u;
constexpr int kCoop = 4;                       // lanes per query in the cooperative rounds (8 until round 4: 256 scans in pairs
                                               // 0.562 -> 0.548 ms; 2 lanes: 0.567 -- a scan's tail of 55 points is 14 queries per wavefront-round now)
constexpr int kCoopPerWave = 64 / kCoop;       // queries a wavefront searches at a time
constexpr int kCoopPerBlock = kWaves * kCoopPerWave;
constexpr unsigned kReduceBytes = (2u * kWaves * kNumAcc + 2u * 8u) * sizeof(double); // reduction + broadcast
constexpr unsigned kQueueBytes = 4u * kWaves + 2u * kBlock; // per-wavefront counts + 64 u16 entries per wavefront
static_assert(kHoist == 3, "the pass loop selects Pc0, Pc1, Pc2 explicitly");
static_assert(kWaves == 16 && 16 % kCoop == 0, "drain_queue keeps one wavefront count per lane of a 16-lane DPP row");
constexpr unsigned kScratchBytes = (kReduceBytes + kQueueBytes + 15u) & ~15u;

// A team = the TB threads (TB / 64 wavefronts) of a workgroup that work on ONE scan: the whole workgroup in the batch
// kernels (TB = kBlock), half of it in the pair kernel, where two scans share one LDS index.  What of a workgroup's
// scratch depends on the team's size:
template <int TB>
struct TeamDims {
    static constexpr int      kW = TB / 64;                      // wavefronts
    static constexpr int      kCoopBlock = kW * kCoopPerWave;    // queries of one cooperative round
    static constexpr unsigned kReduce = (2u * kW * kNumAcc + 2u * 8u) * sizeof(double);
    static constexpr unsigned kQueue = 4u * kW + 2u * TB;
    // a team smaller than the workgroup keeps the word of its own barrier in the last 16 bytes
    static constexpr unsigned kScratch = (kReduce + kQueue + (TB < kBlock ? 16u : 0u) + 15u) & ~15u;
};
static_assert(TeamDims<kBlock>::kScratch == kScratchBytes && TeamDims<kBlock>::kCoopBlock == kCoopPerBlock, "one team = the workgroup");

struct Lattice {
    int   nx, ny;
    float x0, y0, h, inv_h;
    float margin; // subtracted from r*h before squaring: absorbs f32 rounding of the cell assignment
};

// Device view of the model index.  All offsets are bytes into `blob`.
struct ModelView {
    const unsigned char *blob;
    unsigned blob_bytes;
    unsigned off_pts;       // float2[n_all]: class 0 (GA) sorted by cell, then class 1 (NGA)
    unsigned off_start[2];  // StartT[ncells+1] per class, positions relative to the class base
    unsigned off_oidx;      // StartT[n_all]: original index within the class
    int      n_cls[2];
    int      base[2];       // first point of each class in pts
    Lattice  lat;
    double   cx, cy;        // shift origin for the running sums (model centroid)
    const double *normals;  // P2L: double2 per ORIGINAL all-index (GA then NGA), or null.  A point-to-line model is ONE class
                            // (icpPointToPlane.cpp:55-77 has no classes): every point lives in class 1 of the index, in the
                            // order GA then NGA, so that a neighbour's `oidx` is its all-index
    const double2 *lnormals; // P2L with halo lists: the normal of every list ENTRY (parallel to the lists' pts; stays in HBM/L2)
    // Halo lists (list-sweep mode): per class and cell of a second, coarser lattice, every point within the
    // cell dilated by `pad` cells, ordered along the axis of larger extent.  A query whose best distance
    // over its own cell's list is below cert2 has seen every point that close: no neighbour cells.
    const unsigned char *lblob;
    unsigned lblob_bytes;
    unsigned loff_pts;      // float2[n_ent[0] + n_ent[1]]
    unsigned loff_start[2]; // u16[lcells+1] per class, positions relative to the class base
    unsigned loff_axis[2];  // two bits per cell and class: the list's ordering key (list_key)
    float    lkeps;         // rounding allowance of a diagonal key difference, metres
    int      lbase[2];      // first entry of each class
    Lattice  llat;
    float    lpad;          // halo in cell units
    float    cert2;         // squared certified radius (metres^2)
};

template <typename StartT>
struct IndexPtrs {
    const float2 *pts;
    const StartT *start[2];
    const StartT *oidx;
};

template <typename StartT>
__device__ inline IndexPtrs<StartT> make_ptrs(const unsigned char *base, const ModelView &mv)
{
    IndexPtrs<StartT> ix;
    ix.pts = reinterpret_cast<const float2 *>(base + mv.off_pts);
    ix.start[0] = reinterpret_cast<const StartT *>(base + mv.off_start[0]);
    ix.start[1] = reinterpret_cast<const StartT *>(base + mv.off_start[1]);
    ix.oidx = reinterpret_cast<const StartT *>(base + mv.off_oidx);
    return ix;
}

// A device buffer of a handle, taken from the library's pool (common.hpp): a handle is made per match
// where the reference makes its matcher per match, so buffers must not cost a hipMalloc each.
struct DevBuf {
    void  *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SLAM_OK;
        if (p) {
            // the old block may still be in use by enqueued work
            SLAM_HIP(hipDeviceSynchronize());
            pool_free(p);
        }
        p = nullptr;
        cap = 0;
        p = pool_alloc(bytes);
        if (!p) return SLAM_E_NOMEM;
        cap = bytes;
        return SLAM_OK;
    }
    void release() // the owner has synchronised with the device
    {
        if (p) pool_free(p);
        p = nullptr;
        cap = 0;
    }
};

// Ordering key of a halo list: 0 = x, 1 = y, 2 = x + y, 3 = x - y (the diagonals serve lists bent around a corner,
// where either axis would put a whole wall on one key).  Host and device evaluate the same float expression.
__host__ __device__ inline float list_key(int dir, float x, float y)
{
    const float ux = dir == 1 ? 0.0f : 1.0f, uy = dir == 0 ? 0.0f : (dir == 3 ? -1.0f : 1.0f);
    return ux * x + uy * y; // products by 0 and +-1 are exact: x, y, fl(x + y), fl(x - y); built without FMA contraction
}

// Lattice coordinate of a model point along one axis: floor(fl(fl(v - v0) * inv_h)) clamped into [0, n-1];
// a non-finite coordinate goes to 0 (such points are in no list and beyond every gate).  Host and device
// builds evaluate this one expression.
__host__ __device__ inline int lattice_coord(float v, float v0, float inv_h, int n)
{
    if (!(v - v <= 0.0f)) return 0; // NaN or infinity
    float f = floorf((v - v0) * inv_h);
    f = f < 0.0f ? 0.0f : (f > (float)(n - 1) ? (float)(n - 1) : f);
    return (int)f;
}

} // namespace icp
} // namespace slam

struct slam_icp_pending; // icp_build.hip: an index build that is enqueued and not yet adopted

struct slam_icp {
    slam_icp_params prm;
    int             sub_step = 10; // icp.cpp:27
    slam::icp::ModelView       mv;
    bool            in_lds = false;
    bool            start32 = false;
    int             G = 8;
    int             sweep = 0;     // 0 ring search, 2 halo-list sweeps (list_search) for every iteration
    void           *d_lblob = nullptr;
    size_t          list_lds_bytes = 0;
    bool            have_lists = false;
    size_t          lds_bytes = 0;
    void           *d_blob = nullptr;
    double         *d_normals = nullptr;
    double         *d_lnormals = nullptr; // normals per halo-list entry (P2L)
    slam::icp::DevBuf          w_pts, w_stamps, w_ew, w_state, w_single; // w_pts: slam_icp_fit's block (points + header)
    int             spread_points_hint = 0; // points of the batch when the caller knows them (slam_icp_fit), else 0
    size_t          step_pose_off = 0;   // where in w_pts the pose of the last executed step lies
    bool            two_phase = false;   // ring search, then list sweeps (the point-to-point default)
    int             pair = 0;             // two scans per workgroup in the fused schedule: lanes per point of its ring form; 0 = by batch size, -1 = never
    bool            split_launch = false; // SLAM_ICP_SPLIT=1: the two forms as two launches (measurements)
    bool            phase_events = false; // diagnostic: time the two launches separately (slam_icp_debug_phase_ms)
    hipEvent_t      ev[3] = {nullptr, nullptr, nullptr};
    double          phase_ms[2] = {0, 0};
    int             phase_calls = 0;
    bool            ev_pending = false;
    int             far_div = 32;        // see FitArgs
    int             switch_iter = 10;    // ring-search iterations before a scan may change to list sweeps (tools/switch_sweep.sh); point-to-line: 2 (icp_new)
    int             n_stamps = 0;
    int             spread_stamp_iters = 0; // measurement build: iterations per workgroup in the spread form's stamps (n_stamps < 0)
    bool            want_step_pose = false; // set around slam_icp_fit()
    int             last_n = 0, last_nga = 0; // template of the last slam_icp_fit()
    double          last_indist = 0;
    bool            have_last = false;
    int             n_cu = 256;          // CUs of the device the handle was made on
    int             max_cell_points = 0; // points in the fullest cell of the index (either class)
    bool            built_on_device = false;
    // slam_icp_fit: a spread launch whose workgroups did not become resident together (a persistent kernel of this or another
    // process holds CUs) costs its 5 ms first-exchange limit before the one-workgroup form redoes the scan.  After such a fit the
    // handle's next fits go straight to the one-workgroup form and try the spread form again later (icp_single.hip).
    unsigned        spread_tag = 1;             // tag base of the next spread launch (FitArgs::spread_tag)
    int             spread_backoff = 0;         // single fits still to be done without the spread form
    bool            skip_spread = false;        // set around one slam_icp_fit_batch_dev call
    int            *redo_mirror = nullptr;      // set around slam_icp_fit(): where the spread launch leaves its redo flag for the host (pinned)
    bool            build_beside = false;       // the index build's kernels must fit beside a resident registration workgroup (mapper)
    double          build_ms[4] = {0, 0, 0, 0}; // enqueueing the build, its one wait (the other two: unused since the plan moved to the device)
    slam_icp_pending *pending = nullptr;        // between build_index_begin and build_index_finish
};

namespace slam {
namespace icp {
// icp_build.hip: Icp::Icp's model copy + index (icp.cpp:26-70, kdtree.cpp:72-106 stand-in).  m_* are host
// arrays unless on_device; fills h->mv, h->d_blob, h->d_lblob, ...
int build_index(slam_icp *h, const double *m_ga, int n_ga, const double *m_nga, int n_nga, bool on_device);
// The same in two halves, for callers that keep the device busy meanwhile (the mapper's sliding target): begin enqueues the
// whole build on `st` and returns; the model may hold fewer points than cap_* -- d_cnt (int[2] in device memory, complete
// in stream order; null: exactly cap_*) says how many.  The handle is usable after finish (the build's one host wait;
// SLAM_E_TOO_FEW_MODEL_POINTS when d_cnt named fewer than 5); ready = finish would not block.
int  build_index_begin(slam_icp *h, const double *m_ga, int cap_ga, const double *m_nga, int cap_nga, const int *d_cnt, bool on_device,
                       hipStream_t st);
bool build_index_ready(slam_icp *h);
int  build_index_finish(slam_icp *h);
void release_index(slam_icp *h);
void destroy_unsynchronised(slam_icp *h); // slam_icp_destroy without its device synchronisation
} // namespace icp
} // namespace slam
