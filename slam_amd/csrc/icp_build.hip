// icp_build.hip -- the model side of Icp::Icp (ccicp2d/src/icp.cpp:26-70): the f64 -> f32 copy of the two
// model classes (:51-60) and the search structure that stands where the reference builds one kd-tree per
// class (:62-69, kdtree.cpp:72-106): a uniform-cell index (points sorted by lattice cell, per class) and,
// when they fit LDS, the halo lists of the list-sweep form.
//
// The reference constructs its matcher on every doICPMatch (icpTools.cpp:187), so the build is on the path
// of every match.  It runs on the device:
//   cell index : count per cell (atomics) -> exclusive scan -> unordered fill -> rank inside each cell by
//                original index (the order a stable counting sort gives);
//   halo lists : entry counts of every candidate pitch in one launch -> the host picks the first candidate
//                that fits (the one read-back of the build) -> count / scan / fill per list cell -> per cell
//                the ordering key (x, y, x+y, x-y) by the densest-window metric and a rank sort by (key, point).
// Every step is order-free or ranked, so the blobs are bit-identical to the single-threaded host build
// below, which is kept as the reference of that claim (slam_icp_params::build_on_host, tests/test_gpu_icp_build.py).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include "icp_model.hpp"
#include "icp_search.hpp"

using namespace slam;
using namespace slam::icp;

namespace {

__host__ __device__ inline unsigned align16(unsigned v) { return (v + 15u) & ~15u; }

struct BBox {
    float  lo[2] = {FLT_MAX, FLT_MAX}, hi[2] = {-FLT_MAX, -FLT_MAX};
    double sum[2] = {0, 0};
    size_t nfin = 0;
};

// ------------------------------------------------------------------ planning
// One set of expressions for the host build and for the device build's plan kernels (which run them on the device, so that
// the build needs the host only once, at its end): no FMA contraction, IEEE division and square root on both sides.

// Lattice, blob layout and the LDS decision from the model's extent and counts.
struct IndexPlan {
    Lattice  lat;
    int      n_cls[2], base[2];
    double   cx, cy;
    unsigned off_pts, off_start[2], off_oidx, blob_bytes;
    int      in_lds, start32;
    unsigned lds_bytes;
    float    maxabs;
    int      ncells;
};

__host__ __device__ inline IndexPlan plan_core(int n_ga, int n_nga, BBox bb, double cell_size, int force_global, unsigned lds_total)
{
#pragma clang fp contract(off)
    if (bb.nfin == 0) {
        bb.lo[0] = bb.lo[1] = 0.f;
        bb.hi[0] = bb.hi[1] = 1.f;
        bb.nfin = 1;
    }
    const float *lo = bb.lo, *hi = bb.hi;
    const int    n_all = n_ga + n_nga;
    const int    max_cls = n_ga > n_nga ? n_ga : n_nga;
    const unsigned scratch = kScratchBytes;

    // LDS budget for the two start arrays (u16 entries) after points + original indices
    const long fixed16 = (long)scratch + align16(8u * n_all) + align16(2u * n_all) + 64;
    long       cells_lds = ((long)lds_total - fixed16) / (2 * 2) - 1;
    bool       lds = !force_global && max_cls <= 65535 && cells_lds >= 256;

    const float w = hi[0] - lo[0] > 1e-3f ? hi[0] - lo[0] : 1e-3f, ht = hi[1] - lo[1] > 1e-3f ? hi[1] - lo[1] : 1e-3f;
    const float ax = fabsf(lo[0]) > fabsf(hi[0]) ? fabsf(lo[0]) : fabsf(hi[0]), ay = fabsf(lo[1]) > fabsf(hi[1]) ? fabsf(lo[1]) : fabsf(hi[1]);
    const float maxabs = ax > ay ? ax : ay;
    long        hbm = 4L * n_all > 1024 ? 4L * n_all : 1024;
    if (hbm > (1L << 22)) hbm = 1L << 22;
    const long budget = lds ? cells_lds : hbm;
    // target about two cells per point on wall-like maps; never more than the budget
    long want = 2L * n_all > 64 ? 2L * n_all : 64;
    if (want > budget) want = budget;
    double hcell = cell_size > 0 ? cell_size : sqrt((double)w * ht / (double)want);
    if (hcell < (double)maxabs * 1.52587890625e-05 /* 2^-16 */) hcell = (double)maxabs * 1.52587890625e-05;
    if (hcell < 1e-4) hcell = 1e-4;
    int nx, ny;
    for (;;) {
        nx = (int)floor(w / hcell) + 1;
        ny = (int)floor(ht / hcell) + 1;
        if ((long)nx * ny <= budget) break;
        hcell *= 1.05;
    }

    IndexPlan ip;
    ip.lat.nx = nx;
    ip.lat.ny = ny;
    ip.lat.x0 = lo[0];
    ip.lat.y0 = lo[1];
    ip.lat.h = (float)hcell;
    ip.lat.inv_h = 1.0f / ip.lat.h;
    // the cell map floor(fl(fl(x-x0)*inv_h)) is monotone and off by at most ~3*2^-24*nx cells per evaluation,
    // i.e. ~6*2^-24*maxabs metres for a model point and a query together; 2^-19*maxabs covers that 5x
    // (and stays below h/8 by the choice of h above)
    const float m_h = ip.lat.h * 0.0009765625f, m_a = maxabs * 1.9073486328125e-06f;
    ip.lat.margin = m_h > m_a ? m_h : m_a;
    ip.n_cls[0] = n_ga;
    ip.n_cls[1] = n_nga;
    ip.base[0] = 0;
    ip.base[1] = n_ga;
    ip.cx = bb.sum[0] / (double)bb.nfin;
    ip.cy = bb.sum[1] / (double)bb.nfin;

    ip.start32 = !lds; // the HBM-resident index always uses 32-bit positions
    const unsigned esz = ip.start32 ? 4u : 2u;
    const int      ncells = nx * ny;
    unsigned       o = 0;
    ip.off_pts = o;
    o = align16(o + 8u * (unsigned)n_all);
    ip.off_start[0] = o;
    o = align16(o + esz * (unsigned)(ncells + 1));
    ip.off_start[1] = o;
    o = align16(o + esz * (unsigned)(ncells + 1));
    ip.off_oidx = o;
    o = align16(o + esz * (unsigned)n_all);
    ip.blob_bytes = o;
    if (lds && scratch + o > lds_total) lds = false, ip.start32 = 0; // keeps u16 entries, read from HBM
    ip.in_lds = lds;
    ip.lds_bytes = lds ? scratch + o : scratch;
    ip.maxabs = maxabs;
    ip.ncells = ncells;
    return ip;
}

// cells the index can have at most for a model of up to n_all points (what a build reserves before it knows the extent)
inline long cells_bound(int n_all, unsigned lds_total)
{
    const long lds = ((long)lds_total - (long)kScratchBytes - 64) / 4;
    long       hbm = std::max<long>(4L * n_all, 1024);
    hbm = std::min<long>(hbm, 1L << 22);
    return std::max(lds, hbm) + 1;
}

unsigned device_lds_total()
{
    int lds_cap = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&lds_cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return std::min<unsigned>((unsigned)lds_cap, kLdsTotal);
}

// the index's part of a ModelView from its plan (the host adopting a plan; the device making the view its own kernels need)
__host__ __device__ inline void view_index(ModelView &mv, const IndexPlan &ip)
{
    mv.lat = ip.lat;
    for (int c = 0; c < 2; ++c) {
        mv.n_cls[c] = ip.n_cls[c];
        mv.base[c] = ip.base[c];
        mv.off_start[c] = ip.off_start[c];
    }
    mv.cx = ip.cx;
    mv.cy = ip.cy;
    mv.off_pts = ip.off_pts;
    mv.off_oidx = ip.off_oidx;
    mv.blob_bytes = ip.blob_bytes;
}

void adopt_index_plan(slam_icp *h, const IndexPlan &ip)
{
    ModelView &mv = h->mv;
    memset(&mv, 0, sizeof mv);
    view_index(mv, ip);
    h->start32 = ip.start32 != 0;
    h->in_lds = ip.in_lds != 0;
    h->lds_bytes = ip.lds_bytes;
}

int plan_index(slam_icp *h, int n_ga, int n_nga, const BBox &bb, unsigned *lds_total_out, float *maxabs_out)
{
    const unsigned lds_total = device_lds_total();
    SLAM_REQUIRE(lds_total, SLAM_E_HIP, "the device does not report its LDS size");
    const IndexPlan ip = plan_core(n_ga, n_nga, bb, h->prm.cell_size, h->prm.force_global, lds_total);
    adopt_index_plan(h, ip);
    *lds_total_out = lds_total;
    *maxabs_out = ip.maxabs;
    return SLAM_OK;
}

// slam_icp_params::list_min_halo resolved (0 = the library's default)
constexpr double kListMinHaloDefault = 0.125; // round 5: one launch of 256 scans against a 5 k-point model 0.68 -> 0.48 ms in pairs (tools/exp/halo_forms.py)
inline double list_min_halo_of(const slam_icp_params &prm) { return prm.list_min_halo > 0 ? prm.list_min_halo : (prm.list_min_halo < 0 ? 0.0 : kListMinHaloDefault); }

// The list lattices tried, in order: the first whose lists fit LDS beside the scratch is built.
struct ListCand {
    double frac, hs, pad_m, x0, y0;
    long   nx, ny;
};
constexpr int kPitchSteps = 49;               // 0.25 * 1.12^k <= 64
constexpr int kMaxCand = 2 * kPitchSteps;

// candidate (frac, hs) over the extent bb; false when it is not tried
__host__ __device__ inline bool list_candidate(const BBox &bb, float margin_abs, double frac, double hs, ListCand &c)
{
#pragma clang fp contract(off)
    c.frac = frac;
    c.hs = hs;
    c.pad_m = hs * frac; // halo in metres
    if (c.pad_m < 8.0 * margin_abs) return false;
    c.x0 = (double)bb.lo[0] - c.pad_m;
    c.y0 = (double)bb.lo[1] - c.pad_m;
    c.nx = (long)floor(((double)bb.hi[0] + c.pad_m - c.x0) / hs) + 1;
    c.ny = (long)floor(((double)bb.hi[1] + c.pad_m - c.y0) / hs) + 1;
    return c.nx * c.ny <= 60000;
}

void list_candidates(const BBox &bb, float margin_abs, std::vector<ListCand> &out)
{
    for (double frac : {0.25, 0.125}) {
        for (double hs = 0.25; hs <= 64.0; hs *= 1.12) {
            ListCand c;
            if (list_candidate(bb, margin_abs, frac, hs, c)) out.push_back(c);
        }
    }
}

// cells [ax, bx] x [ay, by] of candidate lattice c that the halo of point (px, py) reaches
struct ListGeom {
    double pad_m, x0, y0, hs;
    int    nx, ny;
};
__host__ __device__ inline void halo_cells(const ListGeom &g, double px, double py, long &ax, long &bx, long &ay, long &by)
{
    const long fax = (long)floor((px - g.pad_m - g.x0) / g.hs), fbx = (long)floor((px + g.pad_m - g.x0) / g.hs);
    const long fay = (long)floor((py - g.pad_m - g.y0) / g.hs), fby = (long)floor((py + g.pad_m - g.y0) / g.hs);
    ax = fax > 0 ? fax : 0;
    bx = fbx < g.nx - 1 ? fbx : g.nx - 1;
    ay = fay > 0 ? fay : 0;
    by = fby < g.ny - 1 ? fby : g.ny - 1;
}

__host__ __device__ inline ListGeom geom_of(const ListCand &c)
{
    ListGeom g;
    g.pad_m = c.pad_m;
    g.x0 = c.x0;
    g.y0 = c.y0;
    g.hs = c.hs;
    g.nx = (int)c.nx;
    g.ny = (int)c.ny;
    return g;
}

// The list side of the index for one candidate lattice
struct ListPlan {
    unsigned           loff_pts, loff_start[2], loff_axis[2], lblob_bytes;
    int                lbase[2];
    Lattice            llat;
    float              lpad, lkeps, cert2;
    int                ncells;
    unsigned long long n_ent[2];
};

// Whether candidate c with n_ent[] entries per class is the one to build; lp is its layout if so.
__host__ __device__ inline bool accept_list_core(const ListCand &c, const unsigned long long n_ent[2], double budget, float maxabs,
                                                 float margin_abs, ListPlan &lp)
{
#pragma clang fp contract(off)
    if (n_ent[0] > 65535 || n_ent[1] > 65535) return false;
    const int    ncells = (int)(c.nx * c.ny);
    const size_t bytes = align16(8u * (unsigned)(n_ent[0] + n_ent[1])) + 2 * (size_t)align16(2u * (unsigned)(ncells + 1)) +
                         2 * (size_t)align16(4u * (unsigned)(ncells / 16 + 1));
    if ((double)bytes > budget) return false;
    unsigned o = 0;
    lp.loff_pts = o;
    o = align16(o + 8u * (unsigned)(n_ent[0] + n_ent[1]));
    for (int k = 0; k < 2; ++k) {
        lp.loff_start[k] = o;
        o = align16(o + 2u * (unsigned)(ncells + 1));
    }
    for (int k = 0; k < 2; ++k) {
        lp.loff_axis[k] = o;
        o = align16(o + 4u * (unsigned)(ncells / 16 + 1));
    }
    lp.lblob_bytes = o;
    lp.lbase[0] = 0;
    lp.lbase[1] = (int)n_ent[0];
    lp.llat.nx = (int)c.nx;
    lp.llat.ny = (int)c.ny;
    lp.llat.x0 = (float)c.x0;
    lp.llat.y0 = (float)c.y0;
    lp.llat.h = (float)c.hs;
    lp.llat.inv_h = 1.0f / lp.llat.h;
    const float m_h = lp.llat.h * 0.0009765625f;
    lp.llat.margin = m_h > margin_abs ? m_h : margin_abs;
    lp.lpad = (float)c.frac;
    lp.lkeps = 8.0f * 2.0f * maxabs * 1.1920929e-07f; // 8 ulp of |x| + |y| <= 2 maxabs (query within the lattice)
    // a point within `cert` of a query lies within cert + (cell-map rounding) of the query's nominal cell
    const double cert = c.pad_m - 4.0 * (double)lp.llat.margin - 2.0 * fabs((double)lp.llat.x0 - c.x0) -
                        2.0 * fabs((double)lp.llat.y0 - c.y0);
    if (cert <= 0) return false;
    lp.cert2 = (float)(cert * cert * 0.999);
    lp.ncells = ncells;
    lp.n_ent[0] = n_ent[0];
    lp.n_ent[1] = n_ent[1];
    return true;
}

__host__ __device__ inline void view_lists(ModelView &mv, const ListPlan &lp)
{
    mv.loff_pts = lp.loff_pts;
    for (int k = 0; k < 2; ++k) {
        mv.loff_start[k] = lp.loff_start[k];
        mv.loff_axis[k] = lp.loff_axis[k];
        mv.lbase[k] = lp.lbase[k];
    }
    mv.lblob_bytes = lp.lblob_bytes;
    mv.llat = lp.llat;
    mv.lpad = lp.lpad;
    mv.lkeps = lp.lkeps;
    mv.cert2 = lp.cert2;
}

void adopt_list_plan(slam_icp *h, const ListPlan &lp) { view_lists(h->mv, lp); }

bool accept_list(slam_icp *h, const ListCand &c, const size_t n_ent[2], double budget, float maxabs, float margin_abs)
{
    const unsigned long long n[2] = {(unsigned long long)n_ent[0], (unsigned long long)n_ent[1]};
    ListPlan                 lp;
    if (!accept_list_core(c, n, budget, maxabs, margin_abs, lp)) return false;
    adopt_list_plan(h, lp);
    return true;
}

// a window of the converged search radius (+-3 cm around the query) along key `dir`
__host__ __device__ inline float key_window(int dir) { return (dir < 2 ? 1.0f : 1.41421356f) * 0.06f; }

void list_done(slam_icp *h)
{
    h->list_lds_bytes = kScratchBytes + h->mv.lblob_bytes;
    h->have_lists = true;
}

// ------------------------------------------------------------------ host build (the reference of the device build)

template <typename StartT>
void fill_index_host(std::vector<unsigned char> &blob, const ModelView &mv, const std::vector<float> cls_xy[2],
                     const std::vector<int> cell_of[2])
{
    const int ncells = mv.lat.nx * mv.lat.ny;
    float2   *pts = reinterpret_cast<float2 *>(blob.data() + mv.off_pts);
    StartT   *oidx = reinterpret_cast<StartT *>(blob.data() + mv.off_oidx);
    for (int c = 0; c < 2; ++c) {
        StartT          *start = reinterpret_cast<StartT *>(blob.data() + mv.off_start[c]);
        const int        n = mv.n_cls[c];
        std::vector<int> count(ncells + 1, 0);
        for (int i = 0; i < n; ++i) count[cell_of[c][i] + 1]++;
        for (int k = 0; k < ncells; ++k) count[k + 1] += count[k];
        for (int k = 0; k <= ncells; ++k) start[k] = (StartT)count[k];
        std::vector<int> fill(count.begin(), count.end() - 1);
        for (int i = 0; i < n; ++i) { // stable: equal cells keep original order
            const int pos = fill[cell_of[c][i]]++;
            pts[mv.base[c] + pos] = make_float2(cls_xy[c][2 * i], cls_xy[c][2 * i + 1]);
            oidx[mv.base[c] + pos] = (StartT)i;
        }
    }
}

int build_lists_host(slam_icp *h, const std::vector<float> xy[2], const int cnt[2], const BBox &bb, float maxabs,
                     unsigned lds_total)
{
    ModelView &mv = h->mv;
    h->have_lists = false;
    const double budget = (double)lds_total - (double)kScratchBytes - 64.0;
    const float  margin_abs = maxabs * 1.9073486328125e-06f; // 2^-19 * maxabs, as for the cell lattice
    struct Ent {
        int   cell;
        float key;
        int   pt;
    };
    std::vector<ListCand> cands;
    list_candidates(bb, margin_abs, cands);
    // the candidates whose halo reaches list_min_halo come first (in their order), then the others: a halo the converged
    // neighbour distances do not fit under sends the queries to the cooperative round one by one (list_min_halo, slam_mi355x.h)
    const double min_halo = list_min_halo_of(h->prm);
    std::stable_partition(cands.begin(), cands.end(), [&](const ListCand &c) { return c.pad_m >= min_halo; });
    for (const ListCand &cand : cands) {
        const ListGeom g = geom_of(cand);
        const long     nx = cand.nx;
        const int      ncells = (int)(cand.nx * cand.ny);
        size_t         n_ent[2] = {0, 0};
        bool           ok = true;
        for (int c = 0; c < 2 && ok; ++c) {
            for (int i = 0; i < cnt[c]; ++i) {
                const double px = xy[c][2 * i], py = xy[c][2 * i + 1];
                if (!std::isfinite(px) || !std::isfinite(py)) continue;
                long ax, bx, ay, by;
                halo_cells(g, px, py, ax, bx, ay, by);
                n_ent[c] += (size_t)((bx - ax + 1) * (by - ay + 1));
            }
            if (n_ent[c] > 65535) ok = false;
        }
        if (!ok) continue;
        if (!accept_list(h, cand, n_ent, budget, maxabs, margin_abs)) continue;
        std::vector<unsigned char> blob(mv.lblob_bytes, 0);
        float2                    *lpts = reinterpret_cast<float2 *>(blob.data() + mv.loff_pts);
        for (int c = 0; c < 2; ++c) {
            std::vector<Ent> ent;
            ent.reserve(n_ent[c]);
            for (int i = 0; i < cnt[c]; ++i) {
                const double px = xy[c][2 * i], py = xy[c][2 * i + 1];
                if (!std::isfinite(px) || !std::isfinite(py)) continue;
                long ax, bx, ay, by;
                halo_cells(g, px, py, ax, bx, ay, by);
                for (long yy = ay; yy <= by; ++yy)
                    for (long xx = ax; xx <= bx; ++xx) ent.push_back({(int)(yy * nx + xx), 0.f, i});
            }
            std::stable_sort(ent.begin(), ent.end(), [](const Ent &a, const Ent &b) { return a.cell < b.cell; });
            unsigned short *start = reinterpret_cast<unsigned short *>(blob.data() + mv.loff_start[c]);
            unsigned       *axis = reinterpret_cast<unsigned *>(blob.data() + mv.loff_axis[c]);
            size_t          a = 0;
            for (int k = 0; k < ncells; ++k) {
                start[k] = (unsigned short)a;
                size_t e = a;
                float  mn[2] = {FLT_MAX, FLT_MAX}, mx[2] = {-FLT_MAX, -FLT_MAX};
                while (e < ent.size() && ent[e].cell == k) {
                    for (int d = 0; d < 2; ++d) {
                        const float v = xy[c][2 * ent[e].pt + d];
                        mn[d] = std::min(mn[d], v);
                        mx[d] = std::max(mx[d], v);
                    }
                    ++e;
                }
                // ordering key: the direction (x, y, x+y, x-y) whose densest key window is the sparsest --
                // a window of the converged search radius must hold few entries, or the walk is long
                int best_dir = (mx[1] - mn[1]) > (mx[0] - mn[0]) ? 1 : 0;
                if (e - a >= 8) {
                    size_t             best_metric = SIZE_MAX;
                    std::vector<float> keys(e - a);
                    for (int dir : {best_dir, 1 - best_dir, 2, 3}) {
                        for (size_t j = a; j < e; ++j) keys[j - a] = list_key(dir, xy[c][2 * ent[j].pt], xy[c][2 * ent[j].pt + 1]);
                        std::sort(keys.begin(), keys.end());
                        const float win = key_window(dir);
                        size_t      metric = 0, lo_j = 0;
                        for (size_t j = 0; j < keys.size(); ++j) {
                            while (keys[j] - keys[lo_j] > win) ++lo_j;
                            metric = std::max(metric, j - lo_j + 1);
                        }
                        if (dir >= 2) metric += metric / 4 + 1; // an axis key is cheaper and exact: prefer it when close
                        if (metric < best_metric) {
                            best_metric = metric;
                            best_dir = dir;
                        }
                    }
                }
                axis[k >> 4] |= (unsigned)best_dir << (2 * (k & 15));
                for (size_t j = a; j < e; ++j) ent[j].key = list_key(best_dir, xy[c][2 * ent[j].pt], xy[c][2 * ent[j].pt + 1]);
                std::stable_sort(ent.begin() + a, ent.begin() + e, [](const Ent &p, const Ent &q) { return p.key < q.key; });
                a = e;
            }
            start[ncells] = (unsigned short)a;
            for (size_t j = 0; j < ent.size(); ++j)
                lpts[mv.lbase[c] + j] = make_float2(xy[c][2 * ent[j].pt], xy[c][2 * ent[j].pt + 1]);
        }
        h->d_lblob = pool_alloc(mv.lblob_bytes);
        if (!h->d_lblob) return SLAM_E_NOMEM;
        SLAM_HIP(hipMemcpy(h->d_lblob, blob.data(), mv.lblob_bytes, hipMemcpyHostToDevice));
        mv.lblob = static_cast<const unsigned char *>(h->d_lblob);
        list_done(h);
        return SLAM_OK;
    }
    return SLAM_OK;
}

BBox host_bbox(const double *const src[2], const int cnt[2])
{
    BBox bb;
    for (int c = 0; c < 2; ++c)
        for (int i = 0; i < cnt[c]; ++i) {
            const float x = (float)src[c][2 * i], y = (float)src[c][2 * i + 1]; // icp.cpp:54,60
            if (!std::isfinite(x) || !std::isfinite(y)) continue;
            bb.lo[0] = std::min(bb.lo[0], x);
            bb.hi[0] = std::max(bb.hi[0], x);
            bb.lo[1] = std::min(bb.lo[1], y);
            bb.hi[1] = std::max(bb.hi[1], y);
            bb.sum[0] += x;
            bb.sum[1] += y;
            ++bb.nfin;
        }
    return bb;
}

int build_index_host(slam_icp *h, const double *m_ga, int n_ga, const double *m_nga, int n_nga)
{
    const double *src[2] = {m_ga, m_nga};
    const int     cnt[2] = {n_ga, n_nga};
    const BBox    bb = host_bbox(src, cnt);
    unsigned      lds_total = 0;
    float         maxabs = 0;
    SLAM_TRY(plan_index(h, n_ga, n_nga, bb, &lds_total, &maxabs));
    ModelView &mv = h->mv;

    std::vector<float> xy[2];
    std::vector<int>   cell_of[2];
    for (int c = 0; c < 2; ++c) {
        xy[c].resize(2 * (size_t)cnt[c]);
        for (int i = 0; i < 2 * cnt[c]; ++i) xy[c][i] = (float)src[c][i]; // icp.cpp:54,60
        cell_of[c].resize(cnt[c]);
        for (int i = 0; i < cnt[c]; ++i)
            cell_of[c][i] = lattice_coord(xy[c][2 * i + 1], mv.lat.y0, mv.lat.inv_h, mv.lat.ny) * mv.lat.nx +
                            lattice_coord(xy[c][2 * i], mv.lat.x0, mv.lat.inv_h, mv.lat.nx);
    }
    for (int c = 0; c < 2; ++c) { // points in the fullest cell (picks the one-scan form's lanes per query)
        std::vector<int> per(mv.lat.nx * mv.lat.ny, 0);
        for (int v : cell_of[c]) h->max_cell_points = std::max(h->max_cell_points, ++per[v]);
    }
    std::vector<unsigned char> blob(mv.blob_bytes, 0);
    if (h->start32)
        fill_index_host<uint32_t>(blob, mv, xy, cell_of);
    else
        fill_index_host<uint16_t>(blob, mv, xy, cell_of);

    h->d_blob = pool_alloc(mv.blob_bytes);
    if (!h->d_blob) return SLAM_E_NOMEM;
    SLAM_HIP(hipMemcpy(h->d_blob, blob.data(), mv.blob_bytes, hipMemcpyHostToDevice));
    mv.blob = static_cast<const unsigned char *>(h->d_blob);
    BBox bb_lists = bb;
    if (bb_lists.nfin == 0) {
        bb_lists.lo[0] = bb_lists.lo[1] = 0.f;
        bb_lists.hi[0] = bb_lists.hi[1] = 1.f;
    }
    if (h->sweep == 2 || h->two_phase) SLAM_TRY(build_lists_host(h, xy, cnt, bb_lists, maxabs, lds_total));
    return SLAM_OK;
}

// ------------------------------------------------------------------ device build
// The whole build is enqueued on one stream without a host wait in between: extent -> plan kernel (the lattice, the blob
// layout, the list candidates, all by the expressions above) -> cell index -> candidates' entry counts -> list plan kernel
// (the first candidate that fits) -> lists; every kernel reads the plan from device memory and is launched over what the
// model can be at most (`cap` points per class: the mapper's thinned window knows its counts only on the device).  The
// host reads the plan back ONCE, when it needs the handle (build_index_finish).

struct DevPlan {
    int                cnt[2];           // points per class
    BBox               bb;               // extent, sum and count of the finite points (fixed order)
    IndexPlan          ip;
    float              margin_abs;
    double             budget;           // bytes the lists may take
    int                nc;               // list candidates
    int                pick;             // the candidate built, -1: none
    ListGeom           geom[kMaxCand];
    double             frac[kMaxCand];
    unsigned long long n_ent[kMaxCand][2]; // entries per class of every candidate (zeroed; atomics)
    ListPlan           lp;
    unsigned           most;             // points in the fullest cell of the index (zeroed; atomicMax)
};

struct BuildWs {               // what the host knows when it enqueues the build
    const double  *m[2];       // model classes, f64 xy (device)
    const int     *d_cnt;      // points per class on the device, or null: host_cnt
    int            host_cnt[2];
    int            cap[2];     // points per class at most
    DevPlan       *plan;
    double        *rows;       // [n_rows][8] extent partials
    int            n_rows;
    float2        *xyf;        // [cap_all] the f32 model (icp.cpp:54,60), class order
    int           *cell_of;    // [cap_all]
    int           *tmp;        // [cap_all] original indices bucketed by cell, unordered inside a cell
    unsigned      *cellcnt;    // [2][ncells + 1] counts, then exclusive prefix | [2][ncells] cursors (zeroed)
    unsigned      *tiles;      // scan tile totals
    unsigned      *lcnt;       // the same two arrays for the list lattice (zeroed)
    int           *ent;        // point of every list entry, bucketed by list cell
    unsigned char *blob, *lblob;
    double         cell_size;
    int            force_global, want_lists;
    double         list_min_halo;
    unsigned       lds_total;
    // point-to-line (SLAM_ICP_P2L): the two classes as the caller gave them, merged into ONE class (class 1, GA then NGA: the order
    // of the reference's M_normal) before anything else looks at the model; normals per model point and per halo-list entry
    const double  *p2l_src[2];  // the caller's classes (device), or null: not a point-to-line build
    const int     *p2l_src_cnt; // their counts on the device, or null: p2l_host_cnt
    int            p2l_host_cnt[2];
    double2       *p2l_all;     // [cap_all] the merged model (m[1] points here)
    int           *p2l_cnt;     // [2] = {0, n_all} on the device (d_cnt points here)
    double        *normals;     // [cap_all][2]
    double2       *lnormals;    // [lnormals_cap]
    int           *p2l_lidx;    // [lnormals_cap] the model point of every list entry (written by the list sorts)
    int            lnormals_cap;
};

struct BuildArgs {
    const double *m[2];
    int           cnt[2], base[2], n_all;
    Lattice       lat;
    int           ncells;
    float2       *xyf;
    int          *cell_of;
    unsigned     *start;   // [2][ncells + 1] counts, then exclusive prefix
    unsigned     *cursor;  // [2][ncells]
    int          *tmp;
    unsigned char *blob;
    unsigned      off_pts, off_start[2], off_oidx;
    int           esz;     // bytes per start / oidx entry in the blob: 2 or 4
};

__device__ inline BuildArgs build_args(const BuildWs &w)
{
    const DevPlan &p = *w.plan;
    BuildArgs      a;
    a.m[0] = w.m[0];
    a.m[1] = w.m[1];
    a.cnt[0] = p.cnt[0];
    a.cnt[1] = p.cnt[1];
    a.base[0] = 0;
    a.base[1] = p.cnt[0];
    a.n_all = p.cnt[0] + p.cnt[1];
    a.lat = p.ip.lat;
    a.ncells = p.ip.ncells;
    a.xyf = w.xyf;
    a.cell_of = w.cell_of;
    a.start = w.cellcnt;
    a.cursor = w.cellcnt + 2 * (size_t)(a.ncells + 1);
    a.tmp = w.tmp;
    a.blob = w.blob;
    a.off_pts = p.ip.off_pts;
    a.off_start[0] = p.ip.off_start[0];
    a.off_start[1] = p.ip.off_start[1];
    a.off_oidx = p.ip.off_oidx;
    a.esz = p.ip.start32 ? 4 : 2;
    return a;
}

__device__ inline void store_entry(unsigned char *base, int esz, size_t i, unsigned v)
{
    if (esz == 2)
        reinterpret_cast<unsigned short *>(base)[i] = (unsigned short)v;
    else
        reinterpret_cast<unsigned *>(base)[i] = v;
}

__device__ inline int ws_count(const BuildWs &w, int c) { return w.d_cnt ? w.d_cnt[c] : w.host_cnt[c]; }

// extent, sum and count of the finite points of a device-resident model: one row of partials per workgroup
// (fixed order: the plan kernel adds the rows in index order)
__global__ __launch_bounds__(256) void idx_bbox_kernel(BuildWs w)
{
    __shared__ double s[4][8];
    const int n_ga = min(max(ws_count(w, 0), 0), w.cap[0]), n_nga = min(max(ws_count(w, 1), 0), w.cap[1]);
    const int i = blockIdx.x * 256 + threadIdx.x, n_all = n_ga + n_nga;
    float     lx = FLT_MAX, ly = FLT_MAX, hx = -FLT_MAX, hy = -FLT_MAX;
    double    sx = 0, sy = 0, nf = 0;
    if (i < n_all) {
        const double *p = i < n_ga ? w.m[0] + 2 * (size_t)i : w.m[1] + 2 * (size_t)(i - n_ga);
        const float   x = (float)p[0], y = (float)p[1];
        if ((x - x <= 0.0f) && (y - y <= 0.0f)) {
            lx = hx = x;
            ly = hy = y;
            sx = x;
            sy = y;
            nf = 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lx = fminf(lx, __shfl_xor(lx, o));
        ly = fminf(ly, __shfl_xor(ly, o));
        hx = fmaxf(hx, __shfl_xor(hx, o));
        hy = fmaxf(hy, __shfl_xor(hy, o));
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
        nf += __shfl_xor(nf, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s[wave][0] = lx, s[wave][1] = ly, s[wave][2] = hx, s[wave][3] = hy;
        s[wave][4] = sx, s[wave][5] = sy, s[wave][6] = nf;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *r = w.rows + 8 * (size_t)blockIdx.x;
        r[0] = fmin(fmin(s[0][0], s[1][0]), fmin(s[2][0], s[3][0]));
        r[1] = fmin(fmin(s[0][1], s[1][1]), fmin(s[2][1], s[3][1]));
        r[2] = fmax(fmax(s[0][2], s[1][2]), fmax(s[2][2], s[3][2]));
        r[3] = fmax(fmax(s[0][3], s[1][3]), fmax(s[2][3], s[3][3]));
        r[4] = (s[0][4] + s[1][4]) + (s[2][4] + s[3][4]);
        r[5] = (s[0][5] + s[1][5]) + (s[2][5] + s[3][5]);
        r[6] = (s[0][6] + s[1][6]) + (s[2][6] + s[3][6]);
        r[7] = 0;
    }
}

// One wavefront: the rows of idx_bbox_kernel added in index order, plan_core, the list candidates.
constexpr int kPlanRows = 64; // rows staged per round (1.5 KB of LDS: the kernel sits beside a registration workgroup)
__global__ __launch_bounds__(64, 8) void plan_kernel(BuildWs w) // (8 wavefronts per SIMD: at most 64 registers, see kScanThreads)
{
    __shared__ double s_sum[kPlanRows][3];
    __shared__ double s_hs[kPitchSteps];
    __shared__ int    s_ok[2 * 64];
    const int         lane = threadIdx.x;
    DevPlan          &p = *w.plan;
    const int         n_ga = min(max(ws_count(w, 0), 0), w.cap[0]), n_nga = min(max(ws_count(w, 1), 0), w.cap[1]);
    const int         rows_used = (n_ga + n_nga + 255) / 256;
    BBox              bb;
    float             lx = FLT_MAX, ly = FLT_MAX, hx = -FLT_MAX, hy = -FLT_MAX;
    double            sx = 0, sy = 0, nf = 0; // lane 0's
    for (int r0 = 0; r0 < rows_used; r0 += kPlanRows) {
        const int n = min(kPlanRows, rows_used - r0);
        for (int r = lane; r < n; r += 64) {
            const double *row = w.rows + 8 * (size_t)(r0 + r);
            s_sum[r][0] = row[4], s_sum[r][1] = row[5], s_sum[r][2] = row[6];
            if (row[6] > 0) {
                lx = fminf(lx, (float)row[0]);
                ly = fminf(ly, (float)row[1]);
                hx = fmaxf(hx, (float)row[2]);
                hy = fmaxf(hy, (float)row[3]);
            }
        }
        __syncthreads();
        if (lane == 0)
            for (int r = 0; r < n; ++r)
                if (s_sum[r][2] > 0) sx += s_sum[r][0], sy += s_sum[r][1], nf += s_sum[r][2];
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) {
        lx = fminf(lx, __shfl_xor(lx, o));
        ly = fminf(ly, __shfl_xor(ly, o));
        hx = fmaxf(hx, __shfl_xor(hx, o));
        hy = fmaxf(hy, __shfl_xor(hy, o));
    }
    sx = __shfl(sx, 0), sy = __shfl(sy, 0), nf = __shfl(nf, 0);
    bb.lo[0] = lx, bb.lo[1] = ly, bb.hi[0] = hx, bb.hi[1] = hy;
    bb.sum[0] = sx, bb.sum[1] = sy;
    bb.nfin = (size_t)nf;
    const IndexPlan ip = plan_core(n_ga, n_nga, bb, w.cell_size, w.force_global, w.lds_total); // (every lane: uniform)
    const size_t    nfin = bb.nfin;
    if (bb.nfin == 0) {
        bb.lo[0] = bb.lo[1] = 0.f;
        bb.hi[0] = bb.hi[1] = 1.f;
    }
    const double budget = (double)w.lds_total - (double)kScratchBytes - 64.0;
    const float  margin_abs = ip.maxabs * 1.9073486328125e-06f; // 2^-19 * maxabs, as for the cell lattice
    // candidates: a model whose points alone overflow the budget has no lists
    const bool lists = w.want_lists && 8.0 * (double)nfin <= budget && nfin <= 2 * 65535;
    if (lane == 0) {
        double hs = 0.25;
        for (int k = 0; k < kPitchSteps; ++k) {
            s_hs[k] = hs <= 64.0 ? hs : 0.0;
            hs *= 1.12;
        }
    }
    __syncthreads();
    int nc = 0;
    if (lists) {
        ListCand c[2];
        bool     ok[2];
        for (int t = 0; t < 2; ++t) { // candidate t * 64 + lane = (frac index, pitch step) in the host's order
            const int k = t * 64 + lane, fi = k / kPitchSteps, step = k % kPitchSteps;
            ok[t] = k < kMaxCand && s_hs[step] > 0.0 && list_candidate(bb, margin_abs, fi ? 0.125 : 0.25, s_hs[step], c[t]);
            s_ok[k] = ok[t] ? 1 : 0;
        }
        __syncthreads();
        for (int t = 0; t < 2; ++t) {
            const int k = t * 64 + lane;
            int       pos = 0;
            for (int j = 0; j < k; ++j) pos += s_ok[j];
            if (ok[t]) {
                p.geom[pos] = geom_of(c[t]);
                p.frac[pos] = c[t].frac;
            }
        }
        for (int j = 0; j < 2 * 64; ++j) nc += s_ok[j];
    }
    if (lane == 0) {
        p.cnt[0] = n_ga;
        p.cnt[1] = n_nga;
        p.bb = bb;
        p.bb.nfin = nfin;
        p.ip = ip;
        p.margin_abs = margin_abs;
        p.budget = budget;
        p.nc = nc;
        p.pick = -1;
    }
}

// icp.cpp:54,60: the f32 copy of the model; cell of every point; points per cell
__global__ __launch_bounds__(256) void idx_count_kernel(BuildWs w)
{
    const BuildArgs a = build_args(w);
    const int       i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_all) return;
    const int     c = i >= a.cnt[0] ? 1 : 0;
    const double *p = a.m[c] + 2 * (size_t)(i - a.base[c]);
    const float   x = (float)p[0], y = (float)p[1];
    a.xyf[i] = make_float2(x, y);
    const int cell = lattice_coord(y, a.lat.y0, a.lat.inv_h, a.lat.ny) * a.lat.nx + lattice_coord(x, a.lat.x0, a.lat.inv_h, a.lat.nx);
    a.cell_of[i] = cell;
    atomicAdd(&a.start[(size_t)c * (a.ncells + 1) + cell], 1u);
}

// Exclusive prefix of the per-cell counts of both classes, in place (v[n] = total), also stored as `esz`-byte
// entries in the blob's start arrays.  Two launches over tiles of 2048 values: the tiles' totals, then every
// tile adds the totals before it (a few dozen values) and scans itself -- a single workgroup walking 80 k cells
// took 100 us, one CU's share of the bandwidth.  Workgroups of four wavefronts (round 4; sixteen before): one wavefront
// per SIMD and 40 registers fit into what a registration workgroup leaves of a CU, so a rebuild of the mapper's sliding
// target runs beside the registrations instead of waiting for CUs at the launch boundaries.
constexpr int kScanThreads = 256, kScanPer = 8, kScanTile = kScanThreads * kScanPer;

struct ScanArgs {
    unsigned      *v[2];    // counts of class 0 / 1, n + 1 values each
    unsigned char *out[2];  // the blob's start arrays
    unsigned      *tiles;   // [2][n_tiles] tile totals
    unsigned      *most;    // nullable: receives the largest count (points in the fullest cell)
    int            n, n_tiles, esz;
};

// LISTS = 0: the cell index; 1: the list lattice (false: nothing to scan)
template <int LISTS>
__device__ inline bool scan_args(const BuildWs &w, ScanArgs &a)
{
    const DevPlan &p = *w.plan;
    if (LISTS) {
        if (p.pick < 0) return false;
        a.n = p.lp.ncells;
        a.v[0] = w.lcnt;
        a.v[1] = w.lcnt + (a.n + 1);
        a.out[0] = w.lblob + p.lp.loff_start[0];
        a.out[1] = w.lblob + p.lp.loff_start[1];
        a.esz = 2;
        a.most = nullptr;
    } else {
        a.n = p.ip.ncells;
        a.v[0] = w.cellcnt;
        a.v[1] = w.cellcnt + (a.n + 1);
        a.out[0] = w.blob + p.ip.off_start[0];
        a.out[1] = w.blob + p.ip.off_start[1];
        a.esz = p.ip.start32 ? 4 : 2;
        a.most = &w.plan->most;
    }
    a.tiles = w.tiles;
    a.n_tiles = a.n / kScanTile + 1; // covers position n itself
    return true;
}

__device__ inline unsigned block_sum_scan(unsigned x, unsigned *s_wave)
{
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = x;
    __syncthreads();
    unsigned t = 0;
    for (int w = 0; w < kScanThreads / 64; ++w) t += s_wave[w];
    __syncthreads();
    return t;
}

// grid (tiles at most, 2)
template <int LISTS>
__global__ __launch_bounds__(kScanThreads) void scan_tiles_kernel(BuildWs w)
{
    __shared__ unsigned s_wave[kScanThreads / 64];
    ScanArgs            a;
    if (!scan_args<LISTS>(w, a) || (int)blockIdx.x >= a.n_tiles) return;
    const int       c = blockIdx.y, k0 = blockIdx.x * kScanTile + (int)threadIdx.x * kScanPer;
    const unsigned *v = c ? a.v[1] : a.v[0]; // (a select, not an index into the local struct: that would put the struct into LDS)
    unsigned        sum = 0, most = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        const unsigned x = k0 + j < a.n ? v[k0 + j] : 0u;
        sum += x;
        most = max(most, x);
    }
    const unsigned t = block_sum_scan(sum, s_wave);
    if (threadIdx.x == 0) a.tiles[c * a.n_tiles + blockIdx.x] = t;
    if (a.most) {
        for (int o = 32; o > 0; o >>= 1) most = max(most, (unsigned)__shfl_xor((int)most, o));
        if ((threadIdx.x & 63) == 0 && most > __hip_atomic_load(a.most, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.most, most);
    }
}

template <int LISTS>
__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(BuildWs w)
{
    __shared__ unsigned s_wave[kScanThreads / 64];
    ScanArgs            a;
    if (!scan_args<LISTS>(w, a) || (int)blockIdx.x >= a.n_tiles) return;
    const int      c = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned      *v = c ? a.v[1] : a.v[0];
    unsigned char *out = c ? a.out[1] : a.out[0];
    unsigned       before = 0;
    for (int t = tid; t < tile; t += kScanThreads) before += a.tiles[c * a.n_tiles + t];
    before = block_sum_scan(before, s_wave);
    const int k0 = tile * kScanTile + tid * kScanPer;
    unsigned  x[kScanPer], sum = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        x[j] = k0 + j < a.n ? v[k0 + j] : 0u;
        sum += x[j];
    }
    unsigned incl = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned run = before + incl - sum;
    for (int k = 0; k < wave; ++k) run += s_wave[k];
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        if (k0 + j <= a.n) { // position n receives the total
            v[k0 + j] = run;
            store_entry(out, a.esz, (size_t)(k0 + j), run);
        }
        run += x[j];
    }
}

inline int scan_tiles_for(long n) { return (int)(n / kScanTile + 1); }

template <int LISTS>
void launch_scan(const BuildWs &w, int tiles_max, hipStream_t st)
{
    hipLaunchKernelGGL((scan_tiles_kernel<LISTS>), dim3(tiles_max, 2), dim3(kScanThreads), 0, st, w);
    hipLaunchKernelGGL((scan_apply_kernel<LISTS>), dim3(tiles_max, 2), dim3(kScanThreads), 0, st, w);
}

__global__ __launch_bounds__(256) void idx_fill_kernel(BuildWs w)
{
    const BuildArgs a = build_args(w);
    const int       i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_all) return;
    const int      c = i >= a.cnt[0] ? 1 : 0;
    const int      cell = a.cell_of[i];
    const unsigned pos = a.start[(size_t)c * (a.ncells + 1) + cell] + atomicAdd(&a.cursor[(size_t)c * a.ncells + cell], 1u);
    a.tmp[a.base[c] + pos] = i - a.base[c];
}

// a stable counting sort keeps the original order inside a cell: position = rank of the original index
__global__ __launch_bounds__(256) void idx_rank_kernel(BuildWs w)
{
    const BuildArgs a = build_args(w);
    const int       s = blockIdx.x * 256 + threadIdx.x;
    if (s >= a.n_all) return;
    const int       c = s >= a.cnt[0] ? 1 : 0;
    const int       j = a.tmp[s];
    const int       cell = a.cell_of[a.base[c] + j];
    const unsigned *st = a.start + (size_t)c * (a.ncells + 1);
    const int       lo = (int)st[cell], hi = (int)st[cell + 1];
    int             rank = 0;
    for (int k = lo; k < hi; ++k) rank += a.tmp[a.base[c] + k] < j ? 1 : 0;
    const int out = a.base[c] + lo + rank;
    reinterpret_cast<float2 *>(a.blob + a.off_pts)[out] = a.xyf[a.base[c] + j];
    store_entry(a.blob + a.off_oidx, a.esz, (size_t)out, (unsigned)j);
}

// entries per class of every candidate list lattice (blockIdx.y = candidate)
__global__ __launch_bounds__(256) void list_cand_kernel(BuildWs w)
{
    __shared__ unsigned long long s[4][2];
    DevPlan                      &p = *w.plan;
    if ((int)blockIdx.y >= p.nc) return;
    const ListGeom     g = p.geom[blockIdx.y];
    const int          i = blockIdx.x * 256 + threadIdx.x, n_ga = p.cnt[0], n_all = p.cnt[0] + p.cnt[1];
    unsigned long long c0 = 0, c1 = 0;
    if (i < n_all) {
        const float2 q = w.xyf[i];
        if ((q.x - q.x <= 0.0f) && (q.y - q.y <= 0.0f)) {
            long ax, bx, ay, by;
            halo_cells(g, (double)q.x, (double)q.y, ax, bx, ay, by);
            const unsigned long long k = (unsigned long long)((bx - ax + 1) * (by - ay + 1));
            if (i < n_ga)
                c0 = k;
            else
                c1 = k;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        c0 += __shfl_xor(c0, o);
        c1 += __shfl_xor(c1, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s[wave][0] = c0, s[wave][1] = c1;
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned long long v = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
        if (v) atomicAdd(&p.n_ent[blockIdx.y][threadIdx.x], v);
    }
}

// One wavefront: the first candidate, in the host's order, whose lists fit -- among those whose halo reaches list_min_halo
// if there is one (build_lists_host's partition), among all otherwise
__global__ __launch_bounds__(64) void list_plan_kernel(BuildWs w)
{
    DevPlan  &p = *w.plan;
    const int lane = threadIdx.x;
    int       pick = -1;
    ListPlan  lp;
    for (int tier = 0; tier < 2 && pick < 0; ++tier)
    for (int k0 = 0; k0 < p.nc && pick < 0; k0 += 64) {
        const int k = k0 + lane;
        bool      ok = false;
        if (k < p.nc) {
            const ListGeom &g = p.geom[k];
            ListCand        c;
            c.frac = p.frac[k];
            c.hs = g.hs;
            c.pad_m = g.pad_m;
            c.x0 = g.x0;
            c.y0 = g.y0;
            c.nx = g.nx;
            c.ny = g.ny;
            const unsigned long long n_ent[2] = {p.n_ent[k][0], p.n_ent[k][1]};
            ok = (tier == 1 || c.pad_m >= w.list_min_halo) && accept_list_core(c, n_ent, p.budget, p.ip.maxabs, p.margin_abs, lp);
        }
        const unsigned long long m = __ballot(ok);
        if (m) {
            const int first = __ffsll((long long)m) - 1;
            pick = k0 + first;
            if (lane == first) p.lp = lp;
        }
    }
    if (lane == 0) p.pick = pick;
}

struct ListArgs {
    const float2 *xyf;
    int           cnt[2], base[2], n_all;
    ListGeom      g;
    int           ncells;
    unsigned     *start;  // [2][ncells + 1]
    unsigned     *cursor; // [2][ncells]
    int          *ent;    // [n_ent0 + n_ent1] point (index within its class) of every entry, bucketed by cell
    int           lbase[2];
    unsigned char *lblob;
    unsigned      loff_pts, loff_start[2], loff_axis[2];
    int          *lidx;   // nullable (point-to-line): [entries] the point of every entry in its final, sorted place
};

__device__ inline bool list_args(const BuildWs &w, ListArgs &a)
{
    const DevPlan &p = *w.plan;
    if (p.pick < 0) return false;
    a.xyf = w.xyf;
    a.cnt[0] = p.cnt[0];
    a.cnt[1] = p.cnt[1];
    a.base[0] = 0;
    a.base[1] = p.cnt[0];
    a.n_all = p.cnt[0] + p.cnt[1];
    a.g = p.geom[p.pick];
    a.ncells = p.lp.ncells;
    a.start = w.lcnt;
    a.cursor = w.lcnt + 2 * (size_t)(a.ncells + 1);
    a.ent = w.ent;
    a.lbase[0] = p.lp.lbase[0];
    a.lbase[1] = p.lp.lbase[1];
    a.lblob = w.lblob;
    a.loff_pts = p.lp.loff_pts;
    for (int k = 0; k < 2; ++k) {
        a.loff_start[k] = p.lp.loff_start[k];
        a.loff_axis[k] = p.lp.loff_axis[k];
    }
    a.lidx = w.p2l_lidx;
    return true;
}

template <int FILL>
__global__ __launch_bounds__(256) void list_scatter_kernel(BuildWs w)
{
    ListArgs a;
    if (!list_args(w, a)) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_all) return;
    const float2 p = a.xyf[i];
    if (!((p.x - p.x <= 0.0f) && (p.y - p.y <= 0.0f))) return;
    const int c = i >= a.cnt[0] ? 1 : 0;
    long      ax, bx, ay, by;
    halo_cells(a.g, (double)p.x, (double)p.y, ax, bx, ay, by);
    for (long yy = ay; yy <= by; ++yy)
        for (long xx = ax; xx <= bx; ++xx) {
            const int cell = (int)(yy * a.g.nx + xx);
            if (FILL) {
                const unsigned pos = a.start[(size_t)c * (a.ncells + 1) + cell] + atomicAdd(&a.cursor[(size_t)c * a.ncells + cell], 1u);
                a.ent[a.lbase[c] + pos] = i - a.base[c];
            } else {
                atomicAdd(&a.start[(size_t)c * (a.ncells + 1) + cell], 1u);
            }
        }
}

// One wavefront per (cell, class): the list's ordering key by the densest-window metric, then the entries in
// (key, point) order -- what the host's two stable sorts leave.  Metric of a key: the largest number of entries
// whose key lies in [k_j - win, k_j] over the entries j (the host's sliding window over the sorted keys counts
// exactly that: float subtraction is monotone).
// list_sort_kernel<kSortWaves, kListStage, min wavefronts per SIMD>: wavefronts per workgroup (each takes (cell, class) pairs in
// turn) and entries of a cell a wavefront stages in LDS (kSmallCell and more: every list the kernel takes).  The build of a
// handle made by itself: 4 x 256 entries (16 KB), no register cap.  The build that must run BESIDE a registration workgroup
// (slam_icp::build_beside, the mapper's sliding target): 2 x 128 entries = 4 KB and at most 64 registers (44 against 23 us for
// the 10 k-point map when alone on the chip -- and no wait for a free CU when it is not).
constexpr int kSmallCell = 128; // lists up to this long: one wavefront, quadratic counting; longer: a workgroup that sorts
constexpr int kBigCell = 4096;  // entries a workgroup sorts in LDS (longer lists: quadratic through the cache, correct and slow)
constexpr int kBigCellBeside = 512; // ... of the variant that must fit beside a registration workgroup (4 KB of LDS: slam_icp::build_beside)

template <int kSortWaves, int kListStage, int kMinWaves>
__global__ __launch_bounds__(64 * kSortWaves, kMinWaves) void list_sort_kernel(BuildWs w)
{
    ListArgs a;
    if (!list_args(w, a)) return;
    __shared__ float s_xs[kSortWaves][kListStage], s_ys[kSortWaves][kListStage], s_ks[kSortWaves][kListStage];
    __shared__ int   s_js[kSortWaves][kListStage];
    const int        lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float           *s_x = s_xs[wave], *s_y = s_ys[wave], *s_k = s_ks[wave];
    int             *s_j = s_js[wave];
    const int        n_pairs = 2 * a.ncells, stride = (int)gridDim.x * kSortWaves;
    for (int pair = (int)blockIdx.x * kSortWaves + wave; pair < n_pairs; pair += stride) {
        const int       c = pair >= a.ncells ? 1 : 0, cell = pair - c * a.ncells;
        const unsigned *st = a.start + (size_t)c * (a.ncells + 1);
        const int       lo = (int)st[cell], n = (int)st[cell + 1] - lo;
        if (n <= 0 || n > kSmallCell) continue; // longer lists: list_sort_big_kernel
        const int    *ent = a.ent + a.lbase[c] + lo;
        const float2 *xy = a.xyf + a.base[c];
        const bool    staged = n <= kListStage;
        // (a wavefront writes and reads only its own stage: program order and a wave barrier are enough)
        if (staged) {
            for (int k = lane; k < n; k += 64) {
                const int    j = ent[k];
                const float2 p = xy[j];
                s_x[k] = p.x;
                s_y[k] = p.y;
                s_j[k] = j;
            }
            __builtin_amdgcn_wave_barrier();
        }
        const auto X = [&](int k) { return staged ? s_x[k] : xy[ent[k]].x; };
        const auto Y = [&](int k) { return staged ? s_y[k] : xy[ent[k]].y; };
        const auto J = [&](int k) { return staged ? s_j[k] : ent[k]; };
        // keys of direction `dir`, staged too when the cell is
        const auto stage_keys = [&](int dir) {
            if (staged) {
                __builtin_amdgcn_wave_barrier();
                for (int k = lane; k < n; k += 64) s_k[k] = list_key(dir, s_x[k], s_y[k]);
                __builtin_amdgcn_wave_barrier();
            }
        };
        const auto K = [&](int dir, int k) { return staged ? s_k[k] : list_key(dir, X(k), Y(k)); };

        float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
        for (int k = lane; k < n; k += 64) {
            mnx = fminf(mnx, X(k));
            mxx = fmaxf(mxx, X(k));
            mny = fminf(mny, Y(k));
            mxy = fmaxf(mxy, Y(k));
        }
        for (int o = 32; o > 0; o >>= 1) {
            mnx = fminf(mnx, __shfl_xor(mnx, o));
            mxx = fmaxf(mxx, __shfl_xor(mxx, o));
            mny = fminf(mny, __shfl_xor(mny, o));
            mxy = fmaxf(mxy, __shfl_xor(mxy, o));
        }
        int best_dir = (mxy - mny) > (mxx - mnx) ? 1 : 0;
        if (n >= 8) {
            unsigned  best_metric = 0xffffffffu;
            const int first = best_dir;
            for (int t = 0; t < 4; ++t) {
                const int   dir = t == 0 ? first : (t == 1 ? 1 - first : t);
                const float win = key_window(dir);
                unsigned    metric = 0;
                stage_keys(dir);
                for (int j = lane; j < n; j += 64) {
                    const float kj = K(dir, j);
                    unsigned    cnt = 0;
#pragma unroll 8
                    for (int i = 0; i < n; ++i) {
                        const float ki = K(dir, i);
                        cnt += (ki <= kj && !(kj - ki > win)) ? 1u : 0u;
                    }
                    metric = max(metric, cnt);
                }
                for (int o = 32; o > 0; o >>= 1) metric = max(metric, (unsigned)__shfl_xor((int)metric, o));
                if (dir >= 2) metric += metric / 4 + 1; // an axis key is cheaper and exact: prefer it when close
                if (metric < best_metric) {
                    best_metric = metric;
                    best_dir = dir;
                }
            }
        }
        if (lane == 0 && best_dir)
            atomicOr(reinterpret_cast<unsigned *>(a.lblob + a.loff_axis[c]) + (cell >> 4), (unsigned)best_dir << (2 * (cell & 15)));
        float2 *out = reinterpret_cast<float2 *>(a.lblob + a.loff_pts) + a.lbase[c] + lo;
        stage_keys(best_dir);
        for (int j = lane; j < n; j += 64) {
            const float kj = K(best_dir, j);
            const int   pj = J(j);
            int         rank = 0;
#pragma unroll 8
            for (int i = 0; i < n; ++i) {
                const float ki = K(best_dir, i);
                rank += (ki < kj || (ki == kj && J(i) < pj)) ? 1 : 0;
            }
            out[rank] = make_float2(X(j), Y(j));
            if (a.lidx) a.lidx[a.lbase[c] + lo + rank] = pj;
        }
        __builtin_amdgcn_wave_barrier(); // the stage is rewritten by the next pair
    }
}

// Long lists (a wall a metre from the sensor, seen by a thousand scans): one workgroup per (cell, class), bitonic
// sorts in LDS.  The metric of a key is read off its sorted keys -- for entry j the first i with
// fl(k_j - k_i) <= win by bisection (the host's sliding window) --, the final order is the sort by (key, point).
__device__ inline void bitonic_sort_lds(float *key, int *val, int N /* power of two */)
{
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < N / 2; t += blockDim.x) {
                const int i = ((t / j) * 2 * j) + (t % j), ixj = i + j; // partner pairs of this step
                const bool  up = (i & k) == 0;
                const float ka = key[i], kb = key[ixj];
                const int   va = val[i], vb = val[ixj];
                const bool  gt = ka > kb || (ka == kb && va > vb);
                if (gt == up) {
                    key[i] = kb, key[ixj] = ka;
                    val[i] = vb, val[ixj] = va;
                }
            }
            __syncthreads();
        }
}

// grid (workgroups, 2): a workgroup takes the cells blockIdx.x, blockIdx.x + gridDim.x, ... of class blockIdx.y
template <int kBigCell>
__global__ __launch_bounds__(256) void list_sort_big_kernel(BuildWs w)
{
    __shared__ float s_key[kBigCell];
    __shared__ int   s_val[kBigCell];
    __shared__ unsigned s_red[4];
    __shared__ float s_ext[4][4];
    ListArgs a;
    if (!list_args(w, a)) return;
    const int       c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned *st = a.start + (size_t)c * (a.ncells + 1);
    for (int cell = blockIdx.x; cell < a.ncells; cell += gridDim.x) {
    const int       lo = (int)st[cell], n = (int)st[cell + 1] - lo;
    if (n <= kSmallCell) continue; // (uniform over the workgroup)
    __syncthreads(); // the cell before is through with the shared arrays
    const int    *ent = a.ent + a.lbase[c] + lo;
    const float2 *xy = a.xyf + a.base[c];
    float2       *out = reinterpret_cast<float2 *>(a.lblob + a.loff_pts) + a.lbase[c] + lo;
    // extent
    float mnx = FLT_MAX, mny = FLT_MAX, mxx = -FLT_MAX, mxy = -FLT_MAX;
    for (int k = tid; k < n; k += 256) {
        const float2 p = xy[ent[k]];
        mnx = fminf(mnx, p.x), mxx = fmaxf(mxx, p.x);
        mny = fminf(mny, p.y), mxy = fmaxf(mxy, p.y);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, o)), mxx = fmaxf(mxx, __shfl_xor(mxx, o));
        mny = fminf(mny, __shfl_xor(mny, o)), mxy = fmaxf(mxy, __shfl_xor(mxy, o));
    }
    if (lane == 0) s_ext[wave][0] = mnx, s_ext[wave][1] = mxx, s_ext[wave][2] = mny, s_ext[wave][3] = mxy;
    __syncthreads();
    mnx = fminf(fminf(s_ext[0][0], s_ext[1][0]), fminf(s_ext[2][0], s_ext[3][0]));
    mxx = fmaxf(fmaxf(s_ext[0][1], s_ext[1][1]), fmaxf(s_ext[2][1], s_ext[3][1]));
    mny = fminf(fminf(s_ext[0][2], s_ext[1][2]), fminf(s_ext[2][2], s_ext[3][2]));
    mxy = fmaxf(fmaxf(s_ext[0][3], s_ext[1][3]), fmaxf(s_ext[2][3], s_ext[3][3]));
    int        best_dir = (mxy - mny) > (mxx - mnx) ? 1 : 0;
    const bool lds = n <= kBigCell;
    int        N = 1;
    while (N < n) N <<= 1;
    unsigned  best_metric = 0xffffffffu;
    const int first = best_dir;
    for (int t = 0; t < 4; ++t) { // n >= 8 here
        const int   dir = t == 0 ? first : (t == 1 ? 1 - first : t);
        const float win = key_window(dir);
        unsigned    metric = 0;
        if (lds) {
            __syncthreads();
            for (int k = tid; k < N; k += 256) {
                const float2 p = k < n ? xy[ent[k]] : make_float2(0.f, 0.f);
                s_key[k] = k < n ? list_key(dir, p.x, p.y) : FLT_MAX;
                s_val[k] = k;
            }
            __syncthreads();
            bitonic_sort_lds(s_key, s_val, N);
            for (int j = tid; j < n; j += 256) {
                const float kj = s_key[j];
                int         a0 = 0, b0 = j; // first i in [0, j] with !(kj - key[i] > win)
                while (a0 < b0) {
                    const int mid = (a0 + b0) >> 1;
                    if (kj - s_key[mid] > win)
                        a0 = mid + 1;
                    else
                        b0 = mid;
                }
                metric = max(metric, (unsigned)(j - a0 + 1));
            }
        } else {
            for (int j = tid; j < n; j += 256) {
                const float2 pj = xy[ent[j]];
                const float  kj = list_key(dir, pj.x, pj.y);
                unsigned     cnt = 0;
                for (int i = 0; i < n; ++i) {
                    const float2 pi = xy[ent[i]];
                    const float  ki = list_key(dir, pi.x, pi.y);
                    cnt += (ki <= kj && !(kj - ki > win)) ? 1u : 0u;
                }
                metric = max(metric, cnt);
            }
        }
        for (int o = 32; o > 0; o >>= 1) metric = max(metric, (unsigned)__shfl_xor((int)metric, o));
        __syncthreads();
        if (lane == 0) s_red[wave] = metric;
        __syncthreads();
        metric = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
        if (dir >= 2) metric += metric / 4 + 1; // an axis key is cheaper and exact: prefer it when close
        if (metric < best_metric) {
            best_metric = metric;
            best_dir = dir;
        }
    }
    if (tid == 0 && best_dir)
        atomicOr(reinterpret_cast<unsigned *>(a.lblob + a.loff_axis[c]) + (cell >> 4), (unsigned)best_dir << (2 * (cell & 15)));
    if (lds) {
        __syncthreads();
        for (int k = tid; k < N; k += 256) {
            const int    j = k < n ? ent[k] : 0x7fffffff;
            const float2 p = k < n ? xy[j] : make_float2(0.f, 0.f);
            s_key[k] = k < n ? list_key(best_dir, p.x, p.y) : FLT_MAX;
            s_val[k] = j; // ties by point: the order of the host's stable sort
        }
        __syncthreads();
        bitonic_sort_lds(s_key, s_val, N);
        for (int k = tid; k < n; k += 256) {
            out[k] = xy[s_val[k]];
            if (a.lidx) a.lidx[a.lbase[c] + lo + k] = s_val[k];
        }
    } else {
        for (int j = tid; j < n; j += 256) {
            const int    pj = ent[j];
            const float2 P = xy[pj];
            const float  kj = list_key(best_dir, P.x, P.y);
            int          rank = 0;
            for (int i = 0; i < n; ++i) {
                const int    pi = ent[i];
                const float2 Q = xy[pi];
                const float  ki = list_key(best_dir, Q.x, Q.y);
                rank += (ki < kj || (ki == kj && pi < pj)) ? 1 : 0;
            }
            out[rank] = P;
            if (a.lidx) a.lidx[a.lbase[c] + lo + rank] = pj;
        }
    }
    } // cell
}

// ---- point-to-line builds (SLAM_ICP_P2L; icpPointToPlane.cpp:55-77 knows no classes, :279-349 the normals)

// the caller's two classes into one array, GA then NGA, and {0, n} as the build's device-side counts
__global__ __launch_bounds__(256) void p2l_merge_kernel(BuildWs w)
{
    const int n0 = w.p2l_src_cnt ? w.p2l_src_cnt[0] : w.p2l_host_cnt[0], n1 = w.p2l_src_cnt ? w.p2l_src_cnt[1] : w.p2l_host_cnt[1];
    const int c0 = min(max(n0, 0), w.cap[1]), c1 = min(max(n1, 0), w.cap[1] - c0);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) w.p2l_cnt[0] = 0, w.p2l_cnt[1] = c0 + c1;
    if (i >= c0 + c1) return;
    const double2 *g = reinterpret_cast<const double2 *>(w.p2l_src[0]), *ng = reinterpret_cast<const double2 *>(w.p2l_src[1]);
    w.p2l_all[i] = i < c0 ? g[i] : ng[i - c0];
}

// the view the normals' device functions take, from the plan as it lies in device memory
__device__ inline ModelView p2l_view(const BuildWs &w)
{
    const DevPlan &p = *w.plan;
    ModelView      mv;
    memset(&mv, 0, sizeof mv);
    view_index(mv, p.ip);
    mv.blob = w.blob;
    mv.normals = w.normals;
    if (p.pick >= 0) {
        view_lists(mv, p.lp);
        mv.lblob = w.lblob;
    }
    return mv;
}

// normal_of_model_point for every point of the index just built (one thread per position of the sorted array)
template <int K>
__global__ __launch_bounds__(256) void p2l_normals_kernel(BuildWs w)
{
    const DevPlan  &p = *w.plan;
    const ModelView mv = p2l_view(w);
    const int       n = p.cnt[1], pos = blockIdx.x * 256 + threadIdx.x;
    if (p.cnt[0] + p.cnt[1] < 5) return; // (the handle will be refused at finish)
    if (p.ip.start32)
        normal_of_model_point<K, uint32_t>(mv, n, pos, w.normals);
    else
        normal_of_model_point<K, uint16_t>(mv, n, pos, w.normals);
}

// A normal per halo-list ENTRY, so that a list sweep's neighbour has its normal one load away: the list sorts leave the model
// point of every entry (class 1 holds the whole model: its index within the class is its index into the normals).  Exact
// duplicates sit in a list side by side at the same distance from any query: the sweep calls that a tie and leaves the query to
// the exact search, which takes the lowest index -- an entry's own normal is never used where another's would be right.
__global__ __launch_bounds__(256) void p2l_list_normals_kernel(BuildWs w)
{
    const DevPlan &p = *w.plan;
    if (p.pick < 0 || p.cnt[0] + p.cnt[1] < 5) return;
    const int n_ent = min((int)(p.lp.n_ent[0] + p.lp.n_ent[1]), w.lnormals_cap), e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_ent) return;
    w.lnormals[e] = reinterpret_cast<const double2 *>(w.normals)[w.p2l_lidx[e]];
}

template <int K>
void launch_p2l_normals(const BuildWs &w, int cap_all, hipStream_t st)
{
    hipLaunchKernelGGL((p2l_normals_kernel<K>), dim3((cap_all + 255) / 256), dim3(256), 0, st, w);
}

double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

constexpr int kSortBigGrid = 512; // workgroups per class of list_sort_big_kernel (cells with long lists are few)

} // namespace

// A build that has been enqueued and not yet adopted: its workspace (still in use by the device), the plan's pinned copy
// and the event behind it
struct slam_icp_pending {
    std::vector<void *> blocks;   // pool blocks of the build
    DevPlan            *h_plan = nullptr; // pinned
    hipEvent_t          done = nullptr;
    hipStream_t         st = nullptr;
    bool                recorded = false;  // `done` has been recorded behind the build
    size_t              blob_cap = 0, lblob_cap = 0;
    std::chrono::steady_clock::time_point t_begin;
};

namespace {

void *ws_get(slam_icp_pending *pb, size_t bytes)
{
    void *p = pool_alloc(bytes);
    if (p) pb->blocks.push_back(p);
    return p;
}

// gives the workspace back; wait = the device may still be using it
void drop_pending(slam_icp *h, bool wait)
{
    slam_icp_pending *pb = h->pending;
    if (!pb) return;
    if (wait) {
        if (pb->recorded)
            (void)hipEventSynchronize(pb->done);
        else
            (void)hipStreamSynchronize(pb->st); // a begin that failed half-way: whatever it enqueued
        (void)hipGetLastError();
    }
    for (void *p : pb->blocks) pool_free(p);
    if (pb->h_plan) pinned_block_put(pb->h_plan);
    if (pb->done) (void)hipEventDestroy(pb->done);
    delete pb;
    h->pending = nullptr;
}

int build_begin_device(slam_icp *h, const double *m_ga, int cap_ga, const double *m_nga, int cap_nga, const int *d_cnt,
                       bool on_device, hipStream_t st)
{
    const unsigned lds_total = device_lds_total();
    SLAM_REQUIRE(lds_total, SLAM_E_HIP, "the device does not report its LDS size");
    slam_icp_pending *pb = new (std::nothrow) slam_icp_pending();
    SLAM_REQUIRE(pb, SLAM_E_NOMEM, "out of host memory");
    h->pending = pb;
    pb->st = st;
    pb->t_begin = std::chrono::steady_clock::now();
    const int  cap_all = cap_ga + cap_nga;
    const int  pblocks = std::max(1, (cap_all + 255) / 256);
    const bool want_lists = h->sweep == 2 || h->two_phase;
    h->have_lists = false;

    BuildWs w;
    memset(&w, 0, sizeof w);
    w.d_cnt = d_cnt;
    w.host_cnt[0] = cap_ga;
    w.host_cnt[1] = cap_nga;
    w.cap[0] = cap_ga;
    w.cap[1] = cap_nga;
    w.cell_size = h->prm.cell_size;
    w.force_global = h->prm.force_global;
    w.want_lists = want_lists ? 1 : 0;
    w.list_min_halo = list_min_halo_of(h->prm);
    w.lds_total = lds_total;
    w.m[0] = m_ga;
    w.m[1] = m_nga;
    if (!on_device) { // host arrays (exact counts): one block in HBM
        double *d_in = static_cast<double *>(ws_get(pb, 16 * (size_t)std::max(cap_all, 1)));
        if (!d_in) return SLAM_E_NOMEM;
        if (cap_ga) SLAM_HIP(hipMemcpyAsync(d_in, m_ga, 16 * (size_t)cap_ga, hipMemcpyHostToDevice, st));
        if (cap_nga) SLAM_HIP(hipMemcpyAsync(d_in + 2 * (size_t)cap_ga, m_nga, 16 * (size_t)cap_nga, hipMemcpyHostToDevice, st));
        w.m[0] = d_in;
        w.m[1] = d_in + 2 * (size_t)cap_ga;
    }
    const bool p2l = h->prm.mode == SLAM_ICP_P2L;
    int        normals_k = 0;
    if (p2l) {
        // icpPointToPlane.cpp:55-77 knows no classes: the two arrays become ONE class (class 1), GA then NGA -- the order of the
        // reference's M_normal -- so that a query is searched once and a neighbour's original index is its index into the normals
        normals_k = h->prm.normals_k > 0 ? h->prm.normals_k : 10;
        SLAM_REQUIRE(normals_k >= 2 && normals_k <= kMaxK, SLAM_E_INVALID, "normals_k must be 2..%d (got %d)", kMaxK, normals_k);
        w.p2l_src[0] = w.m[0];
        w.p2l_src[1] = w.m[1];
        w.p2l_src_cnt = d_cnt;
        w.p2l_host_cnt[0] = cap_ga;
        w.p2l_host_cnt[1] = cap_nga;
        w.p2l_all = static_cast<double2 *>(ws_get(pb, 16 * (size_t)std::max(cap_all, 1)));
        w.p2l_cnt = static_cast<int *>(ws_get(pb, 2 * sizeof(int)));
        h->d_normals = static_cast<double *>(pool_alloc(16 * (size_t)std::max(cap_all, 1)));
        if (!w.p2l_all || !w.p2l_cnt || !h->d_normals) return SLAM_E_NOMEM;
        w.normals = h->d_normals;
        w.m[0] = nullptr;
        w.m[1] = reinterpret_cast<const double *>(w.p2l_all);
        w.cap[0] = w.host_cnt[0] = 0;
        w.cap[1] = w.host_cnt[1] = cap_all;
        w.d_cnt = w.p2l_cnt;
    }
    // ---- what the build can need at most
    const long   cb = cells_bound(cap_all, lds_total);
    const long   lcb = want_lists ? 60000 : 0;
    const size_t cnt_words = 2 * (size_t)(cb + 1) + 2 * (size_t)cb + 8;
    const size_t lcnt_words = want_lists ? 2 * (size_t)(lcb + 1) + 2 * (size_t)lcb + 8 : 0;
    const size_t plan_bytes = (sizeof(DevPlan) + 255) & ~(size_t)255;
    const size_t zero_bytes = plan_bytes + 4 * cnt_words + 4 * lcnt_words;
    pb->blob_cap = (size_t)align16(8u * (unsigned)cap_all) + 2 * (size_t)align16(4u * (unsigned)(cb + 1)) + align16(4u * (unsigned)cap_all) + 64;
    pb->lblob_cap = want_lists ? (size_t)lds_total : 0;
    unsigned char *zero = static_cast<unsigned char *>(ws_get(pb, zero_bytes));
    w.rows = static_cast<double *>(ws_get(pb, 64 * (size_t)pblocks));
    w.n_rows = pblocks;
    w.xyf = static_cast<float2 *>(ws_get(pb, 8 * (size_t)std::max(cap_all, 1)));
    w.cell_of = static_cast<int *>(ws_get(pb, 4 * (size_t)std::max(cap_all, 1)));
    w.tmp = static_cast<int *>(ws_get(pb, 4 * (size_t)std::max(cap_all, 1)));
    const int tiles_idx = scan_tiles_for(cb), tiles_lst = scan_tiles_for(lcb);
    w.tiles = static_cast<unsigned *>(ws_get(pb, 4 * 2 * (size_t)std::max(tiles_idx, tiles_lst)));
    w.ent = want_lists ? static_cast<int *>(ws_get(pb, 4 * 2 * (size_t)65536)) : nullptr;
    h->d_blob = pool_alloc(pb->blob_cap);
    h->d_lblob = want_lists ? pool_alloc(pb->lblob_cap) : nullptr;
    pb->h_plan = static_cast<DevPlan *>(pinned_block_get(sizeof(DevPlan)));
    if (!zero || !w.rows || !w.xyf || !w.cell_of || !w.tmp || !w.tiles || (want_lists && (!w.ent || !h->d_lblob)) || !h->d_blob || !pb->h_plan)
        return SLAM_E_NOMEM;
    w.plan = reinterpret_cast<DevPlan *>(zero);
    w.cellcnt = reinterpret_cast<unsigned *>(zero + plan_bytes);
    w.lcnt = w.cellcnt + cnt_words;
    w.blob = static_cast<unsigned char *>(h->d_blob);
    w.lblob = static_cast<unsigned char *>(h->d_lblob);
    SLAM_HIP(hipEventCreateWithFlags(&pb->done, hipEventDisableTiming));
    if (p2l && want_lists) { // a normal per halo-list entry: as many as a list blob can hold at most
        w.lnormals_cap = (int)(pb->lblob_cap / 8);
        h->d_lnormals = static_cast<double *>(pool_alloc(16 * (size_t)w.lnormals_cap));
        if (!h->d_lnormals) return SLAM_E_NOMEM;
        w.lnormals = reinterpret_cast<double2 *>(h->d_lnormals);
        w.p2l_lidx = static_cast<int *>(ws_get(pb, 4 * (size_t)w.lnormals_cap));
        if (!w.p2l_lidx) return SLAM_E_NOMEM;
    }
    if (p2l) hipLaunchKernelGGL(p2l_merge_kernel, dim3(pblocks), dim3(256), 0, st, w);

    SLAM_HIP(hipMemsetAsync(zero, 0, zero_bytes, st));
    SLAM_HIP(hipMemsetAsync(h->d_blob, 0, pb->blob_cap, st)); // the padding between the arrays is part of the blob
    if (want_lists) SLAM_HIP(hipMemsetAsync(h->d_lblob, 0, pb->lblob_cap, st));
    hipLaunchKernelGGL(idx_bbox_kernel, dim3(pblocks), dim3(256), 0, st, w);
    hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(64), 0, st, w);
    hipLaunchKernelGGL(idx_count_kernel, dim3(pblocks), dim3(256), 0, st, w);
    if (want_lists) hipLaunchKernelGGL(list_cand_kernel, dim3(pblocks, kMaxCand), dim3(256), 0, st, w);
    launch_scan<0>(w, tiles_idx, st);
    hipLaunchKernelGGL(idx_fill_kernel, dim3(pblocks), dim3(256), 0, st, w);
    hipLaunchKernelGGL(idx_rank_kernel, dim3(pblocks), dim3(256), 0, st, w);
    if (want_lists) {
        hipLaunchKernelGGL(list_plan_kernel, dim3(1), dim3(64), 0, st, w);
        hipLaunchKernelGGL((list_scatter_kernel<0>), dim3(pblocks), dim3(256), 0, st, w);
        launch_scan<1>(w, tiles_lst, st);
        hipLaunchKernelGGL((list_scatter_kernel<1>), dim3(pblocks), dim3(256), 0, st, w);
        if (h->build_beside)
            hipLaunchKernelGGL((list_sort_kernel<2, 128, 8>), dim3(1024), dim3(64 * 2), 0, st, w);
        else
            hipLaunchKernelGGL((list_sort_kernel<4, 256, 1>), dim3(1024), dim3(64 * 4), 0, st, w);
        // (32 KB of LDS waits for a CU to come free even when no list is long; a caller whose lists are short by construction --
        // the mapper's thinned window -- takes the 4 KB variant, which runs beside a registration workgroup)
        if (h->build_beside)
            hipLaunchKernelGGL((list_sort_big_kernel<kBigCellBeside>), dim3(kSortBigGrid, 2), dim3(256), 0, st, w);
        else
            hipLaunchKernelGGL((list_sort_big_kernel<kBigCell>), dim3(kSortBigGrid, 2), dim3(256), 0, st, w);
    }
    if (p2l) { // the normals of icpPointToPlane.cpp:340-349 from the index that now exists, and one per list entry
        switch (normals_k) {
#define SLAM_K(KK) case KK: launch_p2l_normals<KK>(w, cap_all, st); break;
            SLAM_K(2) SLAM_K(3) SLAM_K(4) SLAM_K(5) SLAM_K(6) SLAM_K(7) SLAM_K(8) SLAM_K(9) SLAM_K(10) SLAM_K(11)
            SLAM_K(12) SLAM_K(13) SLAM_K(14) SLAM_K(15) SLAM_K(16)
#undef SLAM_K
        }
        if (want_lists) hipLaunchKernelGGL(p2l_list_normals_kernel, dim3((w.lnormals_cap + 255) / 256), dim3(256), 0, st, w);
    }
    SLAM_HIP(hipGetLastError());
    SLAM_HIP(hipMemcpyAsync(pb->h_plan, w.plan, sizeof(DevPlan), hipMemcpyDeviceToHost, st));
    SLAM_HIP(hipEventRecord(pb->done, st));
    pb->recorded = true;
    h->build_ms[0] = ms_since(pb->t_begin);
    return SLAM_OK;
}

} // namespace

namespace slam {
namespace icp {

int build_index_begin(slam_icp *h, const double *m_ga, int cap_ga, const double *m_nga, int cap_nga, const int *d_cnt, bool on_device,
                      hipStream_t st)
{
    const int rc = build_begin_device(h, m_ga, cap_ga, m_nga, cap_nga, d_cnt, on_device, st);
    if (rc != SLAM_OK) { // nothing of a failed begin stays behind (the kernels already enqueued write into blocks the device is done with first)
        drop_pending(h, true);
        release_index(h);
    }
    return rc;
}

bool build_index_ready(slam_icp *h)
{
    if (!h->pending) return true;
    const hipError_t e = hipEventQuery(h->pending->done);
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess;
}

int build_index_finish(slam_icp *h)
{
    slam_icp_pending *pb = h->pending;
    if (!pb) return SLAM_OK;
    const auto t_wait = std::chrono::steady_clock::now();
    const hipError_t e = hipEventSynchronize(pb->done); // the build's one wait
    if (e != hipSuccess) {
        (void)hipGetLastError();
        drop_pending(h, false);
        release_index(h);
        return hip_fail(e, "hipEventSynchronize(index build)", __FILE__, __LINE__);
    }
    h->build_ms[1] = ms_since(t_wait);
    const DevPlan &p = *pb->h_plan;
    int            rc = SLAM_OK;
    if (p.cnt[0] + p.cnt[1] < 5) { // icp.cpp:38-43 (counts that only the device knew)
        set_error("LIBICP works only with at least 5 model points (got %d)", p.cnt[0] + p.cnt[1]);
        rc = SLAM_E_TOO_FEW_MODEL_POINTS;
    } else if ((size_t)p.ip.blob_bytes > pb->blob_cap || (p.pick >= 0 && (size_t)p.lp.lblob_bytes > pb->lblob_cap)) {
        set_error("index build: the plan outgrew its reservation (%u of %zu bytes)", p.ip.blob_bytes, pb->blob_cap);
        rc = SLAM_E_HIP;
    }
    if (rc == SLAM_OK) {
        adopt_index_plan(h, p.ip);
        h->mv.blob = static_cast<const unsigned char *>(h->d_blob);
        if (p.pick >= 0) {
            adopt_list_plan(h, p.lp);
            h->mv.lblob = static_cast<const unsigned char *>(h->d_lblob);
            list_done(h);
        } else if (h->d_lblob) {
            pool_free(h->d_lblob);
            h->d_lblob = nullptr;
        }
        h->max_cell_points = (int)p.most;
        h->built_on_device = true;
        if (h->prm.mode == SLAM_ICP_P2L) { // (made by the build's last kernels: p2l_normals_kernel, p2l_list_normals_kernel)
            h->mv.normals = h->d_normals;
            h->mv.lnormals = p.pick >= 0 ? reinterpret_cast<const double2 *>(h->d_lnormals) : nullptr;
        }
    }
    drop_pending(h, false);
    if (rc != SLAM_OK) release_index(h);
    return rc;
}

int build_index(slam_icp *h, const double *m_ga, int n_ga, const double *m_nga, int n_nga, bool on_device)
{
    if (h->prm.build_on_host && !on_device && h->prm.mode != SLAM_ICP_P2L) return build_index_host(h, m_ga, n_ga, m_nga, n_nga);
    // (device arrays: complete when the call is made -- the build does not order itself behind the default stream: that
    // stream shares a hardware queue with whatever the application runs, and an event on it can sit behind a whole
    // registration launch)
    SLAM_TRY(build_index_begin(h, m_ga, n_ga, m_nga, n_nga, nullptr, on_device, build_stream()));
    return build_index_finish(h);
}

void release_index(slam_icp *h)
{
    drop_pending(h, true);
    if (h->d_blob) pool_free(h->d_blob);
    if (h->d_lblob) pool_free(h->d_lblob);
    h->d_blob = h->d_lblob = nullptr;
}

} // namespace icp
} // namespace slam
