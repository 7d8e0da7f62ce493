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

using namespace slam;
using namespace slam::icp;

namespace {

inline unsigned align16(unsigned v) { return (v + 15u) & ~15u; }

struct BBox {
    float  lo[2] = {FLT_MAX, FLT_MAX}, hi[2] = {-FLT_MAX, -FLT_MAX};
    double sum[2] = {0, 0};
    size_t nfin = 0;
};

// ------------------------------------------------------------------ planning (host; shared by both builds)

// Lattice, blob layout and the LDS decision from the model's extent.
int plan_index(slam_icp *h, int n_ga, int n_nga, BBox bb, unsigned *lds_total_out, float *maxabs_out)
{
    if (bb.nfin == 0) {
        bb.lo[0] = bb.lo[1] = 0.f;
        bb.hi[0] = bb.hi[1] = 1.f;
        bb.nfin = 1;
    }
    const float *lo = bb.lo, *hi = bb.hi;
    const int    n_all = n_ga + n_nga;
    const int    max_cls = std::max(n_ga, n_nga);

    int lds_cap = 0;
    int dev = 0;
    SLAM_HIP(hipGetDevice(&dev));
    SLAM_HIP(hipDeviceGetAttribute(&lds_cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
    const unsigned lds_total = std::min<unsigned>((unsigned)lds_cap, kLdsTotal);
    const unsigned scratch = kScratchBytes;

    // LDS budget for the two start arrays (u16 entries) after points + original indices
    const long fixed16 = (long)scratch + align16(8u * n_all) + align16(2u * n_all) + 64;
    long       cells_lds = ((long)lds_total - fixed16) / (2 * 2) - 1;
    bool       lds = !h->prm.force_global && max_cls <= 65535 && cells_lds >= 256;

    const float w = std::max(hi[0] - lo[0], 1e-3f), ht = std::max(hi[1] - lo[1], 1e-3f);
    const float maxabs = std::max(std::max(std::fabs(lo[0]), std::fabs(hi[0])),
                                  std::max(std::fabs(lo[1]), std::fabs(hi[1])));
    long budget = lds ? cells_lds : std::min<long>(std::max<long>(4L * n_all, 1024), 1L << 22);
    // target about two cells per point on wall-like maps; never more than the budget
    long want = std::min<long>(budget, std::max<long>(64, 2L * n_all));
    double hcell = h->prm.cell_size > 0 ? h->prm.cell_size : std::sqrt((double)w * ht / (double)want);
    hcell = std::max(hcell, (double)maxabs * 1.52587890625e-05 /* 2^-16 */);
    hcell = std::max(hcell, 1e-4);
    int nx, ny;
    for (;;) {
        nx = (int)std::floor(w / hcell) + 1;
        ny = (int)std::floor(ht / hcell) + 1;
        if ((long)nx * ny <= budget) break;
        hcell *= 1.05;
    }

    ModelView &mv = h->mv;
    memset(&mv, 0, sizeof mv);
    mv.lat.nx = nx;
    mv.lat.ny = ny;
    mv.lat.x0 = lo[0];
    mv.lat.y0 = lo[1];
    mv.lat.h = (float)hcell;
    mv.lat.inv_h = 1.0f / mv.lat.h;
    // the cell map floor(fl(fl(x-x0)*inv_h)) is monotone and off by at most ~3*2^-24*nx cells per evaluation,
    // i.e. ~6*2^-24*maxabs metres for a model point and a query together; 2^-19*maxabs covers that 5x
    // (and stays below h/8 by the choice of h above)
    mv.lat.margin = std::max(mv.lat.h * 0.0009765625f, maxabs * 1.9073486328125e-06f);
    mv.n_cls[0] = n_ga;
    mv.n_cls[1] = n_nga;
    mv.base[0] = 0;
    mv.base[1] = n_ga;
    mv.cx = bb.sum[0] / (double)bb.nfin;
    mv.cy = bb.sum[1] / (double)bb.nfin;

    h->start32 = !lds; // the HBM-resident index always uses 32-bit positions
    const unsigned esz = h->start32 ? 4u : 2u;
    const int      ncells = nx * ny;
    unsigned       o = 0;
    mv.off_pts = o;
    o = align16(o + 8u * (unsigned)n_all);
    mv.off_start[0] = o;
    o = align16(o + esz * (unsigned)(ncells + 1));
    mv.off_start[1] = o;
    o = align16(o + esz * (unsigned)(ncells + 1));
    mv.off_oidx = o;
    o = align16(o + esz * (unsigned)n_all);
    mv.blob_bytes = o;
    if (lds && scratch + o > lds_total) lds = false, h->start32 = false; // keeps u16 entries, read from HBM
    h->in_lds = lds;
    h->lds_bytes = lds ? scratch + o : scratch;
    *lds_total_out = lds_total;
    *maxabs_out = maxabs;
    return SLAM_OK;
}

// The list lattices tried, in order: the first whose lists fit LDS beside the scratch is built.
struct ListCand {
    double frac, hs, pad_m, x0, y0;
    long   nx, ny;
};

void list_candidates(const BBox &bb, float margin_abs, std::vector<ListCand> &out)
{
    for (double frac : {0.25, 0.125}) {
        for (double hs = 0.25; hs <= 64.0; hs *= 1.12) {
            ListCand c;
            c.frac = frac;
            c.hs = hs;
            c.pad_m = hs * frac; // halo in metres
            if (c.pad_m < 8.0 * margin_abs) continue;
            c.x0 = (double)bb.lo[0] - c.pad_m;
            c.y0 = (double)bb.lo[1] - c.pad_m;
            c.nx = (long)std::floor(((double)bb.hi[0] + c.pad_m - c.x0) / hs) + 1;
            c.ny = (long)std::floor(((double)bb.hi[1] + c.pad_m - c.y0) / hs) + 1;
            if (c.nx * c.ny > 60000) continue;
            out.push_back(c);
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

inline ListGeom geom_of(const ListCand &c)
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

// Whether candidate c with n_ent[] entries per class is the one to build; fills the list side of h->mv if so.
bool accept_list(slam_icp *h, const ListCand &c, const size_t n_ent[2], double budget, float maxabs, float margin_abs)
{
    if (n_ent[0] > 65535 || n_ent[1] > 65535) return false;
    const int    ncells = (int)(c.nx * c.ny);
    const size_t bytes = align16(8u * (unsigned)(n_ent[0] + n_ent[1])) + 2 * (size_t)align16(2u * (unsigned)(ncells + 1)) +
                         2 * (size_t)align16(4u * (unsigned)(ncells / 16 + 1));
    if ((double)bytes > budget) return false;
    ModelView &mv = h->mv;
    unsigned   o = 0;
    mv.loff_pts = o;
    o = align16(o + 8u * (unsigned)(n_ent[0] + n_ent[1]));
    for (int k = 0; k < 2; ++k) {
        mv.loff_start[k] = o;
        o = align16(o + 2u * (unsigned)(ncells + 1));
    }
    for (int k = 0; k < 2; ++k) {
        mv.loff_axis[k] = o;
        o = align16(o + 4u * (unsigned)(ncells / 16 + 1));
    }
    mv.lblob_bytes = o;
    mv.lbase[0] = 0;
    mv.lbase[1] = (int)n_ent[0];
    mv.llat.nx = (int)c.nx;
    mv.llat.ny = (int)c.ny;
    mv.llat.x0 = (float)c.x0;
    mv.llat.y0 = (float)c.y0;
    mv.llat.h = (float)c.hs;
    mv.llat.inv_h = 1.0f / mv.llat.h;
    mv.llat.margin = std::max(mv.llat.h * 0.0009765625f, margin_abs);
    mv.lpad = (float)c.frac;
    mv.lkeps = 8.0f * 2.0f * maxabs * 1.1920929e-07f; // 8 ulp of |x| + |y| <= 2 maxabs (query within the lattice)
    // a point within `cert` of a query lies within cert + (cell-map rounding) of the query's nominal cell
    const double cert = c.pad_m - 4.0 * (double)mv.llat.margin - 2.0 * std::fabs((double)mv.llat.x0 - c.x0) -
                        2.0 * std::fabs((double)mv.llat.y0 - c.y0);
    if (cert <= 0) return false;
    mv.cert2 = (float)(cert * cert * 0.999);
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

struct BuildArgs {
    const double *m[2];   // model classes, f64 xy (device)
    int           cnt[2], base[2], n_all;
    Lattice       lat;
    int           ncells;
    float2       *xyf;     // [n_all] the f32 model (icp.cpp:54,60), class order
    int          *cell_of; // [n_all]
    unsigned     *start;   // [2][ncells + 1] counts, then exclusive prefix
    unsigned     *cursor;  // [2][ncells]
    int          *tmp;     // [n_all] original indices bucketed by cell, unordered inside a cell
    unsigned char *blob;
    unsigned      off_pts, off_start[2], off_oidx;
    int           esz;     // bytes per start / oidx entry in the blob: 2 or 4
};

__device__ inline void store_entry(unsigned char *base, int esz, size_t i, unsigned v)
{
    if (esz == 2)
        reinterpret_cast<unsigned short *>(base)[i] = (unsigned short)v;
    else
        reinterpret_cast<unsigned *>(base)[i] = v;
}

// extent, sum and count of the finite points of a device-resident model: one row of partials per workgroup
// (fixed order: the host adds the rows in index order)
__global__ __launch_bounds__(256) void idx_bbox_kernel(const double *m_ga, int n_ga, const double *m_nga, int n_nga,
                                                       double *rows /* [blocks][8]: lo x,y hi x,y sum x,y n pad */)
{
    __shared__ double s[4][8];
    const int i = blockIdx.x * 256 + threadIdx.x, n_all = n_ga + n_nga;
    float     lx = FLT_MAX, ly = FLT_MAX, hx = -FLT_MAX, hy = -FLT_MAX;
    double    sx = 0, sy = 0, nf = 0;
    if (i < n_all) {
        const double *p = i < n_ga ? m_ga + 2 * (size_t)i : m_nga + 2 * (size_t)(i - n_ga);
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
        double *r = rows + 8 * (size_t)blockIdx.x;
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

// icp.cpp:54,60: the f32 copy of the model; cell of every point; points per cell
__global__ __launch_bounds__(256) void idx_count_kernel(BuildArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
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
// entries in the blob's start arrays.  Two launches over tiles of 8192 values: the tiles' totals, then every
// tile adds the totals before it (a few dozen values) and scans itself -- a single workgroup walking 80 k cells
// took 100 us, one CU's share of the bandwidth.
constexpr int kScanPer = 8, kScanTile = 1024 * kScanPer;

struct ScanArgs {
    unsigned      *v[2];    // counts of class 0 / 1, n + 1 values each
    unsigned char *out[2];  // the blob's start arrays
    unsigned      *tiles;   // [2][n_tiles] tile totals
    unsigned      *most;    // nullable: receives the largest count (points in the fullest cell)
    int            n, n_tiles, esz;
};

__device__ inline unsigned block_sum_1024(unsigned x, unsigned *s_wave)
{
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = x;
    __syncthreads();
    unsigned t = 0;
    for (int w = 0; w < 16; ++w) t += s_wave[w];
    __syncthreads();
    return t;
}

// grid (n_tiles, 2)
__global__ __launch_bounds__(1024) void scan_tiles_kernel(ScanArgs a)
{
    __shared__ unsigned s_wave[16];
    const int           c = blockIdx.y, k0 = blockIdx.x * kScanTile + (int)threadIdx.x * kScanPer;
    unsigned            sum = 0, most = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        const unsigned x = k0 + j < a.n ? a.v[c][k0 + j] : 0u;
        sum += x;
        most = max(most, x);
    }
    const unsigned t = block_sum_1024(sum, s_wave);
    if (threadIdx.x == 0) a.tiles[c * a.n_tiles + blockIdx.x] = t;
    if (a.most) {
        for (int o = 32; o > 0; o >>= 1) most = max(most, (unsigned)__shfl_xor((int)most, o));
        if ((threadIdx.x & 63) == 0 && most > __hip_atomic_load(a.most, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.most, most);
    }
}

__global__ __launch_bounds__(1024) void scan_apply_kernel(ScanArgs a)
{
    __shared__ unsigned s_wave[16];
    const int           c = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned            before = 0;
    for (int t = tid; t < tile; t += 1024) before += a.tiles[c * a.n_tiles + t];
    before = block_sum_1024(before, s_wave);
    const int k0 = tile * kScanTile + tid * kScanPer;
    unsigned  x[kScanPer], sum = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        x[j] = k0 + j < a.n ? a.v[c][k0 + j] : 0u;
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
    for (int w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        if (k0 + j <= a.n) { // position n receives the total
            a.v[c][k0 + j] = run;
            store_entry(a.out[c], a.esz, (size_t)(k0 + j), run);
        }
        run += x[j];
    }
}

int launch_scan(unsigned *v0, unsigned *v1, int n, unsigned char *out0, unsigned char *out1, int esz, unsigned *tiles,
                hipStream_t st, unsigned *most = nullptr)
{
    ScanArgs a;
    a.most = most;
    a.v[0] = v0;
    a.v[1] = v1;
    a.out[0] = out0;
    a.out[1] = out1;
    a.tiles = tiles;
    a.n = n;
    a.n_tiles = n / kScanTile + 1; // covers position n itself
    a.esz = esz;
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(a.n_tiles, 2), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(a.n_tiles, 2), dim3(1024), 0, st, a);
    return SLAM_OK;
}

inline size_t scan_tile_words(int n) { return 2 * (size_t)(n / kScanTile + 1); }

__global__ __launch_bounds__(256) void idx_fill_kernel(BuildArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_all) return;
    const int      c = i >= a.cnt[0] ? 1 : 0;
    const int      cell = a.cell_of[i];
    const unsigned pos = a.start[(size_t)c * (a.ncells + 1) + cell] + atomicAdd(&a.cursor[(size_t)c * a.ncells + cell], 1u);
    a.tmp[a.base[c] + pos] = i - a.base[c];
}

// a stable counting sort keeps the original order inside a cell: position = rank of the original index
__global__ __launch_bounds__(256) void idx_rank_kernel(BuildArgs a)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
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
__global__ __launch_bounds__(256) void list_cand_kernel(const float2 *xyf, int n_all, int n_ga, const ListGeom *cands,
                                                        unsigned long long *n_ent /* [cands][2] */)
{
    __shared__ unsigned long long s[4][2];
    const ListGeom     g = cands[blockIdx.y];
    const int          i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long c0 = 0, c1 = 0;
    if (i < n_all) {
        const float2 p = xyf[i];
        if ((p.x - p.x <= 0.0f) && (p.y - p.y <= 0.0f)) {
            long ax, bx, ay, by;
            halo_cells(g, (double)p.x, (double)p.y, ax, bx, ay, by);
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
        if (v) atomicAdd(&n_ent[2 * (size_t)blockIdx.y + threadIdx.x], v);
    }
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
};

template <int FILL>
__global__ __launch_bounds__(256) void list_scatter_kernel(ListArgs a)
{
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
constexpr int kListStage = 256; // entries of a cell staged in LDS per wavefront (more: read through the cache)
constexpr int kSortWaves = 4;   // wavefronts per workgroup, each taking (cell, class) pairs in turn
constexpr int kSmallCell = 128; // lists up to this long: one wavefront, quadratic counting; longer: a workgroup that sorts
constexpr int kBigCell = 4096;  // entries a workgroup sorts in LDS (longer lists: quadratic through the cache, correct and slow)

__global__ __launch_bounds__(64 * kSortWaves) void list_sort_kernel(ListArgs a)
{
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

__global__ __launch_bounds__(256) void list_sort_big_kernel(ListArgs a)
{
    __shared__ float s_key[kBigCell];
    __shared__ int   s_val[kBigCell];
    __shared__ unsigned s_red[4];
    const int       c = blockIdx.y, cell = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned *st = a.start + (size_t)c * (a.ncells + 1);
    const int       lo = (int)st[cell], n = (int)st[cell + 1] - lo;
    if (n <= kSmallCell) return;
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
    __shared__ float s_ext[4][4];
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
        for (int k = tid; k < n; k += 256) out[k] = xy[s_val[k]];
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
        }
    }
}

double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

struct Workspace { // pool blocks of one build, returned when it ends
    std::vector<void *> blocks;
    hipStream_t         st = nullptr; // the stream the build's kernels run on
    void *get(size_t bytes)
    {
        void *p = pool_alloc(bytes);
        if (p) blocks.push_back(p);
        return p;
    }
    ~Workspace()
    {
        // on EVERY way out, error returns included: kernels already enqueued may still write these blocks, and the pool
        // hands a freed block to the next caller (another thread's rebuild) without synchronising
        if (!blocks.empty()) {
            (void)hipStreamSynchronize(st);
            (void)hipGetLastError();
        }
        for (void *p : blocks) pool_free(p);
    }
};

struct EventGuard { // an event of one build, destroyed on every way out
    hipEvent_t ev = nullptr;
    ~EventGuard()
    {
        if (ev) (void)hipEventDestroy(ev);
    }
};

constexpr int kCandFirst = 24; // candidate pitches counted with the cell index; the rest only if none of them fits

int build_index_device(slam_icp *h, const double *m_ga, int n_ga, const double *m_nga, int n_nga, bool on_device)
{
    const auto   t_begin = std::chrono::steady_clock::now();
    const int    n_all = n_ga + n_nga;
    const int    cnt[2] = {n_ga, n_nga};
    hipStream_t  st = build_stream();
    Workspace    ws;
    ws.st = st;
    const int    pblocks = (n_all + 255) / 256;
    // (device arrays: complete when the call is made -- the build does not order itself behind the default stream: that
    // stream shares a hardware queue with whatever the application runs, and an event on it can sit behind a whole
    // registration launch)

    // ---- the model in HBM (f64, as the caller holds it) and its extent
    BBox          bb;
    const double *d_m[2] = {m_ga, m_nga};
    if (!on_device) {
        const double *src[2] = {m_ga, m_nga};
        bb = host_bbox(src, cnt);
        double *d_in = static_cast<double *>(ws.get(16 * (size_t)std::max(n_all, 1)));
        if (!d_in) return SLAM_E_NOMEM;
        if (n_ga) SLAM_HIP(hipMemcpyAsync(d_in, m_ga, 16 * (size_t)n_ga, hipMemcpyHostToDevice, st));
        if (n_nga) SLAM_HIP(hipMemcpyAsync(d_in + 2 * (size_t)n_ga, m_nga, 16 * (size_t)n_nga, hipMemcpyHostToDevice, st));
        d_m[0] = d_in;
        d_m[1] = d_in + 2 * (size_t)n_ga;
    } else {
        double *rows = static_cast<double *>(ws.get(64 * (size_t)pblocks));
        double *hr = static_cast<double *>(pinned_scratch(64 * (size_t)pblocks));
        if (!rows || !hr) return SLAM_E_NOMEM;
        hipLaunchKernelGGL(idx_bbox_kernel, dim3(pblocks), dim3(256), 0, st, m_ga, n_ga, m_nga, n_nga, rows);
        SLAM_HIP(hipMemcpyAsync(hr, rows, 64 * (size_t)pblocks, hipMemcpyDeviceToHost, st));
        SLAM_HIP(hipStreamSynchronize(st));
        for (int b = 0; b < pblocks; ++b) {
            const double *r = hr + 8 * (size_t)b;
            if (r[6] <= 0) continue;
            bb.lo[0] = std::min(bb.lo[0], (float)r[0]);
            bb.lo[1] = std::min(bb.lo[1], (float)r[1]);
            bb.hi[0] = std::max(bb.hi[0], (float)r[2]);
            bb.hi[1] = std::max(bb.hi[1], (float)r[3]);
            bb.sum[0] += r[4];
            bb.sum[1] += r[5];
            bb.nfin += (size_t)r[6];
        }
    }
    const size_t nfin = bb.nfin;
    unsigned     lds_total = 0;
    float        maxabs = 0;
    SLAM_TRY(plan_index(h, n_ga, n_nga, bb, &lds_total, &maxabs));
    if (bb.nfin == 0) {
        bb.lo[0] = bb.lo[1] = 0.f;
        bb.hi[0] = bb.hi[1] = 1.f;
    }
    ModelView &mv = h->mv;
    const int  ncells = mv.lat.nx * mv.lat.ny;

    // ---- list candidates (host arithmetic only); a model whose points alone overflow the budget has no lists
    h->have_lists = false;
    const double budget = (double)lds_total - (double)kScratchBytes - 64.0;
    const float  margin_abs = maxabs * 1.9073486328125e-06f; // 2^-19 * maxabs, as for the cell lattice
    std::vector<ListCand> cands;
    std::vector<ListGeom> geoms;
    if ((h->sweep == 2 || h->two_phase) && 8.0 * (double)nfin <= budget && nfin <= 2 * 65535) {
        list_candidates(bb, margin_abs, cands);
        geoms.resize(cands.size());
        for (size_t k = 0; k < cands.size(); ++k) geoms[k] = geom_of(cands[k]);
    }
    const int nc = (int)cands.size();
    h->build_ms[0] = ms_since(t_begin);

    // ---- one workspace block, counters first (one memset): cell counts + cursors, candidate entry counts,
    // list counts + cursors (sized for the largest candidate lattice)
    const auto   t_index = std::chrono::steady_clock::now();
    const size_t cnt_words = 2 * (size_t)(ncells + 1) + 2 * (size_t)ncells + 8;
    size_t       lcells_max = 0;
    for (const ListCand &c : cands) lcells_max = std::max(lcells_max, (size_t)(c.nx * c.ny));
    const size_t lcnt_words = nc ? 2 * (lcells_max + 1) + 2 * lcells_max + 8 : 0;
    const size_t zero_bytes = ((4 * cnt_words + 15) & ~(size_t)15) + 16 * (size_t)std::max(nc, 1) + 4 * lcnt_words + 16;
    unsigned char *zero = static_cast<unsigned char *>(ws.get(zero_bytes));
    h->d_blob = pool_alloc(mv.blob_bytes);
    if (!zero || !h->d_blob) return SLAM_E_NOMEM;
    mv.blob = static_cast<const unsigned char *>(h->d_blob);
    BuildArgs a;
    a.m[0] = d_m[0];
    a.m[1] = d_m[1];
    a.cnt[0] = n_ga;
    a.cnt[1] = n_nga;
    a.base[0] = 0;
    a.base[1] = n_ga;
    a.n_all = n_all;
    a.lat = mv.lat;
    a.ncells = ncells;
    a.xyf = static_cast<float2 *>(ws.get(8 * (size_t)n_all));
    a.cell_of = static_cast<int *>(ws.get(4 * (size_t)n_all));
    a.tmp = static_cast<int *>(ws.get(4 * (size_t)n_all));
    if (!a.xyf || !a.cell_of || !a.tmp) return SLAM_E_NOMEM;
    unsigned *d_tiles = static_cast<unsigned *>(ws.get(4 * std::max(scan_tile_words(ncells), scan_tile_words((int)lcells_max))));
    if (!d_tiles) return SLAM_E_NOMEM;
    a.start = reinterpret_cast<unsigned *>(zero);
    a.cursor = a.start + 2 * (size_t)(ncells + 1);
    unsigned long long *d_ent = reinterpret_cast<unsigned long long *>(zero + ((4 * cnt_words + 15) & ~(size_t)15));
    unsigned           *d_lcnt = reinterpret_cast<unsigned *>(d_ent + 2 * (size_t)std::max(nc, 1));
    unsigned           *d_most = reinterpret_cast<unsigned *>(zero + zero_bytes - 16); // points in the fullest cell
    a.blob = static_cast<unsigned char *>(h->d_blob);
    a.off_pts = mv.off_pts;
    a.off_start[0] = mv.off_start[0];
    a.off_start[1] = mv.off_start[1];
    a.off_oidx = mv.off_oidx;
    a.esz = h->start32 ? 4 : 2;
    SLAM_HIP(hipMemsetAsync(zero, 0, zero_bytes, st));
    SLAM_HIP(hipMemsetAsync(h->d_blob, 0, mv.blob_bytes, st)); // the padding between the arrays is part of the blob
    hipLaunchKernelGGL(idx_count_kernel, dim3(pblocks), dim3(256), 0, st, a);
    // the first candidates' entry counts travel back while the cell index is finished
    ListGeom           *d_geom = nullptr;
    unsigned long long *h_ent = nullptr;
    EventGuard          ent_done;
    hipEvent_t         &ev_ent = ent_done.ev;
    const int           nc_first = std::min(nc, kCandFirst);
    if (nc) {
        d_geom = static_cast<ListGeom *>(ws.get(sizeof(ListGeom) * (size_t)nc));
        h_ent = static_cast<unsigned long long *>(pinned_scratch(16 * (size_t)nc));
        if (!d_geom || !h_ent) return SLAM_E_NOMEM;
        SLAM_HIP(hipMemcpyAsync(d_geom, geoms.data(), sizeof(ListGeom) * (size_t)nc, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(list_cand_kernel, dim3(pblocks, nc_first), dim3(256), 0, st, a.xyf, n_all, n_ga, d_geom, d_ent);
        SLAM_HIP(hipMemcpyAsync(h_ent, d_ent, 16 * (size_t)nc_first, hipMemcpyDeviceToHost, st));
        SLAM_HIP(hipEventCreateWithFlags(&ev_ent, hipEventDisableTiming));
        SLAM_HIP(hipEventRecord(ev_ent, st));
    }
    SLAM_TRY(launch_scan(a.start, a.start + (ncells + 1), ncells, a.blob + a.off_start[0], a.blob + a.off_start[1], a.esz, d_tiles, st, d_most));
    hipLaunchKernelGGL(idx_fill_kernel, dim3(pblocks), dim3(256), 0, st, a);
    hipLaunchKernelGGL(idx_rank_kernel, dim3(pblocks), dim3(256), 0, st, a);
    SLAM_HIP(hipGetLastError());
    h->build_ms[1] = ms_since(t_index);

    // ---- halo lists
    if (nc) {
        const auto t_plan = std::chrono::steady_clock::now();
        SLAM_HIP(hipEventSynchronize(ev_ent)); // the build's one read-back
        int    pick = -1;
        size_t n_ent[2] = {0, 0};
        auto   choose = [&](int k0, int k1) {
            for (int k = k0; k < k1 && pick < 0; ++k) {
                n_ent[0] = (size_t)h_ent[2 * (size_t)k];
                n_ent[1] = (size_t)h_ent[2 * (size_t)k + 1];
                if (accept_list(h, cands[k], n_ent, budget, maxabs, margin_abs)) pick = k;
            }
        };
        choose(0, nc_first);
        if (pick < 0 && nc > nc_first) { // rare: a second round for the coarser pitches
            hipLaunchKernelGGL(list_cand_kernel, dim3(pblocks, nc - nc_first), dim3(256), 0, st, a.xyf, n_all, n_ga,
                               d_geom + nc_first, d_ent + 2 * (size_t)nc_first);
            SLAM_HIP(hipMemcpyAsync(h_ent + 2 * (size_t)nc_first, d_ent + 2 * (size_t)nc_first, 16 * (size_t)(nc - nc_first),
                                    hipMemcpyDeviceToHost, st));
            SLAM_HIP(hipStreamSynchronize(st));
            choose(nc_first, nc);
        }
        h->build_ms[2] = ms_since(t_plan);
        if (pick >= 0) {
            const auto t_lists = std::chrono::steady_clock::now();
            ListArgs   l;
            l.xyf = a.xyf;
            l.cnt[0] = n_ga;
            l.cnt[1] = n_nga;
            l.base[0] = 0;
            l.base[1] = n_ga;
            l.n_all = n_all;
            l.g = geoms[pick];
            l.ncells = l.g.nx * l.g.ny;
            l.start = d_lcnt;
            l.cursor = l.start + 2 * (size_t)(l.ncells + 1);
            l.ent = static_cast<int *>(ws.get(4 * std::max<size_t>(n_ent[0] + n_ent[1], 1)));
            h->d_lblob = pool_alloc(mv.lblob_bytes);
            if (!l.ent || !h->d_lblob) return SLAM_E_NOMEM;
            mv.lblob = static_cast<const unsigned char *>(h->d_lblob);
            l.lbase[0] = mv.lbase[0];
            l.lbase[1] = mv.lbase[1];
            l.lblob = static_cast<unsigned char *>(h->d_lblob);
            l.loff_pts = mv.loff_pts;
            for (int k = 0; k < 2; ++k) {
                l.loff_start[k] = mv.loff_start[k];
                l.loff_axis[k] = mv.loff_axis[k];
            }
            SLAM_HIP(hipMemsetAsync(h->d_lblob, 0, mv.lblob_bytes, st));
            hipLaunchKernelGGL((list_scatter_kernel<0>), dim3(pblocks), dim3(256), 0, st, l);
            SLAM_TRY(launch_scan(l.start, l.start + (l.ncells + 1), l.ncells, l.lblob + l.loff_start[0], l.lblob + l.loff_start[1], 2, d_tiles, st));
            hipLaunchKernelGGL((list_scatter_kernel<1>), dim3(pblocks), dim3(256), 0, st, l);
            hipLaunchKernelGGL(list_sort_kernel, dim3(std::min((2 * l.ncells + kSortWaves - 1) / kSortWaves, 1024)), dim3(64 * kSortWaves), 0, st, l);
            hipLaunchKernelGGL(list_sort_big_kernel, dim3(l.ncells, 2), dim3(256), 0, st, l);
            SLAM_HIP(hipGetLastError());
            list_done(h);
            h->build_ms[3] = ms_since(t_lists);
        }
    }
    // the workspace goes back to the pool when this returns: the device must be done with it
    unsigned char *h_tail = static_cast<unsigned char *>(pinned_scratch(16 * (size_t)std::max(nc, 1) + 64));
    unsigned      *h_most = h_tail ? reinterpret_cast<unsigned *>(h_tail + 16 * (size_t)std::max(nc, 1)) : nullptr;
    if (h_most) SLAM_HIP(hipMemcpyAsync(h_most, d_most, 4, hipMemcpyDeviceToHost, st));
    SLAM_HIP(hipStreamSynchronize(st));
    if (h_most) h->max_cell_points = (int)*h_most;
    h->built_on_device = true;
    return SLAM_OK;
}

} // namespace

namespace slam {
namespace icp {

int build_index(slam_icp *h, const double *m_ga, int n_ga, const double *m_nga, int n_nga, bool on_device)
{
    if (h->prm.build_on_host && !on_device) return build_index_host(h, m_ga, n_ga, m_nga, n_nga);
    return build_index_device(h, m_ga, n_ga, m_nga, n_nga, on_device);
}

void release_index(slam_icp *h)
{
    if (h->d_blob) pool_free(h->d_blob);
    if (h->d_lblob) pool_free(h->d_lblob);
    h->d_blob = h->d_lblob = nullptr;
}

} // namespace icp
} // namespace slam
