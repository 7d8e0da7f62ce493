// icp_search.hpp -- device side of the exact 1-NN search on the cell index and of one point-to-point step
// (kdtree.cpp:378-391,515-683; icpPointToPoint.cpp:59-172), shared by the ICP translation units.
#pragma once
#include "icp_model.hpp"

namespace slam {
namespace icp {

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Best {
    float    d;    // squared float distance (kdtree.h:33)
    unsigned oidx; // original index within the class (kdtree.h:34); filled by nn_search on return
    int      pos;  // position in the sorted pts array, -1 = none
};

// kdtree.cpp:610-612: dis += squared(data[i][k]-qv[k]), k = 0 then 1, no FMA
__device__ inline float dist2(const float2 m, float qx, float qy)
{
    const float dx = m.x - qx;
    const float dy = m.y - qy;
    return __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));
}

// The same value for the loops that scan a cell range four candidates at a time: written on two-element vectors so that the
// difference and the squares are ONE packed instruction each (v_pk_add_f32, v_pk_mul_f32: a model point is an aligned register
// pair as it comes from ds_read_b64 / global_load_dwordx2), and the sum as an instruction of its own -- left to itself the
// vectoriser packs across neighbouring candidates instead and pays three register moves per pair (16 instructions per four
// candidates; 12 this way).  Every operation is an IEEE single operation as before (-ffp-contract=off: nothing is fused).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ inline float dist2p(const float2 m, float qx, float qy)
{
    const v2f p = {m.x, m.y}, q = {qx, qy};
    const v2f dl = p - q;
    const v2f sq = dl * dl;
    float     r;
    asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(sq.x), "v"(sq.y));
    return r;
}
// One candidate of a fast (EXACT = false) scan in the second-best form: best, its position, and the SECOND best of everything this
// call has examined together with the best it was given -- one v_med3_f32 where the flag form spends a compare, an AND and an OR.
// An exact tie is second == best when the call ends (scan_tie); a tie between candidates that were both beaten later is no tie.
__device__ inline void scan_step(Best &b, float &d2nd, float d, int i)
{
    d2nd = __builtin_amdgcn_fmed3f(b.d, d, d2nd);
    const bool up = d < b.d;
    b.d = up ? d : b.d;
    b.pos = up ? i : b.pos;
}
__device__ inline bool scan_tie(const Best &b, float d2nd) { return (d2nd == b.d) & (b.pos >= 0); }
// A seeded search in this form starts from the seed's distance ONE ULP UP (d >= 0: the next float is the next integer pattern): the
// seed is met again by whichever lane's share of a span holds it -- every cell the disk of its distance touches is scanned -- and
// must then beat the starting value instead of equalling it (which would read as a tie with itself).
__device__ inline float ulp_above(float d) { return __int_as_float(__float_as_int(d) + 1); }

// PK: which forms a fast scan uses -- false: dist2, minimum tree of four, tie flag (what rounds 1-3 ran; the spread form, the
// cooperative rounds, the normals and slam_icp_nearest still do); true: dist2p, scan_step / scan_tie (the batch kernels' ring passes).
// Measured one by one on config 2's scans (tools/pair_time.py, 256 scans in pairs / one per workgroup): the list form's second best
// 0.629 -> 0.612 / 0.394 -> 0.400; + packed distances 0.603 / 0.397; + second best in the ring scans **0.583 / 0.375 ms** (1024 scans in
// pairs 1.207 -> 1.122) -- the first two alone cost the one-scan-per-workgroup kernels a per cent each, all three together gain 4.6.
template <bool PK>
__device__ inline float dist2s(const float2 m, float qx, float qy)
{
    return PK ? dist2p(m, qx, qy) : dist2(m, qx, qy);
}

// (int)floorf(x) as the ONE instruction the hardware has for it (v_cvt_flr_i32_f32: same value for every finite x in range, same
// clamping outside, 0 for NaN; the compiler emits v_floor_f32 + v_cvt_i32_f32).
__device__ inline int ifloor(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// The radius of the disk a search still has to look at, from the best SQUARED distance: the hardware's v_sqrt_f32 as it is (one ulp;
// the correctly rounded sqrtf is six instructions around it) -- the lattice margin (h / 1024 and more) is what makes the disk safe.
__device__ inline float disk_radius(float d2) { return __builtin_amdgcn_sqrtf(d2); }

// Candidate i at squared distance d.  Ties go to the lowest ORIGINAL index (the
// reference leaves ties to the kd-tree's visit order; this is the brute-force
// arbiter's rule, kdtree.cpp:360-375): the index is only read when d == best.
template <typename StartT>
__device__ inline void consider(Best &b, float d, int i, const StartT *oidx)
{
    if (d < b.d) {
        b.d = d;
        b.pos = i;
    } else if (d == b.d && b.pos >= 0) {
        if ((unsigned)oidx[i] < (unsigned)oidx[b.pos]) b.pos = i;
    }
}

// All points of cells [c0, c1] of one lattice row: they are contiguous in the
// sorted array.  The G lanes of the group take consecutive points; four loads
// are kept in flight per lane and their distances are independent chains.
// EXACT = false is the fast form: branch-free minimum by (distance, position)
// plus a flag that says whether an exact tie d == best was ever seen; the
// caller then repeats the search with EXACT = true (ties by original index).
template <int G, typename StartT, bool EXACT, bool PK = false>
__device__ inline void scan_range(Best &b, bool &tie, const float2 *pts, const StartT *oidx, int a, int e, int sub, float qx,
                                  float qy)
{
    int   i = a + sub;
    float d2nd = FLT_MAX;
    for (; i + 3 * G < e; i += 4 * G) {
        const float2 m0 = pts[i], m1 = pts[i + G], m2 = pts[i + 2 * G], m3 = pts[i + 3 * G];
        const float  d0 = dist2s<PK>(m0, qx, qy), d1 = dist2s<PK>(m1, qx, qy), d2 = dist2s<PK>(m2, qx, qy), d3 = dist2s<PK>(m3, qx, qy);
        if (EXACT) {
            if (fminf(fminf(d0, d1), fminf(d2, d3)) <= b.d) {
                consider<StartT>(b, d0, i, oidx);
                consider<StartT>(b, d1, i + G, oidx);
                consider<StartT>(b, d2, i + 2 * G, oidx);
                consider<StartT>(b, d3, i + 3 * G, oidx);
            }
        } else if (PK) {
            scan_step(b, d2nd, d0, i);
            scan_step(b, d2nd, d1, i + G);
            scan_step(b, d2nd, d2, i + 2 * G);
            scan_step(b, d2nd, d3, i + 3 * G);
        } else {
            // min of the four (first position wins), then one compare against the running best
            const bool  s01 = d1 < d0, s23 = d3 < d2;
            const float m01 = s01 ? d1 : d0, m23 = s23 ? d3 : d2;
            const int   p01 = s01 ? i + G : i, p23 = s23 ? i + 3 * G : i + 2 * G;
            const bool  s = m23 < m01;
            const float m = s ? m23 : m01;
            const int   pm = s ? p23 : p01;
            tie |= (d0 == d1) | (d2 == d3) | (m01 == m23) | (m == b.d);
            const bool up = m < b.d;
            b.d = up ? m : b.d;
            b.pos = up ? pm : b.pos;
        }
    }
    for (; i < e; i += G) {
        const float d = dist2(pts[i], qx, qy);
        if (EXACT) {
            consider<StartT>(b, d, i, oidx);
        } else if (PK) {
            scan_step(b, d2nd, d, i);
        } else {
            tie |= (d == b.d);
            const bool up = d < b.d;
            b.d = up ? d : b.d;
            b.pos = up ? i : b.pos;
        }
    }
    if (!EXACT && PK) tie |= scan_tie(b, d2nd);
}

template <int G, typename StartT, bool EXACT, bool PK = false>
__device__ inline void scan_span(Best &b, bool &tie, const StartT *start, const float2 *pts, const StartT *oidx,
                                 int row_base, int c0, int c1, int sub, float qx, float qy)
{
    if (c0 > c1) return;
    scan_range<G, StartT, EXACT, PK>(b, tie, pts, oidx, (int)start[row_base + c0], (int)start[row_base + c1 + 1], sub, qx, qy);
}

// Fast-path minimum over the G lanes of a group: (distance, position) only, on the DPP cross-lane path for
// the steps inside a row of 16 (no LDS round trip); equal distances at different positions raise `tie`,
// which sends the group to the exact pass.  `tie` itself is OR-ed over the group.
template <int CTRL>
__device__ inline void lean_step_dpp(Best &b, bool &tie)
{
    const float od = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b.d), CTRL, 0xf, 0xf, false));
    const int   op = __builtin_amdgcn_update_dpp(0, b.pos, CTRL, 0xf, 0xf, false);
    const int   ot = __builtin_amdgcn_update_dpp(0, (int)tie, CTRL, 0xf, 0xf, false);
    tie |= (bool)ot | ((od == b.d) & (op != b.pos) & (op >= 0) & (b.pos >= 0));
    const bool take = (od < b.d) | ((od == b.d) & (op >= 0) & ((b.pos < 0) | (op < b.pos)));
    b.d = take ? od : b.d;
    b.pos = take ? op : b.pos;
}

__device__ inline void lean_step_shfl(Best &b, bool &tie, int mask)
{
    const float od = __shfl_xor(b.d, mask);
    const int   op = __shfl_xor(b.pos, mask);
    const int   ot = __shfl_xor((int)tie, mask);
    tie |= (bool)ot | ((od == b.d) & (op != b.pos) & (op >= 0) & (b.pos >= 0));
    const bool take = (od < b.d) | ((od == b.d) & (op >= 0) & ((b.pos < 0) | (op < b.pos)));
    b.d = take ? od : b.d;
    b.pos = take ? op : b.pos;
}

template <int G>
__device__ inline void group_min_lean(Best &b, bool &tie)
{
    if (G >= 2) lean_step_dpp<0xB1>(b, tie);  // quad_perm [1,0,3,2]
    if (G >= 4) lean_step_dpp<0x4E>(b, tie);  // quad_perm [2,3,0,1]
    if (G >= 8) lean_step_dpp<0x141>(b, tie); // row_half_mirror
    if (G >= 16) lean_step_dpp<0x140>(b, tie); // row_mirror
    if (G >= 32) lean_step_shfl(b, tie, 16);
    if (G >= 64) lean_step_shfl(b, tie, 32);
}

template <int G, typename StartT>
__device__ inline void group_min(Best &b, const StartT *oidx)
{
    b.oidx = b.pos >= 0 ? (unsigned)oidx[b.pos] : 0xffffffffu;
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
        const float    od = __shfl_xor(b.d, off);
        const unsigned oo = (unsigned)__shfl_xor((int)b.oidx, off);
        const int      op = __shfl_xor(b.pos, off);
        if (od < b.d || (od == b.d && oo < b.oidx)) {
            b.d = od;
            b.oidx = oo;
            b.pos = op;
        }
    }
}

template <int G, typename StartT, bool EXACT, bool PK = false>
__device__ inline Best nn_search_impl(const IndexPtrs<StartT> &ix, const ModelView &mv, int cls, float qx, float qy,
                                      int sub, double gate, bool &tie);

// The search proper: fast pass, and the exact pass only for a group that met an
// exact distance tie (measure zero on noisy data, common on gridded maps).
template <int G, typename StartT, bool PK = false>
__device__ inline Best nn_search(const IndexPtrs<StartT> &ix, const ModelView &mv, int cls,
                                 float qx, float qy, int sub, double gate)
{
    bool tie = false;
    Best b = nn_search_impl<G, StartT, false, PK>(ix, mv, cls, qx, qy, sub, gate, tie); // `tie` is group-wide
    if (tie) {
        bool unused = false;
        b = nn_search_impl<G, StartT, true>(ix, mv, cls, qx, qy, sub, gate, unused);
    }
    return b;
}

// Exact 1-NN of (qx,qy) among the points of class `cls`, searched by the G
// lanes of a group (`sub` = lane within the group).  `gate` (double, squared
// metres) lets the search stop once no unseen point can pass the inlier test
// of icpPointToPoint.cpp:76; pass +inf for an ungated search.  On return all
// G lanes hold the same result; pos < 0 when the class is empty.
//
// Order of visits: the query's own cell, then square rings of radius 1, 2, 4,
// ... cells.  Inside a ring only the cells that intersect the disk of the
// current best distance are read (a skipped cell lies entirely farther than
// the best found so far, which only shrinks), and cells of the previous,
// smaller square are not read again.  The search ends when the best distance
// is below the distance to the ring's outer edge (minus a margin that absorbs
// the f32 rounding of the cell assignment), when that edge is beyond the
// inlier gate, or when the ring covers the whole lattice.
template <int G, typename StartT, bool EXACT, bool PK>
__device__ inline Best nn_search_impl(const IndexPtrs<StartT> &ix, const ModelView &mv, int cls,
                                      float qx, float qy, int sub, double gate, bool &tie)
{
    const Lattice &L = mv.lat;
    const StartT *start = ix.start[cls];
    const float2 *pts = ix.pts + mv.base[cls];
    const StartT *oidx = ix.oidx + mv.base[cls];

    Best b;
    b.d = FLT_MAX;
    b.oidx = 0xffffffffu;
    b.pos = -1;
    if (mv.n_cls[cls] <= 0) return b;

    const float fx = (qx - L.x0) * L.inv_h, fy = (qy - L.y0) * L.inv_h;
    const int   cx = clampi(ifloor(fx), 0, L.nx - 1);
    const int   cy = clampi(ifloor(fy), 0, L.ny - 1);

    scan_span<G, StartT, EXACT, PK>(b, tie, start, pts, oidx, cy * L.nx, cx, cx, sub, qx, qy);
    if (G > 1) {
        if (EXACT)
            group_min<G, StartT>(b, oidx);
        else
            group_min_lean<G>(b, tie);
    }

    int rp = 0; // radius of the square already visited
    for (int r = 1;; r *= 2) {
        // the ring's own extent, then the disk of the current best distance
        int y_lo = max(cy - r, 0), y_hi = min(cy + r, L.ny - 1);
        int x_lo = max(cx - r, 0), x_hi = min(cx + r, L.nx - 1);
        const bool covers = (x_lo == 0) & (y_lo == 0) & (x_hi == L.nx - 1) & (y_hi == L.ny - 1);
        if (b.d < FLT_MAX) {
            // points of a column (row) above cell(q + R) have x (y) > q + R: the cell map is monotone
            const float R = (disk_radius(b.d) + L.margin) * L.inv_h;
            x_lo = max(x_lo, ifloor(fx - R));
            x_hi = min(x_hi, ifloor(fx + R));
            y_lo = max(y_lo, ifloor(fy - R));
            y_hi = min(y_hi, ifloor(fy + R));
        }
        for (int y = y_lo; y <= y_hi; ++y) {
            const int row = y * L.nx;
            if (y >= cy - rp && y <= cy + rp) {
                scan_span<G, StartT, EXACT, PK>(b, tie, start, pts, oidx, row, x_lo, min(x_hi, cx - rp - 1), sub, qx, qy);
                scan_span<G, StartT, EXACT, PK>(b, tie, start, pts, oidx, row, max(x_lo, cx + rp + 1), x_hi, sub, qx, qy);
            } else {
                scan_span<G, StartT, EXACT, PK>(b, tie, start, pts, oidx, row, x_lo, x_hi, sub, qx, qy);
            }
        }
        if (G > 1) {
            if (EXACT)
                group_min<G, StartT>(b, oidx);
            else
                group_min_lean<G>(b, tie);
        }
        const float bound = (float)r * L.h - L.margin;
        const float b2 = bound * bound;
        // every point outside the ring's square is farther than `bound` in x or in y
        if (covers || b.d < b2 || (double)b2 >= gate) break;
        rp = r;
    }
    if (G == 1 || !EXACT) b.oidx = b.pos >= 0 ? (unsigned)oidx[b.pos] : 0xffffffffu;
    return b;
}

// scan_range with the number of lanes that share the span known only at run time.  Seed-aware: the running
// best may be a point of this very span (a search seeded with last iteration's neighbour meets it again), which
// is not a tie.
template <typename StartT, bool EXACT, bool PK = false>
__device__ inline void scan_range_rt(Best &b, bool &tie, const float2 *pts, const StartT *oidx, int a, int e, int sub, int G,
                                     float qx, float qy)
{
    int   i = a + sub;
    float d2nd = FLT_MAX;
    for (; i + 3 * G < e; i += 4 * G) {
        const float2 m0 = pts[i], m1 = pts[i + G], m2 = pts[i + 2 * G], m3 = pts[i + 3 * G];
        const float  d0 = dist2s<PK>(m0, qx, qy), d1 = dist2s<PK>(m1, qx, qy), d2 = dist2s<PK>(m2, qx, qy), d3 = dist2s<PK>(m3, qx, qy);
        if (EXACT) {
            if (fminf(fminf(d0, d1), fminf(d2, d3)) <= b.d) {
                consider<StartT>(b, d0, i, oidx);
                consider<StartT>(b, d1, i + G, oidx);
                consider<StartT>(b, d2, i + 2 * G, oidx);
                consider<StartT>(b, d3, i + 3 * G, oidx);
            }
        } else if (PK) { // (the caller has put a seed's distance one ulp up: meeting the seed again is an update, not a tie)
            scan_step(b, d2nd, d0, i);
            scan_step(b, d2nd, d1, i + G);
            scan_step(b, d2nd, d2, i + 2 * G);
            scan_step(b, d2nd, d3, i + 3 * G);
        } else {
            const bool  s01 = d1 < d0, s23 = d3 < d2;
            const float m01 = s01 ? d1 : d0, m23 = s23 ? d3 : d2;
            const int   p01 = s01 ? i + G : i, p23 = s23 ? i + 3 * G : i + 2 * G;
            const bool  s = m23 < m01;
            const float m = s ? m23 : m01;
            const int   pm = s ? p23 : p01;
            tie |= (d0 == d1) | (d2 == d3) | (m01 == m23) | ((m == b.d) & (pm != b.pos));
            const bool up = m < b.d;
            b.d = up ? m : b.d;
            b.pos = up ? pm : b.pos;
        }
    }
    for (; i < e; i += G) {
        const float d = dist2(pts[i], qx, qy);
        if (EXACT) {
            consider<StartT>(b, d, i, oidx);
        } else if (PK) {
            scan_step(b, d2nd, d, i);
        } else {
            tie |= (d == b.d) & (i != b.pos);
            const bool up = d < b.d;
            b.d = up ? d : b.d;
            b.pos = up ? i : b.pos;
        }
    }
    if (!EXACT && PK) tie |= scan_tie(b, d2nd);
}

// A long span by all G lanes of the group, eight loads in flight per lane (a span of thousands of points is a
// few round trips), then the rest four at a time.  Seed-aware like scan_range_rt.
constexpr int kDeep = 8;
constexpr int kProbeFromLevel = 4; // nn_search_rows_impl: levels of this radius (cells) and beyond are probed before they are read, when no candidate exists yet
template <int G, typename StartT, bool EXACT>
__device__ inline void scan_range_deep(Best &b, bool &tie, const float2 *pts, const StartT *oidx, int a, int e, int lig, float qx,
                                       float qy)
{
    int i = a + lig;
    for (; i + (kDeep - 1) * G < e; i += kDeep * G) {
        float2 m[kDeep];
#pragma unroll
        for (int k = 0; k < kDeep; ++k) m[k] = pts[i + k * G];
#pragma unroll
        for (int k = 0; k < kDeep; ++k) {
            const float d = dist2(m[k], qx, qy);
            if (EXACT) {
                consider<StartT>(b, d, i + k * G, oidx);
            } else {
                tie |= (d == b.d) & (i + k * G != b.pos);
                const bool up = d < b.d;
                b.d = up ? d : b.d;
                b.pos = up ? i + k * G : b.pos;
            }
        }
    }
    scan_range_rt<StartT, EXACT>(b, tie, pts, oidx, i - lig, e, lig, G, qx, qy);
}

// nn_search with last iteration's neighbour of the same scene point as the first candidate (seed = its position
// in the sorted array, -1 = none): the disk of its distance prunes the search from the first cell on, where an
// unseeded search reads its first non-empty ring whole.  When that disk touches at most 3 x 3 cells they are
// read in one pass and the search is over (every cell the disk touches has been seen); otherwise the rings run
// as usual from the seeded best.  Same result as nn_search.
template <int G, typename StartT, bool EXACT, bool PK = false>
__device__ inline Best nn_search_seeded_impl(const IndexPtrs<StartT> &ix, const ModelView &mv, int cls, float qx, float qy,
                                             int sub, double gate, bool &tie, int seed, float empty_in, float move, float &empty_out)
{
    const Lattice &L = mv.lat;
    const StartT  *start = ix.start[cls];
    const float2  *pts = ix.pts + mv.base[cls];
    const StartT  *oidx = ix.oidx + mv.base[cls];

    Best b;
    b.d = FLT_MAX;
    b.oidx = 0xffffffffu;
    b.pos = -1;
    empty_out = 0.0f;
    if (mv.n_cls[cls] <= 0) return b;

    const float fx = (qx - L.x0) * L.inv_h, fy = (qy - L.y0) * L.inv_h;
    const int   cx = clampi(ifloor(fx), 0, L.nx - 1);
    const int   cy = clampi(ifloor(fy), 0, L.ny - 1);
    if (seed >= 0) {
        b.d = dist2(pts[seed], qx, qy);
        if (PK && !EXACT) b.d = ulp_above(b.d);
        b.pos = seed;
        const float R = (disk_radius(b.d) + L.margin) * L.inv_h;
        const int   x_lo = max(0, ifloor(fx - R)), x_hi = min(L.nx - 1, ifloor(fx + R));
        const int   y_lo = max(0, ifloor(fy - R)), y_hi = min(L.ny - 1, ifloor(fy + R));
        if (x_hi - x_lo <= 2 && y_hi - y_lo <= 2 && x_lo <= x_hi && y_lo <= y_hi) {
            for (int y = y_lo; y <= y_hi; ++y)
                scan_range_rt<StartT, EXACT, PK>(b, tie, pts, oidx, (int)start[y * L.nx + x_lo], (int)start[y * L.nx + x_hi + 1], sub, G, qx, qy);
            if (G > 1) {
                if (EXACT)
                    group_min<G, StartT>(b, oidx);
                else
                    group_min_lean<G>(b, tie);
            }
            if (G == 1 || !EXACT) b.oidx = (unsigned)oidx[b.pos];
            empty_out = __fsqrt_rn(b.d) * 0.999999f;
            return b;
        }
    }
    // cells that lie inside the disk last iteration proved empty, shrunk by the query's move since, need no visit
    int rp = -1, r = 0;
    if (empty_in > 0.0f && fx >= 0.0f && fx < (float)L.nx && fy >= 0.0f && fy < (float)L.ny) {
        const float D = empty_in - move - 2.0f * L.margin;
        if (D > 0.0f) rp = ifloor(fminf(D * L.inv_h * 0.70710677f, (float)(L.nx + L.ny))) - 1;
    }
    if (rp < 0) { // the usual start: the query's own cell
        scan_range_rt<StartT, EXACT, PK>(b, tie, pts, oidx, (int)start[cy * L.nx + cx], (int)start[cy * L.nx + cx + 1], sub, G, qx, qy);
        if (G > 1) {
            if (EXACT)
                group_min<G, StartT>(b, oidx);
            else
                group_min_lean<G>(b, tie);
        }
        rp = 0;
    }
    float bound;
    bool  all;
    for (r = rp + 1;; r *= 2) {
        int y_lo = max(cy - r, 0), y_hi = min(cy + r, L.ny - 1);
        int x_lo = max(cx - r, 0), x_hi = min(cx + r, L.nx - 1);
        const bool covers = (x_lo == 0) & (y_lo == 0) & (x_hi == L.nx - 1) & (y_hi == L.ny - 1);
        if (b.d < FLT_MAX) {
            const float R = (disk_radius(b.d) + L.margin) * L.inv_h;
            x_lo = max(x_lo, ifloor(fx - R));
            x_hi = min(x_hi, ifloor(fx + R));
            y_lo = max(y_lo, ifloor(fy - R));
            y_hi = min(y_hi, ifloor(fy + R));
        }
        for (int y = y_lo; y <= y_hi; ++y) {
            const int row = y * L.nx;
            if (y >= cy - rp && y <= cy + rp) {
                const int l1 = min(x_hi, cx - rp - 1), f2 = max(x_lo, cx + rp + 1);
                if (x_lo <= l1) scan_range_rt<StartT, EXACT, PK>(b, tie, pts, oidx, (int)start[row + x_lo], (int)start[row + l1 + 1], sub, G, qx, qy);
                if (f2 <= x_hi) scan_range_rt<StartT, EXACT, PK>(b, tie, pts, oidx, (int)start[row + f2], (int)start[row + x_hi + 1], sub, G, qx, qy);
            } else if (x_lo <= x_hi) {
                scan_range_rt<StartT, EXACT, PK>(b, tie, pts, oidx, (int)start[row + x_lo], (int)start[row + x_hi + 1], sub, G, qx, qy);
            }
        }
        if (G > 1) {
            if (EXACT)
                group_min<G, StartT>(b, oidx);
            else
                group_min_lean<G>(b, tie);
        }
        bound = (float)r * L.h - L.margin;
        const float b2 = bound * bound;
        all = covers;
        if (covers || b.d < b2 || (double)b2 >= gate) break;
        rp = r;
    }
    {
        const float dn = b.d < FLT_MAX ? __fsqrt_rn(b.d) * 0.999999f : 1.0e30f;
        empty_out = (all || b.d < bound * bound) ? dn : fminf(dn, bound);
    }
    if (G == 1 || !EXACT) b.oidx = b.pos >= 0 ? (unsigned)oidx[b.pos] : 0xffffffffu;
    return b;
}

template <int G, typename StartT, bool PK = false>
__device__ inline Best nn_search_seeded(const IndexPtrs<StartT> &ix, const ModelView &mv, int cls, float qx, float qy, int sub,
                                        double gate, int seed, float empty_in, float move, float &empty_out)
{
    bool tie = false;
    Best b = nn_search_seeded_impl<G, StartT, false, PK>(ix, mv, cls, qx, qy, sub, gate, tie, seed, empty_in, move, empty_out);
    if (tie) { // rare: the plain exact search (the seed changes the cost of a search, never its result)
        bool unused = false;
        b = nn_search_impl<G, StartT, true>(ix, mv, cls, qx, qy, sub, gate, unused);
        empty_out = 0.0f;
    }
    return b;
}

// ---- The model TILE of a wavefront (round 5).  A model too large for LDS (the reference's own cap is 2 x 19 999 points,
// icpTools.h:21) leaves the index in HBM/L2, where a query is a chain of dependent gathers: 221 M lane-loads per launch of 256
// scans, each its own trip through the texture path, six or seven round trips deep.  But LDS is then EMPTY, and the queries of a
// wavefront are ADJACENT beams: their 3 x 3 cell blocks overlap and lie along one stretch of wall.  So the wavefront stages the
// union of those blocks (a cell of margin around it) -- per lattice row one CONTIGUOUS span of the sorted array, read with
// coalesced loads, every line once -- into its own LDS region together with the cell starts of the block, and KEEPS it: a
// converging scan moves by millimetres per iteration, so the tile staged around iteration 3 serves the other twenty-seven.  Its
// queries then run the seeded 3 x 3 scan of nn_search_seeded_impl out of LDS, with no load from L2 at all (the seed's coordinates
// stay in registers, the neighbour's come from the tile).
// Exact like the path it replaces: the same cells are scanned for every query (a cell's points are the same bytes, staged);
// queries whose disk is larger than 3 x 3 cells (the first iterations), lanes of the other class, queries that have left the
// staged block (it is staged again around them) and wavefronts whose block does not fit take the ordinary path through L2.
constexpr int      kTilePts = 960, kTileCells = 736, kTileRows = 48;
constexpr unsigned kWaveTileBytes = 8u * kTilePts + 2u * kTileCells + 4u * kTileRows + 2u * kTileRows;
constexpr int      kTileMaxFail = 3, kTileMaxRestage = 20; // a wavefront whose block does not fit, or that keeps staging (two classes in
                                                           // turn, a scan that will not settle), stops trying
constexpr int      kTileFirstIter = 3; // no staging before: the first steps move a query by cells, a tile staged then is stale at once
static_assert(kWaveTileBytes % 16 == 0, "wave tiles keep 16-byte alignment");

struct WaveTile {
    float2         *pts;  // [kTilePts] the staged spans, row after row
    unsigned short *cell; // [rows][ncol] cell starts (ncol - 1 cells and the end of the last), relative to the row's first staged point
    int            *rowA; // [kTileRows] position of the row's first staged point in the class's sorted array
    unsigned short *rowOff; // [kTileRows] where the row's points start in pts
};

// what a wavefront knows about its tile from one iteration to the next (the same in every lane)
struct TileState {
    int  X0, Y0, X1, Y1; // the staged block of cells, inclusive
    int  restages;
    bool staged, cls1, off;
};

__device__ inline WaveTile wave_tile_at(unsigned char *base)
{
    WaveTile t;
    t.pts = reinterpret_cast<float2 *>(base);
    t.cell = reinterpret_cast<unsigned short *>(base + 8u * kTilePts);
    t.rowA = reinterpret_cast<int *>(base + 8u * kTilePts + 2u * kTileCells);
    t.rowOff = reinterpret_cast<unsigned short *>(base + 8u * kTilePts + 2u * kTileCells + 4u * kTileRows);
    return t;
}

__device__ inline int wave_min_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}

// The box of cells the disk of a seeded query's current best distance touches, as nn_search_seeded_impl computes it.
struct SeedBox {
    int  x_lo, x_hi, y_lo, y_hi;
    bool small; // at most 3 x 3 cells: one pass over them ends the search
};

__device__ inline SeedBox seed_box(const Lattice &L, float qx, float qy, float d)
{
    const float fx = (qx - L.x0) * L.inv_h, fy = (qy - L.y0) * L.inv_h;
    const float R = (disk_radius(d) + L.margin) * L.inv_h;
    SeedBox     bx;
    bx.x_lo = max(0, ifloor(fx - R));
    bx.x_hi = min(L.nx - 1, ifloor(fx + R));
    bx.y_lo = max(0, ifloor(fy - R));
    bx.y_hi = min(L.ny - 1, ifloor(fy + R));
    bx.small = bx.x_hi - bx.x_lo <= 2 && bx.y_hi - bx.y_lo <= 2 && bx.x_lo <= bx.x_hi && bx.y_lo <= bx.y_hi;
    return bx;
}

// Stages the block of cells [X0, X1] x [Y0, Y1] of one class into the wavefront's tile; all arguments are the same in every lane
// (scalar registers), every lane takes part.  False (nothing usable staged) when the block does not fit.
template <typename StartT>
__device__ __forceinline__ bool wave_tile_fill(const WaveTile &wt, const IndexPtrs<StartT> &ix, const ModelView &mv, bool cls1, int X0,
                                               int Y0, int X1, int Y1)
{
    const int lane = (int)threadIdx.x & 63;
    const int nrow = Y1 - Y0 + 1, ncol = X1 - X0 + 2;
    if (nrow > kTileRows || nrow * ncol > kTileCells) return false;
    const Lattice &L = mv.lat;
    const StartT  *start = cls1 ? ix.start[1] : ix.start[0]; // (selects: an index would put the arrays on the stack)
    const float2  *pts = ix.pts + (cls1 ? mv.base[1] : mv.base[0]);
    int            A = 0, cnt = 0; // lane r: the span of row Y0 + r
    if (lane < nrow) {
        const int row = (Y0 + lane) * L.nx;
        A = (int)start[row + X0];
        cnt = (int)start[row + X1 + 1] - A;
    }
    int incl = cnt; // inclusive prefix over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        incl += lane >= o ? up : 0;
    }
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total > kTilePts) return false;
    const int off = incl - cnt;
    if (lane < nrow) {
        wt.rowA[lane] = A;
        wt.rowOff[lane] = (unsigned short)off;
    }
    for (int r = 0; r < nrow; ++r) {
        const int Ar = __builtin_amdgcn_readlane(A, r), nr = __builtin_amdgcn_readlane(cnt, r), offr = __builtin_amdgcn_readlane(off, r);
        const int row = (Y0 + r) * L.nx + X0;
        for (int c = lane; c < ncol; c += 64) wt.cell[r * ncol + c] = (unsigned short)((int)start[row + c] - Ar);
        for (int i = lane; i < nr; i += 64) wt.pts[offr + i] = pts[Ar + i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // LDS serves a wavefront's accesses in order: what the lanes wrote is
    __builtin_amdgcn_wave_barrier();                        // there for the reads that follow; the compiler must not move them up
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return true;
}

// Called by ALL lanes of a wavefront (the control flow around the staging is uniform).  `want` = this lane has a seeded query of
// class `cls` whose box `bx` is small.  Returns true for the lanes whose box lies in the wavefront's tile (staged now or in an
// earlier iteration): their scan reads the tile.  False = take the ordinary path.
template <typename StartT>
__device__ __forceinline__ bool wave_tile_ready(const WaveTile &wt, TileState &ts, const IndexPtrs<StartT> &ix, const ModelView &mv,
                                                bool want, int cls, const SeedBox &bx)
{
    if (ts.off) return false;
    const unsigned long long any = __ballot(want);
    if (__popcll(any) < 8) return false; // too few takers to pay for a staging (the first iterations; a nearly empty pass)
    const bool cls1 = __builtin_amdgcn_readlane(cls, __builtin_ctzll(any)) != 0; // the wavefront's class: its first taker's
    const bool mine = want && (cls != 0) == cls1;
    const bool have = ts.staged && ts.cls1 == cls1;
    const bool inside = have && bx.x_lo >= ts.X0 && bx.x_hi <= ts.X1 && bx.y_lo >= ts.Y0 && bx.y_hi <= ts.Y1;
    if (__ballot(mine && !inside)) { // somebody's block is not staged: stage again, around everybody (uniform branch)
        const Lattice &L = mv.lat;
        // (the same in every lane, and SAID so: the loops of the staging then run on scalar registers)
        int X0 = __builtin_amdgcn_readfirstlane(wave_min_i32(mine ? bx.x_lo : 0x7fffffff)) - 1;
        int Y0 = __builtin_amdgcn_readfirstlane(wave_min_i32(mine ? bx.y_lo : 0x7fffffff)) - 1;
        int X1 = -__builtin_amdgcn_readfirstlane(wave_min_i32(mine ? -bx.x_hi : 0x7fffffff)) + 1;
        int Y1 = -__builtin_amdgcn_readfirstlane(wave_min_i32(mine ? -bx.y_hi : 0x7fffffff)) + 1;
        X0 = max(X0, 0), Y0 = max(Y0, 0), X1 = min(X1, L.nx - 1), Y1 = min(Y1, L.ny - 1);
        bool ok = false;
        if (have) // the other pass of this wavefront lives in the same tile: keep its block in
            ok = wave_tile_fill<StartT>(wt, ix, mv, cls1, min(X0, ts.X0), min(Y0, ts.Y0), max(X1, ts.X1), max(Y1, ts.Y1));
        if (ok) {
            X0 = min(X0, ts.X0), Y0 = min(Y0, ts.Y0), X1 = max(X1, ts.X1), Y1 = max(Y1, ts.Y1);
        } else {
            ok = wave_tile_fill<StartT>(wt, ix, mv, cls1, X0, Y0, X1, Y1);
        }
        ts.staged = ok;
        ts.cls1 = cls1;
        ts.X0 = X0, ts.Y0 = Y0, ts.X1 = X1, ts.Y1 = Y1;
        ts.restages += ok ? 1 : (kTileMaxRestage / kTileMaxFail + 1); // (a failure weighs as much as a third of the budget)
        if (ts.restages > kTileMaxRestage) ts.off = true;            // (this staging still serves this pass)
        return mine && ok;
    }
    return mine && inside;
}

// The seeded 3 x 3 scan of nn_search_seeded_impl out of the wavefront's tile.  b comes in as the seed (distance one ulp up, PK form)
// and leaves as the group's best, m as its coordinates; positions are those of the class's sorted array, as everywhere.
template <int G, bool PK>
__device__ __forceinline__ void wave_tile_scan(Best &b, float2 &m, bool &tie, const WaveTile &wt, const TileState &ts, const SeedBox &bx,
                                               int sub, float qx, float qy)
{
    constexpr int kHere = 0x7ffffff0; // "the best so far is not of this row"
    const int     ncol = ts.X1 - ts.X0 + 2;
    for (int y = bx.y_lo; y <= bx.y_hi; ++y) {
        const int r = y - ts.Y0;
        const int a = (int)wt.cell[r * ncol + (bx.x_lo - ts.X0)], e = (int)wt.cell[r * ncol + (bx.x_hi + 1 - ts.X0)];
        Best      bl;
        bl.d = b.d;
        bl.pos = kHere;
        bl.oidx = 0xffffffffu;
        scan_range_rt<unsigned short, false, PK>(bl, tie, wt.pts + (int)wt.rowOff[r], nullptr, a, e, sub, G, qx, qy);
        if (bl.pos != kHere) {
            b.d = bl.d;
            b.pos = wt.rowA[r] + bl.pos;
        }
    }
    if (G > 1) group_min_lean<G>(b, tie);
    // the winner's coordinates, from the tile: it lies in one of the rows of the box (the seed too: it is inside its own disk)
    m = make_float2(0.f, 0.f);
    for (int y = bx.y_lo; y <= bx.y_hi; ++y) {
        const int r = y - ts.Y0, rel = b.pos - wt.rowA[r];
        if (rel >= 0 && rel < (int)wt.cell[r * ncol + ncol - 1]) m = wt.pts[(int)wt.rowOff[r] + rel];
    }
}

// What a finished search proves about the query and what the next iteration's search of the same scene
// point starts from (icp_single.hip): the neighbour's position in the sorted array, and a radius within which
// the class has no point (the neighbour's distance, or the edge of the last ring when the inlier gate ended
// the search first).
struct Seed {
    int   pos;   // -1: none
    float empty; // metres; 0: nothing known
};

// The same search by a WIDE group of G = 16 or 64 lanes (icp_single.hip: few queries, or an index read from
// HBM/L2, where a query is a chain of dependent loads and the chip has lanes to spare).  A ring level's rows are
// dealt over the lanes -- as many lanes per row as the level's row count leaves (16 for the 3 x 3 block, one for
// levels of 33 rows and more) -- so that the extents of ALL rows of the level arrive in one round trip and their
// points in the next, whatever the radius: a query that is metres from the model (or beyond the inlier gate
// altogether) costs one or two round trips per level instead of two per row.  A row with more points than its
// lanes take in two steps (a lidar cloud holds hundreds of points per cell near the sensor) is left to the
// whole group, one such row after the other.  The query's own cell is part of the first level.  Visits the same
// cells as nn_search_impl, so the result is the same (ties: flagged by the fast form, resolved by the exact one).
// the minimum of a float over the G lanes of a query's group, in every lane
template <int G>
__device__ inline float group_fmin(float v)
{
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

template <int G, typename StartT, bool EXACT>
__device__ inline Best nn_search_rows_impl(const IndexPtrs<StartT> &ix, const ModelView &mv, int cls, float qx, float qy,
                                           int lig, double gate, bool &tie, const Seed seed, float move, float &empty_out)
{
    static_assert(G == 16 || G == 64, "16 or 64 lanes per query");
    const Lattice &L = mv.lat;
    const StartT  *start = ix.start[cls];
    const float2  *pts = ix.pts + mv.base[cls];
    const StartT  *oidx = ix.oidx + mv.base[cls];

    Best b;
    b.d = FLT_MAX;
    b.oidx = 0xffffffffu;
    b.pos = -1;
    empty_out = 0.0f;
    if (mv.n_cls[cls] <= 0) return b;

    const float fx = (qx - L.x0) * L.inv_h, fy = (qy - L.y0) * L.inv_h;
    const int   cx = clampi(ifloor(fx), 0, L.nx - 1);
    const int   cy = clampi(ifloor(fy), 0, L.ny - 1);
    const int   group_base = ((int)threadIdx.x & 63) & ~(G - 1); // first lane of this query's group in the wavefront

    int rp = -1, r = 1; // radius of the square already known (visited or proven empty), radius of the next level
    // Seeds.  (1) Last iteration the class had no point within seed.empty of the query; the query has moved by at
    // most `move` since, so there is none within D = seed.empty - move of it now: the square of cells that lies
    // inside that disk needs no visit (only for a query inside the lattice: its own cell is then really its cell).
    // (2) Last iteration's neighbour is a candidate from the start: the disk of its distance prunes the first level.
    if (seed.empty > 0.0f && fx >= 0.0f && fx < (float)L.nx && fy >= 0.0f && fy < (float)L.ny) {
        const float D = seed.empty - move - 2.0f * L.margin;
        if (D > 0.0f && (double)D * (double)D >= gate) {
            // nothing of the class within D, and D is beyond the inlier gate (icpPointToPoint.cpp:76): whatever the
            // nearest point is, the caller drops it -- no read at all.  (Half of config 3's scene queries: a cloud seen
            // from the next pose has parts the target never saw; they cost two ring levels of ~2000 cells per iteration.)
            empty_out = D;
            return b;
        }
        if (D > 0.0f) {
            const float cells = fminf(D * L.inv_h * 0.70710677f, (float)(L.nx + L.ny)); // (rp + 1) h sqrt(2) <= D
            rp = ifloor(cells) - 1;
            if (rp >= 0) r = rp + 1;
        }
    }
    if (seed.pos >= 0) {
        b.d = dist2(pts[seed.pos], qx, qy);
        b.pos = seed.pos;
    }
    float bound;
    bool  all;
    for (;; r *= 2) {
        int y_lo = max(cy - r, 0), y_hi = min(cy + r, L.ny - 1);
        int x_lo = max(cx - r, 0), x_hi = min(cx + r, L.nx - 1);
        const bool covers = (x_lo == 0) & (y_lo == 0) & (x_hi == L.nx - 1) & (y_hi == L.ny - 1);
        float      Rd = 1.0e30f; // the disk of the best so far, in cells
        // A level entered WITHOUT a candidate (no seed, nothing in the levels inside) used to read every point of its square
        // ring -- at r = 16 cells a thousand cells, of which the nearest point's disk touches a few dozen: 20 us for one query of
        // config 3's first iteration (a scene point beside the vehicle, the rings of lidar returns 1-2 m around it), the whole
        // workgroup waiting at the barrier.  Probe first: per lattice row the point(s) of the cells nearest the query's column
        // -- two or three dependent loads, all rows at once -- give an upper bound within about a cell of the nearest
        // distance; the level then reads the cells under THAT disk only.  (The bound only prunes: the nearest point lies
        // inside its disk and is found by the scan itself, ties and all.)
        float cand = b.d;
        if (cand == FLT_MAX && r >= kProbeFromLevel) {
            float pd = FLT_MAX;
            for (int y0 = y_lo; y0 <= y_hi; y0 += G) {
                const int y = y0 + lig, row = y * L.nx;
                if (y <= y_hi) {
                    if (rp >= 0 && y >= cy - rp && y <= cy + rp) {
                        const int l1 = min(x_hi, cx - rp - 1), f2 = max(x_lo, cx + rp + 1);
                        if (x_lo <= l1) {
                            const int a = (int)start[row + x_lo], e = (int)start[row + l1 + 1];
                            if (e > a) pd = fminf(pd, dist2(pts[e - 1], qx, qy)); // the last point left of the inner square
                        }
                        if (f2 <= x_hi) {
                            const int a = (int)start[row + f2], e = (int)start[row + x_hi + 1];
                            if (e > a) pd = fminf(pd, dist2(pts[a], qx, qy)); // the first point right of it
                        }
                    } else if (x_lo <= x_hi) {
                        const int a = (int)start[row + x_lo], e = (int)start[row + x_hi + 1];
                        if (e > a) {
                            const int c = (int)start[row + clampi(cx, x_lo, x_hi)]; // the first point at or right of the query's column
                            pd = fminf(pd, fminf(dist2(pts[min(c, e - 1)], qx, qy), dist2(pts[max(c - 1, a)], qx, qy)));
                        }
                    }
                }
            }
            cand = group_fmin<G>(pd);
        }
        if (cand < FLT_MAX) {
            const float R = (disk_radius(cand) + L.margin) * L.inv_h;
            Rd = R;
            x_lo = max(x_lo, ifloor(fx - R));
            x_hi = min(x_hi, ifloor(fx + R));
            y_lo = max(y_lo, ifloor(fy - R));
            y_hi = min(y_hi, ifloor(fy + R));
        }
        const int nrows = y_hi - y_lo + 1;
        // lanes per row: the largest power of two that still gives every row of the level a slot (at least one)
        int lpr_log = 0;
        while (lpr_log < 4 && (nrows << (lpr_log + 1)) <= G) ++lpr_log;
        const int lpr = 1 << lpr_log, slot = lig >> lpr_log, sub = lig & (lpr - 1), slots = G >> lpr_log;
        for (int y0 = y_lo; y0 <= y_hi; y0 += slots) {
            const int y = y0 + slot, row = y * L.nx;
            int       a1 = 0, e1 = 0, a2 = 0, e2 = 0; // the row's one or two spans of the sorted array
            if (y <= y_hi) {
                // the row's cells under the disk of the level's best (round 6): a point of lattice row y is at least dy rows from the
                // query, so within Rd of it only if no farther than sqrt(Rd^2 - dy^2) columns -- a far query beside a wall of stacked
                // lidar points reads the one or two cells per row that the disk touches where its bounding square holds metres of wall
                int xl = x_lo, xh = x_hi;
                if (Rd < 1.0e29f) {
                    const float dy = fmaxf(fmaxf((float)y - fy, fy - (float)(y + 1)), 0.0f);
                    const float half = __builtin_amdgcn_sqrtf(fmaxf(Rd * Rd - dy * dy, 0.0f)) + 1.0e-3f;
                    xl = max(xl, ifloor(fx - half));
                    xh = min(xh, ifloor(fx + half));
                }
                if (rp >= 0 && y >= cy - rp && y <= cy + rp) {
                    const int l1 = min(xh, cx - rp - 1), f2 = max(xl, cx + rp + 1);
                    if (xl <= l1) a1 = (int)start[row + xl], e1 = (int)start[row + l1 + 1];
                    if (f2 <= xh) a2 = (int)start[row + f2], e2 = (int)start[row + xh + 1];
                } else if (xl <= xh) {
                    a1 = (int)start[row + xl], e1 = (int)start[row + xh + 1];
                }
            }
            const bool heavy = (e1 - a1) + (e2 - a2) > 8 * lpr;
            if (!heavy) {
                scan_range_rt<StartT, EXACT>(b, tie, pts, oidx, a1, e1, sub, lpr, qx, qy);
                scan_range_rt<StartT, EXACT>(b, tie, pts, oidx, a2, e2, sub, lpr, qx, qy);
            }
            unsigned long long m = __ballot(heavy && sub == 0);
            if (G < 64) m = (m >> group_base) & ((1ull << G) - 1ull);
            while (m) {
                const int src = __builtin_ctzll(m);
                m &= m - 1;
                int A1, E1, A2, E2;
                if (G == 64) {
                    A1 = __builtin_amdgcn_readlane(a1, src), E1 = __builtin_amdgcn_readlane(e1, src);
                    A2 = __builtin_amdgcn_readlane(a2, src), E2 = __builtin_amdgcn_readlane(e2, src);
                } else {
                    A1 = __shfl(a1, src, G), E1 = __shfl(e1, src, G);
                    A2 = __shfl(a2, src, G), E2 = __shfl(e2, src, G);
                }
                scan_range_deep<G, StartT, EXACT>(b, tie, pts, oidx, A1, E1, lig, qx, qy);
                scan_range_deep<G, StartT, EXACT>(b, tie, pts, oidx, A2, E2, lig, qx, qy);
            }
        }
        if (EXACT)
            group_min<G, StartT>(b, oidx);
        else
            group_min_lean<G>(b, tie);
        bound = (float)r * L.h - L.margin;
        const float b2 = bound * bound;
        all = covers;
        if (covers || b.d < b2 || (double)b2 >= gate) break;
        rp = r;
    }
    // every point of the class is at least this far from the query: the neighbour itself when the search ran to
    // its end (its distance inside the last ring, or the whole lattice seen), else the last ring's edge
    {
        const float dn = b.d < FLT_MAX ? __fsqrt_rn(b.d) * 0.999999f : 1.0e30f;
        empty_out = (all || b.d < bound * bound) ? dn : fminf(dn, bound);
    }
    if (!EXACT) b.oidx = b.pos >= 0 ? (unsigned)oidx[b.pos] : 0xffffffffu;
    return b;
}

template <int G, typename StartT>
__device__ inline Best nn_search_rows(const IndexPtrs<StartT> &ix, const ModelView &mv, int cls, float qx, float qy, int lig,
                                      double gate, const Seed seed, float move, float &empty_out)
{
    bool tie = false;
    Best b = nn_search_rows_impl<G, StartT, false>(ix, mv, cls, qx, qy, lig, gate, tie, seed, move, empty_out);
    if (tie) {
        bool unused = false;
        b = nn_search_rows_impl<G, StartT, true>(ix, mv, cls, qx, qy, lig, gate, unused, seed, move, empty_out);
    }
    return b;
}

// Wavefront sum of a double on the VALU's DPP cross-lane path (no LDS
// crossbar): a 16-lane prefix by row_shr 1,2,4,8, then row_bcast 15 and 31 fold
// the four rows; lane 63 holds the total, which is returned to all lanes.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_shift_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ inline double uniform(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__device__ inline double wave_sum(double v)
{
    v += dpp_shift_f64<0x111, 0xf>(v); // row_shr:1
    v += dpp_shift_f64<0x112, 0xf>(v); // row_shr:2
    v += dpp_shift_f64<0x114, 0xf>(v); // row_shr:4
    v += dpp_shift_f64<0x118, 0xf>(v); // row_shr:8
    v += dpp_shift_f64<0x142, 0xa>(v); // row_bcast:15 into rows 1 and 3
    v += dpp_shift_f64<0x143, 0xc>(v); // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ inline double shfl_xor_f64(double v, int mask)
{
    return __hiloint2double(__shfl_xor(__double2hiint(v), mask), __shfl_xor(__double2loint(v), mask));
}

// Wavefront sums of EIGHT doubles at once: each exchange step halves the number of values a lane
// carries (lanes 0-31 keep values 0-3 and take the partner's, lanes 32-63 keep 4-7, and so on), so
// 10 additions per lane replace 48.  On return every lane holds the total of value (lane >> 3).
// Fixed order: bitwise reproducible.
__device__ inline double wave_sum8(const double a[8])
{
    const int  lane = (int)threadIdx.x & 63;
    const bool h32 = lane & 32, h16 = lane & 16, h8 = lane & 8;
    double v4[4], v2[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) v4[k] = (h32 ? a[k + 4] : a[k]) + shfl_xor_f64(h32 ? a[k] : a[k + 4], 32);
#pragma unroll
    for (int k = 0; k < 2; ++k) v2[k] = (h16 ? v4[k + 2] : v4[k]) + shfl_xor_f64(h16 ? v4[k] : v4[k + 2], 16);
    double v = (h8 ? v2[1] : v2[0]) + dpp_shift_f64<0x128, 0xf>(h8 ? v2[0] : v2[1]); // row_ror:8 = lane ^ 8
    v += dpp_shift_f64<0x141, 0xf>(v); // row_half_mirror: lane ^ 7
    v += dpp_shift_f64<0xB1, 0xf>(v);  // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_shift_f64<0x4E, 0xf>(v);  // quad_perm [2,3,0,1]: lane ^ 2
    return v;
}

// icpPointToPoint.cpp:159-162, closed form of svd -> V*U^T (oracle: o_p2p_rotation)
__device__ inline void p2p_rotation(const double H[4], double R_[4])
{
    const double det = H[0] * H[3] - H[1] * H[2];
    double a, b;
    if (det >= 0.0) {
        a = H[0] + H[3];
        b = H[1] - H[2];
    } else {
        a = H[0] - H[3];
        b = H[1] + H[2];
    }
    const double n = sqrt(a * a + b * b);
    double c = 1.0, s = 0.0;
    if (n > 0.0) {
        const double inv_n = 1.0 / n;
        c = a * inv_n;
        s = b * inv_n;
    }
    if (det >= 0.0) {
        R_[0] = c;
        R_[1] = -s;
        R_[2] = s;
        R_[3] = c;
    } else {
        R_[0] = c;
        R_[1] = s;
        R_[2] = s;
        R_[3] = -c;
    }
}

// matrix.cpp:420-508 Gauss-Jordan with full pivoting, 3x3, one rhs (oracle: o_solve3).  The pivot's row and column are
// run-time values; every array index below is a compile-time one (rows and columns are picked by selects), so A and b
// stay in registers -- indexed by irow / icol they went to scratch memory, 80 bytes per lane of every kernel that solves.
// The operations and their order are the reference's, result bit for bit.
__device__ inline double pick3(double v0, double v1, double v2, int i) { return i == 0 ? v0 : (i == 1 ? v1 : v2); }

__device__ inline bool solve3(double A[9], double b[3])
{
    int ipiv0 = 0, ipiv1 = 0, ipiv2 = 0;
    int irow = 0, icol = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double big = 0.0;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int pj = j == 0 ? ipiv0 : (j == 1 ? ipiv1 : ipiv2);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int  pk = k == 0 ? ipiv0 : (k == 1 ? ipiv1 : ipiv2);
                const bool c = (pj != 1) & (pk == 0) & (fabs(A[3 * j + k]) >= big);
                big = c ? fabs(A[3 * j + k]) : big;
                irow = c ? j : irow;
                icol = c ? k : icol;
            }
        }
        ipiv0 += icol == 0 ? 1 : 0;
        ipiv1 += icol == 1 ? 1 : 0;
        ipiv2 += icol == 2 ? 1 : 0;
        // rows irow and icol change places (the same row: nothing moves)
        double ri[3], rc[3];
#pragma unroll
        for (int l = 0; l < 3; l++) {
            ri[l] = pick3(A[l], A[3 + l], A[6 + l], irow);
            rc[l] = pick3(A[l], A[3 + l], A[6 + l], icol);
        }
        const double bi = pick3(b[0], b[1], b[2], irow), bcv = pick3(b[0], b[1], b[2], icol);
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int l = 0; l < 3; l++) {
                double v = A[3 * r + l];
                v = r == irow ? rc[l] : v;
                v = r == icol ? ri[l] : v;
                A[3 * r + l] = v;
            }
            double w = b[r];
            w = r == irow ? bcv : w;
            w = r == icol ? bi : w;
            b[r] = w;
        }
        // (row icol now holds ri, b[icol] holds bi)
        const double piv = pick3(ri[0], ri[1], ri[2], icol);
        if (fabs(piv) < 1e-20) return false;
        const double pivinv = 1.0 / piv;
        double       prow[3];
#pragma unroll
        for (int l = 0; l < 3; l++) prow[l] = (l == icol ? 1.0 : ri[l]) * pivinv;
        const double bp = bi * pivinv;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const bool   is_piv = r == icol;
            const double dum = pick3(A[3 * r], A[3 * r + 1], A[3 * r + 2], icol);
#pragma unroll
            for (int l = 0; l < 3; l++) {
                const double cur = l == icol ? 0.0 : A[3 * r + l];
                A[3 * r + l] = is_piv ? prow[l] : cur - prow[l] * dum;
            }
            b[r] = is_piv ? bp : b[r] - bp * dum;
        }
    }
    return true; // (column unscrambling only affects the inverse, not the solution vector)
}

// One point-to-point step from the nine sums S = {n, sum(p_m - c), sum(p_t - c), sum (p_t - c)(p_m - c)^T}
// (c = mv.cx, mv.cy): icpPointToPoint.cpp:128-171.  pose = r00 r01 r10 r11 t0 t1, updated in place when there
// is a correspondence; returns the step's delta (-1 without correspondences, :128-131).
__device__ inline double p2p_step(const double S[kNumAcc], const ModelView &mv, double pose[6], int &n_corr)
{
    n_corr = (int)S[0];
    if (n_corr == 0) return -1.0;
    const double inv = 1.0 / S[0];
    const double ma0 = S[1] * inv, ma1 = S[2] * inv; // mean of (p_m - c)
    const double mb0 = S[3] * inv, mb1 = S[4] * inv; // mean of (p_t - c)
    double       H[4], R_[4], t_[2];
    H[0] = S[5] - S[3] * ma0;
    H[1] = S[6] - S[3] * ma1;
    H[2] = S[7] - S[4] * ma0;
    H[3] = S[8] - S[4] * ma1;
    p2p_rotation(H, R_);
    const double mm0 = mv.cx + ma0, mm1 = mv.cy + ma1;
    const double mt0 = mv.cx + mb0, mt1 = mv.cy + mb1;
    t_[0] = mm0 - (R_[0] * mt0 + R_[1] * mt1); // :163
    t_[1] = mm1 - (R_[2] * mt0 + R_[3] * mt1);
    const double r00 = pose[0], r01 = pose[1], r10 = pose[2], r11 = pose[3], t0 = pose[4], t1 = pose[5];
    pose[0] = R_[0] * r00 + R_[1] * r10; // :166-167 R = R_*R ; t = R_*t + t_
    pose[1] = R_[0] * r01 + R_[1] * r11;
    pose[2] = R_[2] * r00 + R_[3] * r10;
    pose[3] = R_[2] * r01 + R_[3] * r11;
    pose[4] = (R_[0] * t0 + R_[1] * t1) + t_[0];
    pose[5] = (R_[2] * t0 + R_[3] * t1) + t_[1];
    const double a0 = R_[0] - 1.0, a3 = R_[3] - 1.0;
    const double nr2 = a0 * a0 + R_[1] * R_[1] + R_[2] * R_[2] + a3 * a3;
    const double nt2 = t_[0] * t_[0] + t_[1] * t_[1];
    return sqrt(nr2 > nt2 ? nr2 : nt2); // :170 max of the two norms (sqrt is monotone: same value)
}

struct Pose {
    double r00, r01, r10, r11, t0, t1;
};

// icpPointToPoint.cpp:69-70: (r00*x + r01*y) + t0 in double, stored to float
__device__ inline void transform_query(const Pose &T, const double2 P, float &qx, float &qy)
{
    qx = (float)__dadd_rn(__dadd_rn(__dmul_rn(T.r00, P.x), __dmul_rn(T.r01, P.y)), T.t0);
    qy = (float)__dadd_rn(__dadd_rn(__dmul_rn(T.r10, P.x), __dmul_rn(T.r11, P.y)), T.t1);
}

__device__ inline void add_p2p_xy(const ModelView &mv, const float2 m, float qx, float qy, double acc[kNumAcc]);

// icpPointToPoint.cpp:76,96-99,116-126,159: one inlier correspondence into the running sums
template <typename StartT>
__device__ inline void add_p2p(const IndexPtrs<StartT> &ix, const ModelView &mv, int cls, const Best &b, float qx,
                               float qy, double acc[kNumAcc])
{
    add_p2p_xy(mv, ix.pts[mv.base[cls] + b.pos], qx, qy, acc);
}

__device__ inline void add_p2p_xy(const ModelView &mv, const float2 m, float qx, float qy, double acc[kNumAcc])
{
    const double ax = (double)m.x - mv.cx, ay = (double)m.y - mv.cy;
    const double bx = (double)qx - mv.cx, by = (double)qy - mv.cy;
    acc[0] += 1.0;
    acc[1] += ax;
    acc[2] += ay;
    acc[3] += bx;
    acc[4] += by;
    // H[a][b] = sum q_t[a]*q_m[b] (:159).  Fused: one rounding per term instead of two -- the sums are compared with the reference's
    // two-pass form within a tolerance either way (it centres first, this shifts by the model's centroid and corrects in p2p_step)
    acc[5] = fma(bx, ax, acc[5]);
    acc[6] = fma(bx, ay, acc[6]);
    acc[7] = fma(by, ax, acc[7]);
    acc[8] = fma(by, ay, acc[8]);
}

// icpPointToPlane.cpp:61-82: one correspondence of the point-to-line step -- model point d = m, its normal n, template
// point s = the float query widened -- into the nine sums of A^T A | A^T b, A row = [n_y s_x - n_x s_y, n_x, n_y],
// b = n . (d - s).  No inlier gate, no classes (:55-77).
__device__ inline void add_p2l(const float2 m, const double2 nrm, float qx, float qy, double acc[kNumAcc])
{
    const double nx = nrm.x, ny = nrm.y;
    const double dx = (double)m.x, dy = (double)m.y;
    const double sx = (double)qx, sy = (double)qy;
    // (fused multiply-adds: the row and b = n . (d - s) with one rounding per term, the nine sums likewise)
    const double a0 = fma(ny, sx, -(nx * sy)), a1 = nx, a2 = ny;
    const double bb = fma(nx, dx - sx, ny * (dy - sy));
    acc[0] = fma(a0, a0, acc[0]);
    acc[1] = fma(a0, a1, acc[1]);
    acc[2] = fma(a0, a2, acc[2]);
    acc[3] = fma(a1, a1, acc[3]);
    acc[4] = fma(a1, a2, acc[4]);
    acc[5] = fma(a2, a2, acc[5]);
    acc[6] = fma(a0, bb, acc[6]);
    acc[7] = fma(a1, bb, acc[7]);
    acc[8] = fma(a2, bb, acc[8]);
}

// One point-to-line step from the nine sums S = {A^T A upper triangle, A^T b}: icpPointToPlane.cpp:80-107.  The 3x3
// system goes through the reference's Gauss-Jordan (matrix.cpp:420-508); R_ = I + [[0,-w],[w,0]] re-orthonormalised by
// svd -> U*V^T (:88-95) has the closed form [[1,-w],[w,1]] / sqrt(1 + w^2) (oracle: o_orthonormal_from_omega, pinned
// against the compiled matrix.cpp).  pose is updated in place when the system could be solved; returns the step's delta
// (0 with the pose unchanged when it could not: the reference falls out of the if at :85).
__device__ inline double p2l_step(const double S[kNumAcc], double pose[6])
{
    double A[9] = {S[0], S[1], S[2], S[1], S[3], S[4], S[2], S[4], S[5]};
    double b[3] = {S[6], S[7], S[8]};
    if (!solve3(A, b)) return 0.0; // :85
    const double w = b[0], nn = sqrt(1.0 + w * w);
    const double R_[4] = {1.0 / nn, -w / nn, w / nn, 1.0 / nn};
    const double t_[2] = {b[1], b[2]};
    const double r00 = pose[0], r01 = pose[1], r10 = pose[2], r11 = pose[3], t0 = pose[4], t1 = pose[5];
    pose[0] = R_[0] * r00 + R_[1] * r10; // :101-102 R = R_*R ; t = R_*t + t_
    pose[1] = R_[0] * r01 + R_[1] * r11;
    pose[2] = R_[2] * r00 + R_[3] * r10;
    pose[3] = R_[2] * r01 + R_[3] * r11;
    pose[4] = (R_[0] * t0 + R_[1] * t1) + t_[0];
    pose[5] = (R_[2] * t0 + R_[3] * t1) + t_[1];
    const double a0 = R_[0] - 1.0, a3 = R_[3] - 1.0;
    const double nr2 = a0 * a0 + R_[1] * R_[1] + R_[2] * R_[2] + a3 * a3;
    const double nt2 = t_[0] * t_[0] + t_[1] * t_[1];
    return sqrt(nr2 > nt2 ? nr2 : nt2); // :103
}

// ---- normals of a point-to-line model (icpPointToPlane.cpp:279-305, :340-349), from its cell index.
// A point-to-line model is one class, held as class 1 with oidx = all-index (GA then NGA, the order of M_normal).
constexpr int kMaxK = 16;

// The normal of the model point at position `pos_self` of the sorted array: its K nearest model points (itself included,
// n_nearest_around_point(i, 0, K)) found ring by ring around its own cell -- a few dozen candidates -- and kept ordered by
// (distance, all-index), the order a scan of the whole model in index order meets them; their scatter matrix; the direction of
// least spread.  (Rounds 1-3 scanned the whole model per point through an LDS tile: 1.5 ms for 10 k points.)
template <int K, typename StartT>
__device__ inline void normal_of_model_point(const ModelView &mv, int n, int pos_self, double *normals)
{
    const IndexPtrs<StartT> ix = make_ptrs<StartT>(mv.blob, mv);
    const float2 *pts = ix.pts + mv.base[1];
    const StartT *start = ix.start[1], *oidx = ix.oidx + mv.base[1];
    const Lattice &L = mv.lat;
    if (pos_self >= n) return;
    const float2 q = pts[pos_self];
    float        bd[K];
    int          bi[K], bp[K];
#pragma unroll
    for (int j = 0; j < K; ++j) bd[j] = FLT_MAX, bi[j] = 0x7fffffff, bp[j] = -1;
    const float fx = (q.x - L.x0) * L.inv_h, fy = (q.y - L.y0) * L.inv_h;
    const int   cx = clampi(ifloor(fx), 0, L.nx - 1), cy = clampi(ifloor(fy), 0, L.ny - 1);
    const bool  finite = q.x - q.x == 0.0f && q.y - q.y == 0.0f;
    auto        take = [&](int pos) {
        float d = dist2(pts[pos], q.x, q.y);
        int   id = (int)oidx[pos], pp = pos;
        if (!(d < bd[K - 1] || (d == bd[K - 1] && id < bi[K - 1]))) return;
#pragma unroll
        for (int s_ = 0; s_ < K; ++s_) {
            const bool  before = d < bd[s_] || (d == bd[s_] && id < bi[s_]);
            const float td = before ? bd[s_] : d;
            const int   ti = before ? bi[s_] : id, tp = before ? bp[s_] : pp;
            bd[s_] = before ? d : bd[s_];
            bi[s_] = before ? id : bi[s_];
            bp[s_] = before ? pp : bp[s_];
            d = td, id = ti, pp = tp;
        }
    };
    auto row_span = [&](int y, int x0, int x1) {
        if (x0 > x1) return;
        for (int pos = (int)start[y * L.nx + x0], e = (int)start[y * L.nx + x1 + 1]; pos < e; ++pos) take(pos);
    };
    // (a point with a non-finite coordinate sits in cell 0 and is nobody's neighbour at a finite distance: it scans everything
    // and gets whatever that arithmetic gives)
    // Squares of growing radius around the own cell; of each only what the one before left out: whole rows above and below it, the
    // two side spans of the rows beside it -- two look-ups a row whatever the radius.  (One cell per look-up, ring by ring, was fine
    // for walls -- ten neighbours within a cell or two -- and 1 ms for a model of scattered points: 0.6 m to the tenth neighbour at a
    // pitch of 0.25 m is a hundred look-ups, each a chain of dependent loads, and a straggler in the cloud's fringe thousands.)  The
    // radius grows by one up to 2, then by half of itself: a square is never more than 2.25 x the one that would have sufficed.
    int rp = -1;
    for (int r = 0;; r += (r < 2 ? 1 : r / 2)) {
        const int y_lo = max(cy - r, 0), y_hi = min(cy + r, L.ny - 1), x_lo = max(cx - r, 0), x_hi = min(cx + r, L.nx - 1);
        for (int y = y_lo; y <= y_hi; ++y) {
            if (rp >= 0 && y >= cy - rp && y <= cy + rp) { // beside the square already seen: what lies left and right of it
                row_span(y, x_lo, min(cx - rp - 1, x_hi));
                row_span(y, max(cx + rp + 1, x_lo), x_hi);
            } else {
                row_span(y, x_lo, x_hi);
            }
        }
        const bool  covers = x_lo == 0 && y_lo == 0 && x_hi == L.nx - 1 && y_hi == L.ny - 1;
        const float bound = (float)r * L.h - L.margin; // every point outside the square of radius r is farther than this
        if (covers || (finite && bp[K - 1] >= 0 && bound > 0.0f && bd[K - 1] < bound * bound)) break;
        rp = r;
    }
    double mx = 0, my = 0;
    int    k = 0;
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (bp[j] >= 0) {
            const float2 p = pts[bp[j]];
            mx += (double)p.x;
            my += (double)p.y;
            ++k;
        }
    mx /= (double)k;
    my /= (double)k;
    double sxx = 0, sxy = 0, syy = 0;
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (bp[j] >= 0) {
            const float2 p = pts[bp[j]];
            const double dx = (double)p.x - mx, dy = (double)p.y - my;
            sxx += dx * dx;
            sxy += dx * dy;
            syy += dy * dy;
        }
    const double th = 0.5 * atan2(2.0 * sxy, sxx - syy); // eigenvector of the smaller eigenvalue of [[sxx,sxy],[sxy,syy]]
    const int    i = (int)oidx[pos_self];
    normals[2 * i] = -sin(th);
    normals[2 * i + 1] = cos(th);
}

struct FitArgs {
    const double2 *pts;
    const int     *scan_off;
    const int     *scan_nga;
    double        *R;        // the poses found, 4 + 2 doubles per scan
    double        *t;
    const double  *R0, *t0;  // the poses the fits start from (R, t themselves for a fit in place)
    slam_icp_result *result;
    double        *trace;
    int            max_iter;
    double         min_delta;
    double         indist;
    double        *step_pose; // nullable; per scan 6 doubles: R,t as the last executed step found them
    long long     *stamps; // diagnostic only (SLAM_ICP_STAMPS=1): per scan, cycles in [search, reduce, barrier, solve]
    // Two search forms per scan: the ring search runs at least the first switch_iter iterations (by then a scan
    // is normally within the certified radius of the halo lists), the list sweeps the rest.  One launch does
    // both (icp_fit_fused_kernel); as two launches of icp_fit_kernel (SLAM_ICP_SPLIT=1, measurements) the
    // hand-over goes through state[scan] = iterations done, or -1 when the scan finished in the ring search.
    int           *state;
    int            phase;  // icp_fit_kernel: 0 = the only launch, 1 = ring search of two launches, 2 = list sweeps of two
    int            switch_iter;
    int            far_div;       // hand over once at most n / far_div queries are beyond the lists' certified radius
    const int     *only;          // nullable; per scan: the workgroup-per-scan kernels run scan s only if only[s] != 0 (the
                                  // scans a spread launch could not finish, icp_single.hip)
    int           *redo_mirror;   // nullable; spread form: the redo flag of every scan once more, where the HOST reads it (pinned memory:
                                  // slam_icp_fit learns without a copy of its own whether the one-workgroup form had to take the scan)
    unsigned       spread_tag;    // spread form: the launch's tag base -- granule tags are spread_tag + iteration + 1, a scan's abort word
                                  // holds spread_tag itself when raised; no launch repeats another's (icp_single.hip: no fill between launches)
};


__device__ inline void fill_lds(unsigned char *dst, const unsigned char *blob, unsigned bytes)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(blob);
    uint4       *d4 = reinterpret_cast<uint4 *>(dst);
    for (unsigned i = threadIdx.x; i < bytes / 16u; i += kBlock) d4[i] = src[i];
}


// The state of one scan's fit that outlives a run of iterations: the pose (the same in every lane) and the
// counters that go into slam_icp_result.
struct FitState {
    double r00, r01, r10, r11, t0, t1, delta;
    int    iters, n_corr;
    bool   hand_over;
};

constexpr int kSpreadMinParts = 16; // the spread form takes batches that leave every scan at least this many workgroups

// icp_single.hip: few scans (one, in the reference's own usage), each spread over many workgroups of one
// persistent launch
// *redo_flags: device array [n_scans], non-zero for a scan the launch could not finish (its R, t untouched): the caller
// enqueues the workgroup-per-scan form behind it with FitArgs::only = that array
int launch_fit_spread(slam_icp *h, const FitArgs &fa, int n_scans, hipStream_t st, const int **redo_flags);
bool takes_spread_form(const slam_icp *h, int n_scans); // icp.hip

} // namespace icp
} // namespace slam
