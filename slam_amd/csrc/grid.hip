// grid.hip -- occupancy-grid update on gfx950 behind the C-ABI.
//
// Reference path (under /root/reference/mls):
//   MLS::addToOccupancy  src/mls.cpp:59-150   endpoint binning: obstacle +1.0, ground -0.3
//   Grid::operator()     include/mls/mls.h:76-85   toroidal wrap
//   MLS::setPose         src/mls.cpp:408-479  rolling window
// plus the north-star Bresenham free-space traversal, which the reference does
// not have; its definition is oracle/slam_oracle.c
// ogrid_raycast and the closed form in bres_y() below.
//
// Data layout in HBM (DESIGN.md "Grid"): two int32 planes [hits | misses] of
// size_x*size_y cells each, contiguous (one RCCL all-reduce merges both), in
// toroidal STORAGE order; a f64 evidence plane (Cluster::num_pts) and an int8
// occupancy plane derived from the counts by finalize.  The reference's
// per-cell `Cell` object (vector<Cluster> + deque, mls.h:37-51) is not
// reproduced: in occupancy mode only clusters[0].num_pts and `drivable` are
// ever touched.
//
// Raycast, tiled implementation: a pre-pass turns every beam into integer
// (x0,y0,x1,y1) cells once; each workgroup owns one 128x128-cell tile for a
// slice of the beams, accumulates hit/miss counts for that tile in LDS (ds_add,
// 16+16 bit packed per cell), and writes the tile back with coalesced global
// atomics -- one per touched cell per workgroup instead of one per traversed
// cell per beam.  Each beam's cells inside a tile come from the closed form of
// the integer Bresenham line, so clipping to the tile cannot change them.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"

using namespace slam;

namespace {

struct GridView {
    int      sx, sy;       // window size in cells
    int      ox, oy;       // toroidal origin (Grid::origin_x/y, mls.h:69-70)
    double   res;
    double   max_range;
    double   pose_x, pose_y;
    int      rolling;
    int32_t *hits;         // [sx*sy] storage order
    int32_t *misses;       // [sx*sy] storage order
    unsigned long long *updates;
    int     *dirty;        // [4] storage rows {lowest, -highest}: [0..1] touched since the counts were last reset (or folded),
                           // [2..3] rows whose counts changed since the last finalize by something other than an update (a reset)
    const int32_t *acc_hits, *acc_misses; // nullable: counts folded away by slam_grid_fold (merged totals)
};

__device__ inline int storage_index(const GridView &g, int x, int y)
{
    int ix = x + g.ox, iy = y + g.oy; // mls.h:76-85
    if (ix >= g.sx) ix -= g.sx;
    if (iy >= g.sy) iy -= g.sy;
    return ix + g.sx * iy;
}

// (int)(v/res + size/2) of mls.cpp:77-78; false when an int cannot hold it
// (undefined in the reference; x86 gives INT_MIN there, i.e. "skip").
__device__ inline bool cell_coord(float v, double res, int half, int *out)
{
    const double f = __dadd_rn(__ddiv_rn((double)v, res), (double)half);
    if (!(f > -2147483648.0 && f < 2147483648.0)) return false;
    *out = (int)f; // truncation toward zero
    return true;
}

// mls.cpp:77-90: window cell of a point, or false when the range gate or the
// bounds test (with its `y >= size_x` quirk) drops it.
__device__ inline bool point_cell(const GridView &g, float px, float py, int *cx, int *cy)
{
    int x, y;
    if (!cell_coord(px, g.res, g.sx / 2, &x)) return false;
    if (!cell_coord(py, g.res, g.sy / 2, &y)) return false;
    double rng;
    if (g.rolling) {
        // mls.cpp:82: float expression, float sqrt, widened for the compare
        rng = (double)__fsqrt_rn(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py)));
    } else {
        const double rx = g.pose_x - (double)px, ry = g.pose_y - (double)py; // :84-86
        rng = __dsqrt_rn(__dadd_rn(__dmul_rn(rx, rx), __dmul_rn(ry, ry)));
    }
    if (x < 0 || y < 0 || x >= g.sx || y >= g.sx || rng > g.max_range) return false; // :90
    if (y >= g.sy) return false;
    *cx = x;
    *cy = y;
    return true;
}

constexpr int kUpdateSlots = 1024;

// Rows of the planes (storage order) that hold counts: a merge over the GPUs moves only these
// (slam_mi355x_rccl.h).  Wave minimum, then the pair of atomics only where they would change the range: thousands
// of wavefronts hitting one word cost 11 ns each (the lesson of the update counter below), so the wavefront looks
// first -- a stale look only costs a superfluous atomic, the atomic itself is what counts.
__device__ inline void mark_dirty_rows(int *dirty, int row_lo, int row_hi /* -1: none */)
{
    int lo = row_hi >= 0 ? row_lo : 0x7fffffff, nhi = row_hi >= 0 ? -row_hi : 0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off));
        nhi = min(nhi, __shfl_xor(nhi, off));
    }
    if ((threadIdx.x & 63) == 0 && lo != 0x7fffffff) {
        if (lo < __hip_atomic_load(&dirty[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&dirty[0], lo);
        if (nhi < __hip_atomic_load(&dirty[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&dirty[1], nhi);
    }
}

// Counter of cell updates: wave reduction, then one atomic per wavefront into
// one of kUpdateSlots slots (same-address device atomics retire at ~11 ns each,
// so thousands of wavefronts must not share one word); the reader sums the slots.
__device__ inline void block_add_updates(unsigned long long *slots, unsigned n)
{
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor((int)n, off);
    if ((threadIdx.x & 63) == 0 && n) {
        const unsigned wave = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) + blockIdx.y * 7919u;
        atomicAdd(&slots[wave & (kUpdateSlots - 1)], (unsigned long long)n);
    }
}

// ------------------------------------------------------------- endpoints
// MLS::addToOccupancy's two loops (mls.cpp:73-106 obstacle points, :110-142 ground points) as counts: one workgroup takes
// kEndpointsPerBlock consecutive points, lanes take consecutive points (coalesced, and runs of equal cells stay inside a
// wavefront).  The rows it touched and the number of points it accepted leave the workgroup ONCE: the range words and the
// update counter are single addresses, and same-address atomics retire one at a time (11 ns each) -- with one pair per
// wavefront, the thousands of wavefronts that start together right after a reset (range empty: every one of them widens
// it) cost 70 us on config 2's 276 k endpoints, four times the kernel's own work (round 4, bench.py's endpoint leg).
constexpr int kEndpointsPerBlock = 1024;

__global__ __launch_bounds__(256) void endpoints_kernel(GridView g, const float *obs, int n_obs,
                                                        const float *gnd, int n_gnd, int stride)
{
    __shared__ int      s_lo[4], s_nhi[4];
    __shared__ unsigned s_cnt[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int       lo = 0x7fffffff, nhi = 0x7fffffff; // lowest row, minus the highest
    unsigned  cnt = 0;
#pragma unroll
    for (int k = 0; k < kEndpointsPerBlock / 256; ++k) {
        const int i = blockIdx.x * kEndpointsPerBlock + k * 256 + (int)threadIdx.x;
        bool      did = false;
        int       key = -1;
        if (i < n_obs + n_gnd) {
            const bool   is_obs = i < n_obs;
            const float *p = is_obs ? obs + (size_t)i * stride : gnd + (size_t)(i - n_obs) * stride;
            int cx, cy;
            if (point_cell(g, p[0], p[1], &cx, &cy)) {
                const int s = storage_index(g, cx, cy), row = s / g.sx;
                did = true;
                key = (is_obs ? 0 : g.sx * g.sy) + s; // word of [hits | misses] this point counts in
                lo = min(lo, row);
                nhi = min(nhi, -row);
                ++cnt;
            }
        }
        // Consecutive points of a scan or of a lidar ring fall into the same cell in runs (beams 0.25 degrees apart are
        // centimetres apart at the wall): the first lane of a run adds the run's length, one atomic per run instead of one
        // per point (mls.cpp:99 / :135 as counts: integer sums, any grouping gives the same planes).
        const int                prev = __shfl_up(key, 1);
        const bool               head = did && (lane == 0 || prev != key);
        const unsigned long long heads = __ballot(head || !did); // a dropped point ends a run too
        if (head) {
            const unsigned long long after = lane == 63 ? 0ull : heads >> (lane + 1);
            const int                len = after ? __builtin_ctzll(after) + 1 : 64 - lane;
            atomicAdd(&g.hits[key], len);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off));
        nhi = min(nhi, __shfl_xor(nhi, off));
        cnt += (unsigned)__shfl_xor((int)cnt, off);
    }
    if (lane == 0) s_lo[wave] = lo, s_nhi[wave] = nhi, s_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        nhi = min(min(s_nhi[0], s_nhi[1]), min(s_nhi[2], s_nhi[3]));
        cnt = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (lo != 0x7fffffff) { // look first: a stale look only costs a superfluous atomic (mark_dirty_rows)
            if (lo < __hip_atomic_load(&g.dirty[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&g.dirty[0], lo);
            if (nhi < __hip_atomic_load(&g.dirty[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&g.dirty[1], nhi);
        }
        if (cnt) atomicAdd(&g.updates[(blockIdx.x * 7u + blockIdx.y * 7919u) & (kUpdateSlots - 1)], (unsigned long long)cnt);
    }
}

// --------------------------------------------------------------- raycast
struct Beam { // integer Bresenham endpoints in window cells; x0 < 0 marks a dropped beam
    short x0, y0, x1, y1;
};

constexpr int kTile = 128;             // cells per tile side
#ifndef SLAM_WALK_UNROLL
#define SLAM_WALK_UNROLL 4
#endif
constexpr int kWalkUnroll = SLAM_WALK_UNROLL; // steps per trip of the raycast walk loop
// Lanes of a block walk STAGGERED: lane j takes (j % kWalkStagger) single steps before the lock-step trips begin (round 5).  The 64
// beams of a block are adjacent beams of ONE scan: near the sensor they run through the same cells, and in lock-step from the same
// start they are on the same cell in the same instruction -- LDS atomics of one instruction to one address are served one after the
// other (bank-conflict share 55 %, VALU busy 35 %).  Staggered, neighbours are a cell or more apart along the beam when they add:
// the raycast call of config 2 0.1325 -> 0.0983 ms with one workgroup per CU, 0.110 -> 0.090 with two (VALU busy 56 %); the pipelined
// step 0.304 -> 0.295 ms.  Measured (tools/exp/stagger_ab.sh, ms per call with one workgroup per CU): steps ahead 2 / 3 / 4 / 6 / 8 /
// 12 / 16 / 32 -> 0.120 / 0.112 / 0.105 / 0.100 / 0.098 / 0.100 / 0.104 / 0.130; whole trips of delay instead (4 steps each), 2 / 4 /
// 6 / 8 / 16-fold -> 0.115 / 0.103 / 0.103 / 0.106 / 0.122.  Bit-exact either way: the adds commute.
#ifndef SLAM_WALK_STAGGER
#define SLAM_WALK_STAGGER 8
#endif
constexpr int kWalkStagger = SLAM_WALK_STAGGER;
constexpr int kTileStride = kTile + 1; // LDS row pitch in words: vertical neighbours fall on adjacent banks
constexpr int kTileThreads = 1024;
constexpr int kChunk = 1024;           // beams per pre-pass workgroup
constexpr int kBlock = 64;             // beams per culling block = one wavefront

__device__ inline Beam make_beam(const GridView &g, float ox, float oy, float ex, float ey)
{
    Beam b;
    b.x0 = -1;
    b.y0 = b.x1 = b.y1 = 0;
    int x1, y1, x0, y0;
    if (!point_cell(g, ex, ey, &x1, &y1)) return b;
    if (!cell_coord(ox, g.res, g.sx / 2, &x0)) return b;
    if (!cell_coord(oy, g.res, g.sy / 2, &y0)) return b;
    if (x0 < 0 || y0 < 0 || x0 >= g.sx || y0 >= g.sy) return b;
    b.x0 = (short)x0;
    b.y0 = (short)y0;
    b.x1 = (short)x1;
    b.y1 = (short)y1;
    return b;
}

// Stores the lane's integer beam and, per wavefront, the bounding box of its
// 64 beams (the culling unit of the tiled raycast: 64 consecutive beams of a
// scan are a narrow wedge).  Wave reduction only: no LDS, no atomics.
__device__ inline void store_beam_and_box(const Beam &b, bool in_range, int i, Beam *beams, int4 *block_box)
{
    if (in_range) beams[i] = b;
    const bool ok = in_range && b.x0 >= 0;
    int lo_x = ok ? min((int)b.x0, (int)b.x1) : 0x7fffffff, lo_y = ok ? min((int)b.y0, (int)b.y1) : 0x7fffffff;
    int hi_x = ok ? max((int)b.x0, (int)b.x1) : -1, hi_y = ok ? max((int)b.y0, (int)b.y1) : -1;
    for (int off = 32; off > 0; off >>= 1) {
        lo_x = min(lo_x, __shfl_xor(lo_x, off));
        lo_y = min(lo_y, __shfl_xor(lo_y, off));
        hi_x = max(hi_x, __shfl_xor(hi_x, off));
        hi_y = max(hi_y, __shfl_xor(hi_y, off));
    }
    if ((threadIdx.x & 63) == 0) block_box[i >> 6] = make_int4(lo_x, lo_y, hi_x, hi_y);
}

__global__ __launch_bounds__(kChunk) void beams_from_rays_kernel(GridView g, const float2 *origin,
                                                                 const float2 *end, int n, Beam *beams,
                                                                 int4 *block_box)
{
    const int  i = blockIdx.x * kChunk + threadIdx.x;
    const bool in = i < n;
    Beam       b;
    b.x0 = -1;
    b.y0 = b.x1 = b.y1 = 0;
    if (in) {
        const float2 o = origin[i], e = end[i];
        b = make_beam(g, o.x, o.y, e.x, e.y);
    }
    store_beam_and_box(b, in, i, beams, block_box);
}

__global__ __launch_bounds__(kChunk) void beams_from_scans_kernel(GridView g, const double2 *pts,
                                                                  const int *scan_off, int n_scans,
                                                                  const double *R, const double *t, int n,
                                                                  Beam *beams, int4 *block_box)
{
    const int  i = blockIdx.x * kChunk + threadIdx.x;
    const bool in = i < n;
    Beam       b;
    b.x0 = -1;
    b.y0 = b.x1 = b.y1 = 0;
    if (in) {
        // scan of point i: the last s with scan_off[s] <= i.  Scans of a batch are about the same size, so the
        // proportional guess is right or one off: a few steps from there instead of log2(n_scans) dependent loads
        // (the bisection below takes over for ragged batches)
        int lo = (int)(((long long)i * n_scans) / max(n, 1));
        lo = min(max(lo, 0), n_scans - 1);
        int steps = 0;
        while (steps < 4 && scan_off[lo] > i) --lo, ++steps;
        while (steps < 4 && lo + 1 < n_scans && scan_off[lo + 1] <= i) ++lo, ++steps;
        if (scan_off[lo] > i || (lo + 1 < n_scans && scan_off[lo + 1] <= i)) {
            lo = 0;
            int hi = n_scans - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (scan_off[mid] <= i)
                    lo = mid;
                else
                    hi = mid - 1;
            }
        }
        const double *Rs = R + 4 * (size_t)lo, *ts = t + 2 * (size_t)lo;
        const double2 P = pts[i];
        // end point formed as icpPointToPoint.cpp:69-70 forms its query; a rolling window is centred on
        // curPose, so map-frame points are taken relative to it (mls.cpp:36-47 shifts the cloud the same way)
        const double cx = g.rolling ? g.pose_x : 0.0, cy = g.rolling ? g.pose_y : 0.0;
        const double gx = __dadd_rn(__dadd_rn(__dmul_rn(Rs[0], P.x), __dmul_rn(Rs[1], P.y)), ts[0]);
        const double gy = __dadd_rn(__dadd_rn(__dmul_rn(Rs[2], P.x), __dmul_rn(Rs[3], P.y)), ts[1]);
        const float  ex = (float)(g.rolling ? __dsub_rn(gx, cx) : gx), ey = (float)(g.rolling ? __dsub_rn(gy, cy) : gy);
        const float  ox = (float)(g.rolling ? __dsub_rn(ts[0], cx) : ts[0]);
        const float  oy = (float)(g.rolling ? __dsub_rn(ts[1], cy) : ts[1]);
        b = make_beam(g, ox, oy, ex, ey);
    }
    store_beam_and_box(b, in, i, beams, block_box);
}

// one global atomic per traversed cell (baseline implementation)
__global__ __launch_bounds__(256) void raycast_global_kernel(GridView g, const Beam *beams, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned  did = 0;
    int       rlo = 0, rhi = -1;
    if (i < n) {
        const Beam b = beams[i];
        if (b.x0 >= 0) {
            const int dx = abs(b.x1 - b.x0), dy = abs(b.y1 - b.y0);
            const int sx = b.x1 > b.x0 ? 1 : -1, sy = b.y1 > b.y0 ? 1 : -1;
            int x = b.x0, y = b.y0;
            if (dx >= dy) {
                int e = dx;
                for (int k = 0; k < dx; ++k) {
                    atomicAdd(&g.misses[storage_index(g, x, y)], 1);
                    x += sx;
                    e += 2 * dy;
                    if (e >= 2 * dx) {
                        y += sy;
                        e -= 2 * dx;
                    }
                }
                did = dx + 1;
            } else {
                int e = dy;
                for (int k = 0; k < dy; ++k) {
                    atomicAdd(&g.misses[storage_index(g, x, y)], 1);
                    y += sy;
                    e += 2 * dx;
                    if (e >= 2 * dy) {
                        x += sx;
                        e -= 2 * dy;
                    }
                }
                did = dy + 1;
            }
            atomicAdd(&g.hits[storage_index(g, b.x1, b.y1)], 1);
            const int ra = storage_index(g, 0, b.y0) / g.sx, rb = storage_index(g, 0, b.y1) / g.sx;
            const bool wraps = (ra <= rb) != (b.y0 <= b.y1); // the beam crosses the toroidal seam: every row may be touched
            rlo = wraps ? 0 : min(ra, rb);
            rhi = wraps ? g.sy - 1 : max(ra, rb);
        }
    }
    mark_dirty_rows(g.dirty, rlo, rhi);
    block_add_updates(g.updates, did);
}

// Work list for the tiled raycast, built without atomics (so its order is
// deterministic): for every tile the 64-beam blocks whose bounding box overlaps it.
//   tile_items_wg : one workgroup per tile, ballot + popcount over the block boxes; the ids go to the tile's own
//                row of `items` (n_blocks ids wide, so no prefix sum is needed first), the count to cnt[], and the
//                tile's cursor (the next block of its list to hand out) back to 0
//   the raycast workgroups take blocks from a tile's list through that cursor, a chunk at a time.
//   Grids of more than kMaxLdsTiles tiles keep the three-kernel form (count, scan, fill).

__device__ inline bool box_overlaps_tile(const int4 cb, int tx0, int ty0, int tx1, int ty1)
{
    return cb.z >= tx0 && cb.x <= tx1 && cb.w >= ty0 && cb.y <= ty1; // empty boxes have z = w = -1
}

constexpr int kMaxLdsTiles = 2048; // segment offsets of that many tiles fit beside the LDS tile (8 KB)

// Grids of more than kMaxLdsTiles tiles.  FILL: 0 = count only, 1 = write the ids at item_off[t]
template <int FILL>
__global__ __launch_bounds__(256) void tile_items_kernel(const int4 *block_box, int n_blocks, int tiles_x,
                                                         int n_tiles, int sx, int sy, int *cnt,
                                                         const int *item_off, int *items)
{
    const int t = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (t >= n_tiles) return;
    const int tx0 = (t % tiles_x) * kTile, ty0 = (t / tiles_x) * kTile;
    const int tx1 = min(tx0 + kTile, sx) - 1, ty1 = min(ty0 + kTile, sy) - 1;
    int       c = 0;
    const int base_out = FILL == 1 ? item_off[t] : 0;
    for (int base = 0; base < n_blocks; base += 256) { // four independent box loads in flight per lane
        int4 cb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = base + 64 * j + lane;
            cb[j] = ch < n_blocks ? block_box[ch] : make_int4(0, 0, -1, -1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool               ov = box_overlaps_tile(cb[j], tx0, ty0, tx1, ty1);
            const unsigned long long m = __ballot(ov);
            if (FILL && ov) items[base_out + c + __popcll(m & ((1ull << lane) - 1ull))] = base + 64 * j + lane;
            c += __popcll(m);
        }
    }
    if (FILL != 1 && lane == 0) cnt[t] = c;
}

// The single-pass form with one workgroup of 16 wavefronts per tile: every lane looks at up to four block boxes per
// trip (all loads in flight at once), the wavefronts' match counts go through LDS, and the ids land in the tile's row
// in ascending block order as before (17 dependent trips of one wavefront per tile took 11 us on config 2).
__global__ __launch_bounds__(1024) void tile_items_wg_kernel(const int4 *block_box, int n_blocks, int tiles_x, int sx,
                                                             int sy, int *cnt, int *items, int *cursor)
{
    __shared__ int s_cnt[4][16], s_base;
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        s_base = 0;
        cursor[t] = 0;                       // the tile's list is handed out from its start
        if (t == 0) cursor[gridDim.x] = 0;   // behind the cursors: tile write-backs of the launch (slam_grid_raycast_stats)
    }
    const int tx0 = (t % tiles_x) * kTile, ty0 = (t / tiles_x) * kTile;
    const int tx1 = min(tx0 + kTile, sx) - 1, ty1 = min(ty0 + kTile, sy) - 1;
    const int base_out = t * n_blocks;
    for (int base = 0; base < n_blocks; base += 4096) {
        int4 cb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = base + 1024 * j + tid;
            cb[j] = ch < n_blocks ? block_box[ch] : make_int4(0, 0, -1, -1);
        }
        unsigned long long m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = __ballot(box_overlaps_tile(cb[j], tx0, ty0, tx1, ty1));
            if (lane == 0) s_cnt[j][wave] = __popcll(m[j]);
        }
        __syncthreads(); // the counts of this trip (and s_base of the previous one) are visible
        int total = 0; // ids are ordered by (j, wave, lane) = ascending block id
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (int w = 0; w < 16; ++w) {
                const int c = s_cnt[j][w];
                total += c;
            }
        int run = s_base;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int mine = run;
            for (int w = 0; w < 16; ++w) {
                const int c = s_cnt[j][w];
                mine += w < wave ? c : 0;
                run += c;
            }
            if ((m[j] >> lane) & 1ull)
                items[base_out + mine + __popcll(m[j] & ((1ull << lane) - 1ull))] = base + 1024 * j + tid;
        }
        __syncthreads(); // everyone has read s_cnt and s_base
        if (tid == 0) s_base += total;
    }
    __syncthreads();
    if (tid == 0) cnt[t] = s_base;
}

// grids of more than kMaxLdsTiles tiles: exclusive prefix of the tile counts (where each tile's list starts, the total
// behind the last), and the cursors back to 0
__global__ __launch_bounds__(1024) void tile_scan_kernel(const int *cnt, int n_tiles, int *item_off, int *cursor)
{
    __shared__ int carry;
    __shared__ int wsum[16];
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + tid;
        const int v0 = i < n_tiles ? cnt[i] : 0;
        int       x0 = v0;
        for (int off = 1; off < 64; off <<= 1) {
            const int y0 = __shfl_up(x0, off);
            if ((tid & 63) >= off) x0 += y0;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = x0;
        __syncthreads();
        int w0 = 0;
        for (int w = 0; w < (tid >> 6); ++w) w0 += wsum[w];
        const int in0 = carry + w0 + x0;
        if (i < n_tiles) {
            item_off[i] = in0 - v0;
            cursor[i] = 0;
        }
        __syncthreads();
        if (tid == 1023) carry = in0;
        __syncthreads();
    }
    if (tid == 0) {
        item_off[n_tiles] = carry;
        cursor[n_tiles] = 0; // tile write-backs of the launch
    }
}

// floor(num / den) for 0 <= num < 2^31, 1 <= den < 2^16, when the quotient is
// below 2^16 (larger quotients saturate to 65536: callers clamp against <= 32767).
// One float multiply + an exact remainder fix instead of an integer division:
// with rden within 1 ulp of 1/den the float quotient is off by < 2^16 * 2^-21,
// so truncation lands on floor or floor +- 1 and the remainder test repairs it.
__device__ inline int floor_div_small(int num, int den, float rden)
{
    const float qf = (float)num * rden;
    if (qf >= 65536.0f) return 65536;
    int       q = (int)qf;
    const int r = (int)((unsigned)num - (unsigned)__mul24(q, den));
    q += (r >= den) - (r < 0);
    return q;
}

// Tiled raycast.  A persistent workgroup works on ONE tile at a time: it zeroes a 128x128 tile of packed
// (hits<<16 | misses) counters in LDS, takes 64-beam blocks from the tile's work list a chunk at a time (the tile's
// cursor: one global atomic per chunk), its 16 wavefronts pull the chunk's blocks from an LDS counter, and it keeps
// accumulating in the same LDS tile for as long as the tile has blocks left -- the tile is written back (coalesced
// global atomics) once, when the workgroup leaves it.  Workgroups start spread over the tiles in proportion to the
// tiles' lists (a prefix over the tile counts in LDS) and, when their tile runs dry, move to the tile with the most
// blocks left.  (Round 2 cut the lists into fixed segments of one tile, each zeroed and written back by whoever took it:
// 8.5 write-backs per tile on config 2 -- 30 % of the kernel's instructions and 40 MB of atomic traffic per launch;
// a workgroup that stays is one write-back per workgroup and tile.)
// Within a block every lane owns one beam: it clips the beam's Bresenham step
// range [0, du] to the tile exactly -- the cell of step i has the closed form
//   (u0 + su*i, v0 + sv*floor((2*i*dv + du) / (2*du))),
// so the clipped range and the error term at its first step are two small
// integer divisions, not a walk -- and then all lanes step their beams in
// lock-step with the integer error update (4 integer ops + one ds_add per
// cell).  Clipping by closed form is what makes the tiling invisible in the
// result: the cells are exactly those of the unclipped line.
constexpr int kChunkMin = 8;      // (the visit's chunk log in LDS is sized for it)
constexpr int kChunkDefault = 16; // blocks a workgroup takes from a tile's list at a time, where the caller leaves it to the library
constexpr int kMaxAccBlocks = 1023; // blocks accumulated in the packed LDS counters before a write-back is forced: 64 * 1023 misses of the
                                    // sensor's cell cannot carry into the hit half

template <bool MERGE>
__global__ __launch_bounds__(kTileThreads, 8) void raycast_tiled_kernel(GridView g, const Beam *beams, int n,
                                                                        const int *items, const int *item_off,
                                                                        int n_tiles, int *cursor, int tiles_x, int chunk,
                                                                        int ablate_arg, const int *cnt, int item_stride)
{
#ifdef SLAM_MEASURE // timing experiments only (tools/ablate.sh): bits switch parts of the kernel off -- wrong counts
    const int ablate = ablate_arg;
#else
    constexpr int ablate = 0;
    (void)ablate_arg;
#endif
    __shared__ __attribute__((aligned(16))) unsigned tile[kTile * kTileStride];
    __shared__ int s_ticket, s_done, s_end;
    __shared__ unsigned long long s_desc[kMaxAccBlocks / kChunkMin]; // the visit's chunk log
    __shared__ int s_off[kMaxLdsTiles + 1], s_wsum[kTileThreads / 64], s_carry;
    __shared__ unsigned long long s_best[kTileThreads / 64];

    const int tid = threadIdx.x, lane = tid & 63;
    // (item_off != null: a grid of more than kMaxLdsTiles tiles; its lists lie one behind the other and their starts were
    // scanned by tile_scan_kernel.  Otherwise every tile has a row of its own and the prefix is worked out here.)
    const bool in_lds = item_off == nullptr;
    if (in_lds) {
        if (tid == 0) s_carry = 0;
        __syncthreads();
        for (int base = 0; base < n_tiles; base += kTileThreads) {
            const int i = base + tid;
            const int v = i < n_tiles ? cnt[i] : 0;
            int       x = v;
            for (int off = 1; off < 64; off <<= 1) {
                const int y = __shfl_up(x, off);
                if (lane >= off) x += y;
            }
            if (lane == 63) s_wsum[tid >> 6] = x;
            __syncthreads();
            int w = 0;
            for (int k = 0; k < (tid >> 6); ++k) w += s_wsum[k];
            const int incl = s_carry + w + x;
            if (i < n_tiles) s_off[i] = incl - v;
            __syncthreads();
            if (tid == kTileThreads - 1) s_carry = incl;
            __syncthreads();
        }
        if (tid == 0) s_off[n_tiles] = s_carry;
        __syncthreads();
    }
    const auto list_start = [&](int t) { return in_lds ? s_off[t] : item_off[t]; };
    const int  total = list_start(n_tiles);
    unsigned   did = 0;
    int        d_lo = 0x7fffffff, d_hi = -1; // storage rows this lane wrote back
    if (total <= 0 || (ablate & 8)) {
        mark_dirty_rows(g.dirty, d_lo, d_hi);
        block_add_updates(g.updates, did);
        return;
    }
    // where this workgroup starts: the tile that holds its share of all blocks (the last t with list_start(t) <= pos)
    int cur;
    {
        const int pos = (int)(((long long)blockIdx.x * total) / gridDim.x);
        int       lo = 0, hi = n_tiles - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (list_start(mid) <= pos)
                lo = mid;
            else
                hi = mid - 1;
        }
        cur = lo;
    }
    for (int i = tid * 4; i < kTile * kTileStride; i += kTileThreads * 4)
        *reinterpret_cast<uint4 *>(&tile[i]) = make_uint4(0u, 0u, 0u, 0u);

    // writes the LDS tile of tile t back (coalesced: consecutive lanes -> consecutive x of one row) and zeroes it
    const auto flush = [&](int t) {
        const int tx0 = (t % tiles_x) * kTile, ty0 = (t / tiles_x) * kTile;
        for (int i = (ablate & 2) ? kTile * kTile : tid; i < kTile * kTile; i += kTileThreads) {
            const int      lx = i & (kTile - 1), ly = i / kTile;
            const unsigned v = tile[ly * kTileStride + lx];
            if (!v) continue;
            tile[ly * kTileStride + lx] = 0u;
            const int s = storage_index(g, tx0 + lx, ty0 + ly);
            if (ablate & 128) {
                // measurement only (tools/raycast_interleave.sh): what the write-back would cost with {misses, hits} interleaved
                // per cell -- ONE 8-byte atomic per touched cell, the planes' memory taken as cells x 8 bytes (wrong counts)
                atomicAdd(reinterpret_cast<unsigned long long *>(g.hits) + s, ((unsigned long long)(v >> 16) << 32) | (v & 0xffffu));
            } else {
                if (v & 0xffffu) atomicAdd(&g.misses[s], (int)(v & 0xffffu));
                if (v >> 16) atomicAdd(&g.hits[s], (int)(v >> 16));
            }
            int row = ty0 + ly + g.oy; // the storage row of s (storage_index without its division)
            row -= row >= g.sy ? g.sy : 0;
            d_lo = min(d_lo, row);
            d_hi = max(d_hi, row);
        }
        if (tid == 0) atomicAdd(&cursor[n_tiles], 1); // statistics: tile write-backs of the launch
    };

    // One VISIT = the stay of this workgroup on one tile between two write-backs.  Within a visit there is no barrier: a
    // wavefront draws a ticket T from an LDS counter; ticket T is block T % chunk of the visit's chunk T / chunk, and chunk c
    // is the range [base_c, base_c + np_c) of the tile's list that a global atomic on the tile's cursor handed out,
    // published as one 64-bit LDS word {1, np, base}.  Chunks 0 and 1 come from one atomic at the start of the visit; the
    // wavefront that draws the first ticket of chunk c fetches chunk c + 2 -- two chunks of walks ahead of its use, the
    // atomic in flight while the wavefront walks its own block -- once it has seen chunks c and c + 1 full: fetches are
    // therefore issued one after the other's return, the bases grow with c, and the first chunk that is not full ends the
    // list for every later ticket (s_end).  Descriptors are a log, not a ring (one per chunk of the visit: never overwritten
    // while anybody may read them); a visit ends where the log does (kMaxAccBlocks blocks, the limit of the packed LDS
    // counters) or where the tile's list does.  A wavefront never waits while it owes a publication (settle).
    const int  kMaxChunks = kMaxAccBlocks / chunk; // chunks per visit (host: chunk <= kMaxAccBlocks / 2, so at least two)
    const auto publish = [&](int c, int base, int np) { // lane 0 of the fetching wavefront
        __hip_atomic_store(&s_desc[c], (1ull << 48) | ((unsigned long long)(unsigned)np << 32) | (unsigned)base, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
        if (np < chunk) { // the tile's list ends in (np > 0) or before (np == 0) this chunk
            atomicMin(&s_end, np > 0 ? c + 1 : c);
            s_done = 1;
        }
    };
    for (;;) {
        __syncthreads(); // the last visit's walks are in the tile and written back; nobody reads its descriptors any more
        for (int c = tid; c < kMaxChunks; c += kTileThreads) s_desc[c] = 0ull;
        if (tid == 0) {
            s_ticket = 0;
            s_done = 0;
            s_end = kMaxChunks;
        }
        __syncthreads();
        const int t = cur;
        if (tid == 0) { // the visit's first two chunks
            const int have_t = cnt[t];
            const int k = atomicAdd(&cursor[t], 2 * chunk);
            const int np0 = max(0, min(chunk, have_t - k));
            publish(0, k, np0);
            if (np0 == chunk) publish(1, k + chunk, max(0, min(chunk, have_t - k - chunk)));
        }
        const int tile_base = in_lds ? t * item_stride : item_off[t];
        const int tx0 = (t % tiles_x) * kTile, ty0 = (t / tiles_x) * kTile;
        const int tx1 = min(tx0 + kTile, g.sx) - 1, ty1 = min(ty0 + kTile, g.sy) - 1;

        int  fetch_c = -1, fetch_k = 0; // this wavefront owes the publication of chunk fetch_c; lane 0's atomic returned fetch_k
        auto settle = [&]() {
            if (fetch_c >= 0 && lane == 0) publish(fetch_c, fetch_k, max(0, min(chunk, cnt[t] - fetch_k)));
            fetch_c = -1;
        };
        // descriptor of chunk c, or 0 when the list ends before it
        auto wait_desc = [&](int c) -> unsigned long long {
            for (;;) {
                if (c >= __hip_atomic_load(&s_end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return 0ull;
                const unsigned long long d = __hip_atomic_load(&s_desc[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (d != 0ull) return d;
                if (fetch_c >= 0)
                    settle(); // never wait while owing: the chunk awaited may hang on this very publication
                else
                    __builtin_amdgcn_s_sleep(1);
            }
        };
        // a wavefront's next block: its beams are requested before the current block is walked (hides the two dependent
        // loads).  Returns false at the end of the visit.
        auto grab = [&](int2 *raw) -> bool {
            int T = 0;
            if (lane == 0) T = atomicAdd(&s_ticket, 1);
            T = __builtin_amdgcn_readfirstlane(T);
            const int c = T / chunk, idx = T - c * chunk;
            if (c >= kMaxChunks) return false; // the visit's log is full: write back, then the same tile again
            const unsigned long long d = wait_desc(c);
            const int np = (int)((d >> 32) & 0xffffu), base = (int)(unsigned)(d & 0xffffffffull);
            if (idx >= np) return false; // the tile's list ends before this block
            if (idx == 0 && np == chunk && c + 2 < kMaxChunks) {
                // first ticket of a full chunk: if the next chunk is full too, fetch the one after it
                const unsigned long long d1 = wait_desc(c + 1);
                if ((int)((d1 >> 32) & 0xffffu) == chunk) {
                    settle(); // (one fetch at a time per wavefront)
                    if (lane == 0) fetch_k = atomicAdd(&cursor[t], chunk);
                    fetch_c = c + 2;
                }
            }
            const int bi = items[tile_base + base + idx] * kBlock + lane;
            *raw = bi < n ? *reinterpret_cast<const int2 *>(&beams[bi]) : make_int2(-1, 0);
            return true;
        };
        int2 raw = make_int2(-1, 0), raw_next = make_int2(-1, 0);
        bool have = grab(&raw);
        while (have) {
            const bool have_next = grab(&raw_next);

            // ---- clip this lane's beam to the tile
            int rem = 0, a = 0, e = 0, dv2 = 0, den = 1, step_u = 0, step_v = 0, to_end = 0;
            const int x0 = (short)(raw.x & 0xffff), y0 = raw.x >> 16;
            const int x1 = (short)(raw.y & 0xffff), y1 = raw.y >> 16;
            const bool maybe = x0 >= 0 && max(x0, x1) >= tx0 && min(x0, x1) <= tx1 && max(y0, y1) >= ty0 &&
                               min(y0, y1) <= ty1;
            if (__any(maybe) && !(ablate & 16)) {
                const int dx = abs(x1 - x0), dy = abs(y1 - y0);
                // u = major axis, v = minor axis; step i in [0, du]; i == du is the end cell (hit)
                const bool xm = dx >= dy;
                const int  u0 = xm ? x0 : y0, v0 = xm ? y0 : x0, u1 = xm ? x1 : y1, v1 = xm ? y1 : x1;
                const int  du = xm ? dx : dy, dv = xm ? dy : dx;
                const int  tu0 = xm ? tx0 : ty0, tu1 = xm ? tx1 : ty1;
                const int  tv0 = xm ? ty0 : tx0, tv1 = xm ? ty1 : tx1;
                const bool up = u1 > u0, vp = v1 > v0;
                int        i_lo = max(0, up ? tu0 - u0 : u0 - tu1);
                int        i_hi = min(du, up ? tu1 - u0 : u0 - tu0);
                den = max(2 * du, 1);
                dv2 = 2 * dv;
                const float rden = __builtin_amdgcn_rcpf((float)den); // 1 ulp is ample: see floor_div_small
                // minor axis: v0 + sv*k in [tv0, tv1]  <=>  k in [k_lo, k_hi]
                const int k_lo = vp ? tv0 - v0 : v0 - tv1;
                const int k_hi = min(vp ? tv1 - v0 : v0 - tv0, dv);
                bool      enters = maybe && k_hi >= 0 && k_lo <= dv;
                if (enters && dv > 0 && !(ablate & 32)) {
                    const float rdv2 = __builtin_amdgcn_rcpf((float)dv2);
                    // k_i >= k_lo  <=>  i >= ceil((2*du*k_lo - du) / (2*dv))
                    if (k_lo > 0) i_lo = max(i_lo, floor_div_small(__mul24(den, k_lo) - du + dv2 - 1, dv2, rdv2));
                    // k_i <= k_hi  <=>  i <= floor((2*du*(k_hi+1) - du - 1) / (2*dv))
                    i_hi = min(i_hi, floor_div_small(__mul24(den, k_hi + 1) - du - 1, dv2, rdv2));
                }
                enters = enters && i_lo <= i_hi;
                if (enters && !(ablate & 64)) {
                    const int num = __mul24(i_lo, dv2) + du; // < 2^31
                    const int k = du ? floor_div_small(num, den, rden) : 0;
                    e = num - __mul24(k, den); // running remainder in [0, den)
                    const int ul = (up ? u0 + i_lo : u0 - i_lo) - tu0, vl = (vp ? v0 + k : v0 - k) - tv0;
                    a = xm ? vl * kTileStride + ul : ul * kTileStride + vl;
                    step_u = (up ? 1 : -1) * (xm ? 1 : kTileStride);
                    step_v = (vp ? 1 : -1) * (xm ? kTileStride : 1);
                    rem = i_hi - i_lo + 1;
                    to_end = du - i_lo; // steps until the end cell
                    did += (unsigned)rem;
                }
            }
            if (ablate & 1) rem = 0;
            // ---- the end cell's hit if the beam ends in this tile (its last step here is then the end cell),
            // then all lanes step their beams' misses together.  Per step: one ds_add and five integer
            // instructions -- the cell is kept as a byte offset, the error term biased by 2^32 - den so that the
            // carry of `+= dv2` is the test `e + dv2 >= den` -- in trips of kWalkUnroll steps for the lanes that
            // have that many left (no per-step predicate), then the remainders.
            const bool hit_here = rem > 0 && to_end < rem;
            if (hit_here) atomicAdd(&tile[(y1 - ty0) * kTileStride + (x1 - tx0)], 0x10000u);
            int            miss = rem - (hit_here ? 1 : 0);
            unsigned char *tb = reinterpret_cast<unsigned char *>(tile);
            int            a4 = a * 4;
            const int      su4 = step_u * 4, sv4 = step_v * 4;
            const unsigned bias = 0u - (unsigned)den, udv2 = (unsigned)dv2;
            unsigned       eu = (unsigned)e + bias;
            const auto     advance = [&]() {
                const unsigned e2 = eu + udv2;
                const bool     carry = e2 < eu; // e + dv2 >= den
                a4 += su4 + (carry ? sv4 : 0);
                eu = e2 + (carry ? bias : 0u);
            };
            // MERGE: 64 consecutive beams of a scan leave the sensor through the same cells for their first ~230
            // steps (adjacent beams are 0.25 degrees apart), and LDS atomics of one instruction to one address are
            // served one after the other.  Lanes are ordered by angle, so lanes on the same cell are neighbours: the
            // first lane of a run adds the run's length, the others add nothing.  A step on which every lane has a
            // cell of its own (the far field) costs the compare only.
            const auto add_miss = [&]() {
                if (!MERGE) {
                    atomicAdd(reinterpret_cast<unsigned *>(tb + a4), 1u);
                    return;
                }
                const int  prev = __builtin_amdgcn_update_dpp(-1, a4, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                const bool leader = a4 != prev; // also after a lane that is not stepping (its value does not arrive)
                const unsigned long long on = __ballot(true), lead = __ballot(leader);
                if (lead == on) {
                    atomicAdd(reinterpret_cast<unsigned *>(tb + a4), 1u);
                } else {
                    const unsigned long long ends = lead | ~on; // a run ends before the next leader or idle lane
                    const unsigned long long rest = lane == 63 ? 0ull : ends >> (lane + 1);
                    const unsigned run = rest ? (unsigned)__builtin_ctzll(rest) + 1u : (unsigned)(64 - lane);
                    if (leader) atomicAdd(reinterpret_cast<unsigned *>(tb + a4), run);
                }
            };
#pragma unroll
            for (int k = 0; k < kWalkStagger - 1; ++k) { // the stagger: lane j is j % kWalkStagger cells ahead when the trips begin
                if (lane % kWalkStagger > k && miss > 0) {
                    add_miss();
                    advance();
                    --miss;
                }
            }
            while (__any(miss >= kWalkUnroll)) {
                if (miss >= kWalkUnroll) {
#pragma unroll
                    for (int k = 0; k < kWalkUnroll; ++k) {
                        add_miss();
                        advance();
                    }
                    miss -= kWalkUnroll;
                }
            }
#pragma unroll
            for (int k = 0; k < kWalkUnroll - 1; ++k) {
                if (miss > k) add_miss();
                advance(); // a finished lane's cell is not used again
            }
            settle(); // (after the walk: the fetch's round trip is behind it by now)
            have = have_next;
            raw = raw_next;
        }
        settle(); // nothing fetched may stay unpublished: others wait for it
        __syncthreads(); // every wavefront has found the end of the visit
        flush(t);
        if (s_done == 0) continue; // the visit's log was full and the tile has blocks left: the same tile again, fresh counters
        // this tile has nothing left to hand out: on to the tile with the most blocks left, if any
        unsigned long long best = 0ull;
        for (int u = tid; u < n_tiles; u += kTileThreads) {
            const int have_u = cnt[u];
            const int taken = __hip_atomic_load(&cursor[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int left = have_u - min(have_u, taken);
            const unsigned long long key = ((unsigned long long)(unsigned)left << 32) | (unsigned)u;
            best = left > 0 && key > best ? key : best;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off);
            best = o > best ? o : best;
        }
        if (lane == 0) s_best[tid >> 6] = best;
        __syncthreads();
        best = 0ull;
        for (int w = 0; w < kTileThreads / 64; ++w) best = s_best[w] > best ? s_best[w] : best;
        if (best == 0ull) break; // every list is handed out (what is still being walked belongs to others)
        cur = (int)(unsigned)(best & 0xffffffffull);
    }
    mark_dirty_rows(g.dirty, d_lo, d_hi);
    block_add_updates(g.updates, did);
}

// --------------------------------------------------------------- row ranges
// A range {lo, -hi} of storage rows, empty while lo > hi (cleared to 0x7f7f7f7f both).  Counts are nonzero only in rows of
// the touched range (every update marks its rows; a merge over the GPUs marks the rows it summed: slam_grid_mark_rows), so
// a reset zeroes those rows only -- and remembers them as changed, because their evidence and occupancy are stale until
// the next finalize, which covers the touched and the changed rows and nothing else.  At 2000 x 2000 with a 40 x 30 m
// room the ranges are 30 % of the rows: reset 7 -> 2 us, finalize 12 -> 4 us per step.
__device__ inline bool row_in(const int *r, int row) { return row >= r[0] && row <= -r[1]; }

// (a fixed launch whose workgroups stride over the rows of the range: sizing the launch for the whole grid and leaving
// the rows outside the range to empty workgroups cost as much as the memset it replaces -- 16 000 empty workgroups, 9 us)
constexpr int kRangeBlocks = 1024;

__global__ __launch_bounds__(256) void reset_rows_kernel(int32_t *planes, size_t cells, int sx, int sy, const int *ranges)
{
    const int lo = max(ranges[0], 0), hi = min(-ranges[1], sy - 1);
    if (hi < lo) return;
    const int  per_row = (sx + 1023) / 1024; // 256 threads x 4 ints
    const long items = (long)(hi - lo + 1) * per_row;
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        const int row = lo + (int)(it / per_row), x = ((int)(it % per_row) * 256 + threadIdx.x) * 4;
        for (int k = 0; k < 4; ++k)
            if (x + k < sx) {
                planes[(size_t)row * sx + x + k] = 0;
                planes[cells + (size_t)row * sx + x + k] = 0;
            }
    }
}

// the touched rows become changed rows (hull), the touched range starts again.  (A launch of its own, like
// ranges_set_kernel: letting the range kernels' last workgroup do it -- a completion counter, one atomic and a fence per
// workgroup -- made each of them 12 us slower; 1024 same-address atomics again.)
__global__ void ranges_retire_kernel(int *ranges)
{
    if (threadIdx.x || blockIdx.x) return;
    ranges[2] = min(ranges[2], ranges[0]);
    ranges[3] = min(ranges[3], ranges[1]);
    ranges[0] = ranges[1] = 0x7f7f7f7f;
}

// slam_grid_fold of rows lo..hi: they and the touched rows are due for the next finalize; the touched range starts again
// if the fold covered it (rows it did not cover still hold counts: they stay touched)
__global__ void ranges_fold_kernel(int *ranges, int lo, int hi)
{
    if (threadIdx.x || blockIdx.x) return;
    ranges[2] = min(min(ranges[2], ranges[0]), lo);
    ranges[3] = min(min(ranges[3], ranges[1]), -hi);
    if (ranges[0] >= lo && -ranges[1] <= hi) ranges[0] = ranges[1] = 0x7f7f7f7f;
}

// mode 0: nothing changed any more (after a finalize); 1: everything did (a roll, a merge of whole planes); 2: rows lo..hi
// were written from outside (a merge of those rows): they count as touched
__global__ void ranges_set_kernel(int *ranges, int mode, int lo, int hi, int sy)
{
    if (threadIdx.x || blockIdx.x) return;
    if (mode == 0) {
        ranges[2] = ranges[3] = 0x7f7f7f7f;
    } else if (mode == 1) {
        ranges[2] = 0;
        ranges[3] = -(sy - 1);
    } else {
        ranges[0] = min(ranges[0], lo);
        ranges[1] = min(ranges[1], -hi);
    }
}

// --------------------------------------------------------------- finalize
// SURVEY 8(a) G3 on the summed counts, window order out (mls.h:167-175 data[x + size_x*y]).
__device__ inline void cell_value(int h, int m, double inc, double dec, double minp, double &v, int8_t &o)
{
    v = inc * (double)h;
    o = -1;
    if (h > 0 && v > minp) o = 100; // mls.cpp:101-105
    v = v - dec * (double)m;
    if (m > 0 && v < minp) o = 0; // mls.cpp:137-141
}

__device__ inline void finalize_cell(const GridView &g, int x, int y, double inc, double dec, double minp, double *num_pts, int8_t *occ)
{
    const int s = storage_index(g, x, y);
    const int h = g.hits[s] + (g.acc_hits ? g.acc_hits[s] : 0), m = g.misses[s] + (g.acc_misses ? g.acc_misses[s] : 0);
    double    v;
    int8_t    o;
    cell_value(h, m, inc, dec, minp, v, o);
    num_pts[x + (size_t)g.sx * y] = v;
    occ[x + (size_t)g.sx * y] = o;
}

// Four consecutive cells of one storage row per thread, where the row length and the toroidal x origin are multiples of
// four (any non-rolling grid): 16-byte loads of both count planes, 2 x 16-byte stores of the evidence, one 4-byte store of
// the occupancy -- the rows kernels stream 17 B per cell and were bound by the requests in flight per thread, not by HBM
// (one cell per thread: 8.0 us for config 2's 602 rows; four: see DESIGN.md 4.3).  RESET: the counts just folded go back to
// zero (slam_grid_finalize_reset).
template <bool RESET>
__device__ inline void finalize_quad(const GridView &g, int srow, int ix0, int y, double inc, double dec, double minp, double *num_pts,
                                     int8_t *occ)
{
    const size_t s0 = (size_t)srow * g.sx + ix0;
    int4         h = *reinterpret_cast<const int4 *>(g.hits + s0), m = *reinterpret_cast<const int4 *>(g.misses + s0);
    const bool   any = (h.x | h.y | h.z | h.w | m.x | m.y | m.z | m.w) != 0;
    if (g.acc_hits) {
        const int4 ah = *reinterpret_cast<const int4 *>(g.acc_hits + s0), am = *reinterpret_cast<const int4 *>(g.acc_misses + s0);
        h.x += ah.x, h.y += ah.y, h.z += ah.z, h.w += ah.w;
        m.x += am.x, m.y += am.y, m.z += am.z, m.w += am.w;
    }
    double v[4];
    int8_t o[4];
    cell_value(h.x, m.x, inc, dec, minp, v[0], o[0]);
    cell_value(h.y, m.y, inc, dec, minp, v[1], o[1]);
    cell_value(h.z, m.z, inc, dec, minp, v[2], o[2]);
    cell_value(h.w, m.w, inc, dec, minp, v[3], o[3]);
    int x0 = ix0 - g.ox; // the window cells stored there (storage_index's inverse): the four do not straddle the seam
    x0 += x0 < 0 ? g.sx : 0;
    const size_t w0 = (size_t)x0 + (size_t)g.sx * y;
    *reinterpret_cast<double2 *>(num_pts + w0) = make_double2(v[0], v[1]);
    *reinterpret_cast<double2 *>(num_pts + w0 + 2) = make_double2(v[2], v[3]);
    *reinterpret_cast<unsigned *>(occ + w0) = (unsigned)(unsigned char)o[0] | ((unsigned)(unsigned char)o[1] << 8) |
                                               ((unsigned)(unsigned char)o[2] << 16) | ((unsigned)(unsigned char)o[3] << 24);
    if (RESET && any) {
        *reinterpret_cast<int4 *>(g.hits + s0) = make_int4(0, 0, 0, 0);
        *reinterpret_cast<int4 *>(g.misses + s0) = make_int4(0, 0, 0, 0);
    }
}

// the rows lo..hi of the planes: one cell per thread, or (quads) four
template <bool RESET>
__device__ inline void finalize_rows(const GridView &g, int lo, int hi, double inc, double dec, double minp, double *num_pts, int8_t *occ)
{
    const bool quads = (g.sx & 3) == 0 && (g.ox & 3) == 0;
    const int  per_thread = quads ? 4 : 1;
    const int  per_row = (g.sx + 256 * per_thread - 1) / (256 * per_thread);
    const long items = (long)(hi - lo + 1) * per_row;
    for (long it = blockIdx.x; it < items; it += gridDim.x) {
        const int srow = lo + (int)(it / per_row), ix = ((int)(it % per_row) * 256 + (int)threadIdx.x) * per_thread;
        if (ix >= g.sx) continue;
        int y = srow - g.oy; // the window row stored there (storage_index's inverse)
        y += y < 0 ? g.sy : 0;
        if (quads) {
            finalize_quad<RESET>(g, srow, ix, y, inc, dec, minp, num_pts, occ);
        } else { // here ix counts WINDOW cells of the row (any order covers the row)
            finalize_cell(g, ix, y, inc, dec, minp, num_pts, occ);
            if (RESET) {
                const int s = storage_index(g, ix, y);
                if (g.hits[s]) g.hits[s] = 0;
                if (g.misses[s]) g.misses[s] = 0;
            }
        }
    }
}

// every cell (after the in-order mode, or where the caller wants it)
__global__ __launch_bounds__(256) void finalize_kernel(GridView g, double inc, double dec, double minp,
                                                       double *num_pts, int8_t *occ)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.sx) return;
    finalize_cell(g, x, y, inc, dec, minp, num_pts, occ);
}

// only the storage rows whose counts were touched or changed since the last finalize (the hull of the two ranges): a fixed
// launch striding over them
__global__ __launch_bounds__(256) void finalize_rows_kernel(GridView g, double inc, double dec, double minp, double *num_pts,
                                                            int8_t *occ, const int *ranges)
{
    const int lo = max(min(ranges[0], ranges[2]), 0), hi = min(max(-ranges[1], -ranges[3]), g.sy - 1);
    if (hi < lo) return;
    finalize_rows<false>(g, lo, hi, inc, dec, minp, num_pts, occ);
}

// slam_grid_finalize_reset: finalize_rows_kernel that also zeroes the counts it has just folded (what slam_grid_reset_counts
// would do next) and retires the ranges -- one launch where the batch step had four (finalize_rows, ranges_set, reset_rows,
// ranges_retire).  The rows it covers are the hull of the touched and the changed rows, as in finalize_rows_kernel.  Afterwards
// the touched rows are "changed" (their counts are zero again while their evidence still shows this batch: the next finalize
// must visit them) and nothing is touched: that state goes into the OTHER range buffer (`next`; every workgroup reads `ranges`,
// nobody reads `next` during this launch), which the host makes the grid's current one for everything enqueued behind.
__global__ __launch_bounds__(256) void finalize_reset_rows_kernel(GridView g, double inc, double dec, double minp, double *num_pts,
                                                                  int8_t *occ, const int *ranges, int *next)
{
    const int t_lo = ranges[0], t_nhi = ranges[1];
    const int lo = max(min(t_lo, ranges[2]), 0), hi = min(max(-t_nhi, -ranges[3]), g.sy - 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        next[0] = next[1] = 0x7f7f7f7f;
        next[2] = t_lo;
        next[3] = t_nhi;
    }
    if (hi < lo) return;
    finalize_rows<true>(g, lo, hi, inc, dec, minp, num_pts, occ);
}

__global__ __launch_bounds__(256) void gather_counts_kernel(GridView g, int32_t *hits_w, int32_t *misses_w)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.sx) return;
    const int s = storage_index(g, x, y);
    hits_w[x + (size_t)g.sx * y] = g.hits[s] + (g.acc_hits ? g.acc_hits[s] : 0);
    misses_w[x + (size_t)g.sx * y] = g.misses[s] + (g.acc_misses ? g.acc_misses[s] : 0);
}

// slam_grid_fold: rows [row_lo, row_hi] of the count planes are added to the accumulator planes and zeroed
__global__ __launch_bounds__(256) void fold_rows_kernel(int32_t *planes, int32_t *acc, size_t cells, int sx, int row_lo, int n_rows)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)n_rows * sx;
    if (i >= n) return;
    const size_t k = (size_t)row_lo * sx + i;
    for (int p = 0; p < 2; ++p) {
        const size_t j = p * cells + k;
        const int    v = planes[j];
        if (v) {
            acc[j] += v;
            planes[j] = 0;
        }
    }
}

// MLS::setPose roll, mls.cpp:461-468: cells that rolled into the window are cleared
__global__ __launch_bounds__(256) void roll_clear_kernel(GridView g, int dx, int dy, double *num_pts,
                                                         int8_t *occ_state)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.sx) return;
    if (x < -dx || x >= g.sx - dx || y < -dy || y >= g.sy - dy) {
        const int s = storage_index(g, x, y);
        g.hits[s] = 0;
        g.misses[s] = 0;
        if (g.acc_hits) {
            const_cast<int32_t *>(g.acc_hits)[s] = 0;
            const_cast<int32_t *>(g.acc_misses)[s] = 0;
        }
        num_pts[s] = 0.0;
        occ_state[s] = -1;
    }
}

// in-order single scan: per touched cell replay the reference's += / -= sequence
// per-scan delta of a cell: hits in the high, misses in the low 32 bits of one 64-bit word,
// so exactly one point sees the cell untouched and lists it
__global__ __launch_bounds__(256) void inorder_count_kernel(GridView g, const float *obs, int n_obs,
                                                            const float *gnd, int n_gnd, int stride,
                                                            unsigned long long *delta, int *touched,
                                                            int *n_touched)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_obs + n_gnd) return;
    const bool   is_obs = i < n_obs;
    const float *p = is_obs ? obs + (size_t)i * stride : gnd + (size_t)(i - n_obs) * stride;
    int cx, cy;
    if (!point_cell(g, p[0], p[1], &cx, &cy)) return;
    const int s = storage_index(g, cx, cy);
    const unsigned long long old = atomicAdd(&delta[s], is_obs ? (1ull << 32) : 1ull);
    atomicAdd(is_obs ? &g.hits[s] : &g.misses[s], 1);
    if (old == 0ull) { // the first point of the scan in this cell
        const int row = s / g.sx;
        if (row < __hip_atomic_load(&g.dirty[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&g.dirty[0], row);
        if (-row < __hip_atomic_load(&g.dirty[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&g.dirty[1], -row);
    }
    if (old == 0ull) {
        const int k = atomicAdd(n_touched, 1);
        touched[k] = s;
    }
}

__global__ __launch_bounds__(256) void inorder_apply_kernel(GridView g, double inc, double dec, double minp,
                                                            unsigned long long *delta, const int *touched,
                                                            const int *n_touched, double *num_pts,
                                                            int8_t *occ_state, unsigned long long *updates)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned  did = 0;
    if (i < *n_touched) {
        const int s = touched[i];
        const unsigned long long d = delta[s];
        delta[s] = 0ull; // each touched cell is listed once: no other thread reads it
        const int h = (int)(d >> 32), m = (int)(d & 0xffffffffull);
        {
            double v = num_pts[s];
            int8_t o = occ_state[s];
            for (int k = 0; k < h; ++k) v = __dadd_rn(v, inc); // mls.cpp:99, one += per point
            if (h > 0 && v > minp) o = 100;                    // monotone: last test decides
            for (int k = 0; k < m; ++k) v = __dsub_rn(v, dec); // mls.cpp:135
            if (m > 0 && v < minp) o = 0;
            num_pts[s] = v;
            occ_state[s] = o;
            did = (unsigned)(h + m);
        }
    }
    block_add_updates(updates, did);
}

__global__ __launch_bounds__(256) void window_from_storage_kernel(GridView g, const double *num_s,
                                                                  const int8_t *occ_s, double *num_w,
                                                                  int8_t *occ_w)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.sx) return;
    const int s = storage_index(g, x, y);
    num_w[x + (size_t)g.sx * y] = num_s[s];
    occ_w[x + (size_t)g.sx * y] = occ_s[s];
}

__global__ void fill_i8_kernel(int8_t *p, size_t n, int8_t v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

} // namespace

struct slam_grid {
    slam_grid_params prm;
    GridView         gv;
    size_t           cells = 0;
    int32_t         *d_planes = nullptr;  // [hits | misses]
    double          *d_num_w = nullptr;   // window order, written by finalize
    int8_t          *d_occ_w = nullptr;   // window order, written by finalize
    double          *d_num_s = nullptr;   // storage order, in-order mode state
    int8_t          *d_occ_s = nullptr;   // storage order, in-order mode state
    unsigned long long *d_delta = nullptr; // [cells] per-scan deltas of the in-order mode
    int             *d_touched = nullptr; // [2*cap_points] + counter
    size_t           cap_touched = 0;
    unsigned long long *d_updates = nullptr;
    int             *d_dirty = nullptr;    // [2][4] see GridView::dirty; two buffers: slam_grid_finalize_reset reads one and starts the other
    int32_t         *d_acc = nullptr;      // [hits | misses] accumulator planes (slam_grid_enable_accumulator)
    Beam            *d_beams = nullptr;
    size_t           cap_beams = 0;
    int4            *d_chunk_box = nullptr;
    size_t           cap_chunks = 0;
    int             *d_tile_cnt = nullptr; // [n_tiles] overlapping chunks per tile
    int             *d_tile_fill = nullptr; // [n_tiles+1] item_off: where each tile's list starts (grids beyond kMaxLdsTiles tiles)
    int             *d_cursor = nullptr;   // [n_tiles+1] blocks of each tile's list handed out so far; behind them, the launch's tile write-backs
    int             *d_items = nullptr;    // chunk ids bucketed by tile
    size_t           cap_items = 0;
    int              n_cu = 256;
    int              seg_items = 0;  // 64-beam blocks a workgroup takes from a tile's list at a time; 0 = kChunkDefault
    int              last_chunks = 0;
    int              ablate = 0;     // debug: SLAM_RAYCAST_ABLATE bit mask (timing experiments only)
    int              wg_per_cu = 0;  // persistent raycast workgroups per CU; 0 = by the number of tiles (raycast_wg_per_cu)
    bool             merge = false;  // tiled raycast: lanes on one cell add once (SLAM_RAYCAST_TILED_MERGE)
    void            *d_stage = nullptr;   // host-API staging
    size_t           cap_stage = 0;
    bool             state_from_inorder = false;
    long             shift_x = 0, shift_y = 0; // cells the rolling window has moved since creation (sum of setPose's dx, dy)
};

namespace {

int reserve(void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return SLAM_OK;
    if (*p) (void)hipFree(*p); // (waits for the device: callers with varying sizes reserve up front, slam_grid_reserve)
    const size_t grown = *cap + *cap / 4;
    *p = nullptr;
    *cap = 0;
    size_t want = std::max(std::max(bytes, grown), (size_t)4096);
    SLAM_HIP(hipMalloc(p, want));
    *cap = want;
    return SLAM_OK;
}

int reserve_beams(slam_grid *g, size_t n)
{
    size_t cb = g->cap_beams * sizeof(Beam);
    void  *p = g->d_beams;
    SLAM_TRY(reserve(&p, &cb, n * sizeof(Beam)));
    g->d_beams = static_cast<Beam *>(p);
    g->cap_beams = cb / sizeof(Beam);
    const size_t chunks = (n + kBlock - 1) / kBlock + kChunk / kBlock;
    size_t       cc = g->cap_chunks * sizeof(int4);
    p = g->d_chunk_box;
    SLAM_TRY(reserve(&p, &cc, chunks * sizeof(int4)));
    g->d_chunk_box = static_cast<int4 *>(p);
    g->cap_chunks = cc / sizeof(int4);
    if (g->prm.raycast_impl != SLAM_RAYCAST_GLOBAL) {
        const size_t n_tiles = (size_t)((g->gv.sx + kTile - 1) / kTile) * ((g->gv.sy + kTile - 1) / kTile);
        if (!g->d_tile_cnt) {
            SLAM_HIP(hipMalloc((void **)&g->d_tile_cnt, n_tiles * sizeof(int)));
            SLAM_HIP(hipMalloc((void **)&g->d_tile_fill, (n_tiles + 1) * sizeof(int)));
            SLAM_HIP(hipMalloc((void **)&g->d_cursor, (n_tiles + 1) * sizeof(int)));
        }
        size_t ci = g->cap_items * sizeof(int);
        p = g->d_items;
        SLAM_TRY(reserve(&p, &ci, (chunks * n_tiles) * sizeof(int)));
        g->d_items = static_cast<int *>(p);
        g->cap_items = ci / sizeof(int);
    }
    return SLAM_OK;
}

// Persistent raycast workgroups per CU where the caller did not say: two (thirty-two wavefronts, 2 x 75 KB of LDS).  A
// workgroup's walk is a chain of dependent integer steps and LDS adds: the second workgroup's wavefronts issue in its gaps
// (tools/raycast_time.py, chunks of 16 blocks: config 2 0.110 ms per call with two against 0.131 with one; config 4's share
// 0.274 against 0.322).
int raycast_wg_per_cu(const slam_grid *g, int n_tiles) { return g->wg_per_cu > 0 ? g->wg_per_cu : 2; }
// ... and how many in all: the whole chip, unless the caller caps it (slam_grid_params::raycast_max_workgroups: a raycast that
// runs beside registrations holds every CU it has a workgroup on for as long as the launch lasts)
int raycast_workgroups(const slam_grid *g, int n_tiles)
{
    const int all = raycast_wg_per_cu(g, n_tiles) * g->n_cu;
    return g->prm.raycast_max_workgroups > 0 ? std::min(all, g->prm.raycast_max_workgroups) : all;
}

int walk_beams(slam_grid *g, int n, hipStream_t st)
{
    const int n_chunks = (n + kBlock - 1) / kBlock; // culling blocks
    if (g->prm.raycast_impl != SLAM_RAYCAST_GLOBAL) {
        const int tiles_x = (g->gv.sx + kTile - 1) / kTile, tiles_y = (g->gv.sy + kTile - 1) / kTile;
        const int n_tiles = tiles_x * tiles_y;
        int       *item_off = g->d_tile_fill;
        const dim3 tgrid((n_tiles * 64 + 255) / 256);
        const int  chunk = g->seg_items > 0 ? g->seg_items : kChunkDefault;
        const dim3 rgrid(raycast_workgroups(g, n_tiles)); // persistent workgroups (75 KB of LDS each)
        g->last_chunks = n_chunks;
        if (n_tiles <= kMaxLdsTiles) {
            // one pass over the block boxes; the raycast workgroups work the prefix of the tile counts out themselves
            hipLaunchKernelGGL(tile_items_wg_kernel, dim3(n_tiles), dim3(1024), 0, st, g->d_chunk_box, n_chunks, tiles_x,
                               g->gv.sx, g->gv.sy, g->d_tile_cnt, g->d_items, g->d_cursor);
            hipLaunchKernelGGL(g->merge ? raycast_tiled_kernel<true> : raycast_tiled_kernel<false>, rgrid, dim3(kTileThreads), 0, st,
                               g->gv, g->d_beams, n, g->d_items, (const int *)nullptr, n_tiles, g->d_cursor, tiles_x, chunk, g->ablate,
                               g->d_tile_cnt, n_chunks);
        } else {
            hipLaunchKernelGGL((tile_items_kernel<0>), tgrid, dim3(256), 0, st, g->d_chunk_box, n_chunks, tiles_x,
                               n_tiles, g->gv.sx, g->gv.sy, g->d_tile_cnt, item_off, g->d_items);
            hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, g->d_tile_cnt, n_tiles, item_off, g->d_cursor);
            hipLaunchKernelGGL((tile_items_kernel<1>), tgrid, dim3(256), 0, st, g->d_chunk_box, n_chunks, tiles_x,
                               n_tiles, g->gv.sx, g->gv.sy, g->d_tile_cnt, item_off, g->d_items);
            hipLaunchKernelGGL(g->merge ? raycast_tiled_kernel<true> : raycast_tiled_kernel<false>, rgrid, dim3(kTileThreads), 0, st,
                               g->gv, g->d_beams, n, g->d_items, item_off, n_tiles, g->d_cursor, tiles_x, chunk, g->ablate,
                               g->d_tile_cnt, 0);
        }
    } else {
        hipLaunchKernelGGL(raycast_global_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g->gv, g->d_beams, n);
    }
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

dim3 grid2d(const slam_grid *g) { return dim3((g->gv.sx + 255) / 256, g->gv.sy); }

} // namespace

extern "C" {

void slam_grid_default_params(slam_grid_params *p)
{
    if (!p) return;
    p->max_range = 75.0;          // mls.h:161
    p->occupancy_increment = 1.0; // mls.h:188
    p->occupancy_decrement = 0.3; // mls.h:189
    p->min_cluster_points = 10;   // mls.h:165
    p->rolling = 1;               // local_mapper.cpp:29 MLS(200,200,0.2,true)
    p->raycast_impl = SLAM_RAYCAST_TILED;
    p->raycast_seg_items = 0;
    p->raycast_wg_per_cu = 0;
    p->raycast_max_workgroups = 0;
}

int slam_grid_create(int size_x, int size_y, double resolution, const slam_grid_params *params,
                     slam_grid_t **out)
{
    SLAM_REQUIRE(out, SLAM_E_INVALID, "slam_grid_create: null out pointer");
    *out = nullptr;
    SLAM_REQUIRE(size_x > 0 && size_y > 0 && size_x <= 32767 && size_y <= 32767 && resolution > 0,
                 SLAM_E_INVALID, "slam_grid_create: size must be 1..32767 cells and resolution > 0");
    SLAM_TRY(require_device());
    slam_grid *g = new (std::nothrow) slam_grid();
    SLAM_REQUIRE(g, SLAM_E_NOMEM, "slam_grid_create: out of host memory");
    if (params)
        g->prm = *params;
    else
        slam_grid_default_params(&g->prm);
    g->cells = (size_t)size_x * size_y;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g->n_cu = std::max(1, prop.multiProcessorCount);
    }
    // (a visit's log holds at least two chunks and at most kMaxAccBlocks / kChunkMin)
    g->seg_items = g->prm.raycast_seg_items > 0 ? std::min(std::max(g->prm.raycast_seg_items, kChunkMin), kMaxAccBlocks / 2) : 0;
    if (g->prm.raycast_wg_per_cu > 0) g->wg_per_cu = std::min(g->prm.raycast_wg_per_cu, 8);
    g->merge = g->prm.raycast_impl == SLAM_RAYCAST_TILED_MERGE;
#ifdef SLAM_MEASURE
    if (const char *e = getenv("SLAM_RAYCAST_MERGE")) g->merge = atoi(e) != 0;
    if (const char *e = getenv("SLAM_RAYCAST_SEG")) g->seg_items = std::min(std::max(kChunkMin, atoi(e)), kMaxAccBlocks / 2);
    if (const char *e = getenv("SLAM_RAYCAST_ABLATE")) g->ablate = atoi(e);
    if (const char *e = getenv("SLAM_RAYCAST_WGPCU")) g->wg_per_cu = std::max(1, atoi(e));
#endif
    int rc = SLAM_OK;
    auto alloc = [&](void **p, size_t bytes) {
        if (rc == SLAM_OK && hipMalloc(p, bytes) != hipSuccess) {
            set_error("slam_grid_create: hipMalloc of %zu bytes failed", bytes);
            (void)hipGetLastError();
            rc = SLAM_E_NOMEM;
        }
    };
    alloc((void **)&g->d_planes, 2 * g->cells * sizeof(int32_t));
    alloc((void **)&g->d_num_w, g->cells * sizeof(double));
    alloc((void **)&g->d_occ_w, g->cells);
    alloc((void **)&g->d_num_s, g->cells * sizeof(double));
    alloc((void **)&g->d_occ_s, g->cells);
    alloc((void **)&g->d_updates, kUpdateSlots * sizeof(unsigned long long));
    alloc((void **)&g->d_dirty, 8 * sizeof(int));
    if (rc != SLAM_OK) {
        slam_grid_destroy(g);
        return rc;
    }
    GridView &v = g->gv;
    v.sx = size_x;
    v.sy = size_y;
    v.ox = v.oy = 0;
    v.res = resolution;
    v.max_range = g->prm.max_range;
    v.pose_x = v.pose_y = 0.0;
    v.rolling = g->prm.rolling ? 1 : 0;
    v.hits = g->d_planes;
    v.misses = g->d_planes + g->cells; // (endpoints_kernel addresses both planes through `hits` with an int key: misses == hits + cells,
                                       // and 2 * cells fits an int -- slam_grid_create refuses grids of 2^30 cells or more)
    v.updates = g->d_updates;
    v.dirty = g->d_dirty;
    v.acc_hits = v.acc_misses = nullptr;
    rc = slam_grid_clear(g, nullptr);
    if (rc == SLAM_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = SLAM_E_HIP;
    if (rc != SLAM_OK) {
        slam_grid_destroy(g);
        return rc;
    }
    *out = g;
    return SLAM_OK;
}

void slam_grid_destroy(slam_grid_t *g)
{
    if (!g) return;
    void *ptrs[] = {g->d_planes, g->d_num_w, g->d_occ_w, g->d_num_s,     g->d_occ_s, g->d_delta,
                    g->d_touched, g->d_updates, g->d_beams, g->d_chunk_box, g->d_stage,
                    g->d_tile_cnt, g->d_tile_fill, g->d_cursor, g->d_items, g->d_dirty, g->d_acc};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete g;
}

int slam_grid_clear(slam_grid_t *g, slam_stream_t stream)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    hipStream_t st = as_stream(stream);
    SLAM_HIP(hipMemsetAsync(g->d_planes, 0, 2 * g->cells * sizeof(int32_t), st));
    SLAM_HIP(hipMemsetAsync(g->d_num_w, 0, g->cells * sizeof(double), st));
    SLAM_HIP(hipMemsetAsync(g->d_num_s, 0, g->cells * sizeof(double), st));
    SLAM_HIP(hipMemsetAsync(g->d_occ_w, 0xff, g->cells, st)); // -1 = unknown (mls.cpp:26)
    SLAM_HIP(hipMemsetAsync(g->d_occ_s, 0xff, g->cells, st));
    SLAM_HIP(hipMemsetAsync(g->d_updates, 0, kUpdateSlots * sizeof(unsigned long long), st));
    SLAM_HIP(hipMemsetAsync(g->d_dirty, 0x7f, 8 * sizeof(int), st)); // {lowest, -highest} = "no row", twice (both buffers)
    if (g->d_acc) SLAM_HIP(hipMemsetAsync(g->d_acc, 0, 2 * g->cells * sizeof(int32_t), st));
    g->state_from_inorder = false;
    return SLAM_OK;
}

int slam_grid_reset_counts(slam_grid_t *g, slam_stream_t stream)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(reset_rows_kernel, dim3(kRangeBlocks), dim3(256), 0, st, g->d_planes, g->cells, g->gv.sx, g->gv.sy, g->gv.dirty);
    hipLaunchKernelGGL(ranges_retire_kernel, dim3(1), dim3(64), 0, st, g->gv.dirty);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int slam_grid_enable_accumulator(slam_grid_t *g)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    if (g->d_acc) return SLAM_OK;
    SLAM_HIP(hipMalloc((void **)&g->d_acc, 2 * g->cells * sizeof(int32_t)));
    SLAM_HIP(hipMemset(g->d_acc, 0, 2 * g->cells * sizeof(int32_t)));
    g->gv.acc_hits = g->d_acc;
    g->gv.acc_misses = g->d_acc + g->cells;
    return SLAM_OK;
}

int slam_grid_fold(slam_grid_t *g, int row_lo, int row_hi, slam_stream_t stream)
{
    SLAM_REQUIRE(g && g->d_acc, SLAM_E_INVALID, "slam_grid_fold: needs slam_grid_enable_accumulator");
    hipStream_t st = as_stream(stream);
    if (row_hi >= row_lo) {
        SLAM_REQUIRE(row_lo >= 0 && row_hi < g->gv.sy, SLAM_E_INVALID, "slam_grid_fold: rows %d..%d outside the grid", row_lo, row_hi);
        const size_t n = (size_t)(row_hi - row_lo + 1) * g->gv.sx;
        hipLaunchKernelGGL(fold_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g->d_planes, g->d_acc, g->cells,
                           g->gv.sx, row_lo, row_hi - row_lo + 1);
        SLAM_HIP(hipGetLastError());
    }
    // the folded rows have changed since the last finalize (by updates, by a merge): they and whatever else was touched
    // stay due for it; the touched range starts again
    if (row_hi >= row_lo)
        hipLaunchKernelGGL(ranges_fold_kernel, dim3(1), dim3(64), 0, st, g->gv.dirty, row_lo, row_hi);
    else // nothing to fold: as before, the touched range starts again (there is nothing in it that holds counts... or the caller said so)
        hipLaunchKernelGGL(ranges_retire_kernel, dim3(1), dim3(64), 0, st, g->gv.dirty);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int slam_grid_mark_rows(slam_grid_t *g, int row_lo, int row_hi, slam_stream_t stream)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    if (row_hi < row_lo) return SLAM_OK;
    SLAM_REQUIRE(row_lo >= 0 && row_hi < g->gv.sy, SLAM_E_INVALID, "slam_grid_mark_rows: rows %d..%d outside the grid", row_lo, row_hi);
    hipLaunchKernelGGL(ranges_set_kernel, dim3(1), dim3(64), 0, as_stream(stream), g->gv.dirty, 2, row_lo, row_hi, g->gv.sy);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int slam_grid_dirty_rows_dev(slam_grid_t *g, int32_t **d_range)
{
    SLAM_REQUIRE(g && d_range, SLAM_E_INVALID, "slam_grid_dirty_rows_dev: bad arguments");
    *d_range = g->gv.dirty; // (the buffer in use as of the calls enqueued so far: slam_grid_finalize_reset alternates between two)
    return SLAM_OK;
}

int slam_grid_dirty_rows(slam_grid_t *g, int *row_lo, int *row_hi)
{
    SLAM_REQUIRE(g && row_lo && row_hi, SLAM_E_INVALID, "slam_grid_dirty_rows: bad arguments");
    int r[2];
    SLAM_HIP(hipMemcpy(r, g->gv.dirty, sizeof r, hipMemcpyDeviceToHost));
    *row_lo = r[0] > g->gv.sy ? 0 : r[0];
    *row_hi = r[0] > g->gv.sy ? -1 : -r[1];
    return SLAM_OK;
}

int slam_grid_set_min_cluster_points(slam_grid_t *g, int v)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    g->prm.min_cluster_points = v;
    // the threshold enters every cell's occupancy: all rows are due at the next finalize
    hipLaunchKernelGGL(ranges_set_kernel, dim3(1), dim3(64), 0, nullptr, g->gv.dirty, 1, 0, 0, g->gv.sy);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int slam_grid_set_max_range(slam_grid_t *g, double v)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    g->prm.max_range = v;
    g->gv.max_range = v;
    return SLAM_OK;
}

int slam_grid_get_pose(slam_grid_t *g, double *x, double *y)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    if (x) *x = g->gv.pose_x;
    if (y) *y = g->gv.pose_y;
    return SLAM_OK;
}

int slam_grid_set_pose(slam_grid_t *g, double x, double y, slam_stream_t stream)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    GridView &v = g->gv;
    if (!v.rolling) { // mls.cpp:411-415
        v.pose_x = x;
        v.pose_y = y;
        return SLAM_OK;
    }
    const double xdiff = x - v.pose_x, ydiff = y - v.pose_y; // mls.cpp:419-424
    const int    dx = (int)std::round(xdiff / v.res), dy = (int)std::round(ydiff / v.res);
    if (dx == 0 && dy == 0) return SLAM_OK;
    // Grid::shiftOrigin, mls.h:87-97 (one wrap, as there; larger jumps clear everything anyway)
    auto wrap = [](int o, int d, int n) {
        long v2 = ((long)o + d) % n;
        if (v2 < 0) v2 += n;
        return (int)v2;
    };
    v.ox = wrap(v.ox, dx, v.sx);
    v.oy = wrap(v.oy, dy, v.sy);
    v.pose_x += dx * v.res; // mls.cpp:430-431
    v.pose_y += dy * v.res;
    g->shift_x += dx;
    g->shift_y += dy;
    hipLaunchKernelGGL(roll_clear_kernel, grid2d(g), dim3(256), 0, as_stream(stream), v, dx, dy, g->d_num_s,
                       g->d_occ_s);
    // the window moved over the storage: every row of the evidence / occupancy planes (window order) is due
    hipLaunchKernelGGL(ranges_set_kernel, dim3(1), dim3(64), 0, as_stream(stream), g->gv.dirty, 1, 0, 0, v.sy);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int slam_grid_add_endpoints_dev(slam_grid_t *g, const float *d_obs, int n_obs, const float *d_gnd, int n_gnd,
                                int stride, slam_stream_t stream)
{
    SLAM_REQUIRE(g && n_obs >= 0 && n_gnd >= 0 && stride >= 2, SLAM_E_INVALID,
                 "slam_grid_add_endpoints_dev: bad arguments");
    const int n = n_obs + n_gnd;
    if (n == 0) return SLAM_OK;
    hipLaunchKernelGGL(endpoints_kernel, dim3((n + kEndpointsPerBlock - 1) / kEndpointsPerBlock), dim3(256), 0, as_stream(stream), g->gv, d_obs,
                       n_obs, d_gnd, n_gnd, stride);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

static int stage_points(slam_grid *g, const float *obs, int n_obs, const float *gnd, int n_gnd, int stride,
                        float **d_obs, float **d_gnd)
{
    const size_t bo = (size_t)n_obs * stride * sizeof(float), bg = (size_t)n_gnd * stride * sizeof(float);
    SLAM_TRY(reserve(&g->d_stage, &g->cap_stage, bo + bg + 16));
    *d_obs = static_cast<float *>(g->d_stage);
    *d_gnd = *d_obs + (size_t)n_obs * stride;
    if (bo) SLAM_HIP(hipMemcpyAsync(*d_obs, obs, bo, hipMemcpyHostToDevice, nullptr));
    if (bg) SLAM_HIP(hipMemcpyAsync(*d_gnd, gnd, bg, hipMemcpyHostToDevice, nullptr));
    return SLAM_OK;
}

int slam_grid_add_endpoints(slam_grid_t *g, const float *obs, int n_obs, const float *gnd, int n_gnd,
                            int stride)
{
    SLAM_REQUIRE(g && n_obs >= 0 && n_gnd >= 0 && stride >= 2 && (obs || !n_obs) && (gnd || !n_gnd),
                 SLAM_E_INVALID, "slam_grid_add_endpoints: bad arguments");
    SLAM_TRY(require_device());
    float *d_obs, *d_gnd;
    SLAM_TRY(stage_points(g, obs, n_obs, gnd, n_gnd, stride, &d_obs, &d_gnd));
    SLAM_TRY(slam_grid_add_endpoints_dev(g, d_obs, n_obs, d_gnd, n_gnd, stride, nullptr));
    SLAM_HIP(hipStreamSynchronize(nullptr));
    return SLAM_OK;
}

int slam_grid_raycast_dev(slam_grid_t *g, const float *d_origin_xy, const float *d_end_xy, int n,
                          slam_stream_t stream)
{
    SLAM_REQUIRE(g && n >= 0 && (n == 0 || (d_origin_xy && d_end_xy)), SLAM_E_INVALID,
                 "slam_grid_raycast_dev: bad arguments");
    if (n == 0) return SLAM_OK;
    hipStream_t st = as_stream(stream);
    SLAM_TRY(reserve_beams(g, (size_t)n));
    const int n_chunks = (n + kChunk - 1) / kChunk;
    hipLaunchKernelGGL(beams_from_rays_kernel, dim3(n_chunks), dim3(kChunk), 0, st, g->gv,
                       reinterpret_cast<const float2 *>(d_origin_xy), reinterpret_cast<const float2 *>(d_end_xy),
                       n, g->d_beams, g->d_chunk_box);
    SLAM_HIP(hipGetLastError());
    return walk_beams(g, n, st);
}

int slam_grid_raycast(slam_grid_t *g, const float *origin_xy, const float *end_xy, int n)
{
    SLAM_REQUIRE(g && n >= 0 && (n == 0 || (origin_xy && end_xy)), SLAM_E_INVALID,
                 "slam_grid_raycast: bad arguments");
    SLAM_TRY(require_device());
    if (n == 0) return SLAM_OK;
    float *d_o, *d_e;
    SLAM_TRY(stage_points(g, origin_xy, n, end_xy, n, 2, &d_o, &d_e));
    SLAM_TRY(slam_grid_raycast_dev(g, d_o, d_e, n, nullptr));
    SLAM_HIP(hipStreamSynchronize(nullptr));
    return SLAM_OK;
}

int slam_grid_reserve(slam_grid_t *g, int max_beams)
{
    SLAM_REQUIRE(g && max_beams >= 0, SLAM_E_INVALID, "slam_grid_reserve: bad arguments");
    return max_beams ? reserve_beams(g, (size_t)max_beams) : SLAM_OK;
}

int slam_grid_raycast_scans_dev(slam_grid_t *g, const double *d_pts, const int32_t *d_scan_off, int n_scans,
                                int n_points, const double *d_R, const double *d_t, slam_stream_t stream)
{
    SLAM_REQUIRE(g && n_scans >= 0 && n_points >= 0 && d_scan_off && d_R && d_t, SLAM_E_INVALID,
                 "slam_grid_raycast_scans_dev: bad arguments");
    if (n_scans == 0 || n_points == 0) return SLAM_OK;
    hipStream_t st = as_stream(stream);
    const int   n = n_points;
    SLAM_TRY(reserve_beams(g, (size_t)n));
    const int n_chunks = (n + kChunk - 1) / kChunk;
    hipLaunchKernelGGL(beams_from_scans_kernel, dim3(n_chunks), dim3(kChunk), 0, st, g->gv,
                       reinterpret_cast<const double2 *>(d_pts), d_scan_off, n_scans, d_R, d_t, n, g->d_beams,
                       g->d_chunk_box);
    SLAM_HIP(hipGetLastError());
    return walk_beams(g, n, st);
}

int slam_grid_finalize(slam_grid_t *g, slam_stream_t stream)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    // (after the in-order mode, whose evidence lives in planes of its own, every row is recomputed)
    if (g->state_from_inorder)
        hipLaunchKernelGGL(finalize_kernel, grid2d(g), dim3(256), 0, as_stream(stream), g->gv, g->prm.occupancy_increment,
                           g->prm.occupancy_decrement, (double)g->prm.min_cluster_points, g->d_num_w, g->d_occ_w);
    else
        hipLaunchKernelGGL(finalize_rows_kernel, dim3(kRangeBlocks), dim3(256), 0, as_stream(stream), g->gv, g->prm.occupancy_increment,
                           g->prm.occupancy_decrement, (double)g->prm.min_cluster_points, g->d_num_w, g->d_occ_w, g->gv.dirty);
    hipLaunchKernelGGL(ranges_set_kernel, dim3(1), dim3(64), 0, as_stream(stream), g->gv.dirty, 0, 0, 0, g->gv.sy);
    SLAM_HIP(hipGetLastError());
    g->state_from_inorder = false;
    return SLAM_OK;
}

int slam_grid_finalize_reset(slam_grid_t *g, slam_stream_t stream)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    // A stream that is being captured into a hipGraph records this call's kernel arguments once: the switch between the two
    // range buffers below happens on the HOST, per call, and a replay would read the same stale buffer every time (rows never
    // reset or never folded, silently).  Captured, the call is the two-step form, whose arguments do not change from call to call.
    bool capturing = false;
    if (stream) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(as_stream(stream), &cap) == hipSuccess)
            capturing = cap != hipStreamCaptureStatusNone;
        else
            (void)hipGetLastError();
    }
    if (g->state_from_inorder || capturing) { // (the in-order mode keeps its evidence in planes of its own: the two steps as they are)
        SLAM_TRY(slam_grid_finalize(g, stream));
        return slam_grid_reset_counts(g, stream);
    }
    int *cur = g->gv.dirty, *next = cur == g->d_dirty ? g->d_dirty + 4 : g->d_dirty;
    hipLaunchKernelGGL(finalize_reset_rows_kernel, dim3(kRangeBlocks), dim3(256), 0, as_stream(stream), g->gv, g->prm.occupancy_increment,
                       g->prm.occupancy_decrement, (double)g->prm.min_cluster_points, g->d_num_w, g->d_occ_w, cur, next);
    SLAM_HIP(hipGetLastError());
    g->gv.dirty = next; // for every call enqueued from here on (one stream at a time per handle: the header's contract)
    return SLAM_OK;
}

// MLS::addToMap's cloud transform (mls.cpp:34-53: tf::poseMsgToEigen + pcl::transformPointCloud): every point turned by the pose's
// rotation and offset by t, computed in double and stored as float -- the arithmetic of the adapter's host loop term by term
// (products and sums rounded one by one, no contraction), so the floats are the host's.
namespace {
struct CloudTransform {
    double r[9], t[3];
};
__global__ __launch_bounds__(256) void transform_cloud_kernel(const float *in, int n, int stride, CloudTransform T, float *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double px = in[(size_t)i * stride], py = in[(size_t)i * stride + 1], pz = in[(size_t)i * stride + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        out[3 * (size_t)i + k] = (float)__dadd_rn(
            __dadd_rn(__dadd_rn(__dmul_rn(T.r[3 * k], px), __dmul_rn(T.r[3 * k + 1], py)), __dmul_rn(T.r[3 * k + 2], pz)), T.t[k]);
}
} // namespace

int slam_grid_transform_cloud_dev(const float *d_in_xyz, int n, int stride, const double R[9], const double t[3], float *d_out_xyz,
                                  slam_stream_t stream)
{
    SLAM_REQUIRE(n >= 0 && stride >= 3 && R && t && ((d_in_xyz && d_out_xyz) || n == 0), SLAM_E_INVALID,
                 "slam_grid_transform_cloud_dev: bad arguments");
    SLAM_TRY(require_device());
    if (n == 0) return SLAM_OK;
    CloudTransform T;
    for (int k = 0; k < 9; ++k) T.r[k] = R[k];
    for (int k = 0; k < 3; ++k) T.t[k] = t[k];
    hipLaunchKernelGGL(transform_cloud_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), d_in_xyz, n, stride, T, d_out_xyz);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int slam_grid_add_scan_inorder_dev(slam_grid_t *g, const float *d_obs, int n_obs, const float *d_gnd, int n_gnd, int stride,
                                   slam_stream_t stream)
{
    SLAM_REQUIRE(g && n_obs >= 0 && n_gnd >= 0 && stride >= 2 && (d_obs || !n_obs) && (d_gnd || !n_gnd), SLAM_E_INVALID,
                 "slam_grid_add_scan_inorder_dev: bad arguments");
    SLAM_TRY(require_device());
    const int n = n_obs + n_gnd;
    if (n == 0) return SLAM_OK;
    hipStream_t st = as_stream(stream);
    if (!g->d_delta) {
        SLAM_HIP(hipMalloc((void **)&g->d_delta, g->cells * sizeof(unsigned long long)));
        SLAM_HIP(hipMemsetAsync(g->d_delta, 0, g->cells * sizeof(unsigned long long), st));
    }
    {
        void  *p = g->d_touched;
        size_t cap = g->cap_touched;
        SLAM_TRY(reserve(&p, &cap, ((size_t)n + 1) * sizeof(int)));
        g->d_touched = static_cast<int *>(p);
        g->cap_touched = cap;
    }
    int *counter = g->d_touched + n;
    SLAM_HIP(hipMemsetAsync(counter, 0, sizeof(int), st));
    hipLaunchKernelGGL(inorder_count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g->gv, d_obs, n_obs, d_gnd, n_gnd, stride,
                       g->d_delta, g->d_touched, counter);
    hipLaunchKernelGGL(inorder_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g->gv, g->prm.occupancy_increment,
                       g->prm.occupancy_decrement, (double)g->prm.min_cluster_points, g->d_delta, g->d_touched, counter,
                       g->d_num_s, g->d_occ_s, g->d_updates);
    SLAM_HIP(hipGetLastError());
    g->state_from_inorder = true;
    return SLAM_OK;
}

int slam_grid_add_scan_inorder(slam_grid_t *g, const float *obs, int n_obs, const float *gnd, int n_gnd,
                               int stride)
{
    SLAM_REQUIRE(g && n_obs >= 0 && n_gnd >= 0 && stride >= 2 && (obs || !n_obs) && (gnd || !n_gnd),
                 SLAM_E_INVALID, "slam_grid_add_scan_inorder: bad arguments");
    SLAM_TRY(require_device());
    if (n_obs + n_gnd == 0) return SLAM_OK;
    float *d_obs, *d_gnd;
    SLAM_TRY(stage_points(g, obs, n_obs, gnd, n_gnd, stride, &d_obs, &d_gnd));
    SLAM_TRY(slam_grid_add_scan_inorder_dev(g, d_obs, n_obs, d_gnd, n_gnd, stride, nullptr));
    SLAM_HIP(hipStreamSynchronize(nullptr));
    return SLAM_OK;
}

int slam_grid_read_counts(slam_grid_t *g, int32_t *hits, int32_t *misses)
{
    SLAM_REQUIRE(g && hits && misses, SLAM_E_INVALID, "slam_grid_read_counts: bad arguments");
    SLAM_TRY(require_device());
    int32_t *tmp = nullptr;
    SLAM_HIP(hipMalloc((void **)&tmp, 2 * g->cells * sizeof(int32_t)));
    hipLaunchKernelGGL(gather_counts_kernel, grid2d(g), dim3(256), 0, nullptr, g->gv, tmp, tmp + g->cells);
    hipError_t e = hipMemcpy(hits, tmp, g->cells * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(misses, tmp + g->cells, g->cells * sizeof(int32_t), hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    SLAM_HIP(e);
    return SLAM_OK;
}

static int sync_window_state(slam_grid *g)
{
    if (g->state_from_inorder) {
        hipLaunchKernelGGL(window_from_storage_kernel, grid2d(g), dim3(256), 0, nullptr, g->gv, g->d_num_s,
                           g->d_occ_s, g->d_num_w, g->d_occ_w);
        SLAM_HIP(hipGetLastError());
    }
    return SLAM_OK;
}

int slam_grid_read_occupancy(slam_grid_t *g, int8_t *occ)
{
    SLAM_REQUIRE(g && occ, SLAM_E_INVALID, "slam_grid_read_occupancy: bad arguments");
    SLAM_TRY(require_device());
    SLAM_TRY(sync_window_state(g));
    SLAM_HIP(hipMemcpy(occ, g->d_occ_w, g->cells, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

int slam_grid_read_num_pts(slam_grid_t *g, double *num_pts)
{
    SLAM_REQUIRE(g && num_pts, SLAM_E_INVALID, "slam_grid_read_num_pts: bad arguments");
    SLAM_TRY(require_device());
    SLAM_TRY(sync_window_state(g));
    SLAM_HIP(hipMemcpy(num_pts, g->d_num_w, g->cells * sizeof(double), hipMemcpyDeviceToHost));
    return SLAM_OK;
}

int slam_grid_total_updates(slam_grid_t *g, uint64_t *n)
{
    SLAM_REQUIRE(g && n, SLAM_E_INVALID, "slam_grid_total_updates: bad arguments");
    SLAM_TRY(require_device());
    std::vector<unsigned long long> v(kUpdateSlots, 0);
    SLAM_HIP(hipMemcpy(v.data(), g->d_updates, kUpdateSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    uint64_t sum = 0;
    for (unsigned long long x : v) sum += x;
    *n = sum;
    return SLAM_OK;
}

int slam_grid_info(slam_grid_t *g, int *size_x, int *size_y, double *resolution, int *origin_x, int *origin_y)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    if (size_x) *size_x = g->gv.sx;
    if (size_y) *size_y = g->gv.sy;
    if (resolution) *resolution = g->gv.res;
    if (origin_x) *origin_x = g->gv.ox;
    if (origin_y) *origin_y = g->gv.oy;
    return SLAM_OK;
}

int slam_grid_window_cell(slam_grid_t *g, int *cell_x, int *cell_y)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    if (cell_x) *cell_x = (int)g->shift_x;
    if (cell_y) *cell_y = (int)g->shift_y;
    return SLAM_OK;
}

int slam_grid_raycast_stats(slam_grid_t *g, int *n_tiles, int *n_items, int *n_segments)
{
    SLAM_REQUIRE(g, SLAM_E_INVALID, "null handle");
    const int tiles = ((g->gv.sx + kTile - 1) / kTile) * ((g->gv.sy + kTile - 1) / kTile);
    if (n_tiles) *n_tiles = tiles;
    if (n_items) *n_items = 0;
    if (n_segments) *n_segments = 0;
    if (!g->d_tile_fill) return SLAM_OK; // no tiled raycast has run yet
    SLAM_HIP(hipDeviceSynchronize());
    std::vector<int> cnt((size_t)tiles);
    SLAM_HIP(hipMemcpy(cnt.data(), g->d_tile_cnt, sizeof(int) * (size_t)tiles, hipMemcpyDeviceToHost));
    long items = 0;
    for (int c : cnt) items += c;
    if (n_items) *n_items = (int)items;
    if (n_segments) SLAM_HIP(hipMemcpy(n_segments, g->d_cursor + tiles, sizeof(int), hipMemcpyDeviceToHost));
    return SLAM_OK;
}

int slam_grid_counts_dev(slam_grid_t *g, int32_t **d_planes, size_t *n_ints)
{
    SLAM_REQUIRE(g && d_planes, SLAM_E_INVALID, "slam_grid_counts_dev: bad arguments");
    *d_planes = g->d_planes;
    if (n_ints) *n_ints = 2 * g->cells;
    return SLAM_OK;
}

} // extern "C"
