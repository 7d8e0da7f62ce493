// ccicp.hip -- the CCICP facade steps either side of the ICP call:
//   CCICP::setSceneCloud voxel filter     ccicp2d/src/icpTools.cpp:620-633  (pcl::VoxelGrid, leaf 0.5,0.5,2)
//   CCICP::doICPMatch crop + split + cap  ccicp2d/src/icpTools.cpp:225-276  (pcl::PassThrough, isGA, ICP_MAX_PTS-1)
//   CCICP::doHeightInterpolate            ccicp2d/src/icpTools.cpp:301-381  (KdTreeFLANN 1-NN, plane normal)
// PCL is not part of the reference checkout: the published PCL 1.7 algorithms are restated
// (oracle/ccicp_oracle.c); parity is unpinned there and tolerance-based where PCL's float sums are
// order-dependent.  Everything stays on the device between ground segmentation (gseg.hip) and
// slam_icp_create / slam_icp_fit_batch_dev; results are deterministic (integer sums, index-ordered output).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <new>
#include <vector>

#include <rocprim/rocprim.hpp> // device radix sort (the stable bin order of classifyPoints)

#include "common.hpp"

using namespace slam;

namespace {

constexpr int kItems = 1024; // cells / points per compaction block (round 6: 4096 -- a 131 072-point cloud was 32 blocks on 256 CUs)
constexpr int kScanThreads = 256;
constexpr double kFix = 16777216.0; // 2^24: coordinates summed as 64-bit fixed point (exact, order-free)

struct Voxel { // 32 bytes
    long long          sx, sy, sz;
    unsigned           sflag, count; // updated together as one 64-bit word (sflag in the low half)
};
static_assert(sizeof(Voxel) == 32 && offsetof(Voxel, count) == offsetof(Voxel, sflag) + 4 && offsetof(Voxel, sflag) % 8 == 0, "Voxel layout");

struct VoxelGridView {
    int       min_b[3], div_b[3];
    float     inv[3];
    long long n_vox;
};

__device__ inline unsigned order_f32(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float unorder_f32(unsigned u)
{
    const unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float          f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __uint_as_float(v);
#else
    memcpy(&f, &v, 4);
#endif
    return f;
}

__device__ inline bool finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

// getMinMax3D over finite points: wave reduction, one atomic per wavefront and bound
// `n` is the launch's bound; where the number of points is only known on the device (the chain of slam_ccicp_scene_dev)
// d_n holds it and n is the capacity the grid was sized for
__device__ inline int bound(int n, const int *d_n) { return d_n ? min(*d_n, n) : n; }

__global__ __launch_bounds__(256) void minmax_kernel(const float *xyz, const unsigned char *flag, int n, int stride,
                                                     unsigned *mm /*[6]*/, const int *d_n = nullptr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    n = bound(n, d_n);
    unsigned  lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    if (i < n) {
        const float *p = xyz + (size_t)i * stride;
        if (finite3(p) && !(flag && flag[i] == 255)) // 255: dropped by classifyPoints (icpTools.cpp:60,72-77)
            for (int d = 0; d < 3; ++d) lo[d] = hi[d] = order_f32(p[d]);
    }
    // wavefront, then block (LDS), then one atomic per block and bound -- and only where it would change the value: all
    // six words share a cache line, and every same-line atomic costs ~5-10 ns (1100 wavefronts x 6 took 35 us)
    __shared__ unsigned red[4][6];
    for (int d = 0; d < 3; ++d) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = min(lo[d], (unsigned)__shfl_xor((int)lo[d], off));
            hi[d] = max(hi[d], (unsigned)__shfl_xor((int)hi[d], off));
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][d] = lo[d], red[threadIdx.x >> 6][3 + d] = hi[d];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        unsigned  v = red[0][d];
        for (int w = 1; w < 4; ++w) v = d < 3 ? min(v, red[w][d]) : max(v, red[w][d]);
        if (d < 3) {
            if (v != 0xffffffffu && v < __hip_atomic_load(&mm[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mm[d], v);
        } else {
            if (v != 0u && v > __hip_atomic_load(&mm[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&mm[d], v);
        }
    }
}

__device__ inline long long voxel_of(const VoxelGridView &g, const float *p)
{
    // static_cast<int>(floor(pt.x * inverse_leaf_size_[0]) - static_cast<float>(min_b_[0])), voxel_grid.hpp
    const int i = (int)(floorf(p[0] * g.inv[0]) - (float)g.min_b[0]);
    const int j = (int)(floorf(p[1] * g.inv[1]) - (float)g.min_b[1]);
    const int k = (int)(floorf(p[2] * g.inv[2]) - (float)g.min_b[2]);
    return (long long)i + (long long)j * g.div_b[0] + (long long)k * g.div_b[0] * g.div_b[1];
}

// Points of a cloud come ring by ring, azimuth by azimuth: near the sensor, where the voxels are fullest, adjacent
// lanes fall into the same voxel.  Runs of equal voxels in adjacent lanes are added up inside the wavefront (a
// segmented scan over the run heads) and the run's last lane issues the atomics: integer sums, so the result does not
// depend on who adds -- and the hot voxels see a fraction of the same-address atomics (80 -> 35 us per 70 k-point cloud).
__device__ inline VoxelGridView voxel_geometry(const unsigned *mm, float leaf_x, float leaf_y, float leaf_z, long long capacity, bool *overflow);
__global__ __launch_bounds__(256) void voxel_accumulate_kernel(VoxelGridView g, const float *xyz, const unsigned char *flag,
                                                               int n, int stride, Voxel *vox, VoxelGridView *d_g = nullptr,
                                                               const int *d_n = nullptr, const unsigned *d_mm = nullptr,
                                                               float3 leaf = make_float3(0.f, 0.f, 0.f), long long capacity = 0, int *d_err = nullptr)
{
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    if (d_g && !d_mm) g = *d_g; // the lattice worked out on the device (voxel_geometry_kernel)
    if (d_mm) {
        // ... or here, by every block for itself from the extent the classification left (the chain of slam_ccicp_scene_dev: one
        // launch less); block 0 leaves it for the compaction behind this kernel
        __shared__ VoxelGridView gs;
        if (threadIdx.x == 0) {
            bool overflow = false;
            gs = voxel_geometry(d_mm, leaf.x, leaf.y, leaf.z, capacity, &overflow);
            if (blockIdx.x == 0) {
                *d_g = gs;
                if (overflow) atomicOr(d_err, 1);
            }
        }
        __syncthreads();
        g = gs;
    }
    long long          key = -1; // no contribution
    unsigned long long sx = 0, sy = 0, sz = 0, cf = 0; // cf: count in the high word, ground_adj count in the low (Voxel::sflag, ::count)
    if (i < bound(n, d_n)) {
        const float *p = xyz + (size_t)i * stride;
        if (finite3(p) && !(flag && flag[i] == 255)) {
            const long long v = voxel_of(g, p);
            if (v >= 0 && v < g.n_vox) {
                key = v;
                sx = (unsigned long long)llrint((double)p[0] * kFix);
                sy = (unsigned long long)llrint((double)p[1] * kFix);
                sz = (unsigned long long)llrint((double)p[2] * kFix);
                const unsigned f = flag ? (flag[i] == 1 ? 1u : 0u) : (stride > 3 ? (p[3] > 0.5f ? 1u : 0u) : 0u);
                cf = (1ull << 32) | f;
            }
        }
    }
    const long long prev = __shfl_up(key, 1), next = __shfl_down(key, 1);
    bool            head = lane == 0 || prev != key;
    const bool      last = lane == 63 || next != key;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { // segmented inclusive scan: every lane ends with the sum from its run's head to itself
        const unsigned long long ux = __shfl_up(sx, d), uy = __shfl_up(sy, d), uz = __shfl_up(sz, d), uc = __shfl_up(cf, d);
        const bool               uh = __shfl_up((int)head, d) != 0;
        if (lane >= d && !head) {
            sx += ux;
            sy += uy;
            sz += uz;
            cf += uc;
            head = uh;
        }
    }
    if (last && key >= 0) {
        Voxel *c = vox + key;
        atomicAdd((unsigned long long *)&c->sx, sx);
        atomicAdd((unsigned long long *)&c->sy, sy);
        atomicAdd((unsigned long long *)&c->sz, sz);
        atomicAdd((unsigned long long *)&c->sflag, cf); // {sflag, count} as one 64-bit word: sflag <= count, no carry across
    }
}

// ---- stable compaction (compact1_kernel below)
// the domain of a compaction: n items, or fewer where a count on the device says so
struct Domain {
    long long        n;
    const int       *d_n;   // nullable
    const long long *d_n64; // nullable
    __device__ long long size() const
    {
        long long m = n;
        if (d_n) m = min(m, (long long)*d_n);
        if (d_n64) m = min(m, *d_n64);
        return m;
    }
};

struct VoxelUsed {
    const Voxel *vox;
    __device__ bool operator()(long long v) const { return vox[v].count != 0; }
};
struct VoxelEmit {
    const Voxel *vox;
    float       *out;
    __device__ void operator()(long long v, int pos) const
    {
        const Voxel  c = vox[v];
        const double inv = 1.0 / ((double)c.count * kFix);
        out[4 * (size_t)pos + 0] = (float)((double)c.sx * inv);
        out[4 * (size_t)pos + 1] = (float)((double)c.sy * inv);
        out[4 * (size_t)pos + 2] = (float)((double)c.sz * inv);
        // PCL averages the uint16 ground_adj as a float and stores it back to the uint16 field (truncation)
        out[4 * (size_t)pos + 3] = (float)(unsigned short)((float)c.sflag / (float)c.count);
    }
};

struct VoxelEmitClean { // ... and leaves the voxel as the next cloud must find it
    Voxel *vox;
    float *out;
    __device__ void operator()(long long v, int pos) const
    {
        VoxelEmit{vox, out}(v, pos);
        Voxel z;
        z.sx = z.sy = z.sz = 0;
        z.sflag = z.count = 0;
        vox[v] = z;
    }
};

// points whose label is in a mask, in cloud order, as (x, y, z, 0) records
struct LabelPred {
    const unsigned char *labels;
    unsigned             mask;
    __device__ bool operator()(long long i) const { return (mask >> labels[i]) & 1u; }
};
struct Xyz4Emit {
    const float *xyz;
    int          stride;
    float4      *out;
    __device__ void operator()(long long i, int pos) const
    {
        const float *q = xyz + (size_t)i * stride;
        out[pos] = make_float4(q[0], q[1], q[2], 0.f);
    }
};

// crop (pcl::PassThrough x then y) + class predicate of the split
struct SplitPred {
    const float *xyzg;
    int          stride, want_ga;
    float        x_lo, x_hi, y_lo, y_hi;
    int          crop;
    __device__ bool operator()(long long i) const
    {
        const float *p = xyzg + (size_t)i * stride;
        if (crop && !(finite3(p) && p[0] >= x_lo && p[0] <= x_hi && p[1] >= y_lo && p[1] <= y_hi)) return false;
        return (p[3] > 0.5f) == (want_ga != 0); // isGA, PointcloudXYZGD.h:28-30
    }
};
struct SplitEmit {
    const float *xyzg;
    int          stride;
    double      *out;
    const int   *d_base; // nullable: points already in `out` (capped at `base_cap`): this class is written behind them
    int          base_cap;
    __device__ void operator()(long long i, int pos) const
    {
        const float *p = xyzg + (size_t)i * stride;
        const size_t o = (size_t)pos + (d_base ? (size_t)min(*d_base, base_cap) : 0);
        out[2 * o] = (double)p[0]; // icpTools.cpp:252, 267: float coordinates widened
        out[2 * o + 1] = (double)p[1];
    }
};

// four wheel points against all ground points: exact squared L2 in float (KdTreeFLANN, k = 1), packed
// (distance bits, index) minimum -> lowest index on a tie
struct HeightPose { // the pose the wheel points come from, where it lies on the device (null R: the points are given)
    const double *R, *t;
    double        z0, roll, pitch;
};
__device__ inline void height_pose(const double *R, const double *t, double z0, double roll, double pitch, float4 *q);
__global__ __launch_bounds__(256) void height_nn_kernel(const float *ground, int n, int stride, float4 q0, float4 q1,
                                                        float4 q2, float4 q3, unsigned long long *best /*[4]*/,
                                                        const float4 *d_q = nullptr, const int *d_n = nullptr, HeightPose hp = HeightPose{nullptr, nullptr, 0, 0, 0})
{
    const int          i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long b[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    __shared__ float4  qs[4];
    if (hp.R) { // every block works the four wheel points out for itself (a launch of one thread did: round 6, a match is bound by its launches)
        if (threadIdx.x == 0) height_pose(hp.R, hp.t, hp.z0, hp.roll, hp.pitch, qs);
        __syncthreads();
    }
    if (i < bound(n, d_n)) {
        const float *c = ground + (size_t)i * stride;
        const float4 q[4] = {hp.R ? qs[0] : (d_q ? d_q[0] : q0), hp.R ? qs[1] : (d_q ? d_q[1] : q1), hp.R ? qs[2] : (d_q ? d_q[2] : q2),
                             hp.R ? qs[3] : (d_q ? d_q[3] : q3)};
        for (int k = 0; k < 4; ++k) {
            const float dx = c[0] - q[k].x, dy = c[1] - q[k].y, dz = c[2] - q[k].z;
            const float dd = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if (dd == dd) b[k] = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)i; // dd >= 0: bits order as values
        }
    }
    __shared__ unsigned long long red[4][4];
    for (int k = 0; k < 4; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(b[k], off);
            b[k] = o < b[k] ? o : b[k];
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = b[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) { // one atomic per block and wheel point, where it improves on what is there (see minmax_kernel)
        const int          k = threadIdx.x;
        unsigned long long v = red[0][k];
        for (int w = 1; w < 4; ++w) v = red[w][k] < v ? red[w][k] : v;
        if (v != ~0ull && v < __hip_atomic_load(&best[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&best[k], v);
    }
}

// classifyPoints rebuilds the cloud bin by bin (x bin major, y bin minor, icpTools.cpp:64-101), points of a bin in
// their original order: key = (bin << 32 | index), so sorted keys are that order.  Points it drops (outside the
// 1200 x 1200 lattice :60, edge cells :72-77 -- flag 255 from slam_gseg_classify_ga_dev) get the last key.
constexpr int kGaBins = 1200; // icpTools.h:24-26
__global__ __launch_bounds__(256) void bin_keys_kernel(const float *xyz, const unsigned char *flag, int n, int stride,
                                                       unsigned long long *keys, const int *d_n = nullptr)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i >= bound(n, d_n)) { // past the cloud's end (its size is known on the device only): sorts behind everything kept
        keys[i] = (0x1fffffull << 32) | (unsigned)i;
        return;
    }
    const float *q = xyz + (size_t)i * stride;
    const double RES = 0.5, offset = (double)kGaBins * RES / 2;
    const double fx = floor(((double)q[0] + offset) / RES), fy = floor(((double)q[1] + offset) / RES); // :57-58
    unsigned long long bin = 0x1fffffull;
    if (flag[i] != 255 && fx >= 0 && fx < kGaBins && fy >= 0 && fy < kGaBins) bin = (unsigned long long)((int)fx * kGaBins + (int)fy);
    keys[i] = (bin << 32) | (unsigned)i;
}

__global__ __launch_bounds__(256) void bin_gather_kernel(const float *xyz, const unsigned char *flag, int stride,
                                                         const unsigned long long *keys, int n, float4 *out, int *n_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const bool               kept = (k >> 32) != 0x1fffffull;
    if (kept) {
        const unsigned src = (unsigned)(k & 0xffffffffu);
        const float   *q = xyz + (size_t)src * stride;
        out[i] = make_float4(q[0], q[1], q[2], flag[src] == 1 ? 1.f : 0.f);
    }
    // kept points come first: the boundary is where the count stands
    const bool next_kept = i + 1 < n && (keys[i + 1] >> 32) != 0x1fffffull;
    if (kept && !next_kept) *n_out = i + 1;
}

// ---- the chain of slam_ccicp_scene_dev: what the stepwise entry points work out on the host, on the device
// the voxel lattice from the extent (slam_ccicp_voxel_downsample_dev's host code, PCL voxel_grid.hpp); err |= 1 when it
// does not fit the accumulator (n_vox = 0 then: nothing is accumulated)
__device__ inline VoxelGridView voxel_geometry(const unsigned *mm, float leaf_x, float leaf_y, float leaf_z, long long capacity, bool *overflow)
{
    VoxelGridView v;
    const float   leaf[3] = {leaf_x, leaf_y, leaf_z};
    long long     nv = 1;
    bool          bad = mm[0] == 0xffffffffu; // no finite point
    for (int d = 0; d < 3; ++d) {
        v.inv[d] = 1.0f / leaf[d];
        const double lo = floor((double)(unorder_f32(mm[d]) * v.inv[d]));
        const double hi = floor((double)(unorder_f32(mm[3 + d]) * v.inv[d]));
        if (bad || !(fabs(lo) < 1e9) || !(hi - lo + 1.0 <= (double)capacity)) {
            bad = true;
            v.min_b[d] = 0;
            v.div_b[d] = 0;
            continue;
        }
        v.min_b[d] = (int)lo;
        v.div_b[d] = (int)(hi - lo) + 1;
        nv *= v.div_b[d];
        if (nv <= 0 || nv > capacity) bad = true;
    }
    v.n_vox = bad ? 0 : nv;
    *overflow = bad && mm[0] != 0xffffffffu;
    return v;
}

__global__ void voxel_geometry_kernel(const unsigned *mm, float leaf_x, float leaf_y, float leaf_z, long long capacity,
                                      VoxelGridView *g, int *err)
{
    if (threadIdx.x || blockIdx.x) return;
    bool overflow = false;
    *g = voxel_geometry(mm, leaf_x, leaf_y, leaf_z, capacity, &overflow);
    if (overflow) atomicOr(err, 1);
}

__global__ __launch_bounds__(256) void voxel_zero_kernel(const VoxelGridView *g, Voxel *vox)
{
    const long long nv = g->n_vox;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        Voxel z;
        z.sx = z.sy = z.sz = 0;
        z.sflag = z.count = 0;
        vox[i] = z;
    }
}

// {0, n_ga + n_nga, n_ga}: scan_off[0..1] and scan_nga[0] of the one scan slam_icp_fit_batch_dev then registers; the
// class totals are capped as CCICP::doICPMatch caps them (ICP_MAX_PTS - 1, icpTools.cpp:256,259)
__global__ void scene_scan_kernel(const int *tot /*[2]*/, int cap, const int *n_obs, const int *n_gnd, const int *n_flt, int *scan,
                                  int *counts)
{
    if (threadIdx.x || blockIdx.x) return;
    const int ga = min(tot[0], cap - 1), nga = min(tot[1], cap - 1);
    scan[0] = 0;
    scan[1] = ga + nga;
    scan[2] = ga;
    counts[0] = *n_obs;
    counts[1] = *n_gnd;
    counts[2] = *n_flt;
}

// doHeightInterpolate's four wheel points from a pose held on the device: R (2 x 2), t of the match and z0; the
// quaternion of the yaw (qz = sin(yaw / 2), qw = cos(yaw / 2)) goes through the same arithmetic as the host form
__host__ __device__ inline void wheel_points(const double pose[7], float4 q[4])
{
    const double ROBO_HEIGHT = 1.45, wheel = 0.5; // icpTools.cpp:303-305
    const double x = pose[3], y = pose[4], z = pose[5], w = pose[6];
    const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;
    const double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys,
                 xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    const float M[3][4] = {{(float)(1.0 - (yy + zz)), (float)(xy - wz), (float)(xz + wy), (float)pose[0]},
                           {(float)(xy + wz), (float)(1.0 - (xx + zz)), (float)(yz - wx), (float)pose[1]},
                           {(float)(xz - wy), (float)(yz + wx), (float)(1.0 - (xx + yy)), (float)pose[2]}};
    int k = 0;
    for (int i = -1; i <= 1; i += 2)
        for (int j = -1; j <= 1; j += 2, ++k) { // :311-318
            const float p[3] = {(float)(i * wheel), (float)(j * wheel), (float)(-1.0 * ROBO_HEIGHT)};
            float       t[3];
            for (int r = 0; r < 3; ++r) t[r] = M[r][0] * p[0] + M[r][1] * p[1] + M[r][2] * p[2] + M[r][3];
            q[k] = make_float4(t[0], t[1], t[2], 0.f);
        }
}

__device__ inline void height_pose(const double *R, const double *t, double z0, double roll, double pitch, float4 *q /*[4]*/)
{
    const double yaw = atan2(R[2], R[0]); // icpTools.cpp:195-197
    const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    const double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
    const double pose[7] = {t[0], t[1], z0, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                            cr * cp * cy + sr * sp * sy};
    wheel_points(pose, q);
}

__global__ void height_pose_kernel(const double *R, const double *t, double z0, double roll, double pitch, float4 *q /*[4]*/,
                                   unsigned long long *best)
{
    if (threadIdx.x || blockIdx.x) return;
    const double yaw = atan2(R[2], R[0]); // icpTools.cpp:195-197
    // tf::createQuaternionFromRPY(roll, pitch, yaw): the matched yaw with the roll and pitch the initial pose carried (:205-212)
    const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    const double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
    const double pose[7] = {t[0], t[1], z0, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                            cr * cp * cy + sr * sp * sy};
    wheel_points(pose, q);
    for (int k = 0; k < 4; ++k) best[k] = ~0ull;
}

// 3x3 symmetric: eigenvector of the smallest eigenvalue by Jacobi sweeps (four points: one thread, host or device)
__host__ __device__ inline void smallest_eigvec3(double A[3][3], double v[3])
{
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        // converged to rounding: the off-diagonal part is 1e-22 of the diagonal (a sweep squares it; waiting for it to
        // underflow took five sweeps more, 50 us of one GPU thread's f64 divisions and square roots)
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300 || off <= 1e-22 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]))) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double th = 0.5 * (A[q][q] - A[p][p]) / A[p][q];
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    for (int k = 1; k < 3; ++k)
        if (A[k][k] < A[m][m]) m = k;
    for (int k = 0; k < 3; ++k) v[k] = V[k][m];
}


// the plane under the wheel points and the height it gives (icpTools.cpp:345-376), from the four packed nearest
// neighbours; *n_corr = neighbours within 3 m, z stays z0 below four of them
__host__ __device__ inline double height_from_neighbours(const float corr[4][3], int nc, double z0)
{
    const double ROBO_HEIGHT = 1.45;
    if (nc < 4) return z0; // :351,:379 "Height could not be determined"
    double mean[3] = {0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 3; ++r) mean[r] += (double)corr[i][r];
    for (int r = 0; r < 3; ++r) mean[r] /= 4.0;
    double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) C[r][c] += ((double)corr[i][r] - mean[r]) * ((double)corr[i][c] - mean[c]);
    double nrm[3];
    smallest_eigvec3(C, nrm); // computePointNormal -> solvePlaneParameters (:361-365)
    if (nrm[0] != nrm[0] || nrm[1] != nrm[1] || nrm[2] != nrm[2]) return z0; // :367
    if (nrm[2] < 0) nrm[2] = -nrm[2];                                          // :369-372
    return (double)(float)((float)nrm[2] * ROBO_HEIGHT + (float)mean[2]);      // :376
}

__global__ void height_fit_kernel(const float *ground, int stride, unsigned long long *best, double z0, double *out /*[2]*/, bool clear = false,
                                  unsigned long long *mirror_dst = nullptr, const unsigned long long *mirror_src = nullptr, int mirror_words = 0)
{
    if (threadIdx.x || blockIdx.x) return;
    const unsigned long long b[4] = {best[0], best[1], best[2], best[3]};
    if (clear)
        for (int k = 0; k < 4; ++k) best[k] = ~0ull; // as the next call's neighbour search must find them
    float                    all[4][3];
    for (int k = 0; k < 4; ++k) { // four independent gathers (index 0 where there is no neighbour: read, not used)
        const size_t idx = b[k] == ~0ull ? 0 : (size_t)(unsigned)(b[k] & 0xffffffffu);
        for (int r = 0; r < 3; ++r) all[k][r] = ground ? ground[idx * stride + r] : 0.0f; // (an empty cloud has no buffer)
    }
    float corr[4][3];
    int   nc = 0;
    for (int k = 0; k < 4; ++k)
        if (b[k] != ~0ull && __uint_as_float((unsigned)(b[k] >> 32)) < 9.0f) { // :345
            for (int r = 0; r < 3; ++r) corr[nc][r] = all[k][r];
            ++nc;
        }
    out[0] = height_from_neighbours(corr, nc, z0);
    out[1] = (double)nc;
    // the caller's result block, as it stands now, to where the host reads it (pinned memory: no copy behind the match)
    for (int k = 0; k < mirror_words; ++k) mirror_dst[k] = mirror_src[k];
}

// ---- stable compaction in ONE launch (round 6): per-block counts, a decoupled look-back for the block's offset, the ordered
// write -- for up to two outputs fed from one pass over the input (pred(i) = the output item i goes to, -1: none).  A match of
// config 3 is bound by its launches (45 of 3-47 us per cloud in round 5): the five compactions of the scene chain were fifteen of them.
// status[1 + c * nb + b]: {epoch : 34 | flag : 2 | value : 28} of output c and block b -- flag 1: the block's own count, 2: the count of
// all blocks up to and including it; a word of another epoch is a word not yet written (no fill between launches).  The epoch lives
// on the DEVICE, in status[0]: every block reads it when it starts, the last block -- which has seen a word of every other block,
// written after that block's read -- moves it on when the launch is done.  (It was a counter of the handle passed by value until a
// hipGraph replay of the scene chain froze it: the replay took last replay's words for this one's.)  Blocks are dispatched in index
// order and publish their count before they wait for anything, so a block only ever waits for blocks that are running or done (the
// grids are at most a few hundred blocks: all resident).  CONCAT: output 1 is written BEHIND output 0 (min(total 0, limit) on): its
// blocks wait for the last block's prefix of output 0.
constexpr int                kStatValueBits = 28, kStatEpochShift = 30;
constexpr unsigned long long kStatAgg = 1ull << kStatValueBits, kStatPre = 2ull << kStatValueBits, kStatValueMask = (1ull << kStatValueBits) - 1ull,
                             kStatEpochMask = (1ull << (64 - kStatEpochShift)) - 1ull;
constexpr unsigned kSpinLimit = 1u << 22; // (a safeguard, not a path: a few seconds, then the error bit and whatever is there)
typedef unsigned long long __attribute__((address_space(1))) g_u64;
__device__ inline unsigned long long stat_load(const unsigned long long *p)
{
    return __hip_atomic_load((g_u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void stat_store(unsigned long long *p, unsigned long long epoch, unsigned long long flag, unsigned v)
{
    __hip_atomic_store((g_u64 *)p, (epoch << kStatEpochShift) | flag | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned long long stat_wait(const unsigned long long *p, unsigned long long epoch, unsigned long long need, int *err)
{
    unsigned long long w = stat_load(p);
    for (unsigned spin = 0; (w >> kStatEpochShift) != epoch || !(w & need); ++spin) {
        if (spin > kSpinLimit) {
            if (err) atomicOr(err, 4);
            return (epoch << kStatEpochShift) | kStatPre;
        }
        __builtin_amdgcn_s_sleep(1);
        w = stat_load(p);
    }
    return w;
}

template <int NC, bool CONCAT, class Pred, class Emit, class Tail>
__global__ __launch_bounds__(kScanThreads) void compact1_kernel(Pred pred, Emit emit, Tail tail, Domain dom, unsigned long long *status_epoch,
                                                               int limit, int *err)
{
    static_assert(NC == 1 || NC == 2, "one or two outputs");
    const long long n = dom.size();
    constexpr int   kPer = kItems / kScanThreads;
    const long long base = (long long)blockIdx.x * kItems + (long long)threadIdx.x * kPer;
    // the blocks that have items (the launch is sized for the capacity, the count may be the device's: a voxel lattice of a few
    // thousand cells in an accumulator of two million): the others leave at once, the last of THESE runs the tail
    const int nb = (int)min((long long)gridDim.x, max((n + kItems - 1) / kItems, 1ll));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    if (b >= nb) return;
    const unsigned long long epoch = stat_load(status_epoch) & kStatEpochMask; // (the same in every block of the launch: see above)
    unsigned long long      *status = status_epoch + 1;
    unsigned        mask[NC];
    int             cnt[NC], x[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) mask[c] = 0;
    for (int k = 0; k < kPer; ++k)
        if (base + k < n) {
            const int c = pred(base + k);
            if (c == 0) mask[0] |= 1u << k;
            if (NC > 1 && c == 1) mask[NC - 1] |= 1u << k;
        }
    __shared__ int wsum[NC][kScanThreads / 64], excl_s[NC], base1_s;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        cnt[c] = __popc(mask[c]);
        x[c] = cnt[c];
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x[c], off);
            if (lane >= off) x[c] += y;
        }
        if (lane == 63) wsum[c][wave] = x[c];
    }
    __syncthreads();
    int w[NC], agg[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        w[c] = agg[c] = 0;
        for (int k = 0; k < kScanThreads / 64; ++k) {
            w[c] += k < wave ? wsum[c][k] : 0;
            agg[c] += wsum[c][k];
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            unsigned long long *st = status + (size_t)c * nb;
            if (lane == 0) stat_store(st + b, epoch, b == 0 ? kStatPre : kStatAgg, (unsigned)agg[c]);
            int excl = 0;
            for (int look = b - 1; look >= 0; look -= 64) {
                const int                idx = look - lane;
                const unsigned long long wd = idx >= 0 ? stat_wait(st + idx, epoch, kStatAgg | kStatPre, err) : ((epoch << kStatEpochShift) | kStatPre);
                // the nearest block (lowest lane) that knows its inclusive prefix ends the walk: the counts of the blocks nearer than
                // it plus that prefix (lanes beyond block 0 stand for a prefix of nothing)
                const unsigned long long pre = __ballot((wd & kStatPre) != 0);
                const int                first = pre ? __builtin_ctzll(pre) : 64;
                int                      v = lane <= first ? (int)(wd & kStatValueMask) : 0;
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
                excl += v;
                if (pre) break;
            }
            if (lane == 0) {
                if (b > 0) stat_store(st + b, epoch, kStatPre, (unsigned)(excl + agg[c]));
                excl_s[c] = excl;
            }
        }
    }
    __syncthreads();
    int pos[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) pos[c] = excl_s[c] + w[c] + x[c] - cnt[c];
    if (NC > 1 && CONCAT) {
        if (threadIdx.x == 0) base1_s = min((int)(stat_wait(status + (nb - 1), epoch, kStatPre, err) & kStatValueMask), limit);
        __syncthreads();
        pos[NC - 1] += base1_s;
    }
    for (int k = 0; k < kPer; ++k) {
        if (mask[0] & (1u << k)) {
            if (pos[0] < limit) emit(base + k, 0, pos[0]);
            ++pos[0];
        }
        if (NC > 1 && (mask[NC - 1] & (1u << k))) {
            if (pos[NC - 1] - (CONCAT ? base1_s : 0) < limit) emit(base + k, 1, pos[NC - 1]);
            ++pos[NC - 1];
        }
    }
    if (b == nb - 1 && threadIdx.x == 0) {
        int tot[2] = {excl_s[0] + agg[0], NC > 1 ? excl_s[NC - 1] + agg[NC - 1] : 0};
        tail(tot);
        // the launch after this one is another epoch (every block of THIS launch has read its own: the look-back above saw them all)
        __hip_atomic_store((g_u64 *)status_epoch, (epoch + 1ull) & kStatEpochMask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct DevBuf {
    void  *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SLAM_OK;
        // a quarter more than asked: clouds of a sequence differ by a few per cent, and every growth is a free (which waits
        // for the device) and an allocation
        const size_t want = bytes + bytes / 4;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        SLAM_HIP(hipMalloc(&p, want));
        cap = want;
        return SLAM_OK;
    }
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
};

struct NoTail {
    int *total0, *total1;
    __device__ void operator()(const int tot[2]) const
    {
        if (total0) *total0 = tot[0];
        if (total1) *total1 = tot[1];
    }
};
template <class Pred>
struct OneOutput { // a yes/no predicate as the class function of a one-output compaction
    Pred pred;
    __device__ int operator()(long long i) const { return pred(i) ? 0 : -1; }
};
template <class Emit>
struct OneEmit {
    Emit emit;
    __device__ void operator()(long long i, int, int pos) const { emit(i, pos); }
};

} // namespace

struct slam_ccicp {
    DevBuf vox, small;                  // small: 6 min/max words, totals, 4 packed NN results
    DevBuf status;                      // the one-launch compactions' epoch (word 0, kept by the kernels) and look-back words (compact1)
    size_t vox_clean = 0;               // voxels of `vox` known to be zero (the chain's compaction leaves them so)
    const void *best_of = nullptr;      // the chain block whose packed neighbours (ChainSmall::best) have been set to "none"
    DevBuf keys, sort_tmp;
    long long max_voxels = 1ll << 26;
    // the chain (slam_ccicp_scene_dev): per-point scratch for the cloud's capacity, the lattice and the counts on the device
    DevBuf labels, obs, flags, filtered, chain; // chain: VoxelGridView, counts, wheel points, packed neighbours
    long long chain_voxels = 1ll << 21;         // accumulator capacity of the chain (64 MB): 0.5 x 0.5 x 2 m over 360 x 360 x 30 m
};

namespace {
// One launch of compact1_kernel over n items (n = the capacity where d_n / d_n64 hold the count): at least one block, so that
// the tail runs whatever n is.
template <int NC, bool CONCAT, class Pred, class Emit, class Tail>
int compact1(slam_ccicp *h, Pred pred, Emit emit, Tail tail, long long n, int limit, hipStream_t st, const int *d_n = nullptr,
             const long long *d_n64 = nullptr, int *d_err = nullptr)
{
    SLAM_REQUIRE(n < (1ll << kStatValueBits), SLAM_E_INVALID, "compaction over %lld items: the look-back words count to 2^28", n);
    const int    n_blocks = (int)std::max<long long>((n + kItems - 1) / kItems, 1);
    const size_t need = sizeof(unsigned long long) * (1 + 2 * (size_t)n_blocks);
    if (need > h->status.cap) {
        SLAM_TRY(h->status.reserve(std::max<size_t>(need, 1 << 16)));
        SLAM_HIP(hipMemsetAsync(h->status.p, 0, h->status.cap, st)); // epoch 0, and no word carries a flag
    }
    const Domain dom = {n, d_n, d_n64};
    hipLaunchKernelGGL((compact1_kernel<NC, CONCAT, Pred, Emit, Tail>), dim3(n_blocks), dim3(kScanThreads), 0, st, pred, emit, tail, dom,
                       static_cast<unsigned long long *>(h->status.p), limit, d_err);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}
// the three-step compaction's interface on the one-launch kernel
template <class Pred, class Emit>
int compact_one(slam_ccicp *h, Pred pred, Emit emit, long long n, int limit, int *d_total, hipStream_t st, const int *d_n = nullptr,
                const long long *d_n64 = nullptr)
{
    return compact1<1, false>(h, OneOutput<Pred>{pred}, OneEmit<Emit>{emit}, NoTail{d_total, nullptr}, n, limit, st, d_n, d_n64);
}
} // namespace

// Several scenes of slam_ccicp_scene_dev as ONE batch for slam_icp_fit_batch_dev (the throughput form of config 3): scene k's
// points behind those of the scenes before it, scan_off / scan_nga as that call reads them.  blockIdx.y = scene.
constexpr int kPackMax = 32;
struct PackArgs {
    const double2 *pts[kPackMax];
    const int32_t *scan[kPackMax]; // {0, n, n_ga} of each scene, on the device
    int            n;
};
__global__ __launch_bounds__(256) void pack_scans_kernel(PackArgs a, double2 *out, int32_t *scan_off, int32_t *scan_nga)
{
    const int k = blockIdx.y;
    int       base = 0;
    for (int j = 0; j < k; ++j) base += a.scan[j][1];
    const int n = a.scan[k][1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scan_off[k] = base;
        scan_nga[k] = a.scan[k][2];
        if (k == a.n - 1) scan_off[a.n] = base + n;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[base + i] = a.pts[k][i];
}

extern "C" {

int slam_ccicp_create(slam_ccicp_t **out)
{
    SLAM_REQUIRE(out, SLAM_E_INVALID, "slam_ccicp_create: null out pointer");
    *out = nullptr;
    SLAM_TRY(require_device());
    slam_ccicp *h = new (std::nothrow) slam_ccicp();
    SLAM_REQUIRE(h, SLAM_E_NOMEM, "slam_ccicp_create: out of host memory");
    int rc = h->small.reserve(256);
    if (rc != SLAM_OK) {
        delete h;
        return rc;
    }
    *out = h;
    return SLAM_OK;
}

void slam_ccicp_destroy(slam_ccicp_t *h) { delete h; }

int slam_ccicp_voxel_downsample_dev(slam_ccicp_t *h, const float *d_xyz, const uint8_t *d_flag, int n, int stride,
                                    float leaf_x, float leaf_y, float leaf_z, float *d_out, int max_out, int *n_out,
                                    slam_stream_t stream)
{
    SLAM_REQUIRE(h && n >= 0 && stride >= 3 && n_out && (d_xyz || n == 0) && (d_out || max_out == 0), SLAM_E_INVALID,
                 "slam_ccicp_voxel_downsample_dev: bad arguments");
    SLAM_REQUIRE(leaf_x > 0 && leaf_y > 0 && leaf_z > 0, SLAM_E_INVALID, "leaf size must be positive");
    *n_out = 0;
    if (n == 0) return SLAM_OK;
    hipStream_t st = as_stream(stream);
    unsigned   *mm = static_cast<unsigned *>(h->small.p);
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    SLAM_HIP(hipMemcpyAsync(mm, init, sizeof init, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(minmax_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_xyz, d_flag, n, stride, mm);
    unsigned got[6];
    SLAM_HIP(hipMemcpyAsync(got, mm, sizeof got, hipMemcpyDeviceToHost, st));
    SLAM_HIP(hipStreamSynchronize(st)); // the lattice extent sizes the accumulator
    if (got[0] == 0xffffffffu) return SLAM_OK; // no finite point
    VoxelGridView g;
    const float   leaf[3] = {leaf_x, leaf_y, leaf_z};
    long long     nv = 1;
    for (int d = 0; d < 3; ++d) {
        g.inv[d] = 1.0f / leaf[d]; // Eigen::Array4f::Ones() / leaf_size_.array()
        const double lo = std::floor((double)(unorder_f32(got[d]) * g.inv[d]));
        const double hi = std::floor((double)(unorder_f32(got[3 + d]) * g.inv[d]));
        SLAM_REQUIRE(std::fabs(lo) < 1e9 && hi - lo + 1.0 <= (double)h->max_voxels, SLAM_E_INVALID,
                     "voxel lattice too large along axis %d: leaf too small for the extent", d);
        g.min_b[d] = (int)lo;
        g.div_b[d] = (int)(hi - lo) + 1;
        nv *= g.div_b[d];
        // PCL: "Leaf size is too small for the input dataset. Integer indices would overflow."
        SLAM_REQUIRE(nv > 0 && nv <= h->max_voxels, SLAM_E_INVALID,
                     "voxel lattice of %lld cells exceeds the accumulator limit (%lld): leaf too small for the extent",
                     nv, h->max_voxels);
    }
    g.n_vox = nv;
    SLAM_TRY(h->vox.reserve(sizeof(Voxel) * (size_t)nv));
    Voxel *vox = static_cast<Voxel *>(h->vox.p);
    SLAM_HIP(hipMemsetAsync(vox, 0, sizeof(Voxel) * (size_t)nv, st));
    h->vox_clean = 0; // (the chain of slam_ccicp_scene_dev shares the accumulator and expects it zero)
    hipLaunchKernelGGL(voxel_accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g, d_xyz, d_flag, n, stride, vox);
    int *d_total = reinterpret_cast<int *>(mm + 8);
    SLAM_TRY(compact_one(h, VoxelUsed{vox}, VoxelEmit{vox, d_out}, nv, max_out, d_total, st));
    int total = 0;
    SLAM_HIP(hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, st));
    SLAM_HIP(hipStreamSynchronize(st));
    *n_out = total;
    SLAM_REQUIRE(total <= max_out, SLAM_E_INVALID, "output holds %d voxels, %d produced (first %d written)", max_out,
                 total, max_out);
    return SLAM_OK;
}

int slam_ccicp_select_dev(slam_ccicp_t *h, const float *d_xyz, int n, int stride, const uint8_t *d_labels,
                          unsigned label_mask, float *d_out_xyz4, int *n_out, slam_stream_t stream)
{
    SLAM_REQUIRE(h && n >= 0 && stride >= 3 && n_out && (n == 0 || (d_xyz && d_labels && d_out_xyz4)), SLAM_E_INVALID,
                 "slam_ccicp_select_dev: bad arguments");
    *n_out = 0;
    if (n == 0) return SLAM_OK;
    hipStream_t st = as_stream(stream);
    int        *d_tot = reinterpret_cast<int *>(static_cast<unsigned *>(h->small.p) + 26);
    SLAM_TRY(compact_one(h, LabelPred{d_labels, label_mask}, Xyz4Emit{d_xyz, stride, reinterpret_cast<float4 *>(d_out_xyz4)}, n, n, d_tot, st));
    SLAM_HIP(hipMemcpyAsync(n_out, d_tot, sizeof(int), hipMemcpyDeviceToHost, st));
    SLAM_HIP(hipStreamSynchronize(st));
    return SLAM_OK;
}

int slam_ccicp_bin_order_dev(slam_ccicp_t *h, const float *d_xyz, const uint8_t *d_flag, int n, int stride,
                             float *d_out_xyzg, int *n_out, slam_stream_t stream)
{
    SLAM_REQUIRE(h && n >= 0 && stride >= 3 && n_out && (n == 0 || (d_xyz && d_flag && d_out_xyzg)), SLAM_E_INVALID,
                 "slam_ccicp_bin_order_dev: bad arguments");
    *n_out = 0;
    if (n == 0) return SLAM_OK;
    hipStream_t st = as_stream(stream);
    SLAM_TRY(h->keys.reserve(2 * sizeof(unsigned long long) * (size_t)n));
    unsigned long long *k_in = static_cast<unsigned long long *>(h->keys.p), *k_out = k_in + n;
    hipLaunchKernelGGL(bin_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_xyz, d_flag, n, stride, k_in);
    size_t tmp = 0;
    SLAM_HIP(rocprim::radix_sort_keys(nullptr, tmp, k_in, k_out, (size_t)n, 0u, 53u, st));
    SLAM_TRY(h->sort_tmp.reserve(tmp));
    SLAM_HIP(rocprim::radix_sort_keys(h->sort_tmp.p, tmp, k_in, k_out, (size_t)n, 0u, 53u, st));
    int *d_n = reinterpret_cast<int *>(static_cast<unsigned *>(h->small.p) + 24);
    SLAM_HIP(hipMemsetAsync(d_n, 0, sizeof(int), st));
    hipLaunchKernelGGL(bin_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_xyz, d_flag, stride, k_out, n,
                       reinterpret_cast<float4 *>(d_out_xyzg), d_n);
    SLAM_HIP(hipGetLastError());
    SLAM_HIP(hipMemcpyAsync(n_out, d_n, sizeof(int), hipMemcpyDeviceToHost, st));
    SLAM_HIP(hipStreamSynchronize(st));
    return SLAM_OK;
}

int slam_ccicp_split_box_dev(slam_ccicp_t *h, const float *d_xyzg, int n, int stride, const float box[4], int cap, double *d_ga_xy,
                             double *d_nga_xy, int counts[2], int totals[2], slam_stream_t stream)
{
    SLAM_REQUIRE(h && n >= 0 && stride >= 4 && cap >= 1 && d_ga_xy && d_nga_xy && counts && (d_xyzg || n == 0),
                 SLAM_E_INVALID, "slam_ccicp_split_box_dev: bad arguments");
    counts[0] = counts[1] = 0;
    if (totals) totals[0] = totals[1] = 0;
    if (n == 0) return SLAM_OK;
    hipStream_t st = as_stream(stream);
    int        *d_tot = reinterpret_cast<int *>(static_cast<unsigned *>(h->small.p) + 12);
    SplitPred   p;
    p.xyzg = d_xyzg;
    p.stride = stride;
    p.crop = box ? 1 : 0;
    p.x_lo = box ? box[0] : 0.f;
    p.x_hi = box ? box[1] : 0.f;
    p.y_lo = box ? box[2] : 0.f;
    p.y_hi = box ? box[3] : 0.f;
    p.want_ga = 1;
    SLAM_TRY(compact_one(h, p, SplitEmit{d_xyzg, stride, d_ga_xy, nullptr, 0}, n, cap - 1, d_tot, st)); // ICP_MAX_PTS-1 (:256,:259)
    p.want_ga = 0;
    SLAM_TRY(compact_one(h, p, SplitEmit{d_xyzg, stride, d_nga_xy, nullptr, 0}, n, cap - 1, d_tot + 1, st));
    int tot[2];
    SLAM_HIP(hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, st));
    SLAM_HIP(hipStreamSynchronize(st));
    counts[0] = tot[0] < cap - 1 ? tot[0] : cap - 1;
    counts[1] = tot[1] < cap - 1 ? tot[1] : cap - 1;
    if (totals) totals[0] = tot[0], totals[1] = tot[1];
    return SLAM_OK;
}

int slam_ccicp_split_dev(slam_ccicp_t *h, const float *d_xyzg, int n, int stride, int crop, double cur_x, double cur_y,
                         double crop_dist, int cap, double *d_ga_xy, double *d_nga_xy, int counts[2],
                         slam_stream_t stream)
{
    // setFilterLimits takes floats (icpTools.cpp:231,236)
    const float box[4] = {(float)(-crop_dist + cur_x), (float)(crop_dist + cur_x), (float)(-crop_dist + cur_y), (float)(crop_dist + cur_y)};
    return slam_ccicp_split_box_dev(h, d_xyzg, n, stride, crop ? box : nullptr, cap, d_ga_xy, d_nga_xy, counts, nullptr, stream);
}

int slam_ccicp_height_dev(slam_ccicp_t *h, const float *d_ground, int n, int stride, const double pose[7], double *z_out,
                          int *n_corr, int nn_idx[4], slam_stream_t stream)
{
    SLAM_REQUIRE(h && pose && z_out && n >= 0 && stride >= 3 && (d_ground || n == 0), SLAM_E_INVALID,
                 "slam_ccicp_height_dev: bad arguments");
    *z_out = pose[2];
    if (n_corr) *n_corr = 0;
    if (nn_idx) nn_idx[0] = nn_idx[1] = nn_idx[2] = nn_idx[3] = -1;
    if (n == 0) return SLAM_OK;
    // tf::Matrix3x3(q) stored to an Eigen::Matrix4f (:321-329), then pcl::transformPointCloud in float (:332)
    float4 q[4];
    wheel_points(pose, q);
    int k = 0;
    hipStream_t         st = as_stream(stream);
    unsigned long long *best = reinterpret_cast<unsigned long long *>(static_cast<unsigned *>(h->small.p) + 16);
    SLAM_HIP(hipMemsetAsync(best, 0xff, 4 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(height_nn_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_ground, n, stride, q[0], q[1], q[2],
                       q[3], best);
    unsigned long long got[4];
    SLAM_HIP(hipMemcpyAsync(got, best, sizeof got, hipMemcpyDeviceToHost, st));
    SLAM_HIP(hipStreamSynchronize(st));
    float corr[4][3];
    int   nc = 0;
    for (k = 0; k < 4; ++k) {
        if (got[k] == ~0ull) continue;
        const unsigned idx = (unsigned)(got[k] & 0xffffffffu), bits = (unsigned)(got[k] >> 32);
        float          dd;
        memcpy(&dd, &bits, 4);
        if (nn_idx) nn_idx[k] = (int)idx;
        if (dd < 9.0f) { // :345
            SLAM_HIP(hipMemcpyAsync(corr[nc], d_ground + (size_t)idx * stride, 3 * sizeof(float), hipMemcpyDeviceToHost, st));
            ++nc;
        }
    }
    SLAM_HIP(hipStreamSynchronize(st));
    if (n_corr) *n_corr = nc;
    *z_out = height_from_neighbours(corr, nc, pose[2]);
    return SLAM_OK;
}

// ---- the device-resident chain
namespace {
struct ChainSmall { // layout of slam_ccicp::chain
    VoxelGridView      g;
    unsigned           mm[8];
    int                n_obs, n_gnd, n_flt, tot[2], err, pad[2];
    float4             q[4];
    unsigned long long best[4];
};

// obstacle cloud (output 0) and ground cloud (output 1) from one pass over the labels
struct ObsGndPred {
    const unsigned char *labels;
    unsigned             obs_mask;
    int                  want_ground;
    __device__ int operator()(long long i) const
    {
        const unsigned l = labels[i];
        return ((obs_mask >> l) & 1u) ? 0 : ((want_ground && l == 1u) ? 1 : -1);
    }
};
struct ObsGndEmit {
    const float *xyz;
    int          stride;
    float4      *obs, *gnd;
    __device__ void operator()(long long i, int c, int pos) const
    {
        const float *q = xyz + (size_t)i * stride;
        (c ? gnd : obs)[pos] = make_float4(q[0], q[1], q[2], 0.f);
    }
};
// ... its tail starts the chain's bookkeeping afresh (what a launch of its own did): the counts it has, the extent and the
// error bits for the kernels behind it
struct ObsGndTail {
    ChainSmall *c;
    __device__ void operator()(const int tot[2]) const
    {
        c->n_obs = tot[0];
        c->n_gnd = tot[1];
        c->n_flt = 0;
        c->err = 0;
        for (int d = 0; d < 3; ++d) c->mm[d] = 0xffffffffu, c->mm[3 + d] = 0u;
    }
};
struct FltTail {
    int *n_flt;
    __device__ void operator()(const int tot[2]) const { *n_flt = tot[0]; }
};
// doICPMatch's marshalling as ONE pass: class of a point that survives the crop (0 = GA, 1 = NGA), GA in front of NGA
struct SplitClass {
    SplitPred p; // (want_ga unused)
    __device__ int operator()(long long i) const
    {
        const float *q = p.xyzg + (size_t)i * p.stride;
        if (p.crop && !(finite3(q) && q[0] >= p.x_lo && q[0] <= p.x_hi && q[1] >= p.y_lo && q[1] <= p.y_hi)) return -1;
        return q[3] > 0.5f ? 0 : 1; // isGA, PointcloudXYZGD.h:28-30
    }
};
struct SplitEmit2 {
    const float *xyzg;
    int          stride;
    double      *out;
    __device__ void operator()(long long i, int, int pos) const
    {
        const float *q = xyzg + (size_t)i * stride;
        out[2 * (size_t)pos] = (double)q[0]; // icpTools.cpp:252, 267: float coordinates widened
        out[2 * (size_t)pos + 1] = (double)q[1];
    }
};
// ... and the scan descriptor {0, n_ga + n_nga, n_ga} of the one scan slam_icp_fit_batch_dev then registers, the class totals
// capped as CCICP::doICPMatch caps them (ICP_MAX_PTS - 1, icpTools.cpp:256,259), and the chain's counts
struct SceneTail {
    ChainSmall *c;
    int         cap;
    int        *scan, *counts;
    __device__ void operator()(const int tot[2]) const
    {
        c->tot[0] = tot[0], c->tot[1] = tot[1];
        const int ga = min(tot[0], cap - 1), nga = min(tot[1], cap - 1);
        scan[0] = 0;
        scan[1] = ga + nga;
        scan[2] = ga;
        counts[0] = c->n_obs;
        counts[1] = c->n_gnd;
        counts[2] = c->n_flt;
        counts[3] = c->err;
    }
};
} // namespace

// Launches of one scene (voxel filter on): bin, INSAC, label | obstacle + ground | mark, flag + extent | accumulate (+ lattice) |
// voxels | split + descriptor = 9 -- round 5: 30, eight of them fills and copies.
int slam_ccicp_scene_dev(slam_ccicp_t *h, slam_gseg_t *seg, const float *d_xyz, int n, int stride, int voxel, int crop,
                         double cur_x, double cur_y, double crop_dist, int cap, double *d_pts, int32_t *d_scan, float *d_ground,
                         int32_t *d_counts, slam_stream_t stream)
{
    SLAM_REQUIRE(h && seg && n >= 0 && stride >= 3 && cap >= 1 && d_pts && d_scan && d_counts && (d_xyz || n == 0), SLAM_E_INVALID,
                 "slam_ccicp_scene_dev: bad arguments");
    hipStream_t st = as_stream(stream);
    const size_t np = (size_t)std::max(n, 1);
    SLAM_TRY(h->labels.reserve(np));
    SLAM_TRY(h->obs.reserve(16 * np));
    SLAM_TRY(h->flags.reserve(np));
    SLAM_TRY(h->filtered.reserve(16 * np));
    SLAM_TRY(h->chain.reserve(sizeof(ChainSmall)));
    ChainSmall    *c = static_cast<ChainSmall *>(h->chain.p);
    unsigned char *lab = static_cast<unsigned char *>(h->labels.p), *flg = static_cast<unsigned char *>(h->flags.p);
    float         *obs = static_cast<float *>(h->obs.p), *flt = static_cast<float *>(h->filtered.p);
    // segmentGround (icpTools.cpp:106-119): the outcloud CCICP classifies and the ground cloud, one pass over the labels
    SLAM_TRY(slam_gseg_segment_dev(seg, d_xyz, n, stride, lab, stream));
    SLAM_TRY((compact1<2, false>(h, ObsGndPred{lab, (1u << 2) | (1u << 3), d_ground ? 1 : 0},
                                 ObsGndEmit{d_xyz, stride, reinterpret_cast<float4 *>(obs), reinterpret_cast<float4 *>(d_ground)}, ObsGndTail{c}, n, n,
                                 st)));
    if (n > 0) {
        // classifyPoints (:36-103) on however many obstacle points there are, and the extent of what it keeps
        SLAM_TRY(slam_gseg_classify_ga_extent_dev(seg, obs, &c->n_obs, n, 4, flg, c->mm, stream));
        if (voxel) { // setSceneCloud's voxel filter (:620-633), leaf 0.5, 0.5, 2
            // the accumulator is zero where no scene has left a sum: the compaction below clears what it reads
            if (sizeof(Voxel) * (size_t)h->chain_voxels > h->vox.cap) h->vox_clean = 0;
            SLAM_TRY(h->vox.reserve(sizeof(Voxel) * (size_t)h->chain_voxels));
            Voxel *vox = static_cast<Voxel *>(h->vox.p);
            if (h->vox_clean < (size_t)h->chain_voxels) {
                SLAM_HIP(hipMemsetAsync(vox, 0, sizeof(Voxel) * (size_t)h->chain_voxels, st));
                h->vox_clean = (size_t)h->chain_voxels;
            }
            hipLaunchKernelGGL(voxel_accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, st, VoxelGridView(), obs, flg, n, 4, vox, &c->g,
                               &c->n_obs, c->mm, make_float3(0.5f, 0.5f, 2.0f), h->chain_voxels, &c->err);
            // (there are never more occupied voxels than points: n bounds the output)
            SLAM_TRY((compact1<1, false>(h, OneOutput<VoxelUsed>{VoxelUsed{vox}}, OneEmit<VoxelEmitClean>{VoxelEmitClean{vox, flt}}, FltTail{&c->n_flt},
                                         h->chain_voxels, n, st, nullptr, &c->g.n_vox, &c->err)));
        } else { // setTargetCloud: classified, bin by bin, no voxel filter (:591-595)
            SLAM_TRY(h->keys.reserve(2 * sizeof(unsigned long long) * (size_t)n));
            unsigned long long *k_in = static_cast<unsigned long long *>(h->keys.p), *k_out = k_in + n;
            hipLaunchKernelGGL(bin_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, st, obs, flg, n, 4, k_in, &c->n_obs);
            size_t tmp = 0;
            SLAM_HIP(rocprim::radix_sort_keys(nullptr, tmp, k_in, k_out, (size_t)n, 0u, 53u, st));
            SLAM_TRY(h->sort_tmp.reserve(tmp));
            SLAM_HIP(rocprim::radix_sort_keys(h->sort_tmp.p, tmp, k_in, k_out, (size_t)n, 0u, 53u, st));
            hipLaunchKernelGGL(bin_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, obs, flg, 4, k_out, n,
                               reinterpret_cast<float4 *>(flt), &c->n_flt);
        }
    }
    // doICPMatch marshalling (:225-276): crop, split by class with the cap, GA in front of NGA -- and the scan's descriptor
    SplitClass p;
    p.p.xyzg = flt;
    p.p.stride = 4;
    p.p.want_ga = 0;
    p.p.crop = crop;
    p.p.x_lo = (float)(-crop_dist + cur_x);
    p.p.x_hi = (float)(crop_dist + cur_x);
    p.p.y_lo = (float)(-crop_dist + cur_y);
    p.p.y_hi = (float)(crop_dist + cur_y);
    SLAM_TRY((compact1<2, true>(h, p, SplitEmit2{flt, 4, d_pts}, SceneTail{c, cap, d_scan, d_counts}, n, cap - 1, st, &c->n_flt, nullptr, &c->err)));
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int slam_ccicp_height_pose_dev(slam_ccicp_t *h, const float *d_ground, const int32_t *d_n_ground, int n_capacity, int stride,
                               const double *d_R, const double *d_t, double z0, double *d_out, slam_stream_t stream)
{
    return slam_ccicp_height_rpy_pose_dev(h, d_ground, d_n_ground, n_capacity, stride, d_R, d_t, z0, 0.0, 0.0, d_out, stream);
}

int slam_ccicp_scene_cloud_dev(slam_ccicp_t *h, float *d_out_xyzg, int capacity, slam_stream_t stream)
{
    SLAM_REQUIRE(h && capacity >= 0 && (d_out_xyzg || capacity == 0), SLAM_E_INVALID, "slam_ccicp_scene_cloud_dev: bad arguments");
    const size_t have = h->filtered.cap / 16, n = std::min((size_t)capacity, have);
    if (n) SLAM_HIP(hipMemcpyAsync(d_out_xyzg, h->filtered.p, 16 * n, hipMemcpyDeviceToDevice, as_stream(stream)));
    return SLAM_OK;
}

int slam_ccicp_height_rpy_pose_dev(slam_ccicp_t *h, const float *d_ground, const int32_t *d_n_ground, int n_capacity, int stride,
                                   const double *d_R, const double *d_t, double z0, double roll, double pitch, double *d_out,
                                   slam_stream_t stream)
{
    return slam_ccicp_height_rpy_pose_mirror_dev(h, d_ground, d_n_ground, n_capacity, stride, d_R, d_t, z0, roll, pitch, d_out, nullptr, nullptr, 0,
                                                 stream);
}

int slam_ccicp_height_rpy_pose_mirror_dev(slam_ccicp_t *h, const float *d_ground, const int32_t *d_n_ground, int n_capacity, int stride,
                                          const double *d_R, const double *d_t, double z0, double roll, double pitch, double *d_out,
                                          void *mirror_dst, const void *mirror_src, size_t mirror_bytes, slam_stream_t stream)
{
    SLAM_REQUIRE(h && d_n_ground && d_R && d_t && d_out && n_capacity >= 0 && stride >= 3 && (d_ground || n_capacity == 0),
                 SLAM_E_INVALID, "slam_ccicp_height_rpy_pose_dev: bad arguments");
    SLAM_REQUIRE(mirror_bytes == 0 || (mirror_dst && mirror_src && mirror_bytes % 8 == 0 && mirror_bytes <= 4096), SLAM_E_INVALID,
                 "slam_ccicp_height_rpy_pose_mirror_dev: the mirrored block is a multiple of 8 bytes, at most 4096");
    hipStream_t st = as_stream(stream);
    SLAM_TRY(h->chain.reserve(sizeof(ChainSmall)));
    ChainSmall *c = static_cast<ChainSmall *>(h->chain.p);
    // (the four packed neighbours start as "none": set once for the block as it is, put back by every fit)
    if (h->best_of != h->chain.p) {
        SLAM_HIP(hipMemsetAsync(c->best, 0xff, sizeof c->best, st));
        h->best_of = h->chain.p;
    }
    if (n_capacity > 0)
        hipLaunchKernelGGL(height_nn_kernel, dim3((n_capacity + 255) / 256), dim3(256), 0, st, d_ground, n_capacity, stride,
                           make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), c->best,
                           nullptr, d_n_ground, HeightPose{d_R, d_t, z0, roll, pitch});
    hipLaunchKernelGGL(height_fit_kernel, dim3(1), dim3(64), 0, st, d_ground, stride, c->best, z0, d_out, true,
                       static_cast<unsigned long long *>(mirror_dst), static_cast<const unsigned long long *>(mirror_src), (int)(mirror_bytes / 8));
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int slam_ccicp_pack_scans_dev(int n, const double *const *d_pts, const int32_t *const *d_scan, double *d_out_pts, int32_t *d_scan_off,
                              int32_t *d_scan_nga, slam_stream_t stream)
{
    SLAM_REQUIRE(n >= 1 && n <= kPackMax && d_pts && d_scan && d_out_pts && d_scan_off && d_scan_nga, SLAM_E_INVALID,
                 "slam_ccicp_pack_scans_dev: bad arguments (1..%d scenes)", kPackMax);
    PackArgs a;
    memset(&a, 0, sizeof a);
    a.n = n;
    for (int k = 0; k < n; ++k) {
        SLAM_REQUIRE(d_pts[k] && d_scan[k], SLAM_E_INVALID, "slam_ccicp_pack_scans_dev: scene %d is null", k);
        a.pts[k] = reinterpret_cast<const double2 *>(d_pts[k]);
        a.scan[k] = d_scan[k];
    }
    hipLaunchKernelGGL(pack_scans_kernel, dim3(16, n), dim3(256), 0, as_stream(stream), a, reinterpret_cast<double2 *>(d_out_pts), d_scan_off,
                       d_scan_nga);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

} // extern "C"
