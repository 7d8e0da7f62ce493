// gseg.hip -- GP-INSAC ground segmentation on gfx950 behind the C-ABI: the
// pre-filter both halves of the hot path run first (ccicp2d/src/icpTools.cpp:114-115,
// mls/src/mls.cpp:66-67); SURVEY.md section 8(f) row 1.
//
// Reference: ground_segmentation/src/groundSegmentation.cpp
//   genPolarBinGrid :110-162   72 sectors x 200 range bins, lowest-z prototype per bin
//   genGPModel      :165-185   squared-exponential covariance
//   sectorINSAC     :196-468   seeds, iterative GP inlier growth, per-point labels
//
// Three kernels.  (1) binning: one thread per point, integer count + one 64-bit
// atomicMin of (orderable z, point index) per bin -- the reference's "first point
// with the smallest z" is exactly that minimum.  (2) one workgroup per sector
// runs the whole INSAC loop: the model Gram matrix is symmetric positive
// definite (kernel + noise*I), so instead of Eigen's dense inverse (:303) it is
// Cholesky-factored once per outer iteration and every candidate needs one
// forward substitution: f = c^T A^-1 z, Vf = sf - |L^-1 c|^2 (values agree with
// the inverse to rounding).  (3) labels: one thread per point.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "common.hpp"

using namespace slam;

namespace {

constexpr int NA = 72;   // groundSegmentation.h:18 NUMBINSA
constexpr int NL = 200;  // groundSegmentation.h:19 NUMBINSL
constexpr float kInvalid = 1000.0f; // groundSegmentation.h:17 INVALID
constexpr int kSecThreads = 256;

struct GsegParams {
    double rmax;
    int    num_seedpoints;
    double p_l, p_sf, p_sn, p_tmodel, p_tdata, p_tg, robot_height, max_seed_range, max_seed_height;
};

__device__ inline unsigned orderable(float z)
{
    const unsigned u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// :110-162.  bin_of[i] = sector*200 + bin, or -1 beyond RMAX
// (adjacent lanes are adjacent azimuth steps of one ring: a 5-degree sector is a run of ~28 lanes in one bin -- the run
// is counted and its minimum taken inside the wavefront, its last lane issues the two atomics)
__global__ __launch_bounds__(256) void gseg_bin_kernel(GsegParams p, const float *xyz, int n, int stride, int *bin_of,
                                                       int *count, unsigned long long *proto)
{
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    int                b = -1;
    int                cnt = 0;
    unsigned long long key = ~0ull; // (orderable z, index): the prototype is the minimum
    if (i < n) {
        const float *q = xyz + (size_t)i * stride;
        const double px = q[0], py = q[1], pz = q[2];
        if (sqrt(px * px + py * py + pz * pz) < p.rmax) { // :126
            const double bsize_rad = 360.0 / NA, bsize_lin = p.rmax / NL;
            double       ph = atan2(py, px) * (180 / M_PI);
            if (ph < 0) ph = 360.0 + ph;
            unsigned bind_rad = (unsigned)floor(ph / bsize_rad);
            if (bind_rad >= (unsigned)NA) bind_rad = NA - 1; // the reference asserts (:136)
            const double xy = sqrt(px * px + py * py);
            unsigned     bind_lin = (unsigned)floor(xy / bsize_lin);
            if (bind_lin >= (unsigned)NL) bind_lin = NL - 1;
            b = (int)bind_rad * NL + (int)bind_lin;
            cnt = 1;                                                       // binPoints.push_back :145
            if (q[2] < kInvalid)                                           // :149 against INVALID; NaN never passes
                key = ((unsigned long long)orderable(q[2]) << 32) | (unsigned)i;
        }
        bin_of[i] = b;
    }
    const int  prev = __shfl_up(b, 1), next = __shfl_down(b, 1);
    bool       head = lane == 0 || prev != b;
    const bool last = lane == 63 || next != b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { // segmented inclusive scan over the runs of equal bins
        const int                uc = __shfl_up(cnt, d);
        const unsigned long long uk = __shfl_up(key, d);
        const bool               uh = __shfl_up((int)head, d) != 0;
        if (lane >= d && !head) {
            cnt += uc;
            key = uk < key ? uk : key;
            head = uh;
        }
    }
    if (last && b >= 0) {
        atomicAdd(&count[b], cnt);
        if (key != ~0ull) atomicMin(&proto[b], key);
    }
}

// :165-185 with sig_f, p_l narrowed to float as the reference's signature does
__device__ inline double gp_cov(double r1, double r2, float sig_f, float p_l)
{
    const float  coeff = (-1 / (2 * p_l * p_l));
    const double diff = r1 - r2;
    return (double)sig_f * exp((double)coeff * (diff * diff));
}

#ifdef SLAM_MEASURE
__device__ long long g_insac_dbg[NA][8]; // ticks of 10 ns per sector: setup, matrix, factorisation, solves, candidates, verdict; [6] rounds
#define INSAC_T(k)                                                          \
    do {                                                                    \
        if (tid == 0) {                                                     \
            const long long now_ = (long long)__builtin_amdgcn_s_memrealtime(); \
            g_insac_dbg[sec][k] += now_ - t_last_;                          \
            t_last_ = now_;                                                 \
        }                                                                   \
    } while (0)
#else
#define INSAC_T(k) do { } while (0)
#endif
// One workgroup per sector: :196-468 up to the per-bin verdict
//   state[b] = 1: bin is in the ground model, value = its prototype height
//   state[b] = 2: bin stayed a candidate, value = GP mean f_s at its range
//   state[b] = 0: bin holds no signal point (its points are dropped)
__global__ __launch_bounds__(kSecThreads) void gseg_insac_kernel(GsegParams p, const float *xyz, int stride,
                                                                 const int *count, const unsigned long long *proto,
                                                                 unsigned char *state, double *value, double *scratch,
                                                                 int *iters_out)
{
    __shared__ double s_range[NL], s_height[NL]; // candidates (sorted), compacted in place
    __shared__ int    s_idx[NL];
    __shared__ double m_range[NL], m_height[NL], alpha[NL], f_s[NL], v_f[NL];
    __shared__ int    m_idx[NL];
    __shared__ double t_range[NL], t_height[NL];
    __shared__ int    t_idx[NL], t_valid[NL];
    __shared__ int    s_ns, s_nm, s_keep, s_sufficient, s_iters;

    const int    sec = blockIdx.x, tid = threadIdx.x;
    const float  sf = (float)p.p_sf, pl = (float)p.p_l;
    double      *Lm = scratch + (size_t)sec * NL * NL; // lower-triangular factor, row-major [i*NL + j] 
#ifdef SLAM_MEASURE
    long long t_last_ = (long long)__builtin_amdgcn_s_memrealtime();
    if (tid < 8) g_insac_dbg[sec][tid] = 0;
#endif

    // ---- signal points :205-219
    for (int b = tid; b < NL; b += kSecThreads) {
        const unsigned long long key = proto[sec * NL + b];
        int                      ok = 0;
        if (key != ~0ull && count[sec * NL + b] > 5) {
            const float *q = xyz + (size_t)(unsigned)(key & 0xffffffffu) * stride;
            const double px = q[0], py = q[1];
            t_range[b] = (double)(float)sqrt(px * px + py * py); // pcl::PointXY stores floats
            t_height[b] = (double)q[2];
            ok = 1;
        }
        t_valid[b] = ok;
        t_idx[b] = b;
    }
    __syncthreads();
    // ---- sort by (height, bin) :229 : the signal bins gathered (a few dozen of the 200), then the rank of each among them by counting
    // (round 6: the counts, the sort over the signal bins only and the seed selection by the whole workgroup -- one thread walking
    // 200 LDS words, every thread ranking against all 200 bins, one thread walking the sorted list: 16 of the kernel's 38 us per sector)
    __shared__ int s_wcnt[kSecThreads / 64];
    __shared__ int v_bin[NL];
    {
        const bool               valid = tid < NL && t_valid[tid];
        const unsigned long long vm = __ballot(valid);
        if ((tid & 63) == 0) s_wcnt[tid >> 6] = __popcll(vm);
        __syncthreads();
        int at = __popcll(vm & ((1ull << (tid & 63)) - 1ull));
        for (int w = 0; w < (tid >> 6); ++w) at += s_wcnt[w];
        if (valid) v_bin[at] = tid;
    }
    const int ns0 = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
    __syncthreads();
    if (tid < ns0) {
        const int    b = v_bin[tid];
        const double hb = t_height[b];
        int          rank = 0;
        for (int o = 0; o < ns0; ++o) {
            const int    ob = v_bin[o];
            const double ho = t_height[ob];
            rank += (ho < hb || (ho == hb && ob < b)) ? 1 : 0;
        }
        s_range[rank] = t_range[b];
        s_height[rank] = t_height[b];
        s_idx[rank] = b;
    }
    __syncthreads();
    // ---- seeds :235-277: the first npt sorted entries that pass the gates; the others, in order, are the candidates
    {
        const int  npt = ns0 < p.num_seedpoints ? ns0 : p.num_seedpoints;
        const bool pass = tid < ns0 && s_range[tid] < p.max_seed_range && fabs(s_height[tid]) < p.max_seed_height;
        const unsigned long long pm = __ballot(pass);
        if ((tid & 63) == 0) s_wcnt[tid >> 6] = __popcll(pm);
        __syncthreads();
        int before = __popcll(pm & ((1ull << (tid & 63)) - 1ull)); // passes in front of this entry
        for (int w = 0; w < (tid >> 6); ++w) before += s_wcnt[w];
        const int  all_pass = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
        const int  nm = all_pass < npt ? all_pass : npt, taken_before = before < npt ? before : npt;
        const bool take = pass && before < npt;
        if (tid < ns0) {
            if (take) {
                m_range[before] = s_range[tid];
                m_height[before] = s_height[tid];
                m_idx[before] = s_idx[tid];
            } else { // (t_* are free since the sort: the candidates are compacted there, then moved back)
                t_range[tid - taken_before] = s_range[tid];
                t_height[tid - taken_before] = s_height[tid];
                t_idx[tid - taken_before] = s_idx[tid];
            }
        }
        __syncthreads();
        const int w = ns0 - nm;
        if (tid < w) {
            s_range[tid] = t_range[tid];
            s_height[tid] = t_height[tid];
            s_idx[tid] = t_idx[tid];
        }
        if (tid == 0) {
            s_ns = w;
            s_nm = nm;
            s_sufficient = nm >= 2; // :272-277
            s_keep = (nm >= 2) && (w > 0); // :289-290
            s_iters = 0;
        }
    }
    __syncthreads();

    INSAC_T(0);
    while (s_keep) { // :295-377
        const int nm = s_nm, ns = s_ns;
        double *Yg = scratch + (size_t)NA * NL * NL + (size_t)sec * NL * NL; // [candidate][nm]: the candidates' substitution rows
        // (round 6: the factor and these rows in LDS -- 90 KB -- changed nothing: the rounds are chains of f64 roots and divisions)
        auto L = [&](int i, int j) -> double & { return Lm[i * NL + j]; };
        auto Yk = [&](int k, int i) -> double & { return Yg[(size_t)k * NL + i]; };
        // A = C_XX + sn*I (lower triangle), then Cholesky in place, one column per step
        for (int e = tid; e < nm * nm; e += kSecThreads) {
            const int i = e / nm, j = e % nm;
            if (j <= i) L(i, j) = gp_cov(m_range[i], m_range[j], sf, pl) + (i == j ? p.p_sn : 0.0);
        }
        __threadfence_block();
        __syncthreads();
        INSAC_T(1);
        for (int c = 0; c < nm; ++c) {
            // (every thread takes the root of the pivot for itself -- the same value -- and thread 0 stores it: a barrier less per column)
            const double d = sqrt(L(c, c));
            for (int i = c + 1 + tid; i < nm; i += kSecThreads) L(i, c) /= d;
            __threadfence_block();
            __syncthreads();
            if (tid == 0) L(c, c) = d;
            // trailing update of the lower triangle
            const int rem = nm - c - 1;
            for (int e = tid; e < rem * rem; e += kSecThreads) {
                const int i = c + 1 + e / rem, j = c + 1 + e % rem;
                if (j <= i) L(i, j) -= L(i, c) * L(j, c);
            }
            __threadfence_block();
            __syncthreads();
        }
        INSAC_T(2);
        // alpha = A^-1 z: two triangular solves of length nm.  Up to 64 model bins (a sector has a few dozen) by the first wavefront,
        // a row per lane, column by column: lane j's value is final at step j, goes to the others through a scalar register, and
        // every later row takes its term off -- forward the terms arrive in the order one thread would subtract them (the same bits),
        // backward in the opposite order (a last-bit difference in alpha; the labels of every test cloud are the oracle's, whose
        // solve is an LU anyway).  One thread walking both triangles was 7 of the kernel's 38 us per sector.
        if (nm <= 64) {
            if (tid < 64) {
                const int    lane = tid;
                const double dii = lane < nm ? L(lane, lane) : 1.0;
                double       sv = lane < nm ? m_height[lane] : 0.0, yv = 0.0;
                for (int j = 0; j < nm; ++j) {
                    const double mine = sv / dii; // (every lane divides; lane j's quotient is the one that counts)
                    const double aj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), j), __builtin_amdgcn_readlane(__double2loint(mine), j));
                    if (lane == j) yv = aj;
                    if (lane > j && lane < nm) sv -= L(lane, j) * aj;
                }
                sv = yv;
                double xv = 0.0;
                for (int j = nm - 1; j >= 0; --j) {
                    const double mine = sv / dii;
                    const double xj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), j), __builtin_amdgcn_readlane(__double2loint(mine), j));
                    if (lane == j) xv = xj;
                    if (lane < j) sv -= L(j, lane) * xj;
                }
                if (lane < nm) alpha[lane] = xv;
            }
        } else if (tid == 0) {
            for (int i = 0; i < nm; ++i) {
                double s = m_height[i];
                for (int j = 0; j < i; ++j) s -= L(i, j) * alpha[j];
                alpha[i] = s / L(i, i);
            }
            for (int i = nm - 1; i >= 0; --i) {
                double s = alpha[i];
                for (int j = i + 1; j < nm; ++j) s -= L(j, i) * alpha[j];
                alpha[i] = s / L(i, i);
            }
        }
        __syncthreads();
        INSAC_T(3);
        // every candidate: f = c . alpha ; Vf = sf - |L^-1 c|^2   (one thread per candidate, own scratch row)
        for (int k = tid; k < ns; k += kSecThreads) {
            double  f = 0.0, q = 0.0;
            for (int i = 0; i < nm; ++i) {
                const double c = gp_cov(s_range[k], m_range[i], sf, pl);
                f += c * alpha[i];
                double s = c;
                for (int j = 0; j < i; ++j) s -= L(i, j) * Yk(k, j);
                s /= L(i, i);
                Yk(k, i) = s;
                q += s * s;
            }
            f_s[k] = f;
            v_f[k] = gp_cov(s_range[k], s_range[k], sf, pl) - q;
        }
        __syncthreads();
        INSAC_T(4);
        // :331-369: all candidates are judged against THIS iteration's model; inliers join in order -- every candidate by its own
        // thread, their places by counting (one thread walking the list was 3 of the kernel's 38 us per sector)
        {
            bool inl = false;
            if (tid < ns) {
                const double met = (s_height[tid] - f_s[tid]) / sqrt(p.p_sn + v_f[tid] * v_f[tid]);
                inl = v_f[tid] < p.p_tmodel && fabs(met) < p.p_tdata;
            }
            const unsigned long long im = __ballot(inl);
            if ((tid & 63) == 0) s_wcnt[tid >> 6] = __popcll(im);
            __syncthreads();
            int before = __popcll(im & ((1ull << (tid & 63)) - 1ull));
            for (int w = 0; w < (tid >> 6); ++w) before += s_wcnt[w];
            const int n_inl = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
            if (tid < ns) {
                if (inl) {
                    m_range[nm + before] = s_range[tid];
                    m_height[nm + before] = s_height[tid];
                    m_idx[nm + before] = s_idx[tid];
                } else { // (compacted through the free t_* arrays)
                    t_range[tid - before] = s_range[tid];
                    t_height[tid - before] = s_height[tid];
                    t_idx[tid - before] = s_idx[tid];
                    alpha[tid] = f_s[tid]; // (alpha is free until the next round's solve: f_s's values on their way to their new places)
                }
            }
            __syncthreads();
            const int w = ns - n_inl;
            if (tid < ns && !inl) f_s[tid - before] = alpha[tid];
            if (tid < w) {
                s_range[tid] = t_range[tid];
                s_height[tid] = t_height[tid];
                s_idx[tid] = t_idx[tid];
            }
            if (tid == 0) {
                s_keep = !(n_inl == 0 || w == 0); // :374-375
                s_nm = nm + n_inl;
                s_ns = w;
                ++s_iters;
            }
        }
        __syncthreads();
        INSAC_T(5);
#ifdef SLAM_MEASURE
        if (tid == 0) g_insac_dbg[sec][6] += 1;
#endif
    }

    // ---- verdict per bin :385-454
    for (int b = tid; b < NL; b += kSecThreads) state[sec * NL + b] = 0;
    __syncthreads();
    for (int i = tid; i < s_nm; i += kSecThreads) {
        state[sec * NL + m_idx[i]] = 1;
        value[sec * NL + m_idx[i]] = m_height[i];
    }
    if (s_sufficient)
        for (int i = tid; i < s_ns; i += kSecThreads) {
            state[sec * NL + s_idx[i]] = 2;
            value[sec * NL + s_idx[i]] = f_s[i];
        }
    if (tid == 0 && iters_out) iters_out[sec] = s_iters;
}

__global__ __launch_bounds__(256) void gseg_label_kernel(GsegParams p, const float *xyz, int n, int stride,
                                                         const int *bin_of, const unsigned char *state,
                                                         const double *value, unsigned char *labels, int *count,
                                                         unsigned long long *proto)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    // the bins' counts and prototypes have done their work (the INSAC kernel has read them): cleared here for the next cloud,
    // instead of two fills in front of every segmentation (round 6: a match is bound by its launches)
    if (i < NA * NL) {
        count[i] = 0;
        proto[i] = ~0ull;
    }
    if (i >= n) return;
    unsigned char lab = 0; // dropped
    const int     b = bin_of[i];
    if (b >= 0) {
        const unsigned char st = state[b];
        if (st) {
            const double z = (double)xyz[(size_t)i * stride + 2];
            const float  h = (float)fabs(st == 1 ? value[b] - z : z - value[b]); // :397, :437
            if (st == 1 && h < p.p_tg)
                lab = 1; // ground
            else
                lab = h > p.robot_height ? 3 : 2; // :406-413, :439-446
        }
    }
    labels[i] = lab;
}

// ground and drivability-blocking obstacle points as 4-float records (x, y, z, 0), the
// two clouds mls.cpp:73-142 consumes; wave-aggregated append (output order is not input order)
__global__ __launch_bounds__(256) void gseg_split_kernel(const float *xyz, int n, int stride,
                                                         const unsigned char *labels, float4 *ground, float4 *obstacle,
                                                         int *counts)
{
    const int  i = blockIdx.x * 256 + threadIdx.x;
    const int  lab = i < n ? labels[i] : 0;
    const int  lane = threadIdx.x & 63;
    for (int which = 1; which <= 2; ++which) {
        const bool               mine = lab == which;
        const unsigned long long m = __ballot(mine);
        if (!m) continue;
        int base = 0;
        if (lane == (__ffsll((long long)m) - 1)) base = atomicAdd(&counts[which - 1], __popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1);
        if (mine) {
            const float *q = xyz + (size_t)i * stride;
            (which == 1 ? ground : obstacle)[base + __popcll(m & ((1ull << lane) - 1ull))] =
                make_float4(q[0], q[1], q[2], 0.f);
        }
    }
}

// CCICP::classifyPoints, ccicp2d/src/icpTools.cpp:36-103 (icpTools.h:24-26): occupancy of a
// 1200 x 1200 lattice of 0.5 m cells, then per point the number of empty cells among the 8
// around its own: ground adjacent (GA) when >= 2.  flags: 1 GA, 0 NGA, 255 dropped.
constexpr int kGaBins = 1200;

__device__ inline int ga_bin(const float *q)
{
    const double RES = 0.5, offset = (double)kGaBins * RES / 2;
    const double fx = floor(((double)q[0] + offset) / RES), fy = floor(((double)q[1] + offset) / RES); // :57-58
    if (!(fx >= 0 && fx < kGaBins && fy >= 0 && fy < kGaBins)) return -1;                             // :60
    return (int)fx * kGaBins + (int)fy;
}

// (d_n: the number of points where only the device knows it; n is then the capacity the launch was sized for)
// A cell is occupied when it holds the call's EPOCH: the lattice is cleared when it is made, never again (a fill of its own in
// front of every classification was a launch and 1.44 MB).  The epoch lives on the DEVICE -- state[0], read by both kernels of a
// call, moved on by the workgroup of the second kernel that finishes last (state[1] counts them) -- so that a hipGraph replay of
// the two launches is another call like any other (as a counter of the handle passed by value it froze in the replay, and last
// replay's cells counted as this one's: round 6).  32 bits: four billion classifications.
__global__ __launch_bounds__(256) void ga_mark_kernel(const float *xyz, int n, int stride, unsigned *occ, const int *d_n,
                                                      const unsigned *state)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (d_n) n = min(n, *d_n);
    if (i >= n) return;
    const int b = ga_bin(xyz + (size_t)i * stride);
    if (b >= 0) occ[b] = state[0];
}

__device__ inline unsigned ga_order_f32(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ... and, where the caller goes on to the voxel filter (slam_ccicp_scene_dev), the extent of the points the classification
// keeps -- getMinMax3D over the finite points whose flag is not 255, as ccicp.hip's minmax_kernel forms it -- in the same
// pass: mm[6] ordered-float minima and maxima (nullable).
__device__ inline void ga_call_done(unsigned *state, unsigned epoch, int n)
{
    // (the launch is sized for the cloud's capacity, n is what the device says there is: only the workgroups with points take a
    // ticket -- same-address atomics are served one after the other, 11 ns each -- and a call without points marked nothing)
    const unsigned active = (unsigned)((n + 255) / 256);
    if (blockIdx.x >= active) return;
    __syncthreads(); // (every lane of the workgroup has read the lattice)
    if (threadIdx.x == 0 && atomicAdd(&state[1], 1u) == active - 1u) {
        state[1] = 0u;
        state[0] = epoch + 1u == 0u ? 1u : epoch + 1u;
    }
}
__global__ __launch_bounds__(256) void ga_flag_kernel(const float *xyz, int n, int stride, const unsigned *occ,
                                                      unsigned char *flags, const int *d_n, unsigned *state, unsigned *mm)
{
    const int      i = blockIdx.x * 256 + threadIdx.x;
    const unsigned epoch = state[0]; // (moved on only when every workgroup of this launch is through)
    if (d_n) n = min(n, *d_n);
    unsigned char f = 255;
    unsigned      lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    if (i < n) {
        const float *p = xyz + (size_t)i * stride;
        const int    b = ga_bin(p);
        if (b >= 0) {
            const int bi = b / kGaBins, bj = b % kGaBins;
            if (!(bi == 0 || bi == kGaBins - 1 || bj == 0 || bj == kGaBins - 1)) { // :72-77
                int ground = 0;
                for (int q = bi - 1; q <= bi + 1; ++q)
                    for (int r = bj - 1; r <= bj + 1; ++r)
                        if (!(q == bi && r == bj) && occ[q * kGaBins + r] != epoch) ++ground;
                f = ground >= 2; // :96 GRD_ADJ_THRESH
            }
        }
        flags[i] = f;
        if (mm && f != 255 && isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))
            for (int d = 0; d < 3; ++d) lo[d] = hi[d] = ga_order_f32(p[d]);
    }
    if (!mm) { // (uniform)
        ga_call_done(state, epoch, n);
        return;
    }
    __shared__ unsigned red[4][6];
    for (int d = 0; d < 3; ++d) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[d] = min(lo[d], (unsigned)__shfl_xor((int)lo[d], off));
            hi[d] = max(hi[d], (unsigned)__shfl_xor((int)hi[d], off));
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][d] = lo[d], red[threadIdx.x >> 6][3 + d] = hi[d];
    }
    __syncthreads();
    if (threadIdx.x < 6) { // one atomic per block and bound, where it would change the value (ccicp.hip, minmax_kernel)
        const int d = threadIdx.x;
        unsigned  v = red[0][d];
        for (int w = 1; w < 4; ++w) v = d < 3 ? min(v, red[w][d]) : max(v, red[w][d]);
        if (d < 3) {
            if (v != 0xffffffffu && v < __hip_atomic_load(&mm[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mm[d], v);
        } else {
            if (v != 0u && v > __hip_atomic_load(&mm[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&mm[d], v);
        }
    }
    ga_call_done(state, epoch, n);
}

} // namespace

struct slam_gseg {
    GsegParams prm;
    int       *d_count = nullptr;
    unsigned long long *d_proto = nullptr;
    unsigned char *d_state = nullptr;
    double    *d_value = nullptr;
    double    *d_scratch = nullptr; // [2][72][200*200] Cholesky factor + forward-substitution rows
    int       *d_iters = nullptr;
    int       *d_bin_of = nullptr;
    size_t     cap_points = 0;
    unsigned  *d_ga_occ = nullptr; // 1200 x 1200 occupancy of classifyPoints: the epoch of the last call that marked the cell; behind it
                                   // the calls' state {epoch, workgroups through}, kept by the kernels
    void      *d_stage = nullptr; // host-API staging: points + labels
    size_t     cap_stage = 0;
};

static int gseg_reserve(slam_gseg *h, size_t n)
{
    if (n <= h->cap_points) return SLAM_OK;
    const size_t want = n + n / 4; // (clouds of a sequence differ by a few per cent: grow rarely)
    if (h->d_bin_of) (void)hipFree(h->d_bin_of);
    h->d_bin_of = nullptr;
    h->cap_points = 0;
    SLAM_HIP(hipMalloc((void **)&h->d_bin_of, sizeof(int) * want));
    h->cap_points = want;
    return SLAM_OK;
}

extern "C" {

void slam_gseg_default_params(slam_gseg_params *p)
{ // groundSegmentation.cpp:31-55
    if (!p) return;
    p->rmax = 100.0;
    p->num_seedpoints = 10;
    p->gp_lengthparameter = 10;
    p->gp_covariancescale = 1.0;
    p->gp_modelnoise = 0.3;
    p->gp_groundmodelconfidence = 5.0;
    p->gp_grounddataconfidence = 5.0;
    p->gp_groundthreshold = 0.3;
    p->robotheight = 1.2;
    p->seeding_maxrange = 50;
    p->seeding_maxheight = 15;
}

int slam_gseg_create(const slam_gseg_params *params, slam_gseg_t **out)
{
    SLAM_REQUIRE(out, SLAM_E_INVALID, "slam_gseg_create: null out pointer");
    *out = nullptr;
    SLAM_TRY(require_device());
    slam_gseg_params p;
    if (params)
        p = *params;
    else
        slam_gseg_default_params(&p);
    SLAM_REQUIRE(p.rmax > 0 && p.num_seedpoints >= 0 && p.num_seedpoints <= NL, SLAM_E_INVALID,
                 "slam_gseg_create: bad parameters");
    slam_gseg *h = new (std::nothrow) slam_gseg();
    SLAM_REQUIRE(h, SLAM_E_NOMEM, "slam_gseg_create: out of host memory");
    h->prm = {p.rmax, p.num_seedpoints, p.gp_lengthparameter, p.gp_covariancescale, p.gp_modelnoise,
              p.gp_groundmodelconfidence, p.gp_grounddataconfidence, p.gp_groundthreshold, p.robotheight,
              p.seeding_maxrange, p.seeding_maxheight};
    hipError_t e = hipMalloc((void **)&h->d_count, sizeof(int) * NA * NL);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_proto, sizeof(unsigned long long) * NA * NL);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_state, NA * NL);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_value, sizeof(double) * NA * NL);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_scratch, sizeof(double) * 2 * (size_t)NA * NL * NL);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_iters, sizeof(int) * NA);
    if (e == hipSuccess) e = hipMemset(h->d_count, 0, sizeof(int) * NA * NL);
    if (e == hipSuccess) e = hipMemset(h->d_proto, 0xff, sizeof(unsigned long long) * NA * NL);
    if (e != hipSuccess) {
        slam_gseg_destroy(h);
        SLAM_HIP(e);
    }
    *out = h;
    return SLAM_OK;
}

void slam_gseg_destroy(slam_gseg_t *h)
{
    if (!h) return;
    void *ptrs[] = {h->d_count, h->d_proto, h->d_state, h->d_value, h->d_scratch, h->d_iters, h->d_bin_of, h->d_stage,
                    h->d_ga_occ};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete h;
}

int slam_gseg_reserve(slam_gseg_t *h, int max_points)
{
    SLAM_REQUIRE(h && max_points >= 0, SLAM_E_INVALID, "slam_gseg_reserve: bad arguments");
    return gseg_reserve(h, (size_t)max_points);
}

int slam_gseg_segment_dev(slam_gseg_t *h, const float *d_xyz, int n, int stride, uint8_t *d_labels,
                          slam_stream_t stream)
{
    SLAM_REQUIRE(h && n >= 0 && stride >= 3 && (n == 0 || (d_xyz && d_labels)), SLAM_E_INVALID,
                 "slam_gseg_segment_dev: bad arguments");
    SLAM_TRY(require_device());
    SLAM_TRY(gseg_reserve(h, (size_t)(n > 0 ? n : 1)));
    hipStream_t st = as_stream(stream);
    // (the bins' counts and prototypes are clean: slam_gseg_create cleared them, every labelling since has cleared them again)
    if (n > 0)
        hipLaunchKernelGGL(gseg_bin_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h->prm, d_xyz, n, stride,
                           h->d_bin_of, h->d_count, h->d_proto);
    hipLaunchKernelGGL(gseg_insac_kernel, dim3(NA), dim3(kSecThreads), 0, st, h->prm, d_xyz, stride, h->d_count,
                       h->d_proto, h->d_state, h->d_value, h->d_scratch, h->d_iters);
    if (n > 0)
        hipLaunchKernelGGL(gseg_label_kernel, dim3((std::max(n, NA * NL) + 255) / 256), dim3(256), 0, st, h->prm, d_xyz, n, stride,
                           h->d_bin_of, h->d_state, h->d_value, d_labels, h->d_count, h->d_proto);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

int slam_gseg_segment(slam_gseg_t *h, const float *xyz, int n, int stride, uint8_t *labels)
{
    SLAM_REQUIRE(h && n >= 0 && stride >= 3 && (n == 0 || (xyz && labels)), SLAM_E_INVALID,
                 "slam_gseg_segment: bad arguments");
    SLAM_TRY(require_device());
    if (n == 0) return SLAM_OK;
    const size_t bytes = sizeof(float) * (size_t)n * stride;
    if (bytes + (size_t)n > h->cap_stage) {
        if (h->d_stage) (void)hipFree(h->d_stage);
        h->d_stage = nullptr;
        h->cap_stage = 0;
        const size_t want = bytes + (size_t)n + (bytes + (size_t)n) / 4;
        SLAM_HIP(hipMalloc(&h->d_stage, want));
        h->cap_stage = want;
    }
    float   *d_xyz = static_cast<float *>(h->d_stage);
    uint8_t *d_lab = reinterpret_cast<uint8_t *>(h->d_stage) + bytes;
    SLAM_HIP(hipMemcpyAsync(d_xyz, xyz, bytes, hipMemcpyHostToDevice, nullptr));
    SLAM_TRY(slam_gseg_segment_dev(h, d_xyz, n, stride, d_lab, nullptr));
    SLAM_HIP(hipMemcpyAsync(labels, d_lab, (size_t)n, hipMemcpyDeviceToHost, nullptr));
    SLAM_HIP(hipStreamSynchronize(nullptr));
    return SLAM_OK;
}

int slam_gseg_split_dev(slam_gseg_t *h, const float *d_xyz, int n, int stride, const uint8_t *d_labels,
                        float *d_ground_xyz4, float *d_obstacle_xyz4, int32_t *d_counts, slam_stream_t stream)
{
    SLAM_REQUIRE(h && n >= 0 && stride >= 3 && d_counts && (n == 0 || (d_xyz && d_labels && d_ground_xyz4 && d_obstacle_xyz4)),
                 SLAM_E_INVALID, "slam_gseg_split_dev: bad arguments");
    hipStream_t st = as_stream(stream);
    SLAM_HIP(hipMemsetAsync(d_counts, 0, 2 * sizeof(int32_t), st));
    if (n > 0)
        hipLaunchKernelGGL(gseg_split_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_xyz, n, stride, d_labels,
                           reinterpret_cast<float4 *>(d_ground_xyz4), reinterpret_cast<float4 *>(d_obstacle_xyz4),
                           d_counts);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

static int classify_ga(slam_gseg_t *h, const float *d_obstacle_xyz, int n, const int32_t *d_n, int stride, uint8_t *d_flags,
                       slam_stream_t stream, unsigned *d_mm = nullptr)
{
    SLAM_TRY(require_device());
    if (n == 0) return SLAM_OK;
    hipStream_t st = as_stream(stream);
    constexpr size_t kCells = (size_t)kGaBins * kGaBins;
    if (!h->d_ga_occ) {
        SLAM_HIP(hipMalloc((void **)&h->d_ga_occ, sizeof(unsigned) * (kCells + 2)));
        SLAM_HIP(hipMemsetAsync(h->d_ga_occ, 0, sizeof(unsigned) * (kCells + 2), st)); // no cell holds an epoch, no workgroup is through
        SLAM_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_ga_occ + kCells), 1, 1, st)); // the first call's epoch
    }
    unsigned *state = h->d_ga_occ + kCells;
    hipLaunchKernelGGL(ga_mark_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_obstacle_xyz, n, stride, h->d_ga_occ, d_n, state);
    hipLaunchKernelGGL(ga_flag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_obstacle_xyz, n, stride, h->d_ga_occ,
                       d_flags, d_n, state, d_mm);
    SLAM_HIP(hipGetLastError());
    return SLAM_OK;
}

// ccicp.hip's chain: classification + the extent of what it keeps, in the classification's two launches
int slam_gseg_classify_ga_extent_dev(slam_gseg_t *h, const float *d_obstacle_xyz, const int32_t *d_n, int n_capacity, int stride,
                                     uint8_t *d_flags, uint32_t *d_mm, slam_stream_t stream)
{
    SLAM_REQUIRE(h && d_n && n_capacity >= 0 && stride >= 3 && d_mm && (n_capacity == 0 || (d_obstacle_xyz && d_flags)), SLAM_E_INVALID,
                 "slam_gseg_classify_ga_extent_dev: bad arguments");
    return classify_ga(h, d_obstacle_xyz, n_capacity, d_n, stride, d_flags, stream, d_mm);
}

int slam_gseg_classify_ga_dev(slam_gseg_t *h, const float *d_obstacle_xyz, int n, int stride, uint8_t *d_flags,
                              slam_stream_t stream)
{
    SLAM_REQUIRE(h && n >= 0 && stride >= 2 && (n == 0 || (d_obstacle_xyz && d_flags)), SLAM_E_INVALID,
                 "slam_gseg_classify_ga_dev: bad arguments");
    return classify_ga(h, d_obstacle_xyz, n, nullptr, stride, d_flags, stream);
}

int slam_gseg_classify_ga_counted_dev(slam_gseg_t *h, const float *d_obstacle_xyz, const int32_t *d_n, int n_capacity, int stride,
                                      uint8_t *d_flags, slam_stream_t stream)
{
    SLAM_REQUIRE(h && d_n && n_capacity >= 0 && stride >= 2 && (n_capacity == 0 || (d_obstacle_xyz && d_flags)), SLAM_E_INVALID,
                 "slam_gseg_classify_ga_counted_dev: bad arguments");
    return classify_ga(h, d_obstacle_xyz, n_capacity, d_n, stride, d_flags, stream);
}

#ifdef SLAM_MEASURE
// measurement build: mean microseconds per sector of the last segmentation's INSAC kernel in {setup, matrix, factorisation, solves,
// candidates, verdict}, [6] = mean rounds, [7] = the slowest sector's total
int slam_gseg_debug_insac(double out[8])
{
    long long h[NA][8];
    SLAM_HIP(hipDeviceSynchronize());
    SLAM_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_insac_dbg), sizeof h));
    double worst = 0;
    for (int k = 0; k < 8; ++k) out[k] = 0;
    for (int s = 0; s < NA; ++s) {
        double tot = 0;
        for (int k = 0; k < 6; ++k) out[k] += h[s][k] * 0.01 / NA, tot += h[s][k] * 0.01;
        out[6] += (double)h[s][6] / NA;
        worst = tot > worst ? tot : worst;
    }
    out[7] = worst;
    return SLAM_OK;
}
#endif

int slam_gseg_read_model(slam_gseg_t *h, uint8_t *bin_state, double *bin_value, int32_t *sector_iterations)
{
    SLAM_REQUIRE(h, SLAM_E_INVALID, "null handle");
    SLAM_TRY(require_device());
    SLAM_HIP(hipDeviceSynchronize());
    if (bin_state) SLAM_HIP(hipMemcpy(bin_state, h->d_state, NA * NL, hipMemcpyDeviceToHost));
    if (bin_value) SLAM_HIP(hipMemcpy(bin_value, h->d_value, sizeof(double) * NA * NL, hipMemcpyDeviceToHost));
    if (sector_iterations) SLAM_HIP(hipMemcpy(sector_iterations, h->d_iters, sizeof(int) * NA, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

} // extern "C"
