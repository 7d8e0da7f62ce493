/* oracle/ccicp_oracle.c -- CPU restatement of the CCICP facade steps either side of the ICP
 * (SURVEY 8(f) rows 2 and 4).  TEST INFRASTRUCTURE ONLY (see slam_oracle.h).
 *
 * PARITY UNPINNED: these steps call PCL (pcl::PassThrough, pcl::VoxelGrid, pcl::KdTreeFLANN,
 * pcl::NormalEstimation, pcl::transformPointCloud) and tf, none of which is in /root/reference or in
 * this image; ccicp2d/package.xml names the packages without versions (ROS Hydro/Indigo era, PCL 1.7).
 * What follows restates their published algorithms at the reference's call sites:
 *   crop        icpTools.cpp:225-239   PassThrough on x then y, float limits, closed interval
 *   voxel grid  icpTools.cpp:620-633   VoxelGrid leaf (0.5,0.5,2), every field averaged per voxel,
 *                                      output in increasing voxel index (x fastest, then y, then z)
 *   split       icpTools.cpp:248-276   isGA(ground_adj) (PointcloudXYZGD.h:28-30), ICP_MAX_PTS-1 per class
 *   height      icpTools.cpp:301-381   four wheel points, nearest ground point within 3 m, plane normal
 * Where PCL's own arithmetic is order-dependent float (voxel sums follow std::sort's order; the
 * covariance of computePointNormal is a single-pass float formula) this file uses double, and the tests
 * compare with a tolerance instead of bitwise. */
#define _DEFAULT_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "slam_oracle.h"

/* pcl::PassThrough::applyFilterIndices: non-finite points go; a point stays iff min <= v <= max */
int occicp_crop(const float *xyz, int n, int stride, double cur_x, double cur_y, double crop, unsigned char *keep)
{
    const float x_lo = (float)(-crop + cur_x), x_hi = (float)(crop + cur_x); /* setFilterLimits takes floats */
    const float y_lo = (float)(-crop + cur_y), y_hi = (float)(crop + cur_y);
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = xyz + (size_t)i * stride;
        const int ok = isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]) && p[0] >= x_lo && p[0] <= x_hi &&
                       p[1] >= y_lo && p[1] <= y_hi;
        keep[i] = (unsigned char)ok;
        kept += ok;
    }
    return kept;
}

typedef struct {
    long long idx;
    int       pt;
} vox_ref;

static int vox_cmp(const void *a, const void *b)
{
    const vox_ref *x = (const vox_ref *)a, *y = (const vox_ref *)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt - y->pt;
}

/* pcl::VoxelGrid<PointXYZGD>::applyFilter (voxel_grid.hpp): in[i] = x,y,z,ground_adj (as float) with
 * `stride` floats per point; out = 4 floats per voxel (centroid x,y,z and the ground_adj average cast to
 * uint16 as PCL's field copy does).  Returns the number of voxels, or -1 when the lattice overflows an int
 * (PCL then warns and passes the input through). */
int ovoxel_downsample(const float *in, int n, int stride, float lx, float ly, float lz, float *out)
{
    const float inv[3] = {1.0f / lx, 1.0f / ly, 1.0f / lz};
    float       mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int         n_fin = 0;
    for (int i = 0; i < n; ++i) { /* getMinMax3D over finite points */
        const float *p = in + (size_t)i * stride;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (int d = 0; d < 3; ++d) {
            mn[d] = fminf(mn[d], p[d]);
            mx[d] = fmaxf(mx[d], p[d]);
        }
        ++n_fin;
    }
    if (n_fin == 0) return 0;
    long long min_b[3], div_b[3];
    for (int d = 0; d < 3; ++d) {
        min_b[d] = (long long)floorf(mn[d] * inv[d]);
        div_b[d] = (long long)floorf(mx[d] * inv[d]) - min_b[d] + 1;
    }
    if ((double)div_b[0] * (double)div_b[1] * (double)div_b[2] > 2147483647.0) return -1;
    vox_ref *ref = (vox_ref *)malloc(sizeof(vox_ref) * (size_t)n_fin);
    int      m = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = in + (size_t)i * stride;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        long long ijk[3];
        for (int d = 0; d < 3; ++d) ijk[d] = (long long)(floorf(p[d] * inv[d]) - (float)min_b[d]);
        ref[m].idx = ijk[0] + ijk[1] * div_b[0] + ijk[2] * div_b[0] * div_b[1];
        ref[m].pt = i;
        ++m;
    }
    qsort(ref, (size_t)m, sizeof(vox_ref), vox_cmp);
    int n_out = 0;
    for (int a = 0; a < m;) {
        int b = a;
        double s[4] = {0, 0, 0, 0};
        while (b < m && ref[b].idx == ref[a].idx) {
            const float *p = in + (size_t)ref[b].pt * stride;
            for (int d = 0; d < 4; ++d) s[d] += (double)p[d];
            ++b;
        }
        const double c = (double)(b - a);
        for (int d = 0; d < 3; ++d) out[4 * (size_t)n_out + d] = (float)(s[d] / c);
        out[4 * (size_t)n_out + 3] = (float)(uint16_t)(float)(s[3] / c); /* float average stored to a uint16 field */
        ++n_out;
        a = b;
    }
    free(ref);
    return n_out;
}

/* icpTools.cpp:248-276: cloud order, isGA(ground_adj) = ground_adj > 0.5, at most cap-1 points per class */
void occicp_split(const float *xyzg, const unsigned char *keep, int n, int stride, int cap, double *ga, int *n_ga,
                  double *nga, int *n_nga)
{
    int a = 0, b = 0;
    for (int i = 0; i < n; ++i) {
        if (keep && !keep[i]) continue;
        const float *p = xyzg + (size_t)i * stride;
        if (p[3] > 0.5f) {
            if (a >= cap - 1) continue;
            ga[2 * a] = (double)p[0];
            ga[2 * a + 1] = (double)p[1];
            ++a;
        } else {
            if (b >= cap - 1) continue;
            nga[2 * b] = (double)p[0];
            nga[2 * b + 1] = (double)p[1];
            ++b;
        }
    }
    *n_ga = a;
    *n_nga = b;
}

/* smallest-eigenvalue eigenvector of a symmetric 3x3 (Jacobi sweeps, double) */
static void smallest_eigvec3(double A[3][3], double v[3])
{
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
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

/* CCICP::doHeightInterpolate, icpTools.cpp:301-381.  pose = x,y,z,qx,qy,qz,qw.  Returns the number of
 * wheel points that found a ground point within 3 m (:339-349); *z_out = the new z (:376) or the input z
 * when fewer than 4 did (:351,:379) or the normal is NaN (:367).  nn_idx (4 ints, optional) gets the
 * nearest ground point of every wheel point (lowest index on a tie), -1 for an empty cloud. */
int occicp_height(const float *ground, int n, int stride, const double pose[7], double *z_out, int *nn_idx)
{
    const double ROBO_HEIGHT = 1.45, wheel = 0.5; /* :303-305 */
    *z_out = pose[2];
    /* tf::Matrix3x3(q) (setRotation) in double, stored to an Eigen::Matrix4f (:321-329) */
    const double x = pose[3], y = pose[4], z = pose[5], w = pose[6];
    const double d = x * x + y * y + z * z + w * w, s = 2.0 / d;
    const double xs = x * s, ys = y * s, zs = z * s, wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys,
                 xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
    const float M[3][4] = {{(float)(1.0 - (yy + zz)), (float)(xy - wz), (float)(xz + wy), (float)pose[0]},
                           {(float)(xy + wz), (float)(1.0 - (xx + zz)), (float)(yz - wx), (float)pose[1]},
                           {(float)(xz - wy), (float)(yz + wx), (float)(1.0 - (xx + yy)), (float)pose[2]}};
    float  corr[4][3];
    int    n_corr = 0, k = 0;
    for (int i = -1; i <= 1; i += 2)
        for (int j = -1; j <= 1; j += 2, ++k) { /* :311-318 */
            const float p[3] = {(float)(i * wheel), (float)(j * wheel), (float)(-1.0 * ROBO_HEIGHT)};
            float       q[3]; /* pcl::transformPointCloud: float matrix times float point */
            for (int r = 0; r < 3; ++r) q[r] = M[r][0] * p[0] + M[r][1] * p[1] + M[r][2] * p[2] + M[r][3];
            int   best = -1;
            float bd = INFINITY;
            for (int g = 0; g < n; ++g) { /* KdTreeFLANN::nearestKSearch(k = 1): exact, squared L2 in float */
                const float *c = ground + (size_t)g * stride;
                const float  dx = c[0] - q[0], dy = c[1] - q[1], dz = c[2] - q[2];
                const float  dd = dx * dx + dy * dy + dz * dz;
                if (dd < bd) {
                    bd = dd;
                    best = g;
                }
            }
            if (nn_idx) nn_idx[k] = best;
            if (best >= 0 && bd < 9.0f) { /* :345 */
                const float *c = ground + (size_t)best * stride;
                corr[n_corr][0] = c[0];
                corr[n_corr][1] = c[1];
                corr[n_corr][2] = c[2];
                ++n_corr;
            }
        }
    if (n_corr < 4) return n_corr; /* :351, "Height could not be determined" :379 */
    double mean[3] = {0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 3; ++r) mean[r] += (double)corr[i][r];
    for (int r = 0; r < 3; ++r) mean[r] /= 4.0;
    double C[3][3] = {{0}};
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) C[r][c] += ((double)corr[i][r] - mean[r]) * ((double)corr[i][c] - mean[c]);
    double nrm[3];
    smallest_eigvec3(C, nrm); /* NormalEstimation::computePointNormal -> solvePlaneParameters (:361-365) */
    if (isnan(nrm[0]) || isnan(nrm[1]) || isnan(nrm[2])) return n_corr;
    if (nrm[2] < 0) { /* :369-372 */
        nrm[0] = -nrm[0];
        nrm[1] = -nrm[1];
        nrm[2] = -nrm[2];
    }
    *z_out = (double)(float)((float)nrm[2] * ROBO_HEIGHT + (float)mean[2]); /* float dz :376, stored to a double */
    return n_corr;
}
