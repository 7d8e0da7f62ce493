/*
 * slam_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's per-scan hot path (servos/SLAM:
 * ccicp2d class-constrained ICP + mls/local_mapper occupancy update), used
 * ONLY as the checker in tests/, in __graft_entry__.smoke() and as the timed
 * `cpu_baseline` leg of bench.py.  Nothing under slam_amd/ may include, link
 * or call this file; the product path is the HIP library behind
 * include/slam_mi355x.h and fails loudly without a GPU.
 *
 * PARITY STATUS
 *   - pinned against the compiled reference (oracle/_ref, built from
 *     /root/reference/ccicp2d/src/matrix.cpp as it lies): the 2x2 solve of
 *     fitStep (svd -> V*U^T, icpPointToPoint.cpp:149-171), the 3x3
 *     Gauss-Jordan solve (matrix.cpp:420-508) and the U*V^T
 *     re-orthonormalisation (icpPointToPlane.cpp:88-95).  Golden vectors from
 *     that build are committed under tests/golden/.
 *   - PARITY UNPINNED for everything else: the reference holds no tests or
 *     golden vectors, and kdtree.cpp / icp.cpp /
 *     icpPointToPoint.cpp / mls.cpp need boost, ROS and PCL headers this image
 *     lacks, so they cannot be built here.  Those parts follow the reference
 *     text line by line (citations at each function).
 *   - Bresenham ray traversal and the point-to-line step have no compiled
 *     reference at all: this file IS their oracle.
 */
#ifndef SLAM_ORACLE_H
#define SLAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ kd-tree
 * kdtree2 (M. Kennel) as the reference vendors it: float data, bucketsize 12,
 * split at the mean of the max-spread coordinate, data rearranged into leaf
 * order, exact 1-NN with ball/box pruning.  kdtree.cpp:72-233, 378-391,
 * 515-683. */
typedef struct okd_tree okd_tree;
okd_tree *okd_build(const float *xy, int n);
void      okd_free(okd_tree *t);
/* 1-NN: *dis = squared float distance, *idx = ORIGINAL index (kdtree.h:31-35) */
void      okd_nn1(const okd_tree *t, float qx, float qy, float *dis, int *idx);
/* kdtree.cpp:360-375 n_nearest_brute_force restricted to nn=1, first minimum
 * (lowest original index) wins ties. */
void      obf_nn1(const float *xy, int n, float qx, float qy, float *dis, int *idx);
/* k nearest by brute force ordered by (dis, idx); used for 2-D normals
 * (icpPointToPlane.cpp:340-349 calls n_nearest_around_point(i,0,k)). */
void      obf_knn(const float *xy, int n, float qx, float qy, int k, int *idx_out);

/* ------------------------------------------------------------ small solves */
/* R_ = V*U^T of H = U W V^T (icpPointToPoint.cpp:159-162), closed form,
 * including the reflection branch (no determinant fix in the reference). */
void o_p2p_rotation(const double H[4], double R_[4]);
/* Gauss-Jordan with full pivoting, matrix.cpp:420-508. A (3x3,row-major) and
 * b are overwritten; returns 1 on success, 0 if a pivot < eps(1e-20). */
int  o_solve3(double A[9], double b[3]);
/* R_ = U*V^T of [[1,-w],[w,1]] (icpPointToPlane.cpp:88-95). */
void o_orthonormal_from_omega(double w, double R_[4]);

/* -------------------------------------------------------------------- ICP */
typedef struct oicp_model oicp_model;

/* Icp::Icp, icp.cpp:26-70: dim fixed at 2; returns NULL when
 * (n_ga + n_nga) < 5 (the reference logs and leaves null trees). */
oicp_model *oicp_create(const double *m_ga, int n_ga, const double *m_nga, int n_nga);
void        oicp_free(oicp_model *m);
/* computes per-model-point normals for the point-to-line mode over ALL model
 * points (GA then NGA) with k nearest neighbours (reference: 10). */
void        o_normal2(const double *nb_xy, int k, double n_out[2]); /* icpPointToPlane.cpp:279-305 */
void        oicp_compute_normals(oicp_model *m, int k);
const double *oicp_normals(const oicp_model *m);

#define OICP_NN_KDTREE 0
#define OICP_NN_BRUTE  1
#define OICP_MODE_P2P  0   /* IcpPointToPoint::fitStep, icpPointToPoint.cpp:33-172 */
#define OICP_MODE_P2L  1   /* IcpPointToPlane::fitStep 2-D, icpPointToPlane.cpp:37-107 */

typedef struct {
    int    max_iter;   /* icp.cpp:27  (20) */
    double min_delta;  /* icp.cpp:27  (1e-6) */
    double indist;     /* icpTools.cpp:188 (5.0, compared with SQUARED distance) */
    int    nn_method;  /* OICP_NN_* */
    int    mode;       /* OICP_MODE_* */
} oicp_params;

/* One step. R (2x2 row-major) and t are updated in place; returns delta, or
 * -1 with R,t untouched when there is no correspondence.  *n_corr receives
 * the number of correspondences.  corr_idx (optional, n_tga+n_tnga ints)
 * receives per scene point the ORIGINAL model index within its class or -1. */
double oicp_fit_step(const oicp_model *m, const double *t_ga, int n_tga,
                     const double *t_nga, int n_tnga, double R[4], double t[2],
                     const oicp_params *p, int *n_corr, int *corr_idx);

/* Icp::fit + fitIterate, icp.cpp:80-122.  trace (optional) receives 8 doubles
 * per executed step: R00 R01 R10 R11 t0 t1 delta n_corr.  Returns the number
 * of executed steps (0 when the template has < 5 points: fit returns early). */
int oicp_fit(const oicp_model *m, const double *t_ga, int n_tga,
             const double *t_nga, int n_tnga, double R[4], double t[2],
             const oicp_params *p, double *trace, int *n_corr_last, double *delta_last);

/* Batch over independent scans, OpenMP over scans (n_threads<=0: all cores).
 * pts: xy f64 of all scans concatenated, scan s = [scan_off[s], scan_off[s+1]),
 * its first scan_nga[s] points are class GA, the rest NGA. */
void oicp_fit_batch(const oicp_model *m, const double *pts, const int *scan_off,
                    const int *scan_nga, int n_scans, double *R, double *t,
                    const oicp_params *p, int *iters, int *n_corr, double *delta,
                    int n_threads);

/* IcpPointToPoint::getEdgeWeight, icpPointToPoint.cpp:233-316, bug for bug
 * (dy = ax - bx at :262).  pm/pt: n correspondences (xy f64). */
void oicp_edge_weight(const double *pm, const double *pt, int n, double eW[9]);

/* ------------------------------------------------------------------- grid */
typedef struct {
    int    size_x, size_y;
    double resolution;
    double max_range;            /* mls.h:161 (75) */
    double occupancy_increment;  /* mls.h:188 (1.0) */
    double occupancy_decrement;  /* mls.h:189 (0.3) */
    int    min_cluster_points;   /* mls.h:165 (10; local_mapper.cpp:86 sets 20) */
    int    rolling;              /* mls.h:154 */
    double pose_x, pose_y;       /* curPose, used by the non-rolling range gate */
} ogrid_params;

/* mls.cpp:77-90: cell of one point or -1 when gated out.  *cx,*cy = window
 * coordinates (before the toroidal wrap). */
int ogrid_cell(const ogrid_params *g, float px, float py, int *cx, int *cy);

/* mls.cpp:73-142 on flat int32 planes in WINDOW coordinates (hits/misses
 * [x + size_x*y]); xyz arrays have `stride` floats per point (x,y first).
 * cell_out (optional) gets the linear cell index or -1 per point, obstacle
 * points first. Returns the number of counter increments. */
long ogrid_add_endpoints(const ogrid_params *g, const float *obs, int n_obs,
                         const float *gnd, int n_gnd, int stride,
                         int32_t *hits, int32_t *misses, int *cell_out);

/* Integer Bresenham (own oracle; not in the reference): origin and end are
 * map-frame xy (f32).  Every traversed cell before the end cell gets
 * misses+=1, the end cell hits+=1.  A beam is skipped when its end point
 * fails ogrid_cell() or its origin cell lies outside the window.  Returns
 * the number of counter increments. */
long ogrid_raycast(const ogrid_params *g, const float *origin_xy, const float *end_xy,
                   int n, int32_t *hits, int32_t *misses);

/* ogrid_raycast with OpenMP over beams and atomic increments (CPU baseline). */
long ogrid_raycast_mt(const ogrid_params *g, const float *origin_xy, const float *end_xy,
                      int n, int32_t *hits, int32_t *misses, int n_threads);

/* (float)(R*p+t) exactly as icpPointToPoint.cpp:69-70 forms its query:
 * double arithmetic, then a float store.  out_xy: n*2 floats. */
void o_transform_points(const double *pts, int n, const double R[4], const double t[2],
                        float *out_xy);

/* SURVEY 8(a) G3 applied once on summed counts: num_pts and the occupancy byte
 * (100 / 0 / previous) per cell. occ is in/out. */
void ogrid_finalize(const ogrid_params *g, const int32_t *hits, const int32_t *misses,
                    double *num_pts, int8_t *occ);

/* Reference-order semantics for ONE scan (mls.cpp:73-142): sequential +=/-=
 * on the per-cell double and the threshold writes, all obstacle points before
 * all ground points. drivable: -1/0/1 per cell. */
void ogrid_add_scan_inorder(const ogrid_params *g, const float *obs, int n_obs,
                            const float *gnd, int n_gnd, int stride,
                            double *num_pts, int8_t *drivable, int8_t *occ);

/* ------------------------------------------------- ground segmentation
 * GP-INSAC ground segmentation, ground_segmentation/src/groundSegmentation.cpp:
 * genPolarBinGrid :110-162, genGPModel :165-185, sectorINSAC :196-468, with the
 * constructor's parameters (:31-55).  PARITY UNPINNED: the reference needs
 * PCL/Eigen/ROS (absent here) and holds no test vectors; Eigen's dense
 * inverse() (:303) is restated as an LU solve with partial pivoting, so values
 * agree to rounding, not bitwise.  Ties of std::sort (:229) are broken by bin
 * index here (unspecified there). */
#define OGSEG_NUMBINSA 72  /* groundSegmentation.h:18 */
#define OGSEG_NUMBINSL 200 /* groundSegmentation.h:19 */
#define OGSEG_INVALID  1000 /* groundSegmentation.h:17 */

typedef struct {
    double rmax;           /* 100  */
    int    num_seedpoints; /* 10   */
    double p_l, p_sf, p_sn, p_tmodel, p_tdata, p_tg; /* 10, 1, 0.3, 5, 5, 0.3 */
    double robot_height;   /* 1.2  */
    double max_seed_range, max_seed_height; /* 50, 15 */
} ogseg_params;

void ogseg_default_params(ogseg_params *p);

#define OGSEG_DROPPED   0 /* point appears in no output cloud */
#define OGSEG_GROUND    1 /* gCloud */
#define OGSEG_OBSTACLE  2 /* oCloud and dCloud (below robot height: blocks driving) */
#define OGSEG_OVERHEAD  3 /* oCloud only (drivable = 1) */

/* xyz: n points, `stride` floats apart.  labels: one OGSEG_* per point.
 * Optional outputs (may be NULL): bin_of[n] = sector*200+bin or -1;
 * sector_model[72*200] = 1 where the bin ended in the ground model, 2 where it
 * stayed a candidate (value in sector_value = model height or GP mean f_s), 0
 * otherwise. Returns the number of INSAC outer iterations summed over sectors. */
int ogseg_segment(const ogseg_params *p, const float *xyz, int n, int stride, unsigned char *labels,
                  int *bin_of, unsigned char *sector_model, double *sector_value);

/* CCICP::classifyPoints (icpTools.cpp:36-103): per obstacle point 1 = ground adjacent,
 * 0 = not, 255 = dropped (outside the 1200 x 1200 x 0.5 m lattice or in its edge cells). */
void occicp_classify(const float *xyz, int n, int stride, unsigned char *flags);

/* ------------------------------------------------- CCICP facade steps (ccicp_oracle.c)
 * crop, voxel grid, GA/NGA split and the height recovery around the ICP call
 * (icpTools.cpp:222-381, 611-634).  PARITY UNPINNED: restated from the published PCL 1.7 algorithms the
 * reference calls; see the header of ccicp_oracle.c. */
int  occicp_crop(const float *xyz, int n, int stride, double cur_x, double cur_y, double crop, unsigned char *keep);
int  ovoxel_downsample(const float *in, int n, int stride, float lx, float ly, float lz, float *out);
void occicp_split(const float *xyzg, const unsigned char *keep, int n, int stride, int cap, double *ga, int *n_ga,
                  double *nga, int *n_nga);
int  occicp_height(const float *ground, int n, int stride, const double pose[7], double *z_out, int *nn_idx);

#ifdef __cplusplus
}
#endif
#endif
