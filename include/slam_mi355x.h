/*
 * slam_mi355x.h -- C-ABI of the MI355X (gfx950) implementation of the
 * servos/SLAM per-scan hot path: the ccicp2d class-constrained 2-D ICP scan
 * matcher and the mls/local_mapper occupancy-grid update.
 *
 * This is the drop-in boundary.  The reference has no
 * FFI layer: its hot path sits behind two C++ classes, so each entry point
 * below names the reference member it stands for (file:line under the
 * reference checkout).  INTEGRATION.md shows the adapter a maintainer adds to
 * ccicp2d/mls to call it; the headers in include/slam_amd/ are those adapters.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or framework types cross the ABI;
 *   - every function returns SLAM_OK (0) or a negative SLAM_E_* code and never
 *     throws; slam_last_error() gives the text of the calling thread's last
 *     failure;
 *   - "host" entry points take host memory and return when the result is in
 *     the caller's buffers (reference semantics: fit() is synchronous);
 *   - "_dev" entry points take DEVICE pointers, enqueue on `stream` and return
 *     without synchronising; inputs are borrowed until the stream reaches
 *     that point;
 *   - handles are opaque and not thread-safe (the reference is single-threaded:
 *     matrix.cpp:26-31 statics, node globals); use one handle per host thread.
 *   - there is no CPU fallback: without a usable HIP device every compute
 *     entry point fails with SLAM_E_NO_DEVICE.
 */
#ifndef SLAM_MI355X_H
#define SLAM_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_OK                       0
#define SLAM_E_INVALID               -1 /* bad argument */
#define SLAM_E_NO_DEVICE             -2 /* no HIP device / runtime not usable */
#define SLAM_E_HIP                   -3 /* a HIP call failed, see slam_last_error() */
#define SLAM_E_TOO_FEW_MODEL_POINTS  -4 /* icp.cpp:38-43: fewer than 5 model points */
#define SLAM_E_TOO_FEW_SCENE_POINTS  -5 /* icp.cpp:100-103, icpTools.cpp:179-184 */
#define SLAM_E_NOMEM                 -6
#define SLAM_E_UNSUPPORTED           -7
#define SLAM_E_TIMEOUT               -8 /* slam_mi355x_rccl.h: a merge's united range did not arrive: another rank has stopped */
#define SLAM_E_COMM                  -9 /* slam_mi355x_rccl.h: the communicator (or the host transport) reports a failure */

typedef void *slam_stream_t; /* a hipStream_t; NULL = the default stream */
typedef void *slam_event_t;  /* a hipEvent_t */

const char *slam_last_error(void);
const char *slam_version(void);

/* ------------------------------------------------------------ device plumbing */
int slam_device_count(int *n);
int slam_set_device(int ordinal);
int slam_device_info(char *name, int name_len, int *compute_units, size_t *hbm_bytes);
int slam_malloc(void **dptr, size_t bytes);
int slam_free(void *dptr);
int slam_memset(void *dptr, int value, size_t bytes, slam_stream_t stream);
int slam_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes, slam_stream_t stream);
int slam_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes, slam_stream_t stream);
int slam_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, slam_stream_t stream); /* asynchronous */
/* pinned host memory and copies that only enqueue (for overlapping transfers with kernels on
 * other streams; the host buffer must stay valid until the stream reaches the copy) */
int slam_host_alloc(void **hptr, size_t bytes);
/* 1 when hptr points into pinned host memory (slam_host_alloc, hipHostMalloc, hipHostRegister) -- memory an _async copy really
 * is asynchronous from --, 0 otherwise (pageable memory: the runtime stages such a copy and the call waits) */
int slam_host_is_pinned(const void *hptr);
int slam_host_free(void *hptr);
int slam_memcpy_h2d_async(void *dst_dev, const void *src_pinned, size_t bytes, slam_stream_t stream);
int slam_memcpy_d2h_async(void *dst_pinned, const void *src_dev, size_t bytes, slam_stream_t stream);
int slam_stream_wait_event(slam_stream_t stream, slam_event_t ev);
int slam_stream_create(slam_stream_t *stream);
/* priority > 0: the device's highest stream priority, < 0: its lowest, 0: the middle.  When workgroups of several
 * streams wait for CUs, those of the higher priority are placed first. */
int slam_stream_create_with_priority(slam_stream_t *stream, int priority);
/* A stream whose kernels leave `reserve_per_xcd` CUs of every XCD alone (hipExtStreamCreateWithCUMask): for work that
 * would otherwise hold every CU for its whole duration -- the registration batches, 0.6 ms per workgroup -- while short
 * kernels of other streams (an RCCL all-reduce, the grid update) wait for a CU to come free.  A stream made here has a hardware
 * queue of its own (the runtime deals ordinary streams over a few shared queues, where one stream's launches can stand behind
 * another's: DESIGN.md 4.6).  With reserve_per_xcd = 0 the mask names every CU.  Two differences from slam_stream_create remain,
 * both measured (tools/exp/stream_flags.hip, round 5): the stream has a hardware queue of its own, and it is a BLOCKING stream
 * (hipExtStreamCreateWithCUMask makes hipStreamDefault streams: hipStreamGetFlags = 0) -- a kernel on it waits for whatever the
 * process has put on the legacy null stream (a synchronous hipMemcpy, a framework's default stream) and the null stream waits for
 * it, where every other stream of this library is hipStreamNonBlocking.  The library itself enqueues nothing on the null stream in
 * its asynchronous calls; a caller that does serialises with these streams.  reserve_per_xcd > 0 needs the CU numbering this was
 * measured on -- an unpartitioned gfx950 of 256 CUs -- and returns SLAM_E_UNSUPPORTED elsewhere. */
int slam_stream_create_reserving_cus(slam_stream_t *stream, int reserve_per_xcd);
int slam_stream_destroy(slam_stream_t stream);
int slam_stream_synchronize(slam_stream_t stream);
int slam_device_synchronize(void);
/* hipGraph capture of a sequence of this library's asynchronous calls on one (created) stream: record once
 * after a warm-up call of the same sequence, replay with one launch. */
typedef void *slam_graph_t;
int slam_graph_begin_capture(slam_stream_t stream);
int slam_graph_end_capture(slam_stream_t stream, slam_graph_t *out);
int slam_graph_launch(slam_graph_t graph, slam_stream_t stream);
int slam_graph_destroy(slam_graph_t graph);
int slam_event_create(slam_event_t *ev);
int slam_event_destroy(slam_event_t ev);
int slam_event_record(slam_event_t ev, slam_stream_t stream);
int slam_event_synchronize(slam_event_t ev);
int slam_event_query(slam_event_t ev, int *done);   /* hipEventQuery without blocking: *done = 1 when everything recorded before the event has finished (or it was never recorded) */
int slam_event_elapsed_ms(slam_event_t start, slam_event_t stop, float *ms);

/* ------------------------------------------------------------------- ICP
 * Stands for class IcpPointToPoint : Icp (icpPointToPoint.h:26-40, icp.h:33-102). */
typedef struct slam_icp slam_icp_t;

#define SLAM_ICP_P2P 0 /* IcpPointToPoint::fitStep, icpPointToPoint.cpp:33-172 (what the reference runs) */
#define SLAM_ICP_P2L 1 /* point-to-line 3x3 normal equations, icpPointToPlane.cpp:37-107 (not compiled upstream) */

typedef struct {
    int    max_iter;        /* icp.cpp:27 max_iter(20); icp.h:51 setMaxIterations */
    double min_delta;       /* icp.cpp:27 min_delta(1e-6); icp.h:54 setMinDeltaParam */
    int    mode;            /* SLAM_ICP_P2P | SLAM_ICP_P2L */
    int    normals_k;       /* P2L: neighbours per normal (icpPointToPlane.h: 10) */
    int    lanes_per_point; /* 0 = library default: the ring search (2 lanes per scan point) for
                               the first iterations of a scan, then the halo-list sweeps (one lane
                               per point), both in one launch; N = 1,2,4,8,16,32,64:
                               ring search only, N lanes per point; -1 = ring search, lanes chosen
                               per pass; -2 = list sweeps for every iteration (measurements) */
    double cell_size;       /* model lattice pitch in metres; 0 = sized to fit LDS */
    int    force_global;    /* 1 = keep the model index in HBM/L2 even if it fits LDS */
    int    build_on_host;   /* 1 = build the model index with the single-threaded host code instead of the
                               device kernels (the reference the device build is verified against: same bytes) */
    int    first_iterations;/* default schedule: ring-search iterations before a scan may change to list sweeps
                               (0 = library default: 10 point-to-point, 2 point-to-line) */
    int    far_div;         /* default schedule: hand over once at most n / far_div queries are beyond the lists'
                               certified radius (0 = library default, 32) */
    int    split_launch;    /* 1 = the two search forms as two launches instead of one (same results) */
    int    spread_scans;    /* batches of at most this many scans (one, in the reference's usage) run in the
                               spread form: every scan over many workgroups of one persistent launch, 64 lanes
                               per query (icp_single.hip); 0 = library default (CUs / 16), -1 = never */
    int    pair_scans;      /* default schedule, batches: two scans per workgroup sharing one LDS index, each on half the
                               wavefronts (icp_fit_pair_kernel): 1 = one lane per scan point in the ring search,
                               2 = two lanes, 0 = library default (two lanes, for batches of at least two scans per
                               CU), -1 = never (one scan per workgroup) */
    int    spread_wait_us;  /* spread form: how long a scan's workgroups wait at their FIRST exchange for one another to
                               become resident before the scan is handed to the one-workgroup form instead (another
                               spread launch, another process or a persistent kernel may hold the CUs they need; a fit
                               never fails for it, icp.cpp:80-114); 0 = library default (5000), < 0 = hand over at once */
    int    wave_tiles;      /* a model whose index does not fit LDS (the reference's cap is 2 x 19 999 points, icpTools.h:21):
                               1 = every wavefront stages the model TILE of its two passes -- the union of its queries' 3 x 3 cell
                               blocks, one contiguous span per lattice row -- into its own 9 KB of the otherwise empty LDS, keeps it
                               over the iterations and searches there; 0 = library default: every query through L2.  Measured in
                               round 5: the tiles halve the kernel's L2 traffic and make it 20-30 % SLOWER -- the
                               kernel is bound by instruction issue, not by its loads -- so the default is off; the tiled form is
                               kept for models sparse enough to stage once (tests/test_gpu_icp_tile.py pins its results) */
    double list_min_halo;   /* halo lists: the lattice built is the first -- finest -- candidate whose lists fit LDS;
                               with this set, the first whose HALO is at least this many metres, if one fits (any, otherwise).  A
                               list answers a query whose neighbour is nearer than its halo; the others go to the cooperative
                               round.  0 = library default, < 0 = no preference (the finest that fits) */
    int    spread_tile;     /* spread form (one cloud at a time against up to 2 x 19 999 points: scan_registration.cpp:139-159,
                               icpTools.h:21): every workgroup keeps its scene points and their search state -- last neighbour, radius
                               proved empty -- in LDS for the whole fit and searches from last iteration's neighbour in straight-line
                               code (icp_single.hip).  For a model that does not fit LDS the workgroups take scene points
                               that are neighbours in space and either (1) stage the TILE of the index they can reach into LDS -- for
                               cells of more than 64 points, a lidar cloud's walls: config 3 0.45 -> 0.22 ms per fit -- or (2) search the
                               index where it lies.  Same correspondences; the sums in another order.  0 = library default (by the
                               model), 1 / 2 = that form wherever the index is not in LDS, -1 = the round-5 form */
} slam_icp_params;

typedef struct {
    int    iters;  /* fitStep calls executed (icp.cpp:116-122) */
    int    n_corr; /* Icp::getNumberCorrespondences of the last step */
    double delta;  /* last fitStep return value; -1 = no correspondences */
} slam_icp_result;

void slam_icp_default_params(slam_icp_params *p);

/* Icp::Icp(M_GA, M_NGA, M_GA_num, M_NGA_num, dim=2), icp.cpp:26-70.  The host
 * arrays (xy, f64) are copied (stored as f32, icp.cpp:51-60) and a uniform-cell
 * index replaces the two kd-trees; the exact-1-NN-in-float contract is kept. */
int  slam_icp_create(const double *m_ga, int n_ga, const double *m_nga, int n_nga,
                     const slam_icp_params *params, slam_icp_t **out);
/* The same with the model arrays resident in HBM (a target built from registered scans: the sliding local
 * map of a streaming mapper).  Builds on a stream of the library's own and returns with the index complete; the
 * arrays must be complete when the call is made (it does not order itself behind any stream of the caller). */
int  slam_icp_create_dev(const double *d_m_ga, int n_ga, const double *d_m_nga, int n_nga,
                         const slam_icp_params *params, slam_icp_t **out);
void slam_icp_destroy(slam_icp_t *icp);
int  slam_icp_set_max_iterations(slam_icp_t *icp, int val);  /* icp.h:51 */
int  slam_icp_set_min_delta(slam_icp_t *icp, double val);    /* icp.h:54 */
int  slam_icp_set_subsampling_step(slam_icp_t *icp, int val);/* icp.h:48 (stored, unused there too) */

/* Icp::fit(T_GA, T_NGA, T_GA_num, T_NGA_num, R, t, indist, h_dist), icp.cpp:80-114.
 * R (2x2 row-major) and t are in/out; host pointers; synchronous.  Returns
 * SLAM_E_TOO_FEW_SCENE_POINTS (R,t untouched) where the reference logs and
 * returns early. */
int slam_icp_fit(slam_icp_t *icp, const double *t_ga, int n_tga, const double *t_nga,
                 int n_tnga, double R[4], double t[2], double indist,
                 slam_icp_result *result);

/* The same over a batch of independent scans, all operands resident in HBM.
 *   d_pts      xy f64 of all scans, scan s = points [d_scan_off[s], d_scan_off[s+1])
 *   d_scan_nga number of class-GA points at the front of scan s (the rest is NGA)
 *   d_R, d_t   n_scans x 4 / x 2 doubles, in/out (initial -> registered pose)
 *   d_result   n_scans slam_icp_result (nullable)
 *   d_trace    nullable; n_scans x max_iter x 8 doubles: R00 R01 R10 R11 t0 t1
 *              delta n_corr after each executed step
 * Scans with fewer than 5 points are left untouched (iters = 0).
 * Asynchronous on `stream`.  Batches in the workgroup-per-scan forms (more than CUs / 16 scans, or spread_scans = -1)
 * keep nothing in the handle: calls on ONE handle may be in flight on several streams at once, and that is how a
 * stream of batches should be run -- two registration streams with pair_scans = 2, the grid update on a third
 * (DESIGN.md 4.6).  The spread form (few scans) uses scratch of the handle: one such call at a time per handle; calls
 * on DIFFERENT handles and streams are fine -- the library runs a process's spread launches one after the other on the
 * device, and a scan whose workgroups cannot become resident together (another process's launch, a persistent kernel
 * on the CUs) is redone by the workgroup-per-scan form inside the same call: every scan comes back registered. */
int slam_icp_fit_batch_dev(slam_icp_t *icp, const double *d_pts, const int32_t *d_scan_off,
                           const int32_t *d_scan_nga, int n_scans, double *d_R, double *d_t,
                           double indist, slam_icp_result *d_result, double *d_trace,
                           slam_stream_t stream);
/* The same with the initial poses read from d_R0 / d_t0 and the registered poses written to d_R / d_t (the two may
 * be the same arrays: that is slam_icp_fit_batch_dev).  Icp::fit takes R, t in/out because its caller keeps one pose
 * (icp.cpp:80-114); a stream of batches whose initial poses come from elsewhere (an odometry buffer, the batch before)
 * saves the copy into the output arrays and the launch gap behind it.  A scan with fewer than 5 points gets its
 * initial pose. */
int slam_icp_fit_batch_from_dev(slam_icp_t *icp, const double *d_pts, const int32_t *d_scan_off,
                                const int32_t *d_scan_nga, int n_scans, const double *d_R0, const double *d_t0,
                                double *d_R, double *d_t, double indist, slam_icp_result *d_result,
                                double *d_trace, slam_stream_t stream);

/* KDTree::n_nearest(qv, 1, result), kdtree.cpp:378-391, for n float queries
 * against one class (0 = GA, 1 = NGA): squared float distance and ORIGINAL
 * model index (kdtree.h:31-35).  Device pointers. */
int slam_icp_nearest_dev(slam_icp_t *icp, int cls, const float *d_query_xy, int n,
                         float *d_dis, int32_t *d_idx, slam_stream_t stream);

/* IcpPointToPoint::getEdgeWeight(eW), icpPointToPoint.cpp:233-316, for the
 * correspondences of the last slam_icp_fit() call. */
int slam_icp_get_edge_weight(slam_icp_t *icp, double eW[9]);

/* IcpPointToPlane::computeNormals, icpPointToPlane.cpp:340-349 (SLAM_ICP_P2L only): the unit
 * normal of every model point, GA points first then NGA, 2 doubles each. */
int slam_icp_get_normals(slam_icp_t *icp, double *normals_xy);

/* what the index looks like (for DESIGN.md / bench reporting) */
int slam_icp_index_info(slam_icp_t *icp, int *nx, int *ny, double *cell, int *in_lds,
                        size_t *lds_bytes, int *lanes_per_point);
/* How the index was built: on_device = 1 for the device kernels; ms[0] = host wall time of enqueueing the
 * build (upload, extent, plan, cell index, lists: about twenty launches), ms[1] = its one wait, for the plan the
 * device worked out (lattice, blob layout, the list lattice that fits); ms[2..3] = 0. */
int slam_icp_build_info(slam_icp_t *icp, int *on_device, double ms[4]);
/* The index as it lies in HBM: which = 0 the cell index, 1 the halo lists (0 bytes when the model has
 * none).  bytes (optional) receives the size; buf (optional, cap bytes) the content.  Synchronous. */
int slam_icp_index_blob(slam_icp_t *icp, int which, void *buf, size_t cap, size_t *bytes);
/* The default point-to-point schedule: two_forms = 1 when the halo lists fit LDS for this model (the ring
 * search then runs at least the first `first_iterations` iterations of a scan and the list sweeps the rest, in
 * one launch: the workgroup swaps its LDS contents); list lattice pitch, halo and certified radius in metres,
 * size of the list blob. */
int slam_icp_list_info(slam_icp_t *icp, int *two_forms, int *first_iterations, double *pitch, double *halo,
                       double *certified_radius, size_t *list_bytes);

/* ------------------------------------------------------------------ grid
 * Stands for class MLS in rolling/occupancy mode (mls.h:104-242) as
 * local_mapper uses it (local_mapper.cpp:29,86,107). */
typedef struct slam_grid slam_grid_t;

#define SLAM_RAYCAST_TILED  0 /* LDS-binned tiles, coalesced write-back */
#define SLAM_RAYCAST_GLOBAL 1 /* one global atomic per traversed cell */
#define SLAM_RAYCAST_TILED_MERGE 2 /* tiled, and the lanes of a wavefront that stand on one cell add once */

typedef struct {
    double max_range;           /* mls.h:161 (75) */
    double occupancy_increment; /* mls.h:188 (1.0) */
    double occupancy_decrement; /* mls.h:189 (0.3) */
    int    min_cluster_points;  /* mls.h:165 (10); local_mapper.cpp:86 sets 20 */
    int    rolling;             /* MLS(..., bool roll) mls.h:154 */
    int    raycast_impl;        /* SLAM_RAYCAST_* */
    int    raycast_seg_items;   /* tiled raycast: 64-beam blocks a workgroup takes from a tile's work list at a time (it goes
                                   on accumulating in the same LDS tile while the tile has blocks left and writes the tile
                                   back when it leaves it); 0 = library default (16); 8 ... 511 */
    int    raycast_wg_per_cu;   /* tiled raycast: persistent workgroups per CU; 0 = library default (two) */
    int    raycast_max_workgroups; /* tiled raycast: persistent workgroups in all; 0 = no cap (wg_per_cu x CUs) */
} slam_grid_params;

void slam_grid_default_params(slam_grid_params *p);

/* MLS::MLS(size_x, size_y, res, roll), mls.h:154-207 */
int  slam_grid_create(int size_x, int size_y, double resolution,
                      const slam_grid_params *params, slam_grid_t **out);
void slam_grid_destroy(slam_grid_t *g);
int  slam_grid_clear(slam_grid_t *g, slam_stream_t stream);            /* MLS::clearMap, mls.cpp:18-31 */
/* zeroes the hit/miss count planes only (batch / multi-GPU mode: a fresh local map per batch, so that the
 * all-reduce merges this batch's counts and not sums that were merged before) */
int  slam_grid_reset_counts(slam_grid_t *g, slam_stream_t stream);
int  slam_grid_set_min_cluster_points(slam_grid_t *g, int v);          /* mls.h:235 */
int  slam_grid_set_max_range(slam_grid_t *g, double v);                /* mls.h:237 */
/* MLS::setPose, mls.cpp:408-479.  Non-rolling: records curPose for the range
 * gate (mls.cpp:84-86).  Rolling: shifts the toroidal origin by
 * round(delta/res) cells and clears the cells that rolled in. */
int  slam_grid_set_pose(slam_grid_t *g, double x, double y, slam_stream_t stream);
int  slam_grid_get_pose(slam_grid_t *g, double *x, double *y);

/* MLS::addToOccupancy point loops, mls.cpp:73-142, on already segmented
 * clouds: obstacle points raise hits, ground points raise misses.  Points are
 * `stride` floats apart (x,y first; PCL PointXYZGD has stride 4). */
int slam_grid_add_endpoints(slam_grid_t *g, const float *obs, int n_obs, const float *gnd,
                            int n_gnd, int stride);
int slam_grid_add_endpoints_dev(slam_grid_t *g, const float *d_obs, int n_obs,
                                const float *d_gnd, int n_gnd, int stride,
                                slam_stream_t stream);

/* Bresenham free-space update (north-star extension; not in the reference):
 * for every beam the cells from the sensor cell up to (excluding) the end cell
 * get misses += 1, the end cell hits += 1.  origin/end are map-frame xy f32. */
int slam_grid_raycast(slam_grid_t *g, const float *origin_xy, const float *end_xy, int n);
int slam_grid_raycast_dev(slam_grid_t *g, const float *d_origin_xy, const float *d_end_xy,
                          int n, slam_stream_t stream);
/* The same straight from registered scans: end = (float)(R_s * p + t_s) formed
 * as icpPointToPoint.cpp:69-70 forms its query, origin = (float)t_s.
 * n_points = d_scan_off[n_scans] (the host built that array and knows it).  With a rolling
 * window the map-frame points are taken relative to the window centre (slam_grid_get_pose),
 * as mls.cpp:36-47 shifts the cloud before addToOccupancy. */
int slam_grid_raycast_scans_dev(slam_grid_t *g, const double *d_pts, const int32_t *d_scan_off,
                                int n_scans, int n_points, const double *d_R, const double *d_t,
                                slam_stream_t stream);
/* Reserves the raycast's scratch (beams, block boxes, work list) for calls of up to max_beams beams.  A raycast call
 * that finds its scratch too small grows it -- which frees the old block, and a free waits for the whole device: a
 * caller that streams batches of varying size (slam_mapper_* does this itself from max_points) reserves once instead. */
int slam_grid_reserve(slam_grid_t *g, int max_beams);

/* Folds the counts gathered since the last finalize into the per-cell evidence
 * value (Cluster::num_pts) and the occupancy byte by SURVEY 8(a) G3:
 *   c += inc*h; if (h>0 && c>min) occ=100;  c -= dec*m; if (m>0 && c<min) occ=0.
 * The count planes keep accumulating (they are the bit-exact contract). */
int slam_grid_finalize(slam_grid_t *g, slam_stream_t stream);
/* slam_grid_finalize followed by slam_grid_reset_counts, in one pass over the rows and one launch: what a batch step
 * ends with (fold this batch's counts into evidence and occupancy, leave the count planes zero for the next batch;
 * the accumulator planes, if any, are not touched).  Same evidence and occupancy as the two calls.  It alternates between two
 * row-range buffers from call to call, so it must not be recorded into a hipGraph that is replayed (slam_graph_*): a captured
 * step keeps slam_grid_finalize + slam_grid_reset_counts. */
int slam_grid_finalize_reset(slam_grid_t *g, slam_stream_t stream);

/* MLS::addToMap's cloud transform (mls.cpp:34-53, pcl::transformPointCloud with the pose's rotation and an offset): out = (float)(R p + t)
 * per point, R row-major, computed in double term by term as a host loop `r0*x + r1*y + r2*z + t` does (no contraction): the
 * same floats.  d_out_xyz: n x 3 floats; it may not overlap d_in_xyz unless stride == 3 and the two are the same array. */
int slam_grid_transform_cloud_dev(const float *d_in_xyz, int n, int stride, const double R[9], const double t[3], float *d_out_xyz,
                                  slam_stream_t stream);

/* One scan with the reference's own ordering and rounding (mls.cpp:73-142):
 * sequential += / -= on the per-cell double, thresholds after every point. */
int slam_grid_add_scan_inorder(slam_grid_t *g, const float *obs, int n_obs, const float *gnd,
                               int n_gnd, int stride);
/* the same with the two clouds resident in HBM (what slam_gseg_split_dev leaves); enqueues on `stream` */
int slam_grid_add_scan_inorder_dev(slam_grid_t *g, const float *d_obs, int n_obs, const float *d_gnd,
                                   int n_gnd, int stride, slam_stream_t stream);

/* read-back in WINDOW coordinates, row-major data[x + size_x*y] as
 * nav_msgs/OccupancyGrid (mls.h:167-175). Host pointers; synchronous. */
int slam_grid_read_counts(slam_grid_t *g, int32_t *hits, int32_t *misses);
int slam_grid_read_occupancy(slam_grid_t *g, int8_t *occ);      /* MLS::getDrivability()->data */
int slam_grid_read_num_pts(slam_grid_t *g, double *num_pts);
int slam_grid_total_updates(slam_grid_t *g, uint64_t *n);       /* counter increments so far */
int slam_grid_info(slam_grid_t *g, int *size_x, int *size_y, double *resolution,
                   int *origin_x, int *origin_y);
/* where a rolling window sits: the cells it has moved since creation (the sum of MLS::setPose's dx, dy,
 * mls.cpp:419-431; 0, 0 for a non-rolling grid).  Two grids hold the same world cells in the same storage rows
 * exactly when these agree (what a merge over the GPUs requires: slam_mi355x_rccl.h). */
int slam_grid_window_cell(slam_grid_t *g, int *cell_x, int *cell_y);

/* work list of the last tiled raycast: tiles of the window, (tile, 64-beam block) items, and how many times a
 * workgroup wrote a tile back (for reporting) */
int slam_grid_raycast_stats(slam_grid_t *g, int *n_tiles, int *n_items, int *n_segments);

/* the two int32 planes ([hits | misses], 2*size_x*size_y ints, toroidal
 * storage order) for a collective merge; see slam_mi355x_rccl.h */
int slam_grid_counts_dev(slam_grid_t *g, int32_t **d_planes, size_t *n_ints);
/* Whoever writes the planes through those pointers says which storage rows it wrote: the grid resets and finalizes
 * only the rows it knows to be touched (slam_grid_reset_counts, slam_grid_finalize).  The merges of
 * slam_mi355x_rccl.h do; asynchronous on `stream`, after the writes. */
int slam_grid_mark_rows(slam_grid_t *g, int row_lo, int row_hi, slam_stream_t stream);
/* Rows of the planes (storage order) that received counts since the planes were last reset or folded, tracked
 * on the device by the update kernels: a merge moves only these.  row_hi < row_lo = none.  The host form
 * synchronises; d_range = two ints {lowest row, -(highest row)} (0x7f7f7f7f each when none). */
int slam_grid_dirty_rows(slam_grid_t *g, int *row_lo, int *row_hi);
int slam_grid_dirty_rows_dev(slam_grid_t *g, int32_t **d_range);
/* Periodic merges of a running map (BASELINE config 5): with an accumulator, the count planes hold what THIS
 * GPU added since the last merge; after the all-reduce of those rows slam_grid_fold adds them to the
 * accumulator (the merged totals) and zeroes them, so that nothing is summed twice at the next merge.
 * Finalize and the read-backs see accumulator + planes. */
int slam_grid_enable_accumulator(slam_grid_t *g);
int slam_grid_fold(slam_grid_t *g, int row_lo, int row_hi, slam_stream_t stream);

/* ------------------------------------------------------ ground segmentation
 * Stands for class groundSegmentation (ground_segmentation/include/ground_segmentation/
 * groundSegmentation.h:67-128), the GP-INSAC pre-filter both halves of the path run first
 * (icpTools.cpp:114-115, mls.cpp:66-67): setupGroundSegmentation + segmentGround, with
 * the constructor's parameters (groundSegmentation.cpp:31-55; field = reference setter). */
typedef struct slam_gseg slam_gseg_t;

typedef struct {
    double rmax;                     /* set_rmax(100.0) */
    int    num_seedpoints;           /* set_num_seedpoints(10) */
    double gp_lengthparameter;       /* set_gp_lengthparameter(10) */
    double gp_covariancescale;       /* set_gp_covariancescale(1.0) */
    double gp_modelnoise;            /* set_gp_modelnoise(0.3) */
    double gp_groundmodelconfidence; /* set_gp_groundmodelconfidence(5.0) */
    double gp_grounddataconfidence;  /* set_gp_grounddataconfidence(5.0) */
    double gp_groundthreshold;       /* set_gp_groundthreshold(0.3) */
    double robotheight;              /* set_robotheight(1.2) */
    double seeding_maxrange;         /* set_seeding_maxrange(50) */
    double seeding_maxheight;        /* set_seeding_maxheight(15) */
} slam_gseg_params;

#define SLAM_GSEG_DROPPED  0 /* in no output cloud: beyond rmax, or in a bin with <= 5 points */
#define SLAM_GSEG_GROUND   1 /* groundCloud */
#define SLAM_GSEG_OBSTACLE 2 /* obsCloud and drvCloud (below robot height) */
#define SLAM_GSEG_OVERHEAD 3 /* obsCloud only (drivable = 1) */

void slam_gseg_default_params(slam_gseg_params *p);
int  slam_gseg_create(const slam_gseg_params *params, slam_gseg_t **out);
void slam_gseg_destroy(slam_gseg_t *h);
int  slam_gseg_reserve(slam_gseg_t *h, int max_points);
/* one label (SLAM_GSEG_*) per input point; points are `stride` floats apart (x, y, z first) */
int  slam_gseg_segment(slam_gseg_t *h, const float *xyz, int n, int stride, uint8_t *labels);
int  slam_gseg_segment_dev(slam_gseg_t *h, const float *d_xyz, int n, int stride, uint8_t *d_labels,
                           slam_stream_t stream);
/* the ground cloud and the drvCloud as (x, y, z, 0) records -- what slam_grid_add_endpoints_dev
 * takes with stride 4; d_counts[0] = ground points, d_counts[1] = obstacle points written */
int  slam_gseg_split_dev(slam_gseg_t *h, const float *d_xyz, int n, int stride, const uint8_t *d_labels,
                         float *d_ground_xyz4, float *d_obstacle_xyz4, int32_t *d_counts,
                         slam_stream_t stream);
/* CCICP::classifyPoints, icpTools.cpp:36-103: for every point of the obstacle cloud 1 = ground
 * adjacent (GA), 0 = not (NGA), 255 = dropped there too (outside the 1200 x 1200 x 0.5 m lattice
 * or in its outermost cells).  GA/NGA are the two classes the ICP matches separately. */
int  slam_gseg_classify_ga_dev(slam_gseg_t *h, const float *d_obstacle_xyz, int n, int stride,
                               uint8_t *d_flags, slam_stream_t stream);
/* the same where the number of points is known on the device only (*d_n, at most n_capacity): no host round trip */
int  slam_gseg_classify_ga_counted_dev(slam_gseg_t *h, const float *d_obstacle_xyz, const int32_t *d_n, int n_capacity,
                                       int stride, uint8_t *d_flags, slam_stream_t stream);
/* ... and with the extent of the points the classification keeps (pcl::getMinMax3D over the finite points whose flag is not
 * 255: what setSceneCloud's voxel filter starts from, icpTools.cpp:620-633) accumulated in the same pass: d_mm[0..2] minima,
 * d_mm[3..5] maxima as ORDERED floats (the bits with the sign bit flipped for positive, all bits for negative values), which
 * the caller has set to 0xffffffff / 0 beforehand */
int  slam_gseg_classify_ga_extent_dev(slam_gseg_t *h, const float *d_obstacle_xyz, const int32_t *d_n, int n_capacity,
                                      int stride, uint8_t *d_flags, uint32_t *d_mm, slam_stream_t stream);
/* per polar bin (72 x 200): 1 = in the ground model (value = prototype height), 2 = candidate
 * that stayed out (value = GP mean), 0 = no signal point; INSAC iterations per sector */
int  slam_gseg_read_model(slam_gseg_t *h, uint8_t *bin_state, double *bin_value, int32_t *sector_iterations);

/* ------------------------------------------------------------------------
 * CCICP facade steps either side of the ICP call.  The reference does these
 * with PCL (not part of its checkout): the published PCL 1.7 algorithms are restated; parity unpinned.
 * All device-resident; every result is deterministic.
 * ---------------------------------------------------------------------- */
typedef struct slam_ccicp slam_ccicp_t;
int  slam_ccicp_create(slam_ccicp_t **out);
void slam_ccicp_destroy(slam_ccicp_t *h);

/* The points whose ground-segmentation label (SLAM_GSEG_*) is in label_mask (bit 1 << label), in cloud
 * order, as (x, y, z, 0) records: (1 << SLAM_GSEG_OBSTACLE) | (1 << SLAM_GSEG_OVERHEAD) is the outcloud
 * CCICP::segmentGround classifies, 1 << SLAM_GSEG_GROUND its ground cloud (icpTools.cpp:106-119). */
int slam_ccicp_select_dev(slam_ccicp_t *h, const float *d_xyz, int n, int stride, const uint8_t *d_labels,
                          unsigned label_mask, float *d_out_xyz4, int *n_out, slam_stream_t stream);

/* CCICP::setSceneCloud / setTargetCloud voxel filter, icpTools.cpp:620-633 (pcl::VoxelGrid, leaf
 * 0.5,0.5,2 for obstacles, 0.5,0.5,5 for ground): one output point per occupied voxel, x,y,z = centroid,
 * [3] = ground_adj averaged as PCL averages every field (then stored to the uint16 field), in increasing
 * voxel index (x fastest).  d_xyz: `stride` floats per point; the class comes from d_flag (1 = GA, as
 * slam_gseg_classify_ga_dev writes it) or, when d_flag is null and stride > 3, from float [3] > 0.5.
 * d_out: 4 floats per voxel, room for max_out; *n_out = voxels produced.  Synchronises the stream twice
 * (lattice extent, count). */
int slam_ccicp_voxel_downsample_dev(slam_ccicp_t *h, const float *d_xyz, const uint8_t *d_flag, int n, int stride,
                                    float leaf_x, float leaf_y, float leaf_z, float *d_out, int max_out, int *n_out,
                                    slam_stream_t stream);

/* The cloud as CCICP::classifyPoints leaves it (icpTools.cpp:64-101): points it keeps, bin by bin (x bin
 * major, y bin minor), original order inside a bin, as x,y,z,ground_adj records -- the order in which
 * doICPMatch applies the ICP_MAX_PTS cap to the target cloud (SCAN_TO_MAP: setTargetCloud classifies
 * without a voxel filter, icpTools.cpp:591-595).  d_flag: the flags slam_gseg_classify_ga_dev wrote for
 * the same points. */
int slam_ccicp_bin_order_dev(slam_ccicp_t *h, const float *d_xyz, const uint8_t *d_flag, int n, int stride,
                             float *d_out_xyzg, int *n_out, slam_stream_t stream);

/* CCICP::doICPMatch(initPose) marshalling, icpTools.cpp:225-276: optional crop of +-crop_dist around
 * (cur_x,cur_y) (pcl::PassThrough on x then y; the reference crops the target cloud only, 75 m), then the
 * split by isGA(ground_adj) in cloud order with at most cap-1 points per class (ICP_MAX_PTS = 20000,
 * icpTools.h:21) as f64 xy -- the arrays slam_icp_create / slam_icp_fit take.  d_xyzg: x,y,z,ground_adj
 * with `stride` >= 4 floats per point (the voxel filter's output).  counts[0] = GA, counts[1] = NGA. */
int slam_ccicp_split_dev(slam_ccicp_t *h, const float *d_xyzg, int n, int stride, int crop, double cur_x, double cur_y,
                         double crop_dist, int cap, double *d_ga_xy, double *d_nga_xy, int counts[2],
                         slam_stream_t stream);

/* The same with the crop given as the box itself, box = {x_lo, x_hi, y_lo, y_hi} (closed intervals of floats, as
 * pcl::PassThrough compares them; null = no crop): CCICP::doICPMatch filters seg_target IN PLACE (icpTools.cpp:226-239), so the
 * points a later match sees are those inside the intersection of every window since setTargetCloud -- a box again.
 * totals (optional) = points of each class inside the box before the cap (what the reference's seg_target keeps). */
int slam_ccicp_split_box_dev(slam_ccicp_t *h, const float *d_xyzg, int n, int stride, const float box[4], int cap, double *d_ga_xy,
                             double *d_nga_xy, int counts[2], int totals[2], slam_stream_t stream);

/* CCICP::doHeightInterpolate, icpTools.cpp:301-381: the four wheel points of the pose (x,y,z,qx,qy,qz,qw)
 * find their nearest ground point (exact, squared distance < 9); with four of them the new z is
 * n_z * 1.45 + mean z of the four (n = their plane normal, n_z >= 0); otherwise z stays.  nn_idx
 * (optional) gets the four nearest indices. */
int slam_ccicp_height_dev(slam_ccicp_t *h, const float *d_ground, int n, int stride, const double pose[7], double *z_out,
                          int *n_corr, int nn_idx[4], slam_stream_t stream);

/* The steps above as ONE device-resident chain, for a caller that matches cloud after cloud (scan_registration.cpp:
 * 109-199): CCICP::segmentGround + classifyPoints + setSceneCloud's voxel filter (voxel != 0; 0 = setTargetCloud's
 * bin order) + doICPMatch's crop / class split / cap, stage after stage on `stream` with every count left on the device
 * -- the stepwise entry points return each count to the host, seven round trips per cloud.  Same results, bit for bit.
 *   d_pts    out: xy f64, the GA points then the NGA points (room for 2 * (cap - 1) points)
 *   d_scan   out: int32[3] = {0, n_ga + n_nga, n_ga}: d_scan_off = d_scan, d_scan_nga = d_scan + 2 of a
 *            slam_icp_fit_batch_dev call with n_scans = 1 that registers the cloud without the host knowing its size
 *   d_ground out, nullable: the ground cloud as x,y,z,0 records (room for n)
 *   d_counts out: int32[4] = {obstacle points, ground points, points after the filter, 1 if the voxel lattice did not
 *            fit the chain's accumulator (2 M voxels; the stepwise entry point takes larger extents)}
 * Nothing is read back and nothing waits: the caller reads d_counts when it needs them. */
int slam_ccicp_scene_dev(slam_ccicp_t *h, slam_gseg_t *seg, const float *d_xyz, int n, int stride, int voxel, int crop,
                         double cur_x, double cur_y, double crop_dist, int cap, double *d_pts, int32_t *d_scan,
                         float *d_ground, int32_t *d_counts, slam_stream_t stream);
/* doHeightInterpolate for the pose a registration left on the device (d_R 2x2, d_t of one scan; yaw = atan2(R10, R00),
 * icpTools.cpp:195-197) against a ground cloud whose size is known on the device (*d_n_ground <= n_capacity):
 * d_out[0] = z, d_out[1] = neighbours within 3 m.  Asynchronous. */
int slam_ccicp_height_pose_dev(slam_ccicp_t *h, const float *d_ground, const int32_t *d_n_ground, int n_capacity,
                               int stride, const double *d_R, const double *d_t, double z0, double *d_out,
                               slam_stream_t stream);
/* ... with the roll and pitch of the initial pose, which doICPMatch keeps beside the matched yaw (tf::createQuaternionFromRPY,
 * icpTools.cpp:205-212) and doHeightInterpolate turns the wheel points by (:321-332) */
int slam_ccicp_height_rpy_pose_dev(slam_ccicp_t *h, const float *d_ground, const int32_t *d_n_ground, int n_capacity,
                                   int stride, const double *d_R, const double *d_t, double z0, double roll, double pitch,
                                   double *d_out, slam_stream_t stream);
/* ... and, behind the height, `mirror_bytes` (a multiple of 8, at most 4096) copied from mirror_src (device) to mirror_dst by the
 * last kernel of the call: with mirror_dst in pinned host memory (slam_host_alloc) the caller's result block -- pose, result,
 * height, scan descriptor -- is on the host when the stream has drained, without a copy of its own behind the match */
int slam_ccicp_height_rpy_pose_mirror_dev(slam_ccicp_t *h, const float *d_ground, const int32_t *d_n_ground, int n_capacity,
                                          int stride, const double *d_R, const double *d_t, double z0, double roll, double pitch,
                                          double *d_out, void *mirror_dst, const void *mirror_src, size_t mirror_bytes,
                                          slam_stream_t stream);
/* The throughput form of a sequence of matches (BASELINE config 3; scan_registration.cpp:109-173 is one cloud at a time): n <= 32
 * scenes made by slam_ccicp_scene_dev -- on as many streams as the caller likes, each with its own slam_ccicp_t / slam_gseg_t --
 * gathered into ONE batch for slam_icp_fit_batch_dev: d_out_pts = their points one scene after the other (room for the sum),
 * d_scan_off[n + 1] / d_scan_nga[n] as that call reads them.  d_pts[k] / d_scan[k]: the d_pts / d_scan of scene k (host arrays of
 * device pointers; the sizes stay on the device).  Asynchronous: the caller makes `stream` wait for the scenes' streams first. */
int slam_ccicp_pack_scans_dev(int n, const double *const *d_pts, const int32_t *const *d_scan, double *d_out_pts, int32_t *d_scan_off,
                              int32_t *d_scan_nga, slam_stream_t stream);
/* The filtered cloud of the last slam_ccicp_scene_dev call (x, y, z, ground_adj records: seg_scene / seg_target of
 * icpTools.h:77-78; d_counts[2] of that call says how many are valid) copied to d_out_xyzg, at most `capacity` records. */
int slam_ccicp_scene_cloud_dev(slam_ccicp_t *h, float *d_out_xyzg, int capacity, slam_stream_t stream);

/* ------------------------------------------------------------------------
 * Streaming mapper (BASELINE config 5).  Stands where scan_registration (scan_registration.cpp:109-199: one
 * doICPMatch per scan against the target it keeps) feeds local_mapper (local_mapper.cpp:65-130: addToMap at
 * 50 Hz): chunks of scans go from pinned host memory through registration into the occupancy grid on three
 * HIP streams (copy of chunk k+1 | ICP of chunk k | grid update of chunk k-1), the ICP target is a sliding
 * window of the scans registered so far, and every merge_every chunks the grid is merged over the GPUs
 * (slam_mapper_use_comm, slam_mi355x_rccl.h) and finalized.
 * ---------------------------------------------------------------------- */
typedef struct slam_mapper slam_mapper_t;

typedef struct {
    int    grid_size_x, grid_size_y;
    double resolution;
    slam_grid_params grid;     /* rolling = 1: the window follows slam_mapper_push's window_x/y (mls.cpp:408-479) */
    slam_icp_params  icp;
    double indist;             /* icpTools.cpp:188 (5.0) */
    int    max_scans, max_points; /* reservation per chunk */
    int    window_chunks;      /* sliding local map: the target is the registered points of the last W <= 8 chunks
                                  (thinned or decimated); 0 = the model given at create stays the target */
    int    rebuild_every;      /* chunks between rebuilds of the sliding target (>= 1) */
    int    target_points;      /* points the sliding target holds at most, both classes (2 x 19999: icpTools.h:21) */
    int    keep_prior;         /* 1 = the model given at create stays part of every rebuilt target */
    int    merge_every;        /* chunks between merges over the GPUs + finalize; 0 = only at slam_mapper_finish */
    int    pipelined;          /* 1 = copy, registration (two in turn for a fixed target) and grid update on streams of their
                                  own; 0 = one stage after the other on one stream (same results) */
    int    strict_window;      /* 1 = the push a rebuild is due at waits for it: the chunk meets the target built from every
                                  chunk before it (reproducible; the pipeline drains once per rebuild); 0 = the build is
                                  enqueued (same window) and adopted by a later push, see background_rebuild */
    int    slots;              /* chunks in flight (device + pinned buffers each): 2..8; 0 = default: 5 -- with a fixed target the
                                  host enqueues chunk k while two registrations run on the two registration streams and
                                  the chunks before them are mapped and read back (256-scan chunks: two 0.54 ms per chunk,
                                  three 0.43, four 0.38, five 0.37); with a sliding target, whose chunks register one after
                                  the other (config 5, rebuilt every 4 chunks): four 0.410 ms per chunk, five 0.396, six
                                  0.397-0.436, eight 0.408-0.414 -- four is slow only where it equals rebuild_every (every 3 /
                                  5 / 8 chunks: four and five slots within 1 % of each other) */
    double thin_res;           /* > 0: the window is thinned to one point per cell of this pitch (metres) and class over
                                  the grid's extent, the oldest measurement of a cell kept (where pcl::VoxelGrid keeps a
                                  centroid, icpTools.cpp:620-633); 0: every stride-th point of a chunk instead */
    int    background_rebuild; /* 1 (default) = the sliding target's rebuild is enqueued on a stream of its own (no host wait:
                                  the index build plans itself on the device) and adopted by the first push that finds it
                                  complete (a push waits for it only when the next rebuild is due or after
                                  min(rebuild_every, 4) pushes); 0 or strict_window = the push waits for it at once */
    int    registration_streams; /* 0 = default (two in turn, scans in pairs, for a fixed target; one for a sliding target), 1, 2 */
} slam_mapper_params;

void slam_mapper_default_params(slam_mapper_params *p);
/* m_ga / m_nga: the prior map (host, f64 xy), the first ICP target (at least 5 points, icp.cpp:38-43) */
int  slam_mapper_create(const slam_mapper_params *params, const double *m_ga, int n_ga, const double *m_nga, int n_nga,
                        slam_mapper_t **out);
void slam_mapper_destroy(slam_mapper_t *m);
/* The producer fills the PINNED buffers of slot slam_mapper_next_slot() -- points (xy f64), scan_off
 * (n_scans + 1, from 0), scan_nga, initial poses R0 (4 per scan) and t0 (2 per scan) -- and pushes. */
int  slam_mapper_next_slot(slam_mapper_t *m, int *slot);
int  slam_mapper_slots(slam_mapper_t *m, int *n_slots);
int  slam_mapper_chunk_buffers(slam_mapper_t *m, int slot, double **pts, int32_t **scan_off, int32_t **scan_nga,
                               double **R0, double **t0);
/* enqueues the chunk in the next slot and returns at once; window_x/y: where a rolling grid is centred for it */
int  slam_mapper_push(slam_mapper_t *m, int n_scans, int n_points, double window_x, double window_y, int *slot);
/* blocks until the chunk last pushed into `slot` is registered and mapped; its poses to host memory (optional).  (The poses
 * come back through the slot's pinned R0 / t0 buffers, which hold the REGISTERED poses from here until the producer fills
 * them for the slot's next chunk.) */
int  slam_mapper_wait(slam_mapper_t *m, int slot, double *R_out, double *t_out);
/* last merge (if a communicator is installed), finalize, and waits for everything */
int  slam_mapper_finish(slam_mapper_t *m);
int  slam_mapper_grid(slam_mapper_t *m, slam_grid_t **grid);      /* owned by the mapper */
int  slam_mapper_target(slam_mapper_t *m, slam_icp_t **icp);      /* the current ICP target (owned by the mapper) */
int  slam_mapper_stats(slam_mapper_t *m, long *chunks, long *merges, long *rebuilds, double *rebuild_ms,
                       int last_merge_rows[2]);
/* How a merge is carried out: begin is enqueued behind a chunk's grid update, finish when the next chunk's
 * registration has been enqueued (so the GPU is busy while the host waits for the rows to merge).  Installed
 * by slam_mapper_use_comm; with none, the periodic step is finalize alone. */
typedef int (*slam_mapper_merge_fn)(void *ctx, slam_grid_t *grid, slam_stream_t stream, int *row_lo, int *row_hi);
int  slam_mapper_set_merge(slam_mapper_t *m, slam_mapper_merge_fn begin, slam_mapper_merge_fn finish, void *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_MI355X_H */
