/*
 * slam_mi355x_measure.h -- entry points that exist only in the MEASUREMENT build of the library
 * (python -m slam_amd.build --measure: -DSLAM_MEASURE, slam_amd/lib/libslam_mi355x_measure.so).  That build
 * also honours the SLAM_ICP_* / SLAM_RAYCAST_* environment variables of the tools/ sweeps and carries
 * in-kernel stamps and the raycast ablation switches; the shipped library has none of this.
 */
#ifndef SLAM_MI355X_MEASURE_H
#define SLAM_MI355X_MEASURE_H
#include "slam_mi355x.h"
#ifdef __cplusplus
extern "C" {
#endif
/* the default ICP schedule as two launches timed by three events per call; out = mean ms of either launch */
int slam_icp_debug_phase_events(slam_icp_t *icp, int on);
int slam_icp_debug_phase_ms(slam_icp_t *icp, double out[2], int *calls);
/* in-kernel stamps of the last batch launched with SLAM_ICP_STAMPS=1: [scan][wavefront][9] ticks / their means */
int slam_icp_debug_stamps_raw(slam_icp_t *icp, long long *out, int cap_rows, int *rows);
int slam_icp_debug_stamps(slam_icp_t *icp, double out[9]);
/* the spread form's stamps of the last launch made with SLAM_SPREAD_STAMPS=1: [parts][iters][16]: wall-clock ticks of 10 ns and counts (icp_single.hip), scan 0 */
int slam_icp_debug_spread_stamps(slam_icp_t *icp, long long *out, size_t cap, int *parts, int *iters);
#ifdef __cplusplus
}
#endif
#endif
