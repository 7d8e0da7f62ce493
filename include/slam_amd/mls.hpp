// slam_amd/mls.hpp -- header-only adapter with the shape of class MLS
// (mls/include/mls/mls.h:104-242) in the rolling / occupancy mode local_mapper
// uses (local_mapper.cpp:29,86,107), over the C-ABI (slam_mi355x.h).
//
// The reference takes PCL clouds and geometry_msgs poses and runs the ground
// segmentation inside addToOccupancy (mls.cpp:59-67).  Both levels are here:
//   addToOccupancy(cloud_xyz, n, stride)            the one-cloud form: segmentGround on the device
//                                                   (slam_gseg_*), split into drv / ground clouds, the two
//                                                   point loops of mls.cpp:73-142 in their order, and the
//                                                   drv cloud appended to global_cloud (:144-149);
//   addToOccupancy(obstacle, n, ground, n, stride)  below the segmentation, for callers that have the
//                                                   segmented clouds already (PointXYZGD = 8 floats).
// Clouds are float arrays (x, y, z first, `stride` floats per point), poses the Pose struct below
// (the pose part of geometry_msgs::PoseStamped).  getDrivability() fills a struct laid out like
// nav_msgs/OccupancyGrid (mls.h:167-175): data[x + size_x*y] in {-1, 0, 100}, origin -res*size/2.
// global_cloud lives on the host, as the reference's PCL cloud does.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

#include "slam_mi355x.h"

namespace slam_amd {

#ifndef SLAM_AMD_POSE_DEFINED
#define SLAM_AMD_POSE_DEFINED
struct Pose { // geometry_msgs::Pose
    double x = 0, y = 0, z = 0;
    double qx = 0, qy = 0, qz = 0, qw = 1;
};
#endif

struct OccupancyGrid { // the fields of nav_msgs::OccupancyGrid that MLS fills
    struct {
        double   resolution = 0;
        uint32_t width = 0, height = 0;
        double   origin_x = 0, origin_y = 0;
    } info;
    std::vector<int8_t> data;
};

class MLS {
public:
    // mls.h:154: MLS(int size_x_, int size_y_, double res, bool roll, double robot_size = 1.45)
    MLS(int size_x_, int size_y_, double res, bool roll, double /*robot_size*/ = 1.45)
    {
        slam_grid_params p;
        slam_grid_default_params(&p);
        p.rolling = roll ? 1 : 0;
        rolling_ = roll;
        if (slam_grid_create(size_x_, size_y_, res, &p, &h_) != SLAM_OK) {
            std::fprintf(stderr, "%s\n", slam_last_error());
            h_ = nullptr;
        }
        grid_.info.resolution = res;
        grid_.info.width = (uint32_t)size_x_;
        grid_.info.height = (uint32_t)size_y_;
        grid_.info.origin_x = -(res * size_x_ / 2); // mls.h:171-172
        grid_.info.origin_y = -(res * size_y_ / 2);
        grid_.data.assign((size_t)size_x_ * size_y_, (int8_t)-1);
    }
    ~MLS()
    {
        slam_device_synchronize();
        for (void *p : {d_cloud_, d_labels_, d_gnd_, d_obs_, d_counts_}) slam_free(p);
        slam_host_free(h_pin_);
        slam_free(d_gc_in_), slam_free(d_gc_out_);
        slam_ccicp_destroy(cc_);
        slam_gseg_destroy(gseg_);
        slam_grid_destroy(h_);
    }
    MLS(const MLS &) = delete;
    MLS &operator=(const MLS &) = delete;

    void clearMap()                                                                           // mls.cpp:18-31
    {
        if (h_) slam_grid_clear(h_, nullptr);
        global_cloud_.clear();
    }
    void setPose(double x, double y)                                                          // mls.cpp:408-479
    {
        if (!h_) return;
        double x0 = 0, y0 = 0;
        slam_grid_get_pose(h_, &x0, &y0);
        slam_grid_set_pose(h_, x, y, nullptr);
        double x1 = 0, y1 = 0;
        slam_grid_get_pose(h_, &x1, &y1);
        if (rolling_ && !disable_pointcloud_ && (x1 != x0 || y1 != y0)) {
            // global_cloud follows the window (:433-454): shift by -dx*res, -dy*res, then PassThrough x, y within
            // +-min(size)*res/2 (float limits, closed interval)
            const float sx = (float)-(x1 - x0), sy = (float)-(y1 - y0);
            const float crop = (float)(std::min(grid_.info.width, grid_.info.height) * grid_.info.resolution / 2);
            size_t      k = 0;
            for (size_t i = 0; i + 2 < global_cloud_.size(); i += 3) {
                const float px = global_cloud_[i] + sx, py = global_cloud_[i + 1] + sy, pz = global_cloud_[i + 2];
                if (px >= -crop && px <= crop && py >= -crop && py <= crop) {
                    global_cloud_[k] = px, global_cloud_[k + 1] = py, global_cloud_[k + 2] = pz;
                    k += 3;
                }
            }
            global_cloud_.resize(k);
        }
    }
    void setPose(const Pose &p) { setPose(p.x, p.y); }
    // mls.cpp:59-150, the one-cloud form: segmentGround (groundSegmentation.cpp:91-468) on the device, the drv
    // (obstacle below robot height) and ground clouds, then the two point loops in the reference's order
    void addToOccupancy(const float *cloud_xyz, int n, int stride = 3) { add_cloud(cloud_xyz, n, stride, nullptr, nullptr); }
    // ... of the cloud as it is (R null) or turned by R (row-major) and offset by t first
    void add_cloud(const float *cloud_xyz, int n, int stride, const double *R, const double *t)
    {
        if (!h_ || n <= 0) return;
        if (!gseg_ && slam_gseg_create(nullptr, &gseg_) != SLAM_OK) return warn();
        if (n > cap_) {
            slam_device_synchronize();
            for (void *p : {d_cloud_, d_labels_, d_gnd_, d_obs_}) slam_free(p);
            d_cloud_ = d_labels_ = d_gnd_ = d_obs_ = nullptr;
            cap_ = n + n / 4;
            if (slam_malloc(&d_cloud_, sizeof(float) * (size_t)cap_ * 8) != SLAM_OK || slam_malloc(&d_labels_, (size_t)cap_) != SLAM_OK ||
                slam_malloc(&d_gnd_, 16 * (size_t)cap_) != SLAM_OK || slam_malloc(&d_obs_, 16 * (size_t)cap_) != SLAM_OK) {
                cap_ = 0;
                return warn();
            }
        }
        if (!d_counts_ && slam_malloc(&d_counts_, 16) != SLAM_OK) return warn();
        if (stride > 8) return (void)std::fprintf(stderr, "MLS::addToOccupancy: stride %d > 8 floats\n", stride);
        // the cloud as it came, behind the place the segmentation reads (d_cloud_ holds 8 floats per point: a turned copy of at
        // most 3 in front, the upload of at most 5 ... 8 behind it when a transform is asked for)
        const bool turn = R != nullptr;
        float     *d_up = turn ? static_cast<float *>(d_cloud_) + 3 * (size_t)cap_ : static_cast<float *>(d_cloud_);
        if (turn && stride > 5) return (void)std::fprintf(stderr, "MLS::addToMap: stride %d > 5 floats\n", stride);
        if (slam_memcpy_h2d(d_up, cloud_xyz, sizeof(float) * (size_t)n * stride, nullptr) != SLAM_OK) return warn();
        if (turn) { // mls.cpp:34-53 on the device (round 6: the host loop was 0.15 of a cloud's 0.35 ms)
            if (slam_grid_transform_cloud_dev(d_up, n, stride, R, t, static_cast<float *>(d_cloud_), nullptr) != SLAM_OK) return warn();
            stride = 3;
        }
        if (slam_gseg_segment_dev(gseg_, (const float *)d_cloud_, n, stride, (uint8_t *)d_labels_, nullptr) != SLAM_OK) return warn();
        if (slam_gseg_split_dev(gseg_, (const float *)d_cloud_, n, stride, (const uint8_t *)d_labels_, (float *)d_gnd_, (float *)d_obs_,
                                (int32_t *)d_counts_, nullptr) != SLAM_OK)
            return warn();
        // the two counts, and the drv cloud where global_cloud is kept, come back into PINNED memory (a read-back into pageable
        // memory is staged by the runtime: 0.1 ms for the cloud's 300 KB)
        if (!pin_reserve(64)) return warn();
        int32_t *counts = static_cast<int32_t *>(h_pin_); // ground, obstacle
        if (slam_memcpy_d2h(counts, d_counts_, 2 * sizeof(int32_t), nullptr) != SLAM_OK) return warn();
        const int32_t n_gnd = counts[0], n_drv = counts[1];
        const float  *drv_ = nullptr;
        if (!disable_pointcloud_ && n_drv > 0) { // *global_cloud += drv_cloud (:144-149): read back BEFORE the grid update is
            if (!pin_reserve(64 + 16 * (size_t)n_drv)) return warn(); // enqueued (it runs while the host appends)
            drv_ = reinterpret_cast<const float *>(static_cast<unsigned char *>(h_pin_) + 64);
            if (slam_memcpy_d2h(static_cast<unsigned char *>(h_pin_) + 64, d_obs_, 16 * (size_t)n_drv, nullptr) != SLAM_OK) return warn();
        }
        counts = nullptr; // (pin_reserve may have moved the block)
        if (slam_grid_add_scan_inorder_dev(h_, (const float *)d_obs_, n_drv, (const float *)d_gnd_, n_gnd, 4, nullptr) != SLAM_OK)
            return warn();
        if (drv_) {
            const size_t at = global_cloud_.size();
            global_cloud_.resize(at + 3 * (size_t)n_drv);
            float *dst = global_cloud_.data() + at;
            for (int i = 0; i < n_drv; ++i) {
                dst[3 * (size_t)i] = drv_[4 * (size_t)i];
                dst[3 * (size_t)i + 1] = drv_[4 * (size_t)i + 1];
                dst[3 * (size_t)i + 2] = drv_[4 * (size_t)i + 2];
            }
        }
        last_counts_[0] = n_drv;
        last_counts_[1] = n_gnd;
    }
    bool pin_reserve(size_t bytes)
    {
        if (bytes <= pin_cap_) return true;
        slam_device_synchronize();
        slam_host_free(h_pin_);
        h_pin_ = nullptr;
        pin_cap_ = 0;
        const size_t want = bytes + bytes / 2;
        if (slam_host_alloc(&h_pin_, want) != SLAM_OK) return false;
        pin_cap_ = want;
        return true;
    }
    // mls.cpp:34-53: setPose, then (rolling) the cloud turned into the global orientation and offset by the
    // sub-cell residual curPose - pose (tf::poseMsgToEigen + pcl::transformPointCloud: computed in double, stored
    // as float), then addToOccupancy of that cloud
    void addToMap(const float *cloud_xyz, int n, int stride, const Pose &pose)
    {
        setPose(pose);
        if (!h_ || n <= 0) return;
        if (!rolling_) return addToOccupancy(cloud_xyz, n, stride);
        double cx = 0, cy = 0;
        slam_grid_get_pose(h_, &cx, &cy);
        const double tx = cx - pose.x, ty = cy - pose.y, tz = pose.z;
        const double d = pose.qx * pose.qx + pose.qy * pose.qy + pose.qz * pose.qz + pose.qw * pose.qw, s2 = d > 0 ? 2.0 / d : 0.0;
        const double xs = pose.qx * s2, ys = pose.qy * s2, zs = pose.qz * s2, wx = pose.qw * xs, wy = pose.qw * ys, wz = pose.qw * zs,
                     xx = pose.qx * xs, xy = pose.qx * ys, xz = pose.qx * zs, yy = pose.qy * ys, yz = pose.qy * zs, zz = pose.qz * zs;
        const double r[9] = {1.0 - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0 - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0 - (xx + yy)};
        // (float)(r0 x + r1 y + r2 z + t) per coordinate, in double: slam_grid_transform_cloud_dev, the same floats as the host loop
        // this was until round 6 (tests/test_gpu_cpp_adapters.py holds the map against the oracle's, which transforms on the host)
        const double t3[3] = {tx, ty, tz};
        if (stride <= 5) return add_cloud(cloud_xyz, n, stride, r, t3);
        trans_.resize(3 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            const double px = cloud_xyz[(size_t)i * stride], py = cloud_xyz[(size_t)i * stride + 1], pz = cloud_xyz[(size_t)i * stride + 2];
            trans_[3 * (size_t)i] = (float)(r[0] * px + r[1] * py + r[2] * pz + tx);
            trans_[3 * (size_t)i + 1] = (float)(r[3] * px + r[4] * py + r[5] * pz + ty);
            trans_[3 * (size_t)i + 2] = (float)(r[6] * px + r[7] * py + r[8] * pz + tz);
        }
        addToOccupancy(trans_.data(), n, 3);
    }
    // mls.cpp:481-505: the z offset from graph_slam; the occupancy mode keeps no heights, global_cloud moves
    void offsetMap(const Pose &pose)
    {
        if (disable_pointcloud_) return;
        for (size_t i = 2; i < global_cloud_.size(); i += 3) global_cloud_[i] += (float)pose.z;
    }
    // mls.cpp:508-518: pcl::VoxelGrid(xy, xy, z) over global_cloud: one centroid per occupied voxel, in increasing
    // voxel index (x fastest), the voxel lattice anchored at the cloud's minimum as PCL anchors it
    void filterPointCloud(double xy, double z)
    {
        const size_t n = global_cloud_.size() / 3;
        if (!n) return;
        // On the device (round 6): local_mapper filters its global cloud behind EVERY cloud (local_mapper.cpp:111), and the host
        // filter below -- an ordered map over 90 000 points -- was 4.3 of a cloud's 4.6 ms.  The library's pcl::VoxelGrid
        // (slam_ccicp_voxel_downsample_dev: the same lattice and order, centroids from exact 64-bit sums) takes the cloud as it
        // lies; a lattice beyond its accumulator (a stray point far off) falls back to the host.
        if (filter_on_device(xy, z)) return;
        filter_on_host(xy, z);
    }
    bool filter_on_device(double xy, double z)
    {
        const size_t n = global_cloud_.size() / 3;
        if (n > (size_t)1 << 30) return false;
        if (!cc_ && slam_ccicp_create(&cc_) != SLAM_OK) return false;
        if (n > gc_cap_) {
            slam_device_synchronize();
            slam_free(d_gc_in_), slam_free(d_gc_out_);
            d_gc_in_ = d_gc_out_ = nullptr;
            gc_cap_ = 0;
            const size_t want = n + n / 2;
            if (slam_malloc(&d_gc_in_, 12 * want) != SLAM_OK || slam_malloc(&d_gc_out_, 16 * want) != SLAM_OK) return false;
            gc_cap_ = want;
        }
        if (!pin_reserve(64 + 16 * n)) return false;
        if (slam_memcpy_h2d(d_gc_in_, global_cloud_.data(), 12 * n, nullptr) != SLAM_OK) return false;
        int n_out = 0;
        if (slam_ccicp_voxel_downsample_dev(cc_, (const float *)d_gc_in_, nullptr, (int)n, 3, (float)xy, (float)xy, (float)z, (float *)d_gc_out_,
                                            (int)n, &n_out, nullptr) != SLAM_OK)
            return false;
        float *rec = reinterpret_cast<float *>(static_cast<unsigned char *>(h_pin_) + 64);
        if (n_out > 0 && slam_memcpy_d2h(rec, d_gc_out_, 16 * (size_t)n_out, nullptr) != SLAM_OK) return false;
        global_cloud_.resize(3 * (size_t)n_out);
        for (int i = 0; i < n_out; ++i)
            for (int k = 0; k < 3; ++k) global_cloud_[3 * (size_t)i + k] = rec[4 * (size_t)i + k];
        return true;
    }
    void filter_on_host(double xy, double z)
    {
        const size_t n = global_cloud_.size() / 3;
        if (!n) return;
        float mn[3] = {global_cloud_[0], global_cloud_[1], global_cloud_[2]}, mx[3] = {mn[0], mn[1], mn[2]};
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) {
                mn[k] = std::min(mn[k], global_cloud_[3 * i + k]);
                mx[k] = std::max(mx[k], global_cloud_[3 * i + k]);
            }
        const float inv[3] = {(float)(1.0 / xy), (float)(1.0 / xy), (float)(1.0 / z)};
        long        lo[3], div[3];
        for (int k = 0; k < 3; ++k) {
            lo[k] = (long)std::floor(mn[k] * inv[k]);
            div[k] = (long)std::floor(mx[k] * inv[k]) - lo[k] + 1;
        }
        struct Acc { double s[3] = {0, 0, 0}; long n = 0; };
        std::map<long, Acc> vox;
        for (size_t i = 0; i < n; ++i) {
            long idx = 0, mul = 1;
            for (int k = 0; k < 3; ++k) {
                idx += ((long)std::floor(global_cloud_[3 * i + k] * inv[k]) - lo[k]) * mul;
                mul *= div[k];
            }
            Acc &a = vox[idx];
            for (int k = 0; k < 3; ++k) a.s[k] += global_cloud_[3 * i + k];
            ++a.n;
        }
        global_cloud_.clear();
        for (const auto &kv : vox)
            for (int k = 0; k < 3; ++k) global_cloud_.push_back((float)(kv.second.s[k] / (double)kv.second.n));
    }
    const std::vector<float> &getGlobalCloud() const { return global_cloud_; }                // mls.h:216 (x, y, z per point)
    void setDisablePointCloud(bool v) { disable_pointcloud_ = v; }                            // mls.h:223
    const int *lastSegmentCounts() const { return last_counts_; }                             // drv, ground points of the last cloud
    // mls.cpp:59-150 below the segmentation: obstacle points +1.0, ground points -0.3, in that order
    void addToOccupancy(const float *obstacle, int n_obstacle, const float *ground, int n_ground, int stride = 8)
    {
        if (h_ && slam_grid_add_scan_inorder(h_, obstacle, n_obstacle, ground, n_ground, stride) != SLAM_OK)
            std::fprintf(stderr, "%s\n", slam_last_error());
    }
    // mls.cpp:34-53 (rolling branch): setPose, then the points (already rotated into the
    // global orientation and offset by the sub-cell residual, as mls.cpp:41-47 does with PCL)
    void addToMap(const float *obstacle, int n_obstacle, const float *ground, int n_ground, double pose_x,
                  double pose_y, int stride = 8)
    {
        setPose(pose_x, pose_y);
        addToOccupancy(obstacle, n_obstacle, ground, n_ground, stride);
    }
    const OccupancyGrid &getDrivability()                                                    // mls.h:215
    {
        if (h_) slam_grid_read_occupancy(h_, grid_.data.data());
        return grid_;
    }
    void setMinClusterPoints(double v) { if (h_) slam_grid_set_min_cluster_points(h_, (int)v); } // mls.h:235
    void setMaxRange(double v) { if (h_) slam_grid_set_max_range(h_, v); }                     // mls.h:237
    slam_grid_t *handle() { return h_; }

private:
    void warn() const { std::fprintf(stderr, "MLS: %s\n", slam_last_error()); }
    slam_grid_t  *h_ = nullptr;
    slam_gseg_t  *gseg_ = nullptr;
    OccupancyGrid grid_;
    bool          rolling_ = false, disable_pointcloud_ = false;
    void         *d_cloud_ = nullptr, *d_labels_ = nullptr, *d_gnd_ = nullptr, *d_obs_ = nullptr, *d_counts_ = nullptr;
    int           cap_ = 0, last_counts_[2] = {0, 0};
    std::vector<float> global_cloud_, trans_;
    void              *h_pin_ = nullptr; // pinned: [counts | drv cloud, or the filtered global cloud]
    slam_ccicp_t      *cc_ = nullptr;    // filterPointCloud's voxel grid
    void              *d_gc_in_ = nullptr, *d_gc_out_ = nullptr;
    size_t             gc_cap_ = 0;
    size_t             pin_cap_ = 0;
};

} // namespace slam_amd
