// slam_amd/mls.hpp -- header-only adapter with the shape of class MLS
// (mls/include/mls/mls.h:104-242) in the rolling / occupancy mode local_mapper
// uses (local_mapper.cpp:29,86,107), over the C-ABI (slam_mi355x.h).
//
// The reference takes PCL clouds and geometry_msgs poses and runs the ground
// segmentation inside addToOccupancy (mls.cpp:66-67); this adapter starts just
// below that: the caller hands over the already segmented obstacle ("drv") and
// ground points as float arrays (PointXYZGD is a 32-byte record = 8 floats, x, y, z first).
// getDrivability() fills a struct laid out like nav_msgs/OccupancyGrid
// (mls.h:167-175): data[x + size_x*y] in {-1, 0, 100}, origin -res*size/2.
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>

#include "slam_mi355x.h"

namespace slam_amd {

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
    ~MLS() { slam_grid_destroy(h_); }
    MLS(const MLS &) = delete;
    MLS &operator=(const MLS &) = delete;

    void clearMap() { if (h_) slam_grid_clear(h_, nullptr); }                                 // mls.cpp:18-31
    void setPose(double x, double y) { if (h_) slam_grid_set_pose(h_, x, y, nullptr); }      // mls.cpp:408-479
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
    slam_grid_t  *h_ = nullptr;
    OccupancyGrid grid_;
};

} // namespace slam_amd
