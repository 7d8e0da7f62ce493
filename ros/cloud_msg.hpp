// ros/cloud_msg.hpp -- sensor_msgs/PointCloud2 <-> x, y, z float arrays without PCL (the reference goes through
// pcl::fromROSMsg / pcl::toROSMsg, scan_registration.cpp:73,115,163 and local_mapper.cpp:56,113): the three FLOAT32
// fields named x, y, z are looked up by name, every other field is ignored.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <sensor_msgs/PointCloud2.h>

namespace slam_amd_ros {

// x, y, z of every point, 3 floats each; false when the message has no such FLOAT32 fields
inline bool cloud_to_xyz(const sensor_msgs::PointCloud2 &msg, std::vector<float> &xyz)
{
    int off[3] = {-1, -1, -1};
    for (const auto &f : msg.fields)
        for (int k = 0; k < 3; ++k)
            if (f.name == std::string(1, "xyz"[k]) && f.datatype == sensor_msgs::PointField::FLOAT32) off[k] = (int)f.offset;
    if (off[0] < 0 || off[1] < 0 || off[2] < 0 || msg.point_step == 0) return false;
    const size_t n = (size_t)msg.width * msg.height;
    xyz.resize(3 * n);
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *p = msg.data.data() + i * msg.point_step;
        for (int k = 0; k < 3; ++k) std::memcpy(&xyz[3 * i + k], p + off[k], 4);
    }
    return true;
}

// an unorganised cloud of x, y, z FLOAT32 fields (what pcl::toROSMsg writes for pcl::PointXYZ, without its padding)
inline void xyz_to_cloud(const float *xyz, size_t n, sensor_msgs::PointCloud2 &msg)
{
    msg.height = 1;
    msg.width = (uint32_t)n;
    msg.fields.resize(3);
    for (int k = 0; k < 3; ++k) {
        msg.fields[k].name = std::string(1, "xyz"[k]);
        msg.fields[k].offset = 4u * (uint32_t)k;
        msg.fields[k].datatype = sensor_msgs::PointField::FLOAT32;
        msg.fields[k].count = 1;
    }
    msg.is_bigendian = false;
    msg.point_step = 12;
    msg.row_step = 12u * (uint32_t)n;
    msg.is_dense = true;
    msg.data.resize(12 * n);
    if (n) std::memcpy(msg.data.data(), xyz, 12 * n);
}

} // namespace slam_amd_ros
