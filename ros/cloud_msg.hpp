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
    if (off[0] + 4 == off[1] && off[1] + 4 == off[2]) { // x, y, z side by side (what every driver and pcl::toROSMsg write)
        if (msg.point_step == 12) {
            if (n) std::memcpy(xyz.data(), msg.data.data() + off[0], 12 * n);
        } else {
            for (size_t i = 0; i < n; ++i) std::memcpy(&xyz[3 * i], msg.data.data() + i * msg.point_step + off[0], 12);
        }
        return true;
    }
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *p = msg.data.data() + i * msg.point_step;
        for (int k = 0; k < 3; ++k) std::memcpy(&xyz[3 * i + k], p + off[k], 4);
    }
    return true;
}

// The message's own bytes as a strided float array -- x, y, z FLOAT32 side by side at a 4-byte aligned offset, point_step a multiple
// of 4 (a Velodyne driver's x, y, z, intensity, ring; pcl::toROSMsg's padded PointXYZ): no copy at all for a caller that hands the
// cloud on at once (the library's entry points take `stride` floats per point).  False: use cloud_to_xyz.
inline bool cloud_xyz_view(const sensor_msgs::PointCloud2 &msg, const float *&xyz, int &stride, size_t &n)
{
    int off[3] = {-1, -1, -1};
    for (const auto &f : msg.fields)
        for (int k = 0; k < 3; ++k)
            if (f.name == std::string(1, "xyz"[k]) && f.datatype == sensor_msgs::PointField::FLOAT32) off[k] = (int)f.offset;
    if (off[0] < 0 || off[0] + 4 != off[1] || off[1] + 4 != off[2] || off[0] % 4 || msg.point_step == 0 || msg.point_step % 4) return false;
    const uint8_t *base = msg.data.data() + off[0];
    if (reinterpret_cast<uintptr_t>(base) % 4) return false;
    xyz = reinterpret_cast<const float *>(base);
    stride = (int)(msg.point_step / 4);
    n = (size_t)msg.width * msg.height;
    return msg.data.size() >= n * msg.point_step;
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
