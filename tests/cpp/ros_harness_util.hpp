// Shared by the two node harnesses (tests/cpp/ros_*_harness.cpp): files in, sensor_msgs/PointCloud2 out.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <geometry_msgs/PoseStamped.h>
#include <sensor_msgs/PointCloud2.h>

#include "cloud_msg.hpp"

template <class T>
static std::vector<T> read_all(const std::string &path)
{
    std::vector<T> v;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(2); }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / sizeof(T));
    if (n && std::fread(v.data(), 1, (size_t)n, f) != (size_t)n) std::exit(2);
    std::fclose(f);
    return v;
}

// a cloud message as the Velodyne driver sends it: x, y, z, intensity (FLOAT32) and ring (UINT16), 32-byte points --
// the nodes must find x, y, z by name and ignore the rest
static sensor_msgs::PointCloud2 velodyne_cloud(const std::vector<float> &xyz, uint32_t sec, uint32_t nsec, const char *frame)
{
    const size_t             n = xyz.size() / 3;
    sensor_msgs::PointCloud2 m;
    m.header.stamp.sec = sec, m.header.stamp.nsec = nsec;
    m.header.frame_id = frame;
    m.height = 1, m.width = (uint32_t)n;
    const char    *names[5] = {"x", "y", "z", "intensity", "ring"};
    const uint32_t offs[5] = {0, 4, 8, 16, 20};
    for (int k = 0; k < 5; ++k) {
        sensor_msgs::PointField f;
        f.name = names[k], f.offset = offs[k], f.count = 1;
        f.datatype = k < 4 ? sensor_msgs::PointField::FLOAT32 : sensor_msgs::PointField::UINT16;
        m.fields.push_back(f);
    }
    m.point_step = 32, m.row_step = 32u * (uint32_t)n;
    m.data.assign(32 * n, 0xAB); // whatever lies between the fields is not ours to read
    for (size_t i = 0; i < n; ++i) std::memcpy(m.data.data() + 32 * i, &xyz[3 * i], 12);
    return m;
}

static geometry_msgs::PoseStamped pose_msg(const double p[7], uint32_t sec, uint32_t nsec)
{
    geometry_msgs::PoseStamped m;
    m.header.stamp.sec = sec, m.header.stamp.nsec = nsec;
    m.header.frame_id = "/global";
    m.pose.position.x = p[0], m.pose.position.y = p[1], m.pose.position.z = p[2];
    m.pose.orientation.x = p[3], m.pose.orientation.y = p[4], m.pose.orientation.z = p[5], m.pose.orientation.w = p[6];
    return m;
}
