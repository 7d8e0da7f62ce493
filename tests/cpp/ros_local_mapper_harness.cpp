// Runs ros/local_mapper_node.cpp -- the node source as it is, main() included -- inside one process, on the in-process
// roscpp stand-in of tests/cpp/ros_stub/ros/ros.h: the harness plays ekf (/mapping/ekf/pose) and the Velodyne driver
// (/velodyne_points) and records the OccupancyGrid and cloud messages the node publishes (local_mapper.cpp:95-126).
//   ros_local_mapper_harness <dir> <out> <n_clouds>
// <dir>: cloud<k>.f32 (n x 3), poses.f64 (7 per cloud)
// <out>: doubles [n_grid_msgs, n_cloud_msgs, then per grid message: resolution width height origin.x origin.y frame_ok n_data],
//        <out>.occ: the int8 data of every grid message, one after the other; <out>.log: the event order
#define main local_mapper_node_main
#include "local_mapper_node.cpp"
#undef main

#include "ros_harness_util.hpp"

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const std::string dir = argv[1], out = argv[2];
    const int         n_clouds = std::atoi(argv[3]);
    const auto        poses = read_all<double>(dir + "/poses.f64");
    std::vector<std::vector<float>> clouds;
    for (int k = 0; k < n_clouds; ++k) clouds.push_back(read_all<float>(dir + "/cloud" + std::to_string(k) + ".f32"));
    std::vector<std::string> log;
    auto n_grids = [] { return ros::testing::published<nav_msgs::OccupancyGrid>("/mapping/local_drivability").size(); };
    auto note = [&](const std::string &what) { log.push_back(what + " | grids published so far: " + std::to_string(n_grids())); };
    int step = 0;
    // cloud k: pose stamped 100 + k, cloud stamped 100 + k.  local_mapper.cpp:102 maps a cloud only once a pose at least as
    // new as the cloud has arrived: for cloud 1 the pose is held back one round (the cloud must wait for it)
    ros::testing::master().idle = [&]() -> bool {
        using ros::testing::inject;
        const int k = step / 3, phase = step % 3;
        ++step;
        if (k >= n_clouds) {
            note("end");
            return false;
        }
        if (phase == 0) {
            note("cloud " + std::to_string(k) + (k == 1 ? " (its pose comes a round later)" : " with its pose"));
            if (k != 1) inject("/mapping/ekf/pose", pose_msg(&poses[7 * (size_t)k], 100 + (uint32_t)k, 0));
            inject("/velodyne_points", velodyne_cloud(clouds[(size_t)k], 100 + (uint32_t)k, 0, "/velodyne"));
        } else if (phase == 1) {
            note("round after cloud " + std::to_string(k));
            if (k == 1) inject("/mapping/ekf/pose", pose_msg(&poses[7 * (size_t)k], 100 + (uint32_t)k, 0));
        } else {
            note("quiet round");
        }
        return true;
    };
    const int rc = local_mapper_node_main(argc, argv);
    if (rc != 0) return 10 + rc;

    const auto grids = ros::testing::published<nav_msgs::OccupancyGrid>("/mapping/local_drivability");
    const auto cls = ros::testing::published<sensor_msgs::PointCloud2>("/mapping/local_poiuntcloud");
    std::vector<double> v = {(double)grids.size(), (double)cls.size()};
    std::vector<int8_t> occ;
    for (const auto &g : grids) {
        const double row[7] = {(double)g->info.resolution, (double)g->info.width, (double)g->info.height, g->info.origin.position.x,
                               g->info.origin.position.y, g->header.frame_id == "/local_oriented" ? 1.0 : 0.0, (double)g->data.size()};
        v.insert(v.end(), row, row + 7);
        occ.insert(occ.end(), g->data.begin(), g->data.end());
    }
    for (const auto &c : cls) {
        v.push_back((double)c->width * c->height);
        v.push_back(c->header.frame_id == "/local_oriented" && c->point_step == 12 ? 1.0 : 0.0);
    }
    FILE *f = std::fopen(out.c_str(), "wb");
    if (!f) return 2;
    std::fwrite(v.data(), 8, v.size(), f);
    std::fclose(f);
    f = std::fopen((out + ".occ").c_str(), "wb");
    if (!f) return 2;
    std::fwrite(occ.data(), 1, occ.size(), f);
    std::fclose(f);
    f = std::fopen((out + ".log").c_str(), "w");
    if (!f) return 2;
    for (const auto &l : log) std::fprintf(f, "%s\n", l.c_str());
    std::fclose(f);
    return 0;
}
