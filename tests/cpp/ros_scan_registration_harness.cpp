// Runs ros/scan_registration_node.cpp -- the node source as it is, main() included -- inside one process, on the
// in-process roscpp stand-in of tests/cpp/ros_stub/ros/ros.h: the harness plays the other nodes of
// nasa_mapping.launch (ekf: /mapping/ekf/pose; graph_slam: the two target clouds; the Velodyne driver: /velodyne_points)
// and records what the node publishes (scan_registration.cpp:109-199).
//   ros_scan_registration_harness <dir> <out>
// <dir>: target.f32 (obstacle cloud of the map, n x 3), target_ground.f32, scene.f32 (the scan, n x 3), init.f64 (x y z qx qy qz qw)
// <out>: doubles -- [n_pose_msgs, n_scene_msgs, n_warnings, n_errors, per pose message: x y z qx qy qz qw stamp.sec stamp.nsec frame_is_global]
//        then <out>.log: one line per event (what arrived when), for the test's assertions on ORDER
#define main scan_registration_node_main
#include "scan_registration_node.cpp"
#undef main

#include "ros_harness_util.hpp"

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::string dir = argv[1], out = argv[2];
    const auto target = read_all<float>(dir + "/target.f32"), gnd = read_all<float>(dir + "/target_ground.f32");
    const auto scene = read_all<float>(dir + "/scene.f32");
    const auto init = read_all<double>(dir + "/init.f64");
    std::vector<std::string> log;
    auto n_poses = [] { return ros::testing::published<geometry_msgs::PoseStamped>("mapping/scan_reg/pose").size(); };
    auto note = [&](const std::string &what) {
        log.push_back(what + " | poses published so far: " + std::to_string(n_poses()) + ", warnings " +
                      std::to_string(ros::testing::master().warnings.size()) + ", errors " + std::to_string(ros::testing::master().errors.size()));
    };
    int step = 0;
    ros::testing::master().idle = [&]() -> bool {
        using ros::testing::inject;
        switch (step++) {
        case 0: // a scan before any target: ignored (scan_registration.cpp:114-115 first_gnd && first_obs)
            note("scan before the targets");
            inject("/mapping/ekf/pose", pose_msg(init.data(), 10, 0));
            inject("/velodyne_points", velodyne_cloud(scene, 10, 100, "/velodyne"));
            return true;
        case 1: // graph_slam's target clouds, global frame
            note("targets");
            inject("/mapping/global/obstacle_pointcloud", velodyne_cloud(target, 11, 0, "/global"));
            inject("/mapping/global/ground_pointcloud", velodyne_cloud(gnd, 11, 0, "/global"));
            return true;
        case 2: { // a scan of fewer than 20 000 points: dropped with a warning (:122-125)
            note("small scan");
            std::vector<float> few(scene.begin(), scene.begin() + 3 * 5000);
            inject("/velodyne_points", velodyne_cloud(few, 12, 0, "/velodyne"));
            return true;
        }
        case 3: // the scan
            note("scan");
            inject("/velodyne_points", velodyne_cloud(scene, 13, 250, "/velodyne"));
            return true;
        case 4: { // a scan with nothing to match -- 30 000 returns from flat ground only: doICPMatch answers orientation.w = 9999
                  // (icpTools.cpp:179-184), the node logs an error and publishes nothing (:161-165)
            note("ground-only scan");
            std::vector<float> flat;
            for (int i = 0; i < 30000; ++i) {
                const double r = 2.0 + 0.0005 * i, a = 0.0021 * i;
                flat.push_back((float)(r * std::cos(a))), flat.push_back((float)(r * std::sin(a))), flat.push_back(-1.73f);
            }
            inject("/velodyne_points", velodyne_cloud(flat, 14, 0, "/velodyne"));
            return true;
        }
        case 5: // the EKF's pose again (the node keeps its own result as the next initial pose otherwise, :167-172), the scan again
            note("pose + scan again");
            inject("/mapping/ekf/pose", pose_msg(init.data(), 15, 0));
            inject("/velodyne_points", velodyne_cloud(scene, 15, 500, "/velodyne"));
            return true;
        default:
            note("end");
            return false;
        }
    };
    const int rc = scan_registration_node_main(argc, argv);
    if (rc != 0) return 10 + rc;

    const auto poses = ros::testing::published<geometry_msgs::PoseStamped>("mapping/scan_reg/pose");
    const auto scenes = ros::testing::published<sensor_msgs::PointCloud2>("mapping/scan_reg/scene");
    std::vector<double> v = {(double)poses.size(), (double)scenes.size(), (double)ros::testing::master().warnings.size(),
                             (double)ros::testing::master().errors.size()};
    for (const auto &p : poses) {
        const double row[10] = {p->pose.position.x, p->pose.position.y, p->pose.position.z, p->pose.orientation.x, p->pose.orientation.y,
                                p->pose.orientation.z, p->pose.orientation.w, (double)p->header.stamp.sec, (double)p->header.stamp.nsec,
                                p->header.frame_id == "/global" ? 1.0 : 0.0};
        v.insert(v.end(), row, row + 10);
    }
    for (const auto &c : scenes) { // the debug cloud of the segmented scene (:141-148): its size and frame
        v.push_back((double)c->width * c->height);
        v.push_back(c->header.frame_id == "/local" ? 1.0 : 0.0);
    }
    FILE *f = std::fopen(out.c_str(), "wb");
    if (!f) return 2;
    std::fwrite(v.data(), 8, v.size(), f);
    std::fclose(f);
    f = std::fopen((out + ".log").c_str(), "w");
    if (!f) return 2;
    for (const auto &l : log) std::fprintf(f, "%s\n", l.c_str());
    for (const auto &w : ros::testing::master().warnings) std::fprintf(f, "WARN: %s\n", w.c_str());
    for (const auto &e : ros::testing::master().errors) std::fprintf(f, "ERROR: %s\n", e.c_str());
    std::fclose(f);
    return 0;
}
