// A C++ program written like local_mapper.cpp:65-130 uses MLS, over the adapter include/slam_amd/mls.hpp:
// raw clouds and poses in (addToMap segments inside, mls.cpp:34-150), occupancy grid and global cloud out.
//   mls_cloud_test <dir> <out> <n_clouds>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "slam_amd/mls.hpp"

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

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const std::string dir = argv[1], out = argv[2];
    const int         n_clouds = std::atoi(argv[3]);
    auto poses = read_all<double>(dir + "/poses.f64"); // x y z qx qy qz qw per cloud

    // timing (tools/mls_time.py): <passes> > 0 runs the clouds that often through a second map first -- addToMap per cloud as
    // local_mapper's callback does (local_mapper.cpp:107), getDrivability every <every> clouds (:120) -- and prints the wall clock
    const int passes = argc > 4 ? std::atoi(argv[4]) : 0, every = argc > 5 ? std::atoi(argv[5]) : 5;
    if (passes > 0) {
        std::vector<std::vector<float>> clouds;
        for (int k = 0; k < n_clouds; ++k) clouds.push_back(read_all<float>(dir + "/cloud" + std::to_string(k) + ".f32"));
        const bool nocloud = argc > 6 && std::atoi(argv[6]);
        slam_amd::MLS m(200, 200, 0.2, true);
        m.setMinClusterPoints(20);
        m.clearMap();
        if (nocloud) m.setDisablePointCloud(true); // mls.h:223
        double t_add = 0, t_drv = 0, t_flt = 0;
        long   n_add = 0, n_drv = 0;
        size_t n_global = 0;
        for (int pass = 0; pass < passes; ++pass) {
            if (pass == 1) t_add = t_drv = t_flt = 0, n_add = n_drv = 0; // (the first pass makes the buffers)
            for (int k = 0; k < n_clouds; ++k) {
                slam_amd::Pose p;
                const double *q = &poses[7 * (size_t)k];
                p.x = q[0], p.y = q[1], p.z = q[2], p.qx = q[3], p.qy = q[4], p.qz = q[5], p.qw = q[6];
                auto a = std::chrono::steady_clock::now();
                m.addToMap(clouds[(size_t)k].data(), (int)clouds[(size_t)k].size() / 3, 3, p);
                auto b = std::chrono::steady_clock::now();
                t_add += std::chrono::duration<double, std::milli>(b - a).count();
                ++n_add;
                if (!nocloud) { // local_mapper.cpp:111: the global cloud filtered behind every cloud ("so it won't kill rviz")
                    auto c = std::chrono::steady_clock::now();
                    m.filterPointCloud(0.1, 0.1);
                    t_flt += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c).count();
                    n_global = m.getGlobalCloud().size() / 3;
                    b = std::chrono::steady_clock::now();
                }
                if ((k + 1) % every == 0) {
                    (void)m.getDrivability();
                    t_drv += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b).count();
                    ++n_drv;
                }
            }
            if (!nocloud) m.clearMap();
        }
        std::printf("{\"clouds\": %ld, \"ms_per_add_to_map\": %.4f, \"ms_per_get_drivability\": %.4f, \"ms_per_filter_point_cloud\": %.4f, "
                    "\"global_cloud_points\": %zu, \"points_per_cloud\": %zu}\n", n_add,
                    t_add / (double)std::max(n_add, 1l), t_drv / (double)std::max(n_drv, 1l), t_flt / (double)std::max(n_add, 1l), n_global,
                    clouds[0].size() / 3);
    }
    slam_amd::MLS local_map(200, 200, 0.2, true); // local_mapper.cpp:29
    local_map.setMinClusterPoints(20);             // :86
    local_map.clearMap();                          // :93
    for (int k = 0; k < n_clouds; ++k) {
        auto cloud = read_all<float>(dir + "/cloud" + std::to_string(k) + ".f32");
        slam_amd::Pose p;
        const double *q = &poses[7 * (size_t)k];
        p.x = q[0], p.y = q[1], p.z = q[2], p.qx = q[3], p.qy = q[4], p.qz = q[5], p.qw = q[6];
        local_map.addToMap(cloud.data(), (int)cloud.size() / 3, 3, p); // :107
    }
    const size_t n_global = local_map.getGlobalCloud().size() / 3;
    slam_amd::Pose off;
    off.z = 0.25;
    local_map.offsetMap(off);               // local_mapper.cpp:48-51
    local_map.filterPointCloud(0.1, 0.1);   // :111
    const std::vector<float>       &gc = local_map.getGlobalCloud(); // :112
    const slam_amd::OccupancyGrid &g = local_map.getDrivability();   // :120
    double px = 0, py = 0;
    slam_grid_get_pose(local_map.handle(), &px, &py);
    const double head[8] = {px, py, (double)local_map.lastSegmentCounts()[0], (double)local_map.lastSegmentCounts()[1],
                            (double)n_global, (double)g.info.width, g.info.resolution, 0.0};
    FILE *f = std::fopen(out.c_str(), "wb");
    if (!f) return 2;
    std::fwrite(head, 8, 8, f);
    std::fwrite(g.data.data(), 1, g.data.size(), f);
    std::fwrite(gc.data(), 4, gc.size(), f);
    std::fclose(f);
    return 0;
}
