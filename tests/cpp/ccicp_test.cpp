// A C++ program written like scan_registration.cpp:57,82-98,139-159 uses CCICP, over the adapter
// include/slam_amd/ccicp.hpp: target cloud and scene cloud in, pose out.
//   ccicp_test <dir> <out> <type 0|1>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "slam_amd/ccicp.hpp"

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
    const int         type = std::atoi(argv[3]);
    auto target = read_all<float>(dir + "/target.f32"), scene = read_all<float>(dir + "/scene.f32");
    auto gnd = read_all<float>(dir + "/target_ground.f32");
    auto init = read_all<double>(dir + "/init.f64"); // x y z qx qy qz qw

    slam_amd::CCICP icp(type ? slam_amd::SCAN_TO_MAP : slam_amd::SCAN_TO_SCAN); // scan_registration.cpp:57
    slam_amd::Pose  pose;
    pose.x = init[0], pose.y = init[1], pose.z = init[2];
    pose.qx = init[3], pose.qy = init[4], pose.qz = init[5], pose.qw = init[6];
    icp.setTargetCloud(target.data(), (int)target.size() / 3, 3, pose);       // :82, :97
    if (type) icp.setTargetGndCloud(gnd.data(), (int)gnd.size() / 3, 3);       // :83, :98
    icp.setSceneCloud(scene.data(), (int)scene.size() / 3, 3);                 // :139
    const slam_amd::Pose r = icp.doICPMatch(pose);                             // :159

    const double v[17] = {r.x, r.y, r.z, r.qx, r.qy, r.qz, r.qw, (double)icp.getNumberCorrespondences(),
                          (double)icp.targetSize(), (double)icp.sceneSize(), (double)icp.groundTargetSize(),
                          (double)icp.groundSceneSize(), (double)icp.modelCounts()[0], (double)icp.modelCounts()[1],
                          (double)icp.sceneCounts()[0], (double)icp.sceneCounts()[1], (double)icp.stepwiseMatches()};
    FILE *f = std::fopen(out.c_str(), "wb");
    if (!f) return 2;
    std::fwrite(v, 8, 17, f);
    std::fclose(f);
    // getSegmentedClouds (icpTools.cpp:644-650, scan_registration.cpp:142): four clouds of the sizes reported above
    std::vector<float> c_target, c_scene, c_gt, c_gs;
    icp.getSegmentedClouds(c_target, c_scene, c_gt, c_gs);
    if ((int)c_target.size() != 3 * icp.targetSize() || (int)c_scene.size() != 3 * icp.sceneSize() ||
        (int)c_gt.size() != 3 * icp.groundTargetSize() || (int)c_gs.size() != 3 * icp.groundSceneSize())
        return 4;
    f = std::fopen((out + ".scene").c_str(), "wb");
    if (!f) return 2;
    std::fwrite(c_scene.data(), 4, c_scene.size(), f);
    std::fclose(f);
    // scene first, then the target (icpTools.cpp:585-608 leaves seg_scene alone): the same match must come out
    {
        slam_amd::CCICP other(type ? slam_amd::SCAN_TO_MAP : slam_amd::SCAN_TO_SCAN);
        other.setSceneCloud(scene.data(), (int)scene.size() / 3, 3);
        other.setTargetCloud(target.data(), (int)target.size() / 3, 3, pose);
        if (type) other.setTargetGndCloud(gnd.data(), (int)gnd.size() / 3, 3);
        const slam_amd::Pose q = other.doICPMatch(pose);
        if (q.qw == 9999 || q.x != r.x || q.y != r.y || q.z != r.z || q.qz != r.qz || q.qw != r.qw ||
            other.sceneSize() != icp.sceneSize() || other.getNumberCorrespondences() != icp.getNumberCorrespondences()) {
            std::fprintf(stderr, "scene -> target -> match: %.9f %.9f %.9f (w %.9f), target -> scene -> match: %.9f %.9f %.9f\n", q.x, q.y, q.z, q.qw,
                         r.x, r.y, r.z);
            return 5;
        }
    }
    // a scene too small to match: the sentinel of icpTools.cpp:179-184
    icp.setSceneCloud(scene.data(), 3, 3);
    const slam_amd::Pose bad = icp.doICPMatch(pose);
    return bad.qw == 9999 && icp.getResidual() == -1 ? 0 : 3;
}
