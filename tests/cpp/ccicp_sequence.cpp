// BASELINE config 3 as scan_registration runs it (scan_registration.cpp:57,73-104,109-173), over the adapter
// include/slam_amd/ccicp.hpp: a sequence of 64-ring clouds, each registered against the current target with
// setSceneCloud + doICPMatch; every `advance` clouds the cloud just matched becomes the new target (setTargetCloud,
// SCAN_TO_SCAN: what the target callbacks :73-104 do when graph_slam publishes a new map).
//   ccicp_sequence <dir> <n_clouds> <advance> [passes] [form]
// form (round 5): "seq" (default) = one cloud at a time as above; "seqp" = the same on clouds in pinned memory (round 6: the upload only enqueues); "ahead" = the same calls with prepareSceneCloud(cloud k+1) before
// doICPMatch(cloud k) -- two chains in flight; "batch" = CCICP::matchSequence over the clouds between two target replacements
// (the initial poses are the file's either way: in the node they would be the previous results).  "ahead" and "batch" keep the
// clouds in pinned memory (slam_host_alloc), so that their uploads do not hold the host.
// <dir>/cloud<k>.f32 (x y z per point), <dir>/init.f64 (per match k = 1..n-1: x y z qx qy qz qw of the initial pose of
// cloud k in the frame of ITS target), <dir>/truth.f64 (x y yaw of the same).  Prints one JSON line; the last pass is
// the one reported (the first warms buffers and code objects).
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstring>
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
    const std::string dir = argv[1];
    const int n_clouds = std::atoi(argv[2]), advance = std::atoi(argv[3]), passes = argc > 4 ? std::atoi(argv[4]) : 2;
    const std::string form = argc > 5 ? argv[5] : "seq";
    std::vector<std::vector<float>> clouds;
    for (int k = 0; k < n_clouds; ++k) clouds.push_back(read_all<float>(dir + "/cloud" + std::to_string(k) + ".f32"));
    std::vector<const float *> cloud_ptr;
    std::vector<int>           cloud_n;
    for (int k = 0; k < n_clouds; ++k) {
        const float *p = clouds[k].data();
        if (form != "seq") { // pinned copies
            void *pin = nullptr;
            if (slam_host_alloc(&pin, clouds[k].size() * sizeof(float) + 16) != SLAM_OK) return 4;
            std::memcpy(pin, clouds[k].data(), clouds[k].size() * sizeof(float));
            p = static_cast<const float *>(pin);
        }
        cloud_ptr.push_back(p);
        cloud_n.push_back((int)clouds[k].size() / 3);
    }
    const auto init = read_all<double>(dir + "/init.f64"), truth = read_all<double>(dir + "/truth.f64");
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };

    slam_amd::CCICP icp(slam_amd::SCAN_TO_SCAN);
    if (std::getenv("SEQ_GRAPHS")) icp.setSequenceGraphs(true); // (the batch form's scene chains replayed as hipGraphs: off by default since round 6)
    double t_match = 0, t_target = 0, worst = 0, sum_err = 0, iters = 0, corr = 0, t_set = 0; // t_set: of t_match, the host inside setSceneCloud
    int    n_match = 0, n_target = 0, builds0 = 0;
    std::vector<double> poses;
    int    seq_b0 = 0;
    double seq_t0[3] = {0, 0, 0};
    for (int pass = 0; pass < passes; ++pass) {
        t_match = t_target = worst = sum_err = iters = corr = t_set = 0;
        seq_b0 = icp.sequenceBatches();
        for (int j = 0; j < 3; ++j) seq_t0[j] = icp.sequenceTimes()[j];
        n_match = n_target = 0;
        poses.clear();
        builds0 = icp.targetBuilds();
        slam_amd::Pose p0;
        auto a = clk::now();
        icp.setTargetCloud(clouds[0].data(), (int)clouds[0].size() / 3, 3, p0);
        t_target += ms(a, clk::now());
        ++n_target;
        auto pose_of = [&](int k) {
            const double  *q = &init[7 * (size_t)(k - 1)];
            slam_amd::Pose pose;
            pose.x = q[0], pose.y = q[1], pose.z = q[2], pose.qx = q[3], pose.qy = q[4], pose.qz = q[5], pose.qw = q[6];
            return pose;
        };
        auto account = [&](int k, const slam_amd::Pose &r, int it, int nc) {
            const double *tr = &truth[3 * (size_t)(k - 1)];
            const double  err = std::hypot(r.x - tr[0], r.y - tr[1]);
            worst = err > worst ? err : worst;
            sum_err += err;
            iters += it;
            corr += nc;
            poses.insert(poses.end(), {r.x, r.y, r.z, r.qx, r.qy, r.qz, r.qw});
        };
        if (form == "batch") {
            int k = 1;
            while (k < n_clouds) {
                // the matches up to (and including) the next target replacement
                int last = advance > 0 ? std::min(((k - 1) / advance + 1) * advance, n_clouds - 1) : n_clouds - 1;
                std::vector<slam_amd::Pose> ip;
                for (int j = k; j <= last; ++j) ip.push_back(pose_of(j));
                a = clk::now();
                const std::vector<slam_amd::Pose> r = icp.matchSequence(cloud_ptr.data() + k, cloud_n.data() + k, last - k + 1, 3, ip.data());
                t_match += ms(a, clk::now());
                n_match += last - k + 1;
                for (int j = k; j <= last; ++j) {
                    if (r[(size_t)(j - k)].qw == 9999) { std::fprintf(stderr, "match %d: scene too small\n", j); return 3; }
                    account(j, r[(size_t)(j - k)], 0, 0);
                }
                if (advance > 0 && last % advance == 0 && last + 1 < n_clouds) {
                    a = clk::now();
                    icp.setTargetCloud(cloud_ptr[last], cloud_n[last], 3, r.back());
                    t_target += ms(a, clk::now());
                    ++n_target;
                }
                k = last + 1;
            }
        } else
        for (int k = 1; k < n_clouds; ++k) {
            const slam_amd::Pose pose = pose_of(k);
            a = clk::now();
            icp.setSceneCloud(cloud_ptr[k], cloud_n[k], 3); // scan_registration.cpp:139
            t_set += ms(a, clk::now());
            if (form == "ahead" && k + 1 < n_clouds) icp.prepareSceneCloud(cloud_ptr[k + 1], cloud_n[k + 1], 3);
            const slam_amd::Pose r = icp.doICPMatch(pose);                       // :159
            t_match += ms(a, clk::now());
            ++n_match;
            if (r.qw == 9999) { std::fprintf(stderr, "match %d: scene too small\n", k); return 3; }
            account(k, r, icp.lastIterations(), icp.getNumberCorrespondences());
            if (advance > 0 && k % advance == 0 && k + 1 < n_clouds) { // the map moved on: this cloud is the target from here
                a = clk::now();
                icp.setTargetCloud(cloud_ptr[k], cloud_n[k], 3, r);
                t_target += ms(a, clk::now());
                ++n_target;
            }
        }
    }
    if (form == "batch" && icp.sequenceBatches() > seq_b0) { // the last pass's batches
        const int nb = icp.sequenceBatches() - seq_b0;
        std::fprintf(stderr, "matchSequence: %d batches; host ms per batch: scene chains enqueued %.3f, all enqueued %.3f, results back %.3f (uploads, all passes: %.3f ms)\n", nb,
                     (icp.sequenceTimes()[0] - seq_t0[0]) / nb, (icp.sequenceTimes()[1] - seq_t0[1]) / nb, (icp.sequenceTimes()[2] - seq_t0[2]) / nb, icp.sequenceUploadMs());
    }
    if (FILE *f = std::fopen((dir + "/poses_out.f64").c_str(), "wb")) {
        std::fwrite(poses.data(), 8, poses.size(), f);
        std::fclose(f);
    }
    std::printf("{\"form\": \"%s\", \"matches\": %d, \"ms_per_match\": %.4f, \"clouds_per_s\": %.1f, \"target_updates\": %d, \"ms_per_target_update\": %.4f, "
                "\"target_index_builds\": %d, \"ms_per_cloud_with_target_updates\": %.4f, \"mean_icp_iterations\": %.2f, "
                "\"mean_correspondences\": %.1f, \"mean_xy_error_m\": %.4f, \"max_xy_error_m\": %.4f, \"rays_per_cloud\": %zu, "
                "\"ms_in_set_scene_cloud\": %.4f}\n",
                form.c_str(), n_match, t_match / n_match, 1e3 * n_match / t_match, n_target, t_target / n_target, icp.targetBuilds() - builds0,
                (t_match + t_target) / n_match, iters / n_match, corr / n_match, sum_err / n_match, worst, clouds[0].size() / 3,
                n_match ? t_set / n_match : 0.0);
    return 0;
}
