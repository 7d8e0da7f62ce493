// The throughput forms of slam_amd::CCICP (round 5) against the sequential form, on the edges: a prepared scene that is NOT the one
// set next, read-outs after an adopted scene, a sequence longer than one batch, a scene of fewer than 5 points inside a batch, a pose
// whose crop window (icpTools.cpp:225-239) leaves nothing of the target in the middle of a sequence.  Self-checking:
//   ccicp_forms_test <dir> <n_clouds>      (<dir>/cloud<k>.f32, <dir>/init.f64 as for ccicp_sequence; the target is cloud 0)
// prints what it compared; exit code 0 = every comparison held.
#include <cmath>
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

static int    g_bad = 0;
static double diff(const slam_amd::Pose &a, const slam_amd::Pose &b)
{
    double d = 0;
    for (double x : {a.x - b.x, a.y - b.y, a.z - b.z, a.qx - b.qx, a.qy - b.qy, a.qz - b.qz, a.qw - b.qw}) d = std::fmax(d, std::fabs(x));
    return d;
}
static void check(bool ok, const char *what, double v = 0)
{
    std::printf("%s %s (%.3g)\n", ok ? "ok  " : "BAD ", what, v);
    if (!ok) ++g_bad;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::string dir = argv[1];
    const int         n = std::atoi(argv[2]);
    std::vector<std::vector<float>> clouds;
    for (int k = 0; k < n; ++k) clouds.push_back(read_all<float>(dir + "/cloud" + std::to_string(k) + ".f32"));
    const auto init = read_all<double>(dir + "/init.f64");
    std::vector<slam_amd::Pose> ip;
    for (int k = 1; k < n; ++k) {
        const double  *q = &init[7 * (size_t)(k - 1)];
        slam_amd::Pose p;
        p.x = q[0], p.y = q[1], p.z = q[2], p.qx = q[3], p.qy = q[4], p.qz = q[5], p.qw = q[6];
        ip.push_back(p);
    }
    std::vector<float> tiny(clouds[1].begin(), clouds[1].begin() + 12); // 4 points: "Total Scene has N points" (:179-184)
    std::vector<const float *> ptr;
    std::vector<int>           cnt;
    for (int k = 1; k < n; ++k) ptr.push_back(clouds[k].data()), cnt.push_back((int)clouds[k].size() / 3);
    const int i_tiny = 5, i_far = n - 4; // (of the n - 1 scenes; n = 24: batches 0-15, 16-19, 20-22)
    ptr[i_tiny] = tiny.data(), cnt[i_tiny] = 4;
    std::vector<slam_amd::Pose> ip2 = ip;
    ip2[i_far].x += 1000.0; // its window holds nothing of the target; the box stays empty for every match after it (:226-239)
    slam_amd::Pose p0;

    // (1) the reference's usage, cloud by cloud
    std::vector<slam_amd::Pose> seq;
    std::vector<int>            seq_scene, seq_gscene;
    {
        slam_amd::CCICP icp(slam_amd::SCAN_TO_SCAN);
        icp.setTargetCloud(clouds[0].data(), (int)clouds[0].size() / 3, 3, p0);
        for (int k = 0; k < n - 1; ++k) {
            icp.setSceneCloud(ptr[k], cnt[k], 3);
            seq.push_back(icp.doICPMatch(ip2[k]));
            seq_scene.push_back(icp.sceneSize());
            seq_gscene.push_back(icp.groundSceneSize());
        }
        check(seq[i_tiny].qw == 9999, "sequential: a 4-point scene returns orientation.w == 9999");
    }
    // (1b) setSceneCloud(cloud, R, t): the roll / pitch compensation of scan_registration.cpp:128-139 on the device -- the poses of the
    // same clouds turned by a host loop first (what ros/scan_registration_node.cpp did until round 6), bit for bit
    {
        const double roll = 0.03, pitch = -0.02, tz = 0.4;
        const double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch);
        const double R[9] = {cp, sp * sr, sp * cr, 0.0, cr, -sr, -sp, cp * sr, cp * cr}, t3[3] = {0.0, 0.0, tz};
        slam_amd::CCICP a(slam_amd::SCAN_TO_SCAN), b(slam_amd::SCAN_TO_SCAN);
        a.setTargetCloud(clouds[0].data(), (int)clouds[0].size() / 3, 3, p0);
        b.setTargetCloud(clouds[0].data(), (int)clouds[0].size() / 3, 3, p0);
        double worst = 0;
        for (int k = 0; k < 4; ++k) {
            const size_t       m = (size_t)cnt[k];
            std::vector<float> turned(3 * m);
            for (size_t i = 0; i < m; ++i) {
                const double px = ptr[k][3 * i], py = ptr[k][3 * i + 1], pz = ptr[k][3 * i + 2];
                turned[3 * i] = (float)(R[0] * px + R[1] * py + R[2] * pz + t3[0]);
                turned[3 * i + 1] = (float)(R[3] * px + R[4] * py + R[5] * pz + t3[1]);
                turned[3 * i + 2] = (float)(R[6] * px + R[7] * py + R[8] * pz + t3[2]);
            }
            a.setSceneCloud(turned.data(), cnt[k], 3);
            b.setSceneCloud(ptr[k], cnt[k], 3, R, t3);
            worst = std::fmax(worst, diff(a.doICPMatch(ip[(size_t)k]), b.doICPMatch(ip[(size_t)k])));
        }
        check(worst == 0.0, "setSceneCloud(cloud, R, t): the poses of the clouds turned on the host, bit for bit", worst);
    }
    // (2) two chains in flight: a prepared scene that is adopted, one that is not, read-outs behind an adopted scene
    {
        slam_amd::CCICP icp(slam_amd::SCAN_TO_SCAN);
        icp.setTargetCloud(clouds[0].data(), (int)clouds[0].size() / 3, 3, p0);
        double worst = 0;
        bool   sizes = true;
        icp.prepareSceneCloud(ptr[0], cnt[0], 3);
        for (int k = 0; k < n - 1; ++k) {
            icp.setSceneCloud(ptr[k], cnt[k], 3);
            if (k + 1 < n - 1) {
                // every third time the WRONG cloud is prepared: the next setSceneCloud must make its own scene from scratch
                const int j = (k % 3 == 2 && k + 2 < n - 1) ? k + 2 : k + 1;
                icp.prepareSceneCloud(ptr[j], cnt[j], 3);
            }
            const slam_amd::Pose r = icp.doICPMatch(ip2[k]);
            worst = std::fmax(worst, diff(r, seq[(size_t)k]));
            sizes = sizes && icp.sceneSize() == seq_scene[(size_t)k] && icp.groundSceneSize() == seq_gscene[(size_t)k];
        }
        check(worst == 0.0, "prepareSceneCloud (right and wrong clouds prepared): poses bit-identical to the sequential form", worst);
        check(sizes, "sceneSize / groundSceneSize behind adopted scenes equal the sequential form's");
    }
    // (3) the sequence form: two batches (n - 1 > 16), the 4-point scene and the far pose inside them
    {
        slam_amd::CCICP icp(slam_amd::SCAN_TO_SCAN);
        icp.setTargetCloud(clouds[0].data(), (int)clouds[0].size() / 3, 3, p0);
        const std::vector<slam_amd::Pose> r = icp.matchSequence(ptr.data(), cnt.data(), n - 1, 3, ip2.data());
        double worst = 0;
        for (int k = 0; k < n - 1; ++k)
            if (k != i_tiny) worst = std::fmax(worst, diff(r[(size_t)k], seq[(size_t)k]));
        check(r[(size_t)i_tiny].qw == 9999, "matchSequence: the 4-point scene returns orientation.w == 9999 in its place");
        check(worst < 1e-9, "matchSequence: every other pose equals the sequential form's", worst);
        check(icp.sequenceBatches() >= 3, "matchSequence: more than one batch (16 scenes at most; the far pose's window ends one)", icp.sequenceBatches());
    }
    // (4) scenes of DIFFERENT sizes, twice over: a lane's buffers grow (its captured chains are made for other pointers then), a slot
    // sees a size it has not seen (call by call), then the same size again (captured), then replays
    {
        std::vector<int> cut;
        for (int k = 0; k < n - 1; ++k) cut.push_back(k == i_tiny ? 4 : (int)((cnt[k] / 2) + ((long)cnt[k] / 2) * ((k * 7919) % 97) / 97));
        // five passes over the same sizes (a slot's chain: call by call, captured, replayed, replayed) -- but in pass 3 scene 6 is the
        // WHOLE cloud, larger than anything its lane has seen: the lane's handles re-allocate their scratch, and every replay captured
        // on that lane before must be made again, not launched
        constexpr int kReps = 5;
        const auto cut_of = [&](int rep, int k) { return (rep == 3 && k == 6) ? cnt[k] : cut[(size_t)k]; };
        std::vector<slam_amd::Pose> want;
        {
            slam_amd::CCICP icp(slam_amd::SCAN_TO_SCAN);
            icp.setTargetCloud(clouds[0].data(), (int)clouds[0].size() / 3, 3, p0);
            for (int rep = 0; rep < kReps; ++rep)
                for (int k = 0; k < n - 1; ++k) {
                    icp.setSceneCloud(ptr[k], cut_of(rep, k), 3);
                    want.push_back(icp.doICPMatch(ip[k]));
                }
        }
        slam_amd::CCICP icp(slam_amd::SCAN_TO_SCAN);
        icp.setSequenceGraphs(true); // (opt-in since round 6: this section is about the captured chains)
        icp.setTargetCloud(clouds[0].data(), (int)clouds[0].size() / 3, 3, p0);
        double worst = 0, worst_ahead = 0;
        for (int rep = 0; rep < kReps; ++rep) {
            std::vector<int> c2;
            for (int k = 0; k < n - 1; ++k) c2.push_back(cut_of(rep, k));
            const std::vector<slam_amd::Pose> r = icp.matchSequence(ptr.data(), c2.data(), n - 1, 3, ip.data());
            for (int k = 0; k < n - 1; ++k)
                if (k != i_tiny) worst = std::fmax(worst, diff(r[(size_t)k], want[(size_t)(rep * (n - 1) + k)]));
        }
        // ... and the same object one cloud at a time afterwards, with the next one prepared: the forms share the object's state
        for (int k = 0; k < n - 1; ++k) {
            icp.setSceneCloud(ptr[k], cut[k], 3);
            if (k + 1 < n - 1) icp.prepareSceneCloud(ptr[k + 1], cut[k + 1], 3);
            const slam_amd::Pose r = icp.doICPMatch(ip[k]);
            if (k != i_tiny) worst_ahead = std::fmax(worst_ahead, diff(r, want[(size_t)k]));
        }
        check(worst < 1e-9, "matchSequence over scenes of different sizes, five passes (captured, replayed, a lane's scratch re-allocated in between): poses equal the sequential form's", worst);
        check(worst_ahead == 0.0, "the same object cloud by cloud afterwards, next cloud prepared: bit-identical to the sequential form", worst_ahead);
    }
    std::printf("%s\n", g_bad ? "FAILED" : "OK");
    return g_bad ? 5 : 0;
}
