// Drives the reference-shaped C++ adapters (include/slam_amd/*.hpp) the way
// ccicp2d/src/icpTools.cpp:168-197 and local_mapper/src/local_mapper.cpp:29,86,107
// drive the reference classes; reads inputs from and writes results to plain
// binary files so that tests/test_gpu_cpp_adapters.py can compare with the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "slam_amd/icp.hpp"
#include "slam_amd/mls.hpp"

template <class T>
static std::vector<T> read_all(const char *path)
{
    std::vector<T> v;
    FILE *f = std::fopen(path, "rb");
    if (!f) { std::perror(path); std::exit(2); }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / sizeof(T));
    if (n && std::fread(v.data(), 1, (size_t)n, f) != (size_t)n) std::exit(2);
    std::fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const std::string dir = argv[1], out = argv[2];
    auto m_ga = read_all<double>((dir + "/m_ga.f64").c_str());
    auto m_nga = read_all<double>((dir + "/m_nga.f64").c_str());
    auto t_ga = read_all<double>((dir + "/t_ga.f64").c_str());
    auto t_nga = read_all<double>((dir + "/t_nga.f64").c_str());
    auto init = read_all<double>((dir + "/init.f64").c_str()); // x y theta
    auto obs = read_all<float>((dir + "/obs.f32").c_str());    // stride 4
    auto gnd = read_all<float>((dir + "/gnd.f32").c_str());

    using namespace slam_amd;
    // icpTools.cpp:168-176
    Matrix rot(2, 2), trans(2, 1);
    trans.val[0][0] = init[0];
    trans.val[1][0] = init[1];
    rot.val[0][0] = std::cos(init[2]);
    rot.val[0][1] = -std::sin(init[2]);
    rot.val[1][0] = std::sin(init[2]);
    rot.val[1][1] = std::cos(init[2]);
    // icpTools.cpp:187-188
    IcpPointToPoint icp(m_ga.data(), m_nga.data(), (int32_t)m_ga.size() / 2, (int32_t)m_nga.size() / 2, (int32_t)2);
    icp.fit(t_ga.data(), t_nga.data(), (int32_t)t_ga.size() / 2, (int32_t)t_nga.size() / 2, rot, trans, 5, 0);
    // icpTools.cpp:195-197
    const double res[4] = {trans.val[0][0], trans.val[1][0], std::atan2(rot.val[1][0], rot.val[0][0]),
                           (double)icp.getNumberCorrespondences()};

    // the point-to-line matcher (icpPointToPlane.h:26-49, stale upstream) in both constructor shapes: libicp's one cloud, and
    // this fork's two arrays -- the same model (GA then NGA), so the same answer
    std::vector<double> m_all(m_ga), t_all(t_ga);
    m_all.insert(m_all.end(), m_nga.begin(), m_nga.end());
    t_all.insert(t_all.end(), t_nga.begin(), t_nga.end());
    Matrix rot_l(2, 2), trans_l(2, 1), rot_l2(2, 2), trans_l2(2, 1);
    for (Matrix *r : {&rot_l, &rot_l2}) {
        r->val[0][0] = std::cos(init[2]), r->val[0][1] = -std::sin(init[2]);
        r->val[1][0] = std::sin(init[2]), r->val[1][1] = std::cos(init[2]);
    }
    trans_l.val[0][0] = trans_l2.val[0][0] = init[0];
    trans_l.val[1][0] = trans_l2.val[1][0] = init[1];
    IcpPointToPlane icp_l(m_all.data(), (int32_t)m_all.size() / 2, (int32_t)2);
    icp_l.fit(t_all.data(), (int32_t)t_all.size() / 2, rot_l, trans_l, -1);
    IcpPointToPlane icp_l2(m_ga.data(), m_nga.data(), (int32_t)m_ga.size() / 2, (int32_t)m_nga.size() / 2, (int32_t)2, (int32_t)10);
    icp_l2.fit(t_ga.data(), t_nga.data(), (int32_t)t_ga.size() / 2, (int32_t)t_nga.size() / 2, rot_l2, trans_l2, 5, 0);
    const bool same_l = trans_l.val[0][0] == trans_l2.val[0][0] && trans_l.val[1][0] == trans_l2.val[1][0] && rot_l.val[1][0] == rot_l2.val[1][0];
    const double res_l[5] = {trans_l.val[0][0], trans_l.val[1][0], std::atan2(rot_l.val[1][0], rot_l.val[0][0]),
                             (double)icp_l.getNumberCorrespondences(), same_l ? 1.0 : 0.0};

    // too few model points: logs, object unusable, fit leaves R,t alone (icp.cpp:38-43)
    double few[6] = {0, 0, 1, 1, 2, 2};
    IcpPointToPoint bad(few, few, 2, 1, 2);
    Matrix R2 = Matrix::eye(2), t2(2, 1);
    bad.fit(t_ga.data(), t_nga.data(), (int32_t)t_ga.size() / 2, (int32_t)t_nga.size() / 2, R2, t2, 5, 0);
    const bool untouched = !bad.valid() && R2.val[0][0] == 1 && R2.val[0][1] == 0 && t2.val[0][0] == 0;

    // local_mapper.cpp:29,86: MLS local_map(200,200,0.2,true); setMinClusterPoints(20)
    MLS local_map(200, 200, 0.2, true);
    local_map.setMinClusterPoints(20);
    for (int k = 0; k < 3; ++k) local_map.addToMap(obs.data(), (int)obs.size() / 4, gnd.data(), (int)gnd.size() / 4, 0.0, 0.0, 4);
    const OccupancyGrid &g = local_map.getDrivability();

    FILE *f = std::fopen(out.c_str(), "wb");
    std::fwrite(res, sizeof res, 1, f);
    const double flag = untouched ? 1.0 : 0.0;
    std::fwrite(&flag, sizeof flag, 1, f);
    const double meta[4] = {g.info.resolution, (double)g.info.width, g.info.origin_x, g.info.origin_y};
    std::fwrite(meta, sizeof meta, 1, f);
    std::fwrite(g.data.data(), 1, g.data.size(), f);
    std::fclose(f);
    f = std::fopen((out + ".p2l").c_str(), "wb");
    std::fwrite(res_l, sizeof res_l, 1, f);
    std::fclose(f);
    return 0;
}
