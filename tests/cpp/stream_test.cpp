// BASELINE config 5 driver: a stream of scans goes through slam_amd::StreamMapper
// (copy / register / map on three HIP streams, rolling window) and, for
// comparison, through the same C-ABI calls one after another on one stream.
// Both must produce identical poses and counts; the pipelined result is written
// out for the Python test to check against the oracle.
//   stream_test <dir> <out> <chunk_scans> <size> <resolution> [repeat]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "slam_amd/stream_mapper.hpp"

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
static void ok(int rc, const char *what)
{
    if (rc != SLAM_OK) { std::fprintf(stderr, "%s: %s\n", what, slam_last_error()); std::exit(3); }
}
template <class T>
static T *pinned(size_t n)
{
    void *p = nullptr;
    ok(slam_host_alloc(&p, n * sizeof(T) + 16), "host_alloc");
    return static_cast<T *>(p);
}
static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const std::string dir = argv[1], out = argv[2];
    const int    chunk = std::atoi(argv[3]), size = std::atoi(argv[4]);
    const double res = std::atof(argv[5]);
    const int    repeat = argc > 6 ? std::atoi(argv[6]) : 1;
    auto m_ga = read_all<double>(dir + "/m_ga.f64"), m_nga = read_all<double>(dir + "/m_nga.f64");
    auto pts = read_all<double>(dir + "/pts.f64");
    auto off = read_all<int32_t>(dir + "/scan_off.i32"), nga = read_all<int32_t>(dir + "/scan_nga.i32");
    auto R0 = read_all<double>(dir + "/R0.f64"), t0 = read_all<double>(dir + "/t0.f64");
    const int n_scans = (int)nga.size();

    slam_icp_params ip;
    slam_icp_default_params(&ip);
    slam_icp_t *icp = nullptr;
    ok(slam_icp_create(m_ga.data(), (int)m_ga.size() / 2, m_nga.data(), (int)m_nga.size() / 2, &ip, &icp), "icp_create");
    slam_grid_params gp;
    slam_grid_default_params(&gp);
    gp.rolling = 1;
    gp.max_range = 0.45 * size * res;
    slam_grid_t *grid[2] = {nullptr, nullptr};
    for (auto &g : grid) ok(slam_grid_create(size, size, res, &gp, &g), "grid_create");

    // chunks in pinned memory, offsets rebased per chunk
    struct HostChunk { slam_amd::ScanChunk c; int first; };
    std::vector<HostChunk> chunks;
    int max_pts = 0;
    for (int s0 = 0; s0 < n_scans; s0 += chunk) {
        const int ns = std::min(chunk, n_scans - s0), p0 = off[s0], np = off[s0 + ns] - p0;
        double  *hp = pinned<double>(2 * (size_t)np), *hR = pinned<double>(4 * (size_t)ns), *ht = pinned<double>(2 * (size_t)ns);
        int32_t *ho = pinned<int32_t>(ns + 1), *hn = pinned<int32_t>(ns);
        std::memcpy(hp, pts.data() + 2 * (size_t)p0, 16 * (size_t)np);
        std::memcpy(hR, R0.data() + 4 * (size_t)s0, 32 * (size_t)ns);
        std::memcpy(ht, t0.data() + 2 * (size_t)s0, 16 * (size_t)ns);
        for (int i = 0; i <= ns; ++i) ho[i] = off[s0 + i] - p0;
        std::memcpy(hn, nga.data() + s0, 4 * (size_t)ns);
        chunks.push_back({{hp, ho, hn, hR, ht, ns, np, ht[0], ht[1]}, s0});
        max_pts = std::max(max_pts, np);
    }
    std::vector<double> R_pipe(4 * (size_t)n_scans), t_pipe(2 * (size_t)n_scans), R_seq(R_pipe.size()), t_seq(t_pipe.size());
    double sec_pipe = 0, sec_seq = 0;

    for (int rep = 0; rep < repeat; ++rep) {
        if (rep > 0) // fresh windows centred on (0,0) for every repeat
            for (auto &g : grid) { slam_grid_destroy(g); ok(slam_grid_create(size, size, res, &gp, &g), "grid_create"); }
        ok(slam_device_synchronize(), "sync");
        // ---- pipelined
        {
            slam_amd::StreamMapper sm(icp, grid[0], chunk, max_pts);
            const double a = now();
            int slot_of[2] = {-1, -1};
            for (size_t k = 0; k < chunks.size(); ++k) {
                const int s = (int)(k & 1);
                if (slot_of[s] >= 0) {
                    const int f = chunks[(size_t)slot_of[s]].first;
                    sm.wait_slot(s, R_pipe.data() + 4 * (size_t)f, t_pipe.data() + 2 * (size_t)f);
                }
                if (sm.push(chunks[k].c) != s) return 4;
                slot_of[s] = (int)k;
            }
            for (int s = 0; s < 2; ++s)
                if (slot_of[s] >= 0) {
                    const int f = chunks[(size_t)slot_of[s]].first;
                    sm.wait_slot(s, R_pipe.data() + 4 * (size_t)f, t_pipe.data() + 2 * (size_t)f);
                }
            sm.finish();
            sec_pipe = now() - a;
        }
        // ---- one stage after another, one stream, host sync after every chunk
        double *d_pts, *d_R, *d_t; int32_t *d_off, *d_nga;
        ok(slam_malloc((void **)&d_pts, 16 * (size_t)max_pts), "malloc");
        ok(slam_malloc((void **)&d_R, 32 * (size_t)chunk), "malloc");
        ok(slam_malloc((void **)&d_t, 16 * (size_t)chunk), "malloc");
        ok(slam_malloc((void **)&d_off, 4 * (size_t)(chunk + 1)), "malloc");
        ok(slam_malloc((void **)&d_nga, 4 * (size_t)chunk), "malloc");
        ok(slam_device_synchronize(), "sync");
        const double a = now();
        for (auto &hc : chunks) {
            const auto &c = hc.c;
            ok(slam_memcpy_h2d(d_pts, c.pts, 16 * (size_t)c.n_points, nullptr), "h2d");
            ok(slam_memcpy_h2d(d_off, c.scan_off, 4 * (size_t)(c.n_scans + 1), nullptr), "h2d");
            ok(slam_memcpy_h2d(d_nga, c.scan_nga, 4 * (size_t)c.n_scans, nullptr), "h2d");
            ok(slam_memcpy_h2d(d_R, c.R0, 32 * (size_t)c.n_scans, nullptr), "h2d");
            ok(slam_memcpy_h2d(d_t, c.t0, 16 * (size_t)c.n_scans, nullptr), "h2d");
            ok(slam_icp_fit_batch_dev(icp, d_pts, d_off, d_nga, c.n_scans, d_R, d_t, 5.0, nullptr, nullptr, nullptr), "icp");
            ok(slam_grid_set_pose(grid[1], c.window_x, c.window_y, nullptr), "roll");
            ok(slam_grid_raycast_scans_dev(grid[1], d_pts, d_off, c.n_scans, c.n_points, d_R, d_t, nullptr), "raycast");
            ok(slam_memcpy_d2h(R_seq.data() + 4 * (size_t)hc.first, d_R, 32 * (size_t)c.n_scans, nullptr), "d2h");
            ok(slam_memcpy_d2h(t_seq.data() + 2 * (size_t)hc.first, d_t, 16 * (size_t)c.n_scans, nullptr), "d2h");
        }
        ok(slam_grid_finalize(grid[1], nullptr), "finalize");
        ok(slam_device_synchronize(), "sync");
        sec_seq = now() - a;
        slam_free(d_pts); slam_free(d_R); slam_free(d_t); slam_free(d_off); slam_free(d_nga);
    }

    const size_t cells = (size_t)size * size;
    std::vector<int32_t> h[2], m[2];
    std::vector<int8_t>  occ[2];
    double pose[2][2];
    for (int i = 0; i < 2; ++i) {
        h[i].resize(cells); m[i].resize(cells); occ[i].resize(cells);
        ok(slam_grid_read_counts(grid[i], h[i].data(), m[i].data()), "read_counts");
        ok(slam_grid_read_occupancy(grid[i], occ[i].data()), "read_occ");
        ok(slam_grid_get_pose(grid[i], &pose[i][0], &pose[i][1]), "get_pose");
    }
    const bool same = h[0] == h[1] && m[0] == m[1] && occ[0] == occ[1] && R_pipe == R_seq && t_pipe == t_seq &&
                      pose[0][0] == pose[1][0] && pose[0][1] == pose[1][1];
    long long total = 0;
    for (size_t i = 0; i < cells; ++i) total += h[0][i] + m[0][i];
    std::printf("{\"scans\": %d, \"points\": %d, \"chunk_scans\": %d, \"pipelined_ms\": %.3f, \"sequential_ms\": %.3f, "
                "\"points_per_s_pipelined\": %.4g, \"cell_updates\": %lld, \"identical\": %s}\n",
                n_scans, off[n_scans], chunk, sec_pipe * 1e3, sec_seq * 1e3, off[n_scans] / sec_pipe, total,
                same ? "true" : "false");
    FILE *f = std::fopen(out.c_str(), "wb");
    if (!f) return 2;
    std::fwrite(pose[0], 8, 2, f);
    std::fwrite(R_pipe.data(), 8, R_pipe.size(), f);
    std::fwrite(t_pipe.data(), 8, t_pipe.size(), f);
    std::fwrite(h[0].data(), 4, cells, f);
    std::fwrite(m[0].data(), 4, cells, f);
    std::fwrite(occ[0].data(), 1, cells, f);
    std::fclose(f);
    for (auto &hc : chunks) {
        slam_host_free((void *)hc.c.pts); slam_host_free((void *)hc.c.R0); slam_host_free((void *)hc.c.t0);
        slam_host_free((void *)hc.c.scan_off); slam_host_free((void *)hc.c.scan_nga);
    }
    slam_grid_destroy(grid[0]); slam_grid_destroy(grid[1]);
    slam_icp_destroy(icp);
    return same ? 0 : 5;
}
