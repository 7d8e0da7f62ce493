// slam_amd/stream_mapper.hpp -- host-side pipeline for a stream of scans
// (BASELINE config 5): chunks of scans are copied to the GPU, registered and
// ray-cast into a rolling local map on three HIP streams, so the copy of chunk
// k+1, the ICP of chunk k and the grid update of chunk k-1 overlap.  Plain C++
// over the C-ABI (slam_mi355x.h); one StreamMapper per GPU / host thread.
//
// Data flow per chunk (double-buffered device slots, events between stages):
//   copy stream : pinned host chunk -> HBM (points, offsets, class counts, initial poses)
//   icp  stream : wait(copied)  -> slam_icp_fit_batch_dev            -> record(registered)
//   grid stream : wait(registered) -> slam_grid_set_pose (roll) + slam_grid_raycast_scans_dev
// The caller owns the pinned chunk buffers until wait_slot() returns for that slot.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include "slam_mi355x.h"

namespace slam_amd {

struct ScanChunk {              // all pointers are PINNED host memory (slam_host_alloc)
    const double  *pts;         // xy f64 of all scans of the chunk
    const int32_t *scan_off;    // n_scans + 1
    const int32_t *scan_nga;    // n_scans
    const double  *R0, *t0;     // initial poses: n_scans x 4, n_scans x 2
    int            n_scans, n_points;
    double         window_x, window_y; // where the rolling map is centred for this chunk (e.g. the EKF pose)
};

class StreamMapper {
public:
    StreamMapper(slam_icp_t *icp, slam_grid_t *grid, int max_scans, int max_points, double indist = 5.0)
        : icp_(icp), grid_(grid), max_scans_(max_scans), max_points_(max_points), indist_(indist)
    {
        ok(slam_stream_create(&copy_), "stream");
        ok(slam_stream_create(&icp_s_), "stream");
        ok(slam_stream_create(&grid_s_), "stream");
        for (int s = 0; s < 2; ++s) {
            Slot &b = slot_[s];
            ok(slam_malloc((void **)&b.pts, 16 * (size_t)max_points), "malloc");
            ok(slam_malloc((void **)&b.off, 4 * (size_t)(max_scans + 1)), "malloc");
            ok(slam_malloc((void **)&b.nga, 4 * (size_t)max_scans), "malloc");
            ok(slam_malloc((void **)&b.R, 32 * (size_t)max_scans), "malloc");
            ok(slam_malloc((void **)&b.t, 16 * (size_t)max_scans), "malloc");
            ok(slam_event_create(&b.copied), "event");
            ok(slam_event_create(&b.registered), "event");
            ok(slam_event_create(&b.mapped), "event");
        }
    }
    ~StreamMapper()
    {
        slam_device_synchronize();
        for (int s = 0; s < 2; ++s) {
            Slot &b = slot_[s];
            slam_free(b.pts); slam_free(b.off); slam_free(b.nga); slam_free(b.R); slam_free(b.t);
            slam_event_destroy(b.copied); slam_event_destroy(b.registered); slam_event_destroy(b.mapped);
        }
        slam_stream_destroy(copy_); slam_stream_destroy(icp_s_); slam_stream_destroy(grid_s_);
    }
    StreamMapper(const StreamMapper &) = delete;
    StreamMapper &operator=(const StreamMapper &) = delete;

    // Enqueues one chunk and returns at once; returns the slot (0/1) it used.  The slot's previous
    // chunk must have been waited for (wait_slot) before its pinned buffers are reused by the caller.
    int push(const ScanChunk &c)
    {
        if (c.n_scans > max_scans_ || c.n_points > max_points_) throw std::runtime_error("chunk exceeds the reservation");
        const int s = next_;
        next_ ^= 1;
        Slot &b = slot_[s];
        if (b.busy) ok(slam_event_synchronize(b.mapped), "wait");   // device slot still in use by chunk k-2
        ok(slam_memcpy_h2d_async(b.pts, c.pts, 16 * (size_t)c.n_points, copy_), "h2d");
        ok(slam_memcpy_h2d_async(b.off, c.scan_off, 4 * (size_t)(c.n_scans + 1), copy_), "h2d");
        ok(slam_memcpy_h2d_async(b.nga, c.scan_nga, 4 * (size_t)c.n_scans, copy_), "h2d");
        ok(slam_memcpy_h2d_async(b.R, c.R0, 32 * (size_t)c.n_scans, copy_), "h2d");
        ok(slam_memcpy_h2d_async(b.t, c.t0, 16 * (size_t)c.n_scans, copy_), "h2d");
        ok(slam_event_record(b.copied, copy_), "record");
        ok(slam_stream_wait_event(icp_s_, b.copied), "wait");
        ok(slam_icp_fit_batch_dev(icp_, b.pts, b.off, b.nga, c.n_scans, b.R, b.t, indist_, nullptr, nullptr, icp_s_), "icp");
        ok(slam_event_record(b.registered, icp_s_), "record");
        ok(slam_stream_wait_event(grid_s_, b.registered), "wait");
        ok(slam_grid_set_pose(grid_, c.window_x, c.window_y, grid_s_), "roll");   // MLS::setPose, mls.cpp:408-479
        ok(slam_grid_raycast_scans_dev(grid_, b.pts, b.off, c.n_scans, c.n_points, b.R, b.t, grid_s_), "raycast");
        ok(slam_event_record(b.mapped, grid_s_), "record");
        b.busy = true;
        b.n_scans = c.n_scans;
        return s;
    }
    // Blocks until the chunk last pushed into `slot` is registered and mapped; copies its poses out.
    void wait_slot(int slot, double *R_out, double *t_out)
    {
        Slot &b = slot_[slot];
        if (!b.busy) return;
        ok(slam_event_synchronize(b.mapped), "wait");
        if (R_out) ok(slam_memcpy_d2h(R_out, b.R, 32 * (size_t)b.n_scans, nullptr), "d2h");
        if (t_out) ok(slam_memcpy_d2h(t_out, b.t, 16 * (size_t)b.n_scans, nullptr), "d2h");
        b.busy = false;
    }
    void finish() { ok(slam_device_synchronize(), "sync"); ok(slam_grid_finalize(grid_, grid_s_), "finalize"); ok(slam_stream_synchronize(grid_s_), "sync"); }

private:
    struct Slot {
        double *pts = nullptr, *R = nullptr, *t = nullptr;
        int32_t *off = nullptr, *nga = nullptr;
        slam_event_t copied = nullptr, registered = nullptr, mapped = nullptr;
        bool busy = false;
        int  n_scans = 0;
    };
    static void ok(int rc, const char *what)
    {
        if (rc != SLAM_OK) throw std::runtime_error(std::string(what) + ": " + slam_last_error());
    }
    slam_icp_t  *icp_;
    slam_grid_t *grid_;
    int          max_scans_, max_points_;
    double       indist_;
    slam_stream_t copy_ = nullptr, icp_s_ = nullptr, grid_s_ = nullptr;
    Slot         slot_[2];
    int          next_ = 0;
};

} // namespace slam_amd
