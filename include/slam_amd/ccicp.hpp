// slam_amd/ccicp.hpp -- header-only adapter with the shape of class CCICP
// (ccicp2d/include/ccicp2d/icpTools.h:28-87), the facade scan_registration calls
// (scan_registration.cpp:57,82-83,97-98,139,159), over the C-ABI (slam_mi355x.h):
//
//   reference                                       here (clouds as float arrays, `stride` floats per point)
//   CCICP(RegistrationType)          icpTools.cpp:14-33     CCICP(RegistrationType)
//   setTargetCloud(cloud, pose)      :585-608               setTargetCloud(xyz, n, stride, pose)
//   setTargetGndCloud(cloud)         :580-583               setTargetGndCloud(xyz, n, stride)
//   setSceneCloud(cloud)             :611-634               setSceneCloud(xyz, n, stride)
//   doICPMatch(initPose)             :222-298               doICPMatch(initPose)
//   doICPMatch(target, scene, pose)  :571-578               doICPMatch(target..., scene..., initPose)
//   getResidual()                    :637-641 (returns -1)  getResidual()
//
// PCL and tf types are replaced by plain arrays and the Pose struct below (the pose part of
// geometry_msgs::PoseStamped); the tf calls (getYaw, getEulerYPR, createQuaternionFromRPY: icpTools.cpp:174,205-212)
// are restated here as tf defines them.
//
// What runs per scan (scan_registration.cpp:139-159: setSceneCloud, doICPMatch) is ONE device-resident chain on one
// stream: the cloud goes up once, slam_ccicp_scene_dev (segmentGround + classifyPoints + voxel filter + class split
// with the cap) -> slam_icp_fit_batch_dev -> slam_ccicp_height_rpy_pose_dev run without the host learning a single count,
// and one 128-byte block comes back (pose, result, height, counts).  The target's index is built when the target
// changes -- setTargetCloud, or a crop window (icpTools.cpp:225-239) that selects other points than the last one did --
// not once per match as the reference constructs its matcher (icpTools.cpp:187).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <utility>
#include <vector>

#include "slam_mi355x.h"

namespace slam_amd {

enum RegistrationType { SCAN_TO_SCAN = 0, SCAN_TO_MAP = 1 }; // icpTools.h

#ifndef SLAM_AMD_POSE_DEFINED
#define SLAM_AMD_POSE_DEFINED
struct Pose { // geometry_msgs::Pose
    double x = 0, y = 0, z = 0;
    double qx = 0, qy = 0, qz = 0, qw = 1;
};
#endif

namespace detail {
// tf::Matrix3x3(q).getEulerYPR(yaw, pitch, roll, 1)
inline void euler_ypr(const Pose &p, double &yaw, double &pitch, double &roll)
{
    const double d = p.qx * p.qx + p.qy * p.qy + p.qz * p.qz + p.qw * p.qw, s = 2.0 / d;
    const double xs = p.qx * s, ys = p.qy * s, zs = p.qz * s, wx = p.qw * xs, wy = p.qw * ys, wz = p.qw * zs,
                 xx = p.qx * xs, xy = p.qx * ys, xz = p.qx * zs, yy = p.qy * ys, yz = p.qy * zs, zz = p.qz * zs;
    const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy),
                 m01 = xy - wz, m02 = xz + wy;
    if (std::fabs(m20) >= 1) { // gimbal lock branch of tf
        yaw = 0;
        const double delta = std::atan2(m01, m02);
        if (m20 < 0) {
            pitch = M_PI / 2.0;
            roll = delta;
        } else {
            pitch = -M_PI / 2.0;
            roll = delta;
        }
    } else {
        pitch = -std::asin(m20);
        roll = std::atan2(m21 / std::cos(pitch), m22 / std::cos(pitch));
        yaw = std::atan2(m10 / std::cos(pitch), m00 / std::cos(pitch));
    }
}
// tf::createQuaternionFromRPY
inline void quat_from_rpy(double roll, double pitch, double yaw, Pose &p)
{
    const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr),
                 sr = std::sin(hr);
    p.qx = sr * cp * cy - cr * sp * sy;
    p.qy = cr * sp * cy + sr * cp * sy;
    p.qz = cr * cp * sy - sr * sp * cy;
    p.qw = cr * cp * cy + sr * sp * sy;
}
} // namespace detail

class CCICP {
public:
    static constexpr int ICP_MAX_PTS = 20000; // icpTools.h:21

    explicit CCICP(RegistrationType type_ = SCAN_TO_SCAN) : type(type_)
    {
        if (slam_gseg_create(nullptr, &gseg_) != SLAM_OK || slam_ccicp_create(&cc_) != SLAM_OK)
            std::fprintf(stderr, "CCICP: %s\n", slam_last_error());
        ok(slam_stream_create(&stream_));
        ok(slam_malloc((void **)&d_ga_, 16 * (size_t)ICP_MAX_PTS));
        ok(slam_malloc((void **)&d_nga_, 16 * (size_t)ICP_MAX_PTS));
        ok(slam_malloc((void **)&d_scene_pts_, 16 * 2 * (size_t)ICP_MAX_PTS));
        ok(slam_malloc((void **)&d_io_, kIoBytes));
        ok(slam_host_alloc((void **)&h_io_, kIoBytes));
        if (h_io_) std::memset(h_io_, 0, kIoBytes);
        if (d_io_) ok(slam_memset(d_io_, 0, kIoBytes, stream_));
        reset_box();
    }
    ~CCICP()
    {
        slam_device_synchronize();
        if (icp_) slam_icp_destroy(icp_);
        for (Cloud *c : {&raw_, &scene_raw_, &scene_in_, &labels_, &obs_, &flags_, &seg_target_, &seg_scene_, &ground_target_, &ground_scene_, &scene_ground_})
            slam_free(c->p);
        free_ahead();
        free_seq();
        slam_free(d_ga_);
        slam_free(d_nga_);
        slam_free(d_scene_pts_);
        slam_free(d_io_);
        slam_host_free(h_io_);
        slam_ccicp_destroy(cc_);
        slam_gseg_destroy(gseg_);
        slam_stream_destroy(stream_);
    }
    CCICP(const CCICP &) = delete;
    CCICP &operator=(const CCICP &) = delete;

    // icpTools.cpp:585-608.  SCAN_TO_MAP: the target is an obstacle cloud already (the global map):
    // classifyPoints only.  SCAN_TO_SCAN: segmentGround first; its ground cloud becomes ground_target.
    // (Called when the map changes, not per scan: it may wait for the device.)
    void setTargetCloud(const float *xyz, int n, int stride, const Pose & /*initPose*/)
    {
        extent_of(xyz, n, stride);
        if (type == SCAN_TO_MAP) {
            upload(raw_, xyz, n, stride);
            select(0xffu, obs_, obs_n_); // copyPointCloud(*target, *seg_target) (:592)
            classify_into(seg_target_, seg_target_n_, false);
        } else {
            // segmentGround + classifyPoints in bin order (no voxel filter) as one chain; the ground cloud is the new ground_target
            const bool had_scene = scene_ready_; // a scene set before this call stays the scene (:585-608 touch neither seg_scene nor ground_scene)
            upload(raw_, xyz, n, stride);
            reserve(seg_target_, 16 * (size_t)(n + 1));
            reserve(ground_target_, 16 * (size_t)(n + 1));
            seg_target_n_ = 0;
            set_ground_target_count(0);
            if (n > 0) {
                ok(slam_ccicp_scene_dev(cc_, gseg_, (const float *)raw_.p, n, stride, 0, 0, 0.0, 0.0, 0.0, ICP_MAX_PTS, d_scene_pts_, io_scan(),
                                        (float *)ground_target_.p, io_counts(), stream_));
                ok(slam_ccicp_scene_cloud_dev(cc_, (float *)seg_target_.p, n, stream_));
                fetch_io();
                seg_target_n_ = h_counts()[2];
                set_ground_target_count(h_counts()[1]);
            }
            scene_ready_ = false; // (the chain's scene outputs were borrowed ...
            if (had_scene) enqueue_scene_chain(); // ... and are made again from the scene cloud, which is still held: setSceneCloud,
                                                  // setTargetCloud, doICPMatch in this order matches that scene, as upstream)
        }
        target_dirty_ = true;
        reset_box();
    }
    void setTargetGndCloud(const float *xyz, int n, int stride) // :580-583
    {
        upload(raw_, xyz, n, stride);
        int n_out = 0;
        select(0xffu, ground_target_, n_out); // copyPointCloud
        set_ground_target_count(n_out);
    }
    // :611-634: segmentGround, classifyPoints, VoxelGrid 0.5 x 0.5 x 2 -- enqueued, nothing waited for
    void setSceneCloud(const float *xyz, int n, int stride)
    {
        if (ahead_.pending && ahead_.xyz == xyz && ahead_.n == n && ahead_.stride == stride) {
            adopt_ahead();
            return;
        }
        ahead_.pending = ahead_.deferred = false; // (a prepared cloud that is not the one set now is dropped; one not yet enqueued costs nothing)
        upload(scene_raw_, xyz, n, stride);
        scene_n_in_ = n;
        scene_stride_ = stride;
        enqueue_scene_chain();
    }
    // ... of the cloud turned by R (row-major) and offset by t first -- what scan_registration does to every cloud before it hands
    // it over (roll / pitch compensation, scan_registration.cpp:128-139: pcl::transformPointCloud) -- with the transform on the
    // device: (float)(r0 x + r1 y + r2 z + t) per coordinate in double, the floats a host loop gives (slam_grid_transform_cloud_dev)
    void setSceneCloud(const float *xyz, int n, int stride, const double R[9], const double t[3])
    {
        ahead_.pending = ahead_.deferred = false;
        upload(scene_in_, xyz, n, stride);
        reserve(scene_raw_, sizeof(float) * 3 * (size_t)(n + 1));
        ok(slam_grid_transform_cloud_dev((const float *)scene_in_.p, n, stride, R, t, (float *)scene_raw_.p, stream_));
        scene_n_in_ = n;
        scene_stride_ = 3;
        enqueue_scene_chain();
    }
    // the scene's chain from the cloud in scene_raw_ (setSceneCloud; again after a SCAN_TO_SCAN setTargetCloud borrowed its outputs)
    void enqueue_scene_chain()
    {
        reserve(scene_ground_, 16 * (size_t)(scene_n_in_ + 1));
        ok(slam_ccicp_scene_dev(cc_, gseg_, (const float *)scene_raw_.p, scene_n_in_, scene_stride_, 1, 0, 0.0, 0.0, 0.0, ICP_MAX_PTS, d_scene_pts_,
                                io_scan(), (float *)scene_ground_.p, io_counts(), stream_));
        scene_ready_ = true;
        scene_known_ = false;
        seg_scene_valid_ = ground_scene_valid_ = false;
    }

    Pose doICPMatch(const float *target, int n_target, const float *scene, int n_scene, int stride, const Pose &initPose)
    {
        setTargetCloud(target, n_target, stride, initPose); // :571-578
        setSceneCloud(scene, n_scene, stride);
        return doICPMatch(initPose);
    }

    Pose doICPMatch(const Pose &initPose) // :222-298
    {
        ensure_target(initPose); // crop +-75 m around the pose (:225-239), split with the cap (:263-276), index
        Pose   result;
        double yaw0, pitch0, roll0;
        detail::euler_ypr(initPose, yaw0, pitch0, roll0); // tf::getYaw (:174)
        double *hp = reinterpret_cast<double *>(h_io_);
        hp[0] = std::cos(yaw0), hp[1] = -std::sin(yaw0), hp[2] = std::sin(yaw0), hp[3] = std::cos(yaw0); // :168-176
        hp[4] = initPose.x, hp[5] = initPose.y;
        std::memset(h_io_ + kOffRes, 0, 32); // result and height: a scan below 5 points leaves them untouched
        hp[kOffZ / 8] = initPose.z;
        // The match's launches: the fit (two: the spread form and the form that redoes what it could not finish), the height (two).
        // Nothing is copied around them (round 6: a match is bound by its launches, 45 in round 5): the initial pose is read from
        // this block where it lies (pinned memory is the device's to read), and the last kernel leaves the device block's 128 bytes
        // here again.
        const bool in_place = !scene_ready_ || !icp_;
        if (!scene_ready_) { // no scene cloud: an empty one
            std::memset(h_io_ + kOffScan, 0, 32);
            ok(slam_memcpy_h2d_async(d_io_, h_io_, kOffNgt, stream_));
        } else if (in_place) {
            ok(slam_memcpy_h2d_async(d_io_, h_io_, kOffScan, stream_)); // (no target: the height is that of the initial pose)
        }
        // IcpPointToPoint icp(refPts...) + icp.fit(...) (:187-188): the scene's size stays on the device
        if (icp_) {
            if (in_place)
                ok(slam_icp_fit_batch_dev(icp_, d_scene_pts_, io_scan(), io_scan() + 2, 1, io_R(), io_t(), 5.0,
                                          reinterpret_cast<slam_icp_result *>(d_io_ + kOffRes), nullptr, stream_));
            else
                ok(slam_icp_fit_batch_from_dev(icp_, d_scene_pts_, io_scan(), io_scan() + 2, 1, hp, hp + 4, io_R(), io_t(), 5.0,
                                               reinterpret_cast<slam_icp_result *>(d_io_ + kOffRes), nullptr, stream_));
        }
        // doHeightInterpolate(ground_target, result_2d) (:295, :301-381) for the pose the fit left on the device
        ok(slam_ccicp_height_rpy_pose_mirror_dev(cc_, (const float *)ground_target_.p, reinterpret_cast<const int32_t *>(d_io_ + kOffNgt),
                                                 ground_target_n_, 4, io_R(), io_t(), initPose.z, roll0, pitch0,
                                                 reinterpret_cast<double *>(d_io_ + kOffZ), h_io_, d_io_, kIoBytes, stream_));
        if (ahead_.deferred) enqueue_ahead(ahead_.xyz, ahead_.n, ahead_.stride); // the NEXT cloud's chain, on its own stream, behind this match's launches
        ok(slam_stream_synchronize(stream_)); // the match's one wait: its result block is here
        scene_known_ = scene_ready_;
        const int32_t *scan = h_scan();
        n_scene_[0] = scan[2], n_scene_[1] = scan[1] - scan[2];
        if (scene_ready_ && h_counts()[3] != 0) return match_stepwise(initPose, yaw0, pitch0, roll0); // lattice beyond the chain's accumulator
        if (scan[1] < 5) { // :179-184
            std::fprintf(stderr, "ERROR: Total Scene has %d points\n", scan[1]);
            num_corr_ = 0;
            result = Pose();
            result.qw = 9999;
            return result;
        }
        return result_from_io(initPose, pitch0, roll0);
    }

    // ---- Throughput forms (round 5; the reference is one cloud at a time, scan_registration.cpp:109-173, and so are the calls above).
    //
    // (a) Two chains in flight.  prepareSceneCloud(cloud k+1) while cloud k is being matched: the upload and the scene chain
    // (segmentGround + classifyPoints + voxel filter + split) of the NEXT cloud run on a second stream with buffers of their own
    // -- they do not depend on the match in flight (only on roll / pitch / z of the incoming pose, which the caller has applied,
    // scan_registration.cpp:128-138).  The next setSceneCloud with the same pointer and size adopts what was prepared instead of
    // making it again; any other cloud is made from scratch as before.  The pointer must stay valid until then (pinned memory,
    // slam_host_alloc, makes the upload asynchronous).
    void prepareSceneCloud(const float *xyz, int n, int stride)
    {
        if (!ahead_.made && !make_ahead()) return;
        Ahead &a = ahead_;
        if (scene_ready_ && icp_) {
            // A match is about to be asked for (round 6): its launches go first.  Enqueuing this cloud's upload and nine launches
            // takes the host 0.06 ms, during which the match's stream would have nothing to run; doICPMatch enqueues them behind
            // its own launches, before its wait (0.33 -> 0.27 ms per match).  A setSceneCloud with this pointer before any match
            // makes the cloud the ordinary way.
            a.xyz = xyz, a.n = n, a.stride = stride;
            a.deferred = true;
            a.pending = false;
            return;
        }
        enqueue_ahead(xyz, n, stride);
    }
    void enqueue_ahead(const float *xyz, int n, int stride)
    {
        Ahead &a = ahead_;
        a.deferred = false;
        reserve_on(a.raw, sizeof(float) * (size_t)(n + 1) * stride, a.stream);
        reserve_on(a.ground, 16 * (size_t)(n + 1), a.stream);
        if (n > 0) ok(slam_memcpy_h2d_async(a.raw.p, xyz, sizeof(float) * (size_t)n * stride, a.stream));
        ok(slam_ccicp_scene_dev(a.cc, a.gseg, (const float *)a.raw.p, n, stride, 1, 0, 0.0, 0.0, 0.0, ICP_MAX_PTS, a.d_pts,
                                reinterpret_cast<int32_t *>(a.d_io + kOffScan), (float *)a.ground.p,
                                reinterpret_cast<int32_t *>(a.d_io + kOffCounts), a.stream));
        ok(slam_event_record(a.done, a.stream));
        a.xyz = xyz, a.n = n, a.stride = stride;
        a.pending = true;
    }

    // (b) A sequence whose initial poses are known beforehand (an offline log, a re-run against a new map): every scene's chain on
    // one of kSeqLanes streams (a hipGraph replay from a scene slot's third use on), the fits of up to kSeqBatch scenes as ONE slam_icp_fit_batch_dev, the heights on the lanes again, one
    // read-back per batch.  Against the target as the calls above left it (setTargetCloud); the crop window of every pose applies as
    // in doICPMatch (:225-239) -- a window that changes what the target's index is built from ends a batch.  Same poses as
    // setSceneCloud + doICPMatch one by one (tests/test_gpu_cpp_adapters.py).  A scene with fewer than 5 points returns
    // orientation.w == 9999 in its place (:179-184).
#ifndef SLAM_CCICP_SEQ_LANES
#define SLAM_CCICP_SEQ_LANES 4 // (8: 0.21 ms per match in three runs of four and 0.39 in the fourth, the host held inside the graph launches; 4: 0.23-0.25 every run; 2: 0.26 -- tools/exp/c3_lanes.sh)
#endif
    static constexpr int kSeqLanes = SLAM_CCICP_SEQ_LANES, kSeqBatch = 16;
    std::vector<Pose> matchSequence(const float *const *scenes, const int *n_points, int count, int stride, const Pose *init)
    {
        std::vector<Pose> out((size_t)std::max(count, 0));
        if (count <= 0 || !make_seq()) return out;
        int k0 = 0;
        while (k0 < count) {
            ensure_target(init[k0]);
            int k1 = k0 + 1;
            while (k1 < count && k1 - k0 < kSeqBatch && target_stays(init[k1])) ++k1; // (target_stays narrows the crop box as doICPMatch would)
            match_batch(scenes + k0, n_points + k0, k1 - k0, stride, init + k0, out.data() + k0);
            k0 = k1;
        }
        return out;
    }
    int sequenceBatches() const { return seq_batches_; }
    // the scene chains of matchSequence replayed as hipGraphs from a scene slot's third use on.  Off since round 6: with nine launches a
    // chain (thirty in round 5) the replay saves the host 0.2 ms per batch that the device-bound batch does not wait for, and the
    // uploads beside graph launches take three times as long -- 0.144 against 0.133 ms per match (tools/exp/c3_graphs_ab.sh)
    void setSequenceGraphs(bool graphs) { use_graphs_ = graphs; }
    // host clock of the batches so far, ms: scene chains enqueued | everything enqueued | results back (cumulative within a batch)
    const double *sequenceTimes() const { return seq_ms_; }
    double sequenceUploadMs() const { return seq_up_ms_; } // of which: the host inside the scenes' upload calls

    double getResidual() const { return -1; } // :637-641 ("TODO: calculate this somehow")
    // :644-650: copies of seg_target, seg_scene, ground_target, ground_scene as x, y, z per point
    void getSegmentedClouds(std::vector<float> &target, std::vector<float> &scene, std::vector<float> &g_target,
                            std::vector<float> &g_scene)
    {
        copy_out(seg_target_, seg_target_n_, target, box_); // doICPMatch has filtered seg_target in place (:226-239)
        materialise_scene();
        copy_out(seg_scene_, seg_scene_n_, scene, nullptr);
        copy_out(ground_target_, ground_target_n_, g_target, nullptr);
        copy_out(ground_scene_, ground_scene_n_, g_scene, nullptr);
    }
    int getNumberCorrespondences() const { return num_corr_; }
    // sizes of what getSegmentedClouds would copy out (:644-650)
    int targetSize() { return target_dirty_ ? seg_target_n_ : target_in_box_; }
    int sceneSize()
    {
        know_scene();
        return h_counts()[2];
    }
    int groundTargetSize() const { return ground_target_n_; }
    int groundSceneSize()
    {
        materialise_scene();
        return ground_scene_n_;
    }
    const int *modelCounts() const { return n_model_; }
    const int *sceneCounts()
    {
        know_scene();
        return n_scene_;
    }
    int lastIterations() const { return last_iters_; }
    int stepwiseMatches() const { return stepwise_matches_; } // matches redone through the stepwise entry points (lattice beyond the chain's)
    int targetBuilds() const { return target_builds_; } // how often the target's index was built (it is kept across matches)

    RegistrationType type;

private:
    struct Cloud {
        void  *p = nullptr;
        size_t cap = 0;
    };
    // one device block and its pinned mirror: [R t | result | z, neighbours | scan {0, n, n_ga, -} | counts {obs, gnd, flt, err} | n ground target]
    static constexpr size_t kOffRes = 48, kOffZ = 64, kOffScan = 80, kOffCounts = 96, kOffNgt = 112, kIoBytes = 128;
    double  *io_R() { return reinterpret_cast<double *>(d_io_); }
    double  *io_t() { return reinterpret_cast<double *>(d_io_) + 4; }
    int32_t *io_scan() { return reinterpret_cast<int32_t *>(d_io_ + kOffScan); }
    int32_t *io_counts() { return reinterpret_cast<int32_t *>(d_io_ + kOffCounts); }
    const int32_t *h_scan() const { return reinterpret_cast<const int32_t *>(h_io_ + kOffScan); }
    const int32_t *h_counts() const { return reinterpret_cast<const int32_t *>(h_io_ + kOffCounts); }
    void fetch_io()
    {
        ok(slam_memcpy_d2h_async(h_io_, d_io_, kIoBytes, stream_));
        ok(slam_stream_synchronize(stream_));
    }
    void know_scene()
    {
        if (scene_ready_ && !scene_known_) {
            fetch_io();
            scene_known_ = true;
            n_scene_[0] = h_scan()[2], n_scene_[1] = h_scan()[1] - h_scan()[2];
        }
    }
    static void ok(int rc)
    {
        if (rc != SLAM_OK) std::fprintf(stderr, "CCICP: %s\n", slam_last_error());
    }
    static void reserve(Cloud &c, size_t bytes)
    {
        if (bytes <= c.cap) return;
        slam_device_synchronize(); // (the old block may be in use by enqueued work)
        slam_free(c.p);
        c.p = nullptr;
        c.cap = 0;
        const size_t want = bytes + bytes / 4; // clouds of a sequence differ by a few per cent: grow rarely
        if (slam_malloc(&c.p, want) == SLAM_OK) c.cap = want;
    }
    void upload(Cloud &dst, const float *xyz, int n, int stride)
    {
        reserve(dst, sizeof(float) * (size_t)(n + 1) * stride);
        if (&dst == &raw_) raw_n_ = n, raw_stride_ = stride;
        if (n <= 0) return;
        // from pinned memory (slam_host_alloc) the copy only enqueues -- the caller keeps the cloud as it is until the match that uses
        // it has returned, as with prepareSceneCloud --; from pageable memory the runtime stages it and the call waits (0.045 ms
        // for a 131 072-point cloud)
        if (slam_host_is_pinned(xyz))
            ok(slam_memcpy_h2d_async(dst.p, xyz, sizeof(float) * (size_t)n * stride, stream_));
        else
            ok(slam_memcpy_h2d(dst.p, xyz, sizeof(float) * (size_t)n * stride, stream_));
    }
    void set_ground_target_count(int n)
    {
        ground_target_n_ = n;
        *reinterpret_cast<int32_t *>(h_io_ + kOffNgt) = n;
        ok(slam_memcpy_h2d(d_io_ + kOffNgt, h_io_ + kOffNgt, 4, stream_));
    }
    // extent of the finite points of the cloud the target is made of (a superset of seg_target's): what a crop window must
    // cover to select everything
    void extent_of(const float *xyz, int n, int stride)
    {
        // (without branches and library calls: 131 072 points in 0.3 ms where isfinite / fmin / fmax behind a `continue` took 1.5 --
        // a third of a target update)
        float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
        for (int i = 0; i < n; ++i) {
            const float *p = xyz + (size_t)i * stride;
            const float  x = p[0], y = p[1], z = p[2];
            const bool   fin = (x - x == 0.0f) & (y - y == 0.0f) & (z - z == 0.0f);
            const float  xa = fin ? x : INFINITY, xb = fin ? x : -INFINITY, ya = fin ? y : INFINITY, yb = fin ? y : -INFINITY;
            x0 = xa < x0 ? xa : x0, x1 = xb > x1 ? xb : x1;
            y0 = ya < y0 ? ya : y0, y1 = yb > y1 ? yb : y1;
        }
        ext_[0] = x0, ext_[1] = x1, ext_[2] = y0, ext_[3] = y1;
    }
    void reset_box() { box_[0] = box_[2] = -INFINITY, box_[1] = box_[3] = INFINITY; }
    bool covers(const float b[4]) const { return ext_[0] >= b[0] && ext_[1] <= b[1] && ext_[2] >= b[2] && ext_[3] <= b[3]; }
    // The model of this match: seg_target inside the intersection of every crop window since setTargetCloud (the reference
    // filters seg_target in place, :226-239), split by class with the cap (:263-276).  The index is rebuilt only when that
    // selects other points than it was built from.
    // the crop box after this pose's window (the intersection so far, :226-239) and whether the index built last still holds exactly
    // the points it selects
    bool next_box(const Pose &initPose, float nb[4]) const
    {
        const double crop_dist = 75;
        const float  win[4] = {(float)(-crop_dist + initPose.x), (float)(crop_dist + initPose.x), (float)(-crop_dist + initPose.y),
                               (float)(crop_dist + initPose.y)}; // setFilterLimits takes floats (:231,:236)
        nb[0] = std::fmax(box_[0], win[0]), nb[1] = std::fmin(box_[1], win[1]), nb[2] = std::fmax(box_[2], win[2]), nb[3] = std::fmin(box_[3], win[3]);
        return !target_dirty_ && ((covers(nb) && covers(built_box_)) ||
                                  (nb[0] == built_box_[0] && nb[1] == built_box_[1] && nb[2] == built_box_[2] && nb[3] == built_box_[3]));
    }
    // matchSequence: true = this pose's match runs against the index as it is (and the box is narrowed as its doICPMatch would);
    // false = it needs another index: nothing changed, the batch ends before it
    bool target_stays(const Pose &initPose)
    {
        float nb[4];
        if (!next_box(initPose, nb)) return false;
        for (int k = 0; k < 4; ++k) box_[k] = nb[k];
        return true;
    }
    void ensure_target(const Pose &initPose)
    {
        float      nb[4];
        const bool same = next_box(initPose, nb);
        for (int k = 0; k < 4; ++k) box_[k] = nb[k];
        if (same) return;
        if (icp_) {
            slam_icp_destroy(icp_);
            icp_ = nullptr;
        }
        int mc[2] = {0, 0}, tot[2] = {0, 0};
        ok(slam_ccicp_split_box_dev(cc_, (const float *)seg_target_.p, seg_target_n_, 4, box_, ICP_MAX_PTS, d_ga_, d_nga_, mc, tot, stream_));
        n_model_[0] = mc[0], n_model_[1] = mc[1];
        target_in_box_ = tot[0] + tot[1];
        // IcpPointToPoint icp(refPts_GA, refPts_NGA, ...) (:187): fewer than 5 model points -> error, R,t stay (icp.cpp:38-43)
        if (slam_icp_create_dev(d_ga_, mc[0], d_nga_, mc[1], nullptr, &icp_) != SLAM_OK) {
            std::fprintf(stderr, "%s\n", slam_last_error());
            icp_ = nullptr;
        }
        for (int k = 0; k < 4; ++k) built_box_[k] = box_[k];
        target_dirty_ = false;
        ++target_builds_;
    }
    Pose result_from_io(const Pose &initPose, double pitch0, double roll0)
    {
        const double          *hp = reinterpret_cast<const double *>(h_io_);
        const slam_icp_result *res = reinterpret_cast<const slam_icp_result *>(h_io_ + kOffRes);
        num_corr_ = icp_ ? res->n_corr : 0;
        last_iters_ = icp_ ? res->iters : 0;
        Pose         result;
        const double corr_yaw = std::atan2(hp[2], hp[0]); // :197
        result.x = hp[4];
        result.y = hp[5];
        detail::quat_from_rpy(roll0, pitch0, corr_yaw, result); // :205-212
        result.z = hp[kOffZ / 8];
        (void)initPose;
        return result;
    }
    // the points of raw_ whose label is in `mask`, in cloud order, as (x, y, z, 0) records; 0xff = every point
    void select(unsigned mask, Cloud &dst, int &n_dst)
    {
        reserve(dst, 16 * (size_t)(raw_n_ + 1));
        n_dst = 0;
        if (raw_n_ == 0) return;
        if (mask == 0xffu) {
            reserve(labels_, (size_t)raw_n_ + 16);
            ok(slam_memset(labels_.p, 0, (size_t)raw_n_, stream_));
            mask = 1u;
        }
        ok(slam_ccicp_select_dev(cc_, (const float *)raw_.p, raw_n_, raw_stride_, (const uint8_t *)labels_.p, mask,
                                 (float *)dst.p, &n_dst, stream_));
    }
    // classifyPoints over obs_ (:36-103) into x,y,z,ground_adj records: through the voxel filter
    // (setSceneCloud) or in classifyPoints' own bin order (setTargetCloud)
    void classify_into(Cloud &dst, int &n_dst, bool voxel)
    {
        reserve(flags_, (size_t)obs_n_ + 16);
        reserve(dst, 16 * (size_t)(obs_n_ + 1));
        n_dst = 0;
        if (obs_n_ == 0) return;
        ok(slam_gseg_classify_ga_dev(gseg_, (const float *)obs_.p, obs_n_, 4, (uint8_t *)flags_.p, stream_));
        if (voxel)
            ok(slam_ccicp_voxel_downsample_dev(cc_, (const float *)obs_.p, (const uint8_t *)flags_.p, obs_n_, 4, 0.5f, 0.5f,
                                               2.0f, (float *)dst.p, obs_n_, &n_dst, stream_));
        else
            ok(slam_ccicp_bin_order_dev(cc_, (const float *)obs_.p, (const uint8_t *)flags_.p, obs_n_, 4, (float *)dst.p,
                                        &n_dst, stream_));
    }
    // seg_scene and ground_scene as clouds (getSegmentedClouds): not needed by a match, made when asked for
    void materialise_scene()
    {
        if (!scene_ready_) {
            seg_scene_n_ = ground_scene_n_ = 0;
            return;
        }
        know_scene();
        if (!seg_scene_valid_) {
            seg_scene_n_ = h_counts()[2];
            reserve(seg_scene_, 16 * (size_t)(scene_n_in_ + 1));
            ok(slam_ccicp_scene_cloud_dev(cc_, (float *)seg_scene_.p, scene_n_in_, stream_));
            seg_scene_valid_ = true;
        }
        if (!ground_scene_valid_) { // VoxelGrid 0.5 x 0.5 x 5 of the scene's ground cloud (:628-633)
            const int n_gnd = h_counts()[1];
            reserve(ground_scene_, 16 * (size_t)(n_gnd + 1));
            ground_scene_n_ = 0;
            if (n_gnd > 0)
                ok(slam_ccicp_voxel_downsample_dev(cc_, (const float *)scene_ground_.p, nullptr, n_gnd, 4, 0.5f, 0.5f, 5.0f,
                                                   (float *)ground_scene_.p, n_gnd, &ground_scene_n_, stream_));
            ground_scene_valid_ = true;
        }
    }
    // The scene through the stepwise entry points, which take voxel lattices of any extent (the chain's accumulator holds
    // 2 M voxels: d_counts[3] said that this cloud's lattice does not fit)
    Pose match_stepwise(const Pose &initPose, double yaw0, double pitch0, double roll0)
    {
        ++stepwise_matches_;
        raw_n_ = scene_n_in_, raw_stride_ = scene_stride_;
        std::swap(raw_, scene_raw_);
        reserve(labels_, (size_t)raw_n_ + 16);
        int n_gnd = 0, sc[2] = {0, 0};
        ok(slam_gseg_segment_dev(gseg_, (const float *)raw_.p, raw_n_, raw_stride_, (uint8_t *)labels_.p, stream_));
        select((1u << SLAM_GSEG_OBSTACLE) | (1u << SLAM_GSEG_OVERHEAD), obs_, obs_n_);
        select(1u << SLAM_GSEG_GROUND, scene_ground_, n_gnd);
        std::swap(raw_, scene_raw_);
        classify_into(seg_scene_, seg_scene_n_, true);
        seg_scene_valid_ = true;
        ok(slam_ccicp_split_box_dev(cc_, (const float *)seg_scene_.p, seg_scene_n_, 4, nullptr, ICP_MAX_PTS, d_scene_pts_, d_nga_, sc, nullptr, stream_));
        if (sc[1]) ok(slam_memcpy_d2d(d_scene_pts_ + 2 * (size_t)sc[0], d_nga_, 16 * (size_t)sc[1], stream_)); // NGA behind GA
        int32_t *hs = reinterpret_cast<int32_t *>(h_io_ + kOffScan), *hc = reinterpret_cast<int32_t *>(h_io_ + kOffCounts);
        hs[0] = 0, hs[1] = sc[0] + sc[1], hs[2] = sc[0], hs[3] = 0;
        hc[0] = obs_n_, hc[1] = n_gnd, hc[2] = seg_scene_n_, hc[3] = 0;
        n_scene_[0] = sc[0], n_scene_[1] = sc[1];
        double *hp = reinterpret_cast<double *>(h_io_);
        hp[0] = std::cos(yaw0), hp[1] = -std::sin(yaw0), hp[2] = std::sin(yaw0), hp[3] = std::cos(yaw0);
        hp[4] = initPose.x, hp[5] = initPose.y;
        std::memset(h_io_ + kOffRes, 0, 32);
        hp[kOffZ / 8] = initPose.z;
        ok(slam_memcpy_h2d_async(d_io_, h_io_, kOffNgt, stream_));
        if (hs[1] >= 5 && icp_)
            ok(slam_icp_fit_batch_dev(icp_, d_scene_pts_, io_scan(), io_scan() + 2, 1, io_R(), io_t(), 5.0,
                                      reinterpret_cast<slam_icp_result *>(d_io_ + kOffRes), nullptr, stream_));
        ok(slam_ccicp_height_rpy_pose_dev(cc_, (const float *)ground_target_.p, reinterpret_cast<const int32_t *>(d_io_ + kOffNgt),
                                          ground_target_n_, 4, io_R(), io_t(), initPose.z, roll0, pitch0,
                                          reinterpret_cast<double *>(d_io_ + kOffZ), stream_));
        fetch_io();
        if (hs[1] < 5) {
            std::fprintf(stderr, "ERROR: Total Scene has %d points\n", hs[1]);
            Pose bad;
            bad.qw = 9999;
            num_corr_ = 0;
            return bad;
        }
        return result_from_io(initPose, pitch0, roll0);
    }
    // n records (x, y, z, .) as x, y, z triples; box (optional): only the points a PassThrough of these limits keeps
    void copy_out(const Cloud &c, int n, std::vector<float> &xyz, const float *box)
    {
        std::vector<float> rec(4 * (size_t)n + 4);
        if (n > 0) ok(slam_memcpy_d2h(rec.data(), c.p, 16 * (size_t)n, stream_));
        xyz.clear();
        xyz.reserve(3 * (size_t)n);
        for (int i = 0; i < n; ++i) {
            const float *p = &rec[4 * (size_t)i];
            if (box && !(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]) && p[0] >= box[0] && p[0] <= box[1] &&
                         p[1] >= box[2] && p[1] <= box[3]))
                continue;
            xyz.insert(xyz.end(), p, p + 3);
        }
    }

    // ---- (a) the scene prepared ahead
    struct Ahead {
        bool          made = false, pending = false, deferred = false; // deferred: asked for, enqueued by the next doICPMatch behind its own launches
        slam_gseg_t  *gseg = nullptr;
        slam_ccicp_t *cc = nullptr;
        slam_stream_t stream = nullptr;
        slam_event_t  done = nullptr;
        Cloud         raw, ground;
        double       *d_pts = nullptr;
        unsigned char *d_io = nullptr;
        const float  *xyz = nullptr;
        int           n = 0, stride = 3;
    } ahead_;
    static void reserve_on(Cloud &c, size_t bytes, slam_stream_t st)
    {
        if (bytes <= c.cap) return;
        slam_stream_synchronize(st); // (the old block may be in use by work enqueued on its stream)
        slam_free(c.p);
        c.p = nullptr;
        c.cap = 0;
        const size_t want = bytes + bytes / 4;
        if (slam_malloc(&c.p, want) == SLAM_OK) c.cap = want;
    }
    bool make_ahead()
    {
        Ahead &a = ahead_;
        bool   good = slam_gseg_create(nullptr, &a.gseg) == SLAM_OK && slam_ccicp_create(&a.cc) == SLAM_OK && slam_stream_create(&a.stream) == SLAM_OK &&
                    slam_event_create(&a.done) == SLAM_OK && slam_malloc((void **)&a.d_pts, 16 * 2 * (size_t)ICP_MAX_PTS) == SLAM_OK &&
                    slam_malloc((void **)&a.d_io, kIoBytes) == SLAM_OK;
        if (good) good = slam_memset(a.d_io, 0, kIoBytes, a.stream) == SLAM_OK;
        if (!good) std::fprintf(stderr, "CCICP: %s\n", slam_last_error());
        a.made = good;
        return good;
    }
    void free_ahead()
    {
        Ahead &a = ahead_;
        if (a.cc) slam_ccicp_destroy(a.cc);
        if (a.gseg) slam_gseg_destroy(a.gseg);
        slam_free(a.raw.p);
        slam_free(a.ground.p);
        slam_free(a.d_pts);
        slam_free(a.d_io);
        if (a.done) slam_event_destroy(a.done);
        if (a.stream) slam_stream_destroy(a.stream);
    }
    // the prepared scene becomes THE scene: its buffers, handles and chain outputs change places with the current ones (the match's
    // stream waits for the chain; nothing is copied)
    void adopt_ahead()
    {
        Ahead &a = ahead_;
        a.pending = false;
        ok(slam_stream_wait_event(stream_, a.done));
        std::swap(scene_raw_, a.raw);
        std::swap(scene_ground_, a.ground);
        std::swap(d_scene_pts_, a.d_pts);
        std::swap(gseg_, a.gseg); // (the chain's scratch -- what slam_ccicp_scene_cloud_dev reads -- lives in the handles)
        std::swap(cc_, a.cc);
        // the scan block and the counts of the chain: into the match's io block, device to device (32 bytes)
        ok(slam_memcpy_d2d(d_io_ + kOffScan, a.d_io + kOffScan, kOffNgt - kOffScan, stream_));
        // ... and the second stream waits for whatever the match's stream still does with the buffers it gets back -- this copy included:
        // the next prepared chain writes a.d_io
        ok(slam_event_record(a.done, stream_));
        ok(slam_stream_wait_event(a.stream, a.done));
        scene_n_in_ = a.n;
        scene_stride_ = a.stride;
        scene_ready_ = true;
        scene_known_ = false;
        seg_scene_valid_ = ground_scene_valid_ = false;
    }

    // ---- (b) the sequence form
    struct SeqLane {
        slam_gseg_t  *gseg = nullptr;
        slam_ccicp_t *cc = nullptr;
        slam_stream_t stream = nullptr;
        Cloud         raw, ground;
        int           max_n = 0, epoch = 0; // the largest scene its handles have seen; bumped when they may have re-allocated their scratch
    };
    struct SeqIo { // per scene, device and pinned mirror: [scan {0, n, n_ga, -} | counts {obs, gnd, flt, err}]
        int32_t scan[4], counts[4];
    };
    struct SeqSlot { // what the chain of scene k of a batch was last enqueued with, and its captured replay
        slam_graph_t graph = nullptr;
        int          n = -1, stride = 0, epoch = -1;
        const void  *raw = nullptr, *ground = nullptr;
    };
    SeqLane       lane_[kSeqLanes];
    SeqSlot       seq_slot_[kSeqBatch];
    bool          seq_made_ = false, use_graphs_ = false;
    int           seq_batches_ = 0;
    double        seq_ms_[3] = {0, 0, 0}, seq_up_ms_ = 0;
    double       *d_seq_slots_ = nullptr, *d_seq_pack_ = nullptr; // [kSeqBatch][2 * ICP_MAX_PTS] points each
    SeqIo        *d_seq_io_ = nullptr, *h_seq_io_ = nullptr;
    double       *d_seq_pose_ = nullptr, *h_seq_pose_ = nullptr;  // [kSeqBatch][4] R, [kSeqBatch][2] t, [kSeqBatch][2] z + neighbours
    slam_icp_result *d_seq_res_ = nullptr, *h_seq_res_ = nullptr;
    int32_t      *d_seq_off_ = nullptr;                            // [kSeqBatch + 1] scan_off, [kSeqBatch] scan_nga
    slam_event_t  seq_ev_[kSeqBatch] = {}, seq_fit_ = nullptr, seq_lane_ev_[kSeqLanes] = {};
    static constexpr size_t kSlotPts = 2 * (size_t)ICP_MAX_PTS, kPoseDoubles = 8 * (size_t)kSeqBatch;
    bool make_seq()
    {
        if (seq_made_) return true;
        bool good = true;
        auto g = [&](int rc) { good = good && rc == SLAM_OK; };
        for (SeqLane &l : lane_) {
            g(slam_gseg_create(nullptr, &l.gseg));
            g(slam_ccicp_create(&l.cc));
            // (ordinary streams, which the runtime deals over its four shared hardware queues.  A queue of its own per lane,
            // slam_stream_create_reserving_cus(.., 0), was measured: 1.85 -> 3.3-3.6 ms per batch of ten, and one run never came back)
            g(slam_stream_create(&l.stream));
        }
        g(slam_malloc((void **)&d_seq_slots_, 16 * kSlotPts * kSeqBatch));
        g(slam_malloc((void **)&d_seq_pack_, 16 * kSlotPts * kSeqBatch));
        g(slam_malloc((void **)&d_seq_io_, sizeof(SeqIo) * kSeqBatch));
        g(slam_host_alloc((void **)&h_seq_io_, sizeof(SeqIo) * kSeqBatch));
        g(slam_malloc((void **)&d_seq_pose_, 8 * kPoseDoubles));
        g(slam_host_alloc((void **)&h_seq_pose_, 8 * kPoseDoubles));
        g(slam_malloc((void **)&d_seq_res_, sizeof(slam_icp_result) * kSeqBatch));
        g(slam_host_alloc((void **)&h_seq_res_, sizeof(slam_icp_result) * kSeqBatch));
        g(slam_malloc((void **)&d_seq_off_, 4 * (2 * (size_t)kSeqBatch + 1)));
        for (slam_event_t &e : seq_ev_) g(slam_event_create(&e));
        for (slam_event_t &e : seq_lane_ev_) g(slam_event_create(&e));
        g(slam_event_create(&seq_fit_));
        if (good) g(slam_memset(d_seq_io_, 0, sizeof(SeqIo) * kSeqBatch, stream_));
        if (!good) std::fprintf(stderr, "CCICP: %s\n", slam_last_error());
        seq_made_ = good;
        return good;
    }
    void free_seq()
    {
        for (SeqSlot &q : seq_slot_)
            if (q.graph) slam_graph_destroy(q.graph);
        for (SeqLane &l : lane_) {
            if (l.cc) slam_ccicp_destroy(l.cc);
            if (l.gseg) slam_gseg_destroy(l.gseg);
            slam_free(l.raw.p);
            slam_free(l.ground.p);
            if (l.stream) slam_stream_destroy(l.stream);
        }
        slam_free(d_seq_slots_);
        slam_free(d_seq_pack_);
        slam_free(d_seq_io_);
        slam_host_free(h_seq_io_);
        slam_free(d_seq_pose_);
        slam_host_free(h_seq_pose_);
        slam_free(d_seq_res_);
        slam_host_free(h_seq_res_);
        slam_free(d_seq_off_);
        for (slam_event_t e : seq_ev_)
            if (e) slam_event_destroy(e);
        for (slam_event_t e : seq_lane_ev_)
            if (e) slam_event_destroy(e);
        if (seq_fit_) slam_event_destroy(seq_fit_);
    }
    // n <= kSeqBatch scenes against the index as ensure_target / target_stays left it
    void match_batch(const float *const *scenes, const int *n_points, int n, int stride, const Pose *init, Pose *out)
    {
        ++seq_batches_;
        ahead_.pending = ahead_.deferred = false;
        const auto t_begin = std::chrono::steady_clock::now();
        auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
        double *hR = h_seq_pose_, *ht = h_seq_pose_ + 4 * kSeqBatch, *hz = h_seq_pose_ + 6 * kSeqBatch;
        double *dR = d_seq_pose_, *dt = d_seq_pose_ + 4 * kSeqBatch, *dz = d_seq_pose_ + 6 * kSeqBatch;
        std::vector<double> yaw0((size_t)n), pitch0((size_t)n), roll0((size_t)n);
        const double       *pts[kSeqBatch];
        const int32_t      *scan[kSeqBatch];
        // every scene's chain on its lane (lanes in turn: a lane's stream orders the uploads into its buffers behind the chain before)
        auto enqueue_scene = [&](int k) {
            SeqLane &l = lane_[k % kSeqLanes];
            const int np = n_points[k];
            reserve_on(l.raw, sizeof(float) * (size_t)(np + 1) * stride, l.stream);
            reserve_on(l.ground, 16 * (size_t)(np + 1), l.stream);
            const auto t_up = std::chrono::steady_clock::now();
            if (np > 0) ok(slam_memcpy_h2d_async(l.raw.p, scenes[k], sizeof(float) * (size_t)np * stride, l.stream));
            seq_up_ms_ += since(t_up);
            double *slot = d_seq_slots_ + 2 * kSlotPts * (size_t)k;
            // The chain is some thirty short launches: enqueued one by one they cost the host 0.08 ms per scene; replayed as a hipGraph
            // 0.01 (measured, tools/exp/c3_batch_time.sh: the batch itself is bound by the device either way).  A scene of the same size as the one this slot held before (a lidar's clouds
            // are) replays the chain as a hipGraph captured on its second use: same kernels, same arguments, one launch.
            SeqSlot &q = seq_slot_[k];
            // (a captured chain holds the pointers of the lane's buffers AND of its handles' scratch, which grows with the largest scene the
            // lane has seen: a larger scene on the lane -- through any of its slots -- ends every replay made before it)
            if (np > l.max_n) l.max_n = np, ++l.epoch;
            const bool same = q.n == np && q.stride == stride && q.raw == l.raw.p && q.ground == l.ground.p && q.epoch == l.epoch;
            auto chain = [&]() {
                ok(slam_ccicp_scene_dev(l.cc, l.gseg, (const float *)l.raw.p, np, stride, 1, 0, 0.0, 0.0, 0.0, ICP_MAX_PTS, slot, d_seq_io_[k].scan,
                                        (float *)l.ground.p, d_seq_io_[k].counts, l.stream));
            };
            if (same && q.graph) {
                ok(slam_graph_launch(q.graph, l.stream));
            } else if (same && np > 0 && use_graphs_) { // second use: every scratch buffer of the chain exists, nothing in it allocates
                if (q.graph) slam_graph_destroy(q.graph);
                q.graph = nullptr;
                if (slam_graph_begin_capture(l.stream) == SLAM_OK) {
                    chain();
                    if (slam_graph_end_capture(l.stream, &q.graph) != SLAM_OK) q.graph = nullptr;
                }
                if (q.graph)
                    ok(slam_graph_launch(q.graph, l.stream));
                else {
                    use_graphs_ = false; // (a runtime that cannot capture this chain: call by call from here on)
                    chain();
                }
            } else {
                if (q.graph) slam_graph_destroy(q.graph);
                q.graph = nullptr;
                chain();
                q.n = np, q.stride = stride, q.raw = l.raw.p, q.ground = l.ground.p, q.epoch = l.epoch;
            }
            ok(slam_event_record(seq_ev_[k], l.stream));
        };
        for (int k = 0; k < n; ++k) enqueue_scene(k);
        seq_ms_[0] += since(t_begin);
        for (int k = 0; k < n; ++k) {
            double *slot = d_seq_slots_ + 2 * kSlotPts * (size_t)k;
            pts[k] = slot;
            scan[k] = d_seq_io_[k].scan;
            detail::euler_ypr(init[k], yaw0[k], pitch0[k], roll0[k]); // tf::getYaw (:174)
            hR[4 * k + 0] = std::cos(yaw0[k]), hR[4 * k + 1] = -std::sin(yaw0[k]), hR[4 * k + 2] = std::sin(yaw0[k]), hR[4 * k + 3] = std::cos(yaw0[k]);
            ht[2 * k + 0] = init[k].x, ht[2 * k + 1] = init[k].y;
            hz[2 * k + 0] = init[k].z, hz[2 * k + 1] = 0;
            std::memset(&h_seq_res_[k], 0, sizeof(slam_icp_result));
        }
        // the fits as one batch on the match's stream
        ok(slam_memcpy_h2d_async(d_seq_pose_, h_seq_pose_, 8 * kPoseDoubles, stream_));
        ok(slam_memcpy_h2d_async(d_seq_res_, h_seq_res_, sizeof(slam_icp_result) * (size_t)n, stream_));
        for (int k = 0; k < n; ++k) ok(slam_stream_wait_event(stream_, seq_ev_[k]));
        int32_t *d_off = d_seq_off_, *d_nga = d_seq_off_ + kSeqBatch + 1;
        ok(slam_ccicp_pack_scans_dev(n, pts, scan, d_seq_pack_, d_off, d_nga, stream_));
        if (icp_) ok(slam_icp_fit_batch_dev(icp_, d_seq_pack_, d_off, d_nga, n, dR, dt, 5.0, d_seq_res_, nullptr, stream_));
        ok(slam_event_record(seq_fit_, stream_));
        // doHeightInterpolate (:295) of every pose on the lanes again (the handles' scratch is per lane)
        for (int k = 0; k < n; ++k) {
            SeqLane &l = lane_[k % kSeqLanes];
            if (k < kSeqLanes) ok(slam_stream_wait_event(l.stream, seq_fit_));
            ok(slam_ccicp_height_rpy_pose_dev(l.cc, (const float *)ground_target_.p, reinterpret_cast<const int32_t *>(d_io_ + kOffNgt),
                                              ground_target_n_, 4, dR + 4 * k, dt + 2 * k, init[k].z, roll0[k], pitch0[k], dz + 2 * k, l.stream));
        }
        for (int j = 0; j < std::min(n, (int)kSeqLanes); ++j) {
            ok(slam_event_record(seq_lane_ev_[j], lane_[j].stream));
            ok(slam_stream_wait_event(stream_, seq_lane_ev_[j]));
        }
        ok(slam_memcpy_d2h_async(h_seq_pose_, d_seq_pose_, 8 * kPoseDoubles, stream_));
        ok(slam_memcpy_d2h_async(h_seq_res_, d_seq_res_, sizeof(slam_icp_result) * (size_t)n, stream_));
        ok(slam_memcpy_d2h_async(h_seq_io_, d_seq_io_, sizeof(SeqIo) * (size_t)n, stream_));
        seq_ms_[1] += since(t_begin);
        ok(slam_stream_synchronize(stream_));
        seq_ms_[2] += since(t_begin);
        for (int k = 0; k < n; ++k) {
            const SeqIo &io = h_seq_io_[k];
            if (io.counts[3] != 0) { // the voxel lattice did not fit the chain's accumulator: this scene through the stepwise entry points
                setSceneCloud(scenes[k], n_points[k], stride);
                out[k] = doICPMatch(init[k]);
                continue;
            }
            Pose r;
            if (io.scan[1] < 5) { // :179-184
                std::fprintf(stderr, "ERROR: Total Scene has %d points\n", io.scan[1]);
                r.qw = 9999;
                out[k] = r;
                continue;
            }
            num_corr_ = icp_ ? h_seq_res_[k].n_corr : 0;
            last_iters_ = icp_ ? h_seq_res_[k].iters : 0;
            r.x = ht[2 * k + 0];
            r.y = ht[2 * k + 1];
            detail::quat_from_rpy(roll0[k], pitch0[k], std::atan2(hR[4 * k + 2], hR[4 * k + 0]), r); // :197, :205-212
            r.z = hz[2 * k + 0];
            out[k] = r;
        }
        scene_ready_ = false; // (the scenes of a batch lived in the lanes' buffers)
    }

    slam_gseg_t  *gseg_ = nullptr;
    slam_ccicp_t *cc_ = nullptr;
    slam_icp_t   *icp_ = nullptr;   // the target's index, kept across matches
    slam_stream_t stream_ = nullptr;
    Cloud         raw_, scene_raw_, scene_in_ /* a cloud as it came, before setSceneCloud's transform */, labels_, obs_, flags_, seg_target_, seg_scene_, ground_target_, ground_scene_, scene_ground_;
    int           raw_n_ = 0, raw_stride_ = 3, obs_n_ = 0, seg_target_n_ = 0, seg_scene_n_ = 0, ground_target_n_ = 0,
        ground_scene_n_ = 0, num_corr_ = 0, last_iters_ = 0, scene_n_in_ = 0, scene_stride_ = 3, target_in_box_ = 0, target_builds_ = 0, stepwise_matches_ = 0;
    int     n_model_[2] = {0, 0}, n_scene_[2] = {0, 0};
    bool    target_dirty_ = true, scene_ready_ = false, scene_known_ = false, seg_scene_valid_ = false, ground_scene_valid_ = false;
    float   ext_[4] = {0, 0, 0, 0};       // x_lo, x_hi, y_lo, y_hi of the target's finite points
    float   box_[4], built_box_[4] = {0, 0, 0, 0}; // the crop so far (intersection of the windows); the one the index was built for
    double *d_ga_ = nullptr, *d_nga_ = nullptr, *d_scene_pts_ = nullptr;
    unsigned char *d_io_ = nullptr, *h_io_ = nullptr;
};

} // namespace slam_amd
